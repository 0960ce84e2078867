// TEST INFRASTRUCTURE ONLY. Never linked into the product library.
//
// Deterministic harness around the *real* reference renderer. This translation
// unit pulls the reference's main.cpp in by path (nothing is copied into this
// repository), renames its main(), and adds a command-line front end that
//   * re-seeds the reference's global RNG per pixel / per sample, so that two
//     runs (and our own implementations) can be compared value for value,
//   * dumps the loaded scene graph, meshes and BVHs as flat arrays,
//   * dumps known-answer vectors for the individual hot-path functions.
//
// Built only where /root/reference exists (see oracle/Makefile, target `ref`),
// output goes to oracle/_ref/ which is git-ignored.
//
// With -DREF_P12 the same file is compiled against the project-12 snapshot of the
// reference (project12.zip, unpacked to a temporary directory by `make ref12`): the
// snapshot has the Whitted-style shader instead of the path tracer, so the path
// commands are left out and `render_whitted` is added.

#include <chrono>
#include <ostream>
// libstdc++ 11 has no operator<< for std::chrono::duration; the reference
// (written against a newer libstdc++) prints one at the end of its main().
template <class R, class P>
std::ostream &operator<<(std::ostream &os, std::chrono::duration<R, P> const &d)
{
    return os << d.count();
}

#define main reference_main
#include "main.cpp" // resolved through -I/root/reference/src
#undef main

#include "objects.h"
#include "texture.h"
#include "lodepng.h"
#include "materials.h" // the snapshot's main.cpp does not pull it in

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <map>

// -----------------------------------------------------------------------------
// raw array dump: <dir>/<name>.bin + a line in <dir>/manifest.txt
// -----------------------------------------------------------------------------
static std::string g_outdir;

static void dump(const char *name, const char *dtype, const void *data, size_t elemSize,
                 std::vector<size_t> shape)
{
    size_t n = 1;
    for (size_t s : shape) n *= s;
    std::string path = g_outdir + "/" + name + ".bin";
    FILE *fp = fopen(path.c_str(), "wb");
    if (!fp) { fprintf(stderr, "cannot write %s\n", path.c_str()); exit(2); }
    if (n) fwrite(data, elemSize, n, fp);
    fclose(fp);
    FILE *mf = fopen((g_outdir + "/manifest.txt").c_str(), "a");
    fprintf(mf, "%s %s", name, dtype);
    for (size_t s : shape) fprintf(mf, " %zu", s);
    fprintf(mf, "\n");
    fclose(mf);
}
static void dumpF(const char *n, const std::vector<float> &v, std::vector<size_t> s) { dump(n, "f4", v.data(), 4, s); }
static void dumpI(const char *n, const std::vector<int32_t> &v, std::vector<size_t> s) { dump(n, "i4", v.data(), 4, s); }
static void dumpU(const char *n, const std::vector<uint32_t> &v, std::vector<size_t> s) { dump(n, "u4", v.data(), 4, s); }
static void dumpB(const char *n, const std::vector<uint8_t> &v, std::vector<size_t> s) { dump(n, "u1", v.data(), 1, s); }

// -----------------------------------------------------------------------------
// scene flattening (preorder)
// -----------------------------------------------------------------------------
static std::vector<const Node *> g_preorder;
static std::map<const Node *, int> g_nodeIndex;
static std::vector<int> g_parent;

static void walk(const Node *n, int parent)
{
    int me = (int)g_preorder.size();
    g_preorder.push_back(n);
    g_parent.push_back(parent);
    g_nodeIndex[n] = me;
    for (int i = 0; i < n->GetNumChild(); ++i) walk(n->GetChild(i), me);
}

static void flattenScene()
{
    g_preorder.clear(); g_nodeIndex.clear(); g_parent.clear();
    walk(&renderer.GetScene().rootNode, -1);
}

static int nodeId(const HitInfo &h, bool hit)
{
    if (!hit) return -1;
    if (h.light) return -2; // single-light scenes; light index 0
    auto it = g_nodeIndex.find(h.node);
    return it == g_nodeIndex.end() ? -3 : it->second;
}

static void setupCamera(int W, int H)
{
    Camera &cam = renderer.GetCamera();
    if (W > 0) cam.imgWidth = W;
    if (H > 0) cam.imgHeight = H;
    // same derivation as the reference's main() does before rendering
    const Vec3f camZ{ -cam.dir.GetNormalized() };
    const Vec3f camX{ cam.up.Cross(camZ).GetNormalized() };
    const Vec3f camY{ camZ.Cross(camX) };
    tileThreads::cameraToWorld = Matrix3f{ camX, camY, camZ };
    constexpr float Pi = 3.14159265358979323846f;
    tileThreads::imagePlaneHalfHeight = cam.focaldist * tanf((static_cast<float>(cam.fov) * Pi / 180.0f) / 2.0f);
    const float aspectRatio{ static_cast<float>(cam.imgWidth) / static_cast<float>(cam.imgHeight) };
    tileThreads::imagePlaneHalfWidth = aspectRatio * tileThreads::imagePlaneHalfHeight;
    tileThreads::pixelSize = (tileThreads::imagePlaneHalfWidth * 2.0f) / static_cast<float>(cam.imgWidth);
}

static void cmdDumpScene()
{
    flattenScene();
    const Scene &sc = renderer.GetScene();
    size_t nn = g_preorder.size();
    std::vector<float> tm(nn * 12), itm(nn * 12);
    std::vector<int32_t> parent(nn), prim(nn), mesh(nn), mtl(nn);
    // the material list with every MultiMtl replaced by its members (a MultiMtl has no constants of its own)
    std::vector<const Material *> flat;
    std::vector<int32_t> flat_first, flat_members;
    for (size_t k = 0; k < sc.materials.size(); ++k) {
        flat_first.push_back((int32_t)flat.size());
        if (auto *mm = dynamic_cast<const MultiMtl *>(sc.materials[k])) {
            for (const Material *m : mm->mtls) flat.push_back(m);
            flat_members.insert(flat_members.end(), mm->mtls.size(), (int32_t)mm->mtls.size());
        } else { flat.push_back(sc.materials[k]); flat_members.push_back(0); }
    }
    for (size_t i = 0; i < nn; ++i) {
        const Node *n = g_preorder[i];
        memcpy(&tm[i * 12], n->GetTransform().cell, 48);
        memcpy(&itm[i * 12], n->GetInverseTransform().cell, 48);
        parent[i] = g_parent[i];
        const Object *o = n->GetNodeObj();
        prim[i] = 0; mesh[i] = -1;
        if (o) {
            if (dynamic_cast<const Sphere *>(o)) prim[i] = 1;
            else if (dynamic_cast<const Plane *>(o)) prim[i] = 2;
            else if (dynamic_cast<const TriObj *>(o)) {
                prim[i] = 3;
                for (size_t k = 0; k < sc.objList.size(); ++k) if (sc.objList[k] == o) mesh[i] = (int)k;
            }
        }
        mtl[i] = -1;
        for (size_t k = 0; k < flat_first.size(); ++k) if (sc.materials[k] == n->GetMaterial()) mtl[i] = flat_first[k];
    }
    dumpF("node_tm", tm, {nn, 12}); dumpF("node_itm", itm, {nn, 12});
    dumpI("node_parent", parent, {nn}); dumpI("node_prim", prim, {nn});
    dumpI("node_mesh", mesh, {nn}); dumpI("node_material", mtl, {nn});

    size_t nm = flat.size();
    std::vector<float> m(nm * 20);
    dumpI("material_members", flat_members, {nm}); // members of the MultiMtl a material belongs to, 0 = an ordinary material
    for (size_t k = 0; k < nm; ++k) {
        const MtlBasePhongBlinn *b = static_cast<const MtlBasePhongBlinn *>(flat[k]);
        float *d = &m[k * 20];
        Color c = b->Diffuse().GetValue();    d[0] = c.r; d[1] = c.g; d[2] = c.b;
        c = b->Specular().GetValue();         d[3] = c.r; d[4] = c.g; d[5] = c.b;
        d[6] = b->Glossiness().GetValue();
        c = b->Emission().GetValue();         d[7] = c.r; d[8] = c.g; d[9] = c.b;
        c = b->Reflection().GetValue();       d[10] = c.r; d[11] = c.g; d[12] = c.b;
        c = b->Refraction().GetValue();       d[13] = c.r; d[14] = c.g; d[15] = c.b;
        c = b->Absorption();                  d[16] = c.r; d[17] = c.g; d[18] = c.b;
        d[19] = b->IOR();
    }
    dumpF("materials", m, {nm, 20});

    // lights: fields are protected; -fno-access-control lets the harness read them
    size_t nl = sc.lights.size();
    std::vector<float> L(nl * 12, 0.0f);
    for (size_t k = 0; k < nl; ++k) {
        float *d = &L[k * 12];
        if (auto *p = dynamic_cast<const PointLight *>(sc.lights[k])) {
            d[0] = 2; d[1] = p->intensity.r; d[2] = p->intensity.g; d[3] = p->intensity.b;
            d[4] = p->position.x; d[5] = p->position.y; d[6] = p->position.z;
            d[10] = p->size; d[11] = p->attenuation;
        } else if (auto *q = dynamic_cast<const DirectLight *>(sc.lights[k])) {
            d[0] = 1; d[1] = q->intensity.r; d[2] = q->intensity.g; d[3] = q->intensity.b;
            d[7] = q->direction.x; d[8] = q->direction.y; d[9] = q->direction.z;
        } else if (auto *a = dynamic_cast<const AmbientLight *>(sc.lights[k])) {
            d[0] = 0; d[1] = a->intensity.r; d[2] = a->intensity.g; d[3] = a->intensity.b;
        }
    }
    dumpF("lights", L, {nl, 12});

    const Camera &cam = renderer.GetCamera();
    std::vector<float> c = { cam.pos.x, cam.pos.y, cam.pos.z, cam.dir.x, cam.dir.y, cam.dir.z,
                             cam.up.x, cam.up.y, cam.up.z, cam.fov, cam.focaldist, cam.dof,
                             (float)cam.imgWidth, (float)cam.imgHeight, cam.sRGB ? 1.0f : 0.0f };
    dumpF("camera", c, {c.size()});
    std::vector<float> d(12);
    memcpy(d.data(), tileThreads::cameraToWorld.cell, 36);
    d[9] = tileThreads::imagePlaneHalfWidth; d[10] = tileThreads::imagePlaneHalfHeight; d[11] = tileThreads::pixelSize;
    dumpF("camera_derived", d, {12});
    Color bg = sc.background.GetValue();
    std::vector<float> b = { bg.r, bg.g, bg.b };
    dumpF("background", b, {3});

    Color en = sc.environment.GetValue();
    std::vector<float> e3 = { en.r, en.g, en.b };
    dumpF("environment", e3, {3});

    // textures: the maps of background, environment and every material's diffuse colour, with known-answer lookups
    {
        struct Owner { int kind, index; const TextureMap *map; };
        std::vector<Owner> owners;
        if (sc.background.GetTexture()) owners.push_back({0, 0, sc.background.GetTexture()});
        if (sc.environment.GetTexture()) owners.push_back({1, 0, sc.environment.GetTexture()});
        for (size_t k = 0; k < nm; ++k) {
            const MtlBasePhongBlinn *b = static_cast<const MtlBasePhongBlinn *>(flat[k]);
            if (b->Diffuse().GetTexture()) owners.push_back({2, (int)k, b->Diffuse().GetTexture()});
        }
        const size_t nt = owners.size(), nq = 256;
        std::vector<int32_t> own(nt * 2), type(nt), size(nt * 2, 0);
        std::vector<float> itm_(nt * 12), cols(nt * 6, 0.0f), uvw(nt * nq * 3), rgb(nt * nq * 3);
        RNG g(123, 5);
        for (size_t t = 0; t < nt; ++t) {
            const TextureMap *m = owners[t].map;
            own[t * 2] = owners[t].kind; own[t * 2 + 1] = owners[t].index;
            memcpy(&itm_[t * 12], m->GetInverseTransform().cell, 48);
            const Texture *tx = m->texture; // (null for a map whose file could not be read: type 0, evaluates to black)
            if (auto *c = dynamic_cast<const TextureChecker *>(tx)) {
                type[t] = 1;
                Color a = c->color[0].GetValue(), b = c->color[1].GetValue();
                float *d = &cols[t * 6]; d[0] = a.r; d[1] = a.g; d[2] = a.b; d[3] = b.r; d[4] = b.g; d[5] = b.b;
            } else if (auto *f = dynamic_cast<const TextureFile *>(tx)) {
                type[t] = 2;
                size[t * 2] = f->width; size[t * 2 + 1] = f->height;
                std::vector<uint8_t> px((size_t)f->width * f->height * 3);
                memcpy(px.data(), f->data.data(), px.size());
                char nm_[32]; snprintf(nm_, sizeof nm_, "tex%zu_pixels", t);
                dumpB(nm_, px, {(size_t)f->height, (size_t)f->width, 3});
            }
            for (size_t q = 0; q < nq; ++q) {
                Vec3f p{ (g.RandomFloat() - 0.5f) * 8.0f, (g.RandomFloat() - 0.5f) * 8.0f, (g.RandomFloat() - 0.5f) * 8.0f };
                if (q % 4 == 0) p = Vec3f{ g.RandomFloat(), g.RandomFloat(), 0.0f };
                Color c = m->Eval(p);
                memcpy(&uvw[(t * nq + q) * 3], &p.x, 12);
                rgb[(t * nq + q) * 3] = c.r; rgb[(t * nq + q) * 3 + 1] = c.g; rgb[(t * nq + q) * 3 + 2] = c.b;
            }
        }
        dumpI("tex_owner", own, {nt, 2}); dumpI("tex_type", type, {nt}); dumpI("tex_size", size, {nt, 2});
        dumpF("tex_itm", itm_, {nt, 12}); dumpF("tex_colors", cols, {nt, 6});
        dumpF("tex_kat_uvw", uvw, {nt, nq, 3}); dumpF("tex_kat_rgb", rgb, {nt, nq, 3});
        // EvalEnvironment for random directions
        std::vector<float> ed(nq * 3), ec(nq * 3);
        for (size_t q = 0; q < nq; ++q) {
            Vec3f d{ g.RandomFloat() - 0.5f, g.RandomFloat() - 0.5f, g.RandomFloat() - 0.5f };
            if (q & 1) d *= 7.0f;
            Color c = sc.environment.EvalEnvironment(d);
            memcpy(&ed[q * 3], &d.x, 12);
            ec[q * 3] = c.r; ec[q * 3 + 1] = c.g; ec[q * 3 + 2] = c.b;
        }
        dumpF("env_kat_dir", ed, {nq, 3}); dumpF("env_kat_rgb", ec, {nq, 3});
    }

    // meshes + BVH
    for (size_t k = 0; k < sc.objList.size(); ++k) {
        const TriObj *t = static_cast<const TriObj *>(sc.objList[k]);
        char nm_[64];
        auto nameOf = [&](const char *s) { snprintf(nm_, sizeof nm_, "mesh%zu_%s", k, s); return nm_; };
        dump(nameOf("v"), "f4", &t->V(0), 4, {t->NV(), 3});
        dump(nameOf("vn"), "f4", &t->VN(0), 4, {t->NVN(), 3});
        dump(nameOf("vt"), "f4", &t->VT(0), 4, {t->NVT(), 3});
        dump(nameOf("f"), "u4", &t->F(0), 4, {t->NF(), 3});
        dump(nameOf("fn"), "u4", &t->FN(0), 4, {t->NF(), 3});
        dump(nameOf("ft"), "u4", &t->FT(0), 4, {t->NF(), 3});
        if (t->NM() > 0) { // sub-material of every face, as the mesh's own accessor reports it
            std::vector<int32_t> fm(t->NF());
            for (unsigned i = 0; i < t->NF(); ++i) fm[i] = t->GetMaterialIndex((int)i);
            dumpI(nameOf("face_material"), fm, {fm.size()});
        }
        // walk the BVH through its public accessors to find the node count
        const BVHTriMesh &bvh = t->bvh;
        std::vector<uint32_t> stack{ bvh.GetRootNodeID() };
        uint32_t maxId = 1;
        while (!stack.empty()) {
            uint32_t id = stack.back(); stack.pop_back();
            if (id > maxId) maxId = id;
            if (!bvh.IsLeafNode(id)) { stack.push_back(bvh.GetFirstChildNode(id)); stack.push_back(bvh.GetSecondChildNode(id)); }
        }
        std::vector<uint32_t> words((maxId + 1) * 7, 0u);
        memcpy(words.data() + 7, bvh.GetNodeBounds(1), (size_t)maxId * 28); // node = 6 floats + 1 data word, 28 bytes
        dumpU(nameOf("bvh_nodes"), words, {(size_t)maxId + 1, 7});
        // element list: leaves reference a contiguous index array of NF entries
        std::vector<uint32_t> elems(t->NF());
        // recover base pointer from any leaf: offset + pointer
        stack.push_back(1);
        while (!stack.empty()) {
            uint32_t id = stack.back(); stack.pop_back();
            if (bvh.IsLeafNode(id)) {
                uint32_t data = words[id * 7 + 6];
                uint32_t off = data & ((1u << 28) - 1);
                const unsigned int *e = bvh.GetNodeElements(id);
                for (unsigned i = 0; i < bvh.GetNodeElementCount(id); ++i) elems[off + i] = e[i];
            } else { stack.push_back(bvh.GetFirstChildNode(id)); stack.push_back(bvh.GetSecondChildNode(id)); }
        }
        dumpU(nameOf("bvh_elements"), elems, {elems.size()});
    }
}

// -----------------------------------------------------------------------------
// seeded render: per-pixel seed, one PCG stream per sample
// -----------------------------------------------------------------------------
static inline uint64_t pixelSeed(uint64_t frameSeed, uint64_t pixelIndex)
{
    return frameSeed + pixelIndex * 0x9E3779B97F4A7C15ULL;
}

#ifdef REF_P12
// The snapshot's switches (its renderer.cpp, which the harness replaces).  They are compile-time
// constants there too; `make ref12` builds one harness with all four off and one per photon-map
// combination that has goldens (-DPM_DIRECT=1 ... on the compiler line).
#ifndef PM_DIRECT
#define PM_DIRECT 0
#endif
#ifndef PM_INDIRECT
#define PM_INDIRECT 0
#endif
#ifndef PM_CAUSTICS
#define PM_CAUSTICS 0
#endif
#ifndef PM_MONTE_CARLO
#define PM_MONTE_CARLO 0
#endif
Renderer renderer;
const bool doingDirectWithPhotonMapping{ PM_DIRECT != 0 };
const bool doingIndirectWithPhotonMapping{ PM_INDIRECT != 0 };
const bool doingCaustics{ PM_CAUSTICS != 0 };
const bool monteCarloWithPhoton{ PM_MONTE_CARLO != 0 };

// ---- photon maps: the snapshot's emission loops (its main.cpp:292-470) with one PCG stream per photon
// path instead of the one global stream: path p of the global map starts from SetSequence(p, seed ^ "PHOT"),
// path p of the caustics map from SetSequence(p, seed ^ "CAUS").  Paths run in order until the map is
// full, exactly as there; `maxSegments` bounds the TraceRay calls of one path (the snapshot has no bound).
static PhotonMap g_photonMap, g_causticsMap;
static std::vector<PhotonMap::PhotonData> g_rawGlobal, g_rawCaustic;
static uint64_t g_pathsGlobal = 0, g_pathsCaustic = 0;
static const uint64_t kSeedGlobal = 0x50484F54ULL, kSeedCaustic = 0x43415553ULL;

static void fillPhotonMaps(uint64_t seed, int nGlobal, int nCaustic, int maxSegments)
{
    RNG &rng = tileThreads::rng;
    g_photonMap.Resize(nGlobal);
    g_causticsMap.Resize(nCaustic);
    renderer.SetPhotonMap(&g_photonMap);
    renderer.SetCausticsMap(&g_causticsMap);
    const Light *light{ renderer.GetScene().lights[0] };
    g_pathsGlobal = g_pathsCaustic = 0;
    const bool directOnly = doingDirectWithPhotonMapping || monteCarloWithPhoton; // the first branch's condition
    bool mapIsFull = nGlobal <= 0;
    if (directOnly) {
        for (uint64_t p = 0; !mapIsFull; ++p) {
            rng.SetSequence(p, seed ^ kSeedGlobal);
            g_pathsGlobal = p + 1;
            Ray photonRay;
            Color c;
            light->RandomPhoton(rng, photonRay, c);
            photonRay.p += photonRay.dir * 0.0002f;
            HitInfo hInfo{};
            if (!renderer.TraceRay(photonRay, hInfo)) continue;
            if (hInfo.light) continue; // the snapshot would dereference a null node here; cannot happen with one light
            const Material *material{ hInfo.node->GetMaterial() };
            if (!material->IsPhotonSurface()) continue;
            if (!g_photonMap.AddPhoton(hInfo.p, photonRay.dir, c)) mapIsFull = true;
        }
    } else if (doingIndirectWithPhotonMapping) {
        for (uint64_t p = 0; !mapIsFull; ++p) {
            rng.SetSequence(p, seed ^ kSeedGlobal);
            g_pathsGlobal = p + 1;
            Ray photonRay;
            Color c;
            light->RandomPhoton(rng, photonRay, c);
            bool firstHit{ true };
            for (int seg = 0; seg < maxSegments; ++seg) {
                photonRay.p += photonRay.dir * 0.0002f;
                HitInfo hInfo{};
                if (!renderer.TraceRay(photonRay, hInfo) || hInfo.light) break;
                SamplerInfo sInfo{ rng };
                sInfo.SetHit(photonRay, hInfo);
                DirSampler::Info info{};
                Vec3f newDir{};
                const Material *material{ hInfo.node->GetMaterial() };
                const bool bounce{ material->GenerateSample(sInfo, newDir, info) };
                if (firstHit) {
                    if (!bounce || (doingCaustics && info.lobe != DirSampler::Lobe::DIFFUSE)) break;
                    firstHit = false;
                    photonRay.dir = newDir;
                    photonRay.p = hInfo.p;
                    c *= info.mult / info.prob;
                    continue;
                }
                if (material->IsPhotonSurface() && !g_photonMap.AddPhoton(hInfo.p, photonRay.dir, c)) { mapIsFull = true; break; }
                if (!bounce) break;
                photonRay.dir = newDir;
                photonRay.p = hInfo.p;
                c *= info.mult / info.prob;
            }
        }
    }
    g_rawGlobal.assign(g_photonMap.photons.begin() + 1, g_photonMap.photons.begin() + 1 + g_photonMap.NumPhotons());
    g_photonMap.ScalePhotonPowers(1.0f / static_cast<float>(g_photonMap.NumPhotons()));
    g_photonMap.PrepareForIrradianceEstimation();

    g_rawCaustic.clear();
    if (doingCaustics) {
        mapIsFull = nCaustic <= 0;
        for (uint64_t p = 0; !mapIsFull; ++p) {
            rng.SetSequence(p, seed ^ kSeedCaustic);
            g_pathsCaustic = p + 1;
            Ray photonRay;
            Color c;
            light->RandomPhoton(rng, photonRay, c);
            bool firstHit{ true };
            for (int seg = 0; seg < maxSegments; ++seg) {
                photonRay.p += photonRay.dir * 0.0002f;
                HitInfo hInfo{};
                if (!renderer.TraceRay(photonRay, hInfo) || hInfo.light) break;
                SamplerInfo sInfo{ rng };
                sInfo.SetHit(photonRay, hInfo);
                DirSampler::Info info{};
                Vec3f newDir{};
                const Material *material{ hInfo.node->GetMaterial() };
                const bool bounce{ material->GenerateSample(sInfo, newDir, info) };
                if (firstHit) {
                    if (!bounce || info.lobe == DirSampler::Lobe::DIFFUSE) break;
                    firstHit = false;
                    photonRay.dir = newDir;
                    photonRay.p = hInfo.p;
                    c *= info.mult / info.prob;
                    continue;
                }
                if (material->IsPhotonSurface()) { mapIsFull = !g_causticsMap.AddPhoton(hInfo.p, photonRay.dir, c); break; }
                if (!bounce) break;
                photonRay.dir = newDir;
                photonRay.p = hInfo.p;
                c *= info.mult / info.prob;
            }
        }
        g_rawCaustic.assign(g_causticsMap.photons.begin() + 1, g_causticsMap.photons.begin() + 1 + g_causticsMap.NumPhotons());
        g_causticsMap.ScalePhotonPowers(1.0f / static_cast<float>(g_causticsMap.NumPhotons()));
        g_causticsMap.PrepareForIrradianceEstimation();
    }
}

// 32-byte photon records as 8 words each.  AddPhoton leaves the last byte (split plane) indeterminate
// and only nodes with a child get it set, so it is written out as plane for those and 0 otherwise.
static void dumpPhotons(const char *name, const PhotonMap::PhotonData *p, size_t n, bool balanced)
{
    static_assert(sizeof(PhotonMap::PhotonData) == 32, "photon record");
    std::vector<uint32_t> w(n * 8);
    if (n) memcpy(w.data(), p, n * 32);
    for (size_t i = 0; i < n; ++i) {
        uint8_t *b = (uint8_t *)&w[i * 8 + 7];
        const bool hasChild = balanced && 2 * (i + 1) <= n; // 1-based index i+1
        b[3] = hasChild ? (uint8_t)(b[3] & 3) : 0;
    }
    dumpU(name, w, {n, 8});
}

static void dumpPhotonMaps()
{
    dumpPhotons("pm_global_raw", g_rawGlobal.data(), g_rawGlobal.size(), false);
    dumpPhotons("pm_caustic_raw", g_rawCaustic.data(), g_rawCaustic.size(), false);
    dumpPhotons("pm_global_tree", g_photonMap.GetPhotons(), (size_t)g_photonMap.NumPhotons(), true);
    dumpPhotons("pm_caustic_tree", g_rawCaustic.empty() ? nullptr : g_causticsMap.GetPhotons(), g_rawCaustic.size(), true);
    std::vector<int32_t> paths = { (int32_t)g_pathsGlobal, (int32_t)g_pathsCaustic };
    dumpI("pm_paths", paths, {2});
}

// RandomPhoton for paths 0..n-1 and the snapshot's GenerateSample for every material at synthetic surface points
static void cmdKatPhoton(int n, uint64_t seed)
{
    const Scene &sc = renderer.GetScene();
    const Light *light = sc.lights[0];
    RNG &rng = tileThreads::rng;
    std::vector<float> em((size_t)n * 9);
    for (int i = 0; i < n; ++i) {
        rng.SetSequence((uint64_t)i, seed ^ kSeedGlobal);
        Ray r; Color c;
        light->RandomPhoton(rng, r, c);
        float *o = &em[(size_t)i * 9];
        o[0] = r.p.x; o[1] = r.p.y; o[2] = r.p.z; o[3] = r.dir.x; o[4] = r.dir.y; o[5] = r.dir.z; o[6] = c.r; o[7] = c.g; o[8] = c.b;
    }
    dumpF("ph_emit", em, {(size_t)n, 9});
    size_t nm = sc.materials.size();
    RNG g(98, seed);
    std::vector<float> in((size_t)n * 9), mg((size_t)n * nm * 7);
    std::vector<int32_t> mflags((size_t)n * nm * 3), frontv(n);
    for (int i = 0; i < n; ++i) {
        HitInfo h{};
        h.p = Vec3f{ (g.RandomFloat() - 0.5f) * 28.0f, (g.RandomFloat() - 0.5f) * 36.0f, g.RandomFloat() * 23.0f };
        h.N = Vec3f{ g.RandomFloat() - 0.5f, g.RandomFloat() - 0.5f, g.RandomFloat() - 0.5f };
        h.GN = h.N;
        h.z = 1.0f + g.RandomFloat() * 30.0f;
        h.front = g.RandomFloat() < 0.7f;
        Ray r;
        r.dir = Vec3f{ g.RandomFloat() - 0.5f, g.RandomFloat() - 0.5f, g.RandomFloat() - 0.5f } * 3.0f;
        if ((r.dir % h.N) > 0.0f && (i % 5) != 0) r.dir = -r.dir;
        r.p = h.p - r.dir * h.z;
        float *I = &in[(size_t)i * 9];
        memcpy(I, &h.p.x, 12); memcpy(I + 3, &h.N.x, 12); memcpy(I + 6, &r.dir.x, 12);
        frontv[i] = h.front;
        SamplerInfo sI{ rng };
        sI.SetHit(r, h);
        for (size_t m = 0; m < nm; ++m) {
            rng.SetSequence((uint64_t)i + 1, seed ^ (0x2000 + m));
            DirSampler::Info si{};
            Vec3f bd{ 0, 0, 0 };
            bool b = sc.materials[m]->GenerateSample(sI, bd, si);
            float *q = &mg[((size_t)i * nm + m) * 7];
            q[0] = bd.x; q[1] = bd.y; q[2] = bd.z; q[3] = si.mult.r; q[4] = si.mult.g; q[5] = si.mult.b; q[6] = si.prob;
            int32_t *f = &mflags[((size_t)i * nm + m) * 3];
            f[0] = b; f[1] = (int)si.lobe; f[2] = (int)rng.RandomInt(); // the next draw pins the number of draws consumed
        }
    }
    dumpF("ph_smp_in", in, {(size_t)n, 9}); dumpI("ph_smp_front", frontv, {(size_t)n});
    dumpF("ph_smp_gen", mg, {(size_t)n, nm, 7}); dumpI("ph_smp_flags", mflags, {(size_t)n, nm, 3});
}

// photons seed nGlobal nCaustic maxSegments nQueries radius: the two maps, then EstimateIrradiance<128> at the
// primary hits of a coarse grid of camera rays and at random points with random normals
static void cmdPhotons(uint64_t seed, int nGlobal, int nCaustic, int maxSegments, int nQ, float radius)
{
    fillPhotonMaps(seed, nGlobal, nCaustic, maxSegments);
    dumpPhotonMaps();
    const Camera &cam = renderer.GetCamera();
    std::vector<float> q, out;
    RNG g(31, seed);
    int side = 1;
    while (side * side < nQ / 2) ++side;
    for (int k = 0; k < nQ; ++k) {
        Vec3f P, N;
        if (k < side * side) {
            const float sx = -tileThreads::imagePlaneHalfWidth + 2 * tileThreads::imagePlaneHalfWidth * ((k % side) + 0.5f) / side;
            const float sy = tileThreads::imagePlaneHalfHeight - 2 * tileThreads::imagePlaneHalfHeight * ((k / side) + 0.5f) / side;
            const Ray ray{ cam.pos, tileThreads::cameraToWorld * Vec3f{ sx, sy, -cam.focaldist } };
            HitInfo h{};
            if (!renderer.TraceRay(ray, h) || h.light) continue;
            P = h.p; N = h.N.GetNormalized();
        } else {
            P = Vec3f{ (g.RandomFloat() - 0.5f) * 30.0f, (g.RandomFloat() - 0.5f) * 40.0f, g.RandomFloat() * 24.0f };
            N = Vec3f{ g.RandomFloat() - 0.5f, g.RandomFloat() - 0.5f, g.RandomFloat() - 0.5f }.GetNormalized();
        }
        Color ig{ 0.0f }, ic{ 0.0f };
        Vec3f dg{ 0, 0, 0 }, dc{ 0, 0, 0 };
        if (g_photonMap.NumPhotons() > 0) g_photonMap.EstimateIrradiance<128>(ig, dg, radius, P, N, 1.0f);
        if (g_causticsMap.NumPhotons() > 0) g_causticsMap.EstimateIrradiance<128>(ic, dc, radius, P, N, 1.0f);
        q.insert(q.end(), { P.x, P.y, P.z, N.x, N.y, N.z });
        out.insert(out.end(), { ig.r, ig.g, ig.b, ic.r, ic.g, ic.b });
    }
    dumpF("pm_query", q, {q.size() / 6, 6});
    dumpF("pm_irradiance", out, {out.size() / 6, 6});
}

// CanBounce with a run-time limit.  The snapshot copies the ShadeInfo by value into a plain
// ShadeInfo for every secondary ray, so below the camera ray its compiled-in limit of 5 is
// back: only max_bounces 0 and 5 mean the same thing at every depth.
static int g_maxBounces = 5;
struct HarnessShadeInfo : public ShadeInfo {
    using ShadeInfo::ShadeInfo;
    bool CanBounce() const override { return bounce < g_maxBounces; }
};

// the pixel loop of the snapshot (its main.cpp:160-270) with the seeding convention of cmdRender
static void cmdRenderWhitted(uint64_t seed, int minS, int maxS, float deltaMax, int W, int H, int maxBounces,
                             int nGlobal = 0, int nCaustic = 0, int maxSegments = 64)
{
    if (doingDirectWithPhotonMapping || doingIndirectWithPhotonMapping || doingCaustics || monteCarloWithPhoton) {
        fillPhotonMaps(seed, nGlobal, nCaustic, maxSegments);
        dumpPhotonMaps();
    }
    g_maxBounces = maxBounces;
    setupCamera(W, H);
    flattenScene();
    const Camera &cam = renderer.GetCamera();
    W = cam.imgWidth; H = cam.imgHeight;
    const size_t rows = (size_t)H;
    std::vector<uint8_t> rgb(rows * W * 3);
    std::vector<float> lin(rows * W * 3), z(rows * W);
    std::vector<int32_t> cnt(rows * W), ids(rows * W);
    for (int j = 0; j < H; ++j) {
        for (int i = 0; i < W; ++i) {
            const uint64_t ps = pixelSeed(seed, (uint64_t)j * W + i);
            RNG &rng = tileThreads::rng;
            rng.SetSequence(0, ps);
            const float aaX = rng.RandomFloat();
            const float aaY = rng.RandomFloat();
            const float dofT = rng.RandomFloat();
            const float dofR = rng.RandomFloat();
            Color sum{ 0.0f }, sumSq{ 0.0f };
            size_t n = 0;
            const size_t o = (size_t)j * W + i;
            for (int k = 0; k < maxS; ++k) {
                ++n;
                rng.SetSequence((uint64_t)k + 1, ps);
                const float jx{ fmod(tileThreads::aaHaltonSeqX[k] + aaX, 1.0f) };
                const float jy{ fmod(tileThreads::aaHaltonSeqY[k] + aaY, 1.0f) };
                const float px{ static_cast<float>(i) + jx };
                const float py{ static_cast<float>(j) + jy };
                const float sx{ -tileThreads::imagePlaneHalfWidth + tileThreads::pixelSize * px };
                const float sy{ tileThreads::imagePlaneHalfHeight - tileThreads::pixelSize * py };
                const Vec3f dest{ cam.pos + tileThreads::cameraToWorld * Vec3f{ sx, sy, -cam.focaldist } };
                const float jt{ fmod(tileThreads::diskHaltonSeqTheta[k] + dofT, 1.0f) };
                const float jr{ fmod(tileThreads::diskHaltonSeqRadius[k] + dofR, 1.0f) };
                const float diskTheta = jt * 2.0f * M_PI;
                const float diskRadius{ sqrt(jr) };
                const Vec3f off{ diskRadius * cam.dof * cos(diskTheta), diskRadius * cam.dof * sin(diskTheta), 0.0f };
                const Vec3f org{ cam.pos + tileThreads::cameraToWorld * off };
                const Ray ray{ org, dest - org };
                HitInfo h{};
                HarnessShadeInfo sInfo{ renderer.GetScene().lights, renderer.GetScene().environment, rng };
                sInfo.SetPixelSample(i);
                Color c;
                const bool hit = renderer.TraceRay(ray, h);
                if (k == 0) { ids[o] = nodeId(h, hit); z[o] = h.z; }
                if (hit) {
                    sInfo.SetHit(ray, h);
                    c = h.light ? Color{ 1.0f, 1.0f, 1.0f } : h.node->GetMaterial()->Shade(sInfo);
                } else {
                    const float u{ (static_cast<float>(i) + tileThreads::aaHaltonSeqX[k]) / static_cast<float>(cam.imgWidth - 1) };
                    const float v{ (static_cast<float>(j) + tileThreads::aaHaltonSeqY[k]) / static_cast<float>(cam.imgHeight - 1) };
                    c = renderer.GetScene().background.Eval(Vec3f{ u, v, 1.0f });
                }
                sum += c;
                sumSq += c * c;
                if ((int)n < minS) continue;
                Color sigma{ (sumSq - ((sum * sum) / static_cast<float>(n))) / (static_cast<float>(n - 1)) };
                sigma.r = sqrtf(fmaxf(0.0f, sigma.r));
                sigma.g = sqrtf(fmaxf(0.0f, sigma.g));
                sigma.b = sqrtf(fmaxf(0.0f, sigma.b));
                const Color delta{ 3.0f * (sigma / sqrtf(static_cast<float>(n))) };
                if (delta.r < deltaMax && delta.g < deltaMax && delta.b < deltaMax) break;
            }
            Color c{ sum / static_cast<float>(n) };
            lin[o * 3 + 0] = c.r; lin[o * 3 + 1] = c.g; lin[o * 3 + 2] = c.b;
            if (cam.sRGB) c = c.Linear2sRGB();
            const Color24 q{ c };
            rgb[o * 3 + 0] = q.r; rgb[o * 3 + 1] = q.g; rgb[o * 3 + 2] = q.b;
            cnt[o] = (int)n;
        }
    }
    dumpB("rgb", rgb, {rows, (size_t)W, 3});
    dumpF("linear", lin, {rows, (size_t)W, 3});
    dumpF("z", z, {rows, (size_t)W});
    dumpI("sample_count", cnt, {rows, (size_t)W});
    dumpI("object_id", ids, {rows, (size_t)W});
}
#else
static void cmdRender(uint64_t seed, int minS, int maxS, float deltaMax, int W, int H, int y0, int y1)
{
    setupCamera(W, H);
    flattenScene();
    const Camera &cam = renderer.GetCamera();
    W = cam.imgWidth; H = cam.imgHeight;
    if (y1 <= 0 || y1 > H) y1 = H;
    size_t rows = (size_t)(y1 - y0);
    std::vector<uint8_t> rgb(rows * W * 3);
    std::vector<float> lin(rows * W * 3), z(rows * W);
    std::vector<int32_t> cnt(rows * W), ids(rows * W);
    uint64_t nClosest = 0;
    for (int j = y0; j < y1; ++j) {
        for (int i = 0; i < W; ++i) {
            const uint64_t ps = pixelSeed(seed, (uint64_t)j * W + i);
            RNG &rng = tileThreads::rng;
            rng.SetSequence(0, ps);
            const float aaX = rng.RandomFloat();
            const float aaY = rng.RandomFloat();
            const float dofT = rng.RandomFloat();
            const float dofR = rng.RandomFloat();
            Color sum{ 0.0f }, sumSq{ 0.0f };
            size_t n = 0;
            size_t o = (size_t)(j - y0) * W + i;
            for (int k = 0; k < maxS; ++k) {
                ++n;
                rng.SetSequence((uint64_t)k + 1, ps);
                const float jx{ fmod(tileThreads::aaHaltonSeqX[k] + aaX, 1.0f) };
                const float jy{ fmod(tileThreads::aaHaltonSeqY[k] + aaY, 1.0f) };
                const float px{ static_cast<float>(i) + jx };
                const float py{ static_cast<float>(j) + jy };
                const float sx{ -tileThreads::imagePlaneHalfWidth + tileThreads::pixelSize * px };
                const float sy{ tileThreads::imagePlaneHalfHeight - tileThreads::pixelSize * py };
                const Vec3f dest{ cam.pos + tileThreads::cameraToWorld * Vec3f{ sx, sy, -cam.focaldist } };
                const float jt{ fmod(tileThreads::diskHaltonSeqTheta[k] + dofT, 1.0f) };
                const float jr{ fmod(tileThreads::diskHaltonSeqRadius[k] + dofR, 1.0f) };
                const float diskTheta{ jt * 2.0f * M_PI };
                const float diskRadius{ sqrt(jr) };
                const Vec3f off{ diskRadius * cam.dof * cos(diskTheta), diskRadius * cam.dof * sin(diskTheta), 0.0f };
                const Vec3f org{ cam.pos + tileThreads::cameraToWorld * off };
                const Ray ray{ org, dest - org };
                if (k == 0) {
                    HitInfo h{};
                    bool hit = renderer.TraceRay(ray, h, HIT_FRONT_AND_BACK);
                    ids[o] = nodeId(h, hit);
                    z[o] = h.z;
                }
                const Color c{ tracePath(ray) };
                ++nClosest;
                sum += c;
                sumSq += c * c;
                if ((int)n < minS) continue;
                Color sigma{ (sumSq - ((sum * sum) / static_cast<float>(n))) / (static_cast<float>(n - 1)) };
                sigma.r = sqrtf(fmaxf(0.0f, sigma.r));
                sigma.g = sqrtf(fmaxf(0.0f, sigma.g));
                sigma.b = sqrtf(fmaxf(0.0f, sigma.b));
                const Color delta{ 3.0f * (sigma / sqrtf(static_cast<float>(n))) };
                if (delta.r < deltaMax && delta.g < deltaMax && delta.b < deltaMax) break;
            }
            Color c{ sum / static_cast<float>(n) };
            lin[o * 3 + 0] = c.r; lin[o * 3 + 1] = c.g; lin[o * 3 + 2] = c.b;
            if (cam.sRGB) c = c.Linear2sRGB();
            const Color24 q{ c };
            rgb[o * 3 + 0] = q.r; rgb[o * 3 + 1] = q.g; rgb[o * 3 + 2] = q.b;
            cnt[o] = (int)n;
        }
    }
    dumpB("rgb", rgb, {rows, (size_t)W, 3});
    dumpF("linear", lin, {rows, (size_t)W, 3});
    dumpF("z", z, {rows, (size_t)W});
    dumpI("sample_count", cnt, {rows, (size_t)W});
    dumpI("object_id", ids, {rows, (size_t)W});
}

#endif // REF_P12

// -----------------------------------------------------------------------------
// known-answer vectors
// -----------------------------------------------------------------------------
static void cmdKatRng()
{
    const uint64_t seqs[4][2] = { {0, 0}, {1, 42}, {921599, 0x9E3779B97F4A7C15ULL}, {0xFFFFFFFFFFULL, 7} };
    std::vector<uint32_t> u; std::vector<float> f;
    for (auto &s : seqs) {
        RNG a(s[0], s[1]);
        for (int i = 0; i < 64; ++i) u.push_back(a.RandomInt());
        RNG b(s[0], s[1]);
        for (int i = 0; i < 64; ++i) f.push_back(b.RandomFloat());
    }
    dumpU("rng_u32", u, {4, 64}); dumpF("rng_f32", f, {4, 64});
    // one-argument constructor (seed = MixBits(seq)) and Advance
    std::vector<uint32_t> m;
    for (uint64_t s : { 0ULL, 1ULL, 1001ULL, 2002ULL, 123456789012ULL }) { RNG a(s); for (int i = 0; i < 8; ++i) m.push_back(a.RandomInt()); }
    dumpU("rng_onearg", m, {5, 8});
    std::vector<uint32_t> adv;
    for (int64_t d : { 1LL, 5LL, 1000LL, 123456789LL }) { RNG a(3, 9); a.Advance(d); for (int i = 0; i < 4; ++i) adv.push_back(a.RandomInt()); }
    dumpU("rng_advance", adv, {4, 4});
    std::vector<float> h;
    for (int base : { 2, 3, 5, 7 }) for (int i = 0; i < 16; ++i) h.push_back(Halton(i, base));
    dumpF("halton", h, {4, 16});
}

// random rays against the loaded scene through the reference's own TraceRay / TraceShadowRay
// aimed != 0: rays for a height-field mesh lying in |x| <= extent/2, |y| <= extent/4 around z = 0: half of them come
// down onto the mesh from above, half travel nearly horizontally through its relief (long BVH walks, many leaves)
static void cmdKatTrace(int n, uint64_t seed, float extent, int aimed = 0)
{
    flattenScene();
    RNG g(77, seed);
    std::vector<float> rays((size_t)n * 6), tmax(n), out((size_t)n * 12);
    std::vector<int32_t> ids(n), flags((size_t)n * 3);
    const Camera &cam = renderer.GetCamera();
    for (int i = 0; i < n; ++i) {
        Ray r;
        if (aimed) {
            const float hx = extent * 0.5f, hy = extent * 0.25f;
            Vec3f tgt;
            if (i & 1) {
                r.p = Vec3f{ (g.RandomFloat() - 0.5f) * 2.2f * hx, (g.RandomFloat() - 0.5f) * 2.2f * hy, (g.RandomFloat() - 0.5f) * 5.0f };
                tgt = Vec3f{ (g.RandomFloat() - 0.5f) * 2.0f * hx, (g.RandomFloat() - 0.5f) * 2.0f * hy, (g.RandomFloat() - 0.5f) * 5.0f };
            } else {
                r.p = Vec3f{ (g.RandomFloat() - 0.5f) * 2.5f * hx, (g.RandomFloat() - 0.5f) * 2.5f * hy, 3.0f + g.RandomFloat() * 20.0f };
                tgt = Vec3f{ (g.RandomFloat() - 0.5f) * 2.0f * hx, (g.RandomFloat() - 0.5f) * 2.0f * hy, (g.RandomFloat() - 0.5f) * 3.0f };
            }
            r.dir = tgt - r.p;
            if ((i % 4) == 3) r.dir.Normalize();
            float tm = g.RandomFloat() * 2.0f;
            if ((i % 4) == 3) tm *= 20.0f;
            memcpy(&rays[(size_t)i * 6], &r.p.x, 12); memcpy(&rays[(size_t)i * 6 + 3], &r.dir.x, 12);
            tmax[i] = tm;
        } else {
        // mix of rays starting near the camera and rays starting inside the scene volume
        if (i & 1) r.p = cam.pos + Vec3f{ g.RandomFloat() - 0.5f, g.RandomFloat() - 0.5f, g.RandomFloat() - 0.5f };
        else r.p = Vec3f{ (g.RandomFloat() - 0.5f) * extent, (g.RandomFloat() - 0.5f) * extent, g.RandomFloat() * extent * 0.5f };
        Vec3f tgt{ (g.RandomFloat() - 0.5f) * extent, (g.RandomFloat() - 0.5f) * extent, g.RandomFloat() * extent * 0.5f };
        r.dir = tgt - r.p;
        if ((i % 4) == 3) r.dir.Normalize();
        float tm = g.RandomFloat() * 2.0f;
        memcpy(&rays[(size_t)i * 6], &r.p.x, 12); memcpy(&rays[(size_t)i * 6 + 3], &r.dir.x, 12);
        tmax[i] = tm;
        }
        const float tm = tmax[i];
        HitInfo h{};
        bool hit = renderer.TraceRay(r, h, HIT_FRONT_AND_BACK);
        bool occ = renderer.TraceShadowRay(r, tm, HIT_FRONT_AND_BACK);
        ids[i] = nodeId(h, hit);
        flags[(size_t)i * 3 + 0] = hit; flags[(size_t)i * 3 + 1] = h.front; flags[(size_t)i * 3 + 2] = occ;
        float *o = &out[(size_t)i * 12];
        o[0] = h.z; o[1] = h.p.x; o[2] = h.p.y; o[3] = h.p.z; o[4] = h.N.x; o[5] = h.N.y; o[6] = h.N.z;
        o[7] = h.uvw.x; o[8] = h.uvw.y; o[9] = h.uvw.z; o[10] = 0; o[11] = 0;
        if (!hit) for (int k = 1; k < 10; ++k) o[k] = 0; // undefined on a miss
    }
    dumpF("kat_rays", rays, {(size_t)n, 6}); dumpF("kat_tmax", tmax, {(size_t)n});
    dumpF("kat_hit", out, {(size_t)n, 12}); dumpI("kat_id", ids, {(size_t)n}); dumpI("kat_flags", flags, {(size_t)n, 3});
}

#ifndef REF_P12
// sampler KATs: light and material GenerateSample / GetSampleInfo at synthetic surface points
static void cmdKatSamplers(int n, uint64_t seed)
{
    const Scene &sc = renderer.GetScene();
    const Light *light = sc.lights[0];
    size_t nm = sc.materials.size();
    RNG g(99, seed);
    std::vector<float> in((size_t)n * 12), lg((size_t)n * 9), li((size_t)n), mg((size_t)n * nm * 8), mi((size_t)n * nm);
    std::vector<int32_t> mflags((size_t)n * nm * 3), frontv(n);
    for (int i = 0; i < n; ++i) {
        HitInfo h{};
        h.p = Vec3f{ (g.RandomFloat() - 0.5f) * 28.0f, (g.RandomFloat() - 0.5f) * 36.0f, g.RandomFloat() * 23.0f };
        h.N = Vec3f{ g.RandomFloat() - 0.5f, g.RandomFloat() - 0.5f, g.RandomFloat() - 0.5f };
        h.GN = h.N;
        h.z = 1.0f + g.RandomFloat() * 30.0f;
        h.front = g.RandomFloat() < 0.7f;
        Ray r;
        r.dir = Vec3f{ g.RandomFloat() - 0.5f, g.RandomFloat() - 0.5f, g.RandomFloat() - 0.5f } * 3.0f;
        // view vector on the front side of N most of the time
        if ((r.dir % h.N) > 0.0f && (i % 5) != 0) r.dir = -r.dir;
        r.p = h.p - r.dir * h.z;
        Vec3f qdir = Vec3f{ g.RandomFloat() - 0.5f, g.RandomFloat() - 0.5f, g.RandomFloat() - 0.5f }.GetNormalized();
        float *I = &in[(size_t)i * 12];
        memcpy(I, &h.p.x, 12); memcpy(I + 3, &h.N.x, 12); memcpy(I + 6, &r.dir.x, 12); memcpy(I + 9, &qdir.x, 12);
        frontv[i] = h.front;

        RNG &rng = tileThreads::rng;
        SamplerInfo s{ rng };
        s.SetHit(r, h);
        rng.SetSequence((uint64_t)i + 1, seed);
        DirSampler::Info info; info.SetVoid();
        Vec3f d{ 0, 0, 0 };
        bool ok = light->GenerateSample(s, d, info);
        float *o = &lg[(size_t)i * 9];
        o[0] = d.x; o[1] = d.y; o[2] = d.z; o[3] = info.mult.r; o[4] = info.mult.g; o[5] = info.mult.b;
        o[6] = info.prob; o[7] = info.dist; o[8] = ok;
        info.SetVoid();
        // direction towards the light centre (inside the cone) for even i, arbitrary for odd i
        Vec3f toL = qdir;
        if (!(i & 1)) toL = d;
        light->GetSampleInfo(s, toL, info);
        li[i] = info.prob;
        if (!(i & 1)) memcpy(I + 9, &toL.x, 12);

        for (size_t m = 0; m < nm; ++m) {
            rng.SetSequence((uint64_t)i + 1, seed ^ (0x1000 + m));
            DirSampler::Info si; si.SetVoid();
            Vec3f bd{ 0, 0, 0 };
            bool b = sc.materials[m]->GenerateSample(s, bd, si);
            float *q = &mg[((size_t)i * nm + m) * 8];
            q[0] = bd.x; q[1] = bd.y; q[2] = bd.z; q[3] = si.mult.r; q[4] = si.mult.g; q[5] = si.mult.b; q[6] = si.prob; q[7] = 0;
            int32_t *f = &mflags[((size_t)i * nm + m) * 3];
            f[0] = b; f[1] = b ? (int)si.lobe : 0; f[2] = (int)rng.RandomInt(); // next draw pins the number of draws consumed
            if (!b) for (int k = 0; k < 7; ++k) q[k] = 0;
            DirSampler::Info gi; gi.SetVoid();
            sc.materials[m]->GetSampleInfo(s, toL, gi);
            mi[(size_t)i * nm + m] = gi.prob;
        }
    }
    dumpF("smp_in", in, {(size_t)n, 12}); dumpI("smp_front", frontv, {(size_t)n});
    dumpF("smp_light_gen", lg, {(size_t)n, 9}); dumpF("smp_light_pdf", li, {(size_t)n});
    dumpF("smp_mtl_gen", mg, {(size_t)n, nm, 8}); dumpI("smp_mtl_flags", mflags, {(size_t)n, nm, 3});
    dumpF("smp_mtl_pdf", mi, {(size_t)n, nm});
}

// per-sample radiance of tracePath for explicit rays (seeded per ray)
static void cmdKatPaths(int n, uint64_t seed, float extent)
{
    flattenScene();
    RNG g(55, seed);
    std::vector<float> rays((size_t)n * 6), rad((size_t)n * 3);
    const Camera &cam = renderer.GetCamera();
    for (int i = 0; i < n; ++i) {
        Ray r;
        r.p = cam.pos;
        Vec3f tgt{ (g.RandomFloat() - 0.5f) * extent, (g.RandomFloat() - 0.5f) * extent * 0.3f, g.RandomFloat() * extent * 0.5f };
        r.dir = tgt - r.p;
        memcpy(&rays[(size_t)i * 6], &r.p.x, 12); memcpy(&rays[(size_t)i * 6 + 3], &r.dir.x, 12);
        tileThreads::rng.SetSequence((uint64_t)i + 1, seed);
        Color c = tracePath(r);
        rad[(size_t)i * 3] = c.r; rad[(size_t)i * 3 + 1] = c.g; rad[(size_t)i * 3 + 2] = c.b;
    }
    dumpF("path_rays", rays, {(size_t)n, 6}); dumpF("path_radiance", rad, {(size_t)n, 3});
}

#endif // !REF_P12

static void cmdTime(int threads)
{
    // un-seeded, the reference's own thread pool and its compiled-in 128 spp
    setupCamera(0, 0);
    const Camera &cam = renderer.GetCamera();
    tileThreads::numTilesX = (cam.imgWidth + tileThreads::tileSize - 1) / tileThreads::tileSize;
    tileThreads::numTilesY = (cam.imgHeight + tileThreads::tileSize - 1) / tileThreads::tileSize;
    tileThreads::totalNumTiles = tileThreads::numTilesX * tileThreads::numTilesY;
    tileThreads::tileCounter = 0;
    auto t0 = std::chrono::high_resolution_clock::now();
    std::vector<std::thread> th;
    for (int i = 0; i < threads; ++i) th.emplace_back(threadRenderTiles);
    for (auto &t : th) t.join();
    auto t1 = std::chrono::high_resolution_clock::now();
    printf("{\"threads\": %d, \"ms\": %.3f, \"width\": %d, \"height\": %d, \"spp\": 128}\n", threads,
           std::chrono::duration<double, std::milli>(t1 - t0).count(), cam.imgWidth, cam.imgHeight);
}

int main(int argc, char **argv)
{
    if (argc < 4) {
        fprintf(stderr,
                "usage: ref_harness <scene.xml> <outdir> <cmd> [args]\n"
                "  scene                         dump flattened scene, meshes, BVHs\n"
                "  render seed min max delta W H [y0 y1]\n"
                "  kat_rng | kat_trace n seed extent [aimed] | kat_samplers n seed | kat_paths n seed extent\n"
                "  time threads\n"
                "  png file...                   (first argument ignored) decode PNG files to 8-bit RGB\n");
        return 1;
    }
    g_outdir = argv[2];
    std::string cmd = argv[3];
    if (cmd == "png") { // decode the files named after the command exactly as TextureFile::LoadFile does (texture.cpp:77-87)
        for (int i = 4; i < argc; ++i) {
            std::vector<unsigned char> d;
            unsigned w = 0, h = 0;
            const unsigned error = lodepng::decode(d, w, h, argv[i], LCT_RGB);
            char nm_[32];
            snprintf(nm_, sizeof nm_, "png%d", i - 4);
            if (error) { w = h = 0; d.clear(); }
            dumpB(nm_, d, {h, w, 3});
        }
        return 0;
    }
    if (cmd != "kat_rng") {
        if (!renderer.LoadScene(argv[1])) return 3;
        setupCamera(0, 0);
    }
    auto A = [&](int i, const char *def) { return std::string(argc > i ? argv[i] : def); };
    if (cmd == "scene") cmdDumpScene();
#ifdef REF_P12
    else if (cmd == "render_whitted") // seed min max delta W H max_bounces
        cmdRenderWhitted(strtoull(A(4, "7").c_str(), 0, 0), atoi(A(5, "16").c_str()), atoi(A(6, "16").c_str()), (float)atof(A(7, "0.01").c_str()),
                         atoi(A(8, "0").c_str()), atoi(A(9, "0").c_str()), atoi(A(10, "5").c_str()),
                         atoi(A(11, "0").c_str()), atoi(A(12, "0").c_str()), atoi(A(13, "64").c_str()));
    else if (cmd == "kat_photon") cmdKatPhoton(atoi(A(4, "1024").c_str()), strtoull(A(5, "1").c_str(), 0, 0));
    else if (cmd == "photons") // seed n_global n_caustic max_segments n_queries radius
        cmdPhotons(strtoull(A(4, "7").c_str(), 0, 0), atoi(A(5, "1000").c_str()), atoi(A(6, "1000").c_str()), atoi(A(7, "64").c_str()),
                   atoi(A(8, "256").c_str()), (float)atof(A(9, "3").c_str()));
#else
    else if (cmd == "render")
        cmdRender(strtoull(A(4, "7").c_str(), 0, 0), atoi(A(5, "16").c_str()), atoi(A(6, "16").c_str()), (float)atof(A(7, "0.01").c_str()),
                  atoi(A(8, "0").c_str()), atoi(A(9, "0").c_str()), atoi(A(10, "0").c_str()), atoi(A(11, "0").c_str()));
    else if (cmd == "kat_rng") cmdKatRng();
    else if (cmd == "kat_trace") cmdKatTrace(atoi(A(4, "4096").c_str()), strtoull(A(5, "1").c_str(), 0, 0), (float)atof(A(6, "40").c_str()), atoi(A(7, "0").c_str()));
    else if (cmd == "kat_samplers") cmdKatSamplers(atoi(A(4, "1024").c_str()), strtoull(A(5, "1").c_str(), 0, 0));
    else if (cmd == "kat_paths") cmdKatPaths(atoi(A(4, "4096").c_str()), strtoull(A(5, "1").c_str(), 0, 0), (float)atof(A(6, "40").c_str()));
#endif
    else if (cmd == "time") cmdTime(atoi(A(4, "8").c_str()));
    else { fprintf(stderr, "unknown command %s\n", cmd.c_str()); return 1; }
    return 0;
}
