// Triangle-mesh input for the renderer: OBJ reader, vertex normals and the
// mean-split BVH.  The arrays produced here are, element for element, the ones
// the reference's TriObj::Load builds (src/objects.h:58-65), which is what
// makes triangle hits comparable bit for bit.
#include "scene_host.h"
#include "xform.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>

namespace rt {

rt_mesh HostMesh::desc() const
{
    rt_mesh d{};
    d.v = v.data(); d.nv = (uint32_t)(v.size() / 3);
    d.vn = vn.data(); d.nvn = (uint32_t)(vn.size() / 3);
    d.vt = vt.data(); d.nvt = (uint32_t)(vt.size() / 3);
    d.f = f.data(); d.fn = fn.data(); d.ft = ft.data(); d.nf = nf();
    d.bvh_nodes = bvh.empty() ? nullptr : bvh.data(); d.n_bvh_nodes = (uint32_t)bvh.size();
    d.bvh_elements = elements.empty() ? nullptr : elements.data(); d.bvh_depth = depth;
    if (!mtls.empty() && face_mtl.size() == nf()) { d.face_material = face_mtl.data(); d.n_sub_materials = (uint32_t)mtls.size(); }
    return d;
}

// ---------------------------------------------------------------------------
// OBJ
// ---------------------------------------------------------------------------
namespace {

// One logical line: leading blanks and '#' comment lines skipped, runs of blanks
// collapsed to one space (the reader in cyTriMesh.h:279-303 does the same).
bool next_line(FILE *fp, std::string &line)
{
    line.clear();
    int c = fgetc(fp);
    for (;;) {
        while (c != EOF && isspace(c)) c = fgetc(fp);
        if (c == '#') { while (c != EOF && c != '\n' && c != '\r') c = fgetc(fp); }
        else break;
    }
    bool inspace = false;
    while (c != EOF && c != '\n' && c != '\r' && c != '\0') {
        if (isspace(c)) inspace = true;
        else {
            if (inspace) line += ' ';
            inspace = false;
            line += (char)c;
        }
        c = fgetc(fp);
    }
    return !line.empty();
}

bool is_cmd(const std::string &l, const char *cmd)
{
    size_t n = strlen(cmd);
    return l.compare(0, n, cmd) == 0 && (l.size() == n || l[n] == ' ');
}

void read3(const std::string &l, size_t off, std::vector<float> &dst)
{
    float x = 0, y = 0, z = 0;
    sscanf(l.c_str() + off, "%f %f %f", &x, &y, &z);
    dst.push_back(x); dst.push_back(y); dst.push_back(z);
}

// three floats after a two-letter keyword; a single value stands for all three (cyTriMesh.h:306)
void read_rgb(const std::string &l, float c[3])
{
    c[0] = c[1] = c[2] = 0;
    if (l.size() <= 2) return;
    const int n = sscanf(l.c_str() + 2, "%f %f %f", &c[0], &c[1], &c[2]);
    if (n == 1) c[2] = c[1] = c[0];
}

std::string after(const std::string &l, size_t start)
{
    while (start < l.size() && l[start] <= ' ') ++start;
    return start < l.size() ? l.substr(start) : "";
}

// A material library (cyTriMesh.h:511-543): only materials some face names are kept; a library that cannot be
// opened leaves them at their defaults, as in the reference.
void load_mtl(const std::string &path, std::vector<ObjMaterial> &mtls)
{
    FILE *fp = fopen(path.c_str(), "r");
    if (!fp) return;
    std::string line;
    int cur = -1;
    while (next_line(fp, line)) {
        if (is_cmd(line, "newmtl")) {
            cur = -1;
            const std::string name = line.size() > 7 ? line.substr(7) : "";
            for (size_t i = 0; i < mtls.size(); ++i) if (mtls[i].name == name) { cur = (int)i; break; }
        } else if (cur >= 0) {
            ObjMaterial &m = mtls[(size_t)cur];
            if (is_cmd(line, "Ka")) read_rgb(line, m.Ka);
            else if (is_cmd(line, "Kd")) read_rgb(line, m.Kd);
            else if (is_cmd(line, "Ks")) read_rgb(line, m.Ks);
            else if (is_cmd(line, "Tf")) read_rgb(line, m.Tf);
            else if (is_cmd(line, "Ns")) { if (line.size() > 2) sscanf(line.c_str() + 2, "%f", &m.Ns); }
            else if (is_cmd(line, "Ni")) { if (line.size() > 2) sscanf(line.c_str() + 2, "%f", &m.Ni); }
            else if (is_cmd(line, "illum")) { if (line.size() > 5) sscanf(line.c_str() + 5, "%d", &m.illum); }
            else if (is_cmd(line, "map_Kd")) m.map_Kd = after(line, 7);
            else if (is_cmd(line, "map_Ks")) m.map_Ks = after(line, 7);
        }
    }
    fclose(fp);
}

} // namespace

bool load_obj(const std::string &path, HostMesh &m, std::string &err)
{
    FILE *fp = fopen(path.c_str(), "r");
    if (!fp) { err = "cannot open OBJ file " + path; return false; }
    std::vector<float> v, vt, vn;
    std::vector<uint32_t> f, ft, fn;
    std::vector<int> face_mtl;
    std::vector<std::string> mtl_names, mtl_libs;
    int cur_mtl = -1;
    std::string line;
    while (next_line(fp, line)) {
        if (is_cmd(line, "v")) read3(line, 2, v);
        else if (is_cmd(line, "vt")) read3(line, 2, vt);
        else if (is_cmd(line, "vn")) read3(line, 2, vn);
        else if (is_cmd(line, "f")) {
            // corner tokens a, a/b, a//c, a/b/c; negative = relative to the end; fan triangulation
            uint32_t cv[3] = {0, 0, 0}, ct[3] = {0, 0, 0}, cn[3] = {0, 0, 0};
            int corner = 0;
            const char *p = line.c_str() + 1;
            while (*p) {
                while (*p == ' ') ++p;
                if (!*p) break;
                long idx[3] = {0, 0, 0};
                bool have[3] = {false, false, false};
                int type = 0;
                while (*p && *p != ' ') {
                    if (*p == '/') { ++type; ++p; continue; }
                    char *end;
                    long val = strtol(p, &end, 10);
                    if (end == p) { ++p; continue; }
                    if (type < 3) { idx[type] = val; have[type] = true; }
                    p = end;
                }
                auto resolve = [](long i, size_t count) -> uint32_t {
                    return i < 0 ? (uint32_t)((long)count + i) : (uint32_t)(i - 1);
                };
                const uint32_t iv = have[0] ? resolve(idx[0], v.size() / 3) : 0;
                const uint32_t it = have[1] ? resolve(idx[1], vt.size() / 3) : 0;
                const uint32_t in = have[2] ? resolve(idx[2], vn.size() / 3) : 0;
                if (corner < 3) { cv[corner] = iv; ct[corner] = it; cn[corner] = in; }
                else { // further corners extend the fan
                    cv[1] = cv[2]; ct[1] = ct[2]; cn[1] = cn[2];
                    cv[2] = iv; ct[2] = it; cn[2] = in;
                }
                if (++corner >= 3) {
                    f.insert(f.end(), cv, cv + 3); ft.insert(ft.end(), ct, ct + 3); fn.insert(fn.end(), cn, cn + 3);
                    face_mtl.push_back(cur_mtl);
                }
            }
        } else if (is_cmd(line, "usemtl")) {
            const std::string name = line.size() > 7 ? line.substr(7) : "";
            if (name.empty()) cur_mtl = 0;
            else {
                cur_mtl = -1;
                for (size_t i = 0; i < mtl_names.size(); ++i) if (mtl_names[i] == name) cur_mtl = (int)i;
                if (cur_mtl < 0) { mtl_names.push_back(name); cur_mtl = (int)mtl_names.size() - 1; }
            }
        } else if (is_cmd(line, "mtllib") && line.size() > 7) mtl_libs.push_back(line.substr(7));
    }
    fclose(fp);
    if (f.empty()) { err = "OBJ file " + path + " has no faces"; return false; }
    const uint32_t nv = (uint32_t)(v.size() / 3);
    for (uint32_t i : f) if (i >= nv) { err = "OBJ face references a vertex that does not exist"; return false; }

    // faces grouped by material, in order of first use; unassigned faces last (cyTriMesh.h:468-491)
    if (!mtl_names.empty()) {
        std::vector<uint32_t> order;
        for (int mi = 0; mi < (int)mtl_names.size(); ++mi)
            for (uint32_t i = 0; i < face_mtl.size(); ++i) if (face_mtl[i] == mi) order.push_back(i);
        for (uint32_t i = 0; i < face_mtl.size(); ++i) if (face_mtl[i] < 0) order.push_back(i);
        auto permute = [&](std::vector<uint32_t> &a) {
            std::vector<uint32_t> o(a.size());
            for (size_t k = 0; k < order.size(); ++k) memcpy(&o[k * 3], &a[order[k] * 3], 12);
            a.swap(o);
        };
        permute(f); permute(ft); permute(fn);
        // a face's sub-material (cyTriMesh::GetMaterialIndex); faces outside every group count as the first
        m.face_mtl.resize(order.size());
        for (size_t k = 0; k < order.size(); ++k) m.face_mtl[k] = face_mtl[order[k]] < 0 ? 0u : (uint32_t)face_mtl[order[k]];
        const size_t slash = path.find_last_of("/\\");
        m.dir = slash == std::string::npos ? "" : path.substr(0, slash + 1);
        m.mtls.resize(mtl_names.size());
        for (size_t i = 0; i < mtl_names.size(); ++i) m.mtls[i].name = mtl_names[i];
        for (const std::string &lib : mtl_libs) load_mtl(m.dir + lib, m.mtls);
    }
    m.v.swap(v); m.f.swap(f);
    if (vt.empty()) { vt.assign(3, 0.0f); ft.assign(m.f.size(), 0u); }
    const uint32_t nvt = (uint32_t)(vt.size() / 3);
    for (uint32_t i : ft) if (i >= nvt) { err = "OBJ face references a texture vertex that does not exist"; return false; }
    m.vt.swap(vt); m.ft.swap(ft);
    if (vn.empty()) compute_normals(m);
    else {
        const uint32_t nvn = (uint32_t)(vn.size() / 3);
        for (uint32_t i : fn) if (i >= nvn) { err = "OBJ face references a normal that does not exist"; return false; }
        m.vn.swap(vn); m.fn.swap(fn);
    }
    return true;
}

void compute_normals(HostMesh &m)
{
    const size_t nv = m.v.size() / 3;
    m.vn.assign(nv * 3, 0.0f);
    m.fn = m.f;
    auto V = [&](uint32_t i) { return Vec3{m.v[i * 3], m.v[i * 3 + 1], m.v[i * 3 + 2]}; };
    for (size_t i = 0; i < m.f.size(); i += 3) {
        const uint32_t a = m.f[i], b = m.f[i + 1], c = m.f[i + 2];
        const Vec3 N = cross(V(b) - V(a), V(c) - V(a));
        for (uint32_t k : {a, b, c}) { m.vn[k * 3] += N.x; m.vn[k * 3 + 1] += N.y; m.vn[k * 3 + 2] += N.z; }
    }
    for (size_t i = 0; i < nv; ++i) {
        const Vec3 n = normalized(Vec3{m.vn[i * 3], m.vn[i * 3 + 1], m.vn[i * 3 + 2]});
        m.vn[i * 3] = n.x; m.vn[i * 3 + 1] = n.y; m.vn[i * 3 + 2] = n.z;
    }
}

// ---------------------------------------------------------------------------
// BVH
// ---------------------------------------------------------------------------
namespace {

struct Box6 {
    float b[6];
    Box6() { b[0] = b[1] = b[2] = 1e30f; b[3] = b[4] = b[5] = -1e30f; }
    void add(const Box6 &o)
    {
        for (int i = 0; i < 3; ++i) {
            if (b[i] > o.b[i]) b[i] = o.b[i];
            if (b[i + 3] < o.b[i + 3]) b[i + 3] = o.b[i + 3];
        }
    }
};

struct Builder {
    const float *v;
    const uint32_t *f;
    uint32_t max_leaf;
    std::vector<rt_bvh_node> &nodes;
    std::vector<uint32_t> &el;
    uint32_t depth = 0;
    static constexpr uint32_t kMaxElems = 8; // node format holds count-1 in 3 bits

    Box6 tri_box(uint32_t t) const
    {
        Box6 r;
        const float *p = &v[f[t * 3] * 3];
        r.b[0] = r.b[3] = p[0]; r.b[1] = r.b[4] = p[1]; r.b[2] = r.b[5] = p[2];
        for (int j = 1; j < 3; ++j) {
            const float *q = &v[f[t * 3 + j] * 3];
            for (int k = 0; k < 3; ++k) {
                if (r.b[k] > q[k]) r.b[k] = q[k];
                if (r.b[k + 3] < q[k]) r.b[k + 3] = q[k];
            }
        }
        return r;
    }
    float centre(uint32_t t, int dim) const
    {
        return (v[f[t * 3] * 3 + dim] + v[f[t * 3 + 1] * 3 + dim] + v[f[t * 3 + 2] * 3 + dim]) / 3.0f;
    }

    // partition e[0..count) about the box midpoint of the widest axis that separates them
    uint32_t mean_split(uint32_t count, uint32_t *e, const float *box) const
    {
        if (count <= max_leaf) return 0;
        const float d[3] = {box[3] - box[0], box[4] - box[1], box[5] - box[2]};
        uint32_t sd[3];
        sd[0] = d[0] >= d[1] ? (d[0] >= d[2] ? 0 : 2) : (d[1] >= d[2] ? 1 : 2);
        sd[1] = (sd[0] + 1) % 3;
        sd[2] = (sd[0] + 2) % 3;
        if (d[sd[1]] < d[sd[2]]) { uint32_t t = sd[1]; sd[1] = sd[2]; sd[2] = t; }
        for (int s = 0; s < 3; ++s) {
            const uint32_t dim = sd[s];
            const float split = 0.5f * (box[dim] + box[dim + 3]);
            uint32_t i = 0, j = count;
            while (i < j) {
                if (centre(e[i], (int)dim) <= split) ++i;
                else { --j; uint32_t t = e[i]; e[i] = e[j]; e[j] = t; }
            }
            if (i < count && i > 0) return i;
        }
        return 0;
    }

    // Node ids follow a depth-first walk in which the two children of a node take
    // the next two free slots before either subtree is numbered.
    void build(uint32_t id, uint32_t offset, uint32_t count, const Box6 &box, uint32_t &next_free, uint32_t level)
    {
        if (level > depth) depth = level;
        uint32_t *e = &el[offset];
        uint32_t n1 = mean_split(count, e, box.b);
        if (n1 == 0 || n1 >= count) {
            if (count > kMaxElems) n1 = count / 2;
            else {
                memcpy(nodes[id].bounds, box.b, sizeof box.b);
                nodes[id].data = (offset & 0x0FFFFFFFu) | ((count - 1) << 28) | 0x80000000u;
                return;
            }
        }
        Box6 b1, b2;
        for (uint32_t i = 0; i < n1; ++i) b1.add(tri_box(e[i]));
        for (uint32_t i = n1; i < count; ++i) b2.add(tri_box(e[i]));
        const uint32_t c = next_free;
        next_free += 2;
        if (nodes.size() < next_free) nodes.resize(next_free);
        memcpy(nodes[id].bounds, box.b, sizeof box.b);
        nodes[id].data = c & 0x7FFFFFFFu;
        build(c, offset, n1, b1, next_free, level + 1);
        build(c + 1, offset + n1, count - n1, b2, next_free, level + 1);
    }
};

} // namespace

void build_bvh(const float *v, const uint32_t *f, uint32_t nf, uint32_t max_leaf,
               std::vector<rt_bvh_node> &nodes, std::vector<uint32_t> &elements, uint32_t &depth)
{
    nodes.clear(); elements.clear(); depth = 0;
    if (nf == 0) return;
    if (max_leaf > Builder::kMaxElems) max_leaf = Builder::kMaxElems;
    elements.resize(nf);
    for (uint32_t i = 0; i < nf; ++i) elements[i] = i;
    nodes.reserve((size_t)nf);
    nodes.resize(2);
    memset(nodes.data(), 0, sizeof(rt_bvh_node) * 2);
    Builder b{v, f, max_leaf, nodes, elements};
    Box6 box;
    for (uint32_t i = 0; i < nf; ++i) box.add(b.tri_box(i));
    uint32_t next_free = 2;
    b.build(1, 0, nf, box, next_free, 0);
    nodes.resize(next_free);
    depth = b.depth;
}

} // namespace rt
