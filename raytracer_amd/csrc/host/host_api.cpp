// extern "C" entry points that need no GPU: scene.xml / OBJ reading and the BVH build.
#include "png_reader.h"
#include "scene_host.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>

namespace {
void set_err(char *err, size_t n, const std::string &m)
{
    if (err && n) { snprintf(err, n, "%s", m.c_str()); }
}
} // namespace

extern "C" {

int rt_abi_version(void) { return RT_ABI_VERSION; }

int rt_scene_from_xml_string(const char *xml, const char *base_dir, rt_scene_host **out, char *err, size_t err_len)
{
    if (!xml || !out) { set_err(err, err_len, "null argument"); return RT_ERR_INVALID; }
    *out = nullptr;
    auto *s = new (std::nothrow) rt_scene_host();
    if (!s) return RT_ERR_INVALID;
    std::string e;
    bool ok = false;
    try {
        ok = rt::load_scene_xml(xml, base_dir ? base_dir : "", *s, e);
    } catch (const std::exception &ex) {
        e = ex.what();
    }
    if (!ok) {
        set_err(err, err_len, e);
        delete s;
        return RT_ERR_IO;
    }
    *out = s;
    return RT_OK;
}

int rt_scene_load_xml(const char *path, rt_scene_host **out, char *err, size_t err_len)
{
    if (!path || !out) { set_err(err, err_len, "null argument"); return RT_ERR_INVALID; }
    *out = nullptr;
    std::ifstream fp(path, std::ios::binary);
    if (!fp) { set_err(err, err_len, std::string("Failed to load the file \"") + path + "\""); return RT_ERR_IO; }
    std::stringstream ss;
    ss << fp.rdbuf();
    std::string p(path), dir;
    size_t slash = p.find_last_of('/');
    if (slash != std::string::npos) dir = p.substr(0, slash);
    return rt_scene_from_xml_string(ss.str().c_str(), dir.c_str(), out, err, err_len);
}

const rt_scene_desc *rt_scene_host_desc(const rt_scene_host *s) { return s ? &s->desc : nullptr; }
const rt_camera *rt_scene_host_camera(const rt_scene_host *s) { return s ? &s->camera : nullptr; }
void rt_scene_host_free(rt_scene_host *s) { delete s; }

int rt_scene_host_set_mesh(rt_scene_host *s, const char *node_name, const float *v, const float *vt,
                           uint32_t nv, const uint32_t *f, uint32_t nf, char *err, size_t err_len)
{
    if (!s || !node_name || !v || !vt || !f || nv == 0 || nf == 0) { set_err(err, err_len, "null or empty argument"); return RT_ERR_INVALID; }
    for (size_t i = 0; i < (size_t)nf * 3; ++i)
        if (f[i] >= nv) { set_err(err, err_len, "face index out of range"); return RT_ERR_INVALID; }
    int node = -1;
    for (size_t i = 0; i < s->nodes.size(); ++i)
        if (s->node_names[i] == node_name && s->nodes[i].prim == RT_PRIM_MESH) { node = (int)i; break; }
    if (node < 0) { set_err(err, err_len, std::string("no mesh node named ") + node_name); return RT_ERR_INVALID; }
    rt::HostMesh &m = *s->meshes[s->nodes[node].mesh];
    try {
        m.v.assign(v, v + (size_t)nv * 3);
        m.vt.assign(vt, vt + (size_t)nv * 3);
        m.f.assign(f, f + (size_t)nf * 3);
        m.ft = m.f;
        m.mtls.clear(); m.face_mtl.clear(); // an in-memory mesh brings no materials
        rt::compute_normals(m);
        if (s->gpu_bvh) { m.bvh.clear(); m.elements.clear(); m.depth = 0; }
        else rt::build_bvh(m.v.data(), m.f.data(), nf, 4, m.bvh, m.elements, m.depth);
    } catch (const std::exception &ex) {
        set_err(err, err_len, ex.what());
        return RT_ERR_INVALID;
    }
    s->refresh_desc();
    return RT_OK;
}

int rt_scene_host_set_gpu_bvh(rt_scene_host *s, int on)
{
    if (!s) return RT_ERR_INVALID;
    s->gpu_bvh = on != 0;
    try {
        for (auto &m : s->meshes) {
            if (s->gpu_bvh) { m->bvh.clear(); m->elements.clear(); m->depth = 0; }
            else if (m->bvh.empty() && m->nf() > 0) rt::build_bvh(m->v.data(), m->f.data(), m->nf(), 4, m->bvh, m->elements, m->depth);
        }
    } catch (const std::exception &) {
        return RT_ERR_INVALID;
    }
    s->refresh_desc();
    return RT_OK;
}

int rt_scene_host_set_face_materials(rt_scene_host *s, int on)
{
    if (!s) return RT_ERR_INVALID;
    s->desc.face_materials = on ? 1 : 0;
    return RT_OK;
}

const char *rt_scene_host_warning(const rt_scene_host *s, uint32_t n) { return s && n < s->warnings.size() ? s->warnings[n].c_str() : nullptr; }

int rt_png_decode_rgb8(const uint8_t *file, size_t size, int32_t *width, int32_t *height, uint8_t **rgb, char *err, size_t err_len)
{
    if (!file || !width || !height || !rgb) { set_err(err, err_len, "null argument"); return RT_ERR_INVALID; }
    std::vector<uint8_t> px;
    std::string why;
    int w = 0, h = 0;
    try {
        if (!rt::decode_png_rgb8(file, size, w, h, px, why)) { set_err(err, err_len, why); return RT_ERR_IO; }
    } catch (const std::exception &ex) {
        set_err(err, err_len, ex.what());
        return RT_ERR_IO;
    }
    *rgb = (uint8_t *)malloc(px.size() ? px.size() : 1);
    if (!*rgb) { set_err(err, err_len, "out of memory"); return RT_ERR_IO; }
    memcpy(*rgb, px.data(), px.size());
    *width = w; *height = h;
    return RT_OK;
}

int rt_bvh_build(const float *v, uint32_t nv, const uint32_t *f, uint32_t nf, uint32_t max_leaf,
                 rt_bvh_node **nodes, uint32_t *n_nodes, uint32_t **elements, uint32_t *depth)
{
    if (!v || !f || !nodes || !n_nodes || !elements || !depth || max_leaf == 0) return RT_ERR_INVALID;
    for (size_t i = 0; i < (size_t)nf * 3; ++i) if (f[i] >= nv) return RT_ERR_INVALID;
    std::vector<rt_bvh_node> nn;
    std::vector<uint32_t> ee;
    uint32_t d = 0;
    try {
        rt::build_bvh(v, f, nf, max_leaf, nn, ee, d);
    } catch (const std::exception &) {
        return RT_ERR_INVALID;
    }
    *nodes = (rt_bvh_node *)malloc(sizeof(rt_bvh_node) * (nn.size() ? nn.size() : 1));
    *elements = (uint32_t *)malloc(sizeof(uint32_t) * (ee.size() ? ee.size() : 1));
    if (!*nodes || !*elements) { free(*nodes); free(*elements); return RT_ERR_INVALID; }
    memcpy(*nodes, nn.data(), sizeof(rt_bvh_node) * nn.size());
    memcpy(*elements, ee.data(), sizeof(uint32_t) * ee.size());
    *n_nodes = (uint32_t)nn.size();
    *depth = d;
    return RT_OK;
}

void rt_free(void *p) { free(p); }

} // extern "C"
