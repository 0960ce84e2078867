// Host-side scene storage behind the opaque rt_scene_host handle: owns every
// array an rt_scene_desc points into.
#pragma once
#include "rt_mi355x.h"

#include <memory>
#include <string>
#include <vector>

namespace rt {

// One material of an OBJ file's material library, with the defaults of cyTriMesh::Mtl (cyTriMesh.h:75-104)
struct ObjMaterial {
    std::string name;
    float Ka[3] = {0, 0, 0}, Kd[3] = {1, 1, 1}, Ks[3] = {0, 0, 0}, Tf[3] = {0, 0, 0};
    float Ns = 0, Ni = 1;
    int illum = 2;
    std::string map_Kd, map_Ks;
};

struct HostMesh {
    std::string name;
    std::vector<float> v, vn, vt;       // 3 floats per entry
    std::vector<uint32_t> f, fn, ft;    // 3 indices per face
    std::vector<rt_bvh_node> bvh;       // node 0 unused, node 1 = root
    std::vector<uint32_t> elements;
    uint32_t depth = 0;
    std::vector<ObjMaterial> mtls;      // the materials the file's faces name (`usemtl`), in order of first use
    std::vector<uint32_t> face_mtl;     // per face: index into mtls (empty when the file names none)
    bool members_created = false;       // the scene loader has turned mtls into materials of the scene
    std::string dir;                    // directory of the OBJ file ("" = working directory), with its trailing separator

    uint32_t nf() const { return (uint32_t)(f.size() / 3); }
    rt_mesh desc() const;
};

// OBJ text -> mesh arrays, fan-triangulated, faces grouped by `usemtl` in order
// of first use (cyTriMesh.h:264-497), the named materials read from the file's `mtllib`
// libraries (cyTriMesh.h:499-545).  Normals are computed when the file has none.
bool load_obj(const std::string &path, HostMesh &m, std::string &err);
// Area-weighted vertex normals, one per position, fn = f (cyTriMesh.h:249-262).
void compute_normals(HostMesh &m);
// Mean-split BVH with at most max_leaf triangles per leaf (cyBVH.h:122-142,242-328,339-379).
void build_bvh(const float *v, const uint32_t *f, uint32_t nf, uint32_t max_leaf,
               std::vector<rt_bvh_node> &nodes, std::vector<uint32_t> &elements, uint32_t &depth);

} // namespace rt

struct rt_scene_host {
    std::vector<rt_node> nodes;
    std::vector<std::string> node_names;
    std::vector<std::unique_ptr<rt::HostMesh>> meshes;
    std::vector<rt_mesh> mesh_descs;
    std::vector<rt_material> materials;
    std::vector<std::string> material_names;
    std::vector<rt_light> lights;
    std::vector<rt_texture> textures;
    std::vector<std::vector<uint8_t>> texture_pixels; // per texture, empty for a checkerboard
    rt_scene_desc desc{};
    rt_camera camera{};
    std::vector<std::string> warnings;
    bool gpu_bvh = false; // meshes are described without a BVH: rt_scene_upload builds its own

    void refresh_desc();
};

namespace rt {
// scene.xml text -> scene (src/xmlload.cpp:41-270). base_dir resolves OBJ names
// that are not found relative to the working directory.
bool load_scene_xml(const std::string &xml, const std::string &base_dir, rt_scene_host &out, std::string &err);
} // namespace rt
