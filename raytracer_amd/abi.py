"""ctypes mirror of ``include/rt_mi355x.h`` (plain structs, no behaviour).

Kept in one place so the product binding (``raytracer_amd.lib``) and the test
infrastructure (the CPU oracle loaded by ``tests/``) describe scenes with the
same bytes.
"""
import ctypes as C

import numpy as np

RT_OK = 0
RT_ERR_INVALID, RT_ERR_NO_DEVICE, RT_ERR_HIP, RT_ERR_UNSUPPORTED, RT_ERR_IO, RT_ERR_STATE, RT_ERR_CANCELLED = -1, -2, -3, -4, -5, -6, -7
RT_NOALLOC_MAX_RAYS = 1 << 24  # rt_trace_brief_noalloc: rays per call
RT_PRIM_NONE, RT_PRIM_SPHERE, RT_PRIM_PLANE, RT_PRIM_MESH = 0, 1, 2, 3
RT_LIGHT_AMBIENT, RT_LIGHT_DIRECT, RT_LIGHT_POINT = 0, 1, 2
RT_MODE_PATH, RT_MODE_WHITTED = 0, 1

f32p = C.POINTER(C.c_float)
u32p = C.POINTER(C.c_uint32)
i32p = C.POINTER(C.c_int32)
u8p = C.POINTER(C.c_uint8)


class Node(C.Structure):
    _fields_ = [("tm", C.c_float * 12), ("itm", C.c_float * 12), ("parent", C.c_int32),
                ("prim", C.c_int32), ("mesh", C.c_int32), ("material", C.c_int32)]


class BvhNode(C.Structure):
    _fields_ = [("bounds", C.c_float * 6), ("data", C.c_uint32)]


class Mesh(C.Structure):
    _fields_ = [("v", f32p), ("nv", C.c_uint32), ("vn", f32p), ("nvn", C.c_uint32),
                ("vt", f32p), ("nvt", C.c_uint32), ("f", u32p), ("fn", u32p), ("ft", u32p),
                ("nf", C.c_uint32), ("bvh_nodes", C.POINTER(BvhNode)), ("n_bvh_nodes", C.c_uint32),
                ("bvh_elements", u32p), ("bvh_depth", C.c_uint32),
                ("face_material", u32p), ("n_sub_materials", C.c_uint32)]


class Material(C.Structure):
    _fields_ = [("diffuse", C.c_float * 3), ("specular", C.c_float * 3), ("glossiness", C.c_float),
                ("emission", C.c_float * 3), ("reflection", C.c_float * 3), ("refraction", C.c_float * 3),
                ("absorption", C.c_float * 3), ("ior", C.c_float), ("diffuse_texture", C.c_int32)]


class Texture(C.Structure):
    _fields_ = [("type", C.c_int32), ("itm", C.c_float * 12), ("color1", C.c_float * 3), ("color2", C.c_float * 3),
                ("pixels", C.POINTER(C.c_uint8)), ("width", C.c_int32), ("height", C.c_int32)]


class Light(C.Structure):
    _fields_ = [("type", C.c_int32), ("intensity", C.c_float * 3), ("position", C.c_float * 3),
                ("direction", C.c_float * 3), ("size", C.c_float), ("attenuation", C.c_float)]


class SceneDesc(C.Structure):
    _fields_ = [("nodes", C.POINTER(Node)), ("n_nodes", C.c_uint32),
                ("meshes", C.POINTER(Mesh)), ("n_meshes", C.c_uint32),
                ("materials", C.POINTER(Material)), ("n_materials", C.c_uint32),
                ("lights", C.POINTER(Light)), ("n_lights", C.c_uint32),
                ("background", C.c_float * 3), ("environment", C.c_float * 3),
                ("textures", C.POINTER(Texture)), ("n_textures", C.c_uint32),
                ("background_texture", C.c_int32), ("environment_texture", C.c_int32),
                ("face_materials", C.c_int32)]


class Camera(C.Structure):
    _fields_ = [("pos", C.c_float * 3), ("dir", C.c_float * 3), ("up", C.c_float * 3),
                ("fov", C.c_float), ("focaldist", C.c_float), ("dof", C.c_float),
                ("width", C.c_int32), ("height", C.c_int32), ("srgb", C.c_int32)]


class RenderParams(C.Structure):
    _fields_ = [("mode", C.c_int32), ("min_spp", C.c_int32), ("max_spp", C.c_int32),
                ("delta_max", C.c_float), ("max_bounces", C.c_int32), ("bias", C.c_float),
                ("halton_period", C.c_int32), ("tile_size", C.c_int32), ("tile_rank", C.c_int32),
                ("tile_world", C.c_int32), ("seed", C.c_uint64), ("photon_flags", C.c_int32),
                ("gather_radius", C.c_float)]


class PhotonParams(C.Structure):
    _fields_ = [("flags", C.c_int32), ("n_global", C.c_uint32), ("n_caustic", C.c_uint32),
                ("max_segments", C.c_int32), ("seed", C.c_uint64)]


PHOTON_DTYPE = np.dtype([("pos", "<f4", 3), ("power", "<f4"), ("dir", "<f4", 3), ("rgb", "u1", 3), ("plane", "u1")])
assert PHOTON_DTYPE.itemsize == 32
RT_PM_DIRECT, RT_PM_INDIRECT, RT_PM_CAUSTICS, RT_PM_MONTE_CARLO = 1, 2, 4, 8
RT_MAP_GLOBAL, RT_MAP_CAUSTICS = 0, 1
RT_TEX_CHECKER, RT_TEX_IMAGE = 1, 2


class FrameOut(C.Structure):
    _fields_ = [("rgb", C.c_void_p), ("z", C.c_void_p), ("sample_count", C.c_void_p),
                ("object_id", C.c_void_p), ("linear", C.c_void_p), ("device_memory", C.c_int32)]


class Hit(C.Structure):
    _fields_ = [("z", C.c_float), ("p", C.c_float * 3), ("n", C.c_float * 3), ("uvw", C.c_float * 3),
                ("node", C.c_int32), ("front", C.c_int32)]


class HitBrief(C.Structure):
    _fields_ = [("t", C.c_float), ("node", C.c_int32), ("face", C.c_uint32), ("front", C.c_int32),
                ("u", C.c_float), ("v", C.c_float)]


class Counters(C.Structure):
    _fields_ = [("closest_rays", C.c_uint64), ("shadow_rays", C.c_uint64), ("camera_paths", C.c_uint64),
                ("box_tests", C.c_uint64), ("tri_tests", C.c_uint64), ("gathers", C.c_uint64),
                ("photon_visits", C.c_uint64), ("kernel_ms", C.c_double),
                ("trace_kernel_ms", C.c_double), ("trace_launches", C.c_uint32), ("trace_pipeline", C.c_uint32),
                ("phase_cycles", C.c_uint64 * 16), ("bvh_build_ms", C.c_double)]


class ProgressInfo(C.Structure):
    _fields_ = [("pixels_total", C.c_uint64), ("pixels_done", C.c_uint64), ("round_items", C.c_uint64),
                ("round_items_done", C.c_uint64), ("round", C.c_int32), ("rendering", C.c_int32)]


HIT_DTYPE = np.dtype([("z", "<f4"), ("p", "<f4", 3), ("n", "<f4", 3), ("uvw", "<f4", 3),
                      ("node", "<i4"), ("front", "<i4")])
assert HIT_DTYPE.itemsize == C.sizeof(Hit)
# rt_debug_samplers: queries and answers (rt_sampler_query / rt_sampler_answer)
RT_SAMPLER_LIGHT, RT_SAMPLER_MATERIAL, RT_SAMPLER_PHOTON_MATERIAL, RT_SAMPLER_PHOTON_EMIT = 0, 1, 2, 3
SAMPLER_QUERY_DTYPE = np.dtype([("p", "<f4", 3), ("n", "<f4", 3), ("v", "<f4", 3), ("front", "<i4"), ("dir", "<f4", 3),
                                ("material", "<i4"), ("seq", "<u8"), ("seed", "<u8")])
SAMPLER_ANSWER_DTYPE = np.dtype([("dir", "<f4", 3), ("mult", "<f4", 3), ("prob", "<f4"), ("dist", "<f4"), ("ok", "<i4"),
                                 ("lobe", "<i4"), ("pdf", "<f4"), ("next_u32", "<u4"), ("origin", "<f4", 3), ("pad", "<f4")])
assert SAMPLER_QUERY_DTYPE.itemsize == 72 and SAMPLER_ANSWER_DTYPE.itemsize == 64
HIT_BRIEF_DTYPE = np.dtype([("t", "<f4"), ("node", "<i4"), ("face", "<u4"), ("front", "<i4"), ("u", "<f4"), ("v", "<f4")])
assert HIT_BRIEF_DTYPE.itemsize == C.sizeof(HitBrief)


def default_params(**kw):
    """The constants the reference compiles in (src/main.cpp:93,107,164,220,273-274,369)."""
    p = RenderParams(mode=RT_MODE_PATH, min_spp=128, max_spp=128, delta_max=0.01, max_bounces=50,
                     bias=0.002, halton_period=16, tile_size=16, tile_rank=0, tile_world=1, seed=7,
                     photon_flags=0, gather_radius=3.0)
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


def _ptr(a, typ):
    return a.ctypes.data_as(typ) if a is not None and a.size else C.cast(None, typ)


class SceneArrays:
    """A scene held as numpy arrays plus the ctypes ``SceneDesc`` pointing into them.

    ``nodes`` etc. are kept alive by this object; build one with
    :func:`scene_from_arrays` (tests, synthetic scenes) or obtain an equivalent
    description from the library's own scene.xml reader.
    """

    def __init__(self):
        self.keep = []
        self.desc = SceneDesc()
        self.camera = Camera()


def scene_from_arrays(node_tm, node_itm, node_parent, node_prim, node_mesh, node_material,
                      materials, lights, background=(0, 0, 0), meshes=(), camera=None):
    """Assemble a ``SceneDesc`` from flat arrays.

    materials: [n,20] = kd3 ks3 gloss emission3 reflection3 refraction3 absorption3 ior
    lights:    [n,12] = type intensity3 position3 direction3 size attenuation
    meshes:    iterable of dicts with v, vn, vt, f, fn, ft, bvh_nodes ([n,7] uint32 words),
               bvh_elements, bvh_depth
    camera:    15 floats = pos3 dir3 up3 fov focaldist dof width height srgb
    """
    s = SceneArrays()
    n = len(node_parent)
    nodes = (Node * n)()
    for i in range(n):
        nodes[i].tm[:] = [float(x) for x in np.asarray(node_tm[i], np.float32)]
        nodes[i].itm[:] = [float(x) for x in np.asarray(node_itm[i], np.float32)]
        nodes[i].parent, nodes[i].prim = int(node_parent[i]), int(node_prim[i])
        nodes[i].mesh, nodes[i].material = int(node_mesh[i]), int(node_material[i])
    mats = (Material * max(1, len(materials)))()
    for i, m in enumerate(np.asarray(materials, np.float32).reshape(-1, 20)):
        m = [float(x) for x in m]
        mats[i].diffuse[:] = m[0:3]; mats[i].specular[:] = m[3:6]; mats[i].glossiness = m[6]
        mats[i].emission[:] = m[7:10]; mats[i].reflection[:] = m[10:13]; mats[i].refraction[:] = m[13:16]
        mats[i].absorption[:] = m[16:19]; mats[i].ior = m[19]
    lts = (Light * max(1, len(lights)))()
    for i, l in enumerate(np.asarray(lights, np.float32).reshape(-1, 12)):
        l = [float(x) for x in l]
        lts[i].type = int(l[0]); lts[i].intensity[:] = l[1:4]; lts[i].position[:] = l[4:7]
        lts[i].direction[:] = l[7:10]; lts[i].size = l[10]; lts[i].attenuation = l[11]
    ms = (Mesh * max(1, len(meshes)))()
    for i, m in enumerate(meshes):
        a = {k: np.ascontiguousarray(m[k], np.float32) for k in ("v", "vn", "vt")}
        b = {k: np.ascontiguousarray(m[k], np.uint32) for k in ("f", "fn", "ft", "bvh_nodes", "bvh_elements")}
        s.keep += list(a.values()) + list(b.values())
        ms[i].v, ms[i].nv = _ptr(a["v"], f32p), a["v"].size // 3
        ms[i].vn, ms[i].nvn = _ptr(a["vn"], f32p), a["vn"].size // 3
        ms[i].vt, ms[i].nvt = _ptr(a["vt"], f32p), a["vt"].size // 3
        ms[i].f, ms[i].fn, ms[i].ft = _ptr(b["f"], u32p), _ptr(b["fn"], u32p), _ptr(b["ft"], u32p)
        ms[i].nf = b["f"].size // 3
        ms[i].bvh_nodes = C.cast(b["bvh_nodes"].ctypes.data, C.POINTER(BvhNode))
        ms[i].n_bvh_nodes = b["bvh_nodes"].size // 7
        ms[i].bvh_elements = _ptr(b["bvh_elements"], u32p)
        ms[i].bvh_depth = int(m.get("bvh_depth", 0)) or bvh_depth(b["bvh_nodes"].reshape(-1, 7))
        if m.get("face_material") is not None:
            fm = np.ascontiguousarray(m["face_material"], np.uint32)
            s.keep.append(fm)
            ms[i].face_material, ms[i].n_sub_materials = _ptr(fm, u32p), int(m["n_sub_materials"])
    s.keep += [nodes, mats, lts, ms]
    d = s.desc
    d.nodes, d.n_nodes = nodes, n
    d.meshes, d.n_meshes = ms, len(meshes)
    d.materials, d.n_materials = mats, len(np.asarray(materials).reshape(-1, 20))
    d.lights, d.n_lights = lts, len(np.asarray(lights).reshape(-1, 12))
    d.background[:] = [float(x) for x in background]
    d.environment[:] = [0.0, 0.0, 0.0]
    if camera is not None:
        set_camera(s.camera, camera)
    return s


def set_camera(cam, c15):
    c = [float(x) for x in np.asarray(c15, np.float32)]
    cam.pos[:] = c[0:3]; cam.dir[:] = c[3:6]; cam.up[:] = c[6:9]
    cam.fov, cam.focaldist, cam.dof = c[9], c[10], c[11]
    cam.width, cam.height, cam.srgb = int(c[12]), int(c[13]), int(c[14])
    return cam


def bvh_depth(words):
    """Depth of the deepest leaf of a BVH given as [n,7] uint32 words (root = node 1 = depth 0)."""
    depth, stack = 0, [(1, 0)]
    while stack:
        i, d = stack.pop()
        depth = max(depth, d)
        data = int(words[i, 6])
        if not data & 0x80000000:
            c = data & 0x7FFFFFFF
            stack += [(c, d + 1), (c + 1, d + 1)]
    return depth


def scene_from_fixture(g):
    """SceneArrays from the `scene_*` arrays of a fixture file (a flattened scene dump: node transforms, parents,
    primitives, materials, lights, camera, meshes with their BVHs)."""
    meshes = []
    k = 0
    while f"scene_mesh{k}_v" in g:
        meshes.append({n: g[f"scene_mesh{k}_{n}"] for n in ("v", "vn", "vt", "f", "fn", "ft", "bvh_nodes", "bvh_elements")})
        k += 1
    return scene_from_arrays(g["scene_node_tm"], g["scene_node_itm"], g["scene_node_parent"], g["scene_node_prim"],
                             g["scene_node_mesh"], g["scene_node_material"], g["scene_materials"], g["scene_lights"],
                             background=g["scene_background"], meshes=meshes, camera=g["scene_camera"])
