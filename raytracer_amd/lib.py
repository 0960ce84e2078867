"""ctypes binding of ``librt_mi355x.so`` - plumbing only, no arithmetic.

The shared library is built in-tree by ``__graft_entry__.build()`` (or
``make -C raytracer_amd/csrc``).  Loading fails loudly when it is missing:
there is no CPU fallback.
"""
import ctypes as C
import os

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
# RT_LIB selects another build of the same library (kernel tuning experiments only)
LIB_PATH = os.environ.get("RT_LIB") or os.path.join(_HERE, "librt_mi355x.so")

# every symbol include/rt_mi355x.h declares
SYMBOLS = (
    "rt_abi_version", "rt_init", "rt_shutdown", "rt_last_error", "rt_scene_upload", "rt_camera_set", "rt_render",
    "rt_trace", "rt_trace_shadow", "rt_trace_brief", "rt_trace_brief_noalloc", "rt_trace_paths", "rt_get_counters", "rt_set_counting",
    "rt_scene_load_xml", "rt_scene_from_xml_string", "rt_scene_host_desc", "rt_scene_host_camera",
    "rt_scene_host_free", "rt_scene_host_set_mesh", "rt_scene_host_set_face_materials", "rt_scene_host_set_gpu_bvh", "rt_scene_host_warning", "rt_png_decode_rgb8",
    "rt_bvh_build", "rt_free",
    "rt_cancel", "rt_progress", "rt_frame_records", "rt_frame_pack", "rt_frame_unpack", "rt_photons_emit", "rt_photons_build", "rt_photons_fill", "rt_photons_download", "rt_photons_irradiance",
    "rt_debug_samplers", "rt_debug_rng",
)

_lib = None


class RtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"rt error {code}: {msg}")
        self.code = code


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing - build it with `python -c 'import __graft_entry__ as g; g.build()'`")
    # PyTorch carries its own copy of the HIP runtime; of two copies in one process only the one loaded first sees the
    # GPU.  Where torch is installed it goes first, so that a later `import torch` (frame gather, benchmarks) still works.
    import importlib.util
    import sys
    if "torch" not in sys.modules and not os.environ.get("RT_NO_TORCH") and importlib.util.find_spec("torch") is not None:
        import torch  # noqa: F401
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.rt_abi_version.restype = C.c_int
    L.rt_init.argtypes = [C.c_int, C.POINTER(vp)]
    L.rt_shutdown.argtypes = [vp]
    L.rt_shutdown.restype = None
    L.rt_last_error.argtypes = [vp]
    L.rt_last_error.restype = C.c_char_p
    L.rt_scene_upload.argtypes = [vp, C.POINTER(abi.SceneDesc)]
    L.rt_camera_set.argtypes = [vp, C.POINTER(abi.Camera)]
    L.rt_render.argtypes = [vp, C.POINTER(abi.RenderParams), C.POINTER(abi.FrameOut)]
    L.rt_frame_records.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int]
    L.rt_frame_records.restype = C.c_uint32
    L.rt_frame_pack.argtypes = [vp, C.POINTER(abi.FrameOut), C.c_int, C.c_int, C.c_int, vp]
    L.rt_frame_unpack.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.POINTER(abi.FrameOut)]
    L.rt_cancel.argtypes = [vp]
    L.rt_progress.argtypes = [vp, C.POINTER(abi.ProgressInfo)]
    L.rt_trace.argtypes = [vp, abi.f32p, C.c_int, C.POINTER(abi.Hit)]
    L.rt_trace_shadow.argtypes = [vp, abi.f32p, abi.f32p, C.c_int, abi.i32p]
    L.rt_trace_brief.argtypes = [vp, abi.f32p, abi.f32p, C.c_int, C.POINTER(abi.HitBrief)]
    L.rt_trace_brief_noalloc.argtypes = [vp, vp, vp, C.c_int, vp]
    L.rt_trace_paths.argtypes = [vp, abi.f32p, C.c_int, C.c_uint64, C.POINTER(abi.RenderParams), abi.f32p]
    L.rt_get_counters.argtypes = [vp, C.POINTER(abi.Counters)]
    L.rt_set_counting.argtypes = [vp, C.c_int]
    L.rt_scene_load_xml.argtypes = [C.c_char_p, C.POINTER(vp), C.c_char_p, C.c_size_t]
    L.rt_scene_from_xml_string.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(vp), C.c_char_p, C.c_size_t]
    L.rt_scene_host_desc.argtypes = [vp]
    L.rt_scene_host_desc.restype = C.POINTER(abi.SceneDesc)
    L.rt_scene_host_camera.argtypes = [vp]
    L.rt_scene_host_camera.restype = C.POINTER(abi.Camera)
    L.rt_scene_host_free.argtypes = [vp]
    L.rt_scene_host_free.restype = None
    L.rt_scene_host_set_mesh.argtypes = [vp, C.c_char_p, abi.f32p, abi.f32p, C.c_uint32, abi.u32p, C.c_uint32,
                                         C.c_char_p, C.c_size_t]
    L.rt_scene_host_set_face_materials.argtypes = [vp, C.c_int]
    L.rt_scene_host_set_gpu_bvh.argtypes = [vp, C.c_int]
    L.rt_scene_host_warning.argtypes = [vp, C.c_uint32]
    L.rt_scene_host_warning.restype = C.c_char_p
    L.rt_png_decode_rgb8.argtypes = [C.c_char_p, C.c_size_t, abi.i32p, abi.i32p, C.POINTER(abi.u8p), C.c_char_p, C.c_size_t]
    L.rt_bvh_build.argtypes = [abi.f32p, C.c_uint32, abi.u32p, C.c_uint32, C.c_uint32,
                               C.POINTER(C.POINTER(abi.BvhNode)), C.POINTER(C.c_uint32),
                               C.POINTER(abi.u32p), C.POINTER(C.c_uint32)]
    L.rt_free.argtypes = [vp]
    L.rt_free.restype = None
    L.rt_photons_emit.argtypes = [vp, C.POINTER(abi.PhotonParams), C.c_int, C.c_uint64, C.c_uint64, vp, C.c_uint32,
                                  C.POINTER(C.c_uint64), C.c_int]
    L.rt_photons_build.argtypes = [vp, C.c_int, vp, C.c_uint32, C.c_int]
    L.rt_photons_fill.argtypes = [vp, C.POINTER(abi.PhotonParams)]
    L.rt_photons_download.argtypes = [vp, C.c_int, vp, C.POINTER(C.c_uint32)]
    L.rt_photons_irradiance.argtypes = [vp, C.c_int, abi.f32p, abi.f32p, C.c_int, C.c_float, abi.f32p]
    L.rt_debug_samplers.argtypes = [vp, C.c_int, vp, C.c_int, vp]
    L.rt_debug_rng.argtypes = [vp, vp, vp, C.c_int, C.c_int, vp, vp]
    _lib = L
    return L


class HostScene:
    """A scene read by the library's own scene.xml reader (host memory, no GPU needed)."""

    def __init__(self, handle):
        self._h = handle
        self.desc = load().rt_scene_host_desc(handle).contents
        self.camera = load().rt_scene_host_camera(handle).contents

    @classmethod
    def from_file(cls, path):
        h, err = C.c_void_p(), C.create_string_buffer(512)
        rc = load().rt_scene_load_xml(os.fsencode(path), C.byref(h), err, len(err))
        if rc:
            raise RtError(rc, err.value.decode())
        return cls(h)

    @classmethod
    def from_string(cls, xml, base_dir=""):
        h, err = C.c_void_p(), C.create_string_buffer(512)
        rc = load().rt_scene_from_xml_string(xml.encode(), os.fsencode(base_dir), C.byref(h), err, len(err))
        if rc:
            raise RtError(rc, err.value.decode())
        return cls(h)

    def set_mesh(self, node_name, v, vt, f):
        v = np.ascontiguousarray(v, np.float32)
        vt = np.ascontiguousarray(vt, np.float32)
        f = np.ascontiguousarray(f, np.uint32)
        err = C.create_string_buffer(512)
        rc = load().rt_scene_host_set_mesh(self._h, node_name.encode(), v.ctypes.data_as(abi.f32p),
                                           vt.ctypes.data_as(abi.f32p), len(v), f.ctypes.data_as(abi.u32p), len(f),
                                           err, len(err))
        if rc:
            raise RtError(rc, err.value.decode())
        self.desc = load().rt_scene_host_desc(self._h).contents

    def set_face_materials(self, on):
        """Shade the faces of an OBJ object that brings its own materials each with the one it names (the reference
        shades them all with the first: it never sets HitInfo::mtlID)."""
        load().rt_scene_host_set_face_materials(self._h, int(on))

    def set_gpu_bvh(self, on):
        """Describe the meshes without a BVH (rt_scene_upload then builds its own on the GPU) / with the reference's again."""
        rc = load().rt_scene_host_set_gpu_bvh(self._h, int(on))
        if rc:
            raise RtError(rc, "rt_scene_host_set_gpu_bvh")
        self.desc = load().rt_scene_host_desc(self._h).contents

    def warnings(self):
        out, n = [], 0
        while (w := load().rt_scene_host_warning(self._h, n)) is not None:
            out.append(w.decode())
            n += 1
        return out

    def close(self):
        if self._h:
            load().rt_scene_host_free(self._h)
            self._h = None

    def __del__(self):
        self.close()


def png_decode(data):
    """PNG file bytes -> uint8 array [height, width, 3] (rt_png_decode_rgb8)."""
    w, h, px, err = C.c_int32(), C.c_int32(), abi.u8p(), C.create_string_buffer(256)
    rc = load().rt_png_decode_rgb8(data, len(data), C.byref(w), C.byref(h), C.byref(px), err, len(err))
    if rc:
        raise RtError(rc, err.value.decode())
    try:
        return np.ctypeslib.as_array(px, shape=(h.value, w.value, 3)).copy()
    finally:
        load().rt_free(px)


def bvh_build(v, f, max_leaf=4):
    v = np.ascontiguousarray(v, np.float32)
    f = np.ascontiguousarray(f, np.uint32)
    nodes, elems = C.POINTER(abi.BvhNode)(), abi.u32p()
    nn, depth = C.c_uint32(), C.c_uint32()
    rc = load().rt_bvh_build(v.ctypes.data_as(abi.f32p), len(v), f.ctypes.data_as(abi.u32p), len(f), max_leaf,
                             C.byref(nodes), C.byref(nn), C.byref(elems), C.byref(depth))
    if rc:
        raise RtError(rc, "rt_bvh_build failed")
    try:
        words = np.ctypeslib.as_array(C.cast(nodes, abi.u32p), shape=(nn.value, 7)).copy()
        el = np.ctypeslib.as_array(elems, shape=(len(f),)).copy()
    finally:
        load().rt_free(nodes)
        load().rt_free(elems)
    return words, el, depth.value


class Frame:
    """Host output planes of one frame."""

    def __init__(self, w, h):
        self.rgb = np.zeros((h, w, 3), np.uint8)
        self.z = np.zeros((h, w), np.float32)
        self.sample_count = np.zeros((h, w), np.int32)
        self.object_id = np.full((h, w), -9, np.int32)
        self.linear = np.zeros((h, w, 3), np.float32)
        self.out = abi.FrameOut(self.rgb.ctypes.data, self.z.ctypes.data, self.sample_count.ctypes.data,
                                self.object_id.ctypes.data, self.linear.ctypes.data, 0)


class Context:
    """One GPU.  Mirrors the call order of the reference's Renderer: load scene, render, read images."""

    def __init__(self, device=0):
        self._L = load()
        self._h = C.c_void_p()
        self.width = self.height = 1
        rc = self._L.rt_init(device, C.byref(self._h))
        if rc:
            raise RtError(rc, self._L.rt_last_error(None).decode())

    def _check(self, rc):
        if rc:
            raise RtError(rc, self._L.rt_last_error(self._h).decode())

    def upload(self, desc, camera=None):
        self._check(self._L.rt_scene_upload(self._h, C.byref(desc)))
        if camera is not None:
            self.set_camera(camera)

    def set_camera(self, camera, width=None, height=None):
        cam = abi.Camera.from_buffer_copy(camera)
        if width:
            cam.width, cam.height = width, height
        self._check(self._L.rt_camera_set(self._h, C.byref(cam)))
        self.width, self.height = cam.width, cam.height

    def set_counting(self, on):
        self._check(self._L.rt_set_counting(self._h, int(on)))

    def render(self, params, frame=None):
        fr = frame or Frame(self.width, self.height)
        self._check(self._L.rt_render(self._h, C.byref(params), C.byref(fr.out)))
        return fr

    def render_device(self, params, rgb=0, z=0, sample_count=0, object_id=0, linear=0):
        """Render into device memory; arguments are device addresses (0 = plane not wanted)."""
        out = abi.FrameOut(rgb or None, z or None, sample_count or None, object_id or None, linear or None, 1)
        self._check(self._L.rt_render(self._h, C.byref(params), C.byref(out)))

    def frame_pack(self, planes, tile_size, rank, world, packed_ptr):
        """planes = (rgb, z, sample_count, object_id) device addresses -> 16-byte records of this rank's pixels at packed_ptr"""
        out = abi.FrameOut(planes[0], planes[1], planes[2], planes[3], None, 1)
        self._check(self._L.rt_frame_pack(self._h, C.byref(out), tile_size, rank, world, packed_ptr))

    def frame_unpack(self, packed_all_ptr, tile_size, skip_rank, world, planes):
        out = abi.FrameOut(planes[0], planes[1], planes[2], planes[3], None, 1)
        self._check(self._L.rt_frame_unpack(self._h, packed_all_ptr, tile_size, skip_rank, world, C.byref(out)))

    def cancel(self):
        """rt_cancel: callable from another thread while render() runs."""
        return self._L.rt_cancel(self._h)

    def progress(self):
        p = abi.ProgressInfo()
        rc = self._L.rt_progress(self._h, C.byref(p))
        if rc:
            raise RtError(rc, "rt_progress")
        return p

    def counters(self):
        c = abi.Counters()
        self._check(self._L.rt_get_counters(self._h, C.byref(c)))
        return c

    def trace(self, rays):
        rays = np.ascontiguousarray(rays, np.float32)
        hits = np.zeros(len(rays), abi.HIT_DTYPE)
        self._check(self._L.rt_trace(self._h, rays.ctypes.data_as(abi.f32p), len(rays),
                                     C.cast(hits.ctypes.data, C.POINTER(abi.Hit))))
        return hits

    def trace_shadow(self, rays, tmax):
        rays = np.ascontiguousarray(rays, np.float32)
        tmax = np.ascontiguousarray(tmax, np.float32)
        out = np.zeros(len(rays), np.int32)
        self._check(self._L.rt_trace_shadow(self._h, rays.ctypes.data_as(abi.f32p), tmax.ctypes.data_as(abi.f32p),
                                            len(rays), out.ctypes.data_as(abi.i32p)))
        return out

    def trace_brief(self, rays, tmax=None):
        """Closest hits (tmax None) or any-hit flags by the persistent ray-list kernel; see rt_trace_brief."""
        rays = np.ascontiguousarray(rays, np.float32)
        out = np.zeros(len(rays), abi.HIT_BRIEF_DTYPE)
        tm = None
        if tmax is not None:
            tm = np.ascontiguousarray(tmax, np.float32)
        self._check(self._L.rt_trace_brief(self._h, rays.ctypes.data_as(abi.f32p),
                                           tm.ctypes.data_as(abi.f32p) if tm is not None else None, len(rays),
                                           C.cast(out.ctypes.data, C.POINTER(abi.HitBrief))))
        return out

    def trace_brief_noalloc(self, d_rays6, d_t_max, n, d_out):
        """rt_trace_brief_noalloc: device addresses in, nothing allocated by the call."""
        self._check(self._L.rt_trace_brief_noalloc(self._h, d_rays6, d_t_max or None, n, d_out))

    def trace_paths(self, rays, seed, params):
        rays = np.ascontiguousarray(rays, np.float32)
        out = np.zeros((len(rays), 3), np.float32)
        self._check(self._L.rt_trace_paths(self._h, rays.ctypes.data_as(abi.f32p), len(rays), seed, C.byref(params),
                                           out.ctypes.data_as(abi.f32p)))
        return out

    # ---- known-answer queries (tests) ----
    def debug_samplers(self, which, queries):
        """rt_debug_samplers: queries = array of abi.SAMPLER_QUERY_DTYPE -> array of abi.SAMPLER_ANSWER_DTYPE."""
        q = np.ascontiguousarray(queries, abi.SAMPLER_QUERY_DTYPE)
        out = np.zeros(len(q), abi.SAMPLER_ANSWER_DTYPE)
        self._check(self._L.rt_debug_samplers(self._h, which, q.ctypes.data, len(q), out.ctypes.data))
        return out

    def debug_rng(self, seq, seed, n_draws):
        """rt_debug_rng: (words, floats), each [streams, n_draws]."""
        seq, seed = np.ascontiguousarray(seq, np.uint64), np.ascontiguousarray(seed, np.uint64)
        words = np.zeros((len(seq), n_draws), np.uint32)
        floats = np.zeros((len(seq), n_draws), np.float32)
        self._check(self._L.rt_debug_rng(self._h, seq.ctypes.data, seed.ctypes.data, len(seq), n_draws, words.ctypes.data, floats.ctypes.data))
        return words, floats

    # ---- photon maps ----
    def photons_emit(self, pp, which, path_begin, path_count, capacity, device_ptr=0):
        """Photons of paths [path_begin, path_begin + path_count) in path order.  Returns (photons, stored) with at
        most `capacity` photons; with device_ptr they are written to that device address and None is returned for them."""
        n = C.c_uint64()
        if device_ptr:
            self._check(self._L.rt_photons_emit(self._h, C.byref(pp), which, path_begin, path_count, device_ptr, capacity, C.byref(n), 1))
            return None, n.value
        out = np.zeros(max(1, capacity), abi.PHOTON_DTYPE)
        self._check(self._L.rt_photons_emit(self._h, C.byref(pp), which, path_begin, path_count, out.ctypes.data, capacity, C.byref(n), 0))
        return out[:min(n.value, capacity)], n.value

    def photons_build(self, which, photons=None, device_ptr=0, n=0):
        if device_ptr:
            self._check(self._L.rt_photons_build(self._h, which, device_ptr, n, 1))
            return
        photons = np.ascontiguousarray(photons, abi.PHOTON_DTYPE)
        self._check(self._L.rt_photons_build(self._h, which, photons.ctypes.data, len(photons), 0))

    def photons_fill(self, pp):
        self._check(self._L.rt_photons_fill(self._h, C.byref(pp)))

    def photons_download(self, which):
        n = C.c_uint32()
        self._check(self._L.rt_photons_download(self._h, which, None, C.byref(n)))
        out = np.zeros(n.value, abi.PHOTON_DTYPE)
        if n.value:
            self._check(self._L.rt_photons_download(self._h, which, out.ctypes.data, C.byref(n)))
        return out

    def photons_irradiance(self, which, pos, normal, radius):
        pos = np.ascontiguousarray(pos, np.float32)
        normal = np.ascontiguousarray(normal, np.float32)
        out = np.zeros((len(pos), 3), np.float32)
        self._check(self._L.rt_photons_irradiance(self._h, which, pos.ctypes.data_as(abi.f32p), normal.ctypes.data_as(abi.f32p),
                                                  len(pos), radius, out.ctypes.data_as(abi.f32p)))
        return out

    def close(self):
        if self._h:
            self._L.rt_shutdown(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
