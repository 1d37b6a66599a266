"""Python plumbing over the C ABI of libmcrt_hip.so (include/mcrt.h).

The product is the HIP library; this module only loads it with ctypes, mirrors
its entry points one to one and raises on every non-zero status.  There is no
CPU fallback: if the library is missing or no HIP device is present, calls
fail loudly.
"""
import ctypes as C
import os

import numpy as np

from . import abi
from .abi import (  # noqa: F401
    ACCEL_AUTO, ACCEL_BRUTE, ACCEL_FLAT, ACCEL_OCTREE, ACCEL_SCAN, TONEMAP_EXPOSURE, TONEMAP_NONE, TONEMAP_REINHARD, AccelParams, Camera, FlatScene, OctreeInfo,
)

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libmcrt_hip.so")

# every symbol include/mcrt.h declares
ABI_SYMBOLS = (
    "mcrt_create", "mcrt_destroy", "mcrt_last_error", "mcrt_set_stream", "mcrt_upload_scene", "mcrt_upload_scene_device_build", "mcrt_refit_scene", "mcrt_last_build_ms", "mcrt_set_camera", "mcrt_set_accel",
    "mcrt_camera_rays", "mcrt_intersect", "mcrt_intersect_device", "mcrt_intersect_noalloc", "mcrt_trace_samples", "mcrt_trace_rays", "mcrt_render_unlit", "mcrt_rng_seed",
    "mcrt_clear", "mcrt_dropped_samples", "mcrt_bind_accumulator", "mcrt_accumulator", "mcrt_render", "mcrt_last_render_ms", "mcrt_resolve",
    "mcrt_read_accumulator", "mcrt_write_accumulator", "mcrt_synchronize", "mcrt_octree_info_get", "mcrt_octree_download", "mcrt_set_counting", "mcrt_counters",
    "mcrt_lanes_per_pixel", "mcrt_set_lanes_per_pixel", "mcrt_reduce", "mcrt_comm_init_all", "mcrt_comm_destroy", "mcrt_add_accumulator",
    "mcrt_object_intersect", "mcrt_surface_normals", "mcrt_random_surface_points", "mcrt_brdf",
    "mcrt_camera_look_at", "mcrt_scene_image_create", "mcrt_scene_image_destroy", "mcrt_scene_image_info", "mcrt_scene_image_octree",
)

_lib = None


class McrtError(RuntimeError):
    def __init__(self, code, text):
        super().__init__(f"mcrt status {code}: {text}")
        self.code = code


def load_library(path=None):
    """dlopen the HIP library (needs libamdhip64, not a GPU).  MCRT_LIB points at another build for A/B timing."""
    global _lib
    if _lib is None:
        path = path or os.environ.get("MCRT_LIB") or LIB_PATH
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path} not built: run `python -c 'import __graft_entry__ as g; g.build()'` or make -C {HERE}/csrc")
        lib = C.CDLL(path)
        lib.mcrt_last_error.restype = C.c_char_p
        lib.mcrt_last_error.argtypes = [C.c_void_p]
        lib.mcrt_rng_seed.restype = None
        _lib = lib
    return _lib


def lanes_per_pixel(width, height, sample_count):
    lib = load_library()
    lib.mcrt_lanes_per_pixel.restype = C.c_uint32
    return int(lib.mcrt_lanes_per_pixel(C.c_uint32(width), C.c_uint32(height), C.c_uint32(sample_count)))


def rng_seed(seed, pixel, sample):
    out = (C.c_uint64 * 2)()
    load_library().mcrt_rng_seed(C.c_uint64(seed), C.c_uint32(pixel), C.c_uint32(sample), out)
    return int(out[0]), int(out[1])


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def host_camera(width, height, fov_y, position, look_at, up=(0.0, 1.0, 0.0)):
    """Camera ctor + SetView on the host (mcrt_camera_look_at)."""
    cam = Camera()
    rc = load_library().mcrt_camera_look_at(
        C.c_uint32(width), C.c_uint32(height), C.c_float(fov_y), _ptr(_f32(position), C.c_float), _ptr(_f32(look_at), C.c_float),
        _ptr(_f32(up), C.c_float), C.byref(cam),
    )
    if rc:
        raise McrtError(rc, (load_library().mcrt_last_error(None) or b"").decode())
    return cam


def build_scene_image(flat, accel=None):
    """Host-only flatten (no GPU): returns (OctreeInfo, node_box, node_words, prim_refs)."""
    lib = load_library()
    img = C.c_void_p()
    d = flat.desc()
    rc = lib.mcrt_scene_image_create(C.byref(d), C.byref(accel) if accel is not None else None, C.byref(img))
    if rc:
        raise McrtError(rc, (lib.mcrt_last_error(None) or b"").decode())
    try:
        info = OctreeInfo()
        lib.mcrt_scene_image_info(img, C.byref(info))
        box = np.zeros((info.node_count, 6), np.float32)
        words = np.zeros((info.node_count, 2), np.uint32)
        refs = np.zeros(info.prim_ref_count, np.uint32)
        lib.mcrt_scene_image_octree(img, _ptr(box, C.c_float), _ptr(words, C.c_uint32), _ptr(refs, C.c_uint32))
    finally:
        lib.mcrt_scene_image_destroy(img)
    return info, box, words, refs


def comm_init_all(devices):
    """ncclCommInitAll through the library (mcrt_comm_init_all): one communicator handle per listed device."""
    lib = load_library()
    n = len(devices)
    comms = (C.c_void_p * n)()
    rc = lib.mcrt_comm_init_all(C.c_int(n), (C.c_int * n)(*devices), comms)
    if rc:
        raise McrtError(rc, (lib.mcrt_last_error(None) or b"").decode())
    return [c for c in comms]


def comm_destroy(comm):
    load_library().mcrt_comm_destroy(C.c_void_p(comm))


COUNTER_NAMES = ("rays", "nodes", "tris", "spheres", "gates", "samples", "trace_calls", "splits")


class Backend:
    """One mcrt_ctx bound to one GPU."""

    def __init__(self, device=0):
        self.lib = load_library()
        self.ctx = C.c_void_p()
        rc = self.lib.mcrt_create(C.c_int(device), C.byref(self.ctx))
        if rc:
            raise McrtError(rc, (self.lib.mcrt_last_error(None) or b"").decode())
        self.width = self.height = 0
        self._lanes = 0

    def close(self):
        if getattr(self, "ctx", None) and self.ctx.value:
            self.lib.mcrt_destroy(self.ctx)
            self.ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc:
            raise McrtError(rc, (self.lib.mcrt_last_error(self.ctx) or b"").decode())

    # ---------------------------------------------------------------- setup
    def set_stream(self, stream_handle):
        self._ck(self.lib.mcrt_set_stream(self.ctx, C.c_void_p(stream_handle)))

    def upload_scene(self, flat, accel=None, device_build=False):
        """device_build: the octree is built by kernels on the GPU (mcrt_upload_scene_device_build) instead of on the host."""
        d = flat.desc()
        fn = self.lib.mcrt_upload_scene_device_build if device_build else self.lib.mcrt_upload_scene
        self._ck(fn(self.ctx, C.byref(d), C.byref(accel) if accel is not None else None))

    def refit_scene(self, flat, moved_objects):
        """mcrt_refit_scene: True when the tree was kept and refitted, False when the call fell back to a rebuild."""
        d = flat.desc()
        mv = np.ascontiguousarray(moved_objects, np.uint32)
        rebuilt = C.c_int(0)
        self._ck(self.lib.mcrt_refit_scene(self.ctx, C.byref(d), C.c_uint32(len(mv)), _ptr(mv, C.c_uint32), C.byref(rebuilt)))
        return rebuilt.value == 0

    def last_build_ms(self):
        ms = C.c_float()
        self._ck(self.lib.mcrt_last_build_ms(self.ctx, C.byref(ms)))
        return float(ms.value)

    def set_camera(self, cam):
        self._ck(self.lib.mcrt_set_camera(self.ctx, C.byref(cam)))
        self.width, self.height = int(cam.width), int(cam.height)

    def set_accel(self, mode):
        self._ck(self.lib.mcrt_set_accel(self.ctx, C.c_int(mode)))

    def set_counting(self, on):
        self._ck(self.lib.mcrt_set_counting(self.ctx, C.c_int(int(on))))

    def counters(self):
        out = (C.c_uint64 * 8)()
        self._ck(self.lib.mcrt_counters(self.ctx, out))
        return dict(zip(COUNTER_NAMES, (int(x) for x in out)))

    def octree_info(self):
        info = OctreeInfo()
        self._ck(self.lib.mcrt_octree_info_get(self.ctx, C.byref(info)))
        return info

    def octree_download(self):
        info = self.octree_info()
        box = np.zeros((info.node_count, 6), np.float32)
        words = np.zeros((info.node_count, 2), np.uint32)
        refs = np.zeros(info.prim_ref_count, np.uint32)
        self._ck(self.lib.mcrt_octree_download(self.ctx, _ptr(box, C.c_float), _ptr(words, C.c_uint32), _ptr(refs, C.c_uint32)))
        return box, words, refs

    # ---------------------------------------------------------- parity paths
    def camera_rays(self, xs, ys):
        xs, ys = _f32(xs), _f32(ys)
        o = np.zeros((len(xs), 3), np.float32)
        d = np.zeros((len(xs), 3), np.float32)
        self._ck(self.lib.mcrt_camera_rays(self.ctx, C.c_uint64(len(xs)), _ptr(xs, C.c_float), _ptr(ys, C.c_float), _ptr(o, C.c_float), _ptr(d, C.c_float)))
        return o, d

    def intersect(self, origins, dirs):
        o, d = _f32(origins).reshape(-1, 3), _f32(dirs).reshape(-1, 3)
        n = len(o)
        obj = np.zeros(n, np.int32); elem = np.zeros(n, np.uint32); t = np.zeros(n, np.float32)
        self._ck(self.lib.mcrt_intersect(self.ctx, C.c_uint64(n), _ptr(o, C.c_float), _ptr(d, C.c_float), _ptr(obj, C.c_int32), _ptr(elem, C.c_uint32), _ptr(t, C.c_float)))
        return obj, elem, t

    def intersect_noalloc(self, origins, dirs):
        """mcrt_intersect_noalloc: staging in the library's static device array, no device allocation."""
        o, d = _f32(origins).reshape(-1, 3), _f32(dirs).reshape(-1, 3)
        n = len(o)
        obj = np.zeros(n, np.int32); elem = np.zeros(n, np.uint32); t = np.zeros(n, np.float32)
        self._ck(self.lib.mcrt_intersect_noalloc(self.ctx, C.c_uint64(n), _ptr(o, C.c_float), _ptr(d, C.c_float), _ptr(obj, C.c_int32), _ptr(elem, C.c_uint32), _ptr(t, C.c_float)))
        return obj, elem, t

    def intersect_device(self, n, d_origins, d_dirs, d_obj, d_elem, d_t):
        """All arguments are raw device addresses (ints)."""
        self._ck(self.lib.mcrt_intersect_device(self.ctx, C.c_uint64(n), C.c_void_p(d_origins), C.c_void_p(d_dirs), C.c_void_p(d_obj), C.c_void_p(d_elem), C.c_void_p(d_t)))

    def object_intersect(self, object_index, origins, dirs):
        """Object::Intersects of one object of the scene: (hit bool array, elementIndex, hitDistance)."""
        o, d = _f32(origins).reshape(-1, 3), _f32(dirs).reshape(-1, 3)
        n = len(o)
        hit = np.zeros(n, np.uint8); elem = np.zeros(n, np.uint32); t = np.zeros(n, np.float32)
        self._ck(self.lib.mcrt_object_intersect(self.ctx, C.c_uint32(object_index), C.c_uint64(n), _ptr(o, C.c_float), _ptr(d, C.c_float), _ptr(hit, C.c_uint8), _ptr(elem, C.c_uint32), _ptr(t, C.c_float)))
        return hit.astype(bool), elem, t

    def surface_normals(self, objects, elems, positions):
        ob = np.ascontiguousarray(objects, np.uint32); el = np.ascontiguousarray(elems, np.uint32); pos = _f32(positions).reshape(-1, 3)
        out = np.zeros((len(ob), 3), np.float32)
        self._ck(self.lib.mcrt_surface_normals(self.ctx, C.c_uint64(len(ob)), _ptr(ob, C.c_uint32), _ptr(el, C.c_uint32), _ptr(pos, C.c_float), _ptr(out, C.c_float)))
        return out

    def random_surface_points(self, object_index, states):
        """Object::GetRandomPointOnSurface; returns (points, advanced generator states)."""
        st = np.array(states, dtype=np.uint64).reshape(-1, 2).copy()
        out = np.zeros((len(st), 3), np.float32)
        self._ck(self.lib.mcrt_random_surface_points(self.ctx, C.c_uint32(object_index), C.c_uint64(len(st)), _ptr(st, C.c_uint64), _ptr(out, C.c_float)))
        return out, st

    def brdf(self, material, incident, reflection, normal):
        i, r, n = _f32(incident).reshape(-1, 3), _f32(reflection).reshape(-1, 3), _f32(normal).reshape(-1, 3)
        out = np.zeros(len(i), np.float64)
        self._ck(self.lib.mcrt_brdf(self.ctx, C.byref(material), C.c_uint64(len(i)), _ptr(i, C.c_float), _ptr(r, C.c_float), _ptr(n, C.c_float), _ptr(out, C.c_double)))
        return out

    def trace_samples(self, pixels, samples, seed, max_depth):
        px = np.ascontiguousarray(pixels, np.uint32); sm = np.ascontiguousarray(samples, np.uint32)
        out = np.zeros((len(px), 3), np.float64)
        self._ck(self.lib.mcrt_trace_samples(self.ctx, C.c_uint64(len(px)), _ptr(px, C.c_uint32), _ptr(sm, C.c_uint32), C.c_uint64(seed), C.c_uint32(max_depth), _ptr(out, C.c_double)))
        return out

    def trace_rays(self, origins, dirs, states, max_depth, importance=None):
        """Scene::TraceRay on explicit rays; returns (radiance, advanced generator states)."""
        o, d = _f32(origins).reshape(-1, 3), _f32(dirs).reshape(-1, 3)
        st = np.array(states, dtype=np.uint64).reshape(-1, 2).copy()
        imp = None if importance is None else np.ascontiguousarray(importance, np.float64).reshape(-1, 3)
        out = np.zeros((len(o), 3), np.float64)
        self._ck(self.lib.mcrt_trace_rays(self.ctx, C.c_uint64(len(o)), _ptr(o, C.c_float), _ptr(d, C.c_float), _ptr(st, C.c_uint64), C.c_uint32(max_depth), _ptr(imp, C.c_double) if imp is not None else None, _ptr(out, C.c_double)))
        return out, st

    def render_unlit(self):
        out = np.zeros((self.height, self.width, 3), np.float64)
        self._ck(self.lib.mcrt_render_unlit(self.ctx, _ptr(out, C.c_double)))
        return out

    # ----------------------------------------------------------------- render
    def clear(self):
        self._ck(self.lib.mcrt_clear(self.ctx))

    def dropped_samples(self):
        n = C.c_uint64()
        self._ck(self.lib.mcrt_dropped_samples(self.ctx, C.byref(n)))
        return int(n.value)

    def bind_accumulator(self, device_address):
        self._ck(self.lib.mcrt_bind_accumulator(self.ctx, C.c_void_p(device_address)))

    def add_accumulator(self, device_address):
        self._ck(self.lib.mcrt_add_accumulator(self.ctx, C.c_void_p(device_address)))

    def reduce(self, comm, root=0):
        self._ck(self.lib.mcrt_reduce(self.ctx, C.c_void_p(comm), C.c_int(root)))

    def accumulator_address(self):
        p = C.c_void_p()
        self._ck(self.lib.mcrt_accumulator(self.ctx, C.byref(p)))
        return p.value

    def render(self, sample_begin, sample_count, max_depth, seed):
        self._ck(self.lib.mcrt_render(self.ctx, C.c_uint32(sample_begin), C.c_uint32(sample_count), C.c_uint32(max_depth), C.c_uint64(seed)))

    def lanes_per_pixel(self, sample_count):
        """Sample slices per pixel the next render(…, sample_count, …) pass uses (the summation order, see include/mcrt.h)."""
        if self._lanes:
            return min(self._lanes, int(sample_count))
        return lanes_per_pixel(self.width, self.height, sample_count)

    def set_lanes_per_pixel(self, lanes):
        self._ck(self.lib.mcrt_set_lanes_per_pixel(self.ctx, C.c_uint32(lanes)))
        self._lanes = int(lanes)

    def last_render_ms(self):
        ms = C.c_float()
        self._ck(self.lib.mcrt_last_render_ms(self.ctx, C.byref(ms)))
        return float(ms.value)

    def resolve(self, tonemap, spp_total, gamma=2.2, exposure=1.0, want_mean=True):
        bgra = np.zeros((self.height, self.width, 4), np.uint8)
        mean = np.zeros((self.height, self.width, 3), np.float64) if want_mean else None
        self._ck(self.lib.mcrt_resolve(self.ctx, C.c_int(tonemap), C.c_double(gamma), C.c_double(exposure), C.c_uint64(spp_total), _ptr(bgra, C.c_uint8), _ptr(mean, C.c_double) if want_mean else None))
        return bgra, mean

    def write_accumulator(self, sums):
        sums = np.ascontiguousarray(sums, np.float64)
        assert sums.size == self.width * self.height * 3
        self._ck(self.lib.mcrt_write_accumulator(self.ctx, _ptr(sums, C.c_double)))

    def read_accumulator(self):
        out = np.zeros((self.height, self.width, 3), np.float64)
        self._ck(self.lib.mcrt_read_accumulator(self.ctx, _ptr(out, C.c_double)))
        return out

    def synchronize(self):
        self._ck(self.lib.mcrt_synchronize(self.ctx))
