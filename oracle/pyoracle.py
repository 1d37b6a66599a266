"""TEST INFRASTRUCTURE: ctypes front-ends for the two CPU checkers.

  Oracle - oracle/_build/libmcrt_oracle.so, the plain-C restatement
           (mcrt_oracle.c);
  Ref    - oracle/_ref/libmcrt_ref.so, the reference's own translation units
           behind ref_harness.cpp (only where it has been built).

Both expose the same method names so tests can run one against the other.
Nothing in the product imports this module.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _load_abi():
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import mcrt_pkg

    return mcrt_pkg.load().abi


abi = _load_abi()

ORACLE_SO = os.path.join(HERE, "_build", "libmcrt_oracle.so")
REF_SO = os.path.join(HERE, "_ref", "libmcrt_ref.so")


def build(ref=True):
    """Compile the checkers (gcc only).  The reference build is attempted only
    where /root/reference exists (oracle/Makefile)."""
    targets = ["oracle"] + (["ref"] if ref else [])
    subprocess.run(["make", "-s", "-C", HERE] + targets, check=True)


class Options(C.Structure):
    _fields_ = [("portable_math", C.c_int32), ("forward_sum", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [
        (n, C.c_uint64)
        for n in (
            "rays", "gates", "tri_tests", "sphere_tests", "samples", "trace_calls", "splits", "max_pending",
            "max_depth_used", "nonfinite",
        )
    ]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class OctreeView(C.Structure):
    _fields_ = [("node_count", C.c_uint32), ("box", C.POINTER(C.c_float)), ("words", C.POINTER(C.c_uint32)), ("refs", C.POINTER(C.c_uint32)),
                ("root_cell", C.c_float * 6), ("pad", C.c_float), ("cell", C.POINTER(C.c_float))]


class OctreeStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("rays", "nodes", "tris", "spheres", "gates")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _u32(a):
    return np.ascontiguousarray(a, dtype=np.uint32)


def _u64(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def sample_states(lib_seed_fn, seed, pixels, samples):
    out = np.zeros((len(pixels), 2), dtype=np.uint64)
    tmp = (C.c_uint64 * 2)()
    for i, (p, s) in enumerate(zip(pixels, samples)):
        lib_seed_fn(C.c_uint64(seed), C.c_uint32(int(p)), C.c_uint32(int(s)), tmp)
        out[i, 0], out[i, 1] = tmp[0], tmp[1]
    return out


class Oracle:
    """The C restatement."""

    kind = "port"

    def __init__(self, path=ORACLE_SO):
        if not os.path.exists(path):
            build(ref=False)
        self.lib = C.CDLL(path)
        L = self.lib
        L.orc_scene_create.restype = C.c_void_p
        L.orc_scene_create.argtypes = [C.POINTER(abi.SceneDesc)]
        L.orc_scene_destroy.argtypes = [C.c_void_p]
        L.orc_render.restype = C.c_uint64
        L.orc_abi_sizes.restype = C.c_uint32
        sizes = (C.c_uint32 * 4)()
        stats_size = L.orc_abi_sizes(sizes)
        assert list(sizes) == [C.sizeof(abi.Material), C.sizeof(abi.Triangle), C.sizeof(abi.Object), C.sizeof(abi.Camera)]
        assert stats_size == C.sizeof(Stats)

    # -- scene / camera ---------------------------------------------------
    def scene(self, flat):
        d = flat.desc()
        h = self.lib.orc_scene_create(C.byref(d))
        return _Handle(h, self.lib.orc_scene_destroy, keep=flat)

    def scene_lights(self, scene):
        out = np.zeros(len(scene.keep.objects), dtype=np.int32)
        self.lib.orc_scene_lights.restype = C.c_uint32
        n = self.lib.orc_scene_lights(C.c_void_p(scene.h), _p(out, C.c_int32))
        return out[:n].copy()

    def camera(self, w, h, fov, pos, target, up=(0.0, 1.0, 0.0)):
        cam = abi.Camera()
        self.lib.orc_camera_make(
            C.c_uint32(w), C.c_uint32(h), C.c_float(fov), _p(_f32(pos), C.c_float), _p(_f32(target), C.c_float),
            _p(_f32(up), C.c_float), C.byref(cam),
        )
        return cam

    def camera_rays(self, cam, xs, ys):
        xs, ys = _f32(xs), _f32(ys)
        o = np.zeros((len(xs), 3), np.float32)
        d = np.zeros((len(xs), 3), np.float32)
        self.lib.orc_camera_rays(C.byref(cam), C.c_uint64(len(xs)), _p(xs, C.c_float), _p(ys, C.c_float), _p(o, C.c_float), _p(d, C.c_float))
        return o, d

    def triangle_normal(self, v0, v1, v2):
        out = np.zeros(3, np.float32)
        self.lib.orc_triangle_normal(_p(_f32(v0), C.c_float), _p(_f32(v1), C.c_float), _p(_f32(v2), C.c_float), _p(out, C.c_float))
        return out

    def mesh_aabb(self, position, triangles):
        tri = np.ascontiguousarray(triangles, dtype=abi.TRIANGLE_DTYPE)
        mn, mx = np.zeros(3, np.float32), np.zeros(3, np.float32)
        self.lib.orc_mesh_aabb(_p(_f32(position), C.c_float), tri.ctypes.data_as(C.POINTER(abi.Triangle)), C.c_uint32(len(tri)), _p(mn, C.c_float), _p(mx, C.c_float))
        return mn, mx

    # -- rng ----------------------------------------------------------------
    def sample_seed(self, seed, pixel, sample):
        out = (C.c_uint64 * 2)()
        self.lib.orc_sample_seed(C.c_uint64(seed), C.c_uint32(pixel), C.c_uint32(sample), out)
        return int(out[0]), int(out[1])

    def sample_states(self, seed, pixels, samples):
        return sample_states(self.lib.orc_sample_seed, seed, pixels, samples)

    def rng_doubles(self, s0, s1, n):
        out = np.zeros(n, np.float64)
        self.lib.orc_rng_doubles(C.c_uint64(s0), C.c_uint64(s1), C.c_uint64(n), _p(out, C.c_double))
        return out

    def rng_floats(self, s0, s1, n):
        out = np.zeros(n, np.float32)
        self.lib.orc_rng_floats(C.c_uint64(s0), C.c_uint64(s1), C.c_uint64(n), _p(out, C.c_float))
        return out

    def rng_floats_range(self, s0, s1, lo, hi, n):
        out = np.zeros(n, np.float32)
        self.lib.orc_rng_floats_range(C.c_uint64(s0), C.c_uint64(s1), C.c_float(lo), C.c_float(hi), C.c_uint64(n), _p(out, C.c_float))
        return out

    # -- pieces -------------------------------------------------------------
    def intersect(self, scene, origins, dirs, stats=None):
        o, d = _f32(origins).reshape(-1, 3), _f32(dirs).reshape(-1, 3)
        n = len(o)
        obj = np.zeros(n, np.int32); elem = np.zeros(n, np.uint32); t = np.zeros(n, np.float32)
        self.lib.orc_intersect(
            C.c_void_p(scene.h), C.c_uint64(n), _p(o, C.c_float), _p(d, C.c_float), _p(obj, C.c_int32), _p(elem, C.c_uint32),
            _p(t, C.c_float), C.byref(stats) if stats is not None else None,
        )
        return obj, elem, t

    def _tree(self, info, node_box, node_words, prim_refs):
        """OctreeView over the downloaded arrays; the node cells are derived from the root cell of `info` (mcrt_octree_info)."""
        box, words, refs = _f32(node_box), _u32(node_words), _u32(prim_refs)
        n = len(box.reshape(-1, 6))
        tree = OctreeView(n, _p(box, C.c_float), _p(words, C.c_uint32), _p(refs, C.c_uint32))
        tree.root_cell[:] = list(info.root_cell_min) + list(info.root_cell_max)
        tree.pad = float(info.pad)
        cells = np.zeros((n, 4), np.float32)
        tree.cell = None
        self.lib.orc_octree_cells(C.byref(tree), _p(cells, C.c_float))
        tree.cell = _p(cells, C.c_float)
        return tree, (box, words, refs, cells)

    def intersect_octree(self, scene, info, node_box, node_words, prim_refs, origins, dirs, stats=None):
        """Replay the product's flat octree walk on the CPU (leaf tests are the oracle's)."""
        o, d = _f32(origins).reshape(-1, 3), _f32(dirs).reshape(-1, 3)
        tree, keep = self._tree(info, node_box, node_words, prim_refs)
        n = len(o)
        obj = np.zeros(n, np.int32); elem = np.zeros(n, np.uint32); t = np.zeros(n, np.float32)
        self.lib.orc_intersect_octree.restype = C.c_int
        overflow = self.lib.orc_intersect_octree(
            C.c_void_p(scene.h), C.byref(tree), C.c_uint64(n), _p(o, C.c_float), _p(d, C.c_float), _p(obj, C.c_int32), _p(elem, C.c_uint32),
            _p(t, C.c_float), C.byref(stats) if stats is not None else None,
        )
        assert overflow == 0, "traversal stack overflow in the replay"
        return obj, elem, t

    def attach_octree(self, scene, info, node_box, node_words, prim_refs):
        """Every later nearest-hit query of `scene` (intersect, trace, render) walks this flat octree (copied)."""
        tree, keep = self._tree(info, node_box, node_words, prim_refs)
        self.lib.orc_scene_attach_octree(C.c_void_p(scene.h), C.byref(tree), C.c_uint32(len(keep[2])))

    def detach_octree(self, scene):
        self.lib.orc_scene_detach_octree(C.c_void_p(scene.h))

    def brdf(self, material, incident, reflection, normal, portable_math=False):
        i, r, n = _f32(incident).reshape(-1, 3), _f32(reflection).reshape(-1, 3), _f32(normal).reshape(-1, 3)
        out = np.zeros(len(i), np.float64)
        self.lib.orc_brdf(C.byref(material), C.c_uint64(len(i)), _p(i, C.c_float), _p(r, C.c_float), _p(n, C.c_float), C.c_int(int(portable_math)), _p(out, C.c_double))
        return out

    def hemisphere(self, normals, states, portable_math=False):
        nn, st = _f32(normals).reshape(-1, 3), _u64(states).reshape(-1, 2)
        d = np.zeros((len(nn), 3), np.float32); c = np.zeros(len(nn), np.float32)
        self.lib.orc_hemisphere(C.c_uint64(len(nn)), _p(nn, C.c_float), _p(st, C.c_uint64), C.c_int(int(portable_math)), _p(d, C.c_float), _p(c, C.c_float))
        return d, c

    def object_intersects(self, scene, index, origins, dirs):
        """Object::Intersects of one object: (hit, elementIndex, hitDistance) as the call leaves a fresh RayIntersectionInfo."""
        o, d = _f32(origins).reshape(-1, 3), _f32(dirs).reshape(-1, 3)
        n = len(o)
        hit = np.zeros(n, np.uint8); elem = np.zeros(n, np.uint32); t = np.zeros(n, np.float32)
        self.lib.orc_object_intersects(C.c_void_p(scene.h), C.c_uint32(index), C.c_uint64(n), _p(o, C.c_float), _p(d, C.c_float), _p(hit, C.c_uint8), _p(elem, C.c_uint32), _p(t, C.c_float))
        return hit.astype(bool), elem, t

    def object_random_points(self, scene, index, states, portable_math=False):
        """Object::GetRandomPointOnSurface, one point per generator state: (points, advanced states)."""
        st = np.array(states, dtype=np.uint64).reshape(-1, 2).copy()
        out = np.zeros((len(st), 3), np.float32)
        self.lib.orc_object_random_points(C.c_void_p(scene.h), C.c_uint32(index), C.c_uint64(len(st)), _p(st, C.c_uint64), C.c_int(int(portable_math)), _p(out, C.c_float))
        return out, st

    def surface_normal(self, scene, obj, elem, points):
        obj = np.ascontiguousarray(obj, np.int32); elem = _u32(elem); pts = _f32(points).reshape(-1, 3)
        out = np.zeros((len(obj), 3), np.float32)
        self.lib.orc_surface_normal(C.c_void_p(scene.h), C.c_uint64(len(obj)), _p(obj, C.c_int32), _p(elem, C.c_uint32), _p(pts, C.c_float), _p(out, C.c_float))
        return out

    def portable_math(self, which, xs):
        xs = _f32(xs)
        out = np.zeros_like(xs)
        self.lib.orc_portable_math(C.c_int({"sin": 0, "cos": 1, "tan": 2, "acos": 3}[which]), C.c_uint64(len(xs)), _p(xs, C.c_float), _p(out, C.c_float))
        return out

    # -- tracing ------------------------------------------------------------
    @staticmethod
    def _opts(portable_math, forward_sum):
        return Options(int(portable_math), int(forward_sum))

    def trace_rays(self, scene, origins, dirs, states, depth, portable_math=False, forward_sum=False, stats=None):
        o, d, st = _f32(origins).reshape(-1, 3), _f32(dirs).reshape(-1, 3), _u64(states).reshape(-1, 2)
        out = np.zeros((len(o), 3), np.float64)
        op = self._opts(portable_math, forward_sum)
        self.lib.orc_trace_rays(
            C.c_void_p(scene.h), C.c_uint64(len(o)), _p(o, C.c_float), _p(d, C.c_float), _p(st, C.c_uint64), C.c_uint32(depth),
            C.byref(op), _p(out, C.c_double), C.byref(stats) if stats is not None else None,
        )
        return out

    def trace_samples(self, scene, cam, pixels, samples, seed, depth, portable_math=False, forward_sum=False, stats=None):
        px, sm = _u32(pixels), _u32(samples)
        out = np.zeros((len(px), 3), np.float64)
        op = self._opts(portable_math, forward_sum)
        self.lib.orc_trace_samples(
            C.c_void_p(scene.h), C.byref(cam), C.c_uint64(len(px)), _p(px, C.c_uint32), _p(sm, C.c_uint32), C.c_uint64(seed),
            C.c_uint32(depth), C.byref(op), _p(out, C.c_double), C.byref(stats) if stats is not None else None,
        )
        return out

    def render(self, scene, cam, sample_begin, sample_count, depth, seed, threads=1, lanes_per_pixel=1, portable_math=False,
               forward_sum=False, sums=None, stats=None):
        if sums is None:
            sums = np.zeros((cam.height, cam.width, 3), np.float64)
        op = self._opts(portable_math, forward_sum)
        dropped = self.lib.orc_render(
            C.c_void_p(scene.h), C.byref(cam), C.c_uint32(sample_begin), C.c_uint32(sample_count), C.c_uint32(depth), C.c_uint64(seed),
            C.c_uint32(threads), C.c_uint32(lanes_per_pixel), C.byref(op), _p(sums, C.c_double), C.byref(stats) if stats is not None else None,
        )
        return sums, int(dropped)

    def render_unlit(self, scene, cam):
        out = np.zeros((cam.height, cam.width, 3), np.float64)
        self.lib.orc_render_unlit(C.c_void_p(scene.h), C.byref(cam), _p(out, C.c_double))
        return out

    def resolve(self, sums, spp, tonemap, gamma=2.2, exposure=1.0, portable_math=False):
        """portable_math: pow / exp through the fixed-order kernels k_resolve runs (byte-exact against the GPU)."""
        sums = np.ascontiguousarray(sums, np.float64)
        npix = sums.size // 3
        bgra = np.zeros((npix, 4), np.uint8); mean = np.zeros((npix, 3), np.float64)
        self.lib.orc_resolve_ex(C.c_uint64(npix), _p(sums, C.c_double), C.c_uint64(spp), C.c_int(tonemap), C.c_double(gamma), C.c_double(exposure),
                                C.c_int(1 if portable_math else 0), _p(bgra, C.c_uint8), _p(mean, C.c_double))
        return bgra, mean

    def portable_exp_log_pow(self, which, a, b=None):
        """pm_exp (0) / pm_log (1) / pm_pow (2) of the oracle, elementwise on doubles."""
        a = np.ascontiguousarray(a, np.float64)
        b = np.ascontiguousarray(a if b is None else b, np.float64)
        out = np.empty_like(a)
        self.lib.orc_portable_exp_log_pow(C.c_int(which), C.c_uint64(a.size), _p(a, C.c_double), _p(b, C.c_double), _p(out, C.c_double))
        return out


class _Handle:
    def __init__(self, h, destroy, keep=None):
        self.h, self._destroy, self.keep = h, destroy, keep

    def __del__(self):
        try:
            if self.h:
                self._destroy(C.c_void_p(self.h))
        except Exception:
            pass
        self.h = None


class Ref:
    """The reference's own code (oracle/_ref)."""

    kind = "reference"

    @staticmethod
    def available():
        return os.path.exists(REF_SO)

    def __init__(self, path=REF_SO):
        self.lib = C.CDLL(path)
        L = self.lib
        for fn in ("ref_scene_create_cornell", "ref_scene_create_hexagon", "ref_scene_create_from_desc", "ref_camera_create"):
            getattr(L, fn).restype = C.c_void_p
        L.ref_render.restype = C.c_uint64

    # -- scenes ---------------------------------------------------------------
    def cornell(self, length=10.0, width=10.0, height=10.0, radius=1.5, light=10.0, point=False):
        h = self.lib.ref_scene_create_cornell(C.c_float(length), C.c_float(width), C.c_float(height), C.c_float(radius), C.c_double(light), C.c_int(int(point)))
        return _Handle(h, self.lib.ref_scene_destroy)

    def hexagon(self, radius=1.5, light=10.0, point=False):
        h = self.lib.ref_scene_create_hexagon(C.c_float(radius), C.c_double(light), C.c_int(int(point)))
        return _Handle(h, self.lib.ref_scene_destroy)

    def scene(self, flat, reference_octree=True):
        d = flat.desc()
        h = self.lib.ref_scene_create_from_desc(C.byref(d), C.c_int(int(reference_octree)))
        return _Handle(h, self.lib.ref_scene_destroy, keep=flat)

    def dump(self, scene):
        no, nt, nl = C.c_uint32(), C.c_uint32(), C.c_uint32()
        self.lib.ref_scene_counts(C.c_void_p(scene.h), C.byref(no), C.byref(nt), C.byref(nl))
        objs = np.zeros(no.value, abi.OBJECT_DTYPE); tris = np.zeros(nt.value, abi.TRIANGLE_DTYPE)
        bg = (C.c_double * 3)()
        self.lib.ref_scene_dump(C.c_void_p(scene.h), objs.ctypes.data_as(C.POINTER(abi.Object)), tris.ctypes.data_as(C.POINTER(abi.Triangle)), bg)
        lights = np.zeros(nl.value, np.int32)
        self.lib.ref_scene_lights(C.c_void_p(scene.h), _p(lights, C.c_int32))
        return abi.FlatScene(objs, tris, tuple(bg)), lights

    def scene_lights(self, scene):
        return self.dump(scene)[1]

    # -- camera -----------------------------------------------------------------
    def camera_handle(self, w, h, fov, pos, target, up=(0.0, 1.0, 0.0)):
        c = self.lib.ref_camera_create(C.c_uint32(w), C.c_uint32(h), C.c_float(fov), _p(_f32(pos), C.c_float), _p(_f32(target), C.c_float), _p(_f32(up), C.c_float))
        return _Handle(c, self.lib.ref_camera_destroy)

    def camera(self, w, h, fov, pos, target, up=(0.0, 1.0, 0.0)):
        hd = self.camera_handle(w, h, fov, pos, target, up)
        cam = abi.Camera()
        self.lib.ref_camera_dump(C.c_void_p(hd.h), C.byref(cam))
        cam._ref_handle = hd
        cam._ref_args = (w, h, fov, tuple(pos), tuple(target), tuple(up))
        return cam

    def _cam_handle(self, cam):
        return cam._ref_handle

    def camera_rays(self, cam, xs, ys):
        xs, ys = _f32(xs), _f32(ys)
        o = np.zeros((len(xs), 3), np.float32); d = np.zeros((len(xs), 3), np.float32)
        self.lib.ref_camera_rays(C.c_void_p(self._cam_handle(cam).h), C.c_uint64(len(xs)), _p(xs, C.c_float), _p(ys, C.c_float), _p(o, C.c_float), _p(d, C.c_float))
        return o, d

    # -- rng ----------------------------------------------------------------
    def sample_seed(self, seed, pixel, sample):
        out = (C.c_uint64 * 2)()
        self.lib.ref_sample_seed(C.c_uint64(seed), C.c_uint32(pixel), C.c_uint32(sample), out)
        return int(out[0]), int(out[1])

    def sample_states(self, seed, pixels, samples):
        return sample_states(self.lib.ref_sample_seed, seed, pixels, samples)

    def rng_doubles(self, s0, s1, n):
        out = np.zeros(n, np.float64)
        self.lib.ref_rng_doubles(C.c_uint64(s0), C.c_uint64(s1), C.c_uint64(n), _p(out, C.c_double))
        return out

    def rng_floats(self, s0, s1, n):
        out = np.zeros(n, np.float32)
        self.lib.ref_rng_floats(C.c_uint64(s0), C.c_uint64(s1), C.c_uint64(n), _p(out, C.c_float))
        return out

    def rng_floats_range(self, s0, s1, lo, hi, n):
        out = np.zeros(n, np.float32)
        self.lib.ref_rng_floats_range(C.c_uint64(s0), C.c_uint64(s1), C.c_float(lo), C.c_float(hi), C.c_uint64(n), _p(out, C.c_float))
        return out

    # -- pieces -------------------------------------------------------------
    def intersect(self, scene, origins, dirs, stats=None):
        o, d = _f32(origins).reshape(-1, 3), _f32(dirs).reshape(-1, 3)
        n = len(o)
        obj = np.zeros(n, np.int32); elem = np.zeros(n, np.uint32); t = np.zeros(n, np.float32)
        self.lib.ref_intersect(C.c_void_p(scene.h), C.c_uint64(n), _p(o, C.c_float), _p(d, C.c_float), _p(obj, C.c_int32), _p(elem, C.c_uint32), _p(t, C.c_float))
        return obj, elem, t

    def brdf(self, material, incident, reflection, normal, portable_math=False):
        assert not portable_math
        i, r, n = _f32(incident).reshape(-1, 3), _f32(reflection).reshape(-1, 3), _f32(normal).reshape(-1, 3)
        out = np.zeros(len(i), np.float64)
        self.lib.ref_brdf(C.byref(material), C.c_uint64(len(i)), _p(i, C.c_float), _p(r, C.c_float), _p(n, C.c_float), _p(out, C.c_double))
        return out

    def hemisphere(self, normals, states, portable_math=False):
        assert not portable_math
        nn, st = _f32(normals).reshape(-1, 3), _u64(states).reshape(-1, 2)
        d = np.zeros((len(nn), 3), np.float32); c = np.zeros(len(nn), np.float32)
        self.lib.ref_hemisphere(C.c_uint64(len(nn)), _p(nn, C.c_float), _p(st, C.c_uint64), _p(d, C.c_float), _p(c, C.c_float))
        return d, c

    def object_intersects(self, scene, index, origins, dirs):
        o, d = _f32(origins).reshape(-1, 3), _f32(dirs).reshape(-1, 3)
        n = len(o)
        hit = np.zeros(n, np.uint8); elem = np.zeros(n, np.uint32); t = np.zeros(n, np.float32)
        self.lib.ref_object_intersects(C.c_void_p(scene.h), C.c_uint32(index), C.c_uint64(n), _p(o, C.c_float), _p(d, C.c_float), _p(hit, C.c_uint8), _p(elem, C.c_uint32), _p(t, C.c_float))
        return hit.astype(bool), elem, t

    def object_random_points(self, scene, index, states, portable_math=False):
        assert not portable_math
        st = np.array(states, dtype=np.uint64).reshape(-1, 2).copy()
        out = np.zeros((len(st), 3), np.float32)
        self.lib.ref_object_random_points(C.c_void_p(scene.h), C.c_uint32(index), C.c_uint64(len(st)), _p(st, C.c_uint64), _p(out, C.c_float))
        return out, st

    def surface_normal(self, scene, obj, elem, points):
        obj = np.ascontiguousarray(obj, np.int32); elem = _u32(elem); pts = _f32(points).reshape(-1, 3)
        out = np.zeros((len(obj), 3), np.float32)
        self.lib.ref_surface_normal(C.c_void_p(scene.h), C.c_uint64(len(obj)), _p(obj, C.c_int32), _p(elem, C.c_uint32), _p(pts, C.c_float), _p(out, C.c_float))
        return out

    # -- tracing ------------------------------------------------------------
    def trace_rays(self, scene, origins, dirs, states, depth, **_):
        o, d, st = _f32(origins).reshape(-1, 3), _f32(dirs).reshape(-1, 3), _u64(states).reshape(-1, 2)
        out = np.zeros((len(o), 3), np.float64)
        self.lib.ref_trace_rays(C.c_void_p(scene.h), C.c_uint64(len(o)), _p(o, C.c_float), _p(d, C.c_float), _p(st, C.c_uint64), C.c_uint32(depth), _p(out, C.c_double))
        return out

    def trace_samples(self, scene, cam, pixels, samples, seed, depth, **_):
        px, sm = _u32(pixels), _u32(samples)
        out = np.zeros((len(px), 3), np.float64)
        self.lib.ref_trace_samples(C.c_void_p(scene.h), C.c_void_p(self._cam_handle(cam).h), C.c_uint64(len(px)), _p(px, C.c_uint32), _p(sm, C.c_uint32), C.c_uint64(seed), C.c_uint32(depth), _p(out, C.c_double))
        return out

    def render(self, scene, cam, sample_begin, sample_count, depth, seed, threads=1, sums=None, **_):
        if sums is None:
            sums = np.zeros((cam.height, cam.width, 3), np.float64)
        dropped = self.lib.ref_render(C.c_void_p(scene.h), C.c_void_p(self._cam_handle(cam).h), C.c_uint32(sample_begin), C.c_uint32(sample_count), C.c_uint32(depth), C.c_uint64(seed), C.c_uint32(threads), _p(sums, C.c_double))
        return sums, int(dropped)

    def render_unlit(self, scene, cam):
        out = np.zeros((cam.height, cam.width, 3), np.float64)
        self.lib.ref_render_unlit(C.c_void_p(scene.h), C.c_void_p(self._cam_handle(cam).h), _p(out, C.c_double))
        return out

    def resolve(self, sums, spp, tonemap, gamma=2.2, exposure=1.0):
        sums = np.ascontiguousarray(sums, np.float64)
        npix = sums.size // 3
        bgra = np.zeros((npix, 4), np.uint8); mean = np.zeros((npix, 3), np.float64)
        self.lib.ref_resolve(C.c_uint64(npix), _p(sums, C.c_double), C.c_uint64(spp), C.c_int(tonemap), C.c_double(gamma), C.c_double(exposure), _p(bgra, C.c_uint8), _p(mean, C.c_double))
        return bgra, mean
