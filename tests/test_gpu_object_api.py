"""The per-object virtuals of the reference's class Object (source/objects/object.h:23-29) and Material::BRDF
(source/core/material.h:24-53) through the C ABI (mcrt_object_intersect, mcrt_surface_normals, mcrt_random_surface_points,
mcrt_brdf) and through the C++ host mirror, against fixtures the reference's own virtuals produced (oracle/gen_golden.py F9, F7)."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

NAMES = ["cornell", "cornell_point", "cornell_sphere_light", "hexagon"]


@pytest.mark.parametrize("name", NAMES)
def test_object_intersects_and_normals_bit_exact(backend, pkg, scene_of, golden, name):
    g = golden(f"object_api_{name}.npz")
    flat = scene_of(name)
    backend.upload_scene(flat)
    for i in range(len(flat.objects)):
        hit, elem, t = backend.object_intersect(i, g["origins"], g["dirs"])
        assert np.array_equal(hit, g["hit"][i]) and np.array_equal(elem, g["elem"][i]) and np.array_equal(t.view(np.uint32), g["t"][i].view(np.uint32)), i
        if hit.any():
            p = (g["origins"] + g["dirs"] * t[:, None]).astype(np.float32)[hit]
            n = backend.surface_normals(np.full(len(p), i, np.uint32), elem[hit], p)
            assert np.array_equal(n.view(np.uint32), g["normal"][i][hit].view(np.uint32)), i


@pytest.mark.parametrize("name", NAMES)
def test_random_surface_points(backend, pkg, oracle, scene_of, golden, name):
    """Quad lights, meshes and the point light: bit-exact against the reference (no libm in their samplers).  A sphere runs acosf and
    two sincosf (source/objects/sphere.h:59-69): bit-exact against the oracle's fixed-order trigonometry, within 2 ulp of a unit
    sphere's coordinate against the reference's libm."""
    g = golden(f"object_api_{name}.npz")
    flat = scene_of(name)
    backend.upload_scene(flat)
    sc = oracle.scene(flat)
    for i, ob in enumerate(flat.objects):
        pts, st = backend.random_surface_points(i, g["states"])
        assert np.array_equal(st, g["states_out"][i])
        if ob["kind"] == pkg.abi.OBJECT_SPHERE:
            want, _ = oracle.object_random_points(sc, i, g["states"], portable_math=True)
            assert np.array_equal(pts.view(np.uint32), want.view(np.uint32))
            assert np.abs(pts - g["points"][i]).max() <= 4.0e-7 * max(1.0, float(ob["radius"]) + np.abs(ob["position"]).max())
            # a point takes acosf and two sincosf: 1-ulp differences of the fixed-order kernels against glibc touch ~5 % of the coordinates
            assert (pts.view(np.uint32) == g["points"][i].view(np.uint32)).mean() > 0.90
        else:
            assert np.array_equal(pts.view(np.uint32), g["points"][i].view(np.uint32)), i


def test_material_brdf(backend, pkg, oracle, golden):
    g = golden("brdf_hemisphere.npz")
    m = pkg.abi.Material()
    m.color[:] = (1, 1, 1); m.albedo = 0.75; m.diffuse = pkg.abi.DIFFUSE_LAMBERTIAN; m.roughness = 0.5
    assert np.array_equal(backend.brdf(m, g["incident"], g["reflection"], g["normals"]), np.full(len(g["normals"]), 0.75))
    for tag, rough in (("on_05", 0.5), ("on_10", 1.0), ("on_02", 0.2)):
        m.diffuse = pkg.abi.DIFFUSE_OREN_NAYAR; m.albedo = 1.0; m.roughness = rough
        got = backend.brdf(m, g["incident"], g["reflection"], g["normals"])
        want = oracle.brdf(m, g["incident"], g["reflection"], g["normals"], portable_math=True)
        same = (got.view(np.uint64) == want.view(np.uint64)) | (np.isnan(got) & np.isnan(want))
        assert same.all()
        ref = g[f"brdf_{tag}"]
        assert np.array_equal(np.isnan(got), np.isnan(ref))                   # the acos-domain NaN cases are the reference's
        ok = ~np.isnan(ref)
        # libm against the fixed-order trigonometry: 1 ulp of a float angle; tan(beta) near pi/2 amplifies it
        close = np.abs(got[ok] - ref[ok]) <= 1e-5 * np.maximum(1.0, np.abs(ref[ok]))
        assert close.mean() > 0.999 and (got[ok] == ref[ok]).mean() > 0.95      # two acosf, sinf, tanf chained: ~2 % of the values move in the last place


def test_scene_loop_rebuilt_from_per_object_calls(backend, pkg, scene_of, golden):
    """Scene::IntersectRay's loop (source/scene.cpp:62-80: every object's Intersects, strict < keeps the earlier object) rebuilt
    on the host from mcrt_object_intersect equals the scene-level fixture of the reference."""
    flat = scene_of("config3")
    g = golden("hits_config3.npz")
    sel = np.random.default_rng(5).choice(len(g["origins"]), 4000, replace=False)
    o, d = g["origins"][sel], g["dirs"][sel]
    backend.upload_scene(flat)
    best_t = np.full(len(o), np.finfo(np.float32).max, np.float32); best_o = np.full(len(o), -1, np.int32); best_e = np.zeros(len(o), np.uint32)
    for i in range(len(flat.objects)):
        hit, elem, t = backend.object_intersect(i, o, d)
        better = hit & (t < best_t)
        best_t[better] = t[better]; best_o[better] = i; best_e[better] = elem[better]
    assert np.array_equal(best_o, g["obj"][sel]) and np.array_equal(best_e, g["elem"][sel]) and np.array_equal(best_t, g["t"][sel])


def test_object_api_errors(backend, pkg, scene_of):
    flat = scene_of("cornell")
    backend.upload_scene(flat)
    o = np.zeros((2, 3), np.float32); d = np.tile(np.array([0, 0, -1], np.float32), (2, 1))
    with pytest.raises(pkg.McrtError) as e:
        backend.object_intersect(len(flat.objects), o, d)
    assert e.value.code == 1
    with pytest.raises(pkg.McrtError):
        backend.surface_normals([0, 0], [0, 99], o)                            # wall mesh 0 has two triangles
    with pytest.raises(pkg.McrtError):
        backend.random_surface_points(99, np.ones((1, 2), np.uint64))
    hit, elem, t = backend.object_intersect(0, o[:0], d[:0])
    assert len(hit) == 0


def test_host_mirror_virtuals(pkg, golden):
    """The C++ mirror: Object::Intersects / GetSurfaceNormal / GetRandomPointOnSurface called through the virtual interface of the
    objects of a CornellBoxScene, const Scene::TraceUnlit and Material::BRDF - single calls forwarded to the device."""
    host = C.CDLL(os.path.join(ROOT, "monte-carlo-ray-tracer_amd", "libmcrt_host.so"))
    host.mcrt_host_scene_create.restype = C.c_void_p
    host.mcrt_host_scene_create.argtypes = [C.c_int, C.c_float, C.c_double, C.c_int]
    host.mcrt_host_scene_destroy.argtypes = [C.c_void_p]
    host.mcrt_host_scene_prepare.argtypes = [C.c_void_p]
    g = golden("object_api_cornell.npz")
    n = 120
    o, d = np.ascontiguousarray(g["origins"][:n]), np.ascontiguousarray(g["dirs"][:n])
    f = lambda a: a.ctypes.data_as(C.c_void_p)
    s = host.mcrt_host_scene_create(0, 1.5, 10.0, 0)
    try:
        hit = np.zeros(n, np.uint8); elem = np.zeros(n, np.uint32); t = np.zeros(n, np.float32); nrm = np.zeros((n, 3), np.float32)
        host.mcrt_host_object_intersects.argtypes = [C.c_void_p, C.c_uint, C.c_uint64] + [C.c_void_p] * 6
        # before PrepareForRayTracing: a miss and a message, no crash
        assert host.mcrt_host_object_intersects(s, 3, 1, f(o), f(d), f(hit), f(elem), f(t), f(nrm)) == 0 and hit[0] == 0
        assert host.mcrt_host_scene_prepare(s) == 0
        for i in (0, 2, 3, 6, 7, 10):                                             # wall, white segments, spheres, a box, the light
            assert host.mcrt_host_object_intersects(s, i, n, f(o), f(d), f(hit), f(elem), f(t), f(nrm)) == 0
            h = g["hit"][i][:n]
            assert np.array_equal(hit.astype(bool), h) and np.array_equal(elem, g["elem"][i][:n]) and np.array_equal(t, g["t"][i][:n])
            assert np.array_equal(nrm[h].view(np.uint32), g["normal"][i][:n][h].view(np.uint32))
        # GetRandomPointOnSurface: 50 consecutive points of the light from ONE generator = the fixture's states chained
        host.mcrt_host_object_random_points.argtypes = [C.c_void_p, C.c_uint, C.c_uint64, C.c_void_p, C.c_void_p]
        st = g["states"][0].copy(); pts = np.zeros((1, 3), np.float32)
        assert host.mcrt_host_object_random_points(s, 10, 1, f(st), f(pts)) == 0
        assert np.array_equal(pts[0], g["points"][10][0]) and np.array_equal(st, g["states_out"][10][0])
        # const TraceUnlit
        host.mcrt_host_trace_unlit.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
        col = np.zeros((n, 3), np.float64)
        host.mcrt_host_trace_unlit(s, n, f(o), f(d), f(col))
        un = golden("hits_cornell.npz")
        flat = pkg.FlatScene.load(os.path.join(ROOT, "tests", "golden", "scene_cornell.npz"))
        # the first rays of object_api are a random pick of hits_cornell: rebuild the expected colour from the per-object fixture
        best_t = np.full(n, np.finfo(np.float32).max, np.float32); best_o = np.full(n, -1, np.int32)
        for i in range(len(flat.objects)):
            better = g["hit"][i][:n] & (g["t"][i][:n] < best_t)
            best_t[better] = g["t"][i][:n][better]; best_o[better] = i
        want = np.where(best_o[:, None] >= 0, flat.objects["material"]["color"][np.maximum(best_o, 0)], 0.0)
        assert np.array_equal(col, want)
    finally:
        host.mcrt_host_scene_destroy(s)
    # Material::BRDF
    b = golden("brdf_hemisphere.npz")
    k = 64
    host.mcrt_host_material_brdf.argtypes = [C.c_int, C.c_float, C.c_float, C.c_uint64] + [C.c_void_p] * 4
    out = np.zeros(k, np.float64)
    inc, refl, nr = (np.ascontiguousarray(b[x][16:16 + k]) for x in ("incident", "reflection", "normals"))
    host.mcrt_host_material_brdf(1, 1.0, 0.5, k, f(inc), f(refl), f(nr), f(out))
    ref = b["brdf_on_05"][16:16 + k]
    assert np.allclose(out, ref, rtol=1e-5, atol=1e-6) and (out == ref).mean() > 0.9
    host.mcrt_host_material_brdf(0, 0.25, 0.5, k, f(inc), f(refl), f(nr), f(out))
    assert np.array_equal(out, np.full(k, 0.25))


def test_host_mirror_objects_moved(pkg, golden):
    """Scene::ObjectsMoved of the C++ mirror (mcrt_refit_scene): a sphere and a box of the CornellBoxScene pushed aside answer
    differently, pushed back (exact float moves) they answer as the reference's fixture again - through the refitted tree."""
    host = C.CDLL(os.path.join(ROOT, "monte-carlo-ray-tracer_amd", "libmcrt_host.so"))
    host.mcrt_host_scene_create.restype = C.c_void_p
    host.mcrt_host_scene_create.argtypes = [C.c_int, C.c_float, C.c_double, C.c_int]
    host.mcrt_host_scene_destroy.argtypes = [C.c_void_p]
    host.mcrt_host_scene_prepare.argtypes = [C.c_void_p]
    host.mcrt_host_object_intersects.argtypes = [C.c_void_p, C.c_uint, C.c_uint64] + [C.c_void_p] * 6
    host.mcrt_host_scene_move_object.argtypes = [C.c_void_p, C.c_uint, C.c_float, C.c_float, C.c_float, C.c_void_p]
    g = golden("object_api_cornell.npz")
    n = 400
    o, d = np.ascontiguousarray(g["origins"][:n]), np.ascontiguousarray(g["dirs"][:n])
    f = lambda a: a.ctypes.data_as(C.c_void_p)
    s = host.mcrt_host_scene_create(0, 1.5, 10.0, 0)
    try:
        assert host.mcrt_host_scene_move_object(s, 6, 0.0, 0.0, 0.0, None) == 2        # before PrepareForRayTracing: refused (a zero move: the hook edits the object first)
        assert host.mcrt_host_scene_prepare(s) == 0
        hit = np.zeros(n, np.uint8); elem = np.zeros(n, np.uint32); t = np.zeros(n, np.float32); nrm = np.zeros((n, 3), np.float32)
        for i in (6, 7):
            rebuilt = C.c_int(-1)
            assert host.mcrt_host_scene_move_object(s, i, 0.25, 0.0, -0.125, C.byref(rebuilt)) == 0 and rebuilt.value in (0, 1)
            assert host.mcrt_host_object_intersects(s, i, n, f(o), f(d), f(hit), f(elem), f(t), f(nrm)) == 0
            h = g["hit"][i][:n]
            assert h.any() and not (np.array_equal(hit.astype(bool), h) and np.array_equal(t, g["t"][i][:n]))
            assert host.mcrt_host_scene_move_object(s, i, -0.25, 0.0, 0.125, C.byref(rebuilt)) == 0
            assert host.mcrt_host_object_intersects(s, i, n, f(o), f(d), f(hit), f(elem), f(t), f(nrm)) == 0
            if i == 6:            # the sphere's centre (0, -2, -2) moves and returns without rounding: bit-exact again
                assert np.array_equal(hit.astype(bool), h) and np.array_equal(elem, g["elem"][i][:n]) and np.array_equal(t, g["t"][i][:n])
            else:                 # a box vertex may come back an ulp off (x + 0.25 - 0.25 rounds twice)
                both = hit.astype(bool) & h
                assert (hit.astype(bool) == h).mean() > 0.99 and np.allclose(t[both], g["t"][i][:n][both], rtol=1e-5)
    finally:
        host.mcrt_host_scene_destroy(s)
