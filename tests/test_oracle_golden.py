"""The CPU restatement (oracle/mcrt_oracle.c) against fixtures produced by the
reference's own code (oracle/gen_golden.py).  Everything here is bit-exact:
the oracle uses libm like the reference build and the reference's recursion
order."""
import numpy as np
import pytest

from conftest import SCENES


def test_rng_known_answers(oracle, golden):
    g = golden("rng.npz")
    s0, s1 = (int(x) for x in g["seed"])
    assert np.array_equal(oracle.rng_doubles(s0, s1, 64), g["doubles"])
    assert np.array_equal(oracle.rng_floats(s0, s1, 64), g["floats"])
    assert np.array_equal(oracle.rng_floats_range(s0, s1, -2.5, 7.25, 64), g["floats_range"])
    b0, b1 = (int(x) for x in g["seed_big"])
    assert np.array_equal(oracle.rng_doubles(b0, b1, 64), g["doubles_big"])
    # SURVEY.md 2.2 probe: seed {1,2} -> 0x1p-41, 0x1p-39, ...
    assert [float(x).hex() for x in g["doubles"][:2]] == ["0x1.0000000000000p-41", "0x1.0000000000000p-39"]


@pytest.mark.parametrize("tag", ["512", "4k", "640x480", "oblique", "hexagon"])
def test_camera_matrix_and_rays(oracle, golden, pkg, tag):
    g = golden(f"camera_{tag}.npz")
    a = g["args"]
    cam = oracle.camera(int(a[0]), int(a[1]), float(a[2]), a[3:6], a[6:9], a[9:12])
    assert bytes(cam) == g["camera"].tobytes()
    o, d = oracle.camera_rays(cam, g["xs"], g["ys"])
    assert np.array_equal(o, g["origins"])
    assert np.array_equal(d.view(np.uint32), g["dirs"].view(np.uint32))


def test_survey_anchor_rays(oracle, scene_of, golden):
    """Known-answer anchors of SURVEY.md 8(c)."""
    cam = oracle.camera(512, 512, 90.0, (0, 0, 5), (0, 0, 0))
    xs = [0.5, 256.5, 100.25, 400.5, 256.5, 256.5, 180.5, 256.5]
    ys = [0.5, 256.5, 300.75, 128.5, 40.5, 2.5, 256.5, 330.5]
    o, d = oracle.camera_rays(cam, xs, ys)
    assert float(d[0, 0]).hex() == "-0x1.2769180000000p-1"
    obj, elem, t = oracle.intersect(oracle.scene(scene_of("cornell")), o, d)
    assert obj.tolist() == [1, 2, 3, 0, 2, 10, 3, 6]
    assert elem.tolist() == [1, 2, 0, 0, 0, 0, 0, 0]
    assert [float(x).hex() for x in t[:3]] == ["0x1.154f1a0000000p+3", "0x1.4000500000000p+3", "0x1.a9a1b60000000p+2"]


@pytest.mark.parametrize("name", SCENES)
def test_scene_derived_data(oracle, scene_of, golden, name):
    """Triangle normals, mesh AABBs and the light cache as the reference derives them."""
    flat = scene_of(name)
    g = golden(f"scene_{name}.npz")
    assert np.array_equal(oracle.scene_lights(oracle.scene(flat)), g["lights"])
    for o in flat.objects:
        if o["kind"] == 0:
            continue
        tris = flat.triangles[o["first_triangle"]:o["first_triangle"] + o["triangle_count"]]
        mn, mx = oracle.mesh_aabb(o["position"], tris)
        assert np.array_equal(mn, o["aabb_min"]) and np.array_equal(mx, o["aabb_max"])
        for t in tris[:64]:
            n = oracle.triangle_normal(t["v0"], t["v1"], t["v2"])
            assert np.array_equal(n.view(np.uint32), t["normal"].view(np.uint32)) or (np.isnan(n).all() and np.isnan(t["normal"]).all())


@pytest.mark.parametrize("name", SCENES)
def test_intersect_bit_exact(oracle, scene_of, golden, name):
    g = golden(f"hits_{name}.npz")
    obj, elem, t = oracle.intersect(oracle.scene(scene_of(name)), g["origins"], g["dirs"])
    assert np.array_equal(obj, g["obj"])
    assert np.array_equal(elem, g["elem"])
    assert np.array_equal(t.view(np.uint32), g["t"].view(np.uint32))


@pytest.mark.parametrize("name", SCENES)
def test_per_sample_radiance_bit_exact(oracle, pkg, scene_of, golden, name):
    """F5: pins TraceRay's arithmetic AND its random-draw order (refract branch first)."""
    g = golden(f"samples_{name}.npz")
    cam = pkg.abi.camera_from_numpy(g["camera"])
    rad = oracle.trace_samples(oracle.scene(scene_of(name)), cam, g["pixels"], g["samples"], int(g["seed"]), int(g["depth"]))
    assert np.array_equal(rad.view(np.uint64), g["radiance"].view(np.uint64))


def test_images_bit_exact(oracle, pkg, scene_of, golden):
    sc = oracle.scene(scene_of("cornell"))
    g = golden("image_cornell_128_64spp.npz")
    cam = pkg.abi.camera_from_numpy(g["camera"])
    sums, dropped = oracle.render(sc, cam, 0, int(g["spp"]), int(g["depth"]), int(g["seed"]), threads=8)
    assert dropped == int(g["dropped"])
    assert np.array_equal(sums, g["sums"])
    u = golden("unlit_cornell_128.npz")
    assert np.array_equal(oracle.render_unlit(sc, pkg.abi.camera_from_numpy(u["camera"])), u["color"])


def test_brdf_and_hemisphere(oracle, pkg, golden):
    g = golden("brdf_hemisphere.npz")
    for tag, rough in (("on_05", 0.5), ("on_10", 1.0), ("on_02", 0.2)):
        m = pkg.abi.Material()
        m.color[:] = (1, 1, 1); m.diffuse = pkg.abi.DIFFUSE_OREN_NAYAR; m.albedo = 1.0; m.roughness = rough; m.refractive_index = 1.52
        out = oracle.brdf(m, g["incident"], g["reflection"], g["normals"])
        assert np.array_equal(out.view(np.uint64), g[f"brdf_{tag}"].view(np.uint64))
    assert np.isnan(g["brdf_on_05"][:16]).any(), "fixture should contain the acos-domain NaN case"
    d, c = oracle.hemisphere(g["normals"], g["states"])
    assert np.array_equal(d.view(np.uint32), g["hemi_dir"].view(np.uint32)) and np.array_equal(c, g["hemi_cos"])
    d, c = oracle.hemisphere(g["axis_normals"], g["states"][:600])
    assert np.array_equal(d.view(np.uint32), g["hemi_dir_axis"].view(np.uint32)) and np.array_equal(c, g["hemi_cos_axis"])


def test_tonemap_table(oracle, pkg, golden):
    g = golden("tonemap.npz")
    spp = int(g["spp"])
    b, mean = oracle.resolve(g["sums"], spp, pkg.abi.TONEMAP_REINHARD)
    assert np.array_equal(b, g["reinhard"]) and np.array_equal(mean, g["mean"])
    assert np.array_equal(oracle.resolve(g["sums"], spp, pkg.abi.TONEMAP_EXPOSURE, exposure=1.0)[0], g["exposure"])
    assert np.array_equal(oracle.resolve(g["sums"], spp, pkg.abi.TONEMAP_EXPOSURE, gamma=1.8, exposure=0.35)[0], g["exposure_g18_e035"])
    assert np.array_equal(oracle.resolve(g["sums"], spp, pkg.abi.TONEMAP_NONE)[0], g["none"])
    # the fixed-order pow / exp that k_resolve runs give the reference's bytes too
    for name, mode, kw in (("reinhard", pkg.abi.TONEMAP_REINHARD, {}), ("exposure", pkg.abi.TONEMAP_EXPOSURE, {"exposure": 1.0}),
                           ("exposure_g18_e035", pkg.abi.TONEMAP_EXPOSURE, {"gamma": 1.8, "exposure": 0.35}), ("none", pkg.abi.TONEMAP_NONE, {})):
        assert np.array_equal(oracle.resolve(g["sums"], spp, mode, portable_math=True, **kw)[0], g[name]), name


def test_oracle_against_live_reference(oracle, ref, scene_of):
    """Where the reference build is present: fresh random samples, not only the stored ones."""
    rng = np.random.default_rng(7)
    flat = scene_of("config3")
    cam_r = ref.camera(256, 192, 75.0, (0.5, 0.2, 5), (0, -0.5, 0))
    cam_o = oracle.camera(256, 192, 75.0, (0.5, 0.2, 5), (0, -0.5, 0))
    assert bytes(cam_r) == bytes(cam_o)
    px = rng.integers(0, 256 * 192, 3000).astype(np.uint32)
    sm = rng.integers(0, 1 << 20, 3000).astype(np.uint32)
    a = ref.trace_samples(ref.scene(flat), cam_r, px, sm, 99, 100)
    b = oracle.trace_samples(oracle.scene(flat), cam_o, px, sm, 99, 100)
    assert np.array_equal(a.view(np.uint64), b.view(np.uint64))


def test_oracle_slice_order_is_what_the_header_says(oracle, scene_of):
    """orc_render(lanes_per_pixel=L) - the order the GPU parity tests compare against - equals the per-sample radiances
    of orc_trace_samples summed as include/mcrt.h (mcrt_lanes_per_pixel) describes: contiguous slices, each summed in
    sample order, the last one taking the remainder, slice sums added in slice order, then added to the accumulator."""
    sc = oracle.scene(scene_of("cornell"))
    cam = oracle.camera(12, 8, 90.0, (0, 0, 5), (0, 0, 0))
    seed, spp, begin = 0x5EED0001, 11, 3
    start = np.full((8, 12, 3), 0.25)
    for lanes in (1, 2, 4, 11):
        got, _ = oracle.render(sc, cam, begin, spp, 100, seed, threads=2, lanes_per_pixel=lanes, portable_math=True, forward_sum=True, sums=start.copy())
        per = spp // lanes
        for p in (0, 17, 95):
            s = oracle.trace_samples(sc, cam, np.full(spp, p, np.uint32), np.arange(begin, begin + spp, dtype=np.uint32), seed, 100,
                                     portable_math=True, forward_sum=True)
            tot = None
            for l in range(lanes):
                part = np.zeros(3)
                for row in s[l * per:(spp if l + 1 == lanes else (l + 1) * per)]:
                    part = part + row
                tot = part if tot is None else tot + part
            assert np.array_equal(got.reshape(-1, 3)[p], start.reshape(-1, 3)[p] + tot), (lanes, p)


@pytest.mark.parametrize("seed", range(24))
def test_oracle_against_live_reference_on_random_scenes(oracle, ref, scene_of, seed):
    """Random populations of spheres and meshes with every kind of material, with and without a second emitter: hits and per-sample
    radiance of the restatement equal the reference's own, bit for bit."""
    from mcrt_amd import scenes

    flat = scenes.random_scene(scene_of("cornell"), seed)
    rng = np.random.default_rng(100 + seed)
    o = rng.uniform(-4.5, 4.5, (4000, 3)).astype(np.float32)
    d = rng.normal(size=(4000, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    a = ref.intersect(ref.scene(flat), o, d)
    b = oracle.intersect(oracle.scene(flat), o, d)
    for x, y in zip(a, b):
        assert np.array_equal(x.view(np.uint32), y.view(np.uint32))
    cam_r = ref.camera(128, 96, 90.0, (0, 0, 5), (0, 0, 0))
    cam_o = oracle.camera(128, 96, 90.0, (0, 0, 5), (0, 0, 0))
    px = rng.integers(0, 128 * 96, 1500).astype(np.uint32)
    sm = rng.integers(0, 1 << 16, 1500).astype(np.uint32)
    ra = ref.trace_samples(ref.scene(flat), cam_r, px, sm, 4242, 100)
    rb = oracle.trace_samples(oracle.scene(flat), cam_o, px, sm, 4242, 100)
    same = (ra.view(np.uint64) == rb.view(np.uint64)).all(axis=1) | (np.isnan(ra).any(axis=1) & np.isnan(rb).any(axis=1))
    assert same.all(), f"{(~same).sum()} of {len(same)} samples differ"


@pytest.mark.parametrize("name", ["cornell", "cornell_point", "cornell_sphere_light", "hexagon"])
def test_per_object_virtuals_bit_exact(oracle, scene_of, golden, name):
    """F9: Object::Intersects / GetSurfaceNormal / GetRandomPointOnSurface of every object alone, as the reference's virtuals answer."""
    g = golden(f"object_api_{name}.npz")
    flat = scene_of(name)
    sc = oracle.scene(flat)
    for i in range(len(flat.objects)):
        hit, elem, t = oracle.object_intersects(sc, i, g["origins"], g["dirs"])
        assert np.array_equal(hit, g["hit"][i]) and np.array_equal(elem, g["elem"][i]) and np.array_equal(t.view(np.uint32), g["t"][i].view(np.uint32))
        p = (g["origins"] + g["dirs"] * t[:, None]).astype(np.float32)
        n = oracle.surface_normal(sc, np.full(len(p), i, np.int32), elem, p)
        assert np.array_equal(n[hit].view(np.uint32), g["normal"][i][hit].view(np.uint32))
        pts, st = oracle.object_random_points(sc, i, g["states"])
        assert np.array_equal(pts.view(np.uint32), g["points"][i].view(np.uint32)) and np.array_equal(st, g["states_out"][i])
    # the loop of Scene::IntersectRay (source/scene.cpp:62-80) rebuilt from the per-object answers gives the scene-level hits
    best_t = np.full(len(g["origins"]), np.finfo(np.float32).max, np.float32); best_o = np.full(len(best_t), -1, np.int32)
    for i in range(len(flat.objects)):
        better = g["hit"][i] & (g["t"][i] < best_t)
        best_t[better] = g["t"][i][better]; best_o[better] = i
    so, _, st_ = oracle.intersect(sc, g["origins"], g["dirs"])
    assert np.array_equal(best_o, so) and np.array_equal(best_t, st_)
