"""GPU parity tests: the HIP path, called through the C ABI (ctypes), against
  (1) the golden fixtures produced by the reference's own code, and
  (2) the CPU oracle on the same seeded inputs.

Bars: camera rays, hit (object, element, t) and per-sample radiance are
BIT-EXACT against the oracle run with the same fixed-order trig
(portable_math) and the kernel's summation order (forward_sum); against the
reference's own numbers (libm trig, recursion-order sums) the tolerance is
stated in each test.
"""
import numpy as np
import pytest

from conftest import SCENES

pytestmark = pytest.mark.gpu

SEED = 0x5EED0001


def _upload(backend, pkg, scene_of, name, accel=None):
    flat = scene_of(name)
    backend.upload_scene(flat, accel)
    return flat


@pytest.mark.parametrize("tag", ["512", "4k", "640x480", "oblique", "hexagon"])
def test_camera_rays_bit_exact(backend, pkg, golden, tag):
    g = golden(f"camera_{tag}.npz")
    backend.set_camera(pkg.abi.camera_from_numpy(g["camera"]))
    o, d = backend.camera_rays(g["xs"], g["ys"])
    assert np.array_equal(o, g["origins"])
    assert np.array_equal(d.view(np.uint32), g["dirs"].view(np.uint32))


@pytest.mark.parametrize("accel", ["octree", "brute", "scan", "flat"])
@pytest.mark.parametrize("name", SCENES)
def test_hits_bit_exact_vs_reference(backend, pkg, scene_of, golden, name, accel):
    """Fixed-ray queries: hit object, element index AND distance equal the reference's Scene::IntersectRay."""
    _upload(backend, pkg, scene_of, name)
    if accel == "flat" and not backend.octree_info().lds_resident:
        pytest.skip("the flat scan serves LDS-resident scenes")
    backend.set_accel({"octree": pkg.ACCEL_OCTREE, "brute": pkg.ACCEL_BRUTE, "scan": pkg.ACCEL_SCAN, "flat": pkg.ACCEL_FLAT}[accel])
    g = golden(f"hits_{name}.npz")
    obj, elem, t = backend.intersect(g["origins"], g["dirs"])
    backend.set_accel(pkg.ACCEL_AUTO)
    bad = np.flatnonzero((obj != g["obj"]) | (elem != g["elem"]) | (t.view(np.uint32) != g["t"].view(np.uint32)))
    assert len(bad) == 0, f"{len(bad)} of {len(obj)} rays differ, first {bad[:5]}"


@pytest.mark.parametrize("params", [(1, 12), (2, 6), (16, 3), (64, 1)])
def test_hits_independent_of_octree_shape(backend, pkg, scene_of, golden, params):
    """Any leaf size / depth gives the same hits (the structure only orders and culls)."""
    accel = pkg.AccelParams(params[0], params[1], 0.0)
    for name in ("config2", "config4_2k"):
        _upload(backend, pkg, scene_of, name, accel)
        backend.set_accel(pkg.ACCEL_OCTREE)
        g = golden(f"hits_{name}.npz")
        obj, elem, t = backend.intersect(g["origins"], g["dirs"])
        backend.set_accel(pkg.ACCEL_AUTO)
        assert np.array_equal(obj, g["obj"]) and np.array_equal(elem, g["elem"]) and np.array_equal(t, g["t"])


@pytest.mark.parametrize("name", SCENES)
def test_per_sample_radiance(backend, pkg, oracle, scene_of, golden, name):
    flat = _upload(backend, pkg, scene_of, name)
    g = golden(f"samples_{name}.npz")
    cam = pkg.abi.camera_from_numpy(g["camera"])
    backend.set_camera(cam)
    seed, depth = int(g["seed"]), int(g["depth"])
    got = backend.trace_samples(g["pixels"], g["samples"], seed, depth)
    # (1) bit-exact against the oracle with the kernel's trig and summation order
    want = oracle.trace_samples(oracle.scene(flat), cam, g["pixels"], g["samples"], seed, depth, portable_math=True, forward_sum=True)
    same = (got.view(np.uint64) == want.view(np.uint64)).all(axis=1) | (np.isnan(got).any(axis=1) & np.isnan(want).any(axis=1))
    assert same.all(), f"{(~same).sum()} of {len(same)} samples differ from the oracle, first {np.flatnonzero(~same)[:5]}"
    # (2) against the reference's own numbers (glibc trig, recursion-order sums): a path can only
    # differ where a <=1-ulp trig difference exists; require 99 % of samples within 1e-9 relative.
    # A sphere light runs acosf + two sincosf per light sample (source/objects/sphere.h:59-69), so there the
    # 1-ulp differences reach 5 % of the samples, as a float-rounding-sized change of the light direction:
    # 94 % within 1e-9 and 99.9 % within 1e-5 (glibc 2.35's acosf itself differs from the correctly rounded
    # value on 7.7 % of inputs, its tanf on 3.9 %, sinf/cosf on 1.3 %: DESIGN.md section 4).
    ref = g["radiance"]
    scale = np.maximum(np.abs(ref).max(axis=1), 1e-3)
    rel = np.abs(got - ref).max(axis=1) / scale
    both_nan = np.isnan(got).any(axis=1) & np.isnan(ref).any(axis=1)
    ok = (rel < 1e-9) | both_nan
    need = 0.94 if name in ("cornell_sphere_light", "three_lights") else 0.99      # the scenes with a sphere light of radius > 0
    assert ok.mean() > need, f"only {ok.mean():.4f} of samples within 1e-9 of the reference"
    assert ((rel < 1e-5) | both_nan).mean() > 0.998
    assert abs(np.nanmean(got) - np.nanmean(ref)) < 0.02 * abs(np.nanmean(ref)) + 1e-6


def _sliced_sum(rows, lanes):
    """Sum of per-sample radiances in the order a render pass uses (include/mcrt.h, mcrt_lanes_per_pixel)."""
    per = len(rows) // lanes
    tot = None
    for l in range(lanes):
        part = np.zeros(3)
        for row in rows[l * per:(len(rows) if l + 1 == lanes else (l + 1) * per)]:
            part = part + row
        tot = part if tot is None else tot + part
    return tot


def test_render_through_hbm_resident_octree(backend, pkg, oracle, scene_of, golden):
    """k_render with free-running lanes, the resumable walk and the LDS/scratch stack (2000-triangle mesh: the scene
    image does not fit the LDS budget) equals the oracle's render bit for bit; so does a deep tree (leaf size 1,
    depth 12) whose stacks overflow the LDS part, and sample counts that do not divide into the slices."""
    flat = scene_of("config4_2k")
    cam = oracle.camera(72, 40, 90.0, (0, 0, 5), (0, 0, 0))
    sc = oracle.scene(flat)
    for accel, spp in ((None, 7), (pkg.AccelParams(1, 12, 0.0), 5)):
        backend.upload_scene(flat, accel)
        assert not backend.octree_info().lds_resident
        backend.set_camera(cam)
        backend.set_lanes_per_pixel(4)
        backend.clear(); backend.render(3, spp, 100, SEED)
        got = backend.read_accumulator()
        lanes = backend.lanes_per_pixel(spp)
        backend.set_lanes_per_pixel(0)
        assert lanes == 4 and spp % lanes != 0
        want, _ = oracle.render(sc, cam, 3, spp, 100, SEED, threads=8, portable_math=True, forward_sum=True, lanes_per_pixel=lanes)
        assert np.array_equal(got, want), f"max abs diff {np.abs(got - want).max()}"
    # per-sample radiance through the deep tree (k_trace_list, same walk)
    g = golden("samples_config4_2k.npz")
    backend.set_camera(pkg.abi.camera_from_numpy(g["camera"]))
    n = 4000
    got = backend.trace_samples(g["pixels"][:n], g["samples"][:n], int(g["seed"]), int(g["depth"]))
    want = oracle.trace_samples(sc, pkg.abi.camera_from_numpy(g["camera"]), g["pixels"][:n], g["samples"][:n], int(g["seed"]), int(g["depth"]),
                                portable_math=True, forward_sum=True)
    same = (got.view(np.uint64) == want.view(np.uint64)).all(axis=1) | (np.isnan(got).any(axis=1) & np.isnan(want).any(axis=1))
    assert same.all()


def test_all_accel_modes_render_identically(backend, pkg, oracle, scene_of):
    flat = _upload(backend, pkg, scene_of, "config3")
    cam = oracle.camera(96, 64, 90.0, (0, 0, 5), (0, 0, 0))
    backend.set_camera(cam)
    backend.set_accel(pkg.ACCEL_BRUTE)
    backend.clear(); backend.render(0, 16, 100, SEED)
    a = backend.read_accumulator()
    for mode in (pkg.ACCEL_OCTREE, pkg.ACCEL_SCAN, pkg.ACCEL_FLAT):
        backend.set_accel(mode)
        backend.clear(); backend.render(0, 16, 100, SEED)
        assert np.array_equal(a, backend.read_accumulator())
    backend.set_accel(pkg.ACCEL_AUTO)
    want, _ = oracle.render(oracle.scene(flat), cam, 0, 16, 100, SEED, threads=8, portable_math=True, forward_sum=True,
                             lanes_per_pixel=backend.lanes_per_pixel(16))
    assert np.array_equal(a, want), f"max abs diff {np.abs(a - want).max()}"


def test_image_vs_reference_render(backend, pkg, scene_of, golden):
    """Converged-image parity on the SAME random streams as the reference render: the residual is
    trig-rounding only, far below the Monte-Carlo noise floor (RMSE 0.085 at 64 spp)."""
    _upload(backend, pkg, scene_of, "cornell")
    g = golden("image_cornell_128_64spp.npz")
    backend.set_camera(pkg.abi.camera_from_numpy(g["camera"]))
    spp = int(g["spp"])
    backend.clear(); backend.render(0, spp, int(g["depth"]), int(g["seed"]))
    got = backend.read_accumulator() / spp
    ref = g["sums"] / spp
    rmse = float(np.sqrt(np.mean((got - ref) ** 2)))
    lum = lambda im: (0.2126 * im[..., 0] + 0.7152 * im[..., 1] + 0.0722 * im[..., 2]).mean()
    assert rmse < 0.0849 * 0.05, f"RMSE {rmse} vs noise floor 0.0849"
    assert abs(lum(got) - lum(ref)) / lum(ref) < 1e-3
    assert np.allclose(got.mean(axis=(0, 1)), ref.mean(axis=(0, 1)), rtol=2e-3)
    # pixels whose 64 paths all took the same decisions agree to rounding; a <=1-ulp trig difference
    # can flip a roulette / hit decision in a few
    assert (np.abs(got - ref) < 1e-9).mean() > 0.90
    assert (np.abs(got - ref) < 1e-5).mean() > 0.99


def test_high_spp_image_vs_reference(backend, pkg, scene_of, golden):
    _upload(backend, pkg, scene_of, "cornell")
    g = golden("image_cornell_64_1024spp.npz")
    backend.set_camera(pkg.abi.camera_from_numpy(g["camera"]))
    spp = int(g["spp"])
    backend.clear()
    for s in range(0, spp, 256):                     # progressive passes accumulate
        backend.render(s, 256, int(g["depth"]), int(g["seed"]))
    got = backend.read_accumulator() / spp
    ref, other = g["sums"] / spp, g["sums_other_seed"] / spp
    floor = float(np.sqrt(np.mean((ref - other) ** 2)))          # the reference against itself, other seed
    rmse = float(np.sqrt(np.mean((got - ref) ** 2)))
    assert rmse <= 0.05 * floor, f"RMSE {rmse} vs reference-vs-reference {floor}"


def test_unlit_render(backend, pkg, scene_of, golden):
    _upload(backend, pkg, scene_of, "cornell")
    u = golden("unlit_cornell_128.npz")
    backend.set_camera(pkg.abi.camera_from_numpy(u["camera"]))
    assert np.array_equal(backend.render_unlit(), u["color"])


def test_resolve_tonemap(backend, pkg, oracle, scene_of):
    """mean -> Reinhard / exposure -> gamma -> clamp -> x255 truncate -> BGRA8: every byte equal to the oracle's (pow / exp run
    through the same fixed-order kernels on both sides; tests/test_gpu_driver.py holds the reference's own bytes)."""
    flat = _upload(backend, pkg, scene_of, "cornell")
    cam = oracle.camera(64, 48, 90.0, (0, 0, 5), (0, 0, 0))
    backend.set_camera(cam)
    backend.clear(); backend.render(0, 8, 100, SEED)
    sums = backend.read_accumulator()
    for mode, kw in ((pkg.TONEMAP_REINHARD, {}), (pkg.TONEMAP_EXPOSURE, {"exposure": 0.7, "gamma": 1.8}), (pkg.TONEMAP_NONE, {})):
        bgra, mean = backend.resolve(mode, 8, **kw)
        want, wmean = oracle.resolve(sums, 8, mode, portable_math=True, **kw)
        assert np.array_equal(mean.reshape(-1, 3), wmean)
        assert np.array_equal(bgra.reshape(-1, 4), want)
        assert np.array_equal(want, oracle.resolve(sums, 8, mode, **kw)[0])          # and libm's pow / exp give the same bytes
        assert (bgra[..., 3] == 255).all()


def test_progressive_passes_and_sample_ranges(backend, pkg, scene_of, oracle):
    """Two passes over [0,32) and [32,64) equal the per-slice oracle sums added in the same order."""
    flat = _upload(backend, pkg, scene_of, "cornell")
    cam = oracle.camera(80, 56, 90.0, (0, 0, 5), (0, 0, 0))
    backend.set_camera(cam)
    backend.set_lanes_per_pixel(8)
    backend.clear(); backend.render(0, 32, 100, SEED); backend.render(32, 32, 100, SEED)
    got = backend.read_accumulator()
    sc = oracle.scene(flat)
    lanes = backend.lanes_per_pixel(32)
    backend.set_lanes_per_pixel(0)
    assert lanes == 8
    want, _ = oracle.render(sc, cam, 0, 32, 100, SEED, threads=8, portable_math=True, forward_sum=True, lanes_per_pixel=lanes)
    want, _ = oracle.render(sc, cam, 32, 32, 100, SEED, threads=8, portable_math=True, forward_sum=True, sums=want, lanes_per_pixel=lanes)
    assert np.array_equal(got, want)


def test_full_size_properties_config1(backend, pkg, scene_of, oracle):
    """BASELINE config 1 at full size (512x512x64 = 16.8 M samples): size-independent checks."""
    flat = _upload(backend, pkg, scene_of, "cornell")
    cam = oracle.camera(512, 512, 90.0, (0, 0, 5), (0, 0, 0))
    backend.set_camera(cam)
    backend.set_counting(True)
    backend.clear(); backend.render(0, 64, 100, SEED)
    a = backend.read_accumulator()
    cnt = backend.counters()
    backend.set_counting(False)
    assert cnt["samples"] == 512 * 512 * 64
    # SURVEY.md 3.2 probes: 4.93 IntersectRay, 2.76 TraceRay, 0.123 splits per sample
    assert abs(cnt["rays"] / cnt["samples"] - 4.93) < 0.05
    assert abs(cnt["trace_calls"] / cnt["samples"] - 2.76) < 0.03
    assert abs(cnt["splits"] / cnt["samples"] - 0.123) < 0.005
    assert np.isfinite(a).all()
    # determinism: a second render reproduces the accumulators exactly
    backend.clear(); backend.render(0, 64, 100, SEED)
    assert np.array_equal(a, backend.read_accumulator())
    # linearity over sample ranges up to summation order
    backend.clear(); backend.render(0, 16, 100, SEED); backend.render(16, 48, 100, SEED)
    b = backend.read_accumulator()
    assert np.allclose(a, b, rtol=1e-12, atol=1e-12)
    # frame mean: SURVEY.md section 6 anchor R 0.7447 G 0.6820 B 0.6671 (+-0.001 at 64 spp)
    mean = a.mean(axis=(0, 1)) / 64
    assert np.allclose(mean, [0.7447, 0.6820, 0.6671], atol=0.004), mean
    # spot check of random pixels against the oracle (same streams, kernel order)
    rng = np.random.default_rng(3)
    px = rng.integers(0, 512 * 512, 64).astype(np.uint32)
    sc = oracle.scene(flat)
    for p in px[:16]:
        s = oracle.trace_samples(sc, cam, np.full(64, p, np.uint32), np.arange(64, dtype=np.uint32), SEED, 100, portable_math=True, forward_sum=True)
        assert np.array_equal(_sliced_sum(s, backend.lanes_per_pixel(64)), a.reshape(-1, 3)[p])
    # the reference's own order (one slice): same samples, sums equal up to rounding, and exactly the forward sum
    backend.set_lanes_per_pixel(1)
    backend.clear(); backend.render(0, 64, 100, SEED)
    one = backend.read_accumulator()
    backend.set_lanes_per_pixel(0)
    assert np.allclose(a, one, rtol=1e-12, atol=1e-12)
    for p in px[:4]:
        s = oracle.trace_samples(sc, cam, np.full(64, p, np.uint32), np.arange(64, dtype=np.uint32), SEED, 100, portable_math=True, forward_sum=True)
        assert np.array_equal(_sliced_sum(s, 1), one.reshape(-1, 3)[p])


def test_error_behaviour(pkg, scene_of):
    b = pkg.Backend(0)
    with pytest.raises(pkg.McrtError) as e:
        b.render(0, 1, 5, 1)
    assert e.value.code == 4
    flat = scene_of("cornell")
    bad = pkg.FlatScene(flat.objects.copy(), flat.triangles.copy(), flat.background)
    bad.objects["first_triangle"][9] = 10_000
    with pytest.raises(pkg.McrtError) as e:
        b.upload_scene(bad)
    assert e.value.code == 1
    b.upload_scene(flat)
    with pytest.raises(pkg.McrtError) as e:
        b.render(0, 1, 5, 1)          # still no camera
    assert e.value.code == 4
    with pytest.raises(pkg.McrtError):
        pkg.Backend(10_000)
    b.close()


def test_empty_and_ragged_inputs(backend, pkg, scene_of, oracle):
    _upload(backend, pkg, scene_of, "cornell")
    obj, elem, t = backend.intersect(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32))
    assert len(obj) == 0
    # image sizes that are not multiples of the 8x8 tile
    cam = oracle.camera(37, 21, 90.0, (0, 0, 5), (0, 0, 0))
    backend.set_camera(cam)
    backend.clear(); backend.render(5, 3, 100, SEED)
    got = backend.read_accumulator()
    want, _ = oracle.render(oracle.scene(scene_of("cornell")), cam, 5, 3, 100, SEED, threads=4, portable_math=True, forward_sum=True,
                             lanes_per_pixel=backend.lanes_per_pixel(3))
    assert np.array_equal(got, want)
    # depth 0: only emission of directly visible lights survives
    backend.clear(); backend.render(0, 2, 0, SEED)
    d0 = backend.read_accumulator()
    want0, _ = oracle.render(oracle.scene(scene_of("cornell")), cam, 0, 2, 0, SEED, threads=4, portable_math=True, forward_sum=True,
                              lanes_per_pixel=backend.lanes_per_pixel(2))
    assert np.array_equal(d0, want0)


@pytest.mark.parametrize("size", [(1, 1), (1, 130), (257, 1), (9, 7)])
def test_degenerate_image_sizes(backend, pkg, scene_of, oracle, size):
    """One-pixel, one-column and one-row images (tiles mostly outside the image, more sample slices than some passes have samples);
    a pass of zero samples leaves the sums as they are."""
    _upload(backend, pkg, scene_of, "cornell")
    w, h = size
    cam = oracle.camera(w, h, 90.0, (0, 0, 5), (0, 0, 0))
    backend.set_camera(cam)
    sc = oracle.scene(scene_of("cornell"))
    backend.clear()
    want = None
    done = 0
    for n in (1, 40, 0, 7):
        backend.render(done, n, 100, SEED)
        if n:
            want, _ = oracle.render(sc, cam, done, n, 100, SEED, threads=4, portable_math=True, forward_sum=True, sums=want,
                                     lanes_per_pixel=pkg.lanes_per_pixel(w, h, n))
        done += n
        assert np.array_equal(backend.read_accumulator(), want)
    bgra, mean = backend.resolve(pkg.TONEMAP_REINHARD, done)
    ref_bgra, ref_mean = oracle.resolve(want, done, pkg.TONEMAP_REINHARD, portable_math=True)
    assert np.array_equal(bgra.reshape(-1, 4), ref_bgra.reshape(-1, 4)) and np.array_equal(mean.reshape(-1, 3), np.asarray(ref_mean).reshape(-1, 3))


def test_trace_rays_explicit_generator(backend, pkg, oracle, scene_of, golden):
    """Scene::TraceRay(ray, generator, depth, importance): explicit rays, caller-owned generator state that
    comes back advanced, non-unit importance."""
    flat = _upload(backend, pkg, scene_of, "config3")
    g = golden("hits_config3.npz")
    n = 3000
    o, d = g["origins"][:n], g["dirs"][:n]
    rng = np.random.default_rng(11)
    states = rng.integers(1, 2**63, size=(n, 2)).astype(np.uint64)
    got, after = backend.trace_rays(o, d, states, 100)
    want = oracle.trace_rays(oracle.scene(flat), o, d, states, 100, portable_math=True, forward_sum=True)
    same = (got.view(np.uint64) == want.view(np.uint64)).all(axis=1) | (np.isnan(got).any(axis=1) & np.isnan(want).any(axis=1))
    assert same.all()
    assert (after != states).any(axis=1).mean() > 0.5            # generators advanced where a diffuse surface was hit
    imp = rng.uniform(0.1, 1.0, size=(n, 3))
    scaled, _ = backend.trace_rays(o, d, states, 3, importance=imp)
    assert np.isfinite(scaled).all()


def test_headless_driver_matches_library(backend, pkg, oracle, scene_of, tmp_path):
    """The C++ host layer end to end (CornellBoxScene -> PrepareForRayTracing -> Scene::Render -> Resolve):
    same accumulators and same BGRA8 image as the ctypes path on the reference's dumped scene (the driver against
    the ORACLE: tests/test_gpu_driver.py)."""
    import os
    import subprocess

    from conftest import ROOT

    exe = os.path.join(ROOT, "monte-carlo-ray-tracer_amd", "mcrt_render")
    ppm, raw = tmp_path / "o.ppm", tmp_path / "o.f64"
    subprocess.run([exe, "--width", "96", "--height", "72", "--spp", "12", "--pass", "5", "--out", str(ppm), "--raw", str(raw)], check=True, timeout=120)
    _upload(backend, pkg, scene_of, "cornell")
    cam = pkg.host_camera(96, 72, 90.0, (0, 0, 5), (0, 0, 0))
    backend.set_camera(cam)
    backend.clear()
    for b, n in ((0, 5), (5, 5), (10, 2)):
        backend.render(b, n, 100, SEED)
    sums = backend.read_accumulator()
    assert np.array_equal(np.fromfile(raw, np.float64).reshape(72, 96, 3), sums)
    bgra, _ = backend.resolve(pkg.TONEMAP_REINHARD, 12)
    data = open(ppm, "rb").read()
    head = b"P6\n96 72\n255\n"
    assert data.startswith(head)
    rgb = np.frombuffer(data[len(head):], np.uint8).reshape(72, 96, 3)
    assert np.array_equal(rgb, bgra[..., 2::-1])


def test_checkpoint_and_resume(backend, pkg, scene_of, tmp_path):
    """Accumulator checkpoint / resume: a render continued from a checkpoint in a new process
    equals the uninterrupted one bit for bit; mcrt_write_accumulator is the inverse of mcrt_read_accumulator."""
    import os
    import subprocess

    from conftest import ROOT

    exe = os.path.join(ROOT, "monte-carlo-ray-tracer_amd", "mcrt_render")
    ck, r1, r2 = tmp_path / "ck.bin", tmp_path / "a.f64", tmp_path / "b.f64"
    common = [exe, "--width", "64", "--height", "40", "--pass", "4", "--out", str(tmp_path / "o.ppm")]
    subprocess.run(common + ["--spp", "8", "--checkpoint", str(ck)], check=True, timeout=120)
    subprocess.run(common + ["--spp", "16", "--resume", str(ck), "--raw", str(r1)], check=True, timeout=120)
    subprocess.run(common + ["--spp", "16", "--raw", str(r2)], check=True, timeout=120)
    a, b = np.fromfile(r1, np.float64), np.fromfile(r2, np.float64)
    assert a.size == 64 * 40 * 3 and np.array_equal(a, b)
    # a checkpoint of another image size is refused
    bad = subprocess.run([exe, "--width", "32", "--height", "40", "--spp", "16", "--resume", str(ck), "--out", str(tmp_path / "p.ppm")], timeout=120)
    assert bad.returncode != 0
    _upload(backend, pkg, scene_of, "cornell")
    backend.set_camera(pkg.host_camera(64, 40, 90.0, (0, 0, 5), (0, 0, 0)))
    backend.write_accumulator(b)
    assert np.array_equal(backend.read_accumulator().reshape(-1), b)


def test_plain_c_example(tmp_path):
    """examples/minimal.c: the ABI used from C99 (hand-built scene -> render -> resolve -> PPM)."""
    import os
    import shutil
    import subprocess

    from conftest import ROOT

    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    lib = os.path.join(ROOT, "monte-carlo-ray-tracer_amd")
    exe = str(tmp_path / "minimal")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "minimal.c"),
                    "-L", lib, "-lmcrt_hip", f"-Wl,-rpath,{lib}", "-lm", "-o", exe], check=True, timeout=120)
    r = subprocess.run([exe, str(tmp_path / "m.ppm")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    lit = int(r.stdout.split(",")[1].split()[0])
    assert 12000 < lit < 30000, r.stdout          # floor, back wall and light cover about a third of the frame; the rest is background
    data = open(tmp_path / "m.ppm", "rb").read()
    assert data.startswith(b"P6\n256 192\n255\n") and len(data) == 15 + 256 * 192 * 3


def test_full_size_properties_config2(backend, pkg, scene_of, oracle):
    """BASELINE config 2 at full size (1024x1024x256 = 268 M samples): sample count, determinism, linearity over
    sample ranges, and random pixels against the oracle's per-sample radiance summed in the pass's slice order."""
    flat = _upload(backend, pkg, scene_of, "config2")
    cam = oracle.camera(1024, 1024, 90.0, (0, 0, 5), (0, 0, 0))
    backend.set_camera(cam)
    backend.set_counting(True)
    backend.clear(); backend.render(0, 256, 100, SEED)
    cnt = backend.counters()
    backend.set_counting(False)
    assert cnt["samples"] == 1024 * 1024 * 256
    backend.clear(); backend.render(0, 256, 100, SEED)
    a = backend.read_accumulator()
    backend.clear(); backend.render(0, 256, 100, SEED)
    assert np.array_equal(a, backend.read_accumulator())
    assert np.isfinite(a).all()
    backend.clear(); backend.render(0, 128, 100, SEED); backend.render(128, 128, 100, SEED)
    assert np.allclose(a, backend.read_accumulator(), rtol=1e-12, atol=1e-12)
    lanes = backend.lanes_per_pixel(256)
    assert lanes == 4
    sc = oracle.scene(flat)
    rng = np.random.default_rng(5)
    for p in rng.integers(0, 1024 * 1024, 6).astype(np.uint32):
        s = oracle.trace_samples(sc, cam, np.full(256, p, np.uint32), np.arange(256, dtype=np.uint32), SEED, 100, portable_math=True, forward_sum=True)
        assert np.array_equal(_sliced_sum(s, lanes), a.reshape(-1, 3)[p])


def test_million_triangle_scene(backend, pkg, scene_of, oracle):
    """BASELINE config 4 geometry at full size (1,000,012 triangles, octree walked from HBM): hits equal the oracle's
    brute force over all triangles on a ray sample, per-sample radiance equals the oracle's, a render is deterministic."""
    from mcrt_amd import scenes

    flat = scenes.config4(scene_of("cornell"), n=1_000_000, seed=42)
    backend.upload_scene(flat)
    info = backend.octree_info()
    assert not info.lds_resident and info.node_count > 1_000_000
    sc = oracle.scene(flat)
    cam = oracle.camera(64, 64, 90.0, (0, 0, 5), (0, 0, 0))
    backend.set_camera(cam)
    rng = np.random.default_rng(9)
    o = rng.uniform(-4.4, 4.4, (300, 3)).astype(np.float32)
    d = rng.normal(size=(300, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    obj, elem, t = backend.intersect(o, d)
    wobj, welem, wt = oracle.intersect(sc, o, d)
    assert np.array_equal(obj, wobj) and np.array_equal(elem, welem) and np.array_equal(t.view(np.uint32), wt.view(np.uint32))
    assert (obj == 3).mean() > 0.5                      # most rays end in the soup (object 3)
    px = rng.integers(0, 64 * 64, 40).astype(np.uint32)
    sm = rng.integers(0, 1000, 40).astype(np.uint32)
    got = backend.trace_samples(px, sm, SEED, 100)
    want = oracle.trace_samples(sc, cam, px, sm, SEED, 100, portable_math=True, forward_sum=True)
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64))
    backend.clear(); backend.render(0, 16, 100, SEED)
    a = backend.read_accumulator()
    backend.clear(); backend.render(0, 16, 100, SEED)
    assert np.array_equal(a, backend.read_accumulator()) and np.isfinite(a).all() and a.mean() > 0


def test_full_size_properties_config3(backend, pkg, scene_of, oracle):
    """BASELINE config 3 at full size (Oren-Nayar + heavy refraction, 1024x1024x1024 = 1.07 G samples): sample count, the
    path-tree statistics of SURVEY.md 3.2 (deep refractive paths: more rays and splits than the default scene), and random
    pixels against the oracle's per-sample radiance summed in the pass's slice order, bit for bit."""
    flat = _upload(backend, pkg, scene_of, "config3")
    cam = oracle.camera(1024, 1024, 90.0, (0, 0, 5), (0, 0, 0))
    backend.set_camera(cam)
    backend.set_counting(True)
    backend.clear(); backend.render(0, 64, 100, SEED)                 # the counting build on a 64-spp slice of the job
    cnt = backend.counters()
    backend.set_counting(False)
    assert cnt["samples"] == 1024 * 1024 * 64
    assert cnt["rays"] / cnt["samples"] > 5.0 and cnt["splits"] / cnt["samples"] > 0.2
    backend.clear(); backend.render(0, 1024, 100, SEED)
    a = backend.read_accumulator()
    assert np.isfinite(a).all() and backend.dropped_samples() < 1024
    lanes = backend.lanes_per_pixel(1024)
    sc = oracle.scene(flat)
    rng = np.random.default_rng(6)
    for p in rng.integers(0, 1024 * 1024, 4).astype(np.uint32):
        s = oracle.trace_samples(sc, cam, np.full(1024, p, np.uint32), np.arange(1024, dtype=np.uint32), SEED, 100, portable_math=True, forward_sum=True)
        keep = s[np.isfinite(s).all(axis=1)]                           # non-finite samples are dropped (stated deviation)
        if len(keep) == len(s):
            assert np.array_equal(_sliced_sum(s, lanes), a.reshape(-1, 3)[p])
