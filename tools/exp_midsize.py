"""Experiment: a mid-size scene (2012 triangles, image too big for LDS) through the available paths."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mcrt_pkg
pkg = mcrt_pkg.load()
flat = pkg.FlatScene.load(os.path.join(ROOT, "tests", "golden", "scene_config4_2k.npz"))
for name, mode in (("auto", pkg.ACCEL_AUTO), ("octree", pkg.ACCEL_OCTREE), ("scan", pkg.ACCEL_SCAN), ("brute", pkg.ACCEL_BRUTE)):
    be = pkg.Backend(0); be.upload_scene(flat)
    be.set_camera(pkg.host_camera(1024, 1024, 90.0, (0.0, 0.0, 5.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0)))
    try:
        be.set_accel(mode)
    except pkg.McrtError as e:
        print(name, "refused:", e); be.close(); continue
    ms = []
    for _ in range(2):
        be.clear(); be.render(0, 32, 100, 0x5EED0001); ms.append(be.last_render_ms())
    info = be.octree_info()
    print(name, round(1024 * 1024 * 32 / (min(ms) * 1e-3) / 1e6), "Msamples/s", "lds" if info.lds_resident else "hbm", info.blob_bytes, flush=True)
    be.close()
