"""Experiment: what the Oren-Nayar BRDF (double-precision trig, parity-mandated) costs: config 2 as is vs all-Lambertian."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mcrt_pkg
pkg = mcrt_pkg.load()
def run(flat):
    be = pkg.Backend(0); be.upload_scene(flat)
    be.set_camera(pkg.host_camera(1024, 1024, 90.0, (0.0, 0.0, 5.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0)))
    ms = []
    for _ in range(3):
        be.clear(); be.render(0, 256, 100, 0x5EED0001); ms.append(be.last_render_ms())
    be.close()
    return 1024 * 1024 * 256 / (min(ms) * 1e-3) / 1e6
flat = pkg.FlatScene.load(os.path.join(ROOT, "tests", "golden", "scene_config2.npz"))
print("as is", round(run(flat)))
lam = pkg.FlatScene(flat.objects.copy(), flat.triangles.copy(), flat.background)
lam.objects["material"]["diffuse"][:] = 0
print("all Lambertian", round(run(lam)))
nos = pkg.FlatScene(lam.objects.copy(), lam.triangles.copy(), lam.background)
nos.objects["material"]["surface"][:] = 0
print("all Lambertian diffuse (no mirror / glass)", round(run(nos)))
