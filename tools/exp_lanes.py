"""Experiment: kernel time of one render pass vs forced lanes-per-pixel and image size (not part of the product)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import mcrt_pkg
pkg = mcrt_pkg.load()
SEED = 0x5EED0001
def run(scene, w, h, spp, lanes):
    flat = pkg.FlatScene.load(os.path.join(ROOT, "tests", "golden", f"scene_{scene}.npz"))
    be = pkg.Backend(0)
    be.upload_scene(flat)
    be.set_camera(pkg.host_camera(w, h, 90.0, (0.0, 0.0, 5.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0)))
    be.set_lanes_per_pixel(lanes)
    ms = []
    for _ in range(3):
        be.clear(); be.render(0, spp, 100, SEED); ms.append(be.last_render_ms())
    be.close()
    return w * h * spp / (min(ms) * 1e-3) / 1e6
for scene, w, h, spp in (("cornell", 512, 512, 64), ("config2", 1024, 1024, 256), ("cornell", 1024, 1024, 256), ("cornell", 2048, 2048, 64), ("config2", 3840, 2160, 64)):
    print(scene, w, h, spp, {l: round(run(scene, w, h, spp, l)) for l in (1, 2, 4, 8, 16, 64)}, flush=True)
