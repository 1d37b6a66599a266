"""Experiment (GPU box): sample slices per pixel (mcrt_set_lanes_per_pixel) on the headline workload - round tail against end-of-launch balance."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import mcrt_pkg
pkg = mcrt_pkg.load()
W = H = 1024
spp = 256
flat = pkg.FlatScene.load(os.path.join(ROOT, "tests", "golden", "scene_config2.npz"))
be = pkg.Backend(0)
be.upload_scene(flat)
be.set_camera(pkg.host_camera(W, H, 90.0, (0, 0, 5), (0, 0, 0)))
for lanes in [int(a) for a in sys.argv[1:]] or [0, 1, 2, 4, 8, 16]:
    be.set_lanes_per_pixel(lanes)
    ms = []
    for _ in range(4):
        be.clear(); be.render(0, spp, 100, 0x5EED0001); ms.append(be.last_render_ms())
    m = float(np.median(ms[1:]))
    print(f"lanes_per_pixel {lanes or 'auto':>4}: {m:7.2f} ms  {W * H * spp / m / 1e3:7.1f} Msamples/s", flush=True)
