"""Experiment (GPU box): the octree render kernel on triangle soups of growing size - is the 1M-triangle case bound by memory?
Prints kernel time, rays/s and per-sample visit counts for n triangles in the same box (image size of the octree in MB)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import mcrt_pkg
pkg = mcrt_pkg.load()
from mcrt_amd import scenes
W, H, spp = 1024, 1024, 16
base = pkg.FlatScene.load(os.path.join(ROOT, "tests", "golden", "scene_cornell.npz"))
be = pkg.Backend(0)
cam = pkg.host_camera(W, H, 90.0, (0, 0, 5), (0, 0, 0))
for n in [int(a) for a in sys.argv[1:]] or [10_000, 100_000, 300_000, 1_000_000]:
    flat = scenes.config4(base, n=n, seed=42)
    be.upload_scene(flat); be.set_camera(cam); be.set_accel(pkg.ACCEL_OCTREE)
    info = be.octree_info()
    ms = []
    for _ in range(3):
        be.clear(); be.render(0, spp, 100, 0x5EED0001); ms.append(be.last_render_ms())
    be.set_counting(True); be.clear(); be.render(0, 4, 100, 0x5EED0001); c = be.counters(); be.set_counting(False)
    s = max(c["samples"], 1)
    m = float(np.median(ms))
    print(f"n {n:8d}: {m:7.2f} ms {W*H*spp/m/1e3:7.1f} Ms/s  {c['rays']/s*W*H*spp/m/1e3:7.1f} Mrays/s  image {info.blob_bytes/1e6:6.0f} MB lds {info.lds_resident}  per sample: rays {c['rays']/s:5.2f} nodes {c['nodes']/s:6.1f} tris {c['tris']/s:6.1f}"
          f"  per ray: nodes {c['nodes']/max(c['rays'],1):5.1f} tris {c['tris']/max(c['rays'],1):5.1f}", flush=True)
