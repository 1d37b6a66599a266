"""Experiment (GPU box): octree leaf size / depth on the 1M-triangle scene - kernel time and per-sample visit counts."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import mcrt_pkg
pkg = mcrt_pkg.load()
from mcrt_amd import scenes
W, H, spp = 1024, 1024, 16
flat = scenes.config4(pkg.FlatScene.load(os.path.join(ROOT, "tests", "golden", "scene_cornell.npz")), n=1_000_000, seed=42)
be = pkg.Backend(0)
cam = pkg.host_camera(W, H, 90.0, (0, 0, 5), (0, 0, 0))
combos = [(4, 10), (8, 10), (16, 10), (8, 8), (16, 8), (32, 8), (16, 7), (32, 7), (64, 7), (64, 6)] if len(sys.argv) < 2 else [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]]
for leaf, depth in combos:
    t0 = time.time()
    be.upload_scene(flat, pkg.AccelParams(leaf, depth, 0.0)); be.set_camera(cam)
    tb = time.time() - t0
    info = be.octree_info()
    ms = []
    for _ in range(3):
        be.clear(); be.render(0, spp, 100, 0x5EED0001); ms.append(be.last_render_ms())
    be.set_counting(True); be.clear(); be.render(0, 4, 100, 0x5EED0001); c = be.counters(); be.set_counting(False)
    s = max(c["samples"], 1)
    print(f"leaf {leaf:3d} depth {depth:2d}: {np.median(ms):7.2f} ms {W*H*spp/np.median(ms)/1e3:7.1f} Ms/s  nodes {info.node_count:8d} refs {info.prim_ref_count:8d} "
          f"{info.blob_bytes/1e6:6.0f} MB  per sample: nodes {c['nodes']/s:6.1f} tris {c['tris']/s:6.1f}  build {tb:4.1f} s", flush=True)
