"""Diagnostic (GPU box): where a wavefront of k_render spends its time on an LDS-resident scene and how many of its lanes have work there.
Needs the instrumented build:  hipcc ... -DMCRT_SECTION_PROFILE -o build/ab/prof.so   and   MCRT_LIB=build/ab/prof.so python tools/section_profile.py [scene] [spp]
Times are reference-clock ticks summed over wavefronts (shares of the loop time, not seconds)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import mcrt_pkg

pkg = mcrt_pkg.load()
NAMES = ["iter", "pre_t", "pre_lanes", "cast_t", "cast_lanes", "scan_t", "scan_steps", "scan_lanes", "deal_t", "deal_steps", "deal_lanes", "post_t",
         "post_trace_lanes", "post_shadow_lanes", "post_heavy", "post_heavy_lanes", "waves", "total_t", "idle_fetch", "idle_done", "idle_batch"]
scene = sys.argv[1] if len(sys.argv) > 1 else "config2"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 64
W = H = 1024
flat = pkg.FlatScene.load(os.path.join(ROOT, "tests", "golden", f"scene_{scene}.npz"))
be = pkg.Backend(0)
be.upload_scene(flat)
be.set_camera(pkg.host_camera(W, H, 90.0, (0, 0, 5), (0, 0, 0)))
lib = be.lib
out = (C.c_uint64 * 32)()
be.clear(); be.render(0, spp, 100, 0x5EED0001); be.last_render_ms()      # warm-up (waited for: its wavefronts must not add to the counters read below)
assert lib.mcrt_debug_section_profile(out) == 0
be.clear(); be.render(0, spp, 100, 0x5EED0001)
ms = be.last_render_ms()
assert lib.mcrt_debug_section_profile(out) == 0
p = dict(zip(NAMES, [int(x) for x in out]))
it = max(p["iter"], 1)
tot = max(p["total_t"], 1)
print(f"{scene} {W}x{H}x{spp}: {ms:.2f} ms with the instrumented build, {p['waves']} wavefronts, {it} loop iterations, {W*H*spp/it*1.0:.3f} samples per iteration")
rest = p["cast_t"] - p["scan_t"] - p["deal_t"]
rows = [("pre (next sample / item)", p["pre_t"], p["pre_lanes"] / it, "lanes starting a sample or fetching"),
        ("cast: object loop, gates, spheres", rest, p["cast_lanes"] / it, "lanes with a ray"),
        ("cast: lock-step triangle pairs", p["scan_t"], p["scan_lanes"] / max(p["scan_steps"], 1), "lanes that passed the mesh gate, per pair step"),
        ("cast: dealt (ray, triangle) pairs", p["deal_t"], p["deal_lanes"] / max(p["deal_steps"], 1), "pairs per deal step (64 = full)"),
        ("post (shading)", p["post_t"], (p["post_trace_lanes"] + p["post_shadow_lanes"]) / it, "lanes with a hit to shade")]
acc = 0
for name, t, lanes, what in rows:
    acc += t
    print(f"  {name:36s} {100.0 * t / tot:5.1f} % of wavefront time   {lanes:5.1f} {what}")
print(f"  {'loop control, waiting lanes, tail':36s} {100.0 * (tot - acc) / tot:5.1f} %")
print(f"  post: {p['post_trace_lanes'] / it:.1f} lanes after a camera/bounce ray, {p['post_shadow_lanes'] / it:.1f} after a shadow ray per iteration; a diffuse hit is shaded in "
      f"{100.0 * p['post_heavy'] / it:.0f} % of the iterations, by {p['post_heavy_lanes'] / max(p['post_heavy'], 1):.1f} lanes")
print(f"  scan steps per iteration {p['scan_steps'] / it:.2f}, deal steps per iteration {p['deal_steps'] / it:.2f}")
print(f"  lanes without a ray per iteration: {p['idle_fetch'] / it:.1f} waiting to fetch with the others of their round, {p['idle_batch'] / it:.1f} trace lanes waiting for their batch, "
      f"{p['idle_done'] / it:.1f} out of work (end of the launch)")
