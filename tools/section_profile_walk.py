"""Diagnostic (GPU box): where a wavefront of the render kernel spends its time on the 1M-triangle scene (octree walked from HBM, dealt
leaf records, parked state).  Needs the instrumented build: hipcc ... -DMCRT_SECTION_PROFILE -o build/ab/prof.so, then
MCRT_LIB=build/ab/prof.so python tools/section_profile_walk.py [spp]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import mcrt_pkg

pkg = mcrt_pkg.load()
from mcrt_amd import scenes

NAMES = ["iter", "pre_t", "pre_lanes", "cast_t", "cast_lanes", "scan_t", "scan_steps", "scan_lanes", "deal_t", "deal_steps", "deal_lanes", "post_t",
         "post_trace_lanes", "post_shadow_lanes", "post_heavy", "post_heavy_lanes", "waves", "total_t", "idle_fetch", "idle_done", "idle_batch"]
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 16
W = H = 1024
flat = scenes.config4(pkg.FlatScene.load(os.path.join(ROOT, "tests", "golden", "scene_cornell.npz")), n=1_000_000, seed=42)
be = pkg.Backend(0)
be.upload_scene(flat)
be.set_camera(pkg.host_camera(W, H, 90.0, (0, 0, 5), (0, 0, 0)))
out = (C.c_uint64 * 32)()
be.clear(); be.render(0, spp, 100, 0x5EED0001); be.last_render_ms()
assert be.lib.mcrt_debug_section_profile(out) == 0
be.clear(); be.render(0, spp, 100, 0x5EED0001)
ms = be.last_render_ms()
assert be.lib.mcrt_debug_section_profile(out) == 0
p = dict(zip(NAMES, [int(x) for x in out]))
it, tot, steps, deals = max(p["iter"], 1), max(p["total_t"], 1), max(p["scan_steps"], 1), max(p["deal_steps"], 1)
print(f"config4 {W}x{H}x{spp}: {ms:.2f} ms (instrumented), {p['waves']} wavefronts, {it} loop iterations, {steps} walk steps ({steps / it:.1f} per iteration), {deals} deals")
print(f"  pre / begin region   {100.0 * p['pre_t'] / tot:5.1f} % of wavefront time, {p['pre_lanes'] / it:5.1f} lanes begin a ray per iteration")
print(f"  walk (run_dealt)     {100.0 * p['cast_t'] / tot:5.1f} %, {p['cast_lanes'] / it:5.1f} lanes have a ray when it returns, {p['post_trace_lanes'] / it:5.1f} of them finished")
print(f"  shading region       {100.0 * p['post_t'] / tot:5.1f} %")
print(f"  per walk step: {p['scan_lanes'] / steps:5.1f} lanes look at a node, {64 - p['idle_done'] / steps:5.1f} lanes walking; deal in {100.0 * deals / steps:.0f} % of the steps with {p['deal_lanes'] / deals:5.1f} pairs; "
      f"{p['post_heavy_lanes'] / deals:5.1f} lanes descend (per dealing step)")
print(f"  ticks per walk step {p['cast_t'] / steps:.0f}, per shading region {p['post_t'] / it:.0f}, per pre region {p['pre_t'] / it:.0f} (reference clock, 100 MHz)")
