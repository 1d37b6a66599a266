#!/usr/bin/env python3
"""bench.py - Msamples/s of the path-tracing hot path on N MI355X GPUs of one node.

One step = one full render pass of the workload (clear, render every pixel at
`spp` samples, reduce across ranks, resolve to BGRA8) with the scene already
resident in HBM.  At N>1 every rank renders its own `spp` samples per pixel
(global sample indices rank*spp ...), the double accumulators are summed with
one RCCL reduce, rank 0 resolves: weak scaling, value = all samples / time.

Workload (BASELINE.json configs[1]): Cornell box + the 36-triangle box-stack mesh
through the octree, 1024x1024, 256 spp, depth 100.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

SEED = 0x5EED0001
DEPTH = 100
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md

WORKLOADS = {
    "cornell_512_64": ("cornell", 512, 512, 64),            # BASELINE config 1 (the CPU-runnable case)
    "config2_1024_256": ("config2", 1024, 1024, 256),       # BASELINE config 2: the bench default
    "config3_1024_1024": ("config3", 1024, 1024, 1024),     # BASELINE config 3
    "cornell_4k_64": ("cornell", 3840, 2160, 64),           # one progressive pass of BASELINE config 5
    "config4_2048_256": ("config4", 2048, 2048, 256),       # BASELINE config 4: 1M-triangle mesh (octree, HBM/L2 bound)
    "config4_1024_16": ("config4", 1024, 1024, 16),         # the same scene, short run
}


def algorithmic_bytes_per_sample(cnt, spp):
    """node 32 B, triangle 36 B, sphere 16 B, mesh gate 24 B per visit/test,
    plus the accumulator read-modify-write (24 B per pixel and pass) and the 4-byte BGRA8 store."""
    s = max(cnt["samples"], 1)
    return (cnt["nodes"] * 32 + cnt["tris"] * 36 + cnt["spheres"] * 16 + cnt["gates"] * 24) / s + 24.0 / spp + 4.0 / spp


def measured_counters(workload):
    """The latest committed rocprofv3 PMC measurement of this workload (profiles/r*_pmc_traffic.json, written by
    tools/summarize_profiles.py): HBM bytes per k_render launch and the VALU counters.  PMC counters cannot be
    collected from inside this process, so the committed passes of the same command are reported, or null."""
    import glob

    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), reverse=True):
        try:
            entry = json.load(open(path)).get(workload)
        except (OSError, ValueError):
            continue
        if entry:
            return entry
    return {}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="config2_1024_256", choices=sorted(WORKLOADS))
    ap.add_argument("--accel", default="auto", choices=["auto", "octree", "scan", "brute", "flat"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))

    def note(msg):
        if rank == 0:
            print(f"[bench +{time.perf_counter() - T0:6.1f}s] {msg}", file=sys.stderr, flush=True)

    T0 = time.perf_counter()
    note("importing torch (the first import on a fresh box takes 1-2 minutes)")
    import torch

    import mcrt_pkg
    note("torch imported")

    pkg = mcrt_pkg.load()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    # MCRT_BENCH_REHEARSE=1: every rank on GPU 0 with gloo, to walk the N>1 code path on a one-GPU box
    # (the numbers mean nothing); the real run is one rank per GPU over RCCL.
    rehearse = os.environ.get("MCRT_BENCH_REHEARSE") == "1"
    device_index = 0 if rehearse else local_rank
    torch.cuda.set_device(device_index)
    dist = None
    if world > 1:
        import torch.distributed as dist

        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", device_index))

    scene_name, width, height, spp = WORKLOADS[args.workload]
    if scene_name == "config4":
        from mcrt_amd import scenes

        note("building the 1,000,000-triangle mesh")
        flat = scenes.config4(pkg.FlatScene.load(os.path.join(ROOT, "tests", "golden", "scene_cornell.npz")), n=1_000_000, seed=42)
    else:
        flat = pkg.FlatScene.load(os.path.join(ROOT, "tests", "golden", f"scene_{scene_name}.npz"))
    cam = pkg.host_camera(width, height, 90.0, (0.0, 0.0, 5.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0))

    be = pkg.Backend(device_index)
    stream = torch.cuda.current_stream()
    be.set_stream(stream.cuda_stream)
    be.upload_scene(flat)
    be.set_camera(cam)
    be.set_accel({"auto": pkg.ACCEL_AUTO, "octree": pkg.ACCEL_OCTREE, "scan": pkg.ACCEL_SCAN, "brute": pkg.ACCEL_BRUTE, "flat": pkg.ACCEL_FLAT}[args.accel])
    accum = torch.zeros(height * width * 3, dtype=torch.float64, device="cuda")
    be.bind_accumulator(accum.data_ptr())
    from mcrt_amd import sharding

    first_sample, _ = sharding.rank_sample_range(rank, world, spp)     # global sample indices of this rank

    def step():
        be.clear()
        be.render(first_sample, spp, DEPTH, SEED)
        if rehearse and dist is not None:
            host = accum.cpu()
            sharding.reduce_accumulator(host, dist, dst=0)
            if rank == 0:
                accum.copy_(host)
        else:
            sharding.reduce_accumulator(accum, dist, dst=0)      # RCCL over xGMI: one collective per pass
        if rank == 0:
            be.resolve(pkg.TONEMAP_REINHARD, sharding.total_spp(world, spp), want_mean=False)

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        be.synchronize()

    note("scene uploaded, warming up")
    for _ in range(args.warmup):
        step()
    fence()
    note("timing")
    kernel_ms = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        kernel_ms.append(None)                    # placeholder: event times are read after the timed region
    fence()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearse else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    note(f"timed {args.steps} steps in {elapsed:.3f} s; measuring kernel time and counters")
    # --- outside the timed region: kernel duration (HIP events on the launch stream) and work counters
    be.clear(); be.render(first_sample, spp, DEPTH, SEED)
    k_ms = [be.last_render_ms()]
    for _ in range(2):
        be.clear(); be.render(first_sample, spp, DEPTH, SEED)
        k_ms.append(be.last_render_ms())
    k_ms = float(np.median(k_ms))
    count_spp = min(spp, 8)
    be.set_counting(True)
    be.clear(); be.render(first_sample, count_spp, DEPTH, SEED)
    cnt = be.counters()
    be.set_counting(False)
    info = be.octree_info()

    samples_per_step = width * height * spp * world
    value = samples_per_step * args.steps / elapsed / 1e6
    bytes_per_sample = algorithmic_bytes_per_sample(cnt, spp)
    launch_bytes = bytes_per_sample * width * height * spp
    achieved = launch_bytes / (k_ms * 1e-3) / 1e9

    pmc = measured_counters(args.workload) if args.accel == "auto" else {}
    out = {
        "metric": "Msamples/s (whole node) Cornell box; traversal HBM GB/s",
        "value": round(value, 3),
        "unit": "Msamples/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32 geometry / f64 radiance",
        "data": "synthetic",
        "config": {
            "workload": f"{args.workload}: {scene_name} scene, {width}x{height}, {spp} spp per GPU, depth {DEPTH}, accel {args.accel}",
            "objects": int(len(flat.objects)), "triangles": int(len(flat.triangles)), "octree_nodes": int(info.node_count),
            "scene_in_lds": bool(info.lds_resident), "lanes_per_pixel": be.lanes_per_pixel(spp), "parallelism": f"spp-sharded x{world}",
        },
        "roofline": {
            "bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6),
            "traffic": pmc.get("traffic_bytes"),
            "kernel": "k_render", "kernel_ms": round(k_ms, 4), "algorithmic_bytes_per_sample": round(bytes_per_sample, 2),
            "per_sample": {k: round(v / max(cnt["samples"], 1), 4) for k, v in cnt.items() if k != "samples"},
            "valu_issue_busy": pmc.get("valu_issue_busy"), "active_lanes_per_valu_instruction": pmc.get("active_lanes_per_valu_instruction"),
            "note": ("the algorithmic bytes (SURVEY.md 8d: one record per test) are served from SGPRs / LDS, not from HBM - `traffic` is what "
                     "HBM saw per launch - so `frac` can exceed 1; the kernel is VALU-bound (`valu_issue_busy`, committed PMC pass)"
                     if (info.lds_resident or len(flat.triangles) <= 256) else
                     "scene image (%.0f MB) is walked from HBM/L2 by divergent lanes: bound by lane divergence and cache-line requests "
                     "(`valu_issue_busy`, `active_lanes_per_valu_instruction`), not by streaming bandwidth" % (info.blob_bytes / 1e6)),
        },
    }

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        note("timing the CPU baseline on the host cores")
        out["cpu_baseline"] = cpu_baseline(pkg, flat, scene_name, width, height)

    if rank == 0:
        print(json.dumps(out), flush=True)
    be.close()
    if dist is not None:
        dist.destroy_process_group()


def cpu_baseline(pkg, flat, scene_name, width, height):
    """The reference's own TraceRay (oracle/_ref, kind "reference") or, where that build is absent, the C
    restatement (kind "port"), multithreaded on the host cores: a bounded sample of the same workload."""
    import pyoracle as po

    threads = min(16, os.cpu_count() or 1)
    spp = max(8, min(256, int(1.3e8 / (width * height))))       # ~1.3e8 samples: 10-30 s of work on 16 cores
    if po.Ref.available():
        eng, kind = po.Ref(), "reference"
        cam = eng.camera(width, height, 90.0, (0.0, 0.0, 5.0), (0.0, 0.0, 0.0))
        scene = eng.scene(flat, reference_octree=False)
    else:
        eng, kind = po.Oracle(), "port"
        cam = eng.camera(width, height, 90.0, (0.0, 0.0, 5.0), (0.0, 0.0, 0.0))
        scene = eng.scene(flat)
    t0 = time.perf_counter()
    eng.render(scene, cam, 0, spp, DEPTH, SEED, threads=threads)
    dt = time.perf_counter() - t0
    n = width * height * spp
    return {
        "value": round(n / dt / 1e6, 4), "unit": "Msamples/s", "cores": threads, "kind": kind,
        "sample": f"same scene and image ({scene_name}, {width}x{height}) at {spp} spp = {n} samples, brute-force IntersectRay as the reference does, {dt:.1f} s wall",
    }


if __name__ == "__main__":
    main()
