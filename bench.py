#!/usr/bin/env python3
"""bench.py - Msamples/s of the path-tracing hot path on N MI355X GPUs of one node.

One step = one full render pass of the workload (clear, render every pixel, reduce across ranks,
resolve to BGRA8) with the scene already resident in HBM.  A sample is one jittered camera ray and
its complete path tree.

  --gpus N          N ranks, one per GPU.  Under torchrun (WORLD_SIZE set) N must equal WORLD_SIZE;
                    started plainly with N > 1 this script launches the N ranks itself (a fresh
                    `python -m torch.distributed.run` child, before anything touches a GPU) and exits
                    with the child's status.
  --scaling weak    every rank renders `spp` samples per pixel of its own (global sample indices
                    rank*spp ...): per-GPU work fixed (default; BASELINE configs 2 and 4).
  --scaling strong  the workload's `spp` is the TOTAL, split over the ranks by global sample index
                    (BASELINE configs 3 and 5: 1024 / 4096 spp over 8 GPUs).
Either way the double accumulators are summed with ONE RCCL reduce to rank 0 per pass, rank 0 resolves,
and value = all samples of the step / max-over-ranks time.

Default workload (BASELINE.json configs[1]): Cornell box + the 36-triangle box-stack mesh, 1024x1024,
256 spp, depth 100.  Prints ONE JSON line on rank 0.

At N = 1 the line also carries
  "secondary"     the HBM-bound workload (config4_1024_16: the 1M-triangle mesh of BASELINE config 4, short run) timed in
                  the same process after the headline - 1 warm-up + 3 steps - with its own `roofline` (bound "hbm"): the
                  "traversal HBM GB/s" half of BASELINE's metric (--no-secondary leaves it out);
  "cpu_baseline"  the reference's own TraceRay on the host cores, timed BEFORE the GPU phase (--no-cpu-baseline).
Fields that come from a committed rocprofv3 pass (profiles/r*_pmc.json) are used only when that file records the
SHA-256 of the very libmcrt_hip.so that is loaded now; otherwise they are left out and the line says so.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

SEED = 0x5EED0001
DEPTH = 100
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_CLOCK_HZ = 2.4e9          # max engine clock, same guide
VALU_CYCLES_PER_WAVE_INSTR = 2  # a wave64 VALU instruction occupies a SIMD-32 for 2 cycles (same guide: `v_fma_f32 (wave64) 2 cyc`)

WORKLOADS = {
    "cornell_512_64": ("cornell", 512, 512, 64),            # BASELINE config 1 (the CPU-runnable case)
    "config2_1024_256": ("config2", 1024, 1024, 256),       # BASELINE config 2: the bench default
    "config3_1024_1024": ("config3", 1024, 1024, 1024),     # BASELINE config 3 (use --scaling strong at N > 1)
    "cornell_4k_64": ("cornell", 3840, 2160, 64),           # one progressive pass of BASELINE config 5
    "cornell_4k_4096": ("cornell", 3840, 2160, 4096),       # BASELINE config 5 in full, 64-spp passes (use --scaling strong)
    "config4_2048_256": ("config4", 2048, 2048, 256),       # BASELINE config 4: 1M-triangle mesh (octree, HBM/L2 bound)
    "config4_1024_16": ("config4", 1024, 1024, 16),         # the same scene, short run
}
SECONDARY = "config4_1024_16"                              # the HBM-bound workload every N = 1 run also times ("secondary")
PASS_SPP = {"cornell_4k_4096": 64}                          # progressive workloads: samples per pass (one reduce + resolve each)


def algorithmic_bytes_per_sample(cnt, spp):
    """node 32 B, triangle 36 B, sphere 16 B, mesh gate 24 B per visit/test,
    plus the accumulator read-modify-write (24 B per pixel and pass) and the 4-byte BGRA8 store."""
    s = max(cnt["samples"], 1)
    return (cnt["nodes"] * 32 + cnt["tris"] * 36 + cnt["spheres"] * 16 + cnt["gates"] * 24) / s + 24.0 / spp + 4.0 / spp


def lib_sha256(path):
    import hashlib

    h = hashlib.sha256()
    with open(path, "rb") as f:
        for chunk in iter(lambda: f.read(1 << 20), b""):
            h.update(chunk)
    return h.hexdigest()


def committed_pmc(workload, lib_hash):
    """(entry, repo-relative path, why-not) of the newest committed rocprofv3 PMC summary of this workload
    (profiles/r*_pmc.json, written by tools/summarize_profiles.py from separate --pmc passes of this very command).
    Hardware counters cannot be read from inside this process; everything taken from here is tagged with its source.
    The summary records the SHA-256 of the library it profiled ("_lib_sha256"): counters of another binary say nothing
    about this one, so an entry whose hash differs (or is missing) is not used."""
    import glob

    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc.json")), reverse=True):
        try:
            doc = json.load(open(path))
        except (OSError, ValueError):
            continue
        entry = doc.get(workload)
        if not entry:
            continue
        rel = os.path.relpath(path, ROOT)
        if doc.get("_lib_sha256") != lib_hash:
            return None, rel, f"{rel} profiled another build of libmcrt_hip.so (sha256 {str(doc.get('_lib_sha256'))[:16]}, loaded {lib_hash[:16]}): counter-derived fields left out"
        return entry, rel, None
    return None, None, "no committed PMC summary of this workload"


def host_cpu_budget():
    """Threads the CPU baseline may use: the reference takes std::thread::hardware_concurrency() (source/main.cpp:59);
    inside a container that is the CPU share the cgroup grants, not the machine's CPU count."""
    n = os.cpu_count() or 1
    info = {"os_cpu_count": n}
    try:
        info["affinity"] = len(os.sched_getaffinity(0))
        n = min(n, info["affinity"])
    except (AttributeError, OSError):
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            info["cgroup_cpus"] = float(quota) / float(period)
            n = min(n, max(1, int(info["cgroup_cpus"] + 0.5)))
    except (OSError, ValueError):
        pass
    omp = os.environ.get("OMP_NUM_THREADS")
    if omp and omp.isdigit():                  # the GPU box exports its per-job CPU share this way
        info["omp_num_threads"] = int(omp)
        n = min(n, max(1, int(omp)))
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return n, model, info


def spawn_ranks(args):
    """`python bench.py --gpus N` outside torchrun: start the N ranks as a fresh child (this process has not imported
    torch or touched a GPU) and hand its exit status back."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print(f"[bench] launching {args.gpus} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    sys.exit(subprocess.run(cmd, env=env).returncode)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="config2_1024_256", choices=sorted(WORKLOADS))
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--accel", default="auto", choices=["auto", "octree", "scan", "brute", "flat"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--spp", type=int, default=0, help="override the workload's samples per pixel (rehearsals of the progressive workloads at a reduced pass count; the line says so)")
    ap.add_argument("--no-secondary", action="store_true", help="leave out the HBM-bound secondary workload (N = 1 runs time it after the headline)")
    ap.add_argument("--no-pmc-fields", action="store_true", help="leave out everything read from profiles/ (used while profiling)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be at least 1")

    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            spawn_ranks(args)                  # never returns
        world, rank, local_rank = 1, 0, 0
    else:
        world = int(os.environ["WORLD_SIZE"])
        rank = int(os.environ.get("RANK", "0"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if world != args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")

    def note(msg):
        if rank == 0:
            print(f"[bench +{time.perf_counter() - T0:6.1f}s] {msg}", file=sys.stderr, flush=True)

    T0 = time.perf_counter()
    note("importing torch (the first import on a fresh box takes 1-2 minutes)")
    import numpy as np
    import torch

    import mcrt_pkg
    note("torch imported")

    pkg = mcrt_pkg.load()
    from mcrt_amd import sharding

    # MCRT_BENCH_REHEARSE=1: every rank on GPU 0 with gloo, to walk the N>1 code path on a one-GPU box
    # (the numbers mean nothing); the real run is one rank per GPU over RCCL.
    rehearse = os.environ.get("MCRT_BENCH_REHEARSE") == "1"
    n_dev = torch.cuda.device_count()
    if n_dev < 1:
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    if not rehearse and local_rank >= n_dev:
        raise SystemExit(f"bench.py: rank {rank} wants GPU {local_rank} but this node has {n_dev} (MCRT_BENCH_REHEARSE=1 puts every rank on GPU 0)")
    device_index = 0 if rehearse else local_rank
    torch.cuda.set_device(device_index)
    dist = None
    if world > 1:
        import torch.distributed as dist

        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", device_index))

    lib_hash = lib_sha256(os.environ.get("MCRT_LIB") or pkg.LIB_PATH)
    n_cu = torch.cuda.get_device_properties(device_index).multi_processor_count
    be = pkg.Backend(device_index)
    stream = torch.cuda.current_stream()
    be.set_stream(stream.cuda_stream)

    def load_scene(scene_name):
        if scene_name == "config4":
            from mcrt_amd import scenes

            note("building the 1,000,000-triangle mesh")
            return scenes.config4(pkg.FlatScene.load(os.path.join(ROOT, "tests", "golden", "scene_cornell.npz")), n=1_000_000, seed=42)
        return pkg.FlatScene.load(os.path.join(ROOT, "tests", "golden", f"scene_{scene_name}.npz"))

    def run_workload(workload, flat, steps, warmup, accel, scaling, device_build=False):
        """Upload, warm up, time `steps` steps, then (outside the timed region) the duration of the dominant kernel launch and the
        kernel's own work counters.  Returns the fields of one bench line."""
        scene_name, width, height, spp = WORKLOADS[workload]
        if args.spp > 0 and workload == args.workload:
            spp = args.spp
        cam = pkg.host_camera(width, height, 90.0, (0.0, 0.0, 5.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0))
        be.upload_scene(flat, device_build=device_build)
        be.set_camera(cam)
        be.set_accel({"auto": pkg.ACCEL_AUTO, "octree": pkg.ACCEL_OCTREE, "scan": pkg.ACCEL_SCAN, "brute": pkg.ACCEL_BRUTE, "flat": pkg.ACCEL_FLAT}[accel])
        accum = torch.zeros(height * width * 3, dtype=torch.float64, device="cuda")
        be.bind_accumulator(accum.data_ptr())

        # the sample plan of this rank: a list of passes (first global sample index, count) and the job's total spp
        plan = sharding.SamplePlan(rank, world, spp, scaling, pass_spp=PASS_SPP.get(workload))
        my_samples_per_step = sum(n for _, n in plan.passes)

        def step():
            be.clear()
            for k, (first, count) in enumerate(plan.passes):
                if count:
                    be.render(first, count, DEPTH, SEED)
                # one collective per pass: the W*H*3 doubles of every rank summed into rank 0 (RCCL over xGMI), which resolves;
                # the other ranks start their next pass from zero (their sums now live on rank 0)
                if rehearse and dist is not None:
                    host = accum.cpu()
                    sharding.reduce_accumulator(host, dist, dst=0)
                    if rank == 0:
                        accum.copy_(host)
                else:
                    sharding.reduce_accumulator(accum, dist, dst=0)
                if k + 1 < len(plan.passes):
                    sharding.after_reduce(accum, dist, dst=0)
                if rank == 0:
                    be.resolve(pkg.TONEMAP_REINHARD, plan.spp_after(k + 1), want_mean=False)

        def fence():
            if dist is not None:
                dist.barrier()
            torch.cuda.synchronize()
            be.synchronize()

        note(f"{workload}: scene uploaded; {len(plan.passes)} pass(es) of this rank per step, {my_samples_per_step} spp; warming up")
        for _ in range(warmup):
            step()
        fence()
        note(f"{workload}: timing")
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        fence()
        elapsed = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearse else "cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())

        note(f"{workload}: timed {steps} steps in {elapsed:.3f} s; measuring kernel time and counters")
        # --- outside the timed region: duration of the dominant kernel launch (HIP events on the launch stream, recorded by
        # the library around k_render) and the kernel's own work counters
        first, count = max(plan.passes, key=lambda p: p[1])
        k_ms = []
        for _ in range(3):
            be.clear(); be.render(first, count, DEPTH, SEED)
            k_ms.append(be.last_render_ms())
        k_ms = float(np.median(k_ms))
        count_spp = min(count, 8)
        be.set_counting(True)
        be.clear(); be.render(first, count_spp, DEPTH, SEED)
        cnt = be.counters()
        be.set_counting(False)
        info = be.octree_info()
        be.bind_accumulator(0)                  # `accum` goes out of scope with this function

        samples_per_step = width * height * plan.total_spp
        value = samples_per_step * steps / elapsed / 1e6
        launch_samples = width * height * count
        bytes_per_sample = algorithmic_bytes_per_sample(cnt, count)
        launch_bytes = bytes_per_sample * launch_samples
        hbm_achieved = launch_bytes / (k_ms * 1e-3) / 1e9
        resident = bool(info.lds_resident) or len(flat.triangles) <= 256      # geometry served from SGPRs / LDS, not from HBM
        pmc, pmc_path, pmc_why = (None, None, "not requested") if (args.no_pmc_fields or accel != "auto") else committed_pmc(workload, lib_hash)

        per_sample = {k: round(v / max(cnt["samples"], 1), 4) for k, v in cnt.items() if k != "samples"}
        hbm_algorithmic = {"achieved": round(hbm_achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(hbm_achieved / HBM_PEAK_GBS, 6),
                           "algorithmic_bytes_per_sample": round(bytes_per_sample, 2), "per_sample": per_sample}
        if resident:
            # VALU-bound kernel: achieved = wave-level VALU instructions per second, peak = every SIMD issuing one wave64
            # VALU instruction every 2 cycles at the peak clock.  The instruction count per sample is a hardware counter
            # (SQ_INSTS_VALU of the committed PMC pass of this workload and this binary); the launch duration is measured here.
            valu_peak = n_cu * 4 * PEAK_CLOCK_HZ / VALU_CYCLES_PER_WAVE_INSTR / 1e9
            roof = {"bound": "valu", "achieved": None, "peak": round(valu_peak, 1), "unit": "G wave-instr/s", "frac": None, "traffic": None,
                    "kernel": "k_render", "kernel_ms": round(k_ms, 4)}
            if pmc and pmc.get("valu_insts_per_sample"):
                achieved = pmc["valu_insts_per_sample"] * launch_samples / (k_ms * 1e-3) / 1e9
                roof.update({
                    "achieved": round(achieved, 2), "frac": round(achieved / valu_peak, 4),
                    "traffic": pmc.get("traffic_bytes_per_sample") and int(pmc["traffic_bytes_per_sample"] * launch_samples),
                    "valu_insts_per_sample": pmc["valu_insts_per_sample"], "active_lanes_per_valu_instruction": pmc.get("active_lanes_per_valu_instruction"),
                    "valu_cycle_weighted_frac_in_profile": pmc.get("valu_cycle_weighted_frac"),
                    "source": f"{pmc_path}: valu_insts_per_sample, traffic_bytes_per_sample, active_lanes (rocprofv3 --pmc passes of this command on this binary, sha256 {lib_hash[:16]}); kernel_ms measured in this run",
                })
            else:
                roof["pmc"] = pmc_why
            roof["hbm_algorithmic"] = hbm_algorithmic
            roof["note"] = ("scene tables live in SGPRs / LDS, so the kernel is bound by VALU issue, not by HBM: frac = SQ_INSTS_VALU x 2 cycles / "
                            "(CUs x 4 SIMDs x 2.4 GHz x kernel time).  hbm_algorithmic is SURVEY.md 8(d)'s byte figure (one record per test, served "
                            "on chip) against the HBM peak, kept as a secondary: it is not a bound and may exceed 1")
        else:
            roof = {"bound": "hbm", "achieved": round(hbm_achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(hbm_achieved / HBM_PEAK_GBS, 6),
                    "traffic": None, "kernel": "k_render", "kernel_ms": round(k_ms, 4), "algorithmic_bytes_per_sample": round(bytes_per_sample, 2),
                    "per_sample": per_sample}
            if pmc:
                roof.update({"traffic": pmc.get("traffic_bytes_per_sample") and int(pmc["traffic_bytes_per_sample"] * launch_samples),
                             "active_lanes_per_valu_instruction": pmc.get("active_lanes_per_valu_instruction"),
                             "source": f"{pmc_path}: traffic_bytes_per_sample, active_lanes (rocprofv3 --pmc passes of this command on this binary, sha256 {lib_hash[:16]}); kernel_ms and counts measured in this run"})
            else:
                roof["pmc"] = pmc_why
            roof["note"] = ("scene image (%.0f MB) is walked from HBM / L2 by divergent lanes; achieved = SURVEY.md 8(d) algorithmic bytes / kernel time"
                            % (info.blob_bytes / 1e6))
        return {
            "value": round(value, 3),
            "unit": "Msamples/s",
            "steps": steps,
            "warmup": warmup,
            "ms_per_step": round(elapsed / steps * 1e3, 4),
            "config": {
                "workload": (f"{workload}: {scene_name} scene, {width}x{height}, "
                             + (f"{spp} spp per GPU" if scaling == "weak" else f"{spp} spp in total, split over {world} GPU(s)")
                             + f", depth {DEPTH}, accel {accel}" + (" (spp overridden with --spp)" if spp != WORKLOADS[workload][3] else "")),
                "objects": int(len(flat.objects)), "triangles": int(len(flat.triangles)), "octree_nodes": int(info.node_count),
                "scene_in_lds": bool(info.lds_resident), "lanes_per_pixel": be.lanes_per_pixel(count), "passes_per_step": len(plan.passes),
                "total_spp": plan.total_spp, "parallelism": f"spp-sharded x{world}",
            },
            "roofline": roof,
        }

    scene_name, width, height, _ = WORKLOADS[args.workload]
    flat = load_scene(scene_name)

    # The CPU baseline runs FIRST (host cores only), so that the GPU phase - headline, then the secondary - is the last
    # and uninterrupted stretch of the run.
    baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        note("timing the CPU baseline on the host cores")
        baseline = cpu_baseline(pkg, flat, scene_name, width, height, note)

    head = run_workload(args.workload, flat, args.steps, args.warmup, args.accel, args.scaling)
    out = {
        "metric": "Msamples/s (whole node) Cornell box at 1/2/4/8 GPU; traversal HBM GB/s",
        "value": head["value"],
        "unit": "Msamples/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": head["ms_per_step"],
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": "f32 geometry / f64 radiance",
        "data": "synthetic",
        "config": head["config"],
        "roofline": head["roofline"],
        "lib_sha256": lib_hash,
    }
    if baseline is not None:
        out["cpu_baseline"] = baseline

    # The HBM-bound half of the metric ("traversal HBM GB/s"): the 1M-triangle scene of BASELINE config 4, short run, timed
    # here so that the driver's own run carries it.  The octree is built on the device (45 ms; the host build takes 1.7 s).
    if world == 1 and not args.no_secondary and args.workload != SECONDARY and args.accel == "auto":
        flat2 = load_scene(WORKLOADS[SECONDARY][0])
        sec = run_workload(SECONDARY, flat2, 3, 1, "auto", "weak", device_build=True)
        sec["metric"] = "Msamples/s on the HBM-resident scene; roofline.achieved = traversal GB/s (SURVEY.md 8d algorithmic bytes / k_render time)"
        out["secondary"] = sec

    if rank == 0:
        print(json.dumps(out), flush=True)
    be.close()
    if dist is not None:
        dist.destroy_process_group()


def cpu_baseline(pkg, flat, scene_name, width, height, note):
    """The reference's algorithm on the host cores, beside the GPU number (a baseline, not the target).

    Configs 1-3, 5: the reference's own Scene::TraceRay with its brute-force IntersectRay (oracle/_ref, kind "reference";
    the C restatement, kind "port", where that build is absent), one thread per CPU this job may use (the reference
    starts hardware_concurrency() threads, source/main.cpp:59), sequential-pixel atomic scheduler (source/main.cpp:89-95),
    same scene and image, a bounded number of samples per pixel, median of three runs.
    Config 4 (1M triangles): brute force is infeasible (10^6 triangle tests per ray), so it is timed on a 64x64x1
    subsample, and the restatement's octree path (the same flat octree the GPU walks, replayed on the CPU) carries
    the baseline; both are labelled."""
    import numpy as np

    import pyoracle as po

    threads, model, cpu_info = host_cpu_budget()
    pos, tgt = (0.0, 0.0, 5.0), (0.0, 0.0, 0.0)

    def timed(eng, scene, cam, spp, runs=3):
        rates = []
        for _ in range(runs):
            t0 = time.perf_counter()
            eng.render(scene, cam, 0, spp, DEPTH, SEED, threads=threads)
            rates.append(cam.width * cam.height * spp / (time.perf_counter() - t0) / 1e6)
        return float(np.median(rates)), rates

    def sized_spp(eng, scene, cam, seconds):
        """Samples per pixel that make one run last about `seconds` (pilot run at 1 spp on a 1/16 image)."""
        small = eng.camera(max(width // 4, 16), max(height // 4, 16), 90.0, pos, tgt)
        t0 = time.perf_counter()
        eng.render(scene, small, 0, 2, DEPTH, SEED, threads=threads)
        rate = small.width * small.height * 2 / (time.perf_counter() - t0)
        return max(1, min(256, int(rate * seconds / (cam.width * cam.height))))

    if scene_name != "config4":
        if po.Ref.available():
            eng, kind = po.Ref(), "reference"
            scene = eng.scene(flat, reference_octree=False)
        else:
            eng, kind = po.Oracle(), "port"
            scene = eng.scene(flat)
        cam = eng.camera(width, height, 90.0, pos, tgt)
        spp = sized_spp(eng, scene, cam, 6.0)
        med, rates = timed(eng, scene, cam, spp)
        return {
            "value": round(med, 4), "unit": "Msamples/s", "cores": threads, "kind": kind, "cpu_model": model, "cpu_budget": cpu_info,
            "runs_msamples_per_s": [round(r, 4) for r in rates],
            "sample": (f"same scene and image ({scene_name}, {width}x{height}) at {spp} spp = {width * height * spp} samples per run, brute-force "
                       f"IntersectRay as the reference does, {threads} threads, median of {len(rates)} runs"),
        }

    # ---- config 4
    orc = po.Oracle()
    scene = orc.scene(flat)
    note("config-4 CPU baseline: brute force on a 64x64x1 subsample")
    sub = orc.camera(64, 64, 90.0, pos, tgt)
    brute_eng, brute_kind = (po.Ref(), "reference") if po.Ref.available() else (orc, "port")
    brute_scene = brute_eng.scene(flat, reference_octree=False) if brute_kind == "reference" else scene
    bcam = brute_eng.camera(64, 64, 90.0, pos, tgt)
    t0 = time.perf_counter()
    brute_eng.render(brute_scene, bcam, 0, 1, DEPTH, SEED, threads=threads)
    brute_rate = 64 * 64 / (time.perf_counter() - t0) / 1e6
    note("config-4 CPU baseline: the restatement through the flat octree")
    tree_info, box, words, refs = pkg.build_scene_image(flat)
    orc.attach_octree(scene, tree_info, box, words, refs)
    cam = orc.camera(width, height, 90.0, pos, tgt)
    spp = sized_spp(orc, scene, cam, 6.0)
    med, rates = timed(orc, scene, cam, spp)
    return {
        "value": round(med, 4), "unit": "Msamples/s", "cores": threads, "kind": "port", "cpu_model": model, "cpu_budget": cpu_info,
        "runs_msamples_per_s": [round(r, 4) for r in rates],
        "sample": (f"C restatement of the reference's TraceRay with nearest-hit queries through the same flat octree the GPU walks (the reference's own "
                   f"brute force is infeasible at 10^6 triangles), {scene_name} at {width}x{height}, {spp} spp = {width * height * spp} samples per run, "
                   f"{threads} threads, median of {len(rates)} runs"),
        "brute_force": {"value": round(brute_rate, 6), "unit": "Msamples/s", "kind": brute_kind, "cores": threads,
                        "sample": "the reference's brute-force IntersectRay over all 1,000,012 triangles on a 64x64 image at 1 spp = 4096 samples, one run"},
    }


if __name__ == "__main__":
    main()
