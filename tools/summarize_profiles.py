"""Turn the rocprofv3 outputs that tools/profile_pmc.sh left under gpurun_out/ into the files kept under profiles/:
kernel-trace stats CSVs, the counter CSVs and <round>_pmc.json (per-sample instruction counts and HBM bytes of k_render,
which bench.py reports with their source).

  usage: python tools/summarize_profiles.py <tag> [<out-prefix>]     e.g.  r02  ->  profiles/r02_pmc.json

Units and corrections (MI355X_MICROARCH.md, HBM / rocprofv3 sections): FETCH_SIZE and WRITE_SIZE are in KiB and summed
over the XCDs; on gfx950 FETCH_SIZE reports half of the bytes read, so it is doubled; SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_*
and SQ_BUSY_CYCLES count quad-cycles; every pass is a run of its own.
"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
prefix = sys.argv[2] if len(sys.argv) > 2 else tag
CUS, SIMDS, CLOCK = 256, 4, 2.4e9


def newest(pat):
    files = glob.glob(pat)
    return max(files, key=os.path.getmtime) if files else None


def render_rows(path):
    """Per counter: [(value, seconds)] of the k_render dispatches of one pass."""
    per = {}
    for r in csv.DictReader(open(path)):
        if "k_render" in r["Kernel_Name"]:
            per.setdefault(r["Counter_Name"], []).append((float(r["Counter_Value"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9))
    return per


def full_launch_median(rows):
    """Median over the full-spp launches (bench.py also runs one short counting launch of 8 spp: the small one)."""
    top = max(v for v, _ in rows)
    full = sorted(x for x in rows if x[0] > 0.5 * top)
    return full[len(full) // 2]


out = {}
workloads = sorted({os.path.basename(d)[len(tag) + 5:] for d in glob.glob(os.path.join(G, f"{tag}_sq1_*")) if os.path.isdir(d)})
for w in workloads:
    bench = None
    try:
        bench = json.loads(open(os.path.join(G, f"{tag}_stats_{w}.json")).read().strip().splitlines()[-1])
    except (OSError, ValueError, IndexError):
        pass
    entry = {}
    sq = {}
    secs = None
    for p in ("sq1", "sq2", "grbm"):
        f = newest(os.path.join(G, f"{tag}_{p}_{w}", "*", "*_counter_collection.csv"))
        if not f:
            continue
        for k, rows in render_rows(f).items():
            v, s = full_launch_median(rows)
            sq[k] = v
            if k == "SQ_INSTS_VALU":
                secs = s
        shutil.copy(f, os.path.join(P, f"{prefix}_pmc_{p}_{w}.csv"))
    for kind in ("fetch", "write"):
        f = newest(os.path.join(G, f"{tag}_{kind}_{w}", "*", "*_counter_collection.csv"))
        if not f:
            continue
        rows = [x for rows in render_rows(f).values() for x in rows]
        v, _ = full_launch_median(rows)
        entry[f"raw_{kind}_kib"] = v
        entry[f"{kind}_bytes"] = int(v * 1024 * (2 if kind == "fetch" else 1))
        shutil.copy(f, os.path.join(P, f"{prefix}_pmc_{kind}_size_{w}.csv"))
    f = newest(os.path.join(G, f"{tag}_stats_{w}", "*", "*_kernel_stats.csv"))
    if f:
        shutil.copy(f, os.path.join(P, f"{prefix}_kernel_stats_{w}.csv"))
    if bench:
        json.dump(bench, open(os.path.join(P, f"{prefix}_bench_{w}.json"), "w"))
    # samples of one full launch: from the bench line of the stats pass (same command)
    samples = None
    if bench:
        import re

        m = re.search(r"(\d+)x(\d+), (\d+) spp", bench["config"]["workload"])
        if m:
            samples = int(m.group(1)) * int(m.group(2)) * int(m.group(3)) // max(1, bench["config"].get("passes_per_step", 1))
    entry["samples_per_launch"] = samples
    entry["sq"] = sq
    entry["kernel_s_in_sq1_pass"] = secs
    if "fetch_bytes" in entry and "write_bytes" in entry:
        entry["traffic_bytes"] = entry["fetch_bytes"] + entry["write_bytes"]
        if samples:
            entry["traffic_bytes_per_sample"] = entry["traffic_bytes"] / samples
    if "SQ_INSTS_VALU" in sq:
        valu = sq["SQ_INSTS_VALU"]
        if samples:
            entry["valu_insts_per_sample"] = valu / samples
        if "SQ_THREAD_CYCLES_VALU" in sq:
            entry["active_lanes_per_valu_instruction"] = round(sq["SQ_THREAD_CYCLES_VALU"] / valu, 2)
        if secs:
            # issue-slot utilisation in the profiled launch itself: 2 cycles per wave64 instruction on a SIMD-32
            entry["valu_frac_in_profile"] = round(valu * 2 / (CUS * SIMDS * CLOCK * secs), 4)
            # the same with the documented slower rates: f64 at half rate (4 cycles), transcendentals at quarter rate (8)
            f64 = sum(sq.get(k, 0.0) for k in ("SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64"))
            trans = sq.get("SQ_INSTS_VALU_TRANS_F32", 0.0)
            if f64 or trans:
                entry["valu_cycle_weighted_frac"] = round((valu * 2 + f64 * 2 + trans * 6) / (CUS * SIMDS * CLOCK * secs), 4)
        if "SQ_WAVE_CYCLES" in sq:
            for k in ("SQ_WAIT_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY"):
                if k in sq:
                    entry[k.lower() + "_over_wave_cycles"] = round(sq[k] / sq["SQ_WAVE_CYCLES"], 4)
    if "GRBM_GUI_ACTIVE" in sq and secs:
        entry["effective_clock_ghz"] = round(sq["GRBM_GUI_ACTIVE"] / 8 / secs / 1e9, 3)
    out[w] = entry
    if bench and bench.get("lib_sha256"):
        if out.get("_lib_sha256", bench["lib_sha256"]) != bench["lib_sha256"]:
            raise SystemExit(f"{w} was profiled with another build of libmcrt_hip.so than the workloads before it")
        out["_lib_sha256"] = bench["lib_sha256"]       # bench.py uses this summary only with the same binary loaded
out["_notes"] = ("Per k_render launch (median over the full-spp launches of one `bench.py --workload W` run per rocprofv3 pass; passes are separate "
                 "runs: sq1, sq2, FETCH_SIZE, WRITE_SIZE, GRBM_GUI_ACTIVE, kernel-trace stats). FETCH_SIZE / WRITE_SIZE are KiB; FETCH_SIZE is "
                 "doubled (gfx950 reports half the bytes read, MI355X_MICROARCH.md). valu_frac_in_profile = SQ_INSTS_VALU x 2 cycles / "
                 "(256 CUs x 4 SIMDs x 2.4 GHz x the launch's duration in the sq1 pass); valu_cycle_weighted_frac adds 2 more cycles per f64 "
                 "instruction (half rate) and 6 more per f32 transcendental (quarter rate). bench.py uses valu_insts_per_sample and "
                 "traffic_bytes_per_sample with the kernel time it measures itself.")
json.dump(out, open(os.path.join(P, f"{prefix}_pmc.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
