"""Turn the rocprofv3 outputs that a GPU run left under gpurun_out/ into the files kept under profiles/:
kernel-trace stats CSVs, the FETCH_SIZE / WRITE_SIZE counter CSVs and r01_pmc_traffic.json (per-launch HBM bytes)."""
import csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
rnd = sys.argv[1] if len(sys.argv) > 1 else "r01"
newest = lambda pat: max(glob.glob(pat), key=os.path.getmtime)
out = {}
for w in ("config2_1024_256", "config4_1024_16", "cornell_512_64"):
    vals = {}
    for kind in ("fetch", "write"):
        f = newest(os.path.join(G, f"{rnd}_{kind}_{w}", "*", "*_counter_collection.csv"))
        v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "k_render" in r["Kernel_Name"]]
        full = sorted(x for x in v if x > 0.5 * max(v))          # the counting pass (8 spp) is the small one
        vals[kind] = full[len(full) // 2]
        shutil.copy(f, os.path.join(P, f"{rnd}_pmc_{kind}_size_{w}.csv"))
    fetch, write = vals["fetch"] * 1024 * 2, vals["write"] * 1024
    out[w] = {"fetch_bytes": int(fetch), "write_bytes": int(write), "traffic_bytes": int(fetch + write),
              "raw_fetch_kib": vals["fetch"], "raw_write_kib": vals["write"]}
# SQ pass: instruction counts of the full-spp k_render launches (median) and what they say about the VALU
for w in list(out):
    pat = os.path.join(G, f"{rnd}_sq_{w}", "*", "*_counter_collection.csv")
    if not glob.glob(pat):
        continue
    f = newest(pat)
    per = {}
    for r in csv.DictReader(open(f)):
        if "k_render" in r["Kernel_Name"]:
            per.setdefault(r["Counter_Name"], []).append((float(r["Counter_Value"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9))
    med = {}
    for k, v in per.items():
        full = sorted(x for x in v if x[0] > 0.5 * max(y[0] for y in v))
        med[k] = full[len(full) // 2]
    valu, secs = med["SQ_INSTS_VALU"]
    out[w]["sq"] = {k: v[0] for k, v in med.items()}
    out[w]["sq"]["kernel_s"] = secs
    # 1024 SIMDs (256 CUs x 4), one wave64 VALU instruction occupies a SIMD for 4 cycles, 2.4 GHz peak engine clock
    out[w]["valu_issue_busy"] = round(valu * 4 / (1024 * secs * 2.4e9), 3)
    out[w]["active_lanes_per_valu_instruction"] = round(med["SQ_THREAD_CYCLES_VALU"][0] / valu, 1)
    shutil.copy(f, os.path.join(P, f"{rnd}_pmc_sq_{w}.csv"))
out["_notes"] = ("FETCH_SIZE / WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE undercounts by 2x (MI355X_MICROARCH.md, HBM/rocprofv3 "
                 "section), corrected here. Per k_render launch, median over the full-spp launches of one bench.py run; separate --pmc passes. "
                 "valu_issue_busy = SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x kernel time x 2.4 GHz).")
json.dump(out, open(os.path.join(P, f"{rnd}_pmc_traffic.json"), "w"), indent=1)
for tag, src in (("config2", f"{rnd}_stats_c2"), ("config4", f"{rnd}_stats_c4")):
    shutil.copy(newest(os.path.join(G, src, "*", "*_kernel_stats.csv")), os.path.join(P, f"{rnd}_kernel_stats_{tag}.csv"))
for f in glob.glob(os.path.join(G, "fin_*.json")):
    if os.path.getsize(f):
        shutil.copy(f, os.path.join(P, f"{rnd}_bench_" + os.path.basename(f)[4:]))
print(json.dumps(out, indent=1))
