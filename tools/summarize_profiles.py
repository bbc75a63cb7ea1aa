#!/usr/bin/env python3
"""Turn gpurun_out/<tag>/ (tools/collect_profiles.sh) into the committed summaries under profiles/."""
import csv, glob, json, os, shutil, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join("gpurun_out", tag)
os.makedirs("profiles", exist_ok=True)
shutil.copy(os.path.join(src, "bench.json"), f"profiles/{tag}_bench.json")
for f in glob.glob(os.path.join(src, "stats", "*", "*kernel_stats.csv")):
    shutil.copy(f, f"profiles/{tag}_kernel_stats.csv")
def rows(d, pat):
    out = []
    for f in glob.glob(os.path.join(src, d, "*", "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if pat in r["Kernel_Name"]:
                out.append(r)
    return out
def mean(d, pat):
    v = [float(r["Counter_Value"]) for r in rows(d, pat)]
    return sum(v) / len(v)
fetch_kb, write_kb = mean("pmc_fetch", "mdt_kernel"), mean("pmc_write", "mdt_kernel")
cal = {}
for d, key in (("pmc_fetch_cal", "fetch"), ("pmc_write_cal", "write")):
    rs = [r for r in rows(d, "copy8") if int(r["Grid_Size"]) // int(r["Workgroup_Size"]) == 256 and r["Workgroup_Size"] == "1024"]
    known_kb = 256 * 40000 * 8 * 20 / 1024
    cal[key] = sum(float(r["Counter_Value"]) for r in rs) / len(rs) / known_kb
for d in ("pmc_fetch", "pmc_write", "pmc_fetch_cal", "pmc_write_cal"):
    with open(f"profiles/{tag}_{d}.csv", "w") as fo:
        w = csv.writer(fo); w.writerow(["Kernel_Name", "Grid_Size", "Workgroup_Size", "Counter_Name", "Counter_Value"])
        for r in rows(d, "mdt_kernel") + rows(d, "copy8"):
            w.writerow([r["Kernel_Name"][:60], r["Grid_Size"], r["Workgroup_Size"], r["Counter_Name"], r["Counter_Value"]])
traffic = fetch_kb * 1024 / cal["fetch"] + write_kb * 1024 / cal["write"]
out = {"draws": 10000, "chains": 4, "params": 10000, "kernel": "mdt_kernel<double>",
       "FETCH_SIZE_KB_raw": fetch_kb, "WRITE_SIZE_KB_raw": write_kb,
       "calibration": {"fetch_counter_over_known": cal["fetch"], "write_counter_over_known": cal["write"],
                       "how": "same session, tools/scatter_bench copy8 (8 B/lane coalesced, 256 WG x 1024 threads, 6.55 GB known each way)"},
       "hbm_bytes_per_launch": traffic,
       "note": "memory-side (L2<->fabric) bytes per launch; Infinity-Cache hits are included (upper bound on HBM traffic). Algorithmic bytes per launch: 3.2e9."}
json.dump(out, open("profiles/traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
