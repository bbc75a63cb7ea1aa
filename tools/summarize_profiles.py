#!/usr/bin/env python3
"""Turn gpurun_out/<tag>/ (tools/collect_profiles.sh) into the committed summaries under profiles/:
  <tag>_bench.json, <tag>_bench_under_rocprof.json   the bench line, plain and under rocprofv3
  <tag>_kernel_stats.csv                              rocprofv3 --stats of the FULL bench command (a row per kernel instantiation)
  <tag>_kernel_dispatches.txt                         every dispatch of the engine's kernels in that run, by kernel, in launch order
  <tag>_pmc_<workload>.csv, <tag>_pmc_cal.csv         raw counter rows (FETCH_SIZE / WRITE_SIZE / SQ_*) of the engine's kernels
  <tag>_counters.txt                                  the same as a table: per workload and launch
  <tag>_phase_*.txt, <tag>_strided.txt, <tag>_kinds.txt
  traffic.json                                        what bench.py reads: calibrated memory-side bytes and the instruction-issue
                                                      counters per launch of every measured workload
"""
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
src = os.path.join("gpurun_out", tag)
os.makedirs("profiles", exist_ok=True)
for name in ("bench.json", "bench_under_rocprof.json"):
    p = os.path.join(src, name)
    if os.path.exists(p) and os.path.getsize(p):
        shutil.copy(p, f"profiles/{tag}_{name}")
for f in glob.glob(os.path.join(src, "stats", "*", "*kernel_stats.csv")):
    with open(f) as fi, open(f"profiles/{tag}_kernel_stats.csv", "w") as fo:   # (torch's generator kernels have names of several KB)
        w = csv.writer(fo)
        for row in csv.reader(fi):
            row[0] = row[0][:100]
            w.writerow(row)

ENGINE = ("mdt_kernel", "wave_kernel")
tr = glob.glob(os.path.join(src, "stats", "*", "*kernel_trace.csv"))
if tr:
    disp = {}
    for r in csv.DictReader(open(tr[0])):
        if any(k in r["Kernel_Name"] for k in ENGINE):
            key = (r["Kernel_Name"][:70], r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?")))
            disp.setdefault(key, []).append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
    try:
        under = json.loads(open(os.path.join(src, "bench_under_rocprof.json")).read().strip().splitlines()[-1])
    except Exception:
        under = None
    with open(f"profiles/{tag}_kernel_dispatches.txt", "w") as fo:
        sw = f"--steps {under['steps']} --warmup {under['warmup']} " if under else ""
        fo.write(f"# rocprofv3 --kernel-trace of `python3 bench.py {sw}--no-cpu-baseline` (the FULL line: headline, configs 3 / 5-shard / 1-at-scale /\n"
                 "# 500x4 / 4-shard, shape sweep): duration of every dispatch of the engine's kernels, ms, by (kernel, grid, workgroup) in launch order\n")
        if under:
            fo.write(f"# the same run's own line: ms_per_step {under['ms_per_step']:.3f}, roofline.kernel_ms (HIP events) {under['roofline']['kernel_ms']:.3f}\n")
            oc = under.get("other_configs", {})
            fo.write("# its other_configs (HIP events): " + ", ".join(f"{k} {v['ms']:.2f} ms" for k, v in oc.items() if isinstance(v, dict) and "ms" in v) + "\n")
        for k, v in sorted(disp.items(), key=lambda kv: min(x[0] for x in kv[1])):
            d = [x[1] for x in sorted(v)]
            fo.write(f"{k[0]} grid {k[1]} wg {k[2]}: n={len(d)} mean={sum(d) / len(d):.3f} min={min(d):.3f} max={max(d):.3f}\n  "
                     + " ".join(f"{x:.2f}" for x in d[:80]) + (" ..." if len(d) > 80 else "") + "\n")
            if under and "mdt_small" in k[0] and "double" in k[0] and k[1] == str(512 * 512) and len(d) >= under["steps"] + under["warmup"] + 3 and "headline_done" not in globals():
                headline_done = True
                # the headline's launches come first: the side-stream first call is another context (same kernel, same grid),
                # then first request, second call, warm-up, the timed steps, then 1 + steps launches with MDT_NO_ORDER=1
                w0 = under["warmup"] + 3
                t = d[w0:w0 + under["steps"]]
                fo.write(f"  headline's timed steps (dispatches {w0 + 1}..{w0 + len(t)} of this kernel and grid): mean {sum(t) / len(t):.3f} ms\n")

for name in ("phase_cfg2", "phase_cfg3", "phase_cfg5_rank", "cfg5_split", "wave_crossover", "phase_wave_1000x4", "phase_wave_500x4", "strided", "kinds", "kinds_1000x4"):
    p = os.path.join(src, name + ".txt")
    if os.path.exists(p) and os.path.getsize(p):
        shutil.copy(p, f"profiles/{tag}_{name}.txt")


def rows(pattern, pats):
    out = []
    # gpurun MERGES a collection into the local directory: passes of earlier collections stay beside the new ones (other process ids in
    # the file names) -- only the newest file of every pass directory is this collection's
    newest = {}
    for f in glob.glob(os.path.join(src, pattern, "*", "*counter_collection.csv")):
        d = os.path.dirname(f)
        if d not in newest or os.path.getmtime(f) > os.path.getmtime(newest[d]):
            newest[d] = f
    for f in newest.values():
        for r in csv.DictReader(open(f)):
            if any(p in r["Kernel_Name"] for p in pats):
                out.append(r)
    return out


# D, C, P, bytes per element, kernel, main-kernel pattern
WORK = {"cfg2": (10000, 4, 10000, 8, "mdt_small::mdt_kernel<double>", "mdt_small"),
        "cfg3": (100000, 8, 1000, 8, "mdt_big::mdt_kernel<double>", "mdt_big"),
        "cfg5": (10000, 16, 12500, 4, "mdt_small::mdt_kernel<float> (mcse(mean) + rhat_nested(rank): one fused launch, mdt_mcse_nested)", "mdt_small"),
        "cfg1": (1000, 4, 100000, 8, "mdtw::wave_kernel<double> (config 1's slice shape at scale)", "wave_kernel"),
        "s500": (500, 4, 100000, 8, "mdtw::wave_kernel<double> (500 draws x 4 chains)", "wave_kernel")}
cal = {}
for key in ("FETCH_SIZE", "WRITE_SIZE"):
    rs = [r for r in rows(f"pmc_cal_{key}", ("copy8",)) if int(r["Grid_Size"]) // int(r["Workgroup_Size"]) == 256 and r["Workgroup_Size"] == "1024"]
    known_kb = 256 * 40000 * 8 * 20 / 1024
    cal[key] = sum(float(r["Counter_Value"]) for r in rs) / len(rs) / known_kb if rs else float("nan")
with open(f"profiles/{tag}_pmc_cal.csv", "w") as fo:
    w = csv.writer(fo)
    w.writerow(["Kernel_Name", "Grid_Size", "Workgroup_Size", "Counter_Name", "Counter_Value"])
    for key in ("FETCH_SIZE", "WRITE_SIZE"):
        for r in rows(f"pmc_cal_{key}", ("copy8",)):
            w.writerow([r["Kernel_Name"][:60], r["Grid_Size"], r["Workgroup_Size"], r["Counter_Name"], r["Counter_Value"]])

out = {"profile": tag,
       "calibration": {"fetch_counter_over_known": cal["FETCH_SIZE"], "write_counter_over_known": cal["WRITE_SIZE"],
                       "how": "same session, tools/scatter_bench copy8 (8 B/lane coalesced, 256 WG x 1024 threads, 6.55 GB known each way)"},
       "note": "memory-side (L2<->fabric) bytes, Infinity-Cache hits included (an upper bound on HBM traffic); counters are means over the launches "
               "of the workload's main kernel in separate rocprofv3 --pmc passes (tools/collect_profiles.sh); issue floor: see bench.py",
       "workloads": {}}
table = []
for c, (D, C, P, ts, kern, pat) in WORK.items():
    cnt = {}
    with open(f"profiles/{tag}_pmc_{c}.csv", "w") as fo:
        w = csv.writer(fo)
        w.writerow(["Kernel_Name", "Grid_Size", "Workgroup_Size", "Counter_Name", "Counter_Value"])
        for r in rows(f"pmc_{c}_*", ENGINE):
            w.writerow([r["Kernel_Name"][:60], r["Grid_Size"], r["Workgroup_Size"], r["Counter_Name"], r["Counter_Value"]])
            if pat in r["Kernel_Name"]:
                cnt.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    m = {k: sum(v) / len(v) for k, v in cnt.items()}
    if not m:
        continue
    per_launch = m.get("FETCH_SIZE", float("nan")) * 1024 / cal["FETCH_SIZE"] + m.get("WRITE_SIZE", float("nan")) * 1024 / cal["WRITE_SIZE"]
    alg = P * D * C * ts
    out["workloads"][c] = {"draws": D, "chains": C, "params": P, "kernel": kern,
                           "FETCH_SIZE_KB_raw_per_launch": m.get("FETCH_SIZE"), "WRITE_SIZE_KB_raw_per_launch": m.get("WRITE_SIZE"),
                           "memory_side_bytes_per_launch": per_launch, "algorithmic_bytes_per_call": alg, "traffic_over_algorithmic": per_launch / alg,
                           "valu_wave_insts": m.get("SQ_INSTS_VALU"), "salu_wave_insts": m.get("SQ_INSTS_SALU"), "lds_wave_insts": m.get("SQ_INSTS_LDS"),
                           "vmem_rd_wave_insts": m.get("SQ_INSTS_VMEM_RD"), "vmem_wr_wave_insts": m.get("SQ_INSTS_VMEM_WR"),
                           "mfma_wave_insts": m.get("SQ_INSTS_MFMA"), "mfma_busy_cycles": m.get("SQ_VALU_MFMA_BUSY_CYCLES"),
                           "valu_active_quad_cycles": m.get("SQ_ACTIVE_INST_VALU"), "wave_quad_cycles": m.get("SQ_WAVE_CYCLES"),
                           "wait_any_quad_cycles": m.get("SQ_WAIT_ANY"), "wait_inst_any_quad_cycles": m.get("SQ_WAIT_INST_ANY"),
                           "lds_bank_conflict_cycles": m.get("SQ_LDS_BANK_CONFLICT"), "lds_idx_active_cycles": m.get("SQ_LDS_IDX_ACTIVE"),
                           "grbm_gui_active": m.get("GRBM_GUI_ACTIVE"), "waves": m.get("SQ_WAVES")}
    table.append(f"== {c}: {D} x {C} x {P}, {kern}")
    for k in sorted(m):
        table.append(f"  {k:28s} {m[k]:.5g}  (n={len(cnt[k])})")
    table.append(f"  memory-side bytes per launch  {per_launch:.4g}  = {per_launch / alg:.2f} x the {alg:.3g} algorithmic bytes"
                 f"  (FETCH_SIZE / {cal['FETCH_SIZE']:.5f}, WRITE_SIZE / {cal['WRITE_SIZE']:.5f})")
open(f"profiles/{tag}_counters.txt", "w").write("\n".join(table) + "\n")
# the headline's entry in the old keys as well (bench.py roofline.traffic)
if "cfg2" in out["workloads"]:
    t2 = out["workloads"]["cfg2"]
    out.update({"draws": 10000, "chains": 4, "params": 10000, "kernel": t2["kernel"], "hbm_bytes_per_launch": t2["memory_side_bytes_per_launch"]})
json.dump(out, open("profiles/traffic.json", "w"), indent=1)
print("\n".join(table))
