#!/bin/bash
# Diagnostics: SQ counters, memory-side traffic and duration of one launch cut short after phase k (libraries built with
# tools/build_variant.sh stopK -DMDT_STOP_AFTER=K: 10 = first ranking pass, 1 = ranking of x, 2 = + bulk series,
# 3 = + folded ranking and its moments; the production library = everything) -- per-phase figures by difference.
#   tools/pmc_phases.sh [cfg2|cfg3|cfg5] [params]        (run inside gpurun; FETCH_SIZE counts half the bytes on gfx950)
R=${GRAFT_REPO_ROOT:-$(pwd)}; CFG=${1:-cfg2}; P=${2:-10000}; O=$R/gpurun_out/pp_$CFG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in stop10 stop1 stop2 stop3 full; do
  if [ $v = full ]; then unset MDT_LIB; else export MDT_LIB=$R/tools/exp/lib$v.so; [ -f $MDT_LIB ] || continue; fi
  echo "== $v"
  for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES" "FETCH_SIZE" "WRITE_SIZE"; do
    tag=${v}_$(echo $grp | tr ' ' '_' | cut -c1-30)
    rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/$tag -- python3 $R/tools/run_once.py $P 2 $CFG > $O/$tag.log 2>&1
    for f in $O/$tag/*/*counter_collection.csv; do
      python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "mdt_kernel" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(f"{k:28s} {sum(v)/len(v):.4g}  (n={len(v)})")
PY
    done
    rm -rf $O/$tag
  done
done
