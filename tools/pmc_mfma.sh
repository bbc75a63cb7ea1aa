#!/bin/bash
# MFMA counters: the headline with the matrix form of the z-series products forced (libz2) against the default; config 3 (default)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4q; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() { # tag lib cfg params
  tag=$1; lib=$2; cfg=$3; P=$4
  if [ -n "$lib" ]; then export MDT_LIB=$R/$lib; else unset MDT_LIB; fi
  echo "== $tag"
  for grp in "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_WAIT_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE"; do
    t=$O/$tag; rm -rf $t
    rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $t -- python3 $R/tools/run_once.py $P 2 $cfg > $t.log 2>&1
    for f in $t/*/*counter_collection.csv; do
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
    rm -rf $t
  done
}
run cfg2_vector_products "" cfg2 10000
run cfg2_matrix_products tools/exp/libz2.so cfg2 10000
run cfg3_matrix_products "" cfg3 1000
