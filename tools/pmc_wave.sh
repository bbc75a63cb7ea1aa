#!/bin/bash
# Diagnostics: SQ counters and memory-side traffic of the wave-per-parameter kernel on DRAWS x CHAINS x PARAMS (separate passes per
# counter group, see /opt/skills/guides/MI355X_MICROARCH.md):   tools/pmc_wave.sh 1000 4 100000     (run inside gpurun)
R=${GRAFT_REPO_ROOT:-$(pwd)}; D=${1:-1000}; C=${2:-4}; P=${3:-100000}; O=$R/gpurun_out/pmc_wave; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/$tag -- python3 $R/tools/time_shape.py $D $C $P > $O/$tag.log 2>&1
  for f in $O/$tag/*/*counter_collection.csv; do
    python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "wave_kernel" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(f"{k:28s} {sum(v)/len(v):.5g}  (n={len(v)})")
PY
  done
  rm -rf $O/$tag
done
