#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the fused kernel for one library variant: tools/pmc_rw.sh <lib.so> [params]   (run inside gpurun)
R=${GRAFT_REPO_ROOT:-$(pwd)}; L=$1; P=${2:-10000}
O=$R/gpurun_out/pmc_rw_$(basename $L .so); rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export MDT_LIB=$R/$L
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/f -- python3 $R/tools/run_once.py $P 2 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/w -- python3 $R/tools/run_once.py $P 2 > /dev/null 2>&1
python3 - <<PY
import csv, glob
def mean(d):
    v=[float(r["Counter_Value"]) for f in glob.glob("$O/"+d+"/*/*counter_collection.csv") for r in csv.DictReader(open(f)) if "mdt_kernel" in r["Kernel_Name"]]
    return sum(v)/len(v)
f, w = mean("f"), mean("w")
print("$(basename $L): FETCH raw %.2f GB (x2 = %.2f GB)  WRITE %.2f GB  total %.2f GB per launch" % (f*1024/1e9, 2*f*1024/1e9, w*1024/1e9, (2*f+w)*1024/1e9))
PY
