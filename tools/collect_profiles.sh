#!/bin/bash
# Collect the round's evidence on the GPU box: bench line, rocprofv3 kernel stats, calibrated PMC traffic.
# usage (inside gpurun): bash tools/collect_profiles.sh <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=${1:-r01}
O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R && python bench.py --steps 10 --warmup 2 > $O/bench.json 2> $O/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_under_rocprof.json 2>/dev/null
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/tools/run_once.py 10000 2 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/tools/run_once.py 10000 2 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_cal -- $R/tools/scatter_bench > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_cal -- $R/tools/scatter_bench > /dev/null 2>&1
cat $O/bench.json
