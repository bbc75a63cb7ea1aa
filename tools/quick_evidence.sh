#!/bin/bash
# bench line (all configs) + rocprofv3 kernel stats of the headline-only command
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=${1:-r04b}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R && python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $O/bench_under_rocprof.json 2>/dev/null
find $O -type f ! -name "*.csv" ! -name "*.txt" ! -name "*.json" ! -name "*.err" -delete 2>/dev/null
for f in $(find $O -type f -name "*.csv"); do case "$f" in *kernel_stats.csv|*agent_info.csv) continue;; esac; head -1 "$f" > "$f.tmp"; grep -E "mdt_|copy8|order_kernel|kp_setup" "$f" >> "$f.tmp"; mv "$f.tmp" "$f"; done
cat $O/bench.json | cut -c1-400
