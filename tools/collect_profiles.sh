#!/bin/bash
# Collect the round's evidence on the GPU box: bench line (headline + other_configs), rocprofv3 kernel stats of the same
# command, calibrated PMC traffic for the three measured workloads, per-phase cycle tables.
# usage (inside gpurun): bash tools/collect_profiles.sh <tag>      then, back home: python tools/summarize_profiles.py <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=${1:-r04}
O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R && python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
MDT_PROF=1 python tools/phase_profile.py --params 10000 > $O/phase_cfg2.txt 2>/dev/null
MDT_PROF=1 python tools/phase_profile.py --params 1000 --draws 100000 --chains 8 > $O/phase_cfg3.txt 2>/dev/null
[ -f tools/exp/libprofq.so ] && MDT_LIB=$R/tools/exp/libprofq.so MDT_PROF=1 python tools/phase_profile.py --params 10000 --q > $O/phase_cfg2_rank.txt 2>/dev/null
[ -f tools/exp/libproff.so ] && MDT_LIB=$R/tools/exp/libproff.so MDT_PROF=1 python tools/phase_profile.py --params 10000 --f > $O/phase_cfg2_fold.txt 2>/dev/null
for v in q:rank f:fold; do
  [ -f tools/exp/libprof${v%%:*}.so ] || continue
  MDT_LIB=$R/tools/exp/libprof${v%%:*}.so MDT_PROF=1 python tools/phase_profile.py --${v%%:*} --draws 100000 --chains 8 --params 1000 > $O/phase_cfg3_${v##*:}.txt 2>/dev/null
  MDT_LIB=$R/tools/exp/libprof${v%%:*}.so MDT_PROF=1 python tools/phase_profile.py --${v%%:*} --cfg5 --params 12500 > $O/phase_cfg5_${v##*:}.txt 2>/dev/null
done
# config 1's slice shape at scale (the one-chunk path of round 4), default and with the sub-slots of the fold merge
MDT_PROF=1 python tools/phase_profile.py --params 100000 --draws 1000 --chains 4 > $O/phase_cfg1.txt 2>/dev/null
[ -f tools/exp/libproff.so ] && MDT_LIB=$R/tools/exp/libproff.so MDT_PROF=1 python tools/phase_profile.py --f --params 100000 --draws 1000 --chains 4 > $O/phase_cfg1_fold.txt 2>/dev/null
python tools/time_strided.py > $O/strided.txt 2>/dev/null
python tools/time_kinds.py > $O/kinds.txt 2>/dev/null
bash tools/pmc_sq.sh > $O/sq_counters.txt 2>/dev/null
cd /tmp && export TMPDIR=/tmp
# (--no-other-configs: configs 1-at-scale and 4-shard run the same kernel name as the headline and would sit in its --stats average)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $O/bench_under_rocprof.json 2>/dev/null
for cfg in cfg2:10000 cfg3:1000 cfg5:12500; do
  c=${cfg%%:*}; p=${cfg##*:}
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$c -- python3 $R/tools/run_once.py $p 2 $c > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_$c -- python3 $R/tools/run_once.py $p 2 $c > /dev/null 2>&1
done
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_cal -- $R/tools/scatter_bench > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_cal -- $R/tools/scatter_bench > /dev/null 2>&1
# per-phase counters (SQ, LDS conflicts, MFMA, FETCH / WRITE) of the three measured workloads, by MDT_STOP_AFTER difference
for cfg in cfg2:10000 cfg3:1000 cfg5:12500; do
  bash $R/tools/pmc_phases.sh ${cfg%%:*} ${cfg##*:} > $O/pmc_phases_${cfg%%:*}.txt 2>&1
done
[ -x $R/tools/mfma_f64_bench ] || hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o $R/tools/mfma_f64_bench $R/tools/mfma_f64_bench.hip
$R/tools/mfma_f64_bench > $O/mfma_microbench.txt 2>&1
cd $R
cat $O/bench.json
# keep the merge-back small (64 MiB): of every CSV only the rows of this library's kernels (the synthetic-input generator
# launches thousands of torch kernels), nothing else of rocprofv3's output
prune() {
  find "$1" -type f ! -name "*.csv" ! -name "*.txt" ! -name "*.json" ! -name "*.err" ! -name "*.log" -delete 2>/dev/null
  for f in $(find "$1" -type f -name "*.csv"); do
    case "$f" in *kernel_stats.csv|*agent_info.csv) continue;; esac
    head -1 "$f" > "$f.tmp"; grep -E "mdt_|copy8|order_kernel|kp_setup" "$f" >> "$f.tmp"; mv "$f.tmp" "$f"
  done
}
prune $O; prune $R/gpurun_out/sq
du -sh $R/gpurun_out 1>&2
