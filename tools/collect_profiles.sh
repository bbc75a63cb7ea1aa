#!/bin/bash
# Collect the round's evidence on the GPU box: the bench line (headline, other_configs, shape sweep), rocprofv3 kernel stats of
# THE SAME full command (a row per kernel instantiation: wave_kernel<double>, mdt_small<double>, mdt_big<double>,
# mdt_small<float>, mdt_tiny<double>), calibrated memory-side traffic and the SQ instruction counters of the five measured
# workloads (separate passes per counter group), phase tables.
# usage (inside gpurun): bash tools/collect_profiles.sh <tag>      then, back home: python tools/summarize_profiles.py <tag>
# (gpurun merges the box's gpurun_out/<tag> into the local one: `rm -rf gpurun_out/<tag>` first, or files of an earlier collection stay beside the new ones)
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=${1:-r06}
O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R && python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
python tools/phase_profile.py --params 10000 > $O/phase_cfg2.txt 2>/dev/null
python tools/phase_profile.py --params 1000 --draws 100000 --chains 8 > $O/phase_cfg3.txt 2>/dev/null
if [ -f tools/exp/libwaveprof.so ]; then
  MDT_LIB=$R/tools/exp/libwaveprof.so python tools/dev_wave_prof.py 1000 4 100000 2>/dev/null | grep -v amdgpu.ids > $O/phase_wave_1000x4.txt
  MDT_LIB=$R/tools/exp/libwaveprof.so python tools/dev_wave_prof.py 500 4 100000 2>/dev/null | grep -v amdgpu.ids > $O/phase_wave_500x4.txt
fi
if [ -f tools/exp/libprofq.so ]; then  # (tools/build_variant.sh profq -DMDT_PROF_Q: the sub-slots of the ranking)
  MDT_LIB=$R/tools/exp/libprofq.so python tools/phase_profile.py --cfg5 --q --params 12500 2>/dev/null | grep -v amdgpu.ids > $O/phase_cfg5_rank.txt
fi
python tools/time_cfg5.py 12500 2>/dev/null | tail -1 > $O/cfg5_split.txt
# where the wave-per-parameter kernel starts to pay: default rule (1), forced (2), never (0)
for sh in "1000 4" "500 4" "250 4" "1000 8"; do for P in 1000 2000 4000 8000 16000 32000; do for w in 1 2 0; do
  echo -n "MDT_WAVE=$w "; MDT_TEST_HOOKS=1 MDT_WAVE=$w python tools/time_shape.py $sh $P 2>/dev/null | tail -1 | cut -c1-48; done; done; done > $O/wave_crossover.txt
python tools/time_strided.py > $O/strided.txt 2>/dev/null
python tools/time_kinds.py > $O/kinds.txt 2>/dev/null
python tools/time_kinds.py 100000 1000 4 > $O/kinds_1000x4.txt 2>/dev/null   # (round 6: partial requests on config 1's slice shape)
cd /tmp && export TMPDIR=/tmp
# the program itself directly behind `--` (no env / bash -c hop: the profiler's preload initialises the GPU first)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_under_rocprof.json 2>/dev/null
for cfg in cfg2:10000 cfg3:1000 cfg5:12500 cfg1:100000 s500:100000; do
  c=${cfg%%:*}; p=${cfg##*:}
  for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAVES"; do
    tag=$(echo $grp | tr ' ' '_' | cut -c1-24)
    rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/pmc_${c}_$tag -- python3 $R/tools/run_once.py $p 2 $c > /dev/null 2>&1
  done
done
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_cal_FETCH_SIZE -- $R/tools/scatter_bench > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_cal_WRITE_SIZE -- $R/tools/scatter_bench > /dev/null 2>&1
cd $R
# keep the merge-back small (64 MiB): of every CSV only the rows of this library's kernels (the synthetic-input generator
# launches thousands of torch kernels), nothing else of rocprofv3's output
prune() {
  find "$1" -type f ! -name "*.csv" ! -name "*.txt" ! -name "*.json" ! -name "*.err" ! -name "*.log" -delete 2>/dev/null
  for f in $(find "$1" -type f -name "*.csv"); do
    case "$f" in *kernel_stats.csv|*agent_info.csv) continue;; esac
    head -1 "$f" > "$f.tmp"; grep -E "mdt|copy8|order_kernel|kp_setup|fb_patch|wave_kernel" "$f" >> "$f.tmp"; mv "$f.tmp" "$f"
  done
}
prune $O
du -sh $O 1>&2
cat $O/bench.json | cut -c1-300
