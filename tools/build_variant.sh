#!/bin/bash
# Build a variant of libmdt_hip.so into tools/exp/ (untracked; travels to the GPU box): tools/build_variant.sh <name> [extra hipcc flags]
# The flags go to mdt_hip.hip (the workgroup-per-parameter kernels); the wave-per-parameter unit is linked as built
# (tools/build_wave_variant.sh is its counterpart).  Use it with MDT_LIB=tools/exp/lib<name>.so (A/B timing, diagnostics builds
# such as -DMDT_PROF_Q, -DMDT_STOP_AFTER=2).
set -e
R=$(cd "$(dirname "$0")/.." && pwd); N=$1; shift
C=$R/mcmcdiagnostictools.jl_amd/csrc; mkdir -p $R/tools/exp
[ -f $C/mdt_wave.o ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -I $R/include -c $C/mdt_wave.hip -o $C/mdt_wave.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -I $R/include "$@" -c $C/mdt_hip.hip -o $R/tools/exp/mdt_hip_$N.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/exp/lib$N.so $R/tools/exp/mdt_hip_$N.o $C/mdt_wave.o
