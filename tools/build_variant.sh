#!/bin/bash
# Build a variant of libmdt_hip.so into tools/exp/ (untracked; travels to the GPU box): tools/build_variant.sh <name> [extra hipcc flags]
# The flags go to mdt_hip.hip (the workgroup-per-parameter kernels); the wave-per-parameter unit is linked as the product build has it
# (tools/build_wave_variant.sh is its counterpart).  Use it with MDT_LIB=tools/exp/lib<name>.so (A/B timing, diagnostics builds
# such as -DMDT_PROF_Q, -DMDT_STOP_AFTER=2).
# The two units share struct KP (mdt_kp.hpp): the product objects are first brought up to date by the same staleness rule as
# __graft_entry__.build_hip, so a variant never links a stale partner unit (the library also checks sizeof(KP) of both units when a
# context is created).
set -e
R=$(cd "$(dirname "$0")/.." && pwd); N=$1; shift
C=$R/mcmcdiagnostictools.jl_amd/csrc; mkdir -p $R/tools/exp
(cd $R && python3 -c "import __graft_entry__ as g; g.build_hip()")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -I $R/include "$@" -c $C/mdt_hip.hip -o $R/tools/exp/mdt_hip_$N.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/exp/lib$N.so $R/tools/exp/mdt_hip_$N.o $C/mdt_wave.o
echo $R/tools/exp/lib$N.so
