#!/bin/bash
# Build a variant of libmdt_hip.so into tools/exp/ (untracked; travels to the GPU box): tools/build_variant.sh <name> [extra hipcc flags]
# Use it with MDT_LIB=tools/exp/lib<name>.so (A/B timing, diagnostics builds such as -DMDT_PROF_Q).
R=$(cd "$(dirname "$0")/.." && pwd); N=$1; shift
mkdir -p $R/tools/exp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -shared -fPIC -I $R/include "$@" -o $R/tools/exp/lib$N.so $R/mcmcdiagnostictools.jl_amd/csrc/mdt_hip.hip
