#!/bin/bash
# tools/build_wave_variant.sh NAME [-Dflags...]: mdt_wave.hip rebuilt with the given defines and linked with the current
# mdt_hip.o into tools/exp/libNAME.so (select it with MDT_LIB); e.g. tools/build_wave_variant.sh waveprof -DWV_PROF
# (the product objects are brought up to date first: see tools/build_variant.sh)
set -e
R=$(cd "$(dirname "$0")/.." && pwd); N=$1; shift
C=$R/mcmcdiagnostictools.jl_amd/csrc; mkdir -p $R/tools/exp
(cd $R && python3 -c "import __graft_entry__ as g; g.build_hip()")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -I $R/include "$@" -c $C/mdt_wave.hip -o $R/tools/exp/mdt_wave_$N.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/exp/lib$N.so $C/mdt_hip.o $R/tools/exp/mdt_wave_$N.o
echo $R/tools/exp/lib$N.so
