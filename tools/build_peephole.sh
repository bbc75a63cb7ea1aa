#!/bin/bash
# Build libmdt_hip.so with the VCC peephole (tools/vcc_peephole.py) applied to the device assembly:
#   tools/build_peephole.sh <out.so> [extra hipcc flags]
# device: hipcc -S -> peephole -> assemble -> link (hsaco) -> bundle; host: hipcc --cuda-host-only with that bundle.
set -e
R=$(cd "$(dirname "$0")/.." && pwd); OUT=$1; shift
PEEP_ARGS=${PEEP_ARGS:-}
W=$(mktemp -d /tmp/mdtpeep.XXXXXX)
LL=/opt/rocm/lib/llvm/bin
SRC=$R/mcmcdiagnostictools.jl_amd/csrc/mdt_hip.hip
FL="--offload-arch=gfx950 -O3 -ffp-contract=off -I $R/include $@"
if [ -n "$PEEP_DEV_S" ] && [ -f "$PEEP_DEV_S" ]; then cp $PEEP_DEV_S $W/dev.s; else /opt/rocm/bin/hipcc $FL -S --cuda-device-only -o $W/dev.s $SRC 2>/dev/null; [ -n "$PEEP_DEV_S" ] && cp $W/dev.s $PEEP_DEV_S; fi
python3 $R/tools/vcc_peephole.py $W/dev.s $W/dev2.s $PEEP_ARGS
$LL/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $W/dev2.s -o $W/dev2.o
$LL/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o $W/dev2.out $W/dev2.o
$LL/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input=$W/dev2.out -output=$W/dev2.hipfb
/opt/rocm/bin/hipcc $FL -fPIC --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang $W/dev2.hipfb -c $SRC -o $W/host.o
/opt/rocm/bin/hipcc -shared -fPIC -o $OUT $W/host.o
rm -rf $W
echo "built $OUT"
