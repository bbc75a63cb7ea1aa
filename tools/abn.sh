#!/bin/bash
# timing of several builds of libmdt_hip.so on the same box: tools/abn.sh <rounds> <lib>...
N=$1; shift
L=mcmcdiagnostictools.jl_amd/csrc/libmdt_hip.so
for r in $(seq $N); do
  for v in "$@"; do
    cp $v $L
    echo -n "$(basename $v) "
    python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])"
  done
done
