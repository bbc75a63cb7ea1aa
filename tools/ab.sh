#!/bin/bash
# A/B timing of two builds of libmdt_hip.so on the same box: tools/ab.sh <libA> <libB> [rounds]
A=$1; B=$2; N=${3:-3}
L=mcmcdiagnostictools.jl_amd/csrc/libmdt_hip.so
for r in $(seq $N); do
  for v in $A $B; do
    cp $v $L
    echo -n "$(basename $v) "
    python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])"
  done
done
