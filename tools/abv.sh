#!/bin/bash
# timing of several library variants on the same box: tools/abv.sh <rounds> <libA.so> <libB.so> ...   (paths relative to the repo root)
N=$1; shift
for r in $(seq $N); do
  for v in "$@"; do
    echo -n "$(basename $v) "
    MDT_LIB=$PWD/$v python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])"
  done
done
