#!/bin/bash
# tools/abe.sh <rounds> "<ENV=.. ENV=.. lib>" ... : timing of (environment, library) variants on the same box
N=$1; shift
for r in $(seq $N); do
  for v in "$@"; do
    lib=${v##* }; envs=${v% *}; [ "$envs" = "$v" ] && envs=""
    echo -n "[$envs] $(basename $lib) "
    env $envs MDT_LIB=$PWD/$lib python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])"
  done
done
