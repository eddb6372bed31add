#!/bin/bash
# ab_env.sh ROUNDS "ENV=VAL ..." ["ENV=VAL ..." ...]: bench.py with the current library under each environment in turn
# (an empty string = the default), alternating, on one box; prints rate and mean launch time per class.
ROUNDS=$1; shift
OUT=gpurun_out/ab; mkdir -p $OUT
for r in $(seq $ROUNDS); do
  i=0
  for e in "$@"; do
    i=$((i+1))
    env $e python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras --no-traffic > $OUT/env${i}_$r.json 2>$OUT/env${i}_$r.err || { tail -3 $OUT/env${i}_$r.err; exit 1; }
    echo "[$e]"; python tools/classes_line.py $OUT/env${i}_$r.json | head -1
  done
done
