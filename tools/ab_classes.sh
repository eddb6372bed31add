#!/bin/bash
# ab_classes.sh ROUNDS NAME...: like ab_variants.sh, but prints the mean launch time per class of pass launch.
ROUNDS=$1; shift
OUT=gpurun_out/ab; mkdir -p $OUT
for r in $(seq $ROUNDS); do
  for v in "$@"; do
    PAOS_LIB=$PWD/build/variants/$v.so python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras --no-traffic > $OUT/${v}_$r.json 2>$OUT/${v}_$r.err || { tail -3 $OUT/${v}_$r.err; exit 1; }
    python tools/classes_line.py $OUT/${v}_$r.json | head -1
  done
done
python tools/classes_line.py /dev/null 2>/dev/null | tail -1
