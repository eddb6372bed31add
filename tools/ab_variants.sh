#!/bin/bash
# ab_variants.sh ROUNDS NAME...: bench.py (4096^2 complex128, 20 steps) with each build/ab/NAME.so in turn, ROUNDS times, on ONE box.
# The variant is selected through PAOS_LIB (paos_amd/_lib.py), the shipped library is never overwritten.
ROUNDS=$1; shift
OUT=gpurun_out/ab; mkdir -p $OUT
for r in $(seq $ROUNDS); do
  for v in "$@"; do
    PAOS_LIB=$PWD/build/ab/$v.so python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-traffic --detail $OUT/${v}_$r.json > $OUT/${v}_$r.line 2>$OUT/${v}_$r.err || { tail -3 $OUT/${v}_$r.err; exit 1; }
    python tools/ab_lines.py $OUT/${v}_$r.json $v $r
  done
done
