#!/bin/bash
# Does a field that fits the 256 MiB Infinity Cache make the passes cheaper?  The same chain at small batch sizes, with the
# HBM counters (bench.py's own traffic children): time per launch and item, bytes per launch and item.
set -e
mkdir -p gpurun_out/mall
for spec in "4096 fp64 1" "4096 fp64 2" "4096 fp64 8" "4096 fp32 1" "4096 fp32 2" "2048 fp64 1" "2048 fp64 2" "2048 fp64 4" "2048 fp64 16" "2048 fp32 4" "1024 fp64 4" "1024 fp64 16" "1024 fp64 64"; do
  set -- $spec
  out=gpurun_out/mall/g$1_$2_b$3.json
  timeout -k 10 300 python bench.py --grid $1 --precision $2 --batch $3 --steps 10 --warmup 2 --no-cpu-baseline --no-extras > $out 2> gpurun_out/mall/g$1_$2_b$3.err
  python - "$out" "$spec" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]; b = d["config"]["batch_per_gpu"] if "batch_per_gpu" in d["config"] else None
al = r.get("all_launches", {})
print(sys.argv[2], "| %.1f wavefronts/s | full launch %.4f ms, frac %.3f, traffic/alg %s | all launches: %.2f ms, %.2f GB measured" % (
    d["value"], r["avg_launch_ms"], r["frac"], r.get("traffic_over_algorithmic"), al.get("ms", 0), al.get("bytes_measured", 0) / 1e9), flush=True)
PY
done
