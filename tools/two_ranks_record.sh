#!/bin/bash
# The N > 1 control flow of bench.py on a ONE-GPU box, through its own launcher (no torch): two ranks sharing the GPU
# over the TCP transport (PAOS_BENCH_REHEARSAL=1; the rate means nothing), then the same command asking for RCCL, which
# cannot come up here (rank 1 has no device 1): exit 3 with a JSON line that says how far the bring-up got.
# Output: gpurun_out/two_ranks.txt (copied to profiles/rNN_bench_two_ranks_one_gpu.txt).
OUT=gpurun_out/two_ranks.txt; mkdir -p gpurun_out
{
  echo "## PAOS_BENCH_REHEARSAL=1 python bench.py --gpus 2 --allow-tcp --batch 16 --steps 5 --warmup 1 --no-cpu-baseline --no-extras --no-traffic"
  PAOS_BENCH_REHEARSAL=1 python bench.py --gpus 2 --allow-tcp --batch 16 --steps 5 --warmup 1 --no-cpu-baseline --no-extras --no-traffic 2>gpurun_out/two_ranks_a.err | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print(json.dumps({k:d[k] for k in ('metric','value','unit','n_gpus','steps','ms_per_step','scaling')}))
print('config:', json.dumps({k:d['config'][k] for k in ('parallelism','transport','launcher','ranks_seen','devices_seen','batch_per_gpu')}))
print('roofline.frac', round(d['roofline']['frac'],3), '| sweep rendered per step', d['sweep']['record_sets_rendered_per_step'])
"
  echo "exit code: ${PIPESTATUS[0]}"; echo "stderr:"; tail -5 gpurun_out/two_ranks_a.err
  echo
  echo "## python bench.py --gpus 2 --grid 1024 --batch 4 --steps 2 --no-cpu-baseline --no-extras --no-traffic   (RCCL asked for on a one-GPU box)"
  python bench.py --gpus 2 --grid 1024 --batch 4 --steps 2 --no-cpu-baseline --no-extras --no-traffic 2>gpurun_out/two_ranks_b.err
  echo "exit code: $?"; echo "stderr:"; tail -8 gpurun_out/two_ranks_b.err
} > $OUT 2>&1
cat $OUT
