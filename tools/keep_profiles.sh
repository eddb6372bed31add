#!/bin/bash
# Copy the summaries of gpurun_out/profile_round (tools/profile_round.sh) into profiles/ under this round's names.
R=${1:-r02}
S=gpurun_out/profile_round
cp $S/kernel_stats.txt profiles/${R}_rocprof_kernel_stats_bench.txt
{ echo "# rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes, tools/profile_round.sh) on python3 bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-extras (4096^2 c128, batch 32)"; echo "# FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of wide coalesced reads); algorithmic bytes of one full pass over the batch: read 8.59 GB + write 8.59 GB;"; echo "# averages are over ALL launches of a shape, pruned ones (tiles / loads / stores skipped next to an aperture) included, hence below 8.59 for the shapes that carry or neighbour an aperture"; echo "## FETCH"; cat $S/pmc_fetch.txt; echo "## WRITE"; cat $S/pmc_write.txt; } > /tmp/pmc_$$.txt && mv /tmp/pmc_$$.txt profiles/${R}_pmc_hbm_traffic_bench.txt
{ echo "# rocprofv3 --pmc SQ_* on bench.py --steps 1 (4096^2 c128 batch 32), per-kernel averages per launch (524288 waves per pass launch)"; cat $S/pmc_sq.txt; } > profiles/${R}_sq_counters.txt
cp $S/fftbench.txt profiles/${R}_fftbench_pass_shapes.txt
[ -f $S/timeline.txt ] && cp $S/timeline.txt profiles/${R}_timeline_workgroup_phases.txt
cp $S/bench_default.json profiles/${R}_bench_4096_default.json
[ -f $S/bench_default_line.json ] && cp $S/bench_default_line.json profiles/${R}_bench_4096_contract_line.json
cp $S/bench_fp32.json profiles/${R}_bench_4096_fp32.json
cp $S/bench_noprune.json profiles/${R}_bench_4096_noprune.json
[ -f $S/parity_gpu_vs_oracle.txt ] && cp $S/parity_gpu_vs_oracle.txt profiles/${R}_parity_gpu_vs_oracle.txt
[ -f $S/baseline_configs.txt ] && cp $S/baseline_configs.txt profiles/${R}_baseline_configs.txt
