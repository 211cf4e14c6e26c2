#!/bin/bash
# after `gpurun -- 'bash tools/final_round.sh r05_s'`: the judged artefacts from gpurun_out/ into profiles/ (run in the repo root, here)
set -e
cp gpurun_out/r05_s_pmc_config3.json profiles/pmc_config3.json
cp gpurun_out/r05_s4_pmc_config4_shard.json profiles/pmc_config4_shard.json
cp gpurun_out/r05_s_bench_{default,config3,config4_shard}.json profiles/
cp gpurun_out/r05_s_kernel_stats.txt profiles/r05_s_kernel_stats_config3.txt
cp gpurun_out/r05_s_kernel_stats.csv profiles/r05_s_kernel_stats_config3.csv
cp gpurun_out/r05_s_pmc_summary.txt profiles/r05_s_pmc_summary_config3.txt
cp gpurun_out/r05_s4_kernel_stats.txt profiles/r05_s_kernel_stats_config4_shard.txt
cp gpurun_out/r05_s4_pmc_summary.txt profiles/r05_s_pmc_summary_config4_shard.txt
cp gpurun_out/prof_r05_s.json profiles/r05_s_bench_under_rocprof.json
[ -f gpurun_out/r05_s_gputests.log ] && cp gpurun_out/r05_s_gputests.log profiles/r05_s_gputests.log
[ -f gpurun_out/fullsize_timings.txt ] && cp gpurun_out/fullsize_timings.txt profiles/r05_s_fullsize_timings.txt
python3 -c "
import bench
print(bench.pmc_traffic('k_schur_slots', 10001842))
print(bench.pmc_traffic('k_schur_pairs', 31246709))
"
