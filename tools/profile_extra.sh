#!/bin/bash
# kernel-trace stats of (a) the config-5 SVD leg alone and (b) config 4's per-GPU shard (1.25M points x 500 cameras x 5 %)
# usage: tools/profile_extra.sh TAG    -> gpurun_out/TAG_svd_kernel_stats_config5.txt, gpurun_out/TAG_kernel_stats_config4_shard.txt
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
tag=$1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_svd -- python tools/svd_config5.py > gpurun_out/prof_${tag}_svd.log 2>&1 || { tail -5 gpurun_out/prof_${tag}_svd.log; exit 1; }
python tools/kstats.py gpurun_out/prof_${tag}_svd/*/*kernel_stats.csv > gpurun_out/${tag}_svd_kernel_stats_config5.txt
tail -1 gpurun_out/prof_${tag}_svd.log >> gpurun_out/${tag}_svd_kernel_stats_config5.txt
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_c4 -- python bench.py --points 1250000 --cams 500 --vis 0.05 --steps 4 --warmup 1 --no-cpu-baseline --svd-rows 0 > gpurun_out/${tag}_bench_config4_shard.json 2> gpurun_out/prof_${tag}_c4.err || { tail -5 gpurun_out/prof_${tag}_c4.err; exit 1; }
python tools/kstats.py gpurun_out/prof_${tag}_c4/*/*kernel_stats.csv > gpurun_out/${tag}_kernel_stats_config4_shard.txt
cat gpurun_out/${tag}_svd_kernel_stats_config5.txt; head -14 gpurun_out/${tag}_kernel_stats_config4_shard.txt
