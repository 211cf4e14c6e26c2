#!/bin/bash
# kernel stats of the wide-matrix factorisation (tools/time_svd_wide.py <rows> <n>...) -> gpurun_out/<tag>_svd_wide_kernel_stats.txt
# usage: tools/prof_svd_wide.sh <tag> <rows> <n> [dtype]
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
tag=$1; rows=$2; n=$3; dt=${4:-f64}
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_svdw_${tag} -- python tools/time_svd_wide.py $rows $n --dtype $dt --reps 3 > gpurun_out/${tag}_svd_wide.log 2> gpurun_out/${tag}_svd_wide.err || { tail -5 gpurun_out/${tag}_svd_wide.err; exit 1; }
cp $(find gpurun_out/prof_svdw_${tag} -name "*kernel_stats.csv" | head -1) gpurun_out/${tag}_svd_wide_kernel_stats.csv
python tools/kstats.py gpurun_out/${tag}_svd_wide_kernel_stats.csv > gpurun_out/${tag}_svd_wide_kernel_stats.txt
cat gpurun_out/${tag}_svd_wide.log; head -16 gpurun_out/${tag}_svd_wide_kernel_stats.txt
