#!/bin/bash
# per-kernel durations (rocprofv3 --kernel-trace --stats) of the default bench workload for one MVBA_CHOL mode
# usage: tools/kstats_mode.sh MODE
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
[ "$1" = default ] || export MVBA_CHOL=$1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_chol_$1 -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/prof_chol_$1.json 2> gpurun_out/prof_chol_$1.err || exit 1
python tools/kstats.py gpurun_out/prof_chol_$1/*/*kernel_stats.csv | grep -E "chol|compact"
