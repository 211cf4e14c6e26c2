#!/bin/bash
# per-kernel durations (rocprofv3 --kernel-trace --stats) of the default bench workload for a given libmvba build
# usage: tools/kstats_lib.sh path/to/libmvba.so [kernel-name-pattern]
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
export MVBA_LIBRARY=$PWD/$1
tag=$(basename $1 .so)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_lib_$tag -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/prof_lib_$tag.json 2> gpurun_out/prof_lib_$tag.err || exit 1
python tools/kstats.py gpurun_out/prof_lib_$tag/*/*kernel_stats.csv | grep -E "${2:-k_}"
