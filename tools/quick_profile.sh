#!/bin/bash
# kernel-trace stats of the default bench workload (no PMC); summary -> gpurun_out/<tag>_kernel_stats.txt
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
tag=$1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag} -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/prof_${tag}.json 2> gpurun_out/prof_${tag}.err || exit 1
cp gpurun_out/prof_${tag}/*/*kernel_stats.csv gpurun_out/${tag}_kernel_stats.csv
python tools/kstats.py gpurun_out/${tag}_kernel_stats.csv > gpurun_out/${tag}_kernel_stats.txt
head -8 gpurun_out/${tag}_kernel_stats.txt
python -c "
import json,sys
d=json.loads(open('gpurun_out/prof_${tag}.json').read()); print(d['value'], d['ms_per_step'], d['kernel_ms_per_step'])"
