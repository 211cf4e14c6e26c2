#!/bin/bash
# K1 (k_resid_jac) against its block size on several camera counts: beyond ~100 cameras one block per CU fits beside the LDS
# camera table and its wave count decides the occupancy (mvba_create picks it; MVBA_K1_THREADS overrides)
# usage (on a GPU box): bash tools/sweep_k1.sh
cd $GRAFT_REPO_ROOT
run() { label="$1"; shift; ( for kv in "$@"; do export "$kv"; done; timeout -k 10 200 python bench.py $SHAPE --steps 5 --warmup 2 --no-cpu-baseline --svd-rows 0 > gpurun_out/k1s.json 2>gpurun_out/k1s.err || { tail -2 gpurun_out/k1s.err; exit 0; }; python -c "
import json; d=json.load(open('gpurun_out/k1s.json')); print('$SHAPE', '$label', round(d['kernel_ms_per_step']['resid_jac'],4))" ); }
SHAPE="--config4-shard";                         for t in 512 640 704; do run "threads=$t" MVBA_K1_THREADS=$t; done; run "threads=auto"
SHAPE="--points 600000 --cams 300 --vis 0.05";   for t in 512 896; do run "threads=$t" MVBA_K1_THREADS=$t; done; run "threads=auto"
SHAPE="--points 1000000 --cams 200 --vis 0.10";  for t in 512 1024; do run "threads=$t" MVBA_K1_THREADS=$t; done; run "threads=auto"
SHAPE="";                                        for t in 512 1024; do run "threads=$t" MVBA_K1_THREADS=$t; done; run "threads=auto"
