#!/bin/bash
# the pacing poll as an LDS-DMA of its own (round 5): parity subset, K3 launch time of the variants, bench line
# (every GPU step behind the one before: nothing runs after a failure)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -k 10 120 python tools/time_schur.py 200000 40 0.2 4 2>gpurun_out/pace_small.err || { echo "small scene FAILED"; tail -3 gpurun_out/pace_small.err; exit 1; }
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "every_schur_kernel_form or large_scene or random_scene or config3_full" > gpurun_out/pace_t.log 2>&1 || { echo "pytest FAILED"; tail -5 gpurun_out/pace_t.log; exit 1; }
tail -2 gpurun_out/pace_t.log
for rep in 1 2; do
  for v in tree pace_none_wg pace_sc0_agent pace_sc1_agent; do
    if [ $v = tree ]; then unset MVBA_LIBRARY; else export MVBA_LIBRARY=$PWD/tools/ab3/libmvba_$v.so; fi
    python tools/time_schur.py 2>/dev/null || exit 1
    MVBA_SLOT_POLL_SLOW=1 python tools/time_schur.py 2>/dev/null | sed 's/^/  slow poll: /' || exit 1
  done
done
unset MVBA_LIBRARY
timeout -k 10 200 python bench.py --no-cpu-baseline --svd-rows 0 --depth-rows 0 --steps 20 --warmup 5 > gpurun_out/pace_b.json 2> gpurun_out/pace_b.err || { tail -3 gpurun_out/pace_b.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/pace_b.json')); print('bench', round(d['value'],1), {k: round(v,3) for k,v in d['kernel_ms_per_step'].items()})"
