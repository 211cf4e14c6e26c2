#!/bin/bash
# A/B of the slot kernel's interleaved gather issue (MVBA_SLOT_INTERLEAVE, tools/ab/libmvba_il0.so = the rounds 3-5 order) at config 3
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/il_parity.log 2>&1 || { tail -20 gpurun_out/il_parity.log; exit 1; }
tail -2 gpurun_out/il_parity.log
for i in 1 2 3; do
  python tools/time_schur.py || exit 1
  MVBA_LIBRARY=$PWD/tools/ab/libmvba_il0.so python tools/time_schur.py || exit 1
done
