#!/bin/bash
# what bounds a step of k_schur_slots: the product kernel and its knock-out builds (tools/build_hrec_timing.sh), K3 launch time only
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for v in ${KO_LIST:-tree ko_idx ko_idx_dma ko_idx_valu_dma}; do
  if [ $v = tree ]; then unset MVBA_LIBRARY; elif [ $v = hrec33 ]; then export MVBA_LIBRARY=$PWD/tools/ab/libmvba_$v.so; else export MVBA_LIBRARY=$PWD/tools/ab2/libmvba_$v.so; fi
  timeout -k 10 200 python tools/time_schur.py 2> gpurun_out/ko_$v.err || tail -3 gpurun_out/ko_$v.err
  MVBA_SLOT_SEG=0 timeout -k 10 200 python tools/time_schur.py 2> gpurun_out/ko_$v.err | sed 's/^/   no pacing: /' || tail -3 gpurun_out/ko_$v.err
done
