#!/bin/bash
# what bounds k_schur_pairs on config 4's shard (1.25 M x 500 x 5 %): the product kernel and its knock-out builds (tools/build_hrec_timing.sh)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for v in tree ko_gather ko_lside ko_prow ko_dma ko_valu ko_valu_gather ko_valu_dma; do
  if [ $v = tree ]; then unset MVBA_LIBRARY; else export MVBA_LIBRARY=$PWD/tools/ab2/libmvba_$v.so; fi
  timeout -k 10 300 python tools/time_schur.py 1250000 500 0.05 4 2> gpurun_out/koc4_$v.err || { tail -3 gpurun_out/koc4_$v.err; exit 1; }
done
