#!/bin/bash
# the one-lane-per-item slot kernel (-DMVBA_FS, tools/ab/libmvba_fs*.so): parity of one LM trial on a slot-form scene, K3 time at config 3, knock-outs
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
run() { echo "== $*"; env "$@" timeout -k 10 300 python tools/time_schur.py || exit 1; }
MVBA_LIBRARY=$PWD/tools/ab/libmvba_fs.so timeout -k 10 200 python tools/debug_schur_forms.py 20000 60 0.15 slots || exit 1
for v in fs fs_ko_gather fs_ko_dma fs_ko_valu; do
  run MVBA_LIBRARY=$PWD/tools/ab/libmvba_$v.so
  run MVBA_LIBRARY=$PWD/tools/ab/libmvba_$v.so MVBA_SLOT_SEG=0
done
run MVBA_LIBRARY=$PWD/tools/ab/libmvba_fs.so MVBA_SLOT_SKEW=24576
run FOO=1
