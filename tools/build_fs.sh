#!/bin/bash
# Experimental builds of the slot kernel with ONE LANE PER ITEM (-DMVBA_FS, csrc/mvba_fs.h: 64 lists per wave, 81 accumulators per
# lane; parity-correct for the slot form, the unit form is NOT usable in these builds) and its knock-outs into tools/ab/ -- never
# the product library.   usage: tools/build_fs.sh     (in the build container; the .so files travel with gpurun)
set -e
cd "$(dirname "$0")/../3d-reconstruction-from-multi-view-exp_amd/csrc"
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-result"
[ -f mvsvd.o ] || /opt/rocm/bin/hipcc $F -c mvsvd.hip -o mvsvd.o
mkdir -p ../../tools/ab
b() {  # name defines...
  n=$1; shift
  /opt/rocm/bin/hipcc $F -DMVBA_FS "$@" -c mvba.hip -o /tmp/mvba_$n.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 /tmp/mvba_$n.o mvsvd.o -o ../../tools/ab/libmvba_$n.so -ldl -lpthread
}
b fs &
b fs_ko_gather -DMVBA_KO_GATHER &
b fs_ko_dma -DMVBA_KO_DMA &
b fs_ko_valu -DMVBA_KO_VALU &
wait
ls -la ../../tools/ab | grep fs
