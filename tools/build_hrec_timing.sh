#!/bin/bash
# Timing-only builds of the "h in the record" slot kernel (-DMVBA_HREC_TIMING: today's records read with the new access
# pattern and arithmetic -- wrong numbers, right traffic, right instruction mix) into tools/ab*/ (never the product library).
# usage: tools/build_hrec_timing.sh     (in the build container; the .so files travel with gpurun)
set -e
cd "$(dirname "$0")/../3d-reconstruction-from-multi-view-exp_amd/csrc"
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-result"
[ -f mvsvd.o ] || /opt/rocm/bin/hipcc $F -c mvsvd.hip -o mvsvd.o
mkdir -p ../../tools/ab ../../tools/ab2
build() {  # name dir nbuf waves
  /opt/rocm/bin/hipcc $F -DMVBA_HREC_TIMING -DMVBA_HREC_NBUF=$3 -DMVBA_SLOT_WAVES_PER_SIMD=$4 -c mvba.hip -o /tmp/mvba_$1.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 /tmp/mvba_$1.o mvsvd.o -o ../../tools/$2/libmvba_$1.so -ldl -lpthread
}
build hrec33 ab 3 3 &
build hrec32 ab2 3 2 &
build hrec42 ab2 4 2 &
wait
ls -la ../../tools/ab ../../tools/ab2
# knock-out builds of the PRODUCT slot kernel (what a step costs without its gathers' misses / without its gathers / without its arithmetic)
ko() {  # name define
  /opt/rocm/bin/hipcc $F -D$2 -c mvba.hip -o /tmp/mvba_$1.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 /tmp/mvba_$1.o mvsvd.o -o ../../tools/ab2/libmvba_$1.so -ldl -lpthread
}
ko ko_gather MVBA_KO_GATHER &
ko ko_dma MVBA_KO_DMA &
ko ko_valu MVBA_KO_VALU &
wait
ko ko_valu_gather "MVBA_KO_VALU -DMVBA_KO_GATHER" &
ko ko_valu_dma "MVBA_KO_VALU -DMVBA_KO_DMA" &
ko hrec33_ko_gather "MVBA_HREC_TIMING -DMVBA_KO_GATHER" &
wait
ko ko_idx MVBA_KO_IDX &
ko ko_idx_valu_dma "MVBA_KO_IDX -DMVBA_KO_VALU -DMVBA_KO_DMA" &
ko ko_idx_dma "MVBA_KO_IDX -DMVBA_KO_DMA" &
wait
# the same knock-outs for the UNIT form (config 4's shard: tools/ko_c4.sh): + the l side only, the point rows only
ko ko_lside MVBA_KO_LSIDE &
ko ko_prow MVBA_KO_PROW &
wait
