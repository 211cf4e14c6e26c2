#!/bin/bash
# product-library variants of the slot kernel's pacing traffic (scope of the arrivals / slow polls, cache-policy bits of the poll DMA)
set -e
cd "$(dirname "$0")/../3d-reconstruction-from-multi-view-exp_amd/csrc"
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-result"
mkdir -p ../../tools/ab3
pv() {  # name flags...
  n=$1; shift
  /opt/rocm/bin/hipcc $F "$@" -c mvba.hip -o /tmp/mvba_$n.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 /tmp/mvba_$n.o mvsvd.o -o ../../tools/ab3/libmvba_$n.so -ldl -lpthread
}
pv pace_none_wg '-DMVBA_POLL_BITS=""' &
pv pace_sc0_agent -DMVBA_PACE_SCOPE=__HIP_MEMORY_SCOPE_AGENT &
pv pace_nt_wg '-DMVBA_POLL_BITS="sc0 nt"' &
pv pace_sc1_agent '-DMVBA_POLL_BITS="sc1"' -DMVBA_PACE_SCOPE=__HIP_MEMORY_SCOPE_AGENT &
wait
ls -la ../../tools/ab3
