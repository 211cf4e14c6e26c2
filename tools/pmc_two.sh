#!/bin/bash
# two PMC passes (TCC incl. fabric requests, SQ wave/VALU/wait cycles) of the default bench workload
# usage: tools/pmc_two.sh <tag> [env assignments...]
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
tag=$1; shift
for kv in "$@"; do export "$kv"; done
run() { name=$1; shift; timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmc_${tag}_$name -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_${tag}_$name.log 2>&1 || { tail -5 gpurun_out/pmc_${tag}_$name.log; exit 1; }; }
run tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum && run sq1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_INSTS_VALU && run sq2 SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAVES
python tools/pmc_summary.py gpurun_out/pmc_${tag}_*/ | grep -A22 "^k_schur_[ps]"
