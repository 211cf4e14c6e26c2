#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
tag=$1
run() { name=$1; shift; timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmc_${tag}_$name -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_${tag}_$name.log 2>&1 || { tail -5 gpurun_out/pmc_${tag}_$name.log; exit 1; }; }
run sq1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES && run sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU && run grbm GRBM_GUI_ACTIVE GRBM_COUNT
python tools/pmc_summary.py gpurun_out/pmc_${tag}_*/ | awk '/^k_schur/{p=1} /^k_[a-rt-z]/{if($0!~/schur/)p=0} p'
