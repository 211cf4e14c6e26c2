#!/bin/bash
# kernel-trace stats + PMC (separate passes, never with a trace domain) of a bench workload; summaries -> gpurun_out/
# usage: tools/final_profile.sh <tag> [<pmc json name: pmc_config3 | pmc_config4_shard> [bench args...]]
#   tools/final_profile.sh r04_x                                   (config 3, the default bench workload)
#   tools/final_profile.sh r04_x pmc_config4_shard --config4-shard (config 4's per-GPU shard)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
tag=$1; name=${2:-pmc_config3}; shift; shift
BARGS="$@"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag} -- python bench.py $BARGS --steps 10 --warmup 2 --no-cpu-baseline --no-config4-shard-leg --svd-rows 0 > gpurun_out/prof_${tag}.json 2> gpurun_out/prof_${tag}.err || { tail -5 gpurun_out/prof_${tag}.err; exit 1; }
cp $(find gpurun_out/prof_${tag} -name "*kernel_stats.csv" | head -1) gpurun_out/${tag}_kernel_stats.csv
python tools/kstats.py gpurun_out/${tag}_kernel_stats.csv > gpurun_out/${tag}_kernel_stats.txt
run() { nm=$1; shift; timeout -k 10 400 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmc_${tag}_$nm -- python bench.py $BARGS --steps 3 --warmup 1 --no-cpu-baseline --no-config4-shard-leg --svd-rows 0 > gpurun_out/pmc_${tag}_$nm.log 2>&1 || { tail -5 gpurun_out/pmc_${tag}_$nm.log; exit 1; }; }
run fetch FETCH_SIZE && run write WRITE_SIZE && run tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum && run sq1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_INSTS_VALU && run sq2 SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAVES && run grbm GRBM_GUI_ACTIVE
python tools/pmc_summary.py gpurun_out/pmc_${tag}_*/ > gpurun_out/${tag}_pmc_summary.txt
python - > gpurun_out/${tag}_${name}.json <<PY
import json, subprocess, sys
d = json.load(open("gpurun_out/prof_${tag}.json"))
c = d["config"]
print(subprocess.run([sys.executable, "tools/pmc_to_json.py", str(c["observations_total"]), str(c["points_total"]), "${tag}",
                      "gpurun_out/pmc_${tag}_fetch", "gpurun_out/pmc_${tag}_write", "gpurun_out/pmc_${tag}_tcc"], capture_output=True, text=True, check=True).stdout)
PY
cat gpurun_out/${tag}_kernel_stats.txt | head -12
grep -A3 "k_resid_jac" gpurun_out/${tag}_pmc_summary.txt | head -8
