#!/bin/bash
# The round's judged artefacts in one GPU call: kernel stats + PMC of config 3 and of config 4's shard (tools/final_profile.sh),
# the PMC JSONs put where bench.py looks for them (profiles/, guarded by the source hash), then the bench lines themselves.
# usage: tools/final_round.sh <tag>      -> gpurun_out/<tag>_*; copy what is to be judged into profiles/ afterwards
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
tag=$1
bash tools/final_profile.sh ${tag} pmc_config3 || exit 1
cp gpurun_out/${tag}_pmc_config3.json profiles/pmc_config3.json
bash tools/final_profile.sh ${tag}4 pmc_config4_shard --config4-shard || exit 1
cp gpurun_out/${tag}4_pmc_config4_shard.json profiles/pmc_config4_shard.json
timeout -k 10 500 python bench.py > gpurun_out/${tag}_bench_default.json 2> gpurun_out/${tag}_bench_default.err || { tail -5 gpurun_out/${tag}_bench_default.err; exit 1; }
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-config4-shard-leg > gpurun_out/${tag}_bench_config3.json 2> /dev/null || exit 1
timeout -k 10 300 python bench.py --config4-shard --steps 10 --warmup 3 --no-cpu-baseline --svd-rows 0 > gpurun_out/${tag}_bench_config4_shard.json 2> /dev/null || exit 1
python - <<PY
import json
for n in ("default", "config3", "config4_shard"):
    d = json.load(open("gpurun_out/${tag}_bench_%s.json" % n))
    print(n, round(d["value"], 2), d["unit"], "ms/step", round(d["ms_per_step"], 3), "roofline", d["roofline"]["frac"], d["roofline"].get("bound"), "traffic", d["roofline"].get("traffic"), {k: round(v, 3) for k, v in d["kernel_ms_per_step"].items()})
PY
