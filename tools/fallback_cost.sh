#!/bin/bash
# what the co-residency fallbacks cost when they are taken (a shared GPU): the back-substitution as per-block launches / with
# device-wide barriers instead of point-to-point words, the slot kernel without pacing -- config 3 and D = 4493 (config 4's shard)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
line() {  # tag, env..., then bench args after --
  tag=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout -k 10 300 python bench.py --no-cpu-baseline --no-config4-shard-leg --svd-rows 0 --depth-rows 0 "$@" > gpurun_out/fb_$tag.json 2> gpurun_out/fb_$tag.err || { echo "$tag FAILED: ${envs[*]}"; tail -3 gpurun_out/fb_$tag.err; exit 1; }
  python -c "
import json; d=json.load(open('gpurun_out/fb_$tag.json')); k=d['kernel_ms_per_step']
print('$tag'.ljust(34), 'it/s', round(d['value'],1), 'ms/step', round(d['ms_per_step'],3), 'schur', round(k['schur'],3), 'solve', round(k['solve'],3))"
}
line c3_default -- --steps 20 --warmup 5
line c3_backsub_launches MVBA_CHOL=launches -- --steps 20 --warmup 5
line c3_backsub_barriers MVBA_CHOL=barriers -- --steps 20 --warmup 5
line c3_backsub_barrier_fallback MVBA_CHOL=barriers MVBA_CHOL_BARRIER_POLLS=0 -- --steps 20 --warmup 5
line c3_no_pacing MVBA_SLOT_SEG=0 -- --steps 20 --warmup 5
line c4shard_default -- --config4-shard --steps 6 --warmup 2
line c4shard_backsub_launches MVBA_CHOL=launches -- --config4-shard --steps 6 --warmup 2
