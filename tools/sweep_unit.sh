cd $GRAFT_REPO_ROOT
for v in ${UNITS:-512 640 768 896 1024 1280}; do
  MVBA_PAIR_UNIT=$v timeout -k 10 200 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('unit=$v', round(d['value'],1), 'schur', round(d['kernel_ms_per_step']['schur'],3))"
done
