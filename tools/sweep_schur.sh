cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests -m gpu -x -q > gpurun_out/test_r1_c.log 2>&1 || { tail -30 gpurun_out/test_r1_c.log; exit 1; }
tail -2 gpurun_out/test_r1_c.log
for t in 1024 768 512; do for c in 31 62 124; do
MVBA_SCHUR_THREADS=$t MVBA_SCHUR_CHUNKS=$c timeout -k 10 120 python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('threads $t chunks $c', round(d['value'],1),'it/s', {k:round(v,3) for k,v in d['kernel_ms_per_step'].items()})" || exit 1
done; done
