cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for shape in "1000000 12 1.0" "1000000 20 1.0" "1000000 30 0.5" "2000000 40 0.2"; do
  for mode in slots pairs; do
    MVBA_SCHUR=$mode timeout -k 10 200 python tools/time_schur.py $shape 4 2>/dev/null | sed "s/^/$mode: /"
  done
done
