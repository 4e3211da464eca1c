#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
for g in 0 256 1024 2281; do
  echo "== BALM_FACTORS_GRID=$g"
  BALM_FACTORS_GRID=$g timeout 300 python tools/bench_realshape.py 2>&1 | grep "sparse solve default"
done
