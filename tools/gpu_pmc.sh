#!/bin/bash
# PMC passes (separate runs; never combined with sys/hip traces): HBM read / write bytes per kernel
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp
for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" ; do
  tag=$(echo $C | tr ' ' '_' | cut -c1-24)
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $REPO/gpurun_out/pmc/$tag -o p -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu > $REPO/gpurun_out/pmc/$tag.log 2>&1
  echo "== $C rc=$?"
done
cd $REPO
find gpurun_out/pmc -name "*.csv" | head -20
f=$(find gpurun_out/pmc/FETCH_SIZE -name "*counter_collection.csv" | head -1); echo $f; head -3 $f
