#!/bin/bash
# Round 4's final evidence, on the tree the round stops at: the whole GPU suite, smoke(), then tools/gpu_profile_round.sh r04z (default bench line with the CPU leg,
# rocprofv3 kernel table, the three PMC passes -> pmc_traffic.json, tools/ubench_f64).  Every rocprofv3 run sits under `timeout` (it crashes at exit on this stack).
REPO=$(pwd); OUT=$REPO/gpurun_out/r04z; mkdir -p $OUT
timeout 1500 python -m pytest tests -q -m gpu > $OUT/pytest_gpu.txt 2>&1 < /dev/null; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.txt; tail -3 $OUT/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1 < /dev/null; echo "smoke rc=$?"; tail -2 $OUT/smoke.txt
timeout 2400 bash tools/gpu_profile_round.sh r04z < /dev/null
