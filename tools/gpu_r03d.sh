#!/bin/bash
# after k_ldl_chain became the default: whole GPU suite, solve table, bench line
TAG=r03d
mkdir -p gpurun_out/$TAG
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/$TAG/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/$TAG/pytest_gpu.txt
tail -8 gpurun_out/$TAG/pytest_gpu.txt
echo "# tools/bench_solve.py on MI355X, device ms per balm_solve_damped (HIP events)" > gpurun_out/$TAG/solve_paths_by_window.txt
timeout 600 python tools/bench_solve.py >> gpurun_out/$TAG/solve_paths_by_window.txt 2>&1
cat gpurun_out/$TAG/solve_paths_by_window.txt
timeout 600 python bench.py --no-cpu 2>gpurun_out/$TAG/bench.err | grep "^{" > gpurun_out/$TAG/bench_nocpu.json
python - <<'PY'
import json
b=json.load(open('gpurun_out/r03d/bench_nocpu.json'))
print(b['value'], b['ms_per_step'], b['kernel_ms_per_step'])
PY
timeout 300 python tools/bench_realshape.py 2>&1 | tail -12 | tee gpurun_out/$TAG/realshape_step.txt
