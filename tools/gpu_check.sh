#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
mkdir -p gpurun_out/check
echo "== pytest gpu"; timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -6
echo "== bench default"; timeout 900 python bench.py 2>gpurun_out/check/bench.err | grep "^{" > gpurun_out/check/bench.json; python -c "
import json; d=json.load(open('gpurun_out/check/bench.json')); print(d['value'], d['ms_per_step'], d['steps'], d['kernel_ms_per_step']); print(d['natural_lm_run']['iterations'], d['natural_lm_run']['ms_total']); print(d['cpu_baseline']['value'], d['cpu_baseline']['candidate'])"
echo "== stress"; timeout 900 python tools/stress_shapes.py 2>&1 | tail -4
