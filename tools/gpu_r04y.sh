#!/bin/bash
# Last check of round 4 after bench.py's N > 1 lines gained `same_problem_on_one_gpu`: the default bench line, the N > 1 code path through two loopback shards on the one GPU, the CLI tests.
REPO=$(pwd); OUT=$REPO/gpurun_out/r04y; mkdir -p $OUT
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err < /dev/null; echo "bench rc=$?"; cut -c1-200 $OUT/bench.json; python -c "import json;d=json.load(open('$OUT/bench.json'));print(d['strong_scaling_reference']['ms_per_step'], json.load(open('gpurun_out/strong_scaling_n1_trace.json')).get('timing'))"
BALM_BENCH_LOOPBACK=1 timeout 900 python bench.py --gpus 2 --no-cpu --steps 20 > $OUT/bench_loopback2.json 2> $OUT/bench_loopback2.err < /dev/null; echo "loopback rc=$?"; python -c "import json;d=json.load(open('$OUT/bench_loopback2.json'));print(d['value'], d['scaling'], d['config']['parallelism'][:60], d.get('same_problem_on_one_gpu'), d.get('acceptance',{}).get('ok'))"
timeout 600 python -m pytest tests/test_bench_cli.py tests/test_gpu_multi.py -q -x > $OUT/pytest_cli.txt 2>&1 < /dev/null; echo "pytest rc=$?"; tail -2 $OUT/pytest_cli.txt
