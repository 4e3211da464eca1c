#!/bin/bash
# BALM_SYRK=int8 on the GPU box, into gpurun_out/$TAG/int8/: the full-size Hessian both ways against the reference's values
# (tools/exp_int8_syrk.py), the kernel tables of bench.py's timed loop with and without the switch (rocprofv3 --kernel-trace), the HBM bytes of
# the INT8 product's kernels (separate --pmc passes), and the whole GPU suite with the switch exported.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=${TAG:-i8}; OUT=$REPO/gpurun_out/$TAG/int8; mkdir -p $OUT; export TMPDIR=/tmp PYTHONPATH=$REPO
cd /tmp
(timeout 300 python $REPO/tools/exp_int8_syrk.py --small; timeout 300 python $REPO/tools/exp_int8_syrk.py) 2>&1 | grep -v amdgpu.ids | tee $OUT/accuracy_and_span.txt
for mode in int8 fp64; do
  rm -rf /tmp/p8
  if [ $mode = int8 ]; then export BALM_SYRK=int8; else unset BALM_SYRK; fi
  timeout 600 rocprofv3 --kernel-trace -d /tmp/p8 -- python $REPO/bench.py --no-cpu --steps 100 > $OUT/bench_$mode.json 2> /tmp/p8.log
  python $REPO/tools/rocprof_kernels.py /tmp/p8 | sed -n '/# averages/,$p' > $OUT/kernels_$mode.txt
  echo "== $mode"; python -c "import json; o=json.load(open('$OUT/bench_$mode.json')); print(o['value'], o['ms_per_step'], o['kernel_ms_per_step'])"; head -12 $OUT/kernels_$mode.txt
done
unset BALM_SYRK
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmc8/$C -o p -- python $REPO/tools/exp_int8_syrk.py > /dev/null 2>&1
done
python $REPO/tools/pmc_summary.py /tmp/pmc8 > $OUT/pmc_summary.csv 2>&1; grep -i "i8\|syrk\|kernel" $OUT/pmc_summary.csv | cut -c1-220 | head -8
rm -rf /tmp/pmc8 /tmp/p8
cd $REPO
BALM_SYRK=int8 timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest_gpu_under_int8.txt 2>&1 < /dev/null; echo "pytest under BALM_SYRK=int8 rc=$?" | tee -a $OUT/pytest_gpu_under_int8.txt; tail -3 $OUT/pytest_gpu_under_int8.txt
