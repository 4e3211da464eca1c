# per-kernel times of the bench's timed loop with BALM_SYRK=int8 against the default (rocprofv3 --kernel-trace)
cd /tmp && export TMPDIR=/tmp
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/i8
for mode in int8 fp64; do
  rm -rf /tmp/p8
  if [ $mode = int8 ]; then export BALM_SYRK=int8; else unset BALM_SYRK; fi
  timeout 600 rocprofv3 --kernel-trace -d /tmp/p8 -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --steps 100 > /tmp/p8.json 2> /tmp/p8.log
  echo "== $mode"; python -c "import json; o=json.load(open('/tmp/p8.json')); print(o['value'], o['ms_per_step'], o['kernel_ms_per_step'])"
  python $GRAFT_REPO_ROOT/tools/rocprof_kernels.py /tmp/p8 | sed -n '/# averages/,$p' > $GRAFT_REPO_ROOT/gpurun_out/i8/kernels_$mode.txt
  head -24 $GRAFT_REPO_ROOT/gpurun_out/i8/kernels_$mode.txt
done
