cd /tmp && export TMPDIR=/tmp
for md in 0 2; do
rm -rf /tmp/p8; BALM_I8_MODE=$md timeout 600 rocprofv3 --kernel-trace -d /tmp/p8 -- python $GRAFT_REPO_ROOT/tools/exp_int8_syrk.py > /tmp/p8.log 2>&1
echo "== mode $md"; grep "SYRK span" /tmp/p8.log
python $GRAFT_REPO_ROOT/tools/rocprof_kernels.py /tmp/p8 | sed -n '/# averages/,$p' | grep -i "i8\|hessian_syrk"
done
