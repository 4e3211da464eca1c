#!/bin/bash
# FIRST GPU call of round 4 (written at the end of round 3, after its GPU minutes were gone; every step has its own timeout and no
# step reads stdin -- round 3's last call hung on `head <empty file name>` for eleven minutes):
#   1. the full GPU suite, smoke() and the default bench on the tree round 3 left (the helpers' global_load_lds staging and the
#      macro-tile helpers were verified by tests/test_gpu_solve.py + tools/bench_solve.py only)
#   2. the solve by window size, default and with BALM_BUILD_A=lower / rows / rows+lower (never run), and the solve tests with the last
#   2b. k_ldl_backsolve2 (BALM_BACKSOLVE=fused), k_ldl_finish_u (BALM_FINISH=fast), the tile-major matrix with identity rows (BALM_TILED=ident), never run; then all opt-ins together: suite + bench
#   3. per-kernel times of one solve at n = 3000 and n = 1200 (read from rocprofv3's database: tools/rocprof_kernels.py)
REPO=$(pwd); OUT=$REPO/gpurun_out/r04a; mkdir -p $OUT
timeout 1500 python -m pytest tests -q -m gpu -x > $OUT/pytest_gpu.txt 2>&1 < /dev/null; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.txt 2>&1 < /dev/null; tail -1 $OUT/smoke.txt
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err < /dev/null; tail -c 1500 $OUT/bench.json
timeout 600 python tools/bench_solve.py 100 177 200 256 300 350 400 500 600 800 > $OUT/solve_default.txt 2>&1 < /dev/null
cut -c1-30,100-250 $OUT/solve_default.txt
for mode in lower rows rows+lower; do      # never run: k_build_A without the upper tiles / with the row of H staged in LDS (+ k_rank_diag_u)
  BALM_BUILD_A=$mode timeout 600 python tools/bench_solve.py 100 200 256 300 400 500 600 800 > $OUT/solve_build_a_$mode.txt 2>&1 < /dev/null
  sed "s/^/$mode /" $OUT/solve_build_a_$mode.txt | cut -c1-42,112-262
done
BALM_BUILD_A=rows+lower timeout 900 python -m pytest tests/test_gpu_solve.py -q -m gpu -x > $OUT/pytest_solve_rows_lower.txt 2>&1 < /dev/null; tail -2 $OUT/pytest_solve_rows_lower.txt
# never run: k_ldl_backsolve2 (the last hop as ONE matrix-vector product)
BALM_BACKSOLVE=fused timeout 600 python tools/bench_solve.py 256 300 400 500 600 800 > $OUT/solve_backsolve_fused.txt 2>&1 < /dev/null
sed "s/^/bs-fused /" $OUT/solve_backsolve_fused.txt | cut -c1-39,109-259
BALM_BACKSOLVE=fused timeout 900 python -m pytest tests/test_gpu_solve.py -q -m gpu -x > $OUT/pytest_solve_bs_fused.txt 2>&1 < /dev/null; tail -2 $OUT/pytest_solve_bs_fused.txt
# never run: k_ldl_finish_u (one workgroup, every load of a stage in flight, shuffle reduction)
BALM_FINISH=fast timeout 600 python tools/bench_solve.py 100 200 300 500 > $OUT/solve_finish_fast.txt 2>&1 < /dev/null
sed "s/^/finish-fast /" $OUT/solve_finish_fast.txt | cut -c1-42,112-262
# never run: the tile-major layout WITH identity rows for k_ldl_chain at 5..30 panels (the bench's n = 1200)
BALM_TILED=ident timeout 600 python tools/bench_solve.py 40 100 177 200 240 > $OUT/solve_tiled_ident.txt 2>&1 < /dev/null
sed "s/^/tiled-ident /" $OUT/solve_tiled_ident.txt | cut -c1-42,112-262
BALM_TILED=ident timeout 900 python -m pytest tests/test_gpu_solve.py -q -m gpu -x > $OUT/pytest_solve_tiled_ident.txt 2>&1 < /dev/null; tail -2 $OUT/pytest_solve_tiled_ident.txt
BALM_TILED=ident BALM_SOLVE_TRACE=1 timeout 300 python tools/chain_check.py 200 > $OUT/chain_trace_n1200_tiled_ident.txt 2>&1 < /dev/null; sed -n 4,12p $OUT/chain_trace_n1200_tiled_ident.txt | cut -c1-150
BALM_SOLVE_TRACE=1 timeout 300 python tools/chain_check.py 200 > $OUT/chain_trace_n1200.txt 2>&1 < /dev/null; sed -n 4,12p $OUT/chain_trace_n1200.txt | cut -c1-150
# all five together: the solve by window, the tests that drive whole LM runs, the default bench
export BALM_BUILD_A=rows+lower BALM_BACKSOLVE=fused BALM_FINISH=fast BALM_TILED=ident
timeout 600 python tools/bench_solve.py 100 177 200 256 300 400 500 600 800 > $OUT/solve_all_optins.txt 2>&1 < /dev/null
sed "s/^/all /" $OUT/solve_all_optins.txt | cut -c1-34,104-254
timeout 1500 python -m pytest tests -q -m gpu -x > $OUT/pytest_gpu_all_optins.txt 2>&1 < /dev/null; echo "pytest (opt-ins) rc=$?"; tail -3 $OUT/pytest_gpu_all_optins.txt
timeout 600 python bench.py --no-cpu > $OUT/bench_all_optins.json 2> $OUT/bench_all_optins.err < /dev/null; tail -c 600 $OUT/bench_all_optins.json
unset BALM_BUILD_A BALM_BACKSOLVE BALM_FINISH BALM_TILED
cd /tmp; export TMPDIR=/tmp
for W in 500 200; do
  rm -rf $OUT/prof_solve$W
  timeout 300 rocprofv3 --kernel-trace -d $OUT/prof_solve$W -o s -- python $REPO/tools/bench_solve_one.py $W > /dev/null 2>&1 < /dev/null
  timeout 60 python $REPO/tools/rocprof_kernels.py $OUT/prof_solve$W k_ > $OUT/solve_kernels_W$W.txt 2>&1 < /dev/null
  tail -8 $OUT/solve_kernels_W$W.txt
  find $OUT/prof_solve$W -name "*.db" -size +8M -delete
done
