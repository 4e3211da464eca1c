cd ${GRAFT_REPO_ROOT:-/root/repo}
for bp in 256 384 512; do for r in 1 2; do echo "BP=$bp"; BALM_BUILD_BP=$bp python tools/bench_cluster_build.py | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['kernel_ms'], d['frac_of_8TBps'], d['max_rel_err_vs_host_push'])"; done; done
echo "40-pt runs"; for bp in 256 384 512; do BALM_BUILD_BP=$bp python tools/bench_cluster_build.py --pts 40 --features 3000 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['kernel_ms'], d['frac_of_8TBps'])"; done
