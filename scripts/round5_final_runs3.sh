cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5z3
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r5z3/47_final_pytest_gpu.txt 2>&1; tail -4 gpurun_out/r5z3/47_final_pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r5z3/47_final_smoke.txt 2>&1; tail -3 gpurun_out/r5z3/47_final_smoke.txt
S0=$(date +%s); timeout 1200 python bench.py > gpurun_out/r5z3/48_bench_N1_default.json 2> gpurun_out/r5z3/48.err; echo "bench wall $(( $(date +%s) - S0 )) s" | tee -a gpurun_out/r5z3/48.err
