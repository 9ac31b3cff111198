cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5z4
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r5z4/50_final_pytest_gpu.txt 2>&1; tail -4 gpurun_out/r5z4/50_final_pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r5z4/50_final_smoke.txt 2>&1; tail -3 gpurun_out/r5z4/50_final_smoke.txt
