cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5q
timeout 1200 python bench.py --no-traffic > gpurun_out/r5q/36_bench_N1_balanced_share.json 2> gpurun_out/r5q/36.err; tail -c 600 gpurun_out/r5q/36.err
timeout 600 python -m pytest tests/test_gpu_parallel.py -x -q > gpurun_out/r5q/37_pytest_gpu_parallel.txt 2>&1; tail -3 gpurun_out/r5q/37_pytest_gpu_parallel.txt
