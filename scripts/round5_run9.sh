cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5i
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "attention" > gpurun_out/r5i/01_pytest_attention.txt 2>&1; tail -5 gpurun_out/r5i/01_pytest_attention.txt
timeout 300 python scripts/attn_bench.py > gpurun_out/r5i/02_attn_bench.txt 2>&1; cat gpurun_out/r5i/02_attn_bench.txt
