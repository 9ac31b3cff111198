cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5e
timeout 600 python -m pytest tests/test_gpu_fp8_mfma.py -x -q -m gpu -k outlier -s > gpurun_out/r5e/01_pytest_outlier.txt 2>&1; tail -8 gpurun_out/r5e/01_pytest_outlier.txt
timeout 1500 python scripts/fp8_schemes.py --no-emulation --out gpurun_out/r5e/fp8_schemes_hip.json > gpurun_out/r5e/02_fp8_schemes.txt 2>&1; tail -8 gpurun_out/r5e/02_fp8_schemes.txt
