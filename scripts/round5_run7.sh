cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5g
timeout 900 python -m pytest tests/test_gpu_fp8.py -x -q -m gpu -s > gpurun_out/r5g/01_pytest_fp8.txt 2>&1; tail -6 gpurun_out/r5g/01_pytest_fp8.txt
for P in 64 16 8; do
  echo "## $P pages: bf16 decode | e4m3-weight decode (decode layout of the e4m3 copies) | e4m3-weight decode, CR_DECODE_LAYOUT=0" >> gpurun_out/r5g/02_decode_bench_fp8.txt
  timeout 600 python scripts/decode_bench.py $P 32 2>&1 | grep decode, >> gpurun_out/r5g/02_decode_bench_fp8.txt
  FP8=1 timeout 600 python scripts/decode_bench.py $P 32 2>&1 | grep decode, >> gpurun_out/r5g/02_decode_bench_fp8.txt
  FP8=1 CR_DECODE_LAYOUT=0 timeout 600 python scripts/decode_bench.py $P 32 2>&1 | grep decode, >> gpurun_out/r5g/02_decode_bench_fp8.txt
done
cat gpurun_out/r5g/02_decode_bench_fp8.txt
