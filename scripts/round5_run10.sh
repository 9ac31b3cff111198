cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5j
timeout 900 python -m pytest tests/test_gpu_fp8.py tests/test_gpu_ops.py -x -q -m gpu > gpurun_out/r5j/01_pytest.txt 2>&1; tail -6 gpurun_out/r5j/01_pytest.txt
for P in 64 32; do
  echo "## $P pages: bf16 decode | e4m3-weight decode" >> gpurun_out/r5j/02_decode_bench_fp8_stream.txt
  timeout 600 python scripts/decode_bench.py $P 32 2>&1 | grep decode, >> gpurun_out/r5j/02_decode_bench_fp8_stream.txt
  FP8=1 timeout 600 python scripts/decode_bench.py $P 32 2>&1 | grep decode, >> gpurun_out/r5j/02_decode_bench_fp8_stream.txt
done
cat gpurun_out/r5j/02_decode_bench_fp8_stream.txt
