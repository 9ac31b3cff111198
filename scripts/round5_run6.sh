cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5f
timeout 900 python -m pytest tests/test_gpu_llm.py tests/test_gpu_peaked.py tests/test_gpu_chat.py tests/test_gpu_pipeline.py -x -q -m gpu > gpurun_out/r5f/01_pytest.txt 2>&1; tail -5 gpurun_out/r5f/01_pytest.txt
for v in 1 0; do CR_DECODE_ATTN=$v ROWS=1,2,4,8,16,64 STEPS=32 timeout 600 python scripts/decode_rows.py child 2>&1 | grep -E "RESULT" | sed "s/^/CR_DECODE_ATTN=$v: /" >> gpurun_out/r5f/02_decode_rows_streaming_attention.txt; done
cat gpurun_out/r5f/02_decode_rows_streaming_attention.txt
ROWS=8 STEPS=32 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r5f/prof -o d8 -- python3 scripts/decode_rows.py child > gpurun_out/r5f/03_prof.txt 2>&1
grep -E "decode_attn|flash_attn|combine" gpurun_out/r5f/prof/d8_kernel_stats.csv | cut -c1-160
