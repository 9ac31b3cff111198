cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5c
timeout 900 python -m pytest tests/test_gpu_llm.py tests/test_gpu_peaked.py -x -q -m gpu > gpurun_out/r5c/01_pytest_llm.txt 2>&1; tail -5 gpurun_out/r5c/01_pytest_llm.txt
for st in 4 2; do CR_ATTN_SPLIT_TILES=$st ROWS=1,2,4,8,16,64 STEPS=32 timeout 600 python scripts/decode_rows.py child 2>&1 | grep -E "RESULT" | sed "s/^/split_tiles $st: /" >> gpurun_out/r5c/02_decode_rows_waves_divide_keys.txt; done
cat gpurun_out/r5c/02_decode_rows_waves_divide_keys.txt
ROWS=8 STEPS=32 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r5c/prof -o d8 -- python3 scripts/decode_rows.py child > gpurun_out/r5c/03_prof.txt 2>&1
head -12 gpurun_out/r5c/prof/d8_kernel_stats.csv | cut -c1-200
