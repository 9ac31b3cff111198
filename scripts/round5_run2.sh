cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5b
timeout 900 python -m pytest tests/test_gpu_llm.py tests/test_gpu_ops.py -x -q -m gpu > gpurun_out/r5b/01_pytest_llm_ops.txt 2>&1; tail -5 gpurun_out/r5b/01_pytest_llm_ops.txt
timeout 600 python scripts/decode_variants.py 1 4 8 > gpurun_out/r5b/02_decode_variants.txt 2>&1; tail -20 gpurun_out/r5b/02_decode_variants.txt
for st in 4 2 1; do CR_ATTN_SPLIT_TILES=$st ROWS=1,8,64 STEPS=32 timeout 600 python scripts/decode_rows.py child 2>&1 | grep -E "rows|RESULT" | sed "s/^/split_tiles $st: /" >> gpurun_out/r5b/03_decode_rows_split_tiles.txt; done
cat gpurun_out/r5b/03_decode_rows_split_tiles.txt
