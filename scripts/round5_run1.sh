cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5a
timeout 900 python bench.py --no-traffic > gpurun_out/r5a/01_bench_N1_default.json 2> gpurun_out/r5a/01.err
tail -c 600 gpurun_out/r5a/01.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r5a/prof -o share -- python3 bench.py --pages 8 --steps 2 --warmup 1 --no-pipeline --no-cpu-baseline --no-vit-extra --no-traffic --no-strong-share > gpurun_out/r5a/02_share_under_rocprof.json 2> gpurun_out/r5a/02.err
tail -c 300 gpurun_out/r5a/02.err
ROWS=1,8 STEPS=48 timeout 600 python scripts/decode_rows.py child > gpurun_out/r5a/03_decode_rows.txt 2>&1
ls gpurun_out/r5a gpurun_out/r5a/prof/* | head -30
