cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4f
timeout 900 python bench.py > gpurun_out/r4f/21_bench_N1_default.json 2> gpurun_out/r4f/21.err
timeout 1200 python bench.py --steps 20 --warmup 2 > gpurun_out/r4f/20_bench_N1_steps20.json 2> gpurun_out/r4f/20.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4f/prof -o p64 -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-vit-extra --no-traffic > gpurun_out/r4f/22_bench_under_rocprof.json 2> gpurun_out/r4f/22.err
ls gpurun_out/r4f gpurun_out/r4f/prof/* | head -30
