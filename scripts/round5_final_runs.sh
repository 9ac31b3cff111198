cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5z
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r5z/20_final_pytest_gpu.txt 2>&1; tail -4 gpurun_out/r5z/20_final_pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r5z/20_final_smoke.txt 2>&1; tail -3 gpurun_out/r5z/20_final_smoke.txt
timeout 1200 python bench.py > gpurun_out/r5z/21_bench_N1_default.json 2> gpurun_out/r5z/21.err; tail -c 300 gpurun_out/r5z/21.err
timeout 1500 python bench.py --steps 20 --warmup 2 --no-traffic > gpurun_out/r5z/22_bench_N1_steps20.json 2> gpurun_out/r5z/22.err
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r5z/prof -o p64 -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-vit-extra --no-traffic --no-strong-share > gpurun_out/r5z/23_bench_under_rocprof.json 2> gpurun_out/r5z/23.err
ROWS=1,2,4,8,16,32,64 STEPS=48 timeout 600 python scripts/decode_rows.py child > gpurun_out/r5z/24_final_decode_rows.txt 2>&1
timeout 600 python scripts/config3.py > gpurun_out/r5z/25_config3_phases.txt 2>&1; tail -2 gpurun_out/r5z/25_config3_phases.txt
ls gpurun_out/r5z gpurun_out/r5z/prof | head -30
