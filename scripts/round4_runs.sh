cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4
timeout 900 python bench.py > gpurun_out/r4/07_bench_N1_default.json 2> gpurun_out/r4/07.err
timeout 1200 python bench.py --steps 20 --warmup 2 > gpurun_out/r4/06_bench_N1_steps20.json 2> gpurun_out/r4/06.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4/prof -o p64 -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-vit-extra --no-traffic > gpurun_out/r4/08_bench_under_rocprof.json 2> gpurun_out/r4/08.err
timeout 200 python scripts/config3.py > gpurun_out/r4/09_config3_phases.txt 2>&1
(timeout 200 python scripts/vit_config2.py 32 10; timeout 200 python scripts/vit_config2.py 255 5) > gpurun_out/r4/10_config2.txt 2>&1
timeout 300 python scripts/decode_bench.py 64 48 > gpurun_out/r4/11_decode_bench_64.txt 2>&1
timeout 200 python scripts/attn_bench.py > gpurun_out/r4/12_attn_bench.txt 2>&1
timeout 300 python scripts/decode_gemm_bench.py 1 2 4 8 16 > gpurun_out/r4/03_decode_gemm_bench.txt 2>&1
ls gpurun_out/r4 gpurun_out/r4/prof/* | head -30
