# The GPU runs behind profiles/round5/35_* .. 49_* (the balanced strong-scaling plan), as they were given to gpurun one block at a time.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r5bal; mkdir -p $O
# 35: the decode step by rows between the fused path (<= 8) and 32 rows -- the table behind parallel.MI355X_COST
ROWS=8,9,10,11,12,13,14,16,20,22,24,32 STEPS=48 timeout 900 python scripts/decode_rows.py child > $O/35_decode_rows_9_to_32.txt 2>&1
# 37 / 38: sharded_generate under three plans + ragged pages against single-process ids: world 2 (the suite), 8 gloo ranks on one GPU
timeout 600 python -m pytest tests/test_gpu_parallel.py -x -q > $O/37_pytest_gpu_parallel.txt 2>&1
CR_DIST_BACKEND=gloo CR_DIST_PAGES=11 CR_DIST_JSON=$O/38_dist8_gloo_balanced.json timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29611 scripts/dist_check.py > $O/38_dist8_gloo_balanced.txt 2>&1
# 39 / 40: the N > 1 bench line's strong_scaling.balanced block over gloo (plumbing, ranks share the GPU)
CR_DIST_BACKEND=gloo timeout 1500 python bench.py --gpus 2 --steps 1 --warmup 1 --pages 32 --total-pages 32 --new-tokens 32 --no-cpu-baseline --no-vit-extra --no-traffic > $O/39_bench_gloo2_balanced.json 2> $O/39.err
CR_DIST_BACKEND=gloo timeout 1800 python bench.py --gpus 8 --steps 1 --warmup 1 --pages 4 --total-pages 16 --no-cpu-baseline --no-vit-extra --no-traffic > $O/40_bench_gloo8_balanced.json 2> $O/40.err
# 46: per-kernel view of the page owners' 13-row decode step
ROWS=13 STEPS=32 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o d13 -- python3 scripts/decode_rows.py child > $O/d13.txt 2>&1
# 49: the number of page owners fixed, against the cost model's prediction
for K in 3 4 5 6 7; do timeout 900 python bench.py --no-traffic --no-cpu-baseline --no-vit-extra --balanced-owners $K > $O/owners_$K.json 2> $O/owners_$K.err; done
# 47 / 48: suite, smoke and the driver's bench command on the last tree
timeout 1500 python -m pytest tests -x -q -m gpu > $O/47_final_pytest_gpu.txt 2>&1
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/47_final_smoke.txt 2>&1
timeout 1200 python bench.py > $O/48_bench_N1_default.json 2> $O/48.err
