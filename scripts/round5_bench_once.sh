cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5v
S0=$(date +%s); timeout 1200 python bench.py --no-traffic > gpurun_out/r5v/45_bench_N1_fp8_balanced.json 2> gpurun_out/r5v/45.err; echo "bench wall $(( $(date +%s) - S0 )) s" | tee -a gpurun_out/r5v/45.err; tail -c 400 gpurun_out/r5v/45.err
