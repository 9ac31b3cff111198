cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5k
timeout 1200 python bench.py --no-traffic --no-cpu-baseline --no-vit-extra > gpurun_out/r5k/01_bench.json 2> gpurun_out/r5k/01.err; tail -c 400 gpurun_out/r5k/01.err
python -c "
import json; d=json.load(open('gpurun_out/r5k/01_bench.json')); print(d['value'], json.dumps(d['strong_share'], indent=1)[:3000])"
CR_DIST_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 1 --warmup 1 --pages 4 --total-pages 8 --new-tokens 8 --no-cpu-baseline --no-vit-extra --no-traffic > gpurun_out/r5k/02_bench_gloo2.json 2> gpurun_out/r5k/02.err; tail -c 300 gpurun_out/r5k/02.err
python -c "
import json; d=json.load(open('gpurun_out/r5k/02_bench_gloo2.json')); print(d['n_gpus'], d['value'], json.dumps(d['strong_scaling'])[:1500])"
