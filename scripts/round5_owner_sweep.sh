cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5k
for K in 3 4 5 6 7; do
  timeout 900 python bench.py --no-traffic --no-cpu-baseline --no-vit-extra --balanced-owners $K > gpurun_out/r5k/owners_$K.json 2> gpurun_out/r5k/owners_$K.err; echo "k=$K rc $?"
done
