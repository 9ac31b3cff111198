cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5h
timeout 1500 python scripts/traffic_clock.py > gpurun_out/r5h/01_traffic_clock.txt 2>&1; tail -12 gpurun_out/r5h/01_traffic_clock.txt | cut -c1-400
cp profiles/round5/traffic_clock.json gpurun_out/r5h/ 2>/dev/null
