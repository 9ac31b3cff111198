cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5p
ROWS=8,9,10,11,12,13,14,16,20,22,24,32 STEPS=48 timeout 900 python scripts/decode_rows.py child > gpurun_out/r5p/35_decode_rows_9_to_32.txt 2>&1
tail -3 gpurun_out/r5p/35_decode_rows_9_to_32.txt
