cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5d
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r5d/01_pytest_gpu.txt 2>&1; tail -5 gpurun_out/r5d/01_pytest_gpu.txt
timeout 300 python scripts/decode128_probe.py > gpurun_out/r5d/02_decode128_probe.txt 2>&1; cat gpurun_out/r5d/02_decode128_probe.txt
( echo "## product build"; timeout 300 python scripts/attn_bench.py 2>&1 | grep ViT
  echo "## -DCR_KO_VIT_SOFTMAX (no rounding / FMA / exp / pack: matrix pipe + fragment reads + fills alone)"; CR_HIP_LIB=ab/libko_vit_softmax.so timeout 300 python scripts/attn_bench.py 2>&1 | grep "ViT"
  echo "## -DCR_KO_VIT_MFMA (no MFMAs, no fragment reads: the softmax's vector work + fills alone)"; CR_HIP_LIB=ab/libko_vit_mfma.so timeout 300 python scripts/attn_bench.py 2>&1 | grep "ViT" ) > gpurun_out/r5d/03_vit_attention_knockouts.txt 2>&1
cat gpurun_out/r5d/03_vit_attention_knockouts.txt
timeout 900 python bench.py --no-traffic --no-cpu-baseline > gpurun_out/r5d/04_bench.json 2> gpurun_out/r5d/04.err; tail -c 300 gpurun_out/r5d/04.err
