timeout 300 python -m pytest tests/test_gpu_ops.py -q --timeout 250 2>&1 | grep -E "FAILED|passed|failed" | head -30
for lib in "" /root/repo/ab/libc1.so "" /root/repo/ab/libc1.so; do echo "lib=$lib"; CR_HIP_LIB=$lib timeout 200 python scripts/gemm_bench.py 64575,4096,1024 64575,1024,1024 64575,1024,4096 64575,3072,1024 25312,6144,4096 25312,4096,14336 8192,8192,8192 2>&1 | tail -7; done
