import sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
from callireader_amd import engine as E
g = torch.Generator(device='cuda').manual_seed(0)
def run(name, fn, n=20):
    fn(); torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(n): fn()
    ev[1].record(); torch.cuda.synchronize()
    print(f'{name:40s} {ev[0].elapsed_time(ev[1]) / n:8.3f} ms', flush=True)
M = 32800
for (N, K, epi) in [(3072, 1024, 0), (1024, 1024, 2), (4096, 1024, 1), (1024, 4096, 2)]:
    A = (torch.rand(M, K, device='cuda', generator=g) * 2 - 1).bfloat16()
    W = ((torch.rand(N, K, device='cuda', generator=g) * 2 - 1) * 0.05).bfloat16()
    bias = (torch.rand(N, device='cuda', generator=g) * 0.1).bfloat16()
    scale = (torch.rand(N, device='cuda', generator=g) * 0.1).bfloat16()
    res = (torch.rand(M, N, device='cuda', generator=g)).bfloat16()
    kw = dict(bias=bias) if epi < 2 else dict(bias=bias, scale=scale, res=res)
    run(f'N={N} K={K} epi={epi} dispatcher', lambda: E.op_gemm(epi, A, W, **kw))
    run(f'N={N} K={K} epi={epi} pinned 256', lambda: E.op_gemm(epi, A, W, kernel=2, **kw))
    run(f'N={N} K={K} epi={epi} 32768 rows pinned', lambda: E.op_gemm(epi, A[:32768], W, kernel=2, **{k: (v[:32768] if k == 'res' else v) for k, v in kw.items()}))
    run(f'N={N} K={K} epi={epi} last 32 rows', lambda: E.op_gemm(epi, A[32768:], W, **{k: (v[32768:] if k == 'res' else v) for k, v in kw.items()}))
