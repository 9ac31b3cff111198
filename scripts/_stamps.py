import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from callireader_amd import engine as E
g = torch.Generator(device='cuda').manual_seed(0)
for (M, N, K) in [(64575, 4096, 1024), (64575, 1024, 4096)]:
    A = (torch.rand(M, K, device='cuda', generator=g) * 2 - 1).bfloat16()
    W = (torch.rand(N, K, device='cuda', generator=g) * 2 - 1).bfloat16()
    dbg = torch.zeros(256 * 2 * 32 * 4, dtype=torch.int64, device='cuda')
    for _ in range(3):
        E.op_gemm(0, A, W, kernel=2)
    torch.cuda.synchronize()
    E.op_gemm(0, A, W, scale=dbg, kernel=2)
    torch.cuda.synchronize()
    d = dbg.cpu().reshape(256, 2, 32, 4).double()
    for wg in (0, 1, 100, 255):
        for grp in (0, 1):
            x = d[wg, grp]
            n = int((x[:, 0] > 0).sum())
            t0 = x[0, 0]
            rows = []
            for t in range(min(n, 6)):
                loop = x[t, 1] - x[t, 0]; drain = x[t, 2] - x[t, 1]; epi = x[t, 3] - x[t, 2]
                gap = (x[t + 1, 0] - x[t, 3]) if t + 1 < n else 0
                rows.append(f'[{loop:.0f} {drain:.0f} {epi:.0f} {gap:.0f}]')
            print(M, N, K, 'wg', wg, 'grp', grp, 'tiles', n, 'total', x[n - 1, 3] - t0, ' '.join(rows))
    # averages over all
    x = d[:, :, :, :]
    valid = x[..., 0] > 0
    loop = (x[..., 1] - x[..., 0])[valid].mean(); drain = (x[..., 2] - x[..., 1])[valid].mean(); epi = (x[..., 3] - x[..., 2])[valid].mean()
    print('mean ticks (100 MHz => 10 ns each?): loop', loop.item(), 'drain', drain.item(), 'epi', epi.item())
