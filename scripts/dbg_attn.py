import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from callireader_amd import engine as E
def rb(t): return t.to(torch.bfloat16).float()
def ref(q,k,v):
    s = rb(q @ k.transpose(-1,-2)); p = rb(torch.softmax(s, -1)); return rb(p @ v)
for S in [64, 65, 128, 200, 1025]:
    g = torch.Generator().manual_seed(S)
    D=64
    q = torch.randn(1,S,D,generator=g).bfloat16().cuda(); k = torch.randn(1,S,D,generator=g).bfloat16().cuda(); v = torch.randn(1,S,D,generator=g).bfloat16().cuda()
    o = torch.zeros_like(q)
    E.op_attention(q,k,v,o,[S*D,D,D]*4,1,1,S,S,D)
    torch.cuda.synchronize()
    r = ref(q.float(),k.float(),v.float())
    of = o.float()
    print(S, 'nan', int(torch.isnan(of).sum()), 'inf', int(torch.isinf(of).sum()), 'maxdiff', float((of-r)[torch.isfinite(of)].abs().max()), 'bad rows', torch.nonzero(~torch.isfinite(of).all(-1))[:8,1].tolist())
