#!/usr/bin/env python3
"""Debug aid: which stage makes page 9 of scripts/dist_check.py (11 pages) differ between one process and 8 ranks?  One process, one GPU:
the tile path in shards of 7 against all 55 tiles at once, and the prefill of page 9 alone / with page 1 / with all 11 pages."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from callireader_amd.config import ModelDims
from callireader_amd import synthetic
from callireader_amd.modeling_internvl_chat import InternVLChatModel
IMG, REF = 8990, 8991
PT, CT, NEW = 2, 5, 6
n_pages = 11
dims = ModelDims.reduced(vit_layers=1, llm_layers=1, rs_depth=1, vocab=9000)
sd = synthetic.make_state_dict(dims, seed=0)
m = InternVLChatModel.from_state_dict(sd, dims, device=0, max_tokens=1024, max_pages=n_pages)
m.img_context_token_id, m.aligned_token_id = IMG, REF
page_px = synthetic.make_pixels(n_pages * PT, seed=5).cuda()
char_px = synthetic.make_pixels(n_pages * CT, seed=6).cuda()
ids = [torch.cat([torch.arange(50 + p, 60 + p), torch.full((PT * 256,), IMG), torch.full((CT * 3,), REF), torch.arange(7)]) for p in range(n_pages)]
eng = m.engine
feat_all = m.extract_feature(char_px)
feat_sh = torch.cat([m.extract_feature(char_px[i:i + 7]) for i in range(0, 55, 7)])
print('extract_feature shards == all:', torch.equal(feat_all, feat_sh))
rs_all = eng.resample(feat_all)
rs_sh = torch.cat([eng.resample(feat_all[i:i + 7]) for i in range(0, 55, 7)])
print('resampler shards == all:', torch.equal(rs_all, rs_sh), float((rs_all.float() - rs_sh.float()).abs().max()))
idx_all = eng.vq(rs_all)
idx_sh = torch.cat([eng.vq(rs_all[i:i + 7]) for i in range(0, 55, 7)])
print('vq shards == all:', torch.equal(idx_all, idx_sh), (idx_all != idx_sh).nonzero().flatten().tolist())
p_all, _ = m.align_tiles(char_px)
p_sh = torch.cat([m.align_tiles(char_px[i:i + 7])[0] for i in range(0, 55, 7)])
print('align_tiles shards == all:', torch.equal(p_all, p_sh))
v_all = m.extract_feature(page_px)
pa = p_all.reshape(-1, 3, dims.llm_hidden)
emb = [eng.embed_splice(ids[p], v_all[p * PT:(p + 1) * PT], pa[p * CT:(p + 1) * CT], img_id=IMG, ref_id=REF) for p in range(n_pages)]
kv = eng.kv_alloc(n_pages, 1024)
l_one = eng.prefill(kv, 0, emb[9], want_logits=True).clone()
kv.reset()
l_two = eng.prefill_batch(kv, [0, 1], [emb[1], emb[9]], want_logits=True)[1].clone()
kv.reset()
l_all = eng.prefill_batch(kv, list(range(n_pages)), emb, want_logits=True)[9].clone()
torch.cuda.synchronize()
print('prefill page 9: alone == with page 1:', torch.equal(l_one, l_two), '; alone == all 11:', torch.equal(l_one, l_all), float((l_one - l_all).abs().max()))
t = torch.topk(l_one, 3)
print('page 9 top-3 logits (alone):', t.values.tolist(), t.indices.tolist(), '| all-11:', torch.topk(l_all, 3).values.tolist(), torch.topk(l_all, 3).indices.tolist())
for P in (2, 3, 4, 5, 8, 11):
    kv.reset()
    lg = eng.prefill_batch(kv, list(range(P)), emb[:P], want_logits=True)
    kv.reset()
    single = torch.stack([eng.prefill(kv, 0, emb[i], want_logits=True).clone() for i in range(P) if kv.reset() is None])
    print(f'prefill_batch of {P} pages == singles:', torch.equal(lg, single), [i for i in range(P) if not torch.equal(lg[i], single[i])])
