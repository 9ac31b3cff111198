#!/usr/bin/env python3
"""Run under torch.distributed.run: the tile-sharded + all-gather + round-robin page flow must give, for every page,
exactly the ids a single process produces.  Reduced depth (1 layer each, full width).  Backend: CR_DIST_BACKEND
(gloo lets both ranks share GPU 0 on a single-GPU box; nccl on a real multi-GPU node)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

from callireader_amd.config import ModelDims
from callireader_amd import synthetic
from callireader_amd.modeling_internvl_chat import InternVLChatModel
from callireader_amd.parallel import shard_range, all_gather_rows, owned_pages, plan_balanced

rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
backend = os.environ.get('CR_DIST_BACKEND', 'nccl')
if int(os.environ.get('WORLD_SIZE', '1')) > 2 * torch.cuda.device_count():
    os.environ.setdefault('CR_DECODE_LAYOUT', '0')       # many ranks sharing one GPU: no second copy of the LLM per rank
dev_idx = int(os.environ.get('LOCAL_RANK', '0')) % torch.cuda.device_count()
torch.cuda.set_device(dev_idx)
dist.init_process_group(backend)

IMG, REF = 8990, 8991
PT, CT, NEW = 2, 5, 6                       # page tiles, char tiles per page (deliberately not divisible by world), new tokens
n_pages = int(os.environ.get('CR_DIST_PAGES', '3'))       # 11 with 8 ranks: ragged page ownership (2,2,2,1,...) and ragged tile shards (55 and 22 tiles over 8)
dims = ModelDims.reduced(vit_layers=1, llm_layers=1, rs_depth=1, vocab=9000)
sd = synthetic.make_state_dict(dims, seed=0)
m = InternVLChatModel.from_state_dict(sd, dims, device=dev_idx, max_tokens=1024, max_pages=n_pages)
m.img_context_token_id, m.aligned_token_id = IMG, REF
page_px = synthetic.make_pixels(n_pages * PT, seed=5)
char_px = synthetic.make_pixels(n_pages * CT, seed=6)
ids = [torch.cat([torch.arange(50 + p, 60 + p), torch.full((PT * 256,), IMG), torch.full((CT * 3,), REF), torch.arange(7)]) for p in range(n_pages)]


def run(pages, vit_all, pseudo_all):
    embeds = [m.engine.embed_splice(ids[p], vit_all[p * PT:(p + 1) * PT], pseudo_all[p * CT:(p + 1) * CT], img_id=IMG, ref_id=REF) for p in pages]
    return m.generate_pages(embeds, max_new_tokens=NEW, eos_token_id=None)


# (a) generic path: page tiles sharded contiguously too, both kinds gathered
lo, hi = shard_range(n_pages * PT, world, rank)
clo, chi = shard_range(n_pages * CT, world, rank)
vit_all = all_gather_rows(m.extract_feature(page_px[lo:hi].cuda()), n_pages * PT)
pseudo, _ = m.align_tiles(char_px[clo:chi].cuda())
pseudo_all = all_gather_rows(pseudo.reshape(-1, 3, dims.llm_hidden), n_pages * CT)
mine = owned_pages(n_pages, world, rank)
outs = dict(zip(mine, run(mine, vit_all, pseudo_all)))
# (b) bench.py's flow: page tiles encoded by the page owner, only pseudo-tokens travel
own_px = torch.cat([page_px[p * PT:(p + 1) * PT] for p in mine])
vit_own = m.extract_feature(own_px.cuda())
vit_scatter = torch.zeros_like(vit_all)
for j, p in enumerate(mine):
    vit_scatter[p * PT:(p + 1) * PT] = vit_own[j * PT:(j + 1) * PT]
outs_b = dict(zip(mine, run(mine, vit_scatter, pseudo_all)))
assert outs_b == outs, (outs_b, outs)
# (c) the balanced strong-scaling plan (parallel.plan_balanced): fewer page owners, uneven character-tile shards (a rank may have none), one gather with
# explicit counts.  Twice: the cost model's choice, and ONE rank owning every page (the others only encode tiles).
plans_c = []
for owners in (None, 1):
    pl = plan_balanced(n_pages, world, PT, CT, int(ids[0].numel()), NEW, owners=owners)
    blo, bhi = pl['char_bounds'][rank]
    if bhi > blo:
        pc, _ = m.align_tiles(char_px[blo:bhi].cuda())
    else:
        pc = torch.empty((0, dims.llm_hidden), dtype=torch.bfloat16, device='cuda')
    pseudo_c = all_gather_rows(pc.reshape(-1, 3, dims.llm_hidden), n_pages * CT, counts=pl['char_counts'])
    assert torch.equal(pseudo_c, pseudo_all)
    mine_c = pl['pages'][rank]
    outs_c = dict(zip(mine_c, run(mine_c, vit_all, pseudo_c))) if mine_c else {}
    got_c = [None] * world
    dist.all_gather_object(got_c, outs_c)
    merged_c = {}
    for g in got_c:
        merged_c.update(g)
    plans_c.append((pl, merged_c))
gathered = [None] * world
dist.all_gather_object(gathered, outs)
if rank == 0:
    merged = {}
    for g in gathered:
        merged.update(g)
    v1 = m.extract_feature(page_px.cuda())
    p1, _ = m.align_tiles(char_px.cuda())
    single = run(list(range(n_pages)), v1, p1.reshape(-1, 3, dims.llm_hidden))
    ok = all(merged[p] == single[p] for p in range(n_pages)) and torch.equal(v1, vit_all)
    ok = ok and all(sorted(mc) == list(range(n_pages)) and all(mc[p] == single[p] for p in range(n_pages)) for _, mc in plans_c)
    ok = ok and plans_c[1][0]['k'] == 1
    print('DIST_CHECK', 'OK' if ok else 'MISMATCH', merged, single, flush=True)
    out = os.environ.get('CR_DIST_JSON')
    if out:
        import json
        from callireader_amd.parallel import shard_counts
        os.makedirs(os.path.dirname(out) or '.', exist_ok=True)
        json.dump({'what': 'scripts/dist_check.py: tile shards + all-gather + round-robin page ownership against one process, ids per page',
                   'world_size': world, 'backend': backend, 'visible_gpus': torch.cuda.device_count(), 'pages': n_pages,
                   'pages_per_rank': [len(owned_pages(n_pages, world, r)) for r in range(world)],
                   'char_tile_shards': shard_counts(n_pages * CT, world), 'page_tile_shards': shard_counts(n_pages * PT, world),
                   'balanced_plans': [{'page_owners': pl['k'], 'pages_per_rank': [len(x) for x in pl['pages']], 'char_tiles_per_rank': pl['char_counts']} for pl, _ in plans_c],
                   'ids_equal_single_process': bool(ok), 'ids': {str(k): v for k, v in sorted(merged.items())}}, open(out, 'w'), indent=1)
dist.barrier()
dist.destroy_process_group()
