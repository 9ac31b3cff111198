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
from callireader_amd.parallel import shard_range, all_gather_rows, owned_pages, plan_balanced, plan_even, sharded_generate

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
# (c) parallel.sharded_generate, the library form of flow (b), under three plans: the cost model's (fewer page owners, uneven character-tile shards: a rank
# may own no page or encode no tile), ONE rank owning every page, the even split.  Then RAGGED pages (1..CT character tiles, a page without tiles of its own).
sizes = ([PT] * n_pages, [CT] * n_pages, [int(i.numel()) for i in ids])
pages_pt = [page_px[p * PT:(p + 1) * PT] for p in range(n_pages)]
pages_ct = [char_px[p * CT:(p + 1) * CT] for p in range(n_pages)]
plans_c = []
for pl in (plan_balanced(n_pages, world, *sizes, NEW), plan_balanced(n_pages, world, *sizes, NEW, owners=1), plan_even(n_pages, world, *sizes, NEW)):
    plans_c.append((pl, sharded_generate(m, pages_pt, pages_ct, ids, img_id=IMG, ref_id=REF, max_new_tokens=NEW, eos_token_id=None, plan=pl)))
rag_ct = [1 + (3 * p) % CT for p in range(n_pages)]
rag_pt = [0 if p == 1 else PT for p in range(n_pages)]
rag_pages_pt = [page_px[p * PT:p * PT + rag_pt[p]] for p in range(n_pages)]
rag_pages_ct = [char_px[p * CT:p * CT + rag_ct[p]] for p in range(n_pages)]
rag_ids = [torch.cat([torch.arange(50 + p, 60 + p), torch.full((rag_pt[p] * 256,), IMG), torch.full((rag_ct[p] * 3,), REF), torch.arange(7)]) for p in range(n_pages)]
rag_out = sharded_generate(m, rag_pages_pt, rag_pages_ct, rag_ids, img_id=IMG, ref_id=REF, max_new_tokens=NEW, eos_token_id=None)
# (d) round 6: the plan made under stage costs MEASURED on the GPU at hand (parallel.measure_cost: every rank measures, rank 0's numbers are broadcast so that all
# ranks compute the same plan), and a batch of pages WITHOUT any character tile (nothing to gather: no zero-byte collective)
meas_out = sharded_generate(m, pages_pt, pages_ct, ids, img_id=IMG, ref_id=REF, max_new_tokens=NEW, eos_token_id=None, cost='measure')
cost_used = m._measured_cost
costs = [None] * world
dist.all_gather_object(costs, {k: cost_used[k] for k in ('tile_ms', 'char_tile_ms', 'chunk_ms', 'prefill_ms_per_token', 'decode_ms')})
nochar_ids = [torch.cat([torch.arange(50 + p, 60 + p), torch.full((PT * 256,), IMG), torch.arange(7)]) for p in range(n_pages)]
nochar_out = sharded_generate(m, pages_pt, [char_px[:0]] * n_pages, nochar_ids, img_id=IMG, ref_id=REF, max_new_tokens=NEW, eos_token_id=None)
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
    rag_embeds = []
    for p in range(n_pages):
        v = m.extract_feature(rag_pages_pt[p].cuda()) if rag_pt[p] else None
        ps, _ = m.align_tiles(rag_pages_ct[p].cuda())                      # page by page: a tile's rows do not depend on its batch
        rag_embeds.append(m.engine.embed_splice(rag_ids[p].cuda(), v, ps.reshape(-1, 3, dims.llm_hidden), img_id=IMG, ref_id=REF))
    rag_single = m.generate_pages(rag_embeds, max_new_tokens=NEW, eos_token_id=None)
    ok = ok and sorted(rag_out) == list(range(n_pages)) and all(rag_out[p] == rag_single[p] for p in range(n_pages))
    ok = ok and sorted(meas_out) == list(range(n_pages)) and all(meas_out[p] == single[p] for p in range(n_pages))
    ok = ok and all(c == costs[0] for c in costs) and cost_used.get('measured') is True                 # one table on every rank: rank 0's
    nochar_embeds = [m.engine.embed_splice(nochar_ids[p].cuda(), v1[p * PT:(p + 1) * PT], None, img_id=IMG, ref_id=REF) for p in range(n_pages)]
    nochar_single = m.generate_pages(nochar_embeds, max_new_tokens=NEW, eos_token_id=None)
    ok = ok and sorted(nochar_out) == list(range(n_pages)) and all(nochar_out[p] == nochar_single[p] for p in range(n_pages))
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
                   'sharded_generate_plans': [{'page_owners': pl['k'], 'pages_per_rank': [len(x) for x in pl['pages']], 'char_tiles_per_rank': pl['char_counts']} for pl, _ in plans_c],
                   'ragged_pages': {'char_tiles_per_page': rag_ct, 'page_tiles_per_page': rag_pt},
                   'ids_equal_single_process': bool(ok), 'ids': {str(k): v for k, v in sorted(merged.items())}}, open(out, 'w'), indent=1)
dist.barrier()
dist.destroy_process_group()
