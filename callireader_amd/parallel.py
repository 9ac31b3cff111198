"""Multi-GPU sharding of the tile path: one process per GPU, `torch.distributed` (backend 'nccl' = RCCL on ROCm,
'gloo' in CPU tests).

The reference has no inference parallelism; this is the new exchange step north_star asks for:
tiles of a batch of pages are independent through ViT -> projector -> resampler -> VQ -> de-norm.  The character
tiles (90 % of the visual work, and what varies most from page to page) form one flat list that is split contiguously
and evenly over ranks (reading order preserved); each rank runs the visual stage on its shard and ONE all-gather hands
every rank all pseudo-token embeddings (24 576 B per tile).  Pages are owned round-robin (page p -> rank p % world)
for the LLM; a page's own tiles (2.1 MB of embeddings each, needed by nobody else) are encoded by its owner.
`all_gather_rows` is generic: it also carries the 2.1 MB page-tile rows when a caller shards those too.

STRONG scaling (a fixed number of pages per step over more GPUs) has one term that does not divide: the batched decode streams the
LLM's weights once per step whatever the number of rows (14.7 GB; 3.9 ms at 8 rows, 4.5 ms at 16, 8.9 ms at 64 on MI355X), so eight
ranks decoding 8 pages each pay for it eight times.  `plan_balanced` therefore lets FEWER ranks own pages (fatter decode batches)
and hands the other ranks more of the character tiles -- the tile list is still one contiguous partition and there is still ONE
all-gather, only with uneven shards (`counts`): the split that a cost model of the three stages says finishes first.
"""
import torch
import torch.distributed as dist


def shard_range(total, world, rank):
    """Contiguous, even split: the first `total % world` ranks get one extra element."""
    q, r = divmod(total, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def shard_counts(total, world):
    return [shard_range(total, world, r)[1] - shard_range(total, world, r)[0] for r in range(world)]


def _counts(total, world, counts):
    if counts is None:
        return shard_counts(total, world)
    counts = [int(c) for c in counts]
    if len(counts) != world or sum(counts) != total or min(counts) < 0:
        raise ValueError(f'shard counts {counts} are not a partition of {total} rows over {world} ranks')
    return counts


def all_gather_rows(local, total, group=None, counts=None):
    """local: (n_local, ...) rows of this rank's contiguous shard of a `total`-row tensor -> (total, ...) on every rank.
    Ragged shards are padded to the largest one so a single dist.all_gather_into_tensor moves everything.
    counts: rows per rank when the partition is not the even one (plan_balanced; a rank may have none)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        assert local.shape[0] == total
        return local
    counts = _counts(total, world, counts)
    assert local.shape[0] == counts[dist.get_rank(group)], (local.shape, counts)
    if total == 0:                                        # pages without a character tile: nothing to exchange, and no zero-byte collective
        return local[:0]
    mx = max(counts)
    tail = local.shape[1:]
    send = local
    if local.shape[0] != mx:
        send = torch.zeros((mx,) + tuple(tail), dtype=local.dtype, device=local.device)
        send[:local.shape[0]] = local
    if send.is_cuda and dist.get_backend(group) == 'gloo':
        # functional fallback for single-GPU test boxes (ranks sharing one device): gloo moves host buffers
        host = torch.empty((world * mx,) + tuple(tail), dtype=local.dtype)
        dist.all_gather_into_tensor(host, send.contiguous().cpu(), group=group)
        recv = host.to(local.device)
    else:
        recv = torch.empty((world * mx,) + tuple(tail), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(recv, send.contiguous(), group=group)
    if all(c == mx for c in counts):
        return recv
    return torch.cat([recv[r * mx:r * mx + c] for r, c in enumerate(counts)], dim=0)


def all_gather_rows_async(local, total, group=None, counts=None):
    """Start the gather and return `finish() -> (total, ...) tensor`: on RCCL the collective runs on the communicator's
    own stream, so kernels enqueued before finish() (the page tiles' ViT, in bench.py) overlap the transfer over xGMI.
    World size 1 and the gloo host fallback complete immediately."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1 or (local.is_cuda and dist.get_backend(group) == 'gloo'):
        out = all_gather_rows(local, total, group, counts)
        return lambda: out
    counts = _counts(total, world, counts)
    assert local.shape[0] == counts[dist.get_rank(group)], (local.shape, counts)
    if total == 0:
        return lambda: local[:0]
    mx = max(counts)
    tail = local.shape[1:]
    send = local
    if local.shape[0] != mx:
        send = torch.zeros((mx,) + tuple(tail), dtype=local.dtype, device=local.device)
        send[:local.shape[0]] = local
    recv = torch.empty((world * mx,) + tuple(tail), dtype=local.dtype, device=local.device)
    work = dist.all_gather_into_tensor(recv, send.contiguous(), group=group, async_op=True)

    def finish():
        work.wait()                                   # the current stream waits for the collective; the host does not block
        if all(c == mx for c in counts):
            return recv
        return torch.cat([recv[r * mx:r * mx + c] for r, c in enumerate(counts)], dim=0)
    return finish


def owned_pages(n_pages, world, rank):
    return list(range(rank, n_pages, world))


# ---- strong scaling: which ranks decode, and how many character tiles each rank encodes ----

# The one collective: every rank receives world x (largest shard) rows of 24 576 B (ragged shards are padded to the largest).  xGMI is point-to-point, 7 links
# x ~153 GB/s per GPU (MI355X_MICROARCH.md); an 8-rank all-gather of tens of MB has never been measured here (no node), so the rate is a STATED, conservative
# fifth of the aggregate link rate, and the latency a typical small-collective figure.  The term only decides between plans whose largest shards differ.
ALLGATHER_COST = {'allgather_latency_ms': 0.05, 'allgather_gbs': 200.0, 'char_row_bytes': 3 * 4096 * 2}

# Stage costs on one MI355X at InternVL2-8B shapes, bf16 (profiles/round5: 21_bench_N1_default.json phases, 24_final_decode_rows.txt,
# 35_decode_rows_9_to_32.txt).  Only the RATIOS matter to the plan; a box that is 3 % slower is 3 % slower at everything.
MI355X_COST = {
    'tile_ms': 0.71,                 # one 448 x 448 page tile through ViT + mlp1 ...
    'char_tile_ms': 0.70,            # ... one character tile through ViT + mlp1 + resampler + VQ + de-norm in long runs (1 711 tiles: 1 195 ms, 36_*)
    'prefill_ms_per_token': 0.011,   # 34.6 ms per 3 164-token page (splice + prefill + first token of 13 pages: 450 ms)
    # one decode step of n rows at ~3 200 cached tokens per row (weights 14.7 GB once + 131 KB of KV per row and cached token): linear between the points
    'decode_ms': {1: 2.97, 2: 3.11, 4: 3.30, 8: 3.88, 12: 4.28, 16: 4.58, 24: 5.65, 32: 6.28, 64: 8.86},
    'decode_ctx_tokens': 3228,            # the context the table was measured at (3 164-token prompt + half of 128 new tokens) ...
    'decode_ms_per_row_token': 2.2e-5,    # ... and what a row's step costs per cached token more or less (131 KB at the ~6 TB/s the decode attention streams at)
    # a visual-stage call runs in chunks of <= 255 tiles (csrc/vision.hip: next_chunk); every chunk STARTED costs this on top of its tiles (24 layers x 7 launches
    # of cold start and tail, the weights' first read): ragged shards and pages with few tiles pay it for little work (round 5: "the odd chunks cost ~3 ms of 620")
    'chunk_tiles': 255, 'chunk_ms': 1.5,
    **ALLGATHER_COST,
}

# the same with cr_enable_fp8_mfma level 2 + cr_enable_fp8_decode (BASELINE config 5; profiles/round5: 42_* fp8_mfma phases, 11_* / 14_* decode rows)
MI355X_COST_FP8 = {
    'tile_ms': 0.572, 'char_tile_ms': 0.572, 'prefill_ms_per_token': 0.00713,
    'decode_ms': {1: 2.2, 8: 3.04, 16: 3.54, 32: 4.75, 64: 7.97},
    'decode_ctx_tokens': 3228, 'decode_ms_per_row_token': 2.2e-5,      # (the KV cache stays bf16)
    'chunk_tiles': 255, 'chunk_ms': 1.2, **ALLGATHER_COST,
}


def decode_step_ms(rows, cost=MI355X_COST, ctx_tokens=None):
    """Piecewise-linear reading of the measured decode table (beyond its last point: the last slope); ctx_tokens: the rows' mean cached tokens
    when they differ from the table's (the KV-cache stream is the part of a step that grows with the context)."""
    if rows <= 0:
        return 0.0
    pts = sorted(cost['decode_ms'].items())
    if rows <= pts[0][0]:
        t = float(pts[0][1])
    else:
        (r0, t0), (r1, t1) = pts[-2], pts[-1]
        t = t1 + (t1 - t0) * (rows - r1) / (r1 - r0)
        for (r0, t0), (r1, t1) in zip(pts, pts[1:]):
            if rows <= r1:
                t = t0 + (t1 - t0) * (rows - r0) / (r1 - r0)
                break
    if ctx_tokens is not None and 'decode_ctx_tokens' in cost:
        t = max(t + rows * (ctx_tokens - cost['decode_ctx_tokens']) * cost.get('decode_ms_per_row_token', 0.0), 0.25 * t)
    return t


def _per_page(x, n, what):
    if isinstance(x, int):
        return [x] * n
    x = [int(v) for v in x]
    if len(x) != n or (x and min(x) < 0):
        raise ValueError(f'{what}: {len(x)} entries for {n} pages')
    return x


def _chunks(n, cost):
    return -(-n // cost.get('chunk_tiles', 255)) if n > 0 else 0


def gather_ms(counts, world, cost):
    """The all-gather of the character tiles' pseudo-token rows: world x (largest shard) rows received per rank."""
    if world <= 1 or not counts or max(counts) == 0:
        return 0.0
    return cost.get('allgather_latency_ms', 0.0) + world * max(counts) * cost.get('char_row_bytes', 24576) / (cost.get('allgather_gbs', 200.0) * 1e6)


def rank_ms(pages, n_char, pt, ct, tok, new_tokens, cost):
    """One rank's step under `cost`, without the gather: the character tiles of its shard (one visual call: chunks of <= chunk_tiles), and for a page owner its
    pages' own tiles (one call), prompts and the decode of its pages as one batch."""
    t = n_char * cost['char_tile_ms'] + _chunks(n_char, cost) * cost.get('chunk_ms', 0.0)
    if pages:
        tiles = sum(pt[p] for p in pages)
        t += tiles * cost['tile_ms'] + _chunks(tiles, cost) * cost.get('chunk_ms', 0.0) + sum(tok[p] for p in pages) * cost['prefill_ms_per_token']
        t += (new_tokens - 1) * decode_step_ms(len(pages), cost, sum(tok[p] for p in pages) / len(pages) + new_tokens / 2)
    return t


def evaluate_plan(plan, page_tiles, char_tiles, prompt_tokens, new_tokens, cost=None):
    """Per-rank ms and the step (slowest rank + the gather) of a plan_balanced / plan_even result under `cost` -- the plan's own model, also used to price a plan
    made under one cost table with another (tests: constants off by 10 %)."""
    cost = cost or MI355X_COST
    n_pages = len(plan['owner'])
    world = len(plan['char_counts'])
    pt, ct, tok = _per_page(page_tiles, n_pages, 'page_tiles'), _per_page(char_tiles, n_pages, 'char_tiles'), _per_page(prompt_tokens, n_pages, 'prompt_tokens')
    g = gather_ms(plan['char_counts'], world, cost)
    per = [rank_ms(plan['pages'][r], plan['char_counts'][r], pt, ct, tok, new_tokens, cost) + g for r in range(world)]
    return per, max(per)


def plan_balanced(n_pages, world, page_tiles, char_tiles, prompt_tokens, new_tokens, cost=MI355X_COST, owners=None, max_rows=64, min_gain=0.02):
    """Strong-scaling plan of one step of `n_pages` pages over `world` ranks.  page_tiles / char_tiles / prompt_tokens: one number for every
    page, or one per page (real pages differ in their character count and prompt length).

    The first k ranks own the pages (prefill + decode of their pages, and their pages' own tiles): pages go, dearest first, to the owner
    with the least work so far -- page p -> rank p % k when all pages are alike; ALL ranks share the flat list of character tiles (page
    order) in contiguous shards sized so that every rank finishes at the same time (a rank whose pages already fill the step gets none).
    k is the one whose slowest rank is fastest under `cost` (or `owners`, when the caller fixes it); k = world with even shards is the
    even split.  The model (rank_ms + gather_ms): tiles x ms + a fixed cost per 255-tile chunk started, prompt tokens x ms, the decode table by rows and context,
    and ONE all-gather whose padded size grows with the largest shard.  A plan with fewer owners than ranks is only taken when the model gives it `min_gain`
    (2 %) over the best plan with every rank an owner: constants that are off by a few per cent on another box must not turn a tie into a loss.
    No owner gets more than `max_rows` pages while enough ranks exist (64: the rows the weight-streaming decode kernels take in one launch; the
    table behind `cost` ends there).  Pure host arithmetic, deterministic: every rank computes the same plan.

    Returns {'k', 'pages': [[page ids] per rank], 'owner': [rank per page], 'char_counts': [per rank], 'char_bounds': [(lo, hi) per rank],
             'char_offsets': [first flat index per page] + [total], 'predicted_ms': [per rank], 'predicted_step_ms', 'predicted_even_ms'}."""
    if n_pages < 1 or world < 1:
        raise ValueError((n_pages, world))
    pt, ct, tok = _per_page(page_tiles, n_pages, 'page_tiles'), _per_page(char_tiles, n_pages, 'char_tiles'), _per_page(prompt_tokens, n_pages, 'prompt_tokens')
    total_ct = sum(ct)
    c_char = cost['char_tile_ms']
    page_ms = [pt[p] * cost['tile_ms'] + tok[p] * cost['prefill_ms_per_token'] for p in range(n_pages)]
    by_cost = sorted(range(n_pages), key=lambda p: (-page_ms[p], p))

    def assign(k):
        load, pages = [0.0] * k, [[] for _ in range(k)]
        for p in by_cost:
            r = min(range(k), key=lambda i: (load[i], i))
            load[r] += page_ms[p]
            pages[r].append(p)
        pages = [sorted(x) for x in pages] + [[] for _ in range(world - k)]
        return pages, [rank_ms(x, 0, pt, ct, tok, new_tokens, cost) for x in pages]

    def with_chars(fixed, c):
        return fixed + c * c_char + _chunks(c, cost) * cost.get('chunk_ms', 0.0)

    def fill(fixed):
        """Character tiles per rank: water-filling to a common finishing time, in whole tiles (the per-chunk cost is part of a rank's time)."""
        def take(f, level):                               # the most tiles a rank with fixed work f finishes by `level`
            n = int(max(0.0, (level - f) / c_char))
            while n > 0 and with_chars(f, n) > level:
                n -= max(1, int((with_chars(f, n) - level) / c_char))
            return max(n, 0)
        lo, hi = 0.0, max(fixed) + with_chars(0.0, total_ct)
        for _ in range(64):
            mid = 0.5 * (lo + hi)
            if sum(take(f, mid) for f in fixed) >= total_ct:
                hi = mid
            else:
                lo = mid
        counts = [take(f, hi) for f in fixed]
        left = total_ct - sum(counts)
        while left > 0:                                   # the tiles lost to rounding down: one at a time to the rank that finishes first
            r = min(range(world), key=lambda i: (with_chars(fixed[i], counts[i] + 1), i))
            counts[r] += 1
            left -= 1
        while left < 0:
            r = max((i for i in range(world) if counts[i] > 0), key=lambda i: (with_chars(fixed[i], counts[i]), -i))
            counts[r] -= 1
            left += 1
        return counts

    if owners is not None and not 1 <= owners <= min(world, n_pages):
        raise ValueError(f'{owners} page owners for {n_pages} pages over {world} ranks')
    k_max = min(world, n_pages)
    k_min = min(-(-n_pages // max_rows), k_max) if max_rows else 1
    cands = {}
    for k in ([owners] if owners is not None else range(k_min, k_max + 1)):
        pages, fixed = assign(k)
        counts = fill(fixed)
        g = gather_ms(counts, world, cost)
        t = [with_chars(f, c) + g for f, c in zip(fixed, counts)]
        cands[k] = {'k': k, 'pages': pages, 'char_counts': counts, 'predicted_ms': [round(x, 2) for x in t], 'predicted_step_ms': max(t)}
    best = min(cands.values(), key=lambda c: (c['predicted_step_ms'], -c['k']))
    if owners is None and best['k'] < k_max and best['predicted_step_ms'] > (1.0 - min_gain) * cands[k_max]['predicted_step_ms']:
        best = cands[k_max]
    even_fixed = assign(k_max)[1]
    even_ct = shard_counts(total_ct, world)
    bounds, lo = [], 0
    for c in best['char_counts']:
        bounds.append((lo, lo + c))
        lo += c
    offsets = [0]
    for c in ct:
        offsets.append(offsets[-1] + c)
    owner = [0] * n_pages
    for r, x in enumerate(best['pages']):
        for p in x:
            owner[p] = r
    return {'k': best['k'], 'pages': best['pages'], 'owner': owner, 'char_counts': best['char_counts'], 'char_bounds': bounds, 'char_offsets': offsets,
            'predicted_ms': best['predicted_ms'], 'predicted_step_ms': round(best['predicted_step_ms'], 2),
            'predicted_even_ms': round(max(with_chars(f, c) for f, c in zip(even_fixed, even_ct)) + gather_ms(even_ct, world, cost), 2)}


def default_cost(model=None):
    """The stage-cost table a plan is made with when the caller passes none: the fp8 table when the engine's fp8 options are on (their stage ratios differ:
    the visual stage and the prefill gain more than the decode), else the bf16 one.  Both were measured on ONE MI355X box (profiles/round5); on other hardware
    pass `cost=measure_cost(model)` (or cost='measure')."""
    eng = getattr(model, 'engine', None)
    if eng is not None and (getattr(eng, 'fp8_mfma', 0) or getattr(eng, 'fp8_decode', False)):
        return MI355X_COST_FP8
    return MI355X_COST


def measure_cost(model, group=None, seed=0):
    """The plan's constants MEASURED on this GPU with this model's weights and options, in about two seconds: two visual calls of different sizes (ms per tile and
    per chunk started), the same through the resampler + VQ + de-norm for character tiles, one 2 048-token prefill, decode steps at 1 / 8 / 16 / 64 rows on
    short contexts (the KV-cache term per cached token stays the table's: it is an HBM rate).  With a process group every rank measures (the ranks stay in step)
    and rank 0's numbers are used by all: every rank must compute the same plan."""
    import time
    from . import synthetic
    eng, dev = model.engine, model.engine.device
    dims = eng.dims
    base = dict(default_cost(model))
    px = synthetic.make_pixels(191, seed=seed, device=dev)

    def timed(fn, reps=1):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3
    n0, n1 = 47, 191                                   # both single chunks (next_chunk): t(n) = chunk_ms + n * tile_ms
    tp0, tp1 = timed(lambda: model.extract_feature(px[:n0])), timed(lambda: model.extract_feature(px[:n1]))
    tc0, tc1 = timed(lambda: model.align_tiles(px[:n0])), timed(lambda: model.align_tiles(px[:n1]))
    tile_ms, char_ms = (tp1 - tp0) / (n1 - n0), (tc1 - tc0) / (n1 - n0)
    chunk_ms = max(0.5 * ((tp0 - n0 * tile_ms) + (tc0 - n0 * char_ms)), 0.0)
    del px
    S, NP = max(min(2048, (eng.max_pos - 64) // 64 * 64), 128), 4      # four prompts in one prefill batch, as the page owners prefill (16 pages per batch at full size); within the model's RoPE table
    emb = (torch.randn(S, dims.llm_hidden, generator=torch.Generator().manual_seed(seed)) * 0.02).to(torch.bfloat16).to(dev)
    rows = [1, 8, 16, 64]
    ctx = 64
    kv = eng.kv_alloc(max(rows), S + 64)
    try:
        t_pre = timed(lambda: (kv.reset(), eng.prefill_batch(kv, list(range(NP)), [emb] * NP))) / NP
        kv.reset()
        eng.prefill_batch(kv, list(range(max(rows))), [emb[:ctx]] * max(rows))
        dec = {}
        for r in rows:
            live = list(range(r))
            dec[r] = timed(lambda: eng.decode(kv, live), reps=6)
    finally:
        kv.free()
    per_tok = base.get('decode_ms_per_row_token', 2.2e-5)
    cost = dict(base, tile_ms=round(tile_ms, 4), char_tile_ms=round(char_ms, 4), chunk_ms=round(chunk_ms, 3), prefill_ms_per_token=round(t_pre / S, 5),
                decode_ms={r: round(t, 3) for r, t in dec.items()}, decode_ctx_tokens=ctx + 4, decode_ms_per_row_token=per_tok, measured=True)
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        box = [cost]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        cost = box[0]
    return cost


def _resolve_cost(model, cost, group):
    if cost is None:
        return default_cost(model)
    if cost == 'measure':
        if getattr(model, '_measured_cost', None) is None:
            model._measured_cost = measure_cost(model, group)
        return model._measured_cost
    return cost


def sharded_generate(model, page_tiles, char_tiles, input_ids, *, img_id, ref_id, max_new_tokens, plan=None, cost=None, group=None, gather_results=True, **generate_kw):
    """One step of a batch of pages over the ranks of `group`: the multi-GPU form of `model.generate_pages` (the reference has none: inference.py:47-59
    is a serial page loop).  Every rank calls it with the SAME host-side lists --

        page_tiles[p]  (n_p, 3, 448, 448) the page's own tiles        char_tiles[p]  (c_p, 3, 448, 448) its character tiles, reading order
        input_ids[p]   the page's prompt ids (n_p * 256 `img_id` + c_p * 3 `ref_id` placeholders)

    -- and encodes only its share: its contiguous shard of the flat character-tile list (ViT -> mlp1 -> resampler -> VQ -> de-norm), ONE all-gather of the
    24.5 KB pseudo-token rows, then, for the pages it owns, their own tiles (ViT -> mlp1, under the gather), the splice, prefill and greedy decode.
    plan: a `plan_balanced` result (default: computed here from the lists' sizes -- fewer, fatter decode batches); `plan_even(...)` for one page owner per rank.
    cost: the stage-cost table of that default plan -- None: MI355X_COST, or MI355X_COST_FP8 when the engine's fp8 options are on (`default_cost`); 'measure': measured
    on this GPU once per model (`measure_cost`, ~2 s, rank 0's numbers on every rank); or a table of your own (other hardware).
    Returns {page: ids} for every page (gather_results) or for the pages this rank owns.  Per page the ids are the single-process ids."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    n_pages = len(input_ids)
    if len(page_tiles) != n_pages or len(char_tiles) != n_pages:
        raise ValueError('page_tiles, char_tiles and input_ids are one entry per page')
    if plan is None:
        plan = plan_balanced(n_pages, world, [int(t.shape[0]) for t in page_tiles], [int(t.shape[0]) for t in char_tiles], [int(i.numel()) for i in input_ids], max_new_tokens,
                             cost=_resolve_cost(model, cost, group))
    if len(plan['char_counts']) != world or plan['char_offsets'][-1] != sum(int(t.shape[0]) for t in char_tiles):
        raise ValueError('the plan was made for another world size or another batch')
    dev = model.engine.device
    hidden = model.engine.dims.llm_hidden
    lo, hi = plan['char_bounds'][rank]
    off = plan['char_offsets']
    if hi > lo:                                          # the pieces of the flat list [lo, hi), page by page
        parts = [char_tiles[p][max(lo, off[p]) - off[p]:min(hi, off[p + 1]) - off[p]] for p in range(n_pages) if off[p] < hi and off[p + 1] > lo]
        pseudo, _ = model.align_tiles(torch.cat(parts).to(dev))
        pseudo = pseudo.reshape(-1, 3, hidden)
    else:
        pseudo = torch.empty((0, 3, hidden), dtype=torch.bfloat16, device=dev)
    finish = all_gather_rows_async(pseudo, off[-1], group, counts=plan['char_counts'])
    mine = plan['pages'][rank]
    vit = {p: None for p in mine}                        # (pages without tiles of their own exist: c_p only)
    with_tiles = [p for p in mine if page_tiles[p].shape[0]]
    if with_tiles:                                       # ONE visual call for the owned pages' tiles (255-tile chunks), not one small chunk per page
        feats, o = model.extract_feature(torch.cat([page_tiles[p].to(dev) for p in with_tiles])), 0
        for p in with_tiles:
            vit[p] = feats[o:o + page_tiles[p].shape[0]]
            o += page_tiles[p].shape[0]
    pseudo_all = finish()
    embeds = [model.engine.embed_splice(input_ids[p].to(dev), vit[p], pseudo_all[off[p]:off[p + 1]] if off[p + 1] > off[p] else None, img_id=img_id, ref_id=ref_id)
              for p in mine]
    outs = model.generate_pages(embeds, max_new_tokens=max_new_tokens, **generate_kw) if embeds else []
    result = dict(zip(mine, outs))
    if gather_results and world > 1:
        got = [None] * world
        dist.all_gather_object(got, result, group=group)
        result = {}
        for g in got:
            result.update(g)
    return result


def plan_even(n_pages, world, page_tiles, char_tiles, prompt_tokens, new_tokens, cost=MI355X_COST):
    """The even split in plan_balanced's format: page p -> rank p % world, character tiles in even contiguous shards."""
    pl = plan_balanced(n_pages, world, page_tiles, char_tiles, prompt_tokens, new_tokens, cost=cost, owners=min(world, n_pages))
    k = min(world, n_pages)
    pl['pages'] = [list(range(r, n_pages, k)) if r < k else [] for r in range(world)]
    pl['owner'] = [p % k for p in range(n_pages)]
    pl['char_counts'] = shard_counts(pl['char_offsets'][-1], world)
    pl['char_bounds'] = [shard_range(pl['char_offsets'][-1], world, r) for r in range(world)]
    per, step = evaluate_plan(pl, page_tiles, char_tiles, prompt_tokens, new_tokens, cost)
    pl['predicted_ms'], pl['predicted_step_ms'] = [round(x, 2) for x in per], round(step, 2)
    return pl


def chat_ocr_pages_sharded(model, tokenizer, detect_model, images, question, generation_config, boxes_list=None, hard_vq=False, repetition_penalty=1.5,
                           plan=None, cost=None, group=None, IMG_START_TOKEN='<img>', IMG_END_TOKEN='</img>', IMG_CONTEXT_TOKEN='<IMG_CONTEXT>',
                           ALIGNED_TOKEN='[UNUSED_TOKEN_140]'):
    """`model.chat_ocr_pages` over the ranks of `group`: every rank calls it with the same pages (paths or PIL images) and gets every page's response -- the one
    its own single-GPU `chat_ocr(..., use_p=True, drop_zero=False)` call gives (modeling_internvl_chat.py:649-763 per page; the reference has no multi-GPU form).

    Detection + reading order (when `boxes_list` is None) run on rank p % world for page p and the boxes are exchanged as objects; then the flow of
    `sharded_generate`: each rank crops and encodes ITS contiguous shard of the flat character-box list (GPU resize / pad / normalise, ViT -> mlp1 -> resampler ->
    VQ -> de-norm), one all-gather of the pseudo-token rows, and the owners of a page (plan_balanced over the pages' real sizes, or `plan`) encode its own
    tiles, splice, prefill and decode.  drop_zero is not offered here: it makes a page's prompt length depend on the VQ result of tiles another rank encodes."""
    from PIL import Image
    from . import ordering
    from .conversation import get_conv_template
    from .preprocess import plan_page, plan_chars_array, jobs_array
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    n_pages = len(images)
    eng = model.engine
    hidden = eng.dims.llm_hidden
    model.img_context_token_id = tokenizer.convert_tokens_to_ids(IMG_CONTEXT_TOKEN)
    # ---- pages and the boxes of every page, in reading order: given, or detected where the page lives and exchanged ----
    # A rank that cannot read a page or whose detector fails must not leave the others waiting in the collective (round-5 advice): every failure becomes a
    # marker, the markers are exchanged with the boxes, and EVERY rank raises the same error after the exchange.
    # Every rank needs every page's SIZE (tile counts and prompt lengths are the plan's inputs) but the PIXELS only of the pages it touches: sizes come from the
    # files' headers, pixels are decoded on threads (pageio: 14.5 ms of JPEG decode per example page -- 64 pages decoded one after the other on every rank would
    # cost more than a rank's share of the step at 8 GPUs).  Readability is checked where a page lives (rank p % world decodes it) and exchanged with the boxes.
    from .pageio import decode_pages
    sizes, failures, decoded = [None] * n_pages, {}, {}
    for p, im in enumerate(images):
        try:
            if isinstance(im, str):
                with Image.open(im) as f:
                    sizes[p] = f.size
            else:
                sizes[p] = im.size
        except Exception as e:
            failures[p] = f'page {p}: cannot be read: {type(e).__name__}: {e}'

    def need(idx):
        """decode the pages of `idx` that are not decoded yet (thread pool); a failure becomes a marker"""
        todo = [p for p in idx if p not in decoded and p not in failures]
        for p, (page, arr, _) in zip(todo, decode_pages([images[p] for p in todo], eng.device)):
            if page is None:
                failures[p] = f'page {p}: cannot be read: {type(arr).__name__}: {arr}'
            else:
                decoded[p] = (page, arr)
    need(range(rank, n_pages, world))
    found = {}
    if boxes_list is None:
        for p in range(rank, n_pages, world):
            if p in failures:
                continue
            try:
                found[p] = [[int(v) for v in b[:4]] for b in ordering.acquire_boxes(detect_model, decoded[p][0], model.sorter)]
            except Exception as e:
                failures[p] = f'page {p}: detection failed on rank {rank}: {type(e).__name__}: {e}'
    if world > 1:                                        # boxes (when they were found here) and failure markers, in one exchange
        got = [None] * world
        dist.all_gather_object(got, (found, failures), group=group)
        found = {k: v for g in got for k, v in g[0].items()}
        failures = {k: v for g in got for k, v in g[1].items()}
    if failures:
        raise RuntimeError('chat_ocr_pages_sharded: ' + '; '.join(failures[p] for p in sorted(failures)))
    if boxes_list is None:
        boxes_list = [found.get(p, []) for p in range(n_pages)]
    if any(len(b) == 0 for b in boxes_list):
        raise RuntimeError('chat_ocr_pages_sharded: a page without character boxes (chat_ocr fails on it too: modeling_internvl_chat.py:585)')
    # ---- prompts (host work, the same on every rank): their lengths are the plan's prefill term ----
    page_jobs = [plan_page(*sizes[p]) for p in range(n_pages)]              # (jobs, tiles) per page
    n_chars = [len(b) for b in boxes_list]
    template = get_conv_template(model.template)
    gen = dict(generation_config)
    gen['eos_token_id'] = tokenizer.convert_tokens_to_ids(template.sep)
    max_new, eos = model._gen_args(gen)
    q = question if '<image>' in question else '<image>\n' + question
    ids = [model._prompt_ids(tokenizer, q, page_jobs[p][1], 3 * n_chars[p], IMG_START_TOKEN, IMG_END_TOKEN, IMG_CONTEXT_TOKEN, ALIGNED_TOKEN) for p in range(n_pages)]
    if plan is None:
        plan = plan_balanced(n_pages, world, [j[1] for j in page_jobs], n_chars, [int(i.numel()) for i in ids], max_new, cost=_resolve_cost(model, cost, group))
    off = plan['char_offsets']
    if len(plan['char_counts']) != world or off[-1] != sum(n_chars):
        raise ValueError('the plan was made for another world size or another batch')
    # ---- my shard of the flat character-box list; the pixels of the pages it touches and of the pages I own: decoded now, on threads ----
    lo, hi = plan['char_bounds'][rank]
    mine = plan['pages'][rank]
    touched = [p for p in range(n_pages) if max(lo, off[p]) < min(hi, off[p + 1])]
    need(sorted(set(touched) | set(mine)))
    if failures:                                         # (every page was decoded once where it lives: this only fires if a file changed under us)
        raise RuntimeError('chat_ocr_pages_sharded: ' + '; '.join(failures[p] for p in sorted(failures)))
    dev_px = {p: decoded[p][1].to(eng.device, non_blocking=True) for p in sorted(set(touched) | set(mine))}
    tot = hi - lo
    char_px = torch.empty(tot, 3, eng.dims.image_size, eng.dims.image_size, device=eng.device, dtype=torch.bfloat16) if tot else None
    o = 0
    for p in touched:
        a, b = max(lo, off[p]), min(hi, off[p + 1])
        w, h = sizes[p]
        eng.preprocess(dev_px[p], plan_chars_array(boxes_list[p][a - off[p]:b - off[p]], w, h, tile0=o), tot, out=char_px)
        o += b - a
    if tot:
        pseudo, _ = model.align_tiles(char_px, drop_zero=False, use_hard_vector_quant=hard_vq)
        pseudo = pseudo.reshape(-1, 3, hidden)
    else:
        pseudo = torch.empty((0, 3, hidden), dtype=torch.bfloat16, device=eng.device)
    finish = all_gather_rows_async(pseudo, off[-1], group, counts=plan['char_counts'])
    # ---- the pages I own: their tiles (ONE visual call) under the gather, splice, prefill + decode ----
    feats = {}
    if mine:
        n_own = sum(page_jobs[p][1] for p in mine)
        own_px = torch.empty(n_own, 3, eng.dims.image_size, eng.dims.image_size, device=eng.device, dtype=torch.bfloat16)
        o = 0
        for p in mine:
            eng.preprocess(dev_px[p], jobs_array(plan_page(*sizes[p], tile0=o)[0]), n_own, out=own_px)
            o += page_jobs[p][1]
        f_all, o = model.extract_feature(own_px), 0
        for p in mine:
            feats[p] = f_all[o:o + page_jobs[p][1]]
            o += page_jobs[p][1]
    pseudo_all = finish()
    embeds = [eng.embed_splice(ids[p], feats[p], pseudo_all[off[p]:off[p + 1]], img_id=model.img_context_token_id, ref_id=model.aligned_token_id) for p in mine]
    outs = model.generate_pages(embeds, max_new, eos, repetition_penalty) if embeds else []
    result = {p: tokenizer.batch_decode(torch.tensor([o]), skip_special_tokens=True)[0].split(template.sep)[0].strip() for p, o in zip(mine, outs)}
    if world > 1:
        got = [None] * world
        dist.all_gather_object(got, result, group=group)
        result = {k: v for g in got for k, v in g.items()}
    return [result[p] for p in range(n_pages)]
