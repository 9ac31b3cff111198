"""The optional blocks of bench.py's JSON line (everything here is UNTIMED: it runs after the timed steps, on the same model object):
  strong_share_block    N = 1: one rank's share of BASELINE config 4 as written (64 pages over 8 GPUs) alone on this GPU, incl. the balanced plan (uniform and RAGGED pages)
  strong_scaling_block  N > 1: config 4 as written in the same process group (even split, and the balanced plan)
  single_gpu_extras     N = 1: BASELINE config 2 (ViT only) and 3 (one image), tile preprocessing, the ordering front end, the fp8 options
`S` is bench.py's state (args, model, inputs, step(), timings)."""
import os
import sys
import time

import torch
import torch.distributed as dist

from . import plan
from .plan import PAGE_TILES, CHAR_TILES, TEXT_TOKENS, build_ids, plan_workload, plan_strong_share
from .measure import PEAK_BF16_TFLOPS, PEAK_HBM_GBS
from callireader_amd.config import IMG_CONTEXT_TOKEN_ID, ALIGNED_TOKEN_ID


def measure_balanced(S, pb, full_ms, full_out, pipelined_ms, n_even, cost_name='MI355X_COST'):
    """strong_share.balanced: the two kinds of rank of a plan_balanced plan, each timed alone on this GPU against the one-GPU step `full_ms` (whose ids are `full_out`)."""
    args, world, rank, dev, dims, model, eng = S.args, S.world, S.rank, S.dev, S.dims, S.model, S.eng
    wl, P, S_page, n_pages, mine, ct_lo, ct_hi = S.wl, S.P, S.S_page, S.n_pages, S.mine, S.ct_lo, S.ct_hi
    page_px, char_px, ids, step, sync, make_inputs = S.page_px, S.char_px, S.ids, S.step, S.sync, S.make_inputs
    elapsed, seq_ms, seq_out = S.elapsed, S.seq_ms, S.seq_out
    NEW_TOKENS = plan.NEW_TOKENS
    sw = args.share_world
    if pb['k'] >= sw:
        return None
    ra = max(range(sw), key=lambda r: (len(pb['pages'][r]), pb['char_counts'][r], -r))
    rb = max(range(sw), key=lambda r: (pb['char_counts'][r], -r))
    nA, cA, cB = len(pb['pages'][ra]), pb['char_counts'][ra], pb['char_counts'][rb]

    def timed_share(w, ins_):
        step(w=w, inputs=ins_)
        torch.cuda.synchronize()
        t0_ = time.perf_counter()
        out_ = None
        for _ in range(args.share_steps):
            out_ = step(w=w, inputs=ins_)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0_) / args.share_steps * 1e3, out_
    pseudo_a, _ = model.align_tiles(char_px[:nA * CHAR_TILES])                       # what the gather hands rank A for its pages
    w_a = {'n_pages': nA, 'mine': list(range(nA)), 'pseudo_all': pseudo_a.reshape(-1, 3, dims.llm_hidden)}
    ins_a = (page_px[:nA * PAGE_TILES], char_px[:cA], ids[:nA])
    t_a, out_a = timed_share(w_a, ins_a)
    st4 = [0.0]
    torch.cuda.synchronize(); st4[0] = time.perf_counter()
    step(new_tokens=1, stamps=st4, w=w_a, inputs=ins_a)
    same_a = bool(out_a == full_out[:nA])
    del pseudo_a, w_a, ins_a
    t_b = None
    if not pb['pages'][rb]:
        w_b = {'n_pages': 0, 'mine': [], 'pseudo_all': torch.empty((0, 3, dims.llm_hidden), dtype=torch.bfloat16, device=dev)}
        t_b, _ = timed_share(w_b, (page_px[:0], char_px[:cB], []))
    t_bal = max(t_a, t_b or 0.0)
    return {
        'what': f'the same {args.pages} pages over {sw} GPUs under parallel.plan_balanced: {pb["k"]} ranks own the pages ({nA} rows per decode batch instead of {n_even}; '
                'the decode streams the weights once per step whatever the rows), all ranks share the character tiles in uneven contiguous shards, still ONE all-gather; '
                'the two kinds of rank timed alone on this GPU, one batch at a time',
        'plan': {'page_owners': pb['k'], 'pages_per_rank': [len(x) for x in pb['pages']], 'char_tiles_per_rank': pb['char_counts'],
                 'predicted_ms_per_rank': pb['predicted_ms'], 'predicted_even_plan_ms': pb['predicted_even_ms'],
                 'cost_model': f'callireader_amd/parallel.py: {cost_name} (ms per tile, per prompt token, per decode step by rows), measured in profiles/round5'},
        'page_owner_rank': {'rank': ra, 'pages_owned': nA, 'char_tiles': cA, 't_ms': round(t_a, 2),
                            'phases_ms': {'visual': round((st4[1] - st4[0]) * 1e3, 1), 'splice_prefill_first_token': round((st4[2] - st4[1]) * 1e3, 1),
                                          'decode_remaining_tokens': round(max(t_a - (st4[2] - st4[0]) * 1e3, 0.0), 1),
                                          'decode_ms_per_step': round(max(t_a - (st4[2] - st4[0]) * 1e3, 0.0) / max(NEW_TOKENS - 1, 1), 4)},
                            'ids_equal_the_same_pages_of_the_full_step': same_a},
        'tile_rank': None if t_b is None else {'rank': rb, 'pages_owned': 0, 'char_tiles': cB, 't_ms': round(t_b, 2)},
        't_step_ms': round(t_bal, 2),
        f'projected_speedup_{sw}': round(full_ms / t_bal, 3),
        **({} if pipelined_ms is None else {f'projected_speedup_{sw}_vs_pipelined_n1': round(pipelined_ms / t_bal, 3)}),
        'projection_note': 'ms of the one-GPU step / ms of the slower kind of rank; excludes the all-gather (uneven shards padded to the largest: '
                           f'{sw} x {max(pb["char_counts"])} x 24.5 KB received per rank) and assumes the other ranks of a kind take as long as the one timed'}


def ragged_balanced(S, seed=6):
    """strong_share.balanced on a RAGGED batch (round-5 verdict, item 4d): the same number of pages with 3-13 page tiles and 10-250 character tiles each (seeded), planned
    by parallel.plan_balanced under constants MEASURED on this GPU at start-up (parallel.measure_cost, ~1 s), against the even split of the same batch.  The whole
    ragged step and the two kinds of rank of each plan (the slowest page owner, the rank with the most tiles) are timed alone on this GPU, one batch at a time."""
    import random
    from callireader_amd import parallel, synthetic
    args, world, rank, dev, dims, model, eng = S.args, S.world, S.rank, S.dev, S.dims, S.model, S.eng
    wl, P, S_page, n_pages, mine, ct_lo, ct_hi = S.wl, S.P, S.S_page, S.n_pages, S.mine, S.ct_lo, S.ct_hi
    page_px, char_px, ids, step, sync, make_inputs = S.page_px, S.char_px, S.ids, S.step, S.sync, S.make_inputs
    elapsed, seq_ms, seq_out = S.elapsed, S.seq_ms, S.seq_out
    NEW_TOKENS = plan.NEW_TOKENS
    sw = args.share_world
    rng = random.Random(seed)
    n = args.pages
    pt = [rng.randint(3, 13) for _ in range(n)]
    ct = [rng.randint(10, 250) for _ in range(n)]
    tok = [pt[p] * 256 + ct[p] * 3 + TEXT_TOKENS for p in range(n)]
    t0 = time.perf_counter()
    cost = parallel.measure_cost(model)
    t_cost = time.perf_counter() - t0
    pb = parallel.plan_balanced(n, sw, pt, ct, tok, NEW_TOKENS, cost=cost)
    pe = parallel.plan_even(n, sw, pt, ct, tok, NEW_TOKENS, cost=cost)
    off = pb['char_offsets']
    page_all = synthetic.make_pixels(sum(pt), seed=70, device=dev)
    char_all = synthetic.make_pixels(off[-1], seed=71, device=dev)
    poff = [0]
    for x in pt:
        poff.append(poff[-1] + x)
    ids_r = [build_ids(pt[p], ct[p], TEXT_TOKENS, IMG_CONTEXT_TOKEN_ID, ALIGNED_TOKEN_ID, 3000 + p).to(dev) for p in range(n)]
    pseudo_all, _ = model.align_tiles(char_all)                     # what the all-gather hands every rank
    pseudo_all = pseudo_all.reshape(-1, 3, dims.llm_hidden)

    def rank_work(pages, lo, hi):
        if hi > lo:
            model.align_tiles(char_all[lo:hi])
        outs = []
        if pages:
            feats = model.extract_feature(torch.cat([page_all[poff[p]:poff[p + 1]] for p in pages]))
            embeds, o = [], 0
            for p in pages:
                embeds.append(eng.embed_splice(ids_r[p], feats[o:o + pt[p]], pseudo_all[off[p]:off[p + 1]], img_id=IMG_CONTEXT_TOKEN_ID, ref_id=ALIGNED_TOKEN_ID))
                o += pt[p]
            outs = model.generate_pages(embeds, max_new_tokens=NEW_TOKENS, eos_token_id=None)
        return outs

    def timed(pages, lo, hi, reps):
        rank_work(pages, lo, hi)
        torch.cuda.synchronize()
        t0_ = time.perf_counter()
        out_ = None
        for _ in range(reps):
            out_ = rank_work(pages, lo, hi)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0_) / reps * 1e3, out_
    full_ms, full_out = timed(list(range(n)), 0, off[-1], 1)

    def kinds(pl):
        per = pl['predicted_ms']
        owners = [r for r in range(sw) if pl['pages'][r]]
        tilers = [r for r in range(sw) if not pl['pages'][r]]
        ra = max(owners, key=lambda r: (per[r], -r))
        res = {'page_owner_rank': None, 'tile_rank': None}
        lo, hi = pl['char_bounds'][ra]
        t_a, out_a = timed(pl['pages'][ra], lo, hi, args.share_steps)
        res['page_owner_rank'] = {'rank': ra, 'pages_owned': len(pl['pages'][ra]), 'page_tiles': sum(pt[p] for p in pl['pages'][ra]), 'char_tiles': hi - lo,
                                  'predicted_ms': per[ra], 't_ms': round(t_a, 2), 'ids_equal_the_same_pages_of_the_full_step': bool(out_a == [full_out[p] for p in pl['pages'][ra]])}
        t_b = 0.0
        if tilers:
            rb = max(tilers, key=lambda r: (per[r], -r))
            lo, hi = pl['char_bounds'][rb]
            t_b, _ = timed([], lo, hi, args.share_steps)
            res['tile_rank'] = {'rank': rb, 'pages_owned': 0, 'char_tiles': hi - lo, 'predicted_ms': per[rb], 't_ms': round(t_b, 2)}
        res['t_step_ms'] = round(max(t_a, t_b), 2)
        res[f'projected_speedup_{sw}'] = round(full_ms / max(t_a, t_b), 3)
        return res
    out = {'what': f'{n} RAGGED pages (3-13 page tiles, 10-250 character tiles each; {sum(pt)} + {off[-1]} tiles, prompts of {min(tok)}-{max(tok)} tokens) over {sw} GPUs: parallel.plan_balanced '
                   'under stage costs measured on this GPU at start-up (parallel.measure_cost) against the even split of the same batch; per plan the slowest page owner and the '
                   'rank with the most tiles, each timed alone on this GPU, one batch at a time, against the whole ragged step on this GPU',
           'measured_cost': {k: cost[k] for k in ('tile_ms', 'char_tile_ms', 'chunk_ms', 'prefill_ms_per_token', 'decode_ms', 'decode_ctx_tokens')}, 'measure_cost_s': round(t_cost, 2),
           'full_step_one_gpu_ms': round(full_ms, 1),
           'balanced': dict(kinds(pb), page_owners=pb['k'], pages_per_rank=[len(x) for x in pb['pages']], char_tiles_per_rank=pb['char_counts'], predicted_step_ms=pb['predicted_step_ms']),
           'even': dict(kinds(pe), char_tiles_per_rank=pe['char_counts'], predicted_step_ms=pe['predicted_step_ms']),
           'projection_note': 'ms of the one-GPU ragged step / ms of the slower kind of rank; the all-gather is in the plan\'s model (stated 200 GB/s) but not in the measured times; '
                              'the other ranks of a kind are assumed to take as long as the one timed (the plan balances them to within a tile)'}
    del page_all, char_all, pseudo_all
    return out


def strong_share_block(S):
    """strong_share of the N = 1 line, or None."""
    args, world, rank, dev, dims, model, eng = S.args, S.world, S.rank, S.dev, S.dims, S.model, S.eng
    wl, P, S_page, n_pages, mine, ct_lo, ct_hi = S.wl, S.P, S.S_page, S.n_pages, S.mine, S.ct_lo, S.ct_hi
    page_px, char_px, ids, step, sync, make_inputs = S.page_px, S.char_px, S.ids, S.step, S.sync, S.make_inputs
    elapsed, seq_ms, seq_out = S.elapsed, S.seq_ms, S.seq_out
    NEW_TOKENS = plan.NEW_TOKENS
    # ---- N = 1: one rank's SHARE of BASELINE config 4 as written (64 pages over 8 GPUs), timed on the one GPU there is ----
    # The only evidence for north_star's ">= 6x at 8 GPUs" that can exist without a node: plan ('strong', 64 pages, world 8, rank 0) = 8 pages to own
    # (88 page tiles, 8 prompts, NEW_TOKENS - 1 eight-row decode steps) + an eighth of the character tiles (768), run alone on this GPU, one batch at
    # a time, next to the 64-page step of the same run.  The share's pages are pages 0..7 of the 64-page step (a rank of the real run owns pages
    # r, r + 8, ...: the same amount of work), so that the ids can be compared: a page's result does not depend on its batch.
    strong_share = None
    if world == 1 and rank == 0 and args.scaling == 'weak' and not args.no_strong_share and args.pages >= args.share_world and args.pages % args.share_world == 0:
        w_share = plan_strong_share(args.pages, args.share_world)
        n_own, n_ct = w_share['pages_per_gpu'], w_share['ct_hi']
        ins = (page_px[:n_own * PAGE_TILES], char_px[:n_ct], ids[:n_own])
        step(w=w_share, inputs=ins)                                # untimed warm-up (workspace sizes, kernel attributes of the 8-row forms)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        share_out = None
        for _ in range(args.share_steps):
            share_out = step(w=w_share, inputs=ins)
        torch.cuda.synchronize()
        t_share = (time.perf_counter() - t0) / args.share_steps
        st3 = [0.0]
        torch.cuda.synchronize(); st3[0] = time.perf_counter()
        step(new_tokens=1, stamps=st3, w=w_share, inputs=ins)
        full_ms = seq_ms
        full_out = seq_out
        if full_ms is None:                                        # --no-pipeline: the timed steps were one batch at a time already
            full_ms = elapsed / args.steps * 1e3
            full_out = step()
            torch.cuda.synchronize()
        same_share = bool(share_out == full_out[:n_own])
        vis_ms, pre_ms = (st3[1] - st3[0]) * 1e3, (st3[2] - st3[1]) * 1e3
        # The same rank with its decode batch fed from TWO consecutive steps: visual stage + splice of step A, of step B, then one prefill and ONE decode over both steps'
        # pages (16 rows per rank at 64 pages over 8): the weights are streamed once per two steps' pages.  Throughput view of the same configuration (two steps in
        # flight, as PagePipeline keeps them at N = 1), measurable on one GPU like the share itself; per-step time = the merged pass / 2.
        merged2 = None
        if args.two_steps_one_decode and 2 * n_own <= args.pages and 2 * n_own <= 64:
            ins2 = [(page_px[k * n_own * PAGE_TILES:(k + 1) * n_own * PAGE_TILES], char_px[k * n_ct:(k + 1) * n_ct], ids[k * n_own:(k + 1) * n_own]) for k in range(2)]
            step(w=w_share, merged=ins2)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            m_out = step(w=w_share, merged=ins2)
            torch.cuda.synchronize()
            t_m = (time.perf_counter() - t0) / 2
            merged2 = {'what': f'two consecutive steps of that share with ONE decode over both steps\' pages ({2 * n_own} rows per rank instead of {n_own}): visual stage + splice of step A, of step B, '
                               'one prefill, one decode; a page\'s ids do not depend on its batch',
                       'ms_per_step': round(t_m * 1e3, 2), f'projected_speedup_{args.share_world}': round(full_ms / (t_m * 1e3), 3),
                       'ids_equal_the_same_pages_of_the_full_step': bool(m_out == full_out[:2 * n_own]),
                       'note': 'a throughput arrangement (a page waits for the next step\'s pages before it decodes); the one-batch-at-a-time share above is the latency view'}
            del ins2
        # The same 64 pages over 8 GPUs under the BALANCED plan (parallel.plan_balanced): fewer ranks own pages (fatter decode batches: the weights are
        # streamed once per step whatever the rows), the others encode more character tiles.  Two kinds of rank, each timed alone on this GPU: the page owner
        # with the most work (its pages' other character tiles come out of the all-gather: made beforehand, handed in) and the rank with the most tiles.
        balanced = None
        if not args.no_balanced:
            balanced = measure_balanced(S, plan_workload('strong', args.pages, args.pages, args.share_world, 0, plan='balanced', owners=args.balanced_owners)['balanced'],
                                        full_ms, full_out, elapsed / args.steps * 1e3, n_own)
        ragged = None
        if not args.no_balanced and not args.no_ragged:
            try:
                ragged = ragged_balanced(S)
            except Exception as e:
                import traceback
                traceback.print_exc()
                ragged = {'error': f'{type(e).__name__}: {e}'}
        strong_share = {
            'what': f'one rank\'s share of BASELINE config 4 as written ({args.pages} pages per step over {args.share_world} GPUs, plan_workload(strong, rank 0)) run ALONE on this one GPU, '
                    f'one batch at a time: {n_own} pages owned ({n_own * PAGE_TILES} page tiles, {n_own} prompts of {S_page} tokens, {NEW_TOKENS - 1} decode steps of {n_own} rows) '
                    f'+ {n_ct} of the {args.pages * CHAR_TILES} character tiles',
            'world_projected': args.share_world, 'pages_owned': n_own, 'char_tiles': n_ct, 'steps': args.share_steps,
            't_share_ms': round(t_share * 1e3, 2),
            'phases_ms': {'visual': round(vis_ms, 1), 'splice_prefill_first_token': round(pre_ms, 1),
                          'decode_remaining_tokens': round(max(t_share * 1e3 - vis_ms - pre_ms, 0.0), 1),
                          'decode_ms_per_step': round(max(t_share * 1e3 - vis_ms - pre_ms, 0.0) / max(NEW_TOKENS - 1, 1), 4),
                          'how': 'one extra stamped pass that stops after the first token; decode = timed share - those two'},
            'full_step_one_batch_at_a_time_ms': round(full_ms, 1),
            f'projected_speedup_{args.share_world}': round(full_ms / (t_share * 1e3), 3),
            f'projected_speedup_{args.share_world}_vs_pipelined_n1': round((elapsed / args.steps * 1e3) / (t_share * 1e3), 3),
            'projection_note': f'upper bound: ms of the {args.pages}-page step on one GPU / ms of one rank\'s share; excludes the all-gather (24.5 KB per character tile, '
                               'started under the page tiles\' ViT) and rank skew (every rank has the same tile and page counts at 64 pages over 8); '
                               'the second ratio is against the headline N = 1 step (two batches in flight)',
            'ids_equal_the_same_pages_of_the_full_step': same_share,
            'two_steps_one_decode': merged2,
            'balanced': balanced,
            'balanced_ragged': ragged}
        del ins

    return strong_share


def strong_scaling_block(S, gather_standalone):
    """strong_scaling of the N > 1 line, or None."""
    args, world, rank, dev, dims, model, eng = S.args, S.world, S.rank, S.dev, S.dims, S.model, S.eng
    wl, P, S_page, n_pages, mine, ct_lo, ct_hi = S.wl, S.P, S.S_page, S.n_pages, S.mine, S.ct_lo, S.ct_hi
    page_px, char_px, ids, step, sync, make_inputs = S.page_px, S.char_px, S.ids, S.step, S.sync, S.make_inputs
    elapsed, seq_ms, seq_out = S.elapsed, S.seq_ms, S.seq_out
    NEW_TOKENS = plan.NEW_TOKENS
    # ---- N > 1, weak scaling (what the driver's one command runs): BASELINE config 4 AS WRITTEN in the same process group ----
    # `--total-pages` per step over ALL ranks (64 pages over 8 GPUs = 8 per GPU): the number north_star's ">= 6x at 8 GPUs" is about.  Weak scaling
    # is >= 6x almost by construction (the one collective is 24.5 KB per character tile); strong scaling carries the Amdahl term of the
    # small-batch decode.  One batch at a time (8 pages per GPU leave a second batch nothing to hide behind), untimed warm-up step, then
    # --strong-steps timed steps between barriers, MAX over ranks; phases from one extra stamped pass.
    strong = None
    if world > 1 and args.scaling == 'weak' and not args.no_strong_block:
        ws = plan_workload('strong', args.pages, args.total_pages, world, rank)
        ins = make_inputs(ws)
        step(w=ws, inputs=ins)
        sync()
        t0 = time.perf_counter()
        for _ in range(args.strong_steps):
            step(w=ws, inputs=ins)
        sync()
        el = time.perf_counter() - t0
        st2 = [0.0]
        sync(); st2[0] = time.perf_counter()
        step(new_tokens=1, stamps=st2, w=ws, inputs=ins)
        sync()
        t = torch.tensor([el, st2[1] - st2[0], st2[2] - st2[1]], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el, vis_s, pre_s = (float(x) for x in t.tolist())
        g2 = gather_standalone(ws)
        per_step = el / args.strong_steps
        # the same with every rank's decode batch fed from two consecutive steps (strong_share.two_steps_one_decode at N = 1): two visual stages + gathers, one decode
        merged_ms = None
        if args.two_steps_one_decode and 2 * ws['pages_per_gpu'] <= 64:
            step(w=ws, merged=[ins, ins])
            sync()
            t0 = time.perf_counter()
            step(w=ws, merged=[ins, ins])
            sync()
            tm = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
            merged_ms = float(tm.item()) / 2 * 1e3
        # ... and under the balanced plan (parallel.plan_balanced: fewer page owners, uneven character-tile shards, the same one all-gather)
        bal = None
        wbal = plan_workload('strong', args.pages, args.total_pages, world, rank, plan='balanced')
        if wbal['balanced']['k'] < world and max(len(x) for x in wbal['balanced']['pages']) <= P and not args.no_balanced:
            ins_b = make_inputs(wbal)
            step(w=wbal, inputs=ins_b)
            sync()
            t0 = time.perf_counter()
            for _ in range(args.strong_steps):
                step(w=wbal, inputs=ins_b)
            sync()
            tb = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
            dist.all_reduce(tb, op=dist.ReduceOp.MAX)
            bal_ms = float(tb.item()) / args.strong_steps * 1e3
            pb = wbal['balanced']
            bal = {'what': 'the same step under parallel.plan_balanced: fewer ranks own pages (fatter decode batches), the others encode more character tiles; one all-gather with uneven shards',
                   'page_owners': pb['k'], 'pages_per_rank': [len(x) for x in pb['pages']], 'char_tiles_per_rank': pb['char_counts'],
                   'predicted_ms_per_rank': pb['predicted_ms'], 'predicted_even_plan_ms': pb['predicted_even_ms'],
                   'steps': args.strong_steps, 'ms_per_step': round(bal_ms, 2), 'value': round(wbal['n_pages'] / (bal_ms * 1e-3), 4), 'unit': 'pages/s'}
            del ins_b
        strong = {'what': 'BASELINE config 4 as written: the pages of a step are divided over the ranks (strong scaling), one batch at a time, in the same process group '
                          'as the weak-scaling line above',
                  'scaling': 'strong', 'pages_per_step': ws['n_pages'], 'pages_per_gpu': ws['pages_per_gpu'], 'char_tiles_this_rank': ws['ct_hi'] - ws['ct_lo'],
                  'steps': args.strong_steps, 'value': round(ws['n_pages'] / per_step, 4), 'unit': 'pages/s', 'ms_per_step': round(per_step * 1e3, 2),
                  'phases_ms': {'visual_incl_all_gather': round(vis_s * 1e3, 1), 'splice_prefill_first_token': round(pre_s * 1e3, 1),
                                'decode_remaining_tokens': round(max(per_step - vis_s - pre_s, 0.0) * 1e3, 1),
                                'how': 'MAX over ranks of one extra stamped pass that stops after the first token; decode = timed step - those two'},
                  'all_gather': g2,
                  'two_steps_one_decode': None if merged_ms is None else {
                      'what': 'two consecutive steps with ONE decode over both steps\' pages per rank (the weights are streamed once per two steps\' pages): a throughput arrangement',
                      'ms_per_step': round(merged_ms, 2), 'value': round(ws['n_pages'] / (merged_ms * 1e-3), 4), 'unit': 'pages/s'},
                  'balanced': bal,
                  'n1_denominator': ((f'the N = 1 line of `python bench.py --gpus 1 --pages {ws["n_pages"]}` is this configuration on one GPU'
                                      + (' (= the default N = 1 line)' if ws['n_pages'] == 64 and args.pages == 64 else '')
                                      + ': speed-up = this value / that value; none is printed here because this run did not measure N = 1'))}
        del ins

    return strong


def single_gpu_extras(S, result, strong_share):
    """Adds vit_config2, config3_single_image, preprocess_f1, ordering_f4, fp8_decode (--fp8-extras), fp8_mfma to `result` (rank 0, N = 1)."""
    args, world, rank, dev, dims, model, eng = S.args, S.world, S.rank, S.dev, S.dims, S.model, S.eng
    wl, P, S_page, n_pages, mine, ct_lo, ct_hi = S.wl, S.P, S.S_page, S.n_pages, S.mine, S.ct_lo, S.ct_hi
    page_px, char_px, ids, step, sync, make_inputs = S.page_px, S.char_px, S.ids, S.step, S.sync, S.make_inputs
    elapsed, seq_ms, seq_out = S.elapsed, S.seq_ms, S.seq_out
    NEW_TOKENS = plan.NEW_TOKENS
    from callireader_amd import synthetic
    value, ms_per_step = result['value'], result['ms_per_step']
    if not args.no_vit_extra:
        px32 = synthetic.make_pixels(32, seed=0, device=dev)
        eng.vit_forward(px32)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            eng.vit_forward(px32)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        result['vit_config2'] = {'workload': 'InternViT-300M encoder only, 32 tiles 448x448, bf16', 'tiles_per_s': round(32 / dt, 1),
                                 'ms': round(dt * 1e3, 2), 'tflops': round(32 * 723.6e9 / dt / 1e12, 1),
                                 'mfma_frac': round(32 * 723.6e9 / dt / 1e12 / PEAK_BF16_TFLOPS, 4)}
    if not args.no_vit_extra:
        # BASELINE config 3: one image through the whole path on one GPU (latency view: batch of one page)
        one_page, one_char = page_px[:PAGE_TILES], char_px[:CHAR_TILES]

        def single():
            v = model.extract_feature(one_page)
            r, _ = model.align_tiles(one_char)
            e = eng.embed_splice(ids[0], v, r.reshape(-1, 3, dims.llm_hidden), img_id=IMG_CONTEXT_TOKEN_ID, ref_id=ALIGNED_TOKEN_ID)
            return model.generate_pages([e], max_new_tokens=NEW_TOKENS, eos_token_id=None)
        single(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        single(); torch.cuda.synchronize()
        dt1 = time.perf_counter() - t0
        result['config3_single_image'] = {'workload': 'one page (107 tiles, 3164-token prompt, 128 greedy tokens), batch of one', 's_per_page': round(dt1, 4)}
        # tile preprocessing of one example-shaped page (788x2000, 11 page tiles + 96 character crops)
        import numpy as np
        from PIL import Image
        from callireader_amd import preprocess
        rng = np.random.default_rng(0)
        page = rng.integers(0, 256, (2000, 788, 3), dtype=np.uint8)
        boxes = [(40 + 180 * (i % 4), 30 + 80 * (i // 4), 40 + 180 * (i % 4) + 100 + (i % 5) * 12, 30 + 80 * (i // 4) + 70) for i in range(96)]
        jobs, n = preprocess.plan_page(788, 2000)
        jobs += [preprocess.plan_char(b, n + i) for i, b in enumerate(boxes)]
        page_h = torch.from_numpy(page)
        page_d = page_h.to(dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            page_d = page_h.to(dev)                 # the only host buffer a page needs: 4.7 MB of pixels over PCIe
        torch.cuda.synchronize()
        h2d_ms = (time.perf_counter() - t0) / 10 * 1e3
        eng.preprocess(page_d, jobs, n + 96)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            eng.preprocess(page_d, jobs, n + 96)
        torch.cuda.synchronize()
        gpu_ms = (time.perf_counter() - t0) / 10 * 1e3
        t0 = time.perf_counter()
        pil = Image.fromarray(page)
        preprocess.load_image(pil)
        for (x1, y1, x2, y2) in boxes:
            preprocess.load_image_2(Image.fromarray(page[y1:y2, x1:x2]))
        cpu_ms = (time.perf_counter() - t0) * 1e3
        result['preprocess_f1'] = {'workload': '788x2000 page -> 11 page tiles + 96 character tiles (bf16, normalised)', 'gpu_ms_per_page': round(gpu_ms, 3),
                                   'host_pil_ms_per_page': round(cpu_ms, 1), 'h2d_ms_per_page': round(h2d_ms, 3),
                                   'pcie_inclusive_pages_per_s': round(1.0 / (1.0 / value + (h2d_ms + gpu_ms) * 1e-3), 4) if world == 1 else None,
                                   'parity': 'bit-exact (tests/test_gpu_prep.py)'}
        # the OrderFormer scorer of the ordering front end, 64 pages x 50 boxes per call
        from callireader_amd import synthetic as syn
        from oracle import orderformer as oracle_of
        sd_of = syn.make_orderformer_state_dict(seed=11)
        model.load_orderformer(sd_of)
        xb = torch.rand(64, 50, 4, generator=torch.Generator().manual_seed(3)).to(torch.bfloat16)
        xd = xb.to(dev)
        eng.orderformer(xd)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            eng.orderformer(xd)
        torch.cuda.synchronize()
        of_gpu_ms = (time.perf_counter() - t0) / 10 * 1e3
        cpu_of = oracle_of.CpuScorer(sd_of)
        t0 = time.perf_counter()
        cpu_of.orderformer(xb[:8])
        of_cpu_ms = (time.perf_counter() - t0) * 1e3 / 8
        result['ordering_f4'] = {'workload': 'OrderFormer (4 layers, d 256, 8 heads) on 64 pages x 50 boxes, bf16',
                                 'gpu_ms_per_page': round(of_gpu_ms / 64, 4), 'cpu_oracle_ms_per_page': round(of_cpu_ms, 2),
                                 'parity': 'scores within 4 % of the oracle model, reading order = the reference on 5 pages (tests/test_gpu_ordering.py)'}
    if not args.no_vit_extra:
        # BASELINE config 5's option, as an EXTRA (the headline above is bf16, the reference's arithmetic): batched decode
        # on e4m3 copies of the LLM's linear weights.  Same pages, same prompts; 32 decode steps each way.
        pseudo_all, _ = model.align_tiles(char_px)
        vit_mine = model.extract_feature(page_px)
        pr = pseudo_all.reshape(-1, 3, dims.llm_hidden)
        embeds = [eng.embed_splice(ids[j], vit_mine[j * PAGE_TILES:(j + 1) * PAGE_TILES], pr[p * CHAR_TILES:(p + 1) * CHAR_TILES],
                                   img_id=IMG_CONTEXT_TOKEN_ID, ref_id=ALIGNED_TOKEN_ID) for j, p in enumerate(mine)]
        del pseudo_all, vit_mine

        def decode_run(n_steps=32):
            kv = model.kv()
            kv.reset()
            for i0 in range(0, len(embeds), 16):
                idx = list(range(i0, min(len(embeds), i0 + 16)))
                eng.prefill_batch(kv, idx, [embeds[i] for i in idx])
            live = list(range(len(embeds)))
            first = eng.decode(kv, live, want_logits=True).float()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n_steps):
                eng.decode(kv, live)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / n_steps
            return dt, first, [kv.generated(i)[:n_steps + 2] for i in live]
        if args.fp8_extras:
          dt16, lg16, ids16 = decode_run()
          eng.enable_fp8_decode(True)
          decode_run(4)
          dt8, lg8, ids8 = decode_run()
          eng.enable_fp8_decode(False)
          same = sum(a == b for x, y in zip(ids16, ids8) for a, b in zip(x[:2], y[:2]))
          result['fp8_decode'] = {'what': 'batched greedy decode with e4m3 copies of the LLM linear weights (one fp32 scale per output row, dequantised '
                                        'in registers, same bf16 MFMA, fp32 accumulation; activations / KV cache / prefill / vision stay bf16) '
                                        'next to the bf16 path on the same pages: an option, not the headline',
                                'pages': len(embeds), 'bf16_ms_per_step': round(dt16 * 1e3, 3), 'fp8_ms_per_step': round(dt8 * 1e3, 3),
                                'speedup': round(dt16 / dt8, 3),
                                'first_step_logits_rel_l2_vs_bf16': round(float((lg8 - lg16).double().norm() / lg16.double().norm()), 4),
                                'first_two_picks_equal': f'{same}/{2 * len(embeds)}',
                                'note': 'random-init weights: every linear adds ~3.6 % of independent relative noise (tests/test_gpu_fp8.py); '
                                        'accuracy on real weights is what evaluate.py --type full_page measures (needs the checkpoint and CalliBench)'}
        del embeds
        # fp8 on the matrix cores, also an EXTRA: the same step with the norm-fed / quantised linears of the ViT, the projector
        # and the LLM prefill in e4m3 x e4m3 (v_mfma_f32_16x16x128_f8f6f4)
        eng.enable_fp8_mfma(True, level=1)
        step(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        step(); torch.cuda.synchronize()
        dt_step8_l1 = time.perf_counter() - t0             # level 1 alone: norm-fed linears only, decode in bf16
        eng.enable_fp8_mfma(True, level=2)
        # round 5: the e4m3 copies have their decode layout and the stream kernel an e4m3 form, so the e4m3-weight decode beats the bf16 one again at every
        # row count (64 rows 7.97 against 8.29 ms, 8 rows 3.04 against 3.93; profiles/round5/11_*, 14_*): the fp8 step decodes on it
        eng.enable_fp8_decode(True)
        step(); torch.cuda.synchronize()
        st8 = [0.0]
        torch.cuda.synchronize(); st8[0] = time.perf_counter()
        step(new_tokens=1, stamps=st8)
        t0 = time.perf_counter()
        out_step8 = step(); torch.cuda.synchronize()
        dt_step8 = time.perf_counter() - t0
        # BASELINE config 5 is config 4 with the fp8 weight path: the same rank-0 share as `strong_share`, both sides with the fp8 options on
        share8 = None
        if strong_share is not None:
            w8s = plan_strong_share(args.pages, args.share_world)
            ins8 = (page_px[:w8s['pages_per_gpu'] * PAGE_TILES], char_px[:w8s['ct_hi']], ids[:w8s['pages_per_gpu']])
            step(w=w8s, inputs=ins8); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.share_steps):
                step(w=w8s, inputs=ins8)
            torch.cuda.synchronize()
            t_s8 = (time.perf_counter() - t0) / args.share_steps
            share8 = {'what': 'strong_share with the fp8 options on (level 2 + e4m3-weight decode) against the fp8 step above: config 5\'s per-rank share',
                      't_share_ms': round(t_s8 * 1e3, 2), 'full_step_ms': round(dt_step8 * 1e3, 1), f'projected_speedup_{args.share_world}': round(dt_step8 / t_s8, 3)}
            del ins8
            if not args.no_balanced:                     # ... and the balanced plan under the fp8 options' own stage costs
                from callireader_amd.parallel import MI355X_COST_FP8
                pb8 = plan_workload('strong', args.pages, args.pages, args.share_world, 0, plan='balanced', cost=MI355X_COST_FP8)['balanced']
                share8['balanced'] = measure_balanced(S, pb8, dt_step8 * 1e3, out_step8, None, w8s['pages_per_gpu'], cost_name='MI355X_COST_FP8')
        eng.enable_fp8_mfma(False)
        eng.enable_fp8_decode(False)
        result['fp8_mfma'] = {'what': 'one whole step (one batch at a time) with cr_enable_fp8_mfma level 2 and cr_enable_fp8_decode (e4m3 weight copies in their decode layout): ViT QKV / fc1 / fc2, mlp1[1] and all four LLM '
                                      'prefill linears multiply e4m3 x e4m3 (per-row activation scales from the norm kernels, from fc1\'s own epilogue under a '
                                      'LayerNorm-derived bound, or from a quantiser pass; per-row weight scales; fp32 accumulation); ViT proj, attention, '
                                      'resampler, VQ, KV cache stay bf16: an option, not the headline',
                              'pages_per_s': round(n_pages / dt_step8, 4), 'ms_per_step': round(dt_step8 * 1e3, 1),
                              'level1_only_pages_per_s': round(n_pages / dt_step8_l1, 4), 'strong_share': share8,
                              'accuracy': 'NOT parity-preserving on random-init weights (profiles/round3/full_depth_parity.json: fp8_mfma_full_depth; peaked_streams.json: fp8); the gate on a real checkpoint is evaluate.py --compare_fp8', 'speedup_vs_bf16_step_one_batch_at_a_time': round((seq_ms if seq_ms else ms_per_step) / (dt_step8 * 1e3), 3),
                              'visual_ms': round((st8[1] - st8[0]) * 1e3, 1), 'prefill_ms': round((st8[2] - st8[1]) * 1e3, 1),
                              'parity': 'tests/test_gpu_fp8_mfma.py: exact on e4m3-representable data; model-level difference to the bf16 path stated there'}
