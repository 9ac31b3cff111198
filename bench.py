#!/usr/bin/env python3
"""Benchmark of the CalliReader image->text hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W           (N > 1: launched by torch.distributed.run, one rank per GPU)

Metric (BASELINE.json): calligraphy pages/sec through ViT + resampler + LLM greedy decode.
One "step" = one batch of synthetic pages per GPU through the whole path:
  page shape = examples/0.jpg of the reference: 11 page tiles + 96 character tiles of 448x448 (107 ViT tiles),
  prompt of 3164 tokens (11*256 visual + 96*3 pseudo-tokens + 60 text), NEW_TOKENS greedy tokens
  (random weights never emit EOS; 128 ~ the 96-character transcription of that page);
  visual stage: this rank's even shard of the batch's character tiles (ViT -> projector -> resampler -> VQ -> de-norm)
  and the page tiles of the pages it owns -> (N > 1) all-gather of the character tiles' pseudo-token embeddings over
  RCCL -> embedding splice + batched prefill + batched decode of the pages this rank owns.
Inputs (pixels, token ids, weights) are resident in HBM before the timed region; weights are seeded random
(no checkpoint is available offline).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import platform
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC for RCCL; must be set before HIP initialises



def _self_launch(argv):
    """`python bench.py --gpus N` run directly (no torch.distributed.run around it): start N ranks ourselves.  This
    process has not touched HIP yet (only the standard library is imported above), so it can spawn children freely; it
    never replaces itself by exec, it waits for the launcher and leaves with its exit code."""
    import socket
    import subprocess
    n = 1
    for i, a in enumerate(argv):
        if a == '--gpus' and i + 1 < len(argv):
            n = int(argv[i + 1])
        elif a.startswith('--gpus='):
            n = int(a.split('=', 1)[1])
    if n <= 1 or 'WORLD_SIZE' in os.environ or 'RANK' in os.environ:
        return
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + list(argv)
    sys.exit(subprocess.call(cmd))


if __name__ == '__main__':
    _self_launch(sys.argv[1:])

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
from benchlib import plan  # noqa: E402
from benchlib.plan import PAGE_TILES, CHAR_TILES, TEXT_TOKENS, build_ids, plan_workload, plan_strong_share  # noqa: E402,F401  (tests import them from here)
from benchlib.measure import PEAK_BF16_TFLOPS, PEAK_HBM_GBS, cpu_baseline, cpu_baseline_full, measure_traffic  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=2)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--pages', type=int, default=64, help='pages per GPU per step (all pages of a step decode as one batch; 64 = the most rows the decode kernels take)')
    ap.add_argument('--new-tokens', type=int, default=128)
    ap.add_argument('--scaling', choices=('weak', 'strong'), default='weak',
                    help='weak: --pages per GPU per step whatever N; strong: --total-pages per step over all GPUs (BASELINE config 4 as written: 64 pages over 8 GPUs = 8 per GPU)')
    ap.add_argument('--total-pages', type=int, default=64, help='pages per step over all ranks with --scaling strong')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-traffic', action='store_true', help='skip the rocprofv3 --pmc child runs behind roofline.traffic (N = 1 only; the committed measurement is quoted instead)')
    ap.add_argument('--cpu-baseline', choices=('sample', 'full'), default='sample',
                    help='sample: bounded sample extrapolated to one page (default, ~15 s); full: the oracle on ONE WHOLE example-shaped page (107 tiles, 24 + 4 + 32 layers, '
                         '3164-token prefill, 8 decode steps; minutes of host time and ~25 GB of host memory), printed next to the extrapolation')
    ap.add_argument('--no-vit-extra', action='store_true')
    ap.add_argument('--no-strong-block', action='store_true',
                    help='N > 1 with weak scaling: do not append the strong-scaling block (BASELINE config 4 as written: --total-pages per step over all ranks) to the line')
    ap.add_argument('--strong-steps', type=int, default=2, help='timed steps of that block')
    ap.add_argument('--balanced-owners', type=int, default=None, help='strong_share.balanced: fix the number of page owners instead of taking the cost model\'s (a sweep shows where its optimum lies)')
    ap.add_argument('--no-balanced', action='store_true', help='do not run the balanced strong-scaling plan (strong_scaling.balanced at N > 1, strong_share.balanced at N = 1)')
    ap.add_argument('--no-ragged', action='store_true', help='N = 1: skip strong_share.balanced_ragged (the balanced plan on a ragged batch under measured stage costs)')
    ap.add_argument('--two-steps-one-decode', action='store_true', help='also time the two-steps-one-decode arrangement of strong_share / strong_scaling (off by default since round 6)')
    ap.add_argument('--no-strong-share', action='store_true', help='N = 1: do not time one rank\'s share of the strong-scaling step (strong_share)')
    ap.add_argument('--share-world', type=int, default=8, help='the world size whose rank-0 share strong_share runs on this one GPU')
    ap.add_argument('--share-steps', type=int, default=2, help='timed steps of strong_share')
    ap.add_argument('--fp8-extras', action='store_true', help='also time the batched decode on e4m3 weight copies alone (fp8_decode) and compare its first picks with the bf16 decode')
    ap.add_argument('--no-api', action='store_true', help='N = 1: skip the api_level block (the same batch from JPEG files through chat_ocr_stream / folder_rec to strings)')
    ap.add_argument('--api-batches', type=int, default=6, help='batches of --pages pages the api_level block streams')
    ap.add_argument('--no-pipeline', action='store_true', help='one batch at a time (the decode of a batch does not run beside the visual stage of the next)')
    args = ap.parse_args()
    plan.NEW_TOKENS = NEW_TOKENS = args.new_tokens

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: run `python bench.py --gpus N` (it starts the N ranks itself) '
                         'or launch N ranks with torch.distributed.run')
    backend = os.environ.get('CR_DIST_BACKEND', 'nccl') if world > 1 else 'none'     # 'gloo' lets several ranks share one GPU on a test box
    n_dev = torch.cuda.device_count()                            # counting devices does not initialise HIP
    if n_dev < 1:
        raise SystemExit('bench.py needs a GPU: there is no CPU fallback for the hot path')
    if backend == 'nccl' and world > n_dev:
        raise SystemExit(f'{world} ranks but {n_dev} visible GPU(s): RCCL needs one GPU per rank (CR_DIST_BACKEND=gloo shares a GPU on a test box)')
    if world > 2 * n_dev:
        # plumbing runs with many ranks on one GPU (gloo): every rank would keep the decode-layout copy of the LLM (+15.9 GB each) beside its model
        os.environ.setdefault('CR_DECODE_LAYOUT', '0')
    local_rank = local_rank % n_dev
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if world > 1:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(backend)
        assert dist.get_world_size() == world

    from callireader_amd.config import ModelDims, IMG_CONTEXT_TOKEN_ID, ALIGNED_TOKEN_ID
    from callireader_amd import synthetic, _binding as B
    from callireader_amd.modeling_internvl_chat import InternVLChatModel
    from callireader_amd.parallel import shard_range, all_gather_rows_async, owned_pages

    dims = ModelDims.full()
    wl = plan_workload(args.scaling, args.pages, args.total_pages, world, rank)
    n_pages, mine, ct_lo, ct_hi = wl['n_pages'], wl['mine'], wl['ct_lo'], wl['ct_hi']
    P = wl['pages_per_gpu']                              # pages this rank owns per step (round-robin; = --pages with weak scaling)
    S_page = PAGE_TILES * 256 + CHAR_TILES * 3 + TEXT_TOKENS
    # (room for the ragged batch of strong_share.balanced_ragged: pages of up to 13 page tiles and 250 character tiles; a longer cache changes no step's work)
    longest_prompt = max(S_page, 13 * 256 + 250 * 3 + TEXT_TOKENS) if (world == 1 and not args.no_strong_share and not args.no_ragged) else S_page
    model = InternVLChatModel.from_synthetic(dims, seed=0, device=local_rank, max_tokens=longest_prompt + NEW_TOKENS + 128, max_pages=P)
    model.img_context_token_id = IMG_CONTEXT_TOKEN_ID
    eng = model.engine

    # ---- synthetic inputs, resident in HBM ----
    # character tiles (90 % of the visual work) are sharded evenly over all ranks whatever page they belong to;
    # page tiles stay with the page's owner: their embeddings are 2.1 MB per tile and nobody else needs them
    def make_inputs(w):
        def px(n, seed):                                   # (a rank of the balanced plan may own no page, or encode no character tile)
            return synthetic.make_pixels(n, seed=seed, device=dev) if n else torch.empty((0, 3, 448, 448), dtype=torch.bfloat16, device=dev)
        return (px(len(w['mine']) * PAGE_TILES, 10 + rank), px(w['ct_hi'] - w['ct_lo'], 20 + rank),
                [build_ids(PAGE_TILES, CHAR_TILES, TEXT_TOKENS, IMG_CONTEXT_TOKEN_ID, ALIGNED_TOKEN_ID, 1000 + p).to(dev) for p in w['mine']])
    page_px, char_px, ids = make_inputs(wl)
    headline_inputs = (page_px, char_px, ids)

    def step(new_tokens=None, stamps=None, w=None, inputs=None, merged=None):
        """merged = [inputs, inputs, ...]: the visual stage (with its all-gather) and the splice of SEVERAL consecutive steps of workload `w`, then ONE prefill + decode over all
        their pages -- the decode batch of a rank fed from several steps (a page's ids do not depend on its batch); returns the pages of all of them, in order."""
        new_tokens = NEW_TOKENS if new_tokens is None else new_tokens
        embeds = []
        for one in (merged if merged is not None else [inputs]):
            if w is not None:                             # another workload than the headline's (the strong-scaling block)
                n_pages, mine = w['n_pages'], w['mine']
                page_px, char_px, ids = one
            else:
                n_pages, mine, page_px, char_px, ids = wl['n_pages'], wl['mine'], *headline_inputs
            if char_px.shape[0]:
                pseudo_local, _ = model.align_tiles(char_px)                             # (3 * my char-tile shard, 4096)
            else:                                                                        # balanced plan: a rank whose pages already fill its step
                pseudo_local = torch.empty((0, dims.llm_hidden), dtype=torch.bfloat16, device=dev)
            if w is not None and w.get('pseudo_all') is not None:                        # one-GPU stand-in for the gather of a share that does not encode all of its
                gathered = (lambda t: (lambda: t))(w['pseudo_all'])                      # pages' character tiles itself (strong_share.balanced): handed in, made earlier
            else:
                gathered = all_gather_rows_async(pseudo_local.reshape(-1, 3, dims.llm_hidden), n_pages * CHAR_TILES,      # 24.5 KB per tile, over xGMI ...
                                                 counts=None if w is None else w.get('ct_counts'))
            vit_mine = model.extract_feature(page_px) if len(mine) else None             # ... underneath the owner's page tiles (my pages * 11, 256, 4096)
            if stamps is not None:
                torch.cuda.synchronize(); stamps.append(time.perf_counter())
            pseudo_all = gathered()
            for j, p in enumerate(mine):
                v = vit_mine[j * PAGE_TILES:(j + 1) * PAGE_TILES]
                r = pseudo_all[p * CHAR_TILES:(p + 1) * CHAR_TILES]
                embeds.append(eng.embed_splice(ids[j], v, r, img_id=IMG_CONTEXT_TOKEN_ID, ref_id=ALIGNED_TOKEN_ID))
        outs = model.generate_pages(embeds, max_new_tokens=new_tokens, eos_token_id=None) if embeds else []      # (no pages: a tile-only rank of the balanced plan)
        assert all(len(o) == new_tokens for o in outs)
        if stamps is not None:
            torch.cuda.synchronize(); stamps.append(time.perf_counter())
        return outs

    # The same step with two batches in flight (PagePipeline): batch i's visual stage and prefill run here while a worker thread
    # decodes batch i-1 on its own stream through a context that shares the weights.  A run of K steps ends with the last
    # batch's decode alone (finish()), inside the timed region.
    pipe = None
    if not args.no_pipeline:
        try:
            pipe = model.page_pipeline(max_new_tokens=NEW_TOKENS, eos_token_id=None)
        except Exception as e:                      # e.g. no memory for the second KV cache: one batch at a time, and the line says so
            print(f'[bench] page pipeline unavailable ({e}); running one batch at a time', file=sys.stderr)
    marks = []

    def mark(tag):
        if os.environ.get('CR_PIPE_MARKS'):
            e = torch.cuda.Event(enable_timing=True)
            e.record(torch.cuda.current_stream())
            marks.append((tag, e, time.perf_counter()))

    def step_pipelined():
        mark('step start')
        pseudo_local, _ = model.align_tiles(char_px)
        gathered = all_gather_rows_async(pseudo_local.reshape(-1, 3, dims.llm_hidden), n_pages * CHAR_TILES)
        vit_mine = model.extract_feature(page_px)
        mark('visual done')
        pseudo_all = gathered()
        embeds = []
        for j, p in enumerate(mine):
            v = vit_mine[j * PAGE_TILES:(j + 1) * PAGE_TILES]
            r = pseudo_all[p * CHAR_TILES:(p + 1) * CHAR_TILES]
            embeds.append(eng.embed_splice(ids[j], v, r, img_id=IMG_CONTEXT_TOKEN_ID, ref_id=ALIGNED_TOKEN_ID))
        out = pipe.start(embeds)
        mark('prefill issued + previous collected')
        return out

    def run_steps(k):
        if pipe is None:
            return [step() for _ in range(k)]
        outs = [step_pipelined() for _ in range(k)]
        outs = outs[1:] + [pipe.finish()]
        assert all(o is not None and all(len(x) == NEW_TOKENS for x in o) for o in outs)
        return outs

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # everything below runs on a stream of its own, not on the legacy null stream (which synchronises with other streams' work)
    main_stream = torch.cuda.Stream(device=dev, priority=int(os.environ.get('CR_MAIN_PRIO', '0')))
    torch.cuda.set_stream(main_stream)
    if args.warmup:
        run_steps(args.warmup)
    sync()
    prof_ctx = [eng._h] + ([pipe.dec._h] if pipe is not None else [])      # the decode thread launches through its own context
    for h in prof_ctx:
        B.check(B.lib.cr_profile(h, 1))
    t0 = time.perf_counter()
    run_steps(args.steps)
    sync()
    elapsed = time.perf_counter() - t0
    if marks and rank == 0:
        base_e, base_t = marks[0][1], marks[0][2]
        for tag, e, t in marks:
            print(f'[marks] {tag:40s} gpu {base_e.elapsed_time(e):9.1f} ms   host {1e3 * (t - base_t):9.1f} ms', file=sys.stderr)
    import ctypes as C
    prof, pstat = [0.0] * 8, [0] * 4
    for h in prof_ctx:
        B.check(B.lib.cr_profile(h, 0))
        pr, ps = (C.c_double * 8)(), (C.c_int64 * 4)()
        B.check(B.lib.cr_profile_read(h, pr))
        B.check(B.lib.cr_profile_stats(h, ps))
        prof = [a + b for a, b in zip(prof, pr)]
        pstat = [pstat[0] + ps[0], pstat[1] + ps[1], pstat[2] + ps[2], max(pstat[3], ps[3])]
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # the path's one collective, alone on the wire (untimed extra): the pseudo-token all-gather of one step
    def gather_standalone(w):
        rows = torch.zeros(w['ct_hi'] - w['ct_lo'], 3, dims.llm_hidden, device=dev, dtype=torch.bfloat16)
        all_gather_rows_async(rows, w['n_pages'] * CHAR_TILES)()
        sync()
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            all_gather_rows_async(rows, w['n_pages'] * CHAR_TILES)()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        sync()
        ms = sorted(ts)[2] * 1e3
        total_bytes = w['n_pages'] * CHAR_TILES * 3 * dims.llm_hidden * 2
        return {'collective': 'all_gather_into_tensor (pseudo-token embeddings of the character tiles)', 'backend': dist.get_backend(),
                'bytes_per_rank_sent': (w['ct_hi'] - w['ct_lo']) * 3 * dims.llm_hidden * 2, 'bytes_gathered': total_bytes,
                'standalone_ms': round(ms, 3), 'gb_per_s_received': round(total_bytes * (world - 1) / world / (ms * 1e-3) / 1e9, 1),
                'note': 'inside a step the gather runs underneath the page tiles\' ViT (all_gather_rows_async)'}
    gather = gather_standalone(wl) if world > 1 else None

    # one un-pipelined step for comparison (untimed extra) and a self-check: the pipelined run's ids are the sequential step's
    seq_ms, same_ids, seq_frac, seq_dec, seq_out = None, None, None, None, None
    if pipe is not None:
        last_pipe = run_steps(1)[-1]
        if os.environ.get('CR_PIPE_MARKS') and pipe.host_steps:
            print(f'[marks] decode thread: {1e3 * pipe.host_s / pipe.host_steps:.2f} ms of host time per decode step issued ({pipe.host_steps} steps)', file=sys.stderr)
        pipe.close()                                # the decode thread, its context and both KV caches go before anything else is measured
        step()                                      # (the one-batch path allocates its own KV cache the first time)
        sync()
        B.check(B.lib.cr_profile(eng._h, 1))
        t0s = time.perf_counter()
        seq_out = step()
        sync()
        seq_ms = (time.perf_counter() - t0s) * 1e3
        B.check(B.lib.cr_profile(eng._h, 0))
        pr = (C.c_double * 8)()
        B.check(B.lib.cr_profile_read(eng._h, pr))
        seq_frac = (pr[2] / (pr[1] * 1e-3) / 1e12 / PEAK_BF16_TFLOPS) if pr[1] > 0 else None      # the same kernels with the chip to themselves
        seq_dec = (pr[7] / (pr[5] * 1e-3) / 1e9, pr[5]) if pr[5] > 0 else None                     # weight-streaming class, GB/s and ms of that un-overlapped step
        same_ids = bool(seq_out == last_pipe)

    # north_star's "MFMA utilisation on ViT + LLM prefill": one extra, untimed pass that stops after the first token
    # (visual stage, then splice + prefill + the first LM-head row), algorithmic FLOPs of SURVEY 8(d) over its wall time
    st = [0.0]
    torch.cuda.synchronize(); st[0] = time.perf_counter()
    step(new_tokens=1, stamps=st)
    my_tiles = len(mine) * PAGE_TILES + (ct_hi - ct_lo)
    vis_fl = my_tiles * (723.6e9 + 17.18e9) + (ct_hi - ct_lo) * (12.02e9 + 2.27e9)
    pre_fl = len(mine) * (S_page * 13.96e9 + 0.524e6 * S_page * (S_page + 1) / 2 + 0.758e9)
    vit_prefill = {'what': 'visual stage (ViT + mlp1 + resampler + VQ) and LLM prefill of one batch, attention and norms included; algorithmic FLOPs of SURVEY 8(d) / wall time of an untimed extra pass on rank 0',
                   'visual_ms': round((st[1] - st[0]) * 1e3, 1), 'prefill_ms': round((st[2] - st[1]) * 1e3, 1),
                   'tflops': round((vis_fl + pre_fl) / (st[2] - st[0]) / 1e12, 1),
                   'mfma_frac': round((vis_fl + pre_fl) / (st[2] - st[0]) / 1e12 / PEAK_BF16_TFLOPS, 4)}

    # ---- the optional, untimed blocks (benchlib/extras.py) ----
    import types
    from benchlib import extras
    S = types.SimpleNamespace(args=args, world=world, rank=rank, dev=dev, dims=dims, model=model, eng=eng, wl=wl, P=P, S_page=S_page, n_pages=n_pages, mine=mine,
                              ct_lo=ct_lo, ct_hi=ct_hi, page_px=page_px, char_px=char_px, ids=ids, step=step, sync=sync, make_inputs=make_inputs,
                              elapsed=elapsed, seq_ms=seq_ms, seq_out=seq_out)
    strong_share = None
    if world == 1 and rank == 0 and args.scaling == 'weak' and not args.no_strong_share and args.pages >= args.share_world and args.pages % args.share_world == 0:
        strong_share = extras.strong_share_block(S)
    strong = None
    if world > 1 and args.scaling == 'weak' and not args.no_strong_block:
        strong = extras.strong_scaling_block(S, gather_standalone)

    result = None
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = n_pages / (elapsed / args.steps)
        big_n, big_ms, big_fl, big_by = prof[0], prof[1], prof[2], prof[3]
        traffic, traffic_src = None, None
        for rd in ('round3', 'round2', 'round1'):
            tp = os.path.join(ROOT, 'profiles', rd, 'traffic_pmc.json')
            if os.path.exists(tp):
                traffic = json.load(open(tp)).get('gemm_tiled_big', {}).get('traffic_bytes_per_launch')
                traffic_src = f'profiles/{rd}/traffic_pmc.json'
                break
        sm_n, sm_ms, sm_fl, sm_by = prof[4], prof[5], prof[6], prof[7]
        achieved = big_fl / (big_ms * 1e-3) / 1e12 if big_ms > 0 else 0.0
        result = {
            'metric': 'calligraphy pages/sec (ViT+resampler+LLM greedy)', 'value': round(value, 4), 'unit': 'pages/s',
            'n_gpus': dist.get_world_size() if world > 1 else 1, 'backend': dist.get_backend() if world > 1 else None, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(ms_per_step, 2),
            'higher_is_better': True, 'scaling': args.scaling, 'vs_baseline': None, 'dtype': 'bf16', 'data': 'synthetic',
            'config': {'workload': (f'{args.scaling} scaling: ' + (f'{n_pages} pages/step over {world} GPU(s) = ' if args.scaling == 'strong' else '')) + f'full page path, {P} pages/GPU/step of the examples/0.jpg shape (11 page + 96 char tiles 448x448, '
                                   f'{S_page}-token prompt, {NEW_TOKENS} greedy tokens, repetition_penalty 1.0); InternVL2-8B shapes '
                                   '(InternViT-300M 24L + mlp1 + PerceiverResampler 4L + 92553-row cosine VQ + InternLM2.5-7B 32L), random-init bf16 weights',
                       'scaling': args.scaling, 'pages_per_step': n_pages, 'pages_per_gpu': P, 'tiles_per_page': PAGE_TILES + CHAR_TILES, 'prompt_tokens': S_page, 'new_tokens': NEW_TOKENS,
                       'parallelism': f'character tiles sharded over ranks + RCCL all-gather of their pseudo-token embeddings, page tiles and LLM per page owner (round-robin), dp{world}',
                       'decode_weight_layout': 'nn.Linear rows (CR_DECODE_LAYOUT=0)' if os.environ.get('CR_DECODE_LAYOUT') == '0' else
                                               'second, tile-contiguous copy of every LLM linear for the weight-streaming decode kernels (+15.9 GB per GPU, bit-identical results)'},
            'roofline': {'bound': 'mfma', 'kernel': 'tiled bf16 MFMA GEMM (gemm256_kernel, persistent 256x256, slot-staggered wave groups; gemm128_kernel where it schedules better), launches with M >= 1024: ViT, projector, resampler to_kv, VQ, LLM prefill',
                         'achieved': round(achieved, 1), 'peak': PEAK_BF16_TFLOPS, 'unit': 'TFLOP/s',
                         'frac': round(achieved / PEAK_BF16_TFLOPS, 4),
                         'frac_one_batch_at_a_time': round(seq_frac, 4) if seq_frac else None,      # the same launches with the chip to themselves (see `pipeline`)
                         'traffic': traffic,
                         'traffic_note': 'bytes per launch on the L2 fabric side (Infinity-Cache hits included), (2*FETCH_SIZE + WRITE_SIZE)*1024 from separate rocprofv3 --pmc passes of this bench: ' + str(traffic_src),
                         'algorithmic_bytes_per_launch': round(big_by / max(big_n, 1), 1),
                         'launches': int(big_n), 'avg_launch_ms': round(big_ms / max(big_n, 1), 4),
                         'flops_per_launch': round(big_fl / max(big_n, 1), 1),
                         'how': 'HIP events around every launch on the launch stream during the timed steps (cr_profile)'},
            'vit_prefill': vit_prefill,
            'decode_gemm': {'bound': 'hbm', 'kernel': 'gemm_skinny_kernel (weight streaming, M <= 64: batched decode, LM head) and tiled launches with M < 1024 (resampler rows)',
                            'achieved': round(sm_by / (sm_ms * 1e-3) / 1e9, 1) if sm_ms > 0 else 0.0, 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
                            'frac': round(sm_by / (sm_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4) if sm_ms > 0 else 0.0,
                            'launches': int(sm_n), 'kernel_ms_per_step': round(sm_ms / args.steps, 2),
                            'achieved_one_batch_at_a_time': round(seq_dec[0], 1) if seq_dec else None,
                            'frac_one_batch_at_a_time': round(seq_dec[0] / PEAK_HBM_GBS, 4) if seq_dec else None,
                            'kernel_ms_one_batch_at_a_time': round(seq_dec[1], 2) if seq_dec else None,
                            'note': 'achieved / frac are measured live in the timed steps, where these launches share the chip with the other batch\'s matrix-bound stages '
                                    '(PagePipeline) and stretch; the *_one_batch_at_a_time fields are the same launches on one un-overlapped step -- that is the roofline reading'},
            'gemm_big_ms_per_step': round(big_ms / args.steps, 2),
            'prof_truncated': bool(pstat[2] != 0 or pstat[0] != pstat[1] or int(big_n + sm_n) != pstat[0]),
            'prof': {'launches_bracketed': int(pstat[0]), 'accounted': int(pstat[1]), 'lost': int(pstat[2]), 'peak_pending': int(pstat[3])},
            'all_gather': gather,
            'strong_scaling': strong,
            'strong_share': strong_share,
            'pipeline': None if pipe is None else {
                'what': 'two batches in flight (PagePipeline): a worker thread runs the HBM-bound batched decode of batch i-1 on a second HIP stream '
                        '(second context sharing the weights) beside the matrix-bound visual stage and prefill of batch i; a run of K steps ends '
                        'with the last decode alone, inside the timed region',
                'one_batch_at_a_time_ms_per_step': round(seq_ms, 1), 'ids_equal_one_batch_at_a_time': same_ids,
                'roofline_frac_one_batch_at_a_time': round(seq_frac, 4) if seq_frac else None,
                'note': 'roofline.frac above is measured live in the timed steps, where the tiled GEMMs share the chip with the decode kernels of the '
                        'other batch (each launch takes longer, the step takes less); roofline_frac_one_batch_at_a_time is the same measurement on one '
                        'un-overlapped step'},
        }

    # ---- extras on rank 0 at N == 1: BASELINE config 2 (ViT only, 32 tiles) and the CPU baseline ----
    if rank == 0 and world == 1:
        if not args.no_api:
            # the headline's batch through the reference's API, from image files to strings (benchlib/api.py); untimed extra, same model object
            from benchlib.api import api_level
            try:
                result['api_level'] = api_level(model, ROOT, pages=P, batches=args.api_batches, new_tokens=NEW_TOKENS, folder_pages=2 * P,
                                                headline_ms_per_step=ms_per_step, headline_pages=n_pages)
                # the synthetic step again, right after the API run (same thermal state): does the box still run the one-batch-at-a-time step in the time it took before?
                step(); torch.cuda.synchronize()
                t0 = time.perf_counter()
                step(); torch.cuda.synchronize()
                result['api_level']['synthetic_step_one_batch_at_a_time_ms'] = {'before_the_api_run': round(seq_ms, 1) if seq_ms else None,
                                                                              'after_the_api_run': round((time.perf_counter() - t0) * 1e3, 1)}
            except Exception as e:                      # never at the cost of the line
                import traceback
                traceback.print_exc()
                result['api_level'] = {'error': f'{type(e).__name__}: {e}'}
        if not args.no_vit_extra:
            extras.single_gpu_extras(S, result, strong_share)
            # the kernels one at a time against their own rooflines (benchlib/kernels.py)
            from benchlib.kernels import by_kernel
            try:
                result['roofline']['by_kernel'] = by_kernel(dev)
            except Exception as e:
                result['roofline']['by_kernel'] = {'error': f'{type(e).__name__}: {e}'}
        if not args.no_traffic and args.pages >= 16 and args.scaling == 'weak':
            # HBM-side bytes per launch of the dominant kernel class, measured on THIS box in THIS run (untimed, after everything else)
            try:
                pipe = None
                S.model = S.eng = S.step = S.make_inputs = None
                del model, eng
                import gc
                gc.collect()
                torch.cuda.empty_cache()
            except NameError:
                pass
            tb, note = measure_traffic()
            if tb is not None:
                result['roofline']['traffic'] = tb
                result['roofline']['traffic_note'] = note
            else:
                result['roofline']['traffic_note'] += f' [live measurement unavailable: {note}]'
        if not args.no_cpu_baseline:
            try:
                del model
            except NameError:
                pass
            result['cpu_baseline'] = cpu_baseline()
            if args.cpu_baseline == 'full':
                result['cpu_baseline']['full_page'] = cpu_baseline_full(result['cpu_baseline']['cores'])
    if rank == 0:
        print(json.dumps(result, ensure_ascii=False), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
