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

PAGE_TILES, CHAR_TILES, TEXT_TOKENS = 11, 96, 60
PEAK_BF16_TFLOPS = 2500.0      # dense bf16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"
PEAK_HBM_GBS = 8000.0          # HBM3E spec, same table


def build_ids(n_page_tiles, n_char_tiles, text_tokens, img_id, ref_id, seed):
    g = torch.Generator().manual_seed(seed)
    head = torch.randint(100, 60000, (text_tokens // 2,), generator=g)
    tail = torch.randint(100, 60000, (text_tokens - text_tokens // 2,), generator=g)
    return torch.cat([head, torch.full((n_page_tiles * 256,), img_id), torch.full((n_char_tiles * 3,), ref_id), tail])


def plan_workload(scaling, pages, total_pages, world, rank, plan='even', cost=None, owners=None):
    """Which pages and which character tiles one rank handles in a step.  weak: `pages` per GPU whatever N (n_pages = pages * world);
    strong: `total_pages` per step over all ranks (BASELINE config 4 as written: 64 pages over 8 GPUs).  plan 'even': pages are owned
    round-robin, the flat list of character tiles is split contiguously and evenly; plan 'balanced' (strong only): fewer ranks own pages
    and the others take more character tiles (callireader_amd/parallel.py: plan_balanced).  Pure host arithmetic."""
    from callireader_amd.parallel import shard_range, owned_pages, plan_balanced
    if scaling not in ('weak', 'strong') or plan not in ('even', 'balanced'):
        raise ValueError((scaling, plan))
    n_pages = total_pages if scaling == 'strong' else pages * world
    if plan == 'balanced':
        if scaling != 'strong':
            raise ValueError('the balanced plan is a strong-scaling plan')
        from callireader_amd.parallel import MI355X_COST
        pb = plan_balanced(n_pages, world, PAGE_TILES, CHAR_TILES, PAGE_TILES * 256 + CHAR_TILES * 3 + TEXT_TOKENS, NEW_TOKENS, cost=cost or MI355X_COST, owners=owners)
        lo, hi = pb['char_bounds'][rank]
        return {'scaling': scaling, 'plan': 'balanced', 'n_pages': n_pages, 'mine': pb['pages'][rank], 'ct_lo': lo, 'ct_hi': hi, 'ct_counts': pb['char_counts'],
                'pages_per_gpu': len(pb['pages'][rank]), 'balanced': pb}
    if n_pages < world:
        raise SystemExit(f'{n_pages} pages per step < {world} ranks: every rank needs a page')
    mine = owned_pages(n_pages, world, rank)
    ct_lo, ct_hi = shard_range(n_pages * CHAR_TILES, world, rank)
    return {'scaling': scaling, 'plan': 'even', 'n_pages': n_pages, 'mine': mine, 'ct_lo': ct_lo, 'ct_hi': ct_hi, 'pages_per_gpu': len(mine)}


def plan_strong_share(total_pages, share_world):
    """The workload `strong_share` runs on ONE GPU: as much as rank 0 of plan_workload('strong', total_pages over share_world ranks) has -- the same
    number of owned pages and of character tiles -- but laid on pages 0 .. n-1 of the one-GPU step and their own character tiles (a rank of the real
    run owns pages r, r + world, ... and an arbitrary eighth of the character tiles: the same work), so that the ids can be compared with the
    full step's.  Needs an even split (total_pages % share_world == 0)."""
    sw = plan_workload('strong', total_pages, total_pages, share_world, 0)
    n_own, n_ct = sw['pages_per_gpu'], sw['ct_hi'] - sw['ct_lo']
    if n_ct != n_own * CHAR_TILES:
        raise ValueError(f'{total_pages} pages over {share_world} ranks: uneven split')
    return {'scaling': 'strong', 'n_pages': n_own, 'mine': list(range(n_own)), 'ct_lo': 0, 'ct_hi': n_ct, 'pages_per_gpu': n_own}


# the one whole-page CPU measurement on record (profiles/round3/01_bench_N1_default_full_cpu_baseline.json, AMD EPYC 9575F, 64 threads): the
# sampled extrapolation of the same run said 165 s per page, the oracle measured stage by stage took 231 s
CPU_FULL_PAGE_MEASURED_S, CPU_FULL_PAGE_SAMPLED_S = 231.0, 165.0


def cpu_baseline():
    """The oracle (CPU restatement of the reference's eager path) on a bounded sample of the same page workload,
    extrapolated linearly to one page.  Reported next to the GPU number; it is not the target."""
    from callireader_amd.config import ModelDims
    from callireader_amd import synthetic
    from oracle import vision, calli_align, internlm2
    dims = ModelDims.full()
    t_all = time.time()
    with torch.no_grad():
        sd = synthetic.make_state_dict(dims, parts=('vit', 'mlp1'), seed=0)
        px = synthetic.make_pixels(2, seed=0)
        # eager PyTorch oversubscribes badly on many-core hosts: calibrate the thread count on one ViT layer
        ncpu = os.cpu_count() or 1
        best = (float('inf'), 1)
        for th in sorted({min(ncpu, t) for t in (8, 16, 32, 64, 128, ncpu)}):
            torch.set_num_threads(th)
            vision.vit_forward(sd, px[:1], 1)
            t0 = time.time(); vision.vit_forward(sd, px[:1], 1); dt = time.time() - t0
            best = min(best, (dt, th))
        torch.set_num_threads(best[1])
        cores = best[1]
        t0 = time.time(); feat = vision.extract_feature(sd, px, dims.vit_layers); t_vit = (time.time() - t0) / 2
        del sd
        rdims = ModelDims.reduced(rs_depth=1)
        sd = synthetic.make_state_dict(rdims, parts=('resampler',), seed=0)
        t0 = time.time(); calli_align.resampler_forward(sd, feat, 1); t_rs = (time.time() - t0) / 2 * dims.rs_depth
        del sd
        ldims = ModelDims.reduced(llm_layers=1, vocab=1024)
        sd = synthetic.make_state_dict(ldims, parts=('llm',), seed=0)
        S = 512
        emb = (torch.randn(1, S, 4096) * 0.02).bfloat16()
        rope = internlm2.rope_tables(128, seq_len=4096)
        t0 = time.time(); _, past = internlm2.model_forward(sd, 1, inputs_embeds=emb, rope=rope, all_logits=False)
        t_pre = (time.time() - t0) / S * dims.llm_layers                           # s per prompt token, 32 layers
        t0 = time.time()
        for _ in range(4):
            _, past = internlm2.model_forward(sd, 1, input_ids=torch.tensor([[5]]), past=past, rope=rope)
        t_dec = (time.time() - t0) / 4 * dims.llm_layers                           # s per new token, 32 layers
    S_page = PAGE_TILES * 256 + CHAR_TILES * 3 + TEXT_TOKENS
    page_s = t_vit * (PAGE_TILES + CHAR_TILES) + t_rs * CHAR_TILES + t_pre * S_page + t_dec * NEW_TOKENS
    ratio = CPU_FULL_PAGE_MEASURED_S / CPU_FULL_PAGE_SAMPLED_S
    return {'value': 1.0 / page_s, 'unit': 'pages/s', 'cores': cores, 'kind': 'port',
            'cpu': _cpu_model(), 'host_cores': os.cpu_count(),
            'measured_over_sampled': round(ratio, 2), 'value_calibrated': 1.0 / (page_s * ratio),
            'calibration': (f'the sample under-states a page: the one whole-page run on record (--cpu-baseline full, EPYC 9575F, 64 threads, profiles/round3/'
                            f'01_bench_N1_default_full_cpu_baseline.json) measured {CPU_FULL_PAGE_MEASURED_S:.0f} s per page where its own sample said '
                            f'{CPU_FULL_PAGE_SAMPLED_S:.0f} s; value_calibrated = value / {ratio:.2f}'),
            'sample': (f'oracle bf16 eager: ViT+mlp1 24 layers on 2 tiles ({t_vit:.2f} s/tile), resampler 1 of 4 layers on 2 tiles, '
                       f'InternLM2 1 of 32 layers prefill {S} tokens ({t_pre * 1e3:.1f} ms/token x32) + 4 decode steps '
                       f'({t_dec * 1e3:.0f} ms/token x32); extrapolated linearly to one page (107 tiles, {S_page} prompt tokens, '
                       f'{NEW_TOKENS} new tokens) = {page_s:.0f} s/page; sample wall {time.time() - t_all:.0f} s')}


def measure_traffic(pages=16):
    """roofline.traffic measured IN THIS RUN: two child runs of this script under `rocprofv3 --pmc` (FETCH_SIZE and WRITE_SIZE need
    separate passes: MI355X_MICROARCH.md, rocprofv3 PMC slots), at `pages` pages x 2 new tokens -- the same tiled-GEMM launch shapes
    as the 64-page step (255-tile ViT chunks, 16-page prefill batches) in a quarter of the dispatches (counter mode does not survive
    the full step).  bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 as that guide prescribes for gfx950; these counters sit on the L2's
    fabric side, Infinity-Cache hits included.  Returns (bytes per launch of the M >= 1024 tiled class, note) or (None, why)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    prof = shutil.which('rocprofv3') or '/opt/rocm/bin/rocprofv3'
    if not os.path.exists(prof):
        return None, 'rocprofv3 not found'
    args = ['--steps', '1', '--warmup', '0', '--pages', str(pages), '--new-tokens', '2', '--no-cpu-baseline', '--no-vit-extra', '--no-pipeline', '--no-traffic', '--no-strong-share', '--no-api']
    work = tempfile.mkdtemp(prefix='cr_pmc_', dir='/tmp')
    env = dict(os.environ, TMPDIR='/tmp')
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'LOCAL_WORLD_SIZE', 'GROUP_RANK', 'MASTER_ADDR', 'MASTER_PORT', 'TORCHELASTIC_RUN_ID'):
        env.pop(k, None)                                      # the children are one-process runs whatever launched this one
    sums = {}
    try:
        for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
            d = os.path.join(work, counter)
            r = subprocess.run([prof, '--pmc', counter, '--output-format', 'csv', '-d', d, '--', sys.executable, os.path.abspath(__file__)] + args,
                               cwd='/tmp', env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=240)
            files = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)
            if r.returncode != 0 or not files:
                return None, f'rocprofv3 --pmc {counter} failed (rc {r.returncode}): ' + r.stdout.decode(errors='replace')[-300:]
            vals = []
            for row in csv.DictReader(open(files[0])):
                k = row['Kernel_Name']
                if row['Counter_Name'] != counter or not ('gemm256_kernel' in k or 'gemm128_kernel' in k):
                    continue
                if 'gemm128' in k and int(row['Grid_Size']) < 8 * 8 * 256:
                    continue                                  # M < 1024: not in the roofline class
                vals.append(float(row['Counter_Value']))
            if not vals:
                return None, f'no tiled-GEMM rows in the {counter} pass'
            sums[counter] = (sum(vals), len(vals))
    except Exception as e:                                    # a profiler problem must not cost the bench line
        return None, f'traffic pass failed: {e}'
    finally:
        shutil.rmtree(work, ignore_errors=True)
    fb = 2 * sums['FETCH_SIZE'][0] * 1024 / sums['FETCH_SIZE'][1]
    wb = sums['WRITE_SIZE'][0] * 1024 / sums['WRITE_SIZE'][1]
    return fb + wb, (f'measured in this run: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate child runs of this script at {pages} pages x 2 new tokens: '
                     f'the 64-page step\'s tiled-GEMM launch shapes), {sums["FETCH_SIZE"][1]} launches; bytes per launch on the L2 fabric side (Infinity-Cache hits '
                     f'included) = (2*FETCH_SIZE + WRITE_SIZE)*1024: reads {fb / 1e9:.2f} GB + writes {wb / 1e9:.2f} GB')


def cpu_baseline_full(threads):
    """Calibration of the sample above: the oracle on ONE WHOLE page of the bench's shape -- 107 tiles through
    24 ViT layers + mlp1, 96 of them through the 4-layer resampler + VQ + de-normalisation, splice, 3164-token prefill through 32
    layers and DECODE_STEPS greedy steps -- timed stage by stage on this host; the decode is extrapolated to NEW_TOKENS from its
    own measured steps only.  Never inside the timed region; `--cpu-baseline full`."""
    from callireader_amd.config import ModelDims, IMG_CONTEXT_TOKEN_ID, ALIGNED_TOKEN_ID
    from callireader_amd import synthetic
    from oracle import vision, calli_align, generate, internlm2
    DECODE_STEPS = 8
    dims = ModelDims.full()
    torch.set_num_threads(threads)
    t = {}
    with torch.no_grad():
        sd = synthetic.make_state_dict(dims, parts=('vit', 'mlp1', 'resampler', 'vq'), seed=0)
        page_px, char_px = synthetic.make_pixels(PAGE_TILES, seed=10), synthetic.make_pixels(CHAR_TILES, seed=20)
        t0 = time.time()
        feat_page = vision.extract_feature(sd, page_px, dims.vit_layers)
        feat_char = torch.cat([vision.extract_feature(sd, char_px[i:i + 16], dims.vit_layers) for i in range(0, CHAR_TILES, 16)])
        t['vit_mlp1_107_tiles_s'] = time.time() - t0
        t0 = time.time()
        rs = calli_align.resampler_forward(sd, feat_char, dims.rs_depth)
        idx = calli_align.vq_cos_sim(sd['normed_emb.weight'], rs)
        pseudo, _ = calli_align.denormalise(rs, idx, sd['normed_emb.weight'], sd['calli.mu'], sd['calli.sigma'])
        t['resampler_vq_96_tiles_s'] = time.time() - t0
        del sd
        t0 = time.time()
        lsd = synthetic.make_state_dict(dims, parts=('llm',), seed=0)
        t['llm_weights_generated_s'] = time.time() - t0        # not part of a page
        ids = build_ids(PAGE_TILES, CHAR_TILES, TEXT_TOKENS, IMG_CONTEXT_TOKEN_ID, ALIGNED_TOKEN_ID, 1000)[None]
        emb = generate.splice_embeddings(lsd, ids, feat_page, pseudo, IMG_CONTEXT_TOKEN_ID, ALIGNED_TOKEN_ID)
        rope = internlm2.rope_tables(128)
        t0 = time.time()
        logits, past = internlm2.model_forward(lsd, dims.llm_layers, inputs_embeds=emb, rope=rope, all_logits=False)
        t['prefill_3164_tokens_s'] = time.time() - t0
        nxt = int(torch.argmax(logits[0, -1]))
        t0 = time.time()
        for _ in range(DECODE_STEPS):
            logits, past = internlm2.model_forward(lsd, dims.llm_layers, input_ids=torch.tensor([[nxt]]), past=past, rope=rope)
            nxt = int(torch.argmax(logits[0, -1]))
        t['decode_s_per_token'] = (time.time() - t0) / DECODE_STEPS
    page_s = t['vit_mlp1_107_tiles_s'] + t['resampler_vq_96_tiles_s'] + t['prefill_3164_tokens_s'] + t['decode_s_per_token'] * NEW_TOKENS
    return {'what': f'the oracle on one whole page, stage by stage, {threads} threads of {os.cpu_count()} host cores; decode = {DECODE_STEPS} measured steps x {NEW_TOKENS}',
            'cpu_model': _cpu_model(), 'stages': {k: round(v, 3) for k, v in t.items()}, 's_per_page': round(page_s, 1), 'pages_per_s': 1.0 / page_s}


def _cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return platform.processor() or platform.machine()


NEW_TOKENS = 128


def main():
    global NEW_TOKENS
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
    ap.add_argument('--no-strong-share', action='store_true', help='N = 1: do not time one rank\'s share of the strong-scaling step (strong_share)')
    ap.add_argument('--share-world', type=int, default=8, help='the world size whose rank-0 share strong_share runs on this one GPU')
    ap.add_argument('--share-steps', type=int, default=2, help='timed steps of strong_share')
    ap.add_argument('--fp8-extras', action='store_true', help='also time the batched decode on e4m3 weight copies alone (fp8_decode) and compare its first picks with the bf16 decode')
    ap.add_argument('--no-api', action='store_true', help='N = 1: skip the api_level block (the same batch from JPEG files through chat_ocr_stream / folder_rec to strings)')
    ap.add_argument('--api-batches', type=int, default=4, help='batches of --pages pages the api_level block streams')
    ap.add_argument('--no-pipeline', action='store_true', help='one batch at a time (the decode of a batch does not run beside the visual stage of the next)')
    args = ap.parse_args()
    NEW_TOKENS = args.new_tokens

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
    model = InternVLChatModel.from_synthetic(dims, seed=0, device=local_rank, max_tokens=S_page + NEW_TOKENS + 128, max_pages=P)
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

    def measure_balanced(pb, full_ms, full_out, pipelined_ms, n_even, cost_name='MI355X_COST'):
        """strong_share.balanced: the two kinds of rank of a plan_balanced plan, each timed alone on this GPU against the one-GPU step `full_ms` (whose ids are `full_out`)."""
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
        if 2 * n_own <= args.pages and 2 * n_own <= 64:
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
            balanced = measure_balanced(plan_workload('strong', args.pages, args.pages, args.share_world, 0, plan='balanced', owners=args.balanced_owners)['balanced'],
                                        full_ms, full_out, elapsed / args.steps * 1e3, n_own)
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
            'balanced': balanced}
        del ins

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
        if 2 * ws['pages_per_gpu'] <= 64:
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
            except Exception as e:                      # never at the cost of the line
                import traceback
                traceback.print_exc()
                result['api_level'] = {'error': f'{type(e).__name__}: {e}'}
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
                    share8['balanced'] = measure_balanced(pb8, dt_step8 * 1e3, out_step8, None, w8s['pages_per_gpu'], cost_name='MI355X_COST_FP8')
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
        if not args.no_traffic and args.pages >= 16 and args.scaling == 'weak':
            # HBM-side bytes per launch of the dominant kernel class, measured on THIS box in THIS run (untimed, after everything else)
            try:
                pipe = None
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
