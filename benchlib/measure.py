"""Measurements around bench.py's timed region that do not need the model: the CPU baseline (the oracle on a bounded sample / on one whole page) and the
fabric-side traffic of the tiled-GEMM class from `rocprofv3 --pmc` child runs of bench.py."""
import os
import platform
import sys
import time

import torch

from . import plan
from .plan import PAGE_TILES, CHAR_TILES, TEXT_TOKENS, build_ids

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, 'bench.py')
PEAK_BF16_TFLOPS = 2500.0      # dense bf16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"
PEAK_HBM_GBS = 8000.0          # HBM3E spec, same table

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
    page_s = t_vit * (PAGE_TILES + CHAR_TILES) + t_rs * CHAR_TILES + t_pre * S_page + t_dec * plan.NEW_TOKENS
    ratio = CPU_FULL_PAGE_MEASURED_S / CPU_FULL_PAGE_SAMPLED_S
    return {'value': 1.0 / page_s, 'unit': 'pages/s', 'cores': cores, 'kind': 'port',
            'cpu': _cpu_model(), 'host_cores': os.cpu_count(),
            # (round-5 verdict, weak #10: the whole-page run behind the ratio used 64 threads; the calibrated value is only quoted when this sample ran at that count too)
            'measured_over_sampled': round(ratio, 2), 'value_calibrated': 1.0 / (page_s * ratio) if cores == 64 else None,
            'calibration': (f'the sample under-states a page: the one whole-page run on record (--cpu-baseline full, EPYC 9575F, 64 threads, profiles/round3/'
                            f'01_bench_N1_default_full_cpu_baseline.json) measured {CPU_FULL_PAGE_MEASURED_S:.0f} s per page where its own sample said '
                            f'{CPU_FULL_PAGE_SAMPLED_S:.0f} s; value_calibrated = value / {ratio:.2f}, quoted only when this sample ran at 64 threads as that run did'),
            'sample': (f'oracle bf16 eager: ViT+mlp1 24 layers on 2 tiles ({t_vit:.2f} s/tile), resampler 1 of 4 layers on 2 tiles, '
                       f'InternLM2 1 of 32 layers prefill {S} tokens ({t_pre * 1e3:.1f} ms/token x32) + 4 decode steps '
                       f'({t_dec * 1e3:.0f} ms/token x32); extrapolated linearly to one page (107 tiles, {S_page} prompt tokens, '
                       f'{plan.NEW_TOKENS} new tokens) = {page_s:.0f} s/page; sample wall {time.time() - t_all:.0f} s')}


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
            r = subprocess.run([prof, '--pmc', counter, '--output-format', 'csv', '-d', d, '--', sys.executable, BENCH] + args,
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
    layers and DECODE_STEPS greedy steps -- timed stage by stage on this host; the decode is extrapolated to plan.NEW_TOKENS from its
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
    page_s = t['vit_mlp1_107_tiles_s'] + t['resampler_vq_96_tiles_s'] + t['prefill_3164_tokens_s'] + t['decode_s_per_token'] * plan.NEW_TOKENS
    return {'what': f'the oracle on one whole page, stage by stage, {threads} threads of {os.cpu_count()} host cores; decode = {DECODE_STEPS} measured steps x {plan.NEW_TOKENS}',
            'cpu_model': _cpu_model(), 'stages': {k: round(v, 3) for k, v in t.items()}, 's_per_page': round(page_s, 1), 'pages_per_s': 1.0 / page_s}


def _cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return platform.processor() or platform.machine()
