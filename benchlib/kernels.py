"""`roofline.by_kernel`: the kernels of the path ONE AT A TIME at the launch shapes of bench.py's 64-page step, each against the roofline that bounds it (round-5
verdict, item 7: the bench line's `roofline` names a kernel CLASS; the per-kernel fractions existed only in profiles/).  Untimed extra, HIP events on the launch
stream around 3 launches after a warm-up, random operands (all-zero data would flatter the clock).  FLOPs are the algorithmic 2 M N K (attention: 4 S^2 d per head,
causal half of it); bytes of the decode attention = the K / V rows it has to read."""
import torch

from .measure import PEAK_BF16_TFLOPS, PEAK_HBM_GBS

EPI_STORE, EPI_GELU, EPI_LS_RES, EPI_RES, EPI_SWIGLU = 0, 1, 2, 3, 4


def _timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def by_kernel(dev, vit_tiles=255, prefill_pages=16, prompt=3164, decode_rows=64, ctx_tokens=3228):
    from callireader_amd import engine as E
    g = torch.Generator(device=dev).manual_seed(0)

    def rnd(*shape, scale=1.0):
        return ((torch.rand(*shape, device=dev, generator=g) * 2 - 1) * scale).bfloat16()
    out = {}

    def gemm(name, what, epi, M, N, K, bias=False, scale=False, res=False):
        A, W = rnd(M, K), rnd(N, K, scale=0.05)
        b = rnd(N) if bias else None
        sc = rnd(N) if scale else None
        n_out = N // 2 if epi == EPI_SWIGLU else N
        r = rnd(M, n_out) if res else None
        C = torch.empty(M, n_out, device=dev, dtype=torch.bfloat16)
        ms = _timed(lambda: E.op_gemm(epi, A, W, bias=b, scale=sc, res=r, out=C))
        tf = 2.0 * M * N * K / (ms * 1e-3) / 1e12
        out[name] = {'what': what, 'shape_MNK': [M, N, K], 'ms': round(ms, 4), 'bound': 'mfma', 'achieved': round(tf, 1), 'unit': 'TFLOP/s', 'frac': round(tf / PEAK_BF16_TFLOPS, 4)}
        del A, W, C, r
    Mv = vit_tiles * 1025
    gemm('vit_qkv', f'InternViT qkv linear + bias, {vit_tiles}-tile chunk', EPI_STORE, Mv, 3072, 1024, bias=True)
    gemm('vit_proj', 'InternViT attention proj + bias + LayerScale + residual', EPI_LS_RES, Mv, 1024, 1024, bias=True, scale=True, res=True)
    gemm('vit_fc1_gelu', 'InternViT fc1 + bias + erf-GELU', EPI_GELU, Mv, 4096, 1024, bias=True)
    gemm('vit_fc2', 'InternViT fc2 + bias + LayerScale + residual', EPI_LS_RES, Mv, 1024, 4096, bias=True, scale=True, res=True)
    Mp = prefill_pages * prompt
    gemm('llm_wqkv', f'InternLM2 wqkv, prefill batch of {prefill_pages} x {prompt} rows', EPI_STORE, Mp, 6144, 4096)
    gemm('llm_wo', 'InternLM2 wo + residual', EPI_RES, Mp, 4096, 4096, res=True)
    gemm('llm_w1w3_swiglu', 'InternLM2 w1|w3 + SwiGLU', EPI_SWIGLU, Mp, 28672, 4096)
    gemm('llm_w2', 'InternLM2 w2 + residual', EPI_RES, Mp, 4096, 14336, res=True)
    gemm('resampler_to_kv', 'PerceiverResampler to_kv, 252-tile chunk', EPI_STORE, 252 * 259, 1024, 4096)

    # the weight streams of a 64-row decode step (HBM-bound): the K-sliced partial-sum GEMMs for wqkv / wo / w2 and the X-through-LDS stream kernel for w1|w3, each on the
    # decode-layout copy of its weight as the product keeps it, alternating between three weight copies so that nothing is served from L2 / the Infinity Cache
    def stream(name, what, epi, which, N, K, M=decode_rows):
        A = rnd(M, K)
        Ws = [rnd(N, K, scale=0.05) for _ in range(3)]
        kind = 2 if (which == 0 and epi == 7) else 1
        SW = [E.op_decode_swizzle(which if (kind == 2 or which) else 1, W) for W in Ws]
        C = E.op_gemm(epi, A, Ws[0])
        state = {'i': 0}

        def f():
            i = state['i'] = (state['i'] + 1) % 3
            E.op_gemm(epi, A, Ws[i], out=C, decode_layout=(kind, SW[i]))
        ms = _timed(f, reps=30)
        gbs = N * K * 2 / (ms * 1e-3) / 1e9
        out[name] = {'what': what, 'shape_MNK': [M, N, K], 'ms': round(ms, 4), 'bound': 'hbm', 'achieved': round(gbs, 1), 'unit': 'GB/s', 'frac': round(gbs / PEAK_HBM_GBS, 4),
                     'bytes': N * K * 2, 'note': 'weight bytes only (the activations add M x K x 2 per workgroup column through L2)'}
    stream('decode_wqkv', f'InternLM2 wqkv at {decode_rows} decode rows (gemm_skinny_kernel, K-sliced fp32 partial sums)', 7, 0, 6144, 4096)
    stream('decode_wo', 'InternLM2 wo (K-sliced partial sums)', 7, 1, 4096, 4096)
    stream('decode_w1w3_swiglu', 'InternLM2 w1|w3 + SwiGLU (gemm_stream_kernel: X through LDS)', EPI_SWIGLU, 2, 28672, 4096)
    stream('decode_w2', 'InternLM2 w2 (K-sliced partial sums)', 7, 3, 4096, 14336)
    # LayerNorm of the ViT (HBM-bound: one read, one write)
    x = rnd(Mv, 1024)
    gam, bet = rnd(1024), rnd(1024)
    ms = _timed(lambda: E.op_layernorm(x, gam, bet, 1e-6))
    gbs = 2.0 * Mv * 1024 * 2 / (ms * 1e-3) / 1e9
    out['vit_layernorm'] = {'what': f'InternViT LayerNorm, {vit_tiles}-tile chunk', 'ms': round(ms, 4), 'bound': 'hbm', 'achieved': round(gbs, 1), 'unit': 'GB/s', 'frac': round(gbs / PEAK_HBM_GBS, 4)}
    del x

    # ViT attention (16 heads x 64, 1025 tokens, no mask): modeling_intern_vit.py:215-232
    S, H, D = 1025, 16, 64
    qkv = rnd(vit_tiles, S, 3 * H * D)
    o = torch.empty(vit_tiles, S, H * D, device=dev, dtype=torch.bfloat16)
    C3, C1 = 3 * H * D, H * D
    st = [S * C3, C3, D, S * C3, C3, D, S * C3, C3, D, S * C1, C1, D]
    ms = _timed(lambda: E.op_attention(qkv, qkv[:, :, C1:], qkv[:, :, 2 * C1:], o, st, vit_tiles, H, S, S, D, q_prescale=0.125))
    tf = 4.0 * S * S * D * H * vit_tiles / (ms * 1e-3) / 1e12
    out['vit_attn'] = {'what': f'InternViT attention, {vit_tiles} tiles x 16 heads x 1025 x 64 (vit_attn_kernel)', 'ms': round(ms, 4), 'bound': 'mfma', 'achieved': round(tf, 1), 'unit': 'TFLOP/s',
                       'frac': round(tf / PEAK_BF16_TFLOPS, 4),
                       'note': 'vector-issue bound, not LDS bound (profiles/round6: LDS array 24 % busy, 0 bank-conflict cycles; 45 issue cycles of softmax VALU per 32-cycle MFMA)'}
    del qkv, o
    # LLM causal prefill attention (32 query / 8 KV heads x 128): modeling_internlm2.py:390-410; the pages of a prefill batch as the batch dimension
    NH, NKV, HD, Bp = 32, 8, 128, prefill_pages
    q = rnd(Bp, prompt, NH * HD)
    k, v = rnd(Bp, NKV, prompt, HD), rnd(Bp, NKV, prompt, HD)
    o2 = torch.empty(Bp, prompt, NH * HD, device=dev, dtype=torch.bfloat16)
    st2 = [prompt * NH * HD, NH * HD, HD, NKV * prompt * HD, HD, prompt * HD, NKV * prompt * HD, HD, prompt * HD, prompt * NH * HD, NH * HD, HD]
    ms = _timed(lambda: E.op_attention(q, k, v, o2, st2, Bp, NH, prompt, prompt, HD, kv_group=NH // NKV, causal=True, s_div=11.313708498984761))
    tf = 4.0 * prompt * (prompt + 1) / 2 * HD * NH * Bp / (ms * 1e-3) / 1e12
    out['llm_prefill_attn'] = {'what': f'InternLM2 causal prefill attention, {Bp} pages x 32 heads x {prompt} x 128 (flash_attn_kernel<128, causal>)', 'ms': round(ms, 4), 'bound': 'mfma',
                               'achieved': round(tf, 1), 'unit': 'TFLOP/s', 'frac': round(tf / PEAK_BF16_TFLOPS, 4)}
    del q, k, v, o2
    # decode attention over the cache (HBM-bound): one new token per row, `ctx_tokens` cached tokens per row
    mt = ctx_tokens + 64
    kc, vc = rnd(decode_rows, NKV, mt, HD), rnd(decode_rows, NKV, mt, HD)
    qd = rnd(decode_rows, NH * HD)
    seqs = torch.arange(decode_rows, device=dev, dtype=torch.int32)
    lens = torch.full((decode_rows,), ctx_tokens, device=dev, dtype=torch.int32)
    n_sc = int(E.B.lib.cr_op_decode_attention_scratch_floats(decode_rows, ctx_tokens + 1))
    scratch = torch.empty(n_sc, device=dev, dtype=torch.float32)
    od = torch.empty(decode_rows, NH * HD, device=dev, dtype=torch.bfloat16)
    ms = _timed(lambda: E.op_decode_attention(qd, kc, vc, seqs, lens, max_keys=ctx_tokens + 1, scratch=scratch, out=od), reps=10)
    by = decode_rows * (ctx_tokens + 1) * 2 * NKV * HD * 2
    gbs = by / (ms * 1e-3) / 1e9
    out['decode_attn'] = {'what': f'decode attention over the KV cache, {decode_rows} rows x {ctx_tokens + 1} keys (decode_attn_kernel + combine)', 'ms': round(ms, 4), 'bound': 'hbm',
                          'achieved': round(gbs, 1), 'unit': 'GB/s', 'frac': round(gbs / PEAK_HBM_GBS, 4), 'bytes': by}
    return out
