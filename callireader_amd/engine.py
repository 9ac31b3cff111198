"""Thin torch<->C-ABI plumbing: device buffers and streams come from PyTorch-ROCm,
all compute goes through libcallireader_hip.so.  No torch math on the hot path."""
import ctypes as C

import torch

from . import _binding as B
from .config import ModelDims

_DT = {torch.bfloat16: B.CR_BF16, torch.float32: B.CR_F32, torch.int64: B.CR_I64, torch.int32: B.CR_I32}


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def rope_tables(dims: ModelDims, rows=None):
    """bf16 cos/sin cache exactly as InternLM2DynamicNTKScalingRotaryEmbedding builds it at init
    (reference InternVL/modeling_internlm2.py:154,213-229; returned in bf16 by :177-180).
    Host-side setup (fp32 torch on CPU, like the reference), uploaded once as weights."""
    rows = dims.max_pos if rows is None else rows
    hd = dims.llm_head_dim
    inv_freq = 1.0 / (dims.rope_theta ** (torch.arange(0, hd, 2).float() / hd))
    t = torch.arange(rows).to(inv_freq.dtype)
    freqs = torch.einsum('i,j->ij', t, inv_freq)
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos().to(torch.bfloat16), emb.sin().to(torch.bfloat16)


class Engine:
    """One context per device (include/callireader_hip.h)."""

    def __init__(self, dims: ModelDims = None, device=0, max_pos=None):
        if not torch.cuda.is_available():
            raise RuntimeError('callireader_amd needs a ROCm GPU: there is no CPU fallback for the hot path')
        self.dims = dims or ModelDims.full()
        self.device = torch.device('cuda', device)
        self.max_pos = max_pos or self.dims.max_pos
        d = B.ModelDesc(vit_layers=self.dims.vit_layers, rs_depth=self.dims.rs_depth, llm_layers=self.dims.llm_layers,
                        vocab=self.dims.vocab, max_pos=self.max_pos, vit_ln_eps=self.dims.vit_ln_eps,
                        rms_eps=self.dims.rms_eps)
        h = C.c_void_p()
        torch.cuda.set_device(self.device)
        B.check(B.lib.cr_create(device, C.byref(d), C.byref(h)), 'cr_create')
        self._h = h
        self._kv = []
        self.fp8_decode, self.fp8_mfma = False, 0          # mirrors of the context's options (parallel.default_cost picks the cost table by them)
        self.weights_version = 0                           # bumped by everything after which a borrower (share_weights_from) must share again: loads, finalize, the fp8 switches

    def close(self):
        if getattr(self, '_h', None):
            B.lib.cr_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- weights ----
    def load_weight(self, name, t):
        t = t.detach().contiguous()
        if not name.startswith('orderformer.'):
            self.fp8_decode, self.fp8_mfma = False, 0      # cr_load_weight switches the options off with the copies they read
        self.weights_version += 1
        shape = (C.c_int64 * t.dim())(*t.shape)
        B.check(B.lib.cr_load_weight(self._h, name.encode(), _p(t), _DT[t.dtype], shape, t.dim(),
                                     0 if t.is_cuda else 1, _stream()), f'cr_load_weight({name})')
        if t.is_cuda:
            torch.cuda.current_stream().synchronize()   # the source may be freed by the caller right after

    def load_state_dict(self, sd):
        for k, v in (sd.items() if hasattr(sd, 'items') else sd):
            self.load_weight(k, v if v.dtype in (torch.bfloat16,) or not v.is_floating_point() else v.to(torch.bfloat16))

    def load_rope(self):
        cos, sin = rope_tables(self.dims, self.max_pos)
        self.load_weight('rope.cos', cos)
        self.load_weight('rope.sin', sin)

    def finalize(self):
        B.check(B.lib.cr_finalize(self._h, _stream()), 'cr_finalize')
        self.weights_version += 1

    def share_weights_from(self, other):
        """cr_share_weights: this (fresh) engine uses `other`'s finalized weights without copying them -- a second host thread can
        then run stages through it on its own stream while `other` keeps working (PagePipeline).  `other` must outlive it."""
        B.check(B.lib.cr_share_weights(self._h, other._h), 'cr_share_weights')
        self._shared_from = other

    def enable_fp8_decode(self, on=True):
        """Batched decode streams e4m3 copies of the LLM's linear weights (one fp32 scale per output row) instead of the
        bf16 ones: half the HBM bytes per step.  Off by default: the reference computes in bf16 (include/callireader_hip.h)."""
        B.check(B.lib.cr_enable_fp8_decode(self._h, 1 if on else 0, _stream()), 'cr_enable_fp8_decode')
        self.fp8_decode = bool(on)
        self.weights_version += 1

    # ---- vision ----
    def enable_fp8_mfma(self, on=True, level=1):
        """cr_enable_fp8_mfma.  level 1: norm-fed linears (ViT QKV / fc1, mlp1[1], LLM prefill wqkv / w1|w3) multiply e4m3 x e4m3 on the
        matrix cores; level 2: also ViT fc2 and the prefill's wo / w2.  OFF by default (the reference's arithmetic is bf16); a throughput
        option whose accuracy only a real checkpoint can price (evaluate.py --compare_fp8)."""
        B.check(B.lib.cr_enable_fp8_mfma(self._h, (2 if level >= 2 else 1) if on else 0, _stream()), 'cr_enable_fp8_mfma')
        self.fp8_mfma = ((2 if level >= 2 else 1) if on else 0)
        self.weights_version += 1

    def _chk_pixels(self, px):
        if px.dim() != 4:
            raise ValueError(f'wrong pixel_values size: {px.shape}')       # modeling_intern_vit.py:417-420
        if tuple(px.shape[1:]) != (3, self.dims.image_size, self.dims.image_size):
            raise ValueError(f'pixel_values must be (T,3,448,448), got {tuple(px.shape)}')
        return px.to(self.device, torch.bfloat16).contiguous()

    def vit_forward(self, pixel_values):
        px = self._chk_pixels(pixel_values)
        T = px.shape[0]
        out = torch.empty(T, self.dims.vit_tokens, self.dims.vit_hidden, device=self.device, dtype=torch.bfloat16)
        B.check(B.lib.cr_vit_forward(self._h, _p(px), T, _p(out), _stream()), 'cr_vit_forward')
        return out

    def project(self, vit_out):
        T = vit_out.shape[0]
        vit_out = vit_out.contiguous()
        out = torch.empty(T, self.dims.tokens_per_tile, self.dims.llm_hidden, device=self.device, dtype=torch.bfloat16)
        B.check(B.lib.cr_project(self._h, _p(vit_out), T, _p(out), _stream()), 'cr_project')
        return out

    def extract_feature(self, pixel_values):
        px = self._chk_pixels(pixel_values)
        T = px.shape[0]
        out = torch.empty(T, self.dims.tokens_per_tile, self.dims.llm_hidden, device=self.device, dtype=torch.bfloat16)
        B.check(B.lib.cr_extract_feature(self._h, _p(px), T, _p(out), _stream()), 'cr_extract_feature')
        return out

    # ---- tile preprocessing on the GPU ----
    def preprocess(self, page_u8, jobs, n_tiles, out=None):
        """page_u8: uint8 (H,W,3) RGB tensor (moved to the device once); jobs: dict rows from preprocess.plan_*, or their int32 (n, 11) table
        (preprocess.jobs_array / plan_chars_array); out: the (n_tiles,3,448,448) bf16 tensor the jobs' tile0 slots index (default: a new one)."""
        import numpy as np
        from .preprocess import norm_lut, jobs_array
        if getattr(self, '_lut', None) is None:
            self._lut = norm_lut().to(self.device)
        page = page_u8.to(self.device).contiguous()
        assert page.dtype == torch.uint8 and page.dim() == 3 and page.shape[2] == 3
        tab = np.ascontiguousarray(jobs if isinstance(jobs, np.ndarray) else jobs_array(jobs), dtype=np.int32)
        assert tab.ndim == 2 and tab.shape[1] * 4 == C.sizeof(B.PrepJob)
        if out is None:
            out = torch.empty(n_tiles, 3, self.dims.image_size, self.dims.image_size, device=self.device, dtype=torch.bfloat16)
        assert out.is_contiguous() and out.shape[0] == n_tiles and out.dtype == torch.bfloat16
        B.check(B.lib.cr_preprocess(self._h, _p(page), page.shape[0], page.shape[1], tab.ctypes.data_as(C.c_void_p), tab.shape[0], _p(self._lut), _p(out),
                                    n_tiles, _stream()), 'cr_preprocess')
        return out

    # ---- CalliAlign ----
    def resample(self, feats):
        feats = feats.contiguous()
        T = feats.shape[0]
        out = torch.empty(T, self.dims.rs_queries, self.dims.llm_hidden, device=self.device, dtype=torch.bfloat16)
        B.check(B.lib.cr_resample(self._h, _p(feats), T, _p(out), _stream()), 'cr_resample')
        return out

    def orderformer(self, boxes):
        """boxes (B, L, 4) bf16 -> (B, L) fp32 scores of the OrderFormer (weights `orderformer.*` loaded beforehand)."""
        x = boxes.to(device=self.device, dtype=torch.bfloat16).contiguous()
        Bn, L, four = x.shape
        assert four == 4
        out = torch.empty(Bn, L, device=self.device, dtype=torch.float32)
        B.check(B.lib.cr_orderformer(self._h, _p(x), Bn, L, _p(out), _stream()), 'cr_orderformer')
        return out

    def vq(self, x, with_cos=False):
        x2 = x.reshape(-1, x.shape[-1]).contiguous()
        n = x2.shape[0]
        idx = torch.empty(n, device=self.device, dtype=torch.int64)
        cos = torch.empty(n, device=self.device, dtype=torch.bfloat16) if with_cos else None
        B.check(B.lib.cr_vq(self._h, _p(x2), n, _p(idx), _p(cos), _stream()), 'cr_vq')
        idx = idx.reshape(x.shape[:-1])
        return (idx, cos.reshape(x.shape[:-1])) if with_cos else idx

    def denorm(self, x, idx, cos=None, drop_zero=False, hard_vq=False):
        x2 = x.reshape(-1, x.shape[-1]).contiguous()
        n = x2.shape[0]
        idx1 = idx.reshape(-1).contiguous()
        out = torch.empty_like(x2)
        n_out = torch.zeros(1, device=self.device, dtype=torch.int32)
        flags = (1 if drop_zero else 0) | (2 if hard_vq else 0)
        if hard_vq and cos is None:
            raise ValueError('hard_vq needs the cosine values from vq(..., with_cos=True)')
        cosr = cos.reshape(-1).contiguous() if cos is not None else None
        B.check(B.lib.cr_denorm(self._h, _p(x2), _p(idx1), _p(cosr), n, flags, _p(out), _p(n_out), _stream()), 'cr_denorm')
        return out[:int(n_out.item())] if drop_zero else out

    # ---- language model ----
    def embed_splice(self, input_ids, vit_embeds=None, ref_embeds=None, img_id=92546, ref_id=92537):
        ids = input_ids.reshape(-1).to(self.device, torch.int64).contiguous()
        S = ids.shape[0]
        out = torch.empty(S, self.dims.llm_hidden, device=self.device, dtype=torch.bfloat16)
        v = vit_embeds.reshape(-1, self.dims.llm_hidden).contiguous() if vit_embeds is not None else None
        r = ref_embeds.reshape(-1, self.dims.llm_hidden).to(torch.bfloat16).contiguous() if ref_embeds is not None else None
        B.check(B.lib.cr_embed_splice(self._h, _p(ids), S, _p(v), 0 if v is None else v.shape[0], img_id,
                                      _p(r), 0 if r is None else r.shape[0], ref_id, _p(out), _stream()), 'cr_embed_splice')
        return out

    def kv_alloc(self, n_seqs, max_tokens):
        h = C.c_void_p()
        B.check(B.lib.cr_kv_alloc(self._h, n_seqs, max_tokens, C.byref(h)), 'cr_kv_alloc')
        return KVCache(self, h, n_seqs, max_tokens)

    def prefill(self, kv, seq, embeds, penalty=1.0, want_logits=False):
        e = embeds.reshape(-1, self.dims.llm_hidden).contiguous()
        S = e.shape[0]
        logits = torch.empty(self.dims.vocab, device=self.device, dtype=torch.float32) if want_logits else None
        B.check(B.lib.cr_llm_prefill(self._h, kv._h, seq, _p(e), S, float(penalty), _p(logits), _stream()), 'cr_llm_prefill')
        return logits

    def prefill_batch(self, kv, seqs, embeds_list, penalty=1.0, want_logits=False):
        """Prefill several sequences in one pass (linear layers over all prompt rows, attention per sequence)."""
        n = len(seqs)
        es = [e.reshape(-1, self.dims.llm_hidden) for e in embeds_list]
        lens = [e.shape[0] for e in es]
        cat = es[0].contiguous() if n == 1 else torch.cat(es, dim=0)
        logits = torch.empty(n, self.dims.vocab, device=self.device, dtype=torch.float32) if want_logits else None
        B.check(B.lib.cr_llm_prefill_batch(self._h, kv._h, (C.c_int32 * n)(*seqs), n, _p(cat), (C.c_int32 * n)(*lens),
                                           float(penalty), _p(logits), _stream()), 'cr_llm_prefill_batch')
        return logits

    def hidden_probe(self, rows=None, row0=0):
        """cr_llm_hidden_probe: `rows` > 0 arms it and returns the [llm_layers + 1, rows, 4096] bf16 tensor the next prefill fills
        (residual stream before layer 0 and after each layer); rows=None disarms."""
        if not rows:
            B.check(B.lib.cr_llm_hidden_probe(self._h, C.c_void_p(0), 0, 0), 'cr_llm_hidden_probe')
            self._probe = None
            return None
        self._probe = torch.zeros(self.dims.llm_layers + 1, rows, self.dims.llm_hidden, device=self.device, dtype=torch.bfloat16)
        B.check(B.lib.cr_llm_hidden_probe(self._h, _p(self._probe), row0, rows), 'cr_llm_hidden_probe')
        return self._probe

    def decode(self, kv, seqs, penalty=1.0, force_tokens=None, want_logits=False):
        n = len(seqs)
        seq_arr = (C.c_int32 * n)(*seqs)
        ft = force_tokens.reshape(-1).to(self.device, torch.int64).contiguous() if force_tokens is not None else None
        logits = torch.empty(n, self.dims.vocab, device=self.device, dtype=torch.float32) if want_logits else None
        B.check(B.lib.cr_llm_decode(self._h, kv._h, seq_arr, n, _p(ft), float(penalty), _p(logits), _stream()), 'cr_llm_decode')
        return logits


class KVCache:
    def __init__(self, eng, h, n_seqs, max_tokens):
        self.eng, self._h, self.n_seqs, self.max_tokens = eng, h, n_seqs, max_tokens

    def length(self, seq):
        return B.lib.cr_kv_length(self._h, seq)

    def reset(self, seq=-1):
        """Forget one sequence, or all of them (seq < 0); enqueued on the current stream."""
        B.check(B.lib.cr_kv_reset(self._h, seq, _stream()), 'cr_kv_reset')

    def read(self, layer, seq, pos, which=0):
        """(8, 128) bf16: K (which=0, after RoPE) or V (which=1) of one cached position, all KV heads."""
        out = torch.empty(8, 128, device=self.eng.device, dtype=torch.bfloat16)
        B.check(B.lib.cr_kv_read(self._h, layer, seq, pos, which, _p(out), _stream()), 'cr_kv_read')
        return out

    def generated(self, seq, max_tokens=4096):
        buf = (C.c_int64 * max_tokens)()
        n = B.lib.cr_kv_generated(self._h, seq, buf, max_tokens, _stream())
        if n < 0:
            B.check(n, 'cr_kv_generated')
        return list(buf[:n])

    def free(self):
        if self._h:
            B.lib.cr_kv_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


# ---- single operators (tests / profiling) ----
def op_gemm(epi, A, Wt, bias=None, scale=None, res=None, group=0, out_dtype=torch.bfloat16, n_out=None, kernel=0, out=None, decode_layout=None):
    """kernel: 0 dispatcher's choice, 1 128x128 tiles, 2 256x256 persistent (schedule per shape), 3 weight-streaming, 5 / 6 the 256x256 kernel with
    its 16- / 32-MFMA-slot schedule pinned (tests pin one).  decode_layout = (kind, op_decode_swizzle(which, Wt)): the weight-streaming kernels
    (M <= 64) read that copy of the weight (kind 1: plain tiles, 2: wqkv's RoPE tile order with epi 7); Wt still gives N."""
    M, K = A.shape
    N = Wt.shape[0]
    hi = 0
    if decode_layout is not None:
        hi = decode_layout[0] << 18
        Wt = decode_layout[1].view(-1, K)
    ncols = n_out if n_out is not None else (N // 2 if epi == 4 else N)
    mrows = M if epi != 5 else (M // group) * (group + 1)
    if epi == 7:                     # fp32 K-slice partial sums [S <= 8][M][N]; unused slabs stay zero
        mrows, out_dtype = 8 * M, torch.float32
    if epi == 8:                     # cosine-VQ partials: per row and 64-column block {column (low word), bits of the bf16 max (high word)}
        ncols, out_dtype = ((N + 63) // 64 + 1) & ~1, torch.int64
    Cc = out if out is not None else torch.zeros(mrows, ncols, device=A.device, dtype=out_dtype)
    B.check(B.lib.cr_op_gemm(epi | (kernel << 8) | hi, _p(A), A.stride(0), _p(Wt), Wt.stride(0), _p(Cc), Cc.stride(0), _p(bias), _p(scale),
                             _p(res), res.stride(0) if res is not None else 0, M, N, K, group, _stream()), 'cr_op_gemm')
    return Cc


def op_quantize_fp8(Wt):
    """rows of a bf16 [N, K] matrix -> (uint8 e4m3 [N, K], fp32 scale [N]) exactly as cr_enable_fp8_decode builds them"""
    N, K = Wt.shape
    q = torch.empty(N, K, device=Wt.device, dtype=torch.uint8)
    sc = torch.empty(N, device=Wt.device, dtype=torch.float32)
    B.check(B.lib.cr_op_quantize_fp8(_p(Wt), Wt.stride(0), N, K, _p(q), _p(sc), _stream()), 'cr_op_quantize_fp8')
    return q, sc


def op_decode_swizzle8(q):
    """cr_op_decode_swizzle(which = 8): the decode layout of an e4m3 [N, K] weight copy (1 KiB per 16-row tile and 64-deep k-step)."""
    N, K = q.shape
    out = torch.empty(((N + 15) // 16) * 16 * K, dtype=torch.uint8, device=q.device)
    B.check(B.lib.cr_op_decode_swizzle(8, _p(q), q.stride(0), N, K, _p(out), _stream()), 'cr_op_decode_swizzle(e4m3)')
    return out


def op_gemm_fp8(epi, A, q, sc, res=None, out_dtype=torch.bfloat16, decode_layout=None):
    """cr_op_gemm with e4m3 weights `q` [N, K] and row scales `sc` (decode kernel, M <= 64); decode_layout = op_decode_swizzle8(q): the kernel streams that copy."""
    M, K = A.shape
    N = q.shape[0]
    hi = 0
    if decode_layout is not None:
        hi = 1 << 18
        q = decode_layout.view(-1, K)
    ncols = N // 2 if epi == 4 else N
    mrows = M
    if epi == 7:
        mrows, out_dtype = 8 * M, torch.float32
    Cc = torch.zeros(mrows, ncols, device=A.device, dtype=out_dtype)
    B.check(B.lib.cr_op_gemm(epi | (1 << 16) | hi, _p(A), A.stride(0), _p(q), q.stride(0), _p(Cc), Cc.stride(0), _p(None), _p(sc),
                             _p(res), res.stride(0) if res is not None else 0, M, N, K, 0, _stream()), 'cr_op_gemm(fp8)')
    return Cc


def op_gemm_fp8x8(epi, a8, a_scale, q, sc, bias=None, out_dtype=torch.bfloat16):
    """cr_op_gemm with e4m3 activations `a8` [M, K] + row scales and e4m3 weights `q` [N, K] + row scales: the 256x256 kernel's
    v_mfma_f32_16x16x128_f8f6f4 instance (epi 0 | 1 | 4 | 6)."""
    M, K = a8.shape
    N = q.shape[0]
    Cc = torch.zeros(M, N // 2 if epi == 4 else N, device=a8.device, dtype=out_dtype)
    B.check(B.lib.cr_op_gemm(epi | (1 << 16) | (1 << 17), _p(a8), a8.stride(0), _p(q), q.stride(0), _p(Cc), Cc.stride(0), _p(bias), _p(sc),
                             _p(a_scale), 0, M, N, K, 0, _stream()), 'cr_op_gemm(fp8 x fp8)')
    return Cc


def op_norm_fp8(x, gamma, beta, eps, next_bound=None):
    """LayerNorm (beta given) or RMSNorm (beta None) of bf16 rows -> (uint8 e4m3 [rows, n], fp32 scale [rows]); with next_bound
    (device floats {largest weight-row norm, largest |bias|} of the next linear) also that linear's output-row scales [rows]"""
    rows, n = x.shape
    q = torch.empty(rows, n, device=x.device, dtype=torch.uint8)
    sc = torch.empty(rows, device=x.device, dtype=torch.float32)
    nxt = torch.empty(rows, device=x.device, dtype=torch.float32) if next_bound is not None else None
    B.check(B.lib.cr_op_norm_fp8(_p(x), _p(gamma), _p(beta), rows, n, float(eps), _p(q), _p(sc), _p(nxt), _p(next_bound), _stream()), 'cr_op_norm_fp8')
    return (q, sc, nxt) if nxt is not None else (q, sc)


def op_gemm_q8(a8, a_scale, q, sc, bias, c_scale):
    """cr_op_gemm_q8: fc1 of the fp8 path -- e4m3 x e4m3, GELU, output as e4m3 rows divided by c_scale[m]"""
    M, K = a8.shape
    N = q.shape[0]
    out = torch.zeros(M, N, device=a8.device, dtype=torch.uint8)
    B.check(B.lib.cr_op_gemm_q8(_p(a8), _p(a_scale), _p(q), _p(sc), _p(bias), _p(c_scale), _p(out), M, N, K, _stream()), 'cr_op_gemm_q8')
    return out


def op_layernorm(x, gamma, beta, eps, pixel_shuffle=False):
    if pixel_shuffle:
        T = x.shape[0]
        out = torch.empty(T * 256, 4096, device=x.device, dtype=torch.bfloat16)
        rows, n = T * 256, 4096
    else:
        out = torch.empty_like(x)
        rows, n = x.shape[0], x.shape[1]
    B.check(B.lib.cr_op_layernorm(_p(x), _p(out), _p(gamma), _p(beta), rows, n, eps, 1 if pixel_shuffle else 0, _stream()),
            'cr_op_layernorm')
    return out


def op_rmsnorm(x, gamma, eps):
    out = torch.empty_like(x)
    B.check(B.lib.cr_op_rmsnorm(_p(x), _p(out), _p(gamma), x.shape[0], x.shape[1], eps, _stream()), 'cr_op_rmsnorm')
    return out


def op_attention(q, k, v, o, strides, Bn, H, Sq, Sk, head_dim, kv_group=1, causal=False, q_pos0=0, q_prescale=1.0, s_div=1.0):
    st = (C.c_int64 * 12)(*strides)
    B.check(B.lib.cr_op_attention(_p(q), _p(k), _p(v), _p(o), st, Bn, H, Sq, Sk, head_dim, kv_group, 1 if causal else 0,
                                  q_pos0, q_prescale, s_div, _stream()), 'cr_op_attention')
    return o


def op_decode_swizzle(which, W):
    """cr_op_decode_swizzle: the decode layout of a [N, K] weight for op_decode_gemm's `which` (1 KiB per 16-row tile and 32-deep k-step)."""
    N, K = W.shape
    out = torch.empty(((N + 15) // 16) * 16 * K, dtype=torch.bfloat16, device=W.device)
    B.check(B.lib.cr_op_decode_swizzle(which, _p(W), W.stride(0), N, K, _p(out), _stream()), 'cr_op_decode_swizzle')
    return out


def op_decode_gemm(which, W, M, X=None, xres=None, gamma=None, eps=1e-5, xio=None, C_out=None, rope=None, flags=0, swizzled=None):
    """cr_op_decode_gemm: one small-batch decode GEMM with its neighbours folded in (gemm_decode.hip).  which: 0 wqkv (rope = dict with
    cos, sin, q_out, kc, vc, seqs, lens, max_tokens), 1 wo, 2 w1|w3, 3 w2, 4 LM head.  swizzled: op_decode_swizzle(which, W) -- the kernel then
    streams that copy (W still gives N and K)."""
    N, K = W.shape
    r = rope or {}
    Wp = W if swizzled is None else swizzled
    B.check(B.lib.cr_op_decode_gemm(which, flags | (256 if swizzled is not None else 0), _p(Wp), K, M, N, K, _p(X), X.shape[1] if X is not None else 0,
                                    _p(xres), _p(gamma), eps, _p(xio),
                                    _p(C_out), C_out.shape[1] if C_out is not None else 0, _p(r.get('cos')), _p(r.get('sin')), _p(r.get('q_out')),
                                    _p(r.get('kc')), _p(r.get('vc')), _p(r.get('seqs')), _p(r.get('lens')), r.get('max_tokens', 0), _stream()),
            'cr_op_decode_gemm')


def op_decode_attention(q, kc, vc, seqs, lens, s_div=11.313708498984761, which=0, max_keys=None, scratch=None, out=None):
    """cr_op_decode_attention: q (B, 4096) bf16 = one new token per row, kc / vc (n_slots, 8, max_tokens, 128) bf16 caches that already hold the token's own K / V row at
    position lens[slot], seqs / lens int32 device tensors; returns (B, 4096) bf16.  which = 1 pins the matrix-core split kernel (A/B)."""
    Bn = q.shape[0]
    max_tokens = kc.shape[2]
    if max_keys is None:
        max_keys = int(lens.max().item()) + 1                      # (a host round trip: benchmarks pass max_keys, scratch and out)
    if scratch is None:
        scratch = torch.empty(int(B.lib.cr_op_decode_attention_scratch_floats(Bn, max_keys)), device=q.device, dtype=torch.float32)
    if out is None:
        out = torch.empty(Bn, 4096, device=q.device, dtype=torch.bfloat16)
    B.check(B.lib.cr_op_decode_attention(which, _p(q), _p(kc), _p(vc), max_tokens, _p(seqs), _p(lens), Bn, max_keys, float(s_div), _p(scratch), _p(out), _stream()),
            'cr_op_decode_attention')
    return out
