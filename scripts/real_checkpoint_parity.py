#!/usr/bin/env python3
"""Real-checkpoint parity, ready to run the day weights exist.

Given the checkpoint the reference loads (INTERNVL_PATH: config.json + sharded safetensors + tokenizer files; ./params with
gauss_norm_mu_sigma.pth ...) it sends ONE page with known boxes -- by default the reference's own examples/0.jpg with the 96 labelled boxes of
examples/0.json, kept as tests/golden/example0.* -- through

  (a) the HIP path (libcallireader_hip.so through the drop-in InternVLChatModel) and
  (b) the CPU oracle (oracle/: the restatement of the reference's eager path, pinned bit-for-bit to the reference's own modules by
      tests/test_oracle_golden.py; the reference itself cannot travel to a GPU box) on the host cores,

the way inference.py:37-42,92-96 runs chat_ocr: bf16, use_p=True, hard_vq=False, drop_zero=False, repetition_penalty=1.0, greedy, and prints per stage:
visual features / resampler output / pseudo tokens rel-L2, VQ index agreement, prefill logits rel-L2 and max |d|, the FIRST token at which the two
greedy streams part and the oracle's top-2 margin there, and the same for each fp8 level (informational).

Exit status: 0 when the streams are identical or part at a margin <= ONE bf16 step of the logits involved AND the HIP logits measured at that step
straddle the gap (oracle/generate.py: near_tie_straddles); 1 otherwise -- i.e. non-zero on any divergence the arithmetic cannot excuse.

  INTERNVL_PATH=/path/to/InternVL python scripts/real_checkpoint_parity.py [--params ./params] [--image p.jpg --boxes p.json] [--max-new-tokens 64]

Development aid: imports oracle/ (allowed for scripts that check, never for the product path); not imported by the package."""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def bf16_step(v):
    """spacing of bf16 values at magnitude |v| (8 significant bits)"""
    v = abs(float(v))
    return 2.0 ** (math.floor(math.log2(v)) - 7) if v > 0 else 2.0 ** -133


def rel_l2(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def first_divergence(a, b):
    for i, (x, y) in enumerate(zip(a, b)):
        if x != y:
            return i
    return None if len(a) == len(b) else min(len(a), len(b))


def judge_divergence(ref_logits_t, hip_logits_t, ref_id, hip_id, generated, penalty):
    """-> dict(excusable, gap, step, d_ref_id, d_hip_id): the rule of this script's exit status."""
    from oracle.generate import near_tie_straddles, apply_repetition_penalty
    sc = apply_repetition_penalty(ref_logits_t.float(), generated, penalty)
    step = max(bf16_step(sc[ref_id]), bf16_step(sc[hip_id]))
    ok, gap, d_ref, d_hip = near_tie_straddles(ref_logits_t, hip_logits_t, ref_id, hip_id, generated, penalty, atol=step)
    return {'excusable': bool(ok), 'oracle_gap': gap, 'one_bf16_step': step, 'd_ref_id': d_ref, 'd_hip_id': d_hip}


def load_host_state_dict(path, params_dir):
    """The checkpoint's tensors on the host, under the reference's key names, for the oracle (bf16; ~19 GB for InternVL2-8B)."""
    import torch
    from callireader_amd.weights import iter_safetensors, strip_ddp
    sd = {k: (t.to(torch.bfloat16) if t.is_floating_point() else t) for k, t in iter_safetensors(path)}
    w = torch.load(os.path.join(params_dir, 'gauss_norm_mu_sigma.pth'), map_location='cpu')['weight']
    sd['calli.mu'], sd['calli.sigma'] = w[:, 0].reshape(-1, 1).contiguous(), w[:, 1].reshape(-1, 1).contiguous()
    if not any(k.startswith('resampler.') for k in sd):
        for k, t in strip_ddp(torch.load(os.path.join(params_dir, 'callialign.pth'), map_location='cpu', weights_only=False)).items():
            sd['resampler.' + k] = t.to(torch.bfloat16)
    if 'normed_emb.weight' not in sd:
        sd['normed_emb.weight'] = torch.load(os.path.join(params_dir, 'gauss_norm.pth'), map_location='cpu', weights_only=True)['weight'].to(torch.bfloat16)
    return sd


def run(model_path, params_dir, image, boxes, question='这幅书法作品内容是什么？', max_new_tokens=64, penalty=1.0, fp8=True, threads=None,
        model=None, tokenizer=None, sd=None, aligned_token='[UNUSED_TOKEN_140]', log=print):
    """Returns (report dict, exit status).  `model` / `tokenizer` / `sd` may be handed in (the GPU test does, on a synthetic checkpoint)."""
    import numpy as np
    import torch
    from PIL import Image
    from callireader_amd import preprocess
    from callireader_amd.conversation import get_conv_template
    from callireader_amd.modeling_internvl_chat import InternVLChatModel
    from callireader_amd.tokenization_internlm2 import InternLM2Tokenizer
    from oracle import vision, calli_align, generate

    if threads:
        torch.set_num_threads(threads)
    t_all = time.time()
    if model is None:
        model = InternVLChatModel.from_pretrained(model_path, params_dir=params_dir, torch_dtype=torch.bfloat16, max_tokens=8192, max_pages=1).eval().cuda()
    tok = tokenizer or InternLM2Tokenizer.from_pretrained(model_path)
    if sd is None:
        sd = load_host_state_dict(model_path, params_dir)
    dims, eng = model.dims, model.engine
    IMG, REF, EOS = (tok.convert_tokens_to_ids(t) for t in ('<IMG_CONTEXT>', aligned_token, '<|im_end|>'))
    img = image if isinstance(image, Image.Image) else Image.open(image).convert('RGB')
    rep = {'checkpoint': model_path, 'image': getattr(image, 'filename', None) or (image if isinstance(image, str) else '<PIL image>'),
           'boxes': len(boxes), 'question': question, 'max_new_tokens': max_new_tokens, 'repetition_penalty': penalty,
           'layers': {'vit': dims.vit_layers, 'resampler': dims.rs_depth, 'llm': dims.llm_layers}, 'vocab': dims.vocab}

    # ---- pixels: the reference's host pipeline for both sides (the GPU preprocessing is pinned bit-exactly to it elsewhere)
    page_px = preprocess.load_image(img).to(torch.bfloat16)
    arr = np.array(img)
    char_px = torch.cat([preprocess.load_image_2(Image.fromarray(arr[y1:y2, x1:x2])).to(torch.bfloat16) for x1, y1, x2, y2 in boxes])
    rep['tiles'] = {'page': int(page_px.shape[0]), 'characters': int(char_px.shape[0])}

    # ---- visual stage
    with torch.no_grad():
        t0 = time.time()
        o_page = vision.extract_feature(sd, page_px, dims.vit_layers)
        o_char = torch.cat([vision.extract_feature(sd, char_px[i:i + 16], dims.vit_layers) for i in range(0, char_px.shape[0], 16)])
        o_rs = calli_align.resampler_forward(sd, o_char, dims.rs_depth)
        o_idx, o_cos = calli_align.vq_cos_sim(sd['normed_emb.weight'], o_rs, use_dynamic_p=True)
        o_idx = o_idx.reshape(o_rs.shape[0], 3)
        o_ref, _ = calli_align.denormalise(o_rs, o_idx, sd['normed_emb.weight'], sd['calli.mu'], sd['calli.sigma'])
        rep['oracle_visual_s'] = round(time.time() - t0, 1)
    h_page = model.extract_feature(page_px.cuda())
    h_char = model.extract_feature(char_px.cuda())
    h_rs = eng.resample(h_char)
    h_idx = eng.vq(h_rs)
    h_ref = eng.denorm(h_rs, h_idx)
    torch.cuda.synchronize()
    same = (h_idx.cpu().reshape(-1, 3) == o_idx)
    rep['visual'] = {'page_features_rel_l2': rel_l2(h_page.float().cpu(), o_page.float()), 'char_features_rel_l2': rel_l2(h_char.float().cpu(), o_char.float()),
                     'resampler_rel_l2': rel_l2(h_rs.float().cpu(), o_rs.float()), 'vq_indices_equal': int(same.sum()), 'vq_indices': int(same.numel()),
                     'pseudo_tokens_rel_l2': rel_l2(h_ref.float().cpu(), o_ref.float())}
    if not bool(same.all()):
        # every differing index through the ONE VQ rule (oracle/calli_align.py: vq_tie_rule, the tests' rule): the oracle's similarities at the two rows at most
        # one bf16 step apart AND the HIP tiled GEMM's own similarities (HIP resampler row, normalised table) straddling that gap
        from callireader_amd import engine as E
        xn = torch.nn.functional.normalize(o_rs, p=2, dim=2)
        hn = torch.nn.functional.normalize(h_rs.cpu(), p=2, dim=2)
        tn = torch.nn.functional.normalize(sd['normed_emb.weight'], p=2, dim=1)
        diffs = []
        for t, q in (~same).nonzero().tolist():
            i_o, i_h = int(o_idx[t, q]), int(h_idx.reshape(-1, 3)[t, q])
            s = torch.matmul(xn[t, q], tn[[i_o, i_h]].t()).float()
            rows = torch.stack([tn[i_o], tn[i_h]] + [tn[(i_o + 1 + k) % tn.shape[0]] for k in range(14)])
            sh = E.op_gemm(0, hn[t, q][None].expand(16, -1).contiguous().cuda(), rows.cuda(), kernel=1).float().cpu()[0]
            ok, gap, step = calli_align.vq_tie_rule(s[0], s[1], sh[0], sh[1])
            diffs.append({'tile': t, 'query': q, 'oracle_id': i_o, 'hip_id': i_h, 'oracle_gap': gap, 'one_bf16_step': step,
                          'hip_similarities': [float(sh[0]), float(sh[1])], 'measured_tie': bool(ok)})
        rep['visual']['vq_differences'] = diffs
        rep['visual']['vq_differences_all_measured_ties'] = all(d['measured_tie'] for d in diffs)
    log(f"visual stage: page features rel-L2 {rep['visual']['page_features_rel_l2']:.3e}, character features {rep['visual']['char_features_rel_l2']:.3e}, "
        f"resampler {rep['visual']['resampler_rel_l2']:.3e}, VQ indices {rep['visual']['vq_indices_equal']}/{rep['visual']['vq_indices']} equal, "
        f"pseudo tokens {rep['visual']['pseudo_tokens_rel_l2']:.3e}")

    # ---- prompt (chat_ocr's assembly, modeling_internvl_chat.py:690-726) and splice, each side from its own embeddings
    q = '<image>\n' + question + aligned_token * int(o_ref.shape[0])
    tpl = get_conv_template('internlm2-chat')
    tpl.append_message(tpl.roles[0], q)
    tpl.append_message(tpl.roles[1], None)
    query = tpl.get_prompt().replace('<image>', '<img>' + '<IMG_CONTEXT>' * 256 * page_px.shape[0] + '</img>', 1)
    ids = tok(query, return_tensors='pt')['input_ids']
    rep['prompt_tokens'] = int(ids.shape[1])
    o_emb = generate.splice_embeddings(sd, ids, o_page, o_ref, IMG, REF)
    h_emb = eng.embed_splice(ids[0].cuda(), h_page, h_ref.reshape(-1, 3, dims.llm_hidden), img_id=IMG, ref_id=REF)
    rep['spliced_embeddings_rel_l2'] = rel_l2(h_emb.float().cpu().reshape(-1), o_emb.float().reshape(-1))

    # ---- the oracle's free-running greedy stream (with the logits of every step)
    with torch.no_grad():
        t0 = time.time()
        o_ids, o_logits = generate.greedy_generate(sd, dims.llm_layers, o_emb, max_new_tokens=max_new_tokens, eos_token_id=EOS,
                                                   repetition_penalty=penalty, return_logits=True)
        rep['oracle_llm_s'] = round(time.time() - t0, 1)
    o_ids = o_ids[0].tolist()
    from oracle.generate import apply_repetition_penalty
    margins = []
    for t, lg in enumerate(o_logits):
        top2 = torch.topk(apply_repetition_penalty(lg.float(), o_ids[:t], penalty), 2).values
        margins.append(float(top2[0] - top2[1]))
    rep['oracle'] = {'ids': o_ids, 'text': tok.batch_decode(torch.tensor([o_ids]), skip_special_tokens=True)[0].split('<|im_end|>')[0].strip(),
                     'min_margin': min(margins), 'median_margin': sorted(margins)[len(margins) // 2]}

    def hip_stream(label):
        """free-running HIP ids from the HIP embeddings; on a divergence, the teacher-forced HIP logits of that step"""
        kv = eng.kv_alloc(1, int(ids.shape[1]) + max_new_tokens + 8)
        lg0 = eng.prefill(kv, 0, h_emb, penalty=penalty, want_logits=True).float().cpu().reshape(-1)
        for _ in range(len(o_ids) - 1):
            if kv.generated(0)[-1] == EOS:
                break
            eng.decode(kv, [0], penalty=penalty)
        torch.cuda.synchronize()
        got = kv.generated(0)
        if EOS in got:
            got = got[:got.index(EOS) + 1]
        out = {'ids': got, 'identical': got == o_ids, 'prefill_logits_rel_l2': rel_l2(lg0, o_logits[0].float()),
               'prefill_logits_max_abs': float((lg0 - o_logits[0].float()).abs().max())}
        d = first_divergence(got, o_ids)
        out['first_divergence'] = d
        if d is not None and d < min(len(got), len(o_ids)):
            kv.reset()
            lg = eng.prefill(kv, 0, h_emb, penalty=penalty, want_logits=True)
            for t in range(d):
                lg = eng.decode(kv, [0], penalty=penalty, force_tokens=torch.tensor([o_ids[t]]), want_logits=True)
            torch.cuda.synchronize()
            out['at_divergence'] = dict(judge_divergence(o_logits[d], lg.float().cpu().reshape(-1), o_ids[d], got[d], o_ids[:d], penalty),
                                        oracle_id=o_ids[d], hip_id=got[d], oracle_margin_top2=margins[d])
        elif d is not None:
            out['at_divergence'] = {'excusable': False, 'note': 'one stream stopped earlier than the other'}
        kv.free()
        log(f"{label}: prefill logits rel-L2 {out['prefill_logits_rel_l2']:.3e} (max |d| {out['prefill_logits_max_abs']:.4f}); "
            + ('ids identical to the oracle\'s (' + str(len(got)) + ' tokens)' if out['identical'] else
               f"streams part at token {d}: {json.dumps(out.get('at_divergence'))}"))
        return out

    rep['bf16'] = hip_stream('bf16')
    if fp8:
        for level in (1, 2):
            eng.enable_fp8_mfma(True, level=level)
            if level == 2:
                eng.enable_fp8_decode(True)
            # the fp8 switches act on the visual stage too: redo it, then the stream
            h_page8, h_char8 = model.extract_feature(page_px.cuda()), model.extract_feature(char_px.cuda())
            h_rs8 = eng.resample(h_char8)
            h_idx8 = eng.vq(h_rs8)
            h_ref8 = eng.denorm(h_rs8, h_idx8)
            keep = h_emb
            h_emb = eng.embed_splice(ids[0].cuda(), h_page8, h_ref8.reshape(-1, 3, dims.llm_hidden), img_id=IMG, ref_id=REF)
            r8 = hip_stream(f'fp8 level {level}' + (' + fp8 decode' if level == 2 else ''))
            r8['char_features_rel_l2'] = rel_l2(h_char8.float().cpu(), o_char.float())
            r8['vq_indices_equal'] = int((h_idx8.cpu().reshape(-1, 3) == o_idx).sum())
            rep[f'fp8_level{level}'] = r8
            h_emb = keep
            eng.enable_fp8_mfma(False)
            eng.enable_fp8_decode(False)
    rep['wall_s'] = round(time.time() - t_all, 1)
    b = rep['bf16']
    status = 0 if (b['identical'] or b.get('at_divergence', {}).get('excusable')) else 1
    if not rep['visual'].get('vq_differences_all_measured_ties', True):
        status = 1                                   # a VQ index that differs without a measured tie is a divergence too (vq_tie_rule)
    rep['verdict'] = ('token-exact' if b['identical'] else
                      ('streams part at a measured tie within one bf16 step' if status == 0 else 'DIVERGENCE at a margin the arithmetic cannot excuse'))
    log(f"verdict: {rep['verdict']} (oracle margins: min {rep['oracle']['min_margin']:.4f}, median {rep['oracle']['median_margin']:.4f})")
    return rep, status


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument('--model', default=os.environ.get('INTERNVL_PATH'), help='checkpoint directory (default: $INTERNVL_PATH)')
    ap.add_argument('--params', default='./params')
    ap.add_argument('--image', default=os.path.join(ROOT, 'tests', 'golden', 'example0.jpg'))
    ap.add_argument('--boxes', default=os.path.join(ROOT, 'tests', 'golden', 'example0_boxes.json'), help='labelme-style JSON (examples/0.json)')
    ap.add_argument('--question', default='这幅书法作品内容是什么？')
    ap.add_argument('--max-new-tokens', type=int, default=64)
    ap.add_argument('--repetition-penalty', type=float, default=1.0)
    ap.add_argument('--no-fp8', action='store_true')
    ap.add_argument('--threads', type=int, default=None, help='host threads for the oracle (default: torch\'s)')
    ap.add_argument('--out', default='real_checkpoint_parity.json')
    args = ap.parse_args(argv)
    if not args.model or not os.path.isdir(args.model):
        ap.error('give the checkpoint directory with --model or INTERNVL_PATH (config.json, safetensors shards, tokenizer files)')
    from callireader_amd import preprocess
    boxes = preprocess.boxes_from_labelme(json.load(open(args.boxes)))
    rep, status = run(args.model, args.params, args.image, boxes, args.question, args.max_new_tokens, args.repetition_penalty, not args.no_fp8, args.threads)
    with open(args.out, 'w', encoding='utf-8') as f:
        json.dump(rep, f, ensure_ascii=False, indent=1)
    print(f'report written to {args.out}; exit status {status}')
    return status


if __name__ == '__main__':
    sys.exit(main())
