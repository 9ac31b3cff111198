#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own modules (build container only).

Imports /root/reference (read-only) with three import-time stubs (timm DropPath,
cv2, torchvision.transforms — SURVEY.md 8c recipe), loads seeded weights from
`callireader_amd.synthetic` into the reference nn.Modules, runs them on CPU and
stores small samples of the outputs.  Only DATA (inputs are re-derivable from
seeds; expected outputs are sub-sampled tensors) is written — no reference
source travels.  tests/test_oracle_golden.py then checks `oracle/` against
these vectors, which pins the oracle to the reference.

Usage:  python scripts/make_golden.py            (takes ~2-4 min on 8 cores)
"""
import os
import sys
import types
import json

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = '/root/reference'
OUT = os.path.join(ROOT, 'tests', 'golden')


def install_stubs():
    import transformers.modeling_utils  # noqa: F401  (must be imported BEFORE the torchvision stub)
    timm = types.ModuleType('timm')
    timm_models = types.ModuleType('timm.models')
    timm_layers = types.ModuleType('timm.models.layers')

    class DropPath(nn.Identity):
        def __init__(self, p=0.0):
            super().__init__()
    timm_layers.DropPath = DropPath
    sys.modules.update({'timm': timm, 'timm.models': timm_models, 'timm.models.layers': timm_layers})
    for name in ['cv2', 'torchvision', 'torchvision.transforms', 'torchvision.transforms.functional',
                 'ultralytics', 'opencc', 'Levenshtein']:
        if name not in sys.modules:
            m = types.ModuleType(name)
            sys.modules[name] = m
    tv = sys.modules['torchvision']
    tv.transforms = sys.modules['torchvision.transforms']
    tv.transforms.functional = sys.modules['torchvision.transforms.functional']
    sys.modules['torchvision.transforms.functional'].InterpolationMode = types.SimpleNamespace(BICUBIC=3)
    sys.modules['torchvision.transforms'].InterpolationMode = types.SimpleNamespace(BICUBIC=3)
    sys.modules['ultralytics'].YOLO = object
    sys.path.insert(0, REF)


def sample(t, n=4096):
    """Deterministic strided sub-sample + fp64 checksums of a tensor."""
    f = t.detach().float().reshape(-1)
    step = max(1, f.numel() // n)
    return {'sample': f[::step][:n].numpy().astype(np.float32), 'step': np.int64(step),
            'numel': np.int64(f.numel()), 'sum': np.float64(f.double().sum().item()),
            'abssum': np.float64(f.double().abs().sum().item())}


def flat(prefix, d):
    return {f'{prefix}.{k}': v for k, v in d.items()}


def main():
    os.makedirs(OUT, exist_ok=True)
    install_stubs()
    torch.manual_seed(0)
    torch.set_num_threads(os.cpu_count())
    from callireader_amd.config import ModelDims
    from callireader_amd import synthetic

    cfg = json.load(open(os.path.join(REF, 'InternVL', 'config.json')))
    dims = ModelDims.reduced(vit_layers=2, llm_layers=2, rs_depth=2)
    gold = {}
    meta = {'seed': 0, 'dims': dims.asdict(), 'torch': torch.__version__}

    # ---------------- ViT (G1-G3) ----------------
    from InternVL.configuration_intern_vit import InternVisionConfig
    from InternVL.modeling_intern_vit import InternVisionModel
    vcfg = dict(cfg['vision_config'])
    vcfg['num_hidden_layers'] = dims.vit_layers
    vcfg['use_flash_attn'] = False
    vit = InternVisionModel(InternVisionConfig(**vcfg)).to(torch.bfloat16).eval()
    sd = synthetic.make_state_dict(dims, parts=('vit', 'mlp1'), seed=0)
    vit_sd = {k[len('vision_model.'):]: v for k, v in sd.items() if k.startswith('vision_model.')}
    missing = vit.load_state_dict(vit_sd, strict=True)
    print('vit load', missing)
    px = synthetic.make_pixels(2, seed=1)
    with torch.no_grad():
        emb = vit.embeddings(px)
        gold.update(flat('vit_embeddings', sample(emb)))
        lay0 = vit.encoder.layers[0](emb)
        gold.update(flat('vit_layer0', sample(lay0)))
        last = vit(pixel_values=px, output_hidden_states=False, return_dict=True).last_hidden_state
        gold.update(flat('vit_last', sample(last)))
        # extract_feature, called unbound on a namespace as SURVEY 8c describes
        from InternVL.modeling_internvl_chat import InternVLChatModel
        mlp1 = nn.Sequential(nn.LayerNorm(4096), nn.Linear(4096, 4096), nn.GELU(), nn.Linear(4096, 4096)).to(torch.bfloat16)
        mlp1.load_state_dict({k[len('mlp1.'):]: v for k, v in sd.items() if k.startswith('mlp1.')})
        ns = types.SimpleNamespace(vision_model=vit, mlp1=mlp1, select_layer=-1, downsample_ratio=0.5, ps_version='v2')
        ns.pixel_shuffle = lambda x, scale_factor=0.5: InternVLChatModel.pixel_shuffle(ns, x, scale_factor)
        feat = InternVLChatModel.extract_feature(ns, px)
        gold.update(flat('extract_feature', sample(feat)))
        # pixel_shuffle on a known integer pattern (G10-like index check)
        pat = torch.arange(2 * 4 * 4 * 8, dtype=torch.float32).reshape(2, 4, 4, 8)
        gold['pixel_shuffle.pattern_out'] = InternVLChatModel.pixel_shuffle(ns, pat, 0.5).numpy()
    del vit, mlp1

    # ---------------- Resampler (G4) ----------------
    from models.perceiver_resampler import PerceiverResampler
    rs = PerceiverResampler(dim=4096, depth=dims.rs_depth).to(torch.bfloat16).eval()
    rsd = synthetic.make_state_dict(dims, parts=('resampler',), seed=0)
    rs.load_state_dict({k[len('resampler.'):]: v for k, v in rsd.items()}, strict=True)
    with torch.no_grad():
        out = rs(feat)
        gold.update(flat('resampler', sample(out)))
        gold['resampler.full'] = out.float().numpy()        # (2,3,4096): small enough to keep whole
    del rs

    # ---------------- VQ (G5) ----------------
    from models.similarity import vq_cos_sim
    vdims = ModelDims.reduced(vocab=4096)
    vsd = synthetic.make_state_dict(vdims, parts=('vq',), seed=0)
    table = nn.Embedding(4096, 4096).to(torch.bfloat16)
    table.weight.data.copy_(vsd['normed_emb.weight'])
    with torch.no_grad():
        # plant exact neighbours so the argmax is not at the mercy of bf16 ties
        q = out.clone()
        q[0, 0] = vsd['normed_emb.weight'][123] * 3.0
        q[1, 2] = vsd['normed_emb.weight'][4000] * 0.5
        idx, cosv = vq_cos_sim(table, q, use_dynamic_p=True)
        gold['vq.indices'] = idx.numpy().astype(np.int64)
        gold['vq.cos'] = cosv.float().numpy()
        idx1 = vq_cos_sim(table, q[:1])                     # T == 1 -> shape (3,)  (similarity.py:27 squeeze)
        gold['vq.indices_T1'] = idx1.numpy().astype(np.int64)

    # ---------------- InternLM2 (G7-G9) ----------------
    from InternVL.configuration_internlm2 import InternLM2Config
    from InternVL.modeling_internlm2 import InternLM2ForCausalLM
    lcfg = dict(cfg['llm_config'])
    lcfg['num_hidden_layers'] = dims.llm_layers
    lcfg['attn_implementation'] = 'eager'
    ldims = ModelDims.reduced(llm_layers=2, vocab=8192)     # reduced vocab keeps the fixture run short
    lcfg['vocab_size'] = ldims.vocab
    llm = InternLM2ForCausalLM(InternLM2Config(**lcfg)).to(torch.bfloat16).eval()
    lsd = synthetic.make_state_dict(ldims, parts=('llm',), seed=0)
    llm.load_state_dict({k[len('language_model.'):]: v for k, v in lsd.items()}, strict=True)
    rot = llm.model.layers[0].attention.rotary_emb
    rows = [0, 1, 1023, 3163, 32767]
    gold['rope.rows'] = np.array(rows)
    gold['rope.cos'] = rot.cos_cached[rows].to(torch.bfloat16).float().numpy()
    gold['rope.sin'] = rot.sin_cached[rows].to(torch.bfloat16).float().numpy()
    for S in (17, 300):
        g = torch.Generator().manual_seed(100 + S)
        emb = (torch.randn(1, S, 4096, generator=g) * 0.02).to(torch.bfloat16)
        with torch.no_grad():
            o = llm(inputs_embeds=emb, use_cache=True, return_dict=True)
            logits = o.logits
            past = o.past_key_values
            gold[f'llm.S{S}.last_logits'] = logits[0, -1].numpy()
            gold[f'llm.S{S}.k0_last'] = past[0][0][0, :, -1, :].float().numpy()
            gold[f'llm.S{S}.v1_first'] = past[1][1][0, :, 0, :].float().numpy()
            toks = []
            step_logits = []
            for _ in range(8):
                nxt = int(torch.argmax(logits[0, -1]))
                toks.append(nxt)
                pos = torch.tensor([[past[0][0].shape[2]]])
                o = llm(input_ids=torch.tensor([[nxt]]), past_key_values=past, position_ids=pos,
                        use_cache=True, return_dict=True)
                logits, past = o.logits, o.past_key_values
                step_logits.append(logits[0, -1].numpy())
            gold[f'llm.S{S}.greedy_tokens'] = np.array(toks, dtype=np.int64)
            gold[f'llm.S{S}.step_logits_sample'] = np.stack(step_logits)[:, ::64]
    np.savez_compressed(os.path.join(OUT, 'reference_vectors.npz'), **gold)
    json.dump(meta, open(os.path.join(OUT, 'reference_vectors.json'), 'w'), indent=1)
    sz = os.path.getsize(os.path.join(OUT, 'reference_vectors.npz'))
    print('wrote', len(gold), 'arrays,', sz, 'bytes')


if __name__ == '__main__':
    main()
