"""Oracle: embedding splice + greedy generation loop (TEST INFRASTRUCTURE ONLY).

PARITY UNPINNED for the loop itself: the reference delegates it to
transformers==4.45.2 (requirements.txt:76) `GenerationMixin.generate -> _sample`
with `RepetitionPenaltyLogitsProcessor`, `EosTokenCriteria`, `MaxLengthCriteria`;
that version is not installed here and the installed 5.x cannot drive the
reference model.  This file restates the published 4.45.2
semantics for the exact arguments the reference passes:
  inference.py:92-96                       num_beams=1, max_new_tokens=1024, do_sample=False
  modeling_internvl_chat.py:709,732        eos_token_id = id('<|im_end|>') = 92542 (kwarg overrides config)
  modeling_internvl_chat.py:741,1111-1120  inputs_embeds=..., use_cache=True, repetition_penalty=...
Semantics:
  * only inputs_embeds is given, so generate() starts from an EMPTY input_ids
    and returns only the new tokens (EOS included when hit);
  * each step: logits[:, -1, :] (already fp32, modeling_internlm2.py:1082);
    RepetitionPenaltyLogitsProcessor (only when penalty != 1.0): for every id
    already in input_ids (= generated so far), score = score/penalty if score>0
    else score*penalty; next = argmax (first max wins);
  * stop when next == eos or max_new_tokens tokens were produced;
  * subsequent steps feed the sampled id through tok_embeddings with the tuple
    cache (prepare_inputs_for_generation, modeling_internlm2.py:1112-1149);
    position = cache length (attention_mask all ones -> cumsum-1, :1130).
"""
import torch
import torch.nn.functional as F

from . import internlm2


def splice_embeddings(sd, input_ids, vit_embeds=None, reference_embeds=None,
                      img_context_token_id=92546, aligned_token_id=92537):
    """generate_ocr head, modeling_internvl_chat.py:1081-1107 (generate_origin :1033-1052 w/o reference_embeds)."""
    emb = F.embedding(input_ids, sd['language_model.model.tok_embeddings.weight'])      # :1087
    B, N, C = emb.shape
    emb = emb.reshape(B * N, C).clone()
    ids = input_ids.reshape(B * N)
    if vit_embeds is not None:
        selected = ids == img_context_token_id                                           # :1094
        assert selected.sum() != 0                                                       # :1095
        emb[selected] = vit_embeds.reshape(-1, C).to(emb.dtype)                          # :1096
        if reference_embeds is not None:
            selected = ids == aligned_token_id                                           # :1100
            assert selected.sum() != 0                                                   # :1101
            emb[selected] = reference_embeds.reshape(-1, C).to(emb.dtype)                # :1102
    return emb.reshape(B, N, C)


def apply_repetition_penalty(scores, generated, penalty):
    """transformers 4.45.2 RepetitionPenaltyLogitsProcessor.__call__ (published algorithm)."""
    if penalty == 1.0 or len(generated) == 0:
        return scores
    ids = torch.tensor(sorted(set(generated)), dtype=torch.long)
    s = scores[ids]
    scores = scores.clone()
    scores[ids] = torch.where(s < 0, s * penalty, s / penalty)
    return scores


def near_tie_straddles(ref_logits, hip_logits, ref_id, hip_id, generated, penalty, atol):
    """The only excuse a differing greedy pick has (checker rule shared by tests/test_gpu_llm.py, tests/test_gpu_full_depth.py and
    __graft_entry__.smoke): the oracle's gap between its pick and the other one, on the PROCESSED scores, must be covered by the logit
    differences measured at exactly those two ids (the other implementation's scores straddle) and stay inside `atol`.  "Straddle" is meant
    literally (round-4 advice): the other implementation's processed scores must favour ITS pick, sg[hip_id] >= sg[ref_id], i.e.
    gap <= d_hip - d_ref with the signs -- two shifts in the same direction that leave the oracle's id ahead explain nothing.
    Returns (ok, gap, d_ref_id, d_hip_id)."""
    sc = apply_repetition_penalty(ref_logits.float(), generated, penalty)
    sg = apply_repetition_penalty(hip_logits.float(), generated, penalty)
    gap = float(sc[ref_id] - sc[hip_id])
    d_ref, d_hip = float(sg[ref_id] - sc[ref_id]), float(sg[hip_id] - sc[hip_id])
    straddles = float(sg[hip_id]) >= float(sg[ref_id]) - 1e-6
    return (straddles and gap <= d_hip - d_ref + 1e-6 and gap <= atol), gap, d_ref, d_hip


def greedy_generate(sd, n_layers, inputs_embeds, max_new_tokens=1024, eos_token_id=92542,
                    repetition_penalty=1.0, n_heads=32, n_kv=8, return_logits=False):
    """Greedy loop over `internlm2.model_forward` for ONE sequence (B == 1)."""
    assert inputs_embeds.shape[0] == 1
    rope = internlm2.rope_tables(inputs_embeds.shape[-1] // n_heads)
    logits, past = internlm2.model_forward(sd, n_layers, inputs_embeds=inputs_embeds, rope=rope,
                                           n_heads=n_heads, n_kv=n_kv, all_logits=False)
    out, all_logits = [], []
    while True:
        row = logits[0, -1, :]
        if return_logits:
            all_logits.append(row.clone())
        row = apply_repetition_penalty(row, out, repetition_penalty)
        nxt = int(torch.argmax(row))
        out.append(nxt)
        if nxt == eos_token_id or len(out) >= max_new_tokens:
            break
        logits, past = internlm2.model_forward(sd, n_layers, input_ids=torch.tensor([[nxt]]), past=past,
                                               rope=rope, n_heads=n_heads, n_kv=n_kv)
    ids = torch.tensor([out], dtype=torch.long)
    return (ids, all_logits) if return_logits else ids
