"""Host-side mirror of the reference's InternVLChatModel for the image->text path.

Same method names, argument meaning, return values and error behaviour as
/root/reference/InternVL/modeling_internvl_chat.py (extract_feature :299, calli_align :322,
chat_ocr :649, batch_chat :903, chat :955, generate_origin :1021, generate_ocr :1067), so a
caller of the reference (inference.py:37-57, evaluate.py) can switch classes.  All tensor
math runs in libcallireader_hip.so through `Engine`; this file only builds prompts, moves
ids and pixels, and drives the greedy loop.  There is no PyTorch compute fallback.

`detect_model` is what the reference passes (inference.py:37-42,98): an ultralytics `YOLO` object (or any callable
returning raw boxes); with the sorter loaded (`params/orderformer.pth`, as the reference's __init__ :159 does) the
whole front end of calli_align runs on its detections (ordering.py).  The detector NETWORK itself is third-party
(ultralytics) and is not re-implemented; `boxes=` bypasses it with ordered boxes (examples/0.json).
"""
import json
import os
import sys
import time

import numpy as np
import torch
from PIL import Image

from .config import ModelDims, IMG_CONTEXT_TOKEN_ID, ALIGNED_TOKEN_ID, EOS_TOKEN_ID
from .conversation import get_conv_template
from .engine import Engine
from .preprocess import load_image, load_image_2, plan_page, plan_chars_array


class InternVLChatModel:
    main_input_name = 'pixel_values'

    def __init__(self, dims: ModelDims = None, device=0, max_tokens=8192, max_pages=1):
        self.dims = dims or ModelDims.full()
        self.engine = Engine(self.dims, device=device, max_pos=min(self.dims.max_pos, max(max_tokens, 64)))
        self.device = self.engine.device
        self.template = 'internlm2-chat'
        self.num_image_token = self.dims.tokens_per_tile                    # :146
        self.downsample_ratio = self.dims.downsample_ratio
        self.ps_version = 'v2'
        self.select_layer = -1
        self.img_context_token_id = None                                    # :192
        self.aligned_token_id = ALIGNED_TOKEN_ID                            # 92537, hard-coded at :1100
        self.conv_template = get_conv_template(self.template)
        self.system_message = self.conv_template.system_message            # :194
        self.max_tokens = max_tokens
        self.max_pages = max_pages
        self.sorter = None                 # ordering.OrderFormer once load_orderformer() has run (modeling_internvl_chat.py:159)
        self.gpu_preprocess = True       # tiles are cut/resized/normalised on the GPU (bit-identical to the PIL path)
        self._kv = None
        self._ready = False

    # ---- construction --------------------------------------------------------------------------
    @classmethod
    def from_state_dict(cls, sd, dims=None, **kw):
        m = cls(dims, **kw)
        m.engine.load_state_dict(sd)
        m._finish()
        return m

    @classmethod
    def from_synthetic(cls, dims=None, seed=0, parts=('vit', 'mlp1', 'resampler', 'vq', 'llm'), outlier_shift=0, **kw):
        """Seeded random weights of the checkpoint's architecture, drawn on the GPU tensor by tensor (outlier_shift: synthetic.outlier_transform)."""
        from . import synthetic
        m = cls(dims, **kw)
        for k, v in synthetic.iter_state_dict(m.dims, parts=parts, seed=seed, device=m.device, outlier_shift=outlier_shift):
            m.engine.load_weight(k, v)
            del v
        m._finish()
        return m

    @classmethod
    def from_pretrained(cls, path, *model_args, params_dir='./params', torch_dtype=torch.bfloat16, config=None, **kw):
        """HF sharded safetensors (model.safetensors.index.json) + params/gauss_norm_mu_sigma.pth, as
        AutoModel.from_pretrained(INTERNVL_PATH, torch_dtype=torch.bfloat16, low_cpu_mem_usage=True,
        trust_remote_code=True) does in inference.py:85-89 / __init__ :153-159.  Shapes come from <path>/config.json.
        Keyword arguments of the HF loader that mean nothing here (low_cpu_mem_usage, trust_remote_code, revision,
        code_revision, ...) are accepted and ignored, so the reference's call works unchanged through the auto_map shim
        (callireader_amd/hf_entry/)."""
        from .weights import load_checkpoint
        if torch_dtype not in (torch.bfloat16, None, 'auto', 'bfloat16'):
            raise ValueError('the engine computes in bf16, the dtype the reference loads the model in (inference.py:87)')
        ours = {k: kw.pop(k) for k in ('device', 'max_tokens', 'max_pages') if k in kw}
        dims = kw.pop('dims', None)
        if dims is None:
            dims = ModelDims.from_hf_config(path) if os.path.exists(os.path.join(path, 'config.json')) else ModelDims.full()
        m = cls(dims, **ours)
        load_checkpoint(m.engine, path, params_dir)
        m._finish()
        of = os.path.join(params_dir, 'orderformer.pth')                    # :159 ORDERFORMER_CHECKPOINT
        if os.path.exists(of):
            m.load_orderformer(torch.load(of, map_location='cpu', weights_only=True))
        return m

    # hooks transformers' AutoModel.from_pretrained(..., trust_remote_code=True) calls on the class auto_map names
    @classmethod
    def register_for_auto_class(cls, auto_class='AutoModel'):
        cls._auto_class = auto_class

    _auto_class = None

    def _finish(self):
        self.engine.load_rope()
        self.engine.finalize()
        self._ready = True

    def eval(self):
        return self

    def cuda(self):
        return self

    def close(self):
        """Release what the model holds on the device and the host: the page feeder (its decode threads, stream and context), the KV cache, the engine's context."""
        f = getattr(self, '_pagefeeder', None)
        if f is not None:
            f.close()
            self._pagefeeder = None
        if self._kv is not None:
            self._kv.free()
            self._kv = None
        self.engine.close()

    def kv(self, n_seqs=None):
        """The model's own KV cache (the pipeline of chat_ocr_stream has two of its own): `n_seqs` sequences when none exists yet (default: max_pages).  The one-page
        calls ask for ONE sequence -- 1 GB at max_tokens 8192, where a 64-page cache is 69 GB that folder mode's per-image fallback would hold beside the pipeline's."""
        if self._kv is None:
            self._kv = self.engine.kv_alloc(n_seqs or self.max_pages, self.max_tokens)
        return self._kv

    # ---- the * stages --------------------------------------------------------------------------
    def extract_feature(self, pixel_values):
        """:299-319  (T,3,448,448) -> (T,256,4096)"""
        return self.engine.extract_feature(pixel_values)

    def resampler(self, image_embeddings):
        """models/perceiver_resampler.py:81-100  (T,256,4096) -> (T,3,4096)"""
        return self.engine.resample(image_embeddings)

    @staticmethod
    def pixel_shuffle(x, scale_factor=0.5):
        """:283-297, ps_version 'v2': (N, W, H, C) -> (N, W * s, H * s, C / s^2), a pure re-indexing.  Kept for callers of the reference's method; extract_feature does
        not call it (csrc/vision.hip reads the ViT output in this order while it normalises the rows for mlp1)."""
        n, w, h, c = x.shape
        hs, ws = int(h * scale_factor), int(w * scale_factor)
        x = x.reshape(n, w, hs, int(c / scale_factor)).transpose(1, 2)                    # fold 1 / s of H into the channels
        x = x.reshape(n, hs, ws, int(c / (scale_factor * scale_factor)))                  # ... then 1 / s of W
        return x.transpose(1, 2).contiguous()

    @staticmethod
    def find_coordinates(text):
        """:642-648: every run of digits in the question, as ints (region_wise: x1, x2, y1, y2)."""
        import re
        return [int(n) for n in re.findall(r'\d+', text)]

    def align_tiles(self, pixel_values, drop_zero=False, use_hard_vector_quant=False, verbose=False):
        """The tile path of calli_align after the boxes are known (:587-640):
        extract_feature -> resampler -> vq_cos_sim -> (hard VQ) -> (drop_zero) -> sigma/mu de-normalisation.
        Returns (back_to_origin_flat (n,4096), indices (T,3))."""
        st = time.time()
        image_embeddings = self.extract_feature(pixel_values)
        output = self.resampler(image_embeddings)
        outs = self.engine.vq(output, with_cos=use_hard_vector_quant)
        if use_hard_vector_quant:
            indices, cos_sim_values = outs
            print('Dynamic vector quantization...')                          # :610
        else:
            indices, cos_sim_values = outs, None
        back = self.engine.denorm(output, indices, cos_sim_values, drop_zero=drop_zero, hard_vq=use_hard_vector_quant)
        if verbose:
            torch.cuda.synchronize()
            print(f'extract feat + resampler + vq {time.time() - st:.2f}s')
        # the reference returns vq_cos_sim's squeezed indices: shape (3,) for a single tile (similarity.py:27)
        return back, (indices.squeeze(0) if indices.shape[0] == 1 else indices)

    def calli_align(self, img_path, detect_model, drop_zero=False, use_hard_vector_quant=False, save_path=None,
                    verbose=False, boxes=None):
        """:322-640.  Boxes come from `boxes=` (reading order), or from `detect_model`:
          * with a sorter loaded (`load_orderformer`): `detect_model(image array)` returns RAW detections in any order
            and the reference's front end runs on them -- repeated detection passes, duplicate removal, column merge /
            area split, OrderFormer, per-column top-to-bottom (ordering.py, :346-556);
          * without one: `detect_model(PIL image)` must return the boxes already in reading order.
        The detector network itself (YOLOv10 through ultralytics) is third-party and stays outside this package."""
        if img_path is None:
            return None, None                                               # :554-555
        img = Image.open(img_path).convert('RGB') if isinstance(img_path, str) else img_path.convert('RGB')
        if boxes is None:
            from . import ordering
            boxes = ordering.acquire_boxes(detect_model, img, self.sorter)      # :346-394, :558 (one helper for chat_ocr and the page batches)
        if len(boxes) == 0:
            # the reference reaches torch.cat([]) at :585 with no box on the page
            raise RuntimeError('calli_align: no character box on the page (the reference fails here too: torch.cat() of an empty list, modeling_internvl_chat.py:585)')
        arr = np.array(img)
        if self.gpu_preprocess:
            # one page upload, every crop resized/padded/normalised by cr_preprocess (replaces the per-box PIL loop :580-583)
            h, w = arr.shape[:2]
            jobs = plan_chars_array(boxes, w, h)                                          # clipped to the page, as numpy slicing clips the reference's crops
            results = self.engine.preprocess(torch.from_numpy(arr), jobs, len(jobs))
        else:
            tiles = []
            for xyxy in boxes:                                              # :580-583
                x1, y1, x2, y2 = int(xyxy[0]), int(xyxy[1]), int(xyxy[2]), int(xyxy[3])
                tiles.append(load_image_2(Image.fromarray(arr[y1:y2, x1:x2])).to(torch.bfloat16))
            results = torch.cat(tiles).to(self.device)                      # :585
        return self.align_tiles(results, drop_zero, use_hard_vector_quant, verbose)

    def load_orderformer(self, state_dict, max_nums=50):
        """:159, models/model.py:530-552: `state_dict` = params/orderformer.pth (the reference's `Transformer` keys)."""
        from .ordering import OrderFormer
        self.sorter = OrderFormer.from_state_dict(self.engine, state_dict, max_nums=max_nums)
        return self.sorter

    # ---- generation ----------------------------------------------------------------------------
    def _greedy(self, input_embeds, max_new_tokens, eos_token_id, repetition_penalty, check_every=16):
        """transformers 4.45.2 GenerationMixin._sample with do_sample=False, num_beams=1, for ONE sequence:
        returns only the new ids, EOS included (oracle/generate.py documents the semantics)."""
        kv = self.kv(1)
        kv.reset(0)
        if input_embeds.shape[0] + max_new_tokens > self.max_tokens:
            raise ValueError(f'prompt of {input_embeds.shape[0]} tokens + {max_new_tokens} new exceeds max_tokens={self.max_tokens}')
        self.engine.prefill(kv, 0, input_embeds, penalty=repetition_penalty)
        n = 1
        while n < max_new_tokens:
            steps = min(check_every, max_new_tokens - n)
            if eos_token_id is not None and eos_token_id in kv.generated(0):
                break
            for _ in range(steps):
                self.engine.decode(kv, [0], penalty=repetition_penalty)
            n += steps
        ids = kv.generated(0)[:max_new_tokens]
        if eos_token_id is not None and eos_token_id in ids:
            ids = ids[:ids.index(eos_token_id) + 1]       # ids past EOS were speculative; the reference stops here
        return torch.tensor([ids], dtype=torch.long, device=self.device)

    @staticmethod
    def _gen_args(generate_kwargs):
        if generate_kwargs.get('num_beams', 1) != 1 or generate_kwargs.get('do_sample', False):
            raise NotImplementedError('the CalliReader path is greedy (inference.py:92-96): num_beams=1, do_sample=False')
        return generate_kwargs.get('max_new_tokens', 1024), generate_kwargs.get('eos_token_id', EOS_TOKEN_ID)

    @torch.no_grad()
    def generate_ocr(self, pixel_values=None, input_ids=None, attention_mask=None, visual_features=None,
                     generation_config=None, reference_embeds=None, output_hidden_states=None, return_dict=None,
                     repetition_penalty=1.5, **generate_kwargs):
        """:1067-1122"""
        assert self.img_context_token_id is not None                        # :1081
        if input_ids.shape[0] != 1:
            raise NotImplementedError('generate_ocr handles one page per call, as chat_ocr does')
        if pixel_values is not None:
            vit_embeds = visual_features if visual_features is not None else self.extract_feature(pixel_values)
            ids = input_ids.reshape(-1)
            assert (ids == self.img_context_token_id).sum() != 0            # :1095
            if reference_embeds is not None:
                assert (ids == self.aligned_token_id).sum() != 0            # :1101
            input_embeds = self.engine.embed_splice(ids, vit_embeds, reference_embeds,
                                                    img_id=self.img_context_token_id, ref_id=self.aligned_token_id)
        else:
            input_embeds = self.engine.embed_splice(input_ids.reshape(-1))  # :1107
        max_new, eos = self._gen_args(generate_kwargs)
        return self._greedy(input_embeds, max_new, eos, repetition_penalty)

    @torch.no_grad()
    def generate_origin(self, pixel_values=None, input_ids=None, attention_mask=None, visual_features=None,
                        generation_config=None, output_hidden_states=None, return_dict=None, **generate_kwargs):
        """:1021-1065 (no pseudo-tokens, no repetition penalty unless passed)"""
        penalty = generate_kwargs.pop('repetition_penalty', 1.0)
        return self.generate_ocr(pixel_values, input_ids, attention_mask, visual_features, generation_config, None,
                                 output_hidden_states, return_dict, repetition_penalty=penalty, **generate_kwargs)

    @torch.no_grad()
    def generate(self, pixel_values=None, input_ids=None, attention_mask=None, visual_features=None, generation_config=None,
                 output_hidden_states=None, return_dict=None, **generate_kwargs):
        """:1124-1183.  The CalliAlign-only path: every tile's 256 visual tokens are resampled to 3 pseudo-tokens,
        snapped to the normalised table, de-normalised (no drop_zero, no hard VQ) and spliced at the <IMG_CONTEXT>
        positions (3 per tile: dynamic_chat sets num_image_token = 3).  One row per call, like the other generate_*."""
        assert self.img_context_token_id is not None                        # :1137
        if input_ids.shape[0] != 1:
            raise NotImplementedError('generate handles one prompt per call; dynamic_chat(batch=True) loops over rows')
        penalty = generate_kwargs.pop('repetition_penalty', 1.0)
        if pixel_values is not None:
            vit_embeds = visual_features if visual_features is not None else self.extract_feature(pixel_values)   # :1139-1143
            vit_embeds = self.resampler(vit_embeds)                         # :1147
            indices = self.engine.vq(vit_embeds)                            # :1153
            vit_embeds = self.engine.denorm(vit_embeds, indices)            # :1156  x * sigma[idx] + mu[idx]
            ids = input_ids.reshape(-1)
            assert (ids == self.img_context_token_id).sum() != 0            # :1164
            input_embeds = self.engine.embed_splice(ids, vit_embeds, None, img_id=self.img_context_token_id)
        else:
            input_embeds = self.engine.embed_splice(input_ids.reshape(-1))  # :1172
        max_new, eos = self._gen_args(generate_kwargs)
        return self._greedy(input_embeds, max_new, eos, penalty)

    # ---- chat API ------------------------------------------------------------------------------
    def _build_query(self, question, history, num_patches_list, IMG_START_TOKEN, IMG_END_TOKEN, IMG_CONTEXT_TOKEN):
        template = get_conv_template(self.template)
        template.system_message = self.system_message
        history = [] if history is None else history
        for (old_question, old_answer) in history:
            template.append_message(template.roles[0], old_question)
            template.append_message(template.roles[1], old_answer)
        template.append_message(template.roles[0], question)
        template.append_message(template.roles[1], None)
        query = template.get_prompt()
        for num_patches in num_patches_list:
            image_tokens = IMG_START_TOKEN + IMG_CONTEXT_TOKEN * self.num_image_token * num_patches + IMG_END_TOKEN
            query = query.replace('<image>', image_tokens, 1)
        return query, template, history

    def chat(self, tokenizer, pixel_values, question, generation_config, history=None, return_history=False,
             num_patches_list=None, IMG_START_TOKEN='<img>', IMG_END_TOKEN='</img>', IMG_CONTEXT_TOKEN='<IMG_CONTEXT>',
             verbose=False):
        """:955-1018"""
        if history is None and pixel_values is not None and '<image>' not in question:
            question = '<image>\n' + question
        if num_patches_list is None:
            num_patches_list = [pixel_values.shape[0]] if pixel_values is not None else []
        assert pixel_values is None or len(pixel_values) == sum(num_patches_list)       # :965
        self.img_context_token_id = tokenizer.convert_tokens_to_ids(IMG_CONTEXT_TOKEN)
        query, template, history = self._build_query(question, history, num_patches_list, IMG_START_TOKEN,
                                                     IMG_END_TOKEN, IMG_CONTEXT_TOKEN)
        model_inputs = tokenizer(query, return_tensors='pt')
        generation_config = dict(generation_config)
        generation_config['eos_token_id'] = tokenizer.convert_tokens_to_ids(template.sep)
        out = self.generate_origin(pixel_values=pixel_values, input_ids=model_inputs['input_ids'],
                                   attention_mask=model_inputs['attention_mask'], **generation_config)
        response = tokenizer.batch_decode(out, skip_special_tokens=True)[0]
        response = response.split(template.sep)[0].strip()
        history.append((question, response))
        return (response, history) if return_history else response

    def chat_ocr(self, tokenizer, detect_model, img_path, questions, generation_config, num_patches_list=None,
                 history=None, return_history=False, IMG_START_TOKEN='<img>', IMG_END_TOKEN='</img>',
                 IMG_CONTEXT_TOKEN='<IMG_CONTEXT>', ALIGNED_TOKEN='[UNUSED_TOKEN_140]', verbose=False, image_counts=None,
                 batch=False, use_p=True, drop_zero=False, hard_vq=False, repetition_penalty=1.5, region_wise=False,
                 boxes=None):
        """:649-762"""
        pixel_values = None
        sub_img = None
        page = None                       # the decoded page, handed on to calli_align (the reference opens the file a second time there, :558: 14.5 ms of JPEG decode on the example page)
        if img_path is not None:
            try:
                if region_wise:
                    img = np.array(Image.open(img_path).convert('RGB'))
                    x1, x2, y1, y2 = self.find_coordinates(questions)                    # :661-663
                    sub_img = Image.fromarray(img[y1:y2, x1:x2])
                    questions = '输出图片中所有文字:'
                    pixel_values = load_image(sub_img).to(torch.bfloat16).to(self.device)
                elif self.gpu_preprocess:
                    page = Image.open(img_path).convert('RGB') if isinstance(img_path, str) else img_path.convert('RGB')
                    jobs, n = plan_page(*page.size)
                    pixel_values = self.engine.preprocess(torch.from_numpy(np.array(page)), jobs, n)
                else:
                    pixel_values = load_image(img_path).to(torch.bfloat16).to(self.device)
            except Exception:
                raise FileNotFoundError                                         # :670-671
        out_tokens = None
        if use_p:
            if region_wise:
                try:
                    out_tokens, indices = self.calli_align(sub_img, detect_model, drop_zero=drop_zero,
                                                           use_hard_vector_quant=hard_vq, verbose=verbose, boxes=boxes)
                except Exception:
                    return '检测失败'                                            # :676-679
            else:
                out_tokens, indices = self.calli_align(page if page is not None else img_path, detect_model, drop_zero=drop_zero,
                                                       use_hard_vector_quant=hard_vq, verbose=verbose, boxes=boxes)
        question = questions
        if pixel_values is not None and '<image>' not in questions:
            question = '<image>\n' + questions                                  # :690-691
        if history is None and use_p and ALIGNED_TOKEN not in question:
            question = question + ALIGNED_TOKEN * out_tokens.shape[0]           # :698-699
        if num_patches_list is None:
            num_patches_list = [pixel_values.shape[0]] if pixel_values is not None else []
        assert pixel_values is None or len(pixel_values) == sum(num_patches_list)          # :702
        self.img_context_token_id = tokenizer.convert_tokens_to_ids(IMG_CONTEXT_TOKEN)
        query, template, history = self._build_query(question, history, num_patches_list, IMG_START_TOKEN,
                                                     IMG_END_TOKEN, IMG_CONTEXT_TOKEN)
        model_inputs = tokenizer(query, return_tensors='pt')
        generation_config = dict(generation_config)
        generation_config['eos_token_id'] = tokenizer.convert_tokens_to_ids(template.sep)   # :709,732
        out = self.generate_ocr(pixel_values=pixel_values, input_ids=model_inputs['input_ids'],
                                attention_mask=model_inputs['attention_mask'],
                                reference_embeds=out_tokens if use_p else None,
                                repetition_penalty=repetition_penalty, **generation_config)
        response = tokenizer.batch_decode(out, skip_special_tokens=True)[0]
        response = response.split(template.sep)[0].strip()                      # :752-753
        history.append((question, response))
        return (response, history) if return_history else response

    def dynamic_chat(self, tokenizer, pixel_values, questions, generation_config, num_patches_list=None, history=None,
                     return_history=False, IMG_START_TOKEN='<img>', IMG_END_TOKEN='</img>', IMG_CONTEXT_TOKEN='<IMG_CONTEXT>',
                     verbose=False, image_counts=None, batch=False, use_p=True):
        """:765-901.  `use_p` switches every tile to 3 pseudo-tokens (and, as upstream, leaves num_image_token at 3 on
        the object afterwards).  batch=True: a list of questions, one answer each (the reference pads them into one
        batch; here every row runs at its own length and the rows decode together).  batch=False: the reference's
        hard-wired single-turn prompt (:857-866), history only feeds the returned list."""
        if use_p:
            self.num_image_token = 3                                        # :768-769
        gen = self.generate if use_p else self.generate_origin
        generation_config = dict(generation_config)
        self.img_context_token_id = tokenizer.convert_tokens_to_ids(IMG_CONTEXT_TOKEN)
        if batch:
            assert isinstance(questions, list) and len(questions) > 0 and isinstance(questions[0], str)
            if history is not None or return_history:
                print('Now multi-turn chat is not supported in batch_chat.')
                raise NotImplementedError
            if image_counts is not None:
                num_patches_list = image_counts
                print('Warning: `image_counts` is deprecated. Please use `num_patches_list` instead.')
            if verbose and pixel_values is not None:
                print(f'dynamic ViT batch size: {pixel_values.shape[0]}')
            template = get_conv_template(self.template)
            generation_config['eos_token_id'] = tokenizer.convert_tokens_to_ids(template.sep)
            responses, off = [], 0
            for idx, num_patches in enumerate(num_patches_list):
                question = questions[idx]
                if pixel_values is not None and '<image>' not in question:
                    question = '<image>\n' + question
                t = get_conv_template(self.template)                         # no system_message override here (:790)
                t.append_message(t.roles[0], question)
                t.append_message(t.roles[1], None)
                query = t.get_prompt().replace('<image>', IMG_START_TOKEN + IMG_CONTEXT_TOKEN * self.num_image_token * num_patches + IMG_END_TOKEN, 1)
                ids = tokenizer(query, return_tensors='pt')['input_ids']
                px = pixel_values[off:off + num_patches] if pixel_values is not None else None
                off += num_patches
                out = gen(pixel_values=px, input_ids=ids, **generation_config)
                responses.append(tokenizer.batch_decode(out, skip_special_tokens=True)[0].split(template.sep)[0].strip())
            return responses
        assert isinstance(questions, str)
        if num_patches_list is None:
            num_patches_list = [pixel_values.shape[0]] if pixel_values is not None else []
        assert pixel_values is None or len(pixel_values) == sum(num_patches_list)          # :829
        template = get_conv_template(self.template)
        template.system_message = self.system_message
        generation_config['eos_token_id'] = tokenizer.convert_tokens_to_ids(template.sep)
        history = [] if history is None else history
        if verbose and pixel_values is not None:
            print(f'dynamic ViT batch size: {pixel_values.shape[0]}')
        # the reference discards the template's prompt and builds this string (:857-866): no newline after the roles,
        # no <img> markers, `num_image_token` context tokens per <image> whatever the tile count
        query = ('<|im_start|>system你是由上海人工智能实验室联合商汤科技开发的书生多模态大模型，英文名叫InternVL, 是一个有用无害的人工智能助手。<|im_end|>\n'
                 f'<|im_start|>user{questions}') + '<image>'
        for _ in num_patches_list:
            query = query.replace('<image>', IMG_CONTEXT_TOKEN * self.num_image_token, 1)
        query += '<|im_end|>\n<|im_start|>assistant'
        model_inputs = tokenizer(query, return_tensors='pt')
        out = gen(pixel_values=pixel_values, input_ids=model_inputs['input_ids'], attention_mask=model_inputs['attention_mask'],
                  **generation_config)
        response = tokenizer.batch_decode(out, skip_special_tokens=True)[0].split(template.sep)[0].strip()
        history.append((questions, response))
        if return_history:
            return response, history
        if verbose:
            print(query.replace(IMG_CONTEXT_TOKEN, '').replace(f'{IMG_START_TOKEN}{IMG_END_TOKEN}', '<image>'), response)
        return response

    def chat_ocr_pages(self, tokenizer, detect_model, images, question, generation_config, boxes_list=None, use_p=True,
                       drop_zero=False, hard_vq=False, repetition_penalty=1.5, IMG_START_TOKEN='<img>', IMG_END_TOKEN='</img>',
                       IMG_CONTEXT_TOKEN='<IMG_CONTEXT>', ALIGNED_TOKEN='[UNUSED_TOKEN_140]', errors='raise'):
        """Many pages at once (new: the reference's chat_ocr is one page per call, evaluate.py loops over it).  The
        character tiles of ALL pages go through the visual stage as one batch, the prompts are prefilled together and
        the pages decode as one batch; every page gets exactly the response its own chat_ocr call would produce.
        errors='return': a page whose own chat_ocr call would raise before generation (unreadable image, detector failure, no box, prompt too long) gets that
        exception object in its slot and the other pages run -- what folder mode's per-image try/except needs (inference.py:55-57)."""
        embeds, meta = self._ocr_embeds(tokenizer, detect_model, images, question, generation_config, boxes_list, use_p, drop_zero,
                                        hard_vq, IMG_START_TOKEN, IMG_END_TOKEN, IMG_CONTEXT_TOKEN, ALIGNED_TOKEN, errors=errors)
        outs = self.generate_pages(embeds, meta['max_new'], meta['eos'], repetition_penalty) if embeds else []
        return self._responses(tokenizer, outs, meta)

    @staticmethod
    def _responses(tokenizer, outs, meta):
        """ids of the pages that ran + the exceptions of those that did not -> one entry per page of the batch, in order (:752-753 per page)."""
        sep = meta['template'].sep
        res = [None] * meta['n']
        for i, o in zip(meta['ok'], outs):
            res[i] = tokenizer.batch_decode(torch.tensor([o]), skip_special_tokens=True)[0].split(sep)[0].strip()
        for i, e in meta['failed'].items():
            res[i] = e
        return res

    def chat_ocr_stream(self, tokenizer, detect_model, image_batches, question, generation_config, boxes_batches=None, use_p=True,
                        drop_zero=False, hard_vq=False, repetition_penalty=1.5, IMG_START_TOKEN='<img>', IMG_END_TOKEN='</img>',
                        IMG_CONTEXT_TOKEN='<IMG_CONTEXT>', ALIGNED_TOKEN='[UNUSED_TOKEN_140]', errors='raise', stats=None):
        """chat_ocr_pages over a sequence of page batches with two batches in flight (PagePipeline): a generator that yields one
        list of responses per batch, in order; batch i decodes on a second stream while batch i+1 is detected, tiled, encoded and
        prefilled, and the image files of batch i+2 are being decoded on the feeder's threads (pageio.PageFeeder).  Every page gets the
        response of its own chat_ocr call.  stats (a dict) receives the host seconds per stage and, per batch, how long the compute stream
        sat between the previous batch's prefill and this batch's first kernel."""
        pipe, pending = None, None
        feeder = self._feeder()
        marks, host0 = [], dict(feeder.stats, tokenize_s=self._tok_s)
        it = iter(image_batches)
        boxes_it = iter(boxes_batches) if boxes_batches is not None else None

        def pull():
            images = next(it, None)
            if images is None:
                return None
            images = list(images)
            return [images, (next(boxes_it) if boxes_it is not None else None), feeder.decode(images), None]

        def tile(item):
            """uploads + tile kernels of a pulled batch on the feeder's stream; a failure is kept for the batch's own turn"""
            if item is not None and item[3] is None:
                try:
                    item[3] = feeder.tiles(item[2], item[1], detect_model, use_p, errors)
                except Exception as e:
                    item[3] = e
        import collections
        ahead = collections.deque()              # pulled batches, oldest first: [0] next to run (tiled while its predecessor ran), [1] decoding

        def fill():
            while len(ahead) < 2:
                item = pull()
                if item is None:
                    return
                ahead.append(item)
        try:
            fill()
            while ahead:
                cur = ahead.popleft()
                tile(cur)
                try:
                    if isinstance(cur[3], Exception):
                        raise cur[3]
                    embeds, meta = self._ocr_embeds(tokenizer, None, cur[0], question, generation_config, None, use_p, drop_zero, hard_vq,
                                                    IMG_START_TOKEN, IMG_END_TOKEN, IMG_CONTEXT_TOKEN, ALIGNED_TOKEN, tiles=cur[3], errors=errors,
                                                    mark=marks if stats is not None else None)
                except Exception:
                    # this batch fails as a whole (errors='raise', or the visual stage itself): the batch before it is still decoding and its responses are good --
                    # they are handed out first, then the failure surfaces (folder mode re-runs exactly the failing batch page by page)
                    if pending is not None:
                        done, pending = pending, None
                        yield self._responses(tokenizer, pipe.finish(), done)
                    raise
                # batch i+1's tiles BEFORE this thread blocks for batch i-1's decode: when it comes back the compute stream is close to the end of batch i's
                # prefill, and the first kernel of batch i+1 must not wait for 64 uploads to be issued (measured: 40-60 ms of idle compute stream per batch)
                if ahead:
                    tile(ahead[0])
                fill()                                          # batch i+2: its files start decoding
                if not embeds:                                  # nothing of this batch reached the language model
                    if pending is not None:
                        yield self._responses(tokenizer, pipe.finish(), pending)
                        pending = None
                    yield self._responses(tokenizer, [], meta)
                    continue
                if pipe is None:
                    pipe = self.page_pipeline(max_new_tokens=meta['max_new'], eos_token_id=meta['eos'], repetition_penalty=repetition_penalty)
                prev = pipe.start(embeds)
                if stats is not None:
                    e = torch.cuda.Event(enable_timing=True)
                    e.record(torch.cuda.current_stream())
                    marks.append(('end', e))
                if pending is not None:
                    yield self._responses(tokenizer, prev, pending)
                pending = meta
            if pending is not None:
                done, pending = pending, None
                yield self._responses(tokenizer, pipe.finish(), done)
        finally:
            for item in ahead:
                if item is not None:
                    for f in item[2]:
                        f.cancel()
            if pipe is not None:
                pipe.close()
            if stats is not None:
                torch.cuda.current_stream().synchronize()
                idle, last_end = [], None
                for kind, e in marks:
                    if kind == 'begin' and last_end is not None:
                        idle.append(round(last_end.elapsed_time(e), 2))
                    elif kind == 'end':
                        last_end = e
                stats['compute_stream_idle_ms_between_batches'] = idle
                stats['host'] = {k: v - host0[k] for k, v in dict(feeder.stats, tokenize_s=self._tok_s).items()}       # this call's share of the feeder's counters

    def _feeder(self):
        if getattr(self, '_pagefeeder', None) is None:
            from .pageio import PageFeeder
            self._pagefeeder = PageFeeder(self)
            if getattr(self, '_tok_cache', None) is None:
                self._tok_cache, self._tok_s = {}, 0.0
        return self._pagefeeder

    def _prompt_ids(self, tokenizer, q, n_tiles, n_ref, IMG_START_TOKEN, IMG_END_TOKEN, IMG_CONTEXT_TOKEN, ALIGNED_TOKEN):
        """ids of one page's prompt = tokenizer(query of chat_ocr, :690-726).  A page prompt is ~40 KB of text of which all but ~100 characters are two RUNS of one
        added token each (256 x tiles <IMG_CONTEXT>, n_ref [UNUSED_TOKEN_140] appended to the question): tokenizers split added tokens out first and
        tokenise the stretches between them on their own, so the ids are those of the SKELETON (both runs at length one) with the two ids repeated.
        The skeleton is tokenised once per question; the FIRST page of every skeleton is also tokenised in full and compared -- a tokenizer for which the
        shortcut does not hold keeps the full path."""
        t0 = time.perf_counter()
        if getattr(self, '_tok_cache', None) is None:
            self._tok_cache, self._tok_s = {}, 0.0
        ctx = IMG_CONTEXT_TOKEN * (self.num_image_token * n_tiles)
        appended = n_ref is not None and ALIGNED_TOKEN not in q
        def full():
            query, _, _ = self._build_query(q + ALIGNED_TOKEN * n_ref if appended else q, None, [n_tiles], IMG_START_TOKEN, IMG_END_TOKEN, IMG_CONTEXT_TOKEN)
            return tokenizer(query, return_tensors='pt')['input_ids'].reshape(-1)
        ids = None
        if n_tiles > 0 and (not appended or n_ref > 0) and IMG_CONTEXT_TOKEN not in q:
            skel, _, _ = self._build_query(q + ALIGNED_TOKEN if appended else q, None, [n_tiles], IMG_START_TOKEN, IMG_END_TOKEN, IMG_CONTEXT_TOKEN)
            skel = skel.replace(ctx, IMG_CONTEXT_TOKEN, 1)
            key = (id(tokenizer), skel, appended)
            ent = self._tok_cache.get(key)
            if ent is None:
                sid = tokenizer(skel, return_tensors='pt')['input_ids'].reshape(-1)
                pi = (sid == self.img_context_token_id).nonzero().reshape(-1)
                pr = (sid == self.aligned_token_id).nonzero().reshape(-1)
                ent = False
                if pi.numel() == 1 and (not appended or (pr.numel() == 1 and int(pr[0]) > int(pi[0]))):
                    ent = (sid, int(pi[0]), int(pr[0]) if appended else None)
                if ent:
                    ref = full()
                    if not torch.equal(self._expand(ent, n_tiles, n_ref), ref):
                        ent = False
                    ids = ref
                if len(self._tok_cache) > 256:
                    self._tok_cache.clear()
                self._tok_cache[key] = ent
            if ent and ids is None:
                ids = self._expand(ent, n_tiles, n_ref)
        if ids is None:
            ids = full()
        self._tok_s += time.perf_counter() - t0
        return ids

    def _expand(self, ent, n_tiles, n_ref):
        sid, pi, pr = ent
        img = torch.full((self.num_image_token * n_tiles,), self.img_context_token_id, dtype=sid.dtype)
        if pr is None:
            return torch.cat([sid[:pi], img, sid[pi + 1:]])
        return torch.cat([sid[:pi], img, sid[pi + 1:pr], torch.full((n_ref,), self.aligned_token_id, dtype=sid.dtype), sid[pr + 1:]])

    def _ocr_embeds(self, tokenizer, detect_model, images, question, generation_config, boxes_list, use_p, drop_zero, hard_vq,
                    IMG_START_TOKEN, IMG_END_TOKEN, IMG_CONTEXT_TOKEN, ALIGNED_TOKEN, tiles=None, errors='raise', mark=None):
        """Everything of chat_ocr_pages before the language model: the pages' prompt embeddings with both splices done, and
        meta = {n, ok: batch indices of the pages in `embeds`, failed: {index: exception}, max_new, eos, template}.  The host side (decode, boxes, tiles) is
        pageio.PageFeeder's; nothing on this thread's stream waits for a pageable copy."""
        self.img_context_token_id = tokenizer.convert_tokens_to_ids(IMG_CONTEXT_TOKEN)
        feeder = self._feeder()
        batch = tiles if tiles is not None else feeder.tiles(feeder.decode(images), boxes_list, detect_model, use_p, errors)
        template = get_conv_template(self.template)
        generation_config = dict(generation_config)
        generation_config['eos_token_id'] = tokenizer.convert_tokens_to_ids(template.sep)
        max_new, eos = self._gen_args(generation_config)
        meta = {'n': len(images), 'ok': [], 'failed': dict(batch.failed), 'max_new': max_new, 'eos': eos, 'template': template}
        if not batch.ok:
            return [], meta
        feeder.hand_over(batch)
        if mark is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record(torch.cuda.current_stream())
            mark.append(('begin', e))
        feats = self.extract_feature(batch.page_px)
        if use_p:
            feat_c = self.extract_feature(batch.char_px)
            rs = self.resampler(feat_c)
            outs = self.engine.vq(rs, with_cos=hard_vq)
            idx, cos = outs if hard_vq else (outs, None)
        q = question if '<image>' in question else '<image>\n' + question
        refs, all_ids, coff = [], [], 0
        for k in range(len(batch.ok)):
            ref = None
            if use_p:
                sl = slice(coff, coff + batch.n_chars[k])
                ref = self.engine.denorm(rs[sl], idx[sl], cos[sl] if cos is not None else None, drop_zero=drop_zero, hard_vq=hard_vq)
                coff += batch.n_chars[k]
            refs.append(ref)
            all_ids.append(self._prompt_ids(tokenizer, q, batch.n_tiles[k], ref.shape[0] if use_p else None, IMG_START_TOKEN, IMG_END_TOKEN,
                                            IMG_CONTEXT_TOKEN, ALIGNED_TOKEN))
        ids_dev = torch.cat(all_ids).pin_memory().to(self.device, non_blocking=True)          # ONE asynchronous upload for the batch's prompts
        embeds, toff, ioff = [], 0, 0
        for k, i in enumerate(batch.ok):
            S = all_ids[k].numel()
            if S + max_new > self.max_tokens:
                err = ValueError(f'prompt of {S} tokens + {max_new} new exceeds max_tokens={self.max_tokens}')       # _greedy's check, per page
                if errors == 'raise':
                    raise err
                meta['failed'][i] = err
            else:
                embeds.append(self.engine.embed_splice(ids_dev[ioff:ioff + S], feats[toff:toff + batch.n_tiles[k]], refs[k], img_id=self.img_context_token_id,
                                                       ref_id=self.aligned_token_id))
                meta['ok'].append(i)
            toff += batch.n_tiles[k]
            ioff += S
        return embeds, meta

    def batch_chat(self, tokenizer, pixel_values, questions, generation_config, num_patches_list=None, history=None,
                   return_history=False, IMG_START_TOKEN='<img>', IMG_END_TOKEN='</img>', IMG_CONTEXT_TOKEN='<IMG_CONTEXT>',
                   verbose=False, image_counts=None):
        """:903-951.  The reference left-pads and runs one padded batch; here every query is prefilled at its own
        length and the pages then decode together (cr_llm_decode), which gives each row exactly its unpadded result."""
        if history is not None or return_history:
            print('Now multi-turn chat is not supported in batch_chat.')
            raise NotImplementedError
        if image_counts is not None:
            num_patches_list = image_counts
        self.img_context_token_id = tokenizer.convert_tokens_to_ids(IMG_CONTEXT_TOKEN)
        generation_config = dict(generation_config)
        embeds, off = [], 0
        feats = self.extract_feature(pixel_values) if pixel_values is not None else None
        for idx, num_patches in enumerate(num_patches_list):
            question = questions[idx]
            if pixel_values is not None and '<image>' not in question:
                question = '<image>\n' + question
            query, template, _ = self._build_query(question, None, [num_patches], IMG_START_TOKEN, IMG_END_TOKEN, IMG_CONTEXT_TOKEN)
            ids = tokenizer(query, return_tensors='pt')['input_ids'].reshape(-1)
            embeds.append(self.engine.embed_splice(ids, feats[off:off + num_patches] if feats is not None else None,
                                                   img_id=self.img_context_token_id))
            off += num_patches
        generation_config['eos_token_id'] = tokenizer.convert_tokens_to_ids(template.sep)
        max_new, eos = self._gen_args(generation_config)
        outs = self.generate_pages(embeds, max_new, eos, generation_config.get('repetition_penalty', 1.0))
        responses = [tokenizer.batch_decode(torch.tensor([o]), skip_special_tokens=True)[0] for o in outs]
        return [r.split(template.sep)[0].strip() for r in responses]

    # ---- page-parallel generation (new: the reference decodes one page at a time) ---------------
    def page_pipeline(self, **kw):
        """A PagePipeline over this model: batches of pages with the decode of one batch beside the visual stage of the next."""
        return PagePipeline(self, **kw)

    def generate_pages(self, embeds_list, max_new_tokens=1024, eos_token_id=EOS_TOKEN_ID, repetition_penalty=1.0,
                       check_every=16, prefill_batch=16):
        """Prefill every page, then decode all unfinished pages as ONE batch per step: the 14.7 GB of LLM weights are
        streamed once per step for all pages.  Per page the ids equal the single-page greedy result."""
        P = len(embeds_list)
        if self._kv is None or self._kv.n_seqs < P:
            if self._kv is not None:
                self._kv.free()
            self.max_pages = max(P, self.max_pages)
            self._kv = self.engine.kv_alloc(self.max_pages, self.max_tokens)
        kv = self._kv
        kv.reset()
        for i0 in range(0, P, prefill_batch):                       # prompts of several pages share the linear layers' GEMMs
            idx = list(range(i0, min(P, i0 + prefill_batch)))
            self.engine.prefill_batch(kv, idx, [embeds_list[i] for i in idx], penalty=repetition_penalty)
        live = list(range(P))
        n = 1
        done = {}
        while n < max_new_tokens and live:
            steps = min(check_every, max_new_tokens - n)
            for _ in range(steps):
                self.engine.decode(kv, live, penalty=repetition_penalty)
            n += steps
            if eos_token_id is not None:
                for i in list(live):
                    ids = kv.generated(i)
                    if eos_token_id in ids:
                        done[i] = ids[:ids.index(eos_token_id) + 1]
                        live.remove(i)
        outs = []
        for i in range(P):
            ids = done.get(i)
            if ids is None:
                ids = kv.generated(i)[:max_new_tokens]
                if eos_token_id is not None and eos_token_id in ids:
                    ids = ids[:ids.index(eos_token_id) + 1]
            outs.append(ids)
        return outs


class PagePipeline:
    """Two batches of pages in flight.  The batched decode is HBM-bound (weights + KV cache streamed once per step) and leaves the
    matrix cores idle; the visual stage and the prefill of the NEXT batch are matrix-bound.  A worker thread decodes batch i-1 on
    its own HIP stream, through a second context that shares this model's weights (Engine.share_weights_from: own workspace, no
    copy), while the caller runs batch i's visual stage and prefill as usual.  Measured on MI355X: a 63-tile ViT chunk takes 54 ms
    instead of 50 beside a decode step that takes 26 ms instead of 11, i.e. 1.33x the work per unit time while both run.  The
    two streams have to be fed by two host threads: one thread alternating between them got a quarter of that (scripts/overlap_probe.py).
    Per page the ids are those of generate_pages (same kernels, same order per page).

        pipe = model.page_pipeline(max_new_tokens=..., eos_token_id=...)
        for batch in batches:
            embeds = ... visual stage of `batch`, as for generate_pages ...
            outs_prev = pipe.start(embeds)                 # prefill; returns the PREVIOUS batch's ids (None the first time)
        outs_last = pipe.finish()
        pipe.close()

    The two KV caches alternate between batches."""

    def __init__(self, model, max_new_tokens=1024, eos_token_id=EOS_TOKEN_ID, repetition_penalty=1.0, check_every=16, prefill_batch=16):
        import queue
        import threading
        from .engine import Engine
        self.m, self.eng = model, model.engine
        self.max_new_tokens, self.eos, self.penalty = max_new_tokens, eos_token_id, repetition_penalty
        self.check_every, self.prefill_batch = int(os.environ.get('CR_PIPE_CHECK_EVERY', check_every)), prefill_batch      # (env: development aid, how often the decode thread looks for EOS)
        self.dec = Engine(self.eng.dims, device=self.eng.device.index, max_pos=self.eng.max_pos)
        self.dec.share_weights_from(self.eng)
        # a stream of another priority level also lands on another hardware queue than the caller's (two streams of one level may share one)
        self.side = torch.cuda.Stream(device=self.eng.device, priority=int(os.environ.get('CR_PIPE_PRIO', '-1')))
        self.kvs = [None, None]
        self.turn = 0
        self.jobs, self.results = queue.Queue(), queue.Queue()
        self.in_flight = 0
        self.host_s, self.host_steps = 0.0, 0         # host time the decode thread spent issuing steps (diagnostic)
        self.worker = threading.Thread(target=self._decode_loop, daemon=True)
        self.worker.start()

    def _kv(self, slot, P):
        kv = self.kvs[slot]
        if kv is None or kv.n_seqs < P:
            if kv is not None:
                kv.free()
            kv = self.kvs[slot] = self.eng.kv_alloc(max(P, self.m.max_pages), self.m.max_tokens)
        return kv

    def _decode_loop(self):
        torch.cuda.set_device(self.eng.device)
        while True:
            job = self.jobs.get()
            if job is None:
                return
            kv, P, ready = job
            try:
                with torch.cuda.stream(self.side):
                    self.side.wait_event(ready)                        # the prefill that filled this cache, on the caller's stream
                    live, done, n = list(range(P)), {}, 1
                    span = [torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)] if os.environ.get('CR_PIPE_MARKS') else None
                    if span:
                        span[0].record(self.side)
                        t_first = time.perf_counter()
                    while n < self.max_new_tokens and live:
                        steps = min(self.check_every, self.max_new_tokens - n)
                        t0 = time.perf_counter()
                        for _ in range(steps):
                            self.dec.decode(kv, live, penalty=self.penalty)
                        self.host_s += time.perf_counter() - t0
                        self.host_steps += steps
                        n += steps
                        if self.eos is not None:
                            for i in list(live):
                                ids = kv.generated(i)                  # synchronises the side stream only
                                if self.eos in ids:
                                    done[i] = ids[:ids.index(self.eos) + 1]
                                    live.remove(i)
                    if span:
                        t_issued = time.perf_counter()
                        span[1].record(self.side)
                        span[1].synchronize()
                        print(f'[marks] decode of a batch: {span[0].elapsed_time(span[1]):.0f} ms on the GPU, all steps issued after {1e3 * (t_issued - t_first):.0f} ms of host time',
                              file=sys.stderr, flush=True)
                    outs = []
                    for i in range(P):
                        ids = done.get(i)
                        if ids is None:
                            ids = kv.generated(i)[:self.max_new_tokens]
                            if self.eos is not None and self.eos in ids:
                                ids = ids[:ids.index(self.eos) + 1]
                        outs.append(ids)
                self.results.put(outs)
            except BaseException as e:                                 # hand the failure to the caller's thread
                self.results.put(e)

    def _collect(self):
        if not self.in_flight:
            return None
        self.in_flight -= 1
        out = self.results.get()
        if isinstance(out, BaseException):
            raise out
        return out

    def start(self, embeds_list):
        """Prefill a batch on the caller's stream and hand it to the decode thread; returns the previous batch's ids (None if there
        was none), waiting for them if they are not there yet."""
        P = len(embeds_list)
        # The OTHER slot may be under the worker's decode right now (the previous batch, not collected yet): it is only ever
        # CREATED here, on the first batch (28 GB at 64 pages: not inside a later step), never regrown -- a batch larger than it
        # holds regrows it when that slot's turn comes, i.e. after _collect() has returned the batch that last used it.
        if self.kvs[self.turn ^ 1] is None:
            self._kv(self.turn ^ 1, P)
        kv = self._kv(self.turn, P)                # last used two batches ago, and that batch has been collected
        self.turn ^= 1
        kv.reset()
        for i0 in range(0, P, self.prefill_batch):
            idx = list(range(i0, min(P, i0 + self.prefill_batch)))
            self.eng.prefill_batch(kv, idx, [embeds_list[i] for i in idx], penalty=self.penalty)
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream())
        prev = self._collect()
        self.jobs.put((kv, P, ready))
        self.in_flight += 1
        return prev

    def finish(self):
        return self._collect()

    def close(self):
        if self.worker is not None:
            while self.in_flight:
                self._collect()
            self.jobs.put(None)
            self.worker.join()
            self.worker = None
        for kv in self.kvs:
            if kv is not None:
                kv.free()
        self.kvs = [None, None]
        if self.dec is not None:
            self.dec.close()
            self.dec = None


def load_boxes_json(path):
    from .preprocess import boxes_from_labelme
    with open(path, 'r', encoding='utf-8') as f:
        return boxes_from_labelme(json.load(f))
