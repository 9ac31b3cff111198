"""`python inference.py --tgt ...` with the reference's flags (/root/reference/inference.py:65-130), on the HIP engine.

The detector is what the reference uses: `ultralytics.YOLO(<params>/best.pt)` (inference.py:98), handed to chat_ocr as
is; everything behind it (repeated passes, de-duplication, column merge, OrderFormer, reading order) runs in this
package (ordering.py).  ultralytics is third-party and not part of this image: without it, or with --boxes / a
labelme-style `<image>.json` next to the image (the format of examples/0.json), the ordered boxes are read from JSON.
The tokenizer is the engine's own reader of the reference's tokenizer files under --model
(callireader_amd/tokenization_internlm2.py, pinned against the sentencepiece library in tests/test_tokenizer.py).
"""
import argparse
import json
import os

import torch

from .modeling_internvl_chat import InternVLChatModel, load_boxes_json

IMG_EXT = ('.jpg', '.jpeg', '.png', '.bmp', '.webp')


def _sniff(p):
    """imghdr.what of the reference's utils/utils.py:493-512 (a file is an image when its header says so); the module left the standard library in 3.13: by extension then."""
    try:
        import imghdr
    except ImportError:
        return 'by-extension' if p.lower().endswith(IMG_EXT) else None
    try:
        return imghdr.what(p)
    except Exception:
        return None


def is_image(p):
    return os.path.isfile(p) and _sniff(p) is not None


def get_image_paths(folder_path):
    """utils/utils.py:493-503: every file under the folder (recursively) whose header is an image's; sorted here, so that the batches and the JSON are reproducible."""
    return sorted(os.path.join(root, f) for root, _, files in os.walk(folder_path) for f in files if _sniff(os.path.join(root, f)) is not None)


def load_detector(params_dir='./params'):
    """inference.py:98 `YOLO(YOLO_CHECKPOINT)` (config/configu.py: ./params/best.pt) when ultralytics and the weights
    are present, else None (boxes then come from JSON)."""
    path = os.path.join(params_dir, 'best.pt')
    try:
        from ultralytics import YOLO
    except Exception:
        return None
    return YOLO(path) if os.path.exists(path) else None


def boxes_for(image_path, boxes_arg=None, required=True):
    cand = boxes_arg or os.path.splitext(image_path)[0] + '.json'
    if not os.path.exists(cand):
        if not required:
            return None
        raise FileNotFoundError(f'character boxes for {image_path}: no detector (ultralytics + params/best.pt) and no {cand} '
                                '(labelme-style, see examples/0.json)')
    return load_boxes_json(cand)


def single_rec(model, tokenizer, detect_model, generation_config, image_path, prompt, use_p, hard_vq, drop_zero, repetition_penalty, verbose,
               boxes=None, quiet=False):
    bx = None
    if use_p and (boxes is not None or detect_model is None):
        bx = boxes_for(image_path, boxes)
    response, history = model.chat_ocr(tokenizer, detect_model, image_path, prompt, generation_config, use_p=use_p, hard_vq=hard_vq,
                                       drop_zero=drop_zero, repetition_penalty=repetition_penalty, return_history=True,
                                       verbose=verbose, boxes=bx)
    if not quiet:
        print(f'User: {prompt}\nAssistant: {response}')
    return response


DEFAULT_BATCH_PAGES = 64       # pages per decode batch in folder mode: the most rows the batched decode kernels take, and the batch the headline is measured on


def folder_rec(model, tokenizer, detect_model, generation_config, folder_path, prompt, save_name, use_p, hard_vq, drop_zero, repetition_penalty,
               verbose, batch_pages=DEFAULT_BATCH_PAGES):
    """inference.py:47-62: every image of the folder -> results JSON [{imagePath, prompt, response}], an image that fails -> "ERROR!" for that entry only.
    The reference calls chat_ocr image after image; here the images go through `chat_ocr_stream` in batches of `batch_pages` (two batches in flight, files
    decoded ahead on threads): each response is the one the image's own chat_ocr call gives, pages/s is what changes (batch_pages <= 1: the serial loop).
    Failures stay per image: a page that cannot be read, has no boxes JSON / no box, or whose prompt is too long is reported alone; if a batch fails as a whole
    its pages are run one by one, each in its own try/except, as the reference does."""
    paths = get_image_paths(folder_path)
    responses = {}

    def one(pic_path):
        try:
            return single_rec(model, tokenizer, detect_model, generation_config, pic_path, prompt, use_p, hard_vq, drop_zero, repetition_penalty, verbose,
                              quiet=True)
        except Exception as e:                               # inference.py:55-57
            return e

    def record(pic_path, response):
        if isinstance(response, BaseException):
            print(f'An error has occured:\n{response}')
            response = 'ERROR!'
        print(f'User: {prompt}\nAssistant: {response}')
        responses[pic_path] = response

    if batch_pages <= 1 or not hasattr(model, 'chat_ocr_stream'):
        for pic_path in paths:
            record(pic_path, one(pic_path))
    else:
        # boxes from JSON where there is no detector (single_rec's rule); a page without them fails alone, before the batch is formed
        boxes, ready = {}, []
        for pic_path in paths:
            try:
                boxes[pic_path] = boxes_for(pic_path) if (use_p and detect_model is None) else None
                ready.append(pic_path)
            except Exception as e:
                responses[pic_path] = e
        batches = [ready[i:i + batch_pages] for i in range(0, len(ready), batch_pages)]
        done = 0
        while done < len(batches):
            stream = model.chat_ocr_stream(tokenizer, detect_model, batches[done:], prompt, generation_config,
                                           boxes_batches=[[boxes[p] for p in b] for b in batches[done:]] if (use_p and detect_model is None) else None,
                                           use_p=use_p, drop_zero=drop_zero, hard_vq=hard_vq, repetition_penalty=repetition_penalty, errors='return')
            try:
                for res in stream:
                    for pic_path, r in zip(batches[done], res):
                        responses[pic_path] = r
                    done += 1
            except Exception as e:
                print(f'[folder_rec] a batch failed as a whole ({e}); its {len(batches[done])} pages run one by one')
                for pic_path in batches[done]:
                    responses[pic_path] = one(pic_path)
                done += 1
        for pic_path in paths:                               # the reference's console output and order
            record(pic_path, responses[pic_path])
    results = [{'imagePath': p, 'prompt': prompt, 'response': responses[p]} for p in paths]
    if not save_name.endswith('json'):
        save_name += '_result.json'
    with open(save_name, 'w', encoding='utf-8') as f:
        json.dump(results, f, ensure_ascii=False, indent=4)           # utils/utils.py:59-62 save_json
    return results


def main(argv=None):
    parser = argparse.ArgumentParser(description='args for inference task')
    parser.add_argument('--tgt', type=str, help='Recognition target')
    parser.add_argument('--prompt', type=str, default='这幅书法作品内容是什么？', help='Prompt for recognition')
    parser.add_argument('--save_name', type=str, default='recognition.json')
    parser.add_argument('--use_p', type=bool, default=True)          # type=bool kept: any non-empty string is truthy, as upstream
    parser.add_argument('--hard_vq', type=bool, default=False)
    parser.add_argument('--drop_zero', type=bool, default=False)
    parser.add_argument('--verbose', type=bool, default=False)
    parser.add_argument('--repetition_penalty', type=float, default=1.0)
    parser.add_argument('--model', type=str, default='InternVL', help='checkpoint dir (INTERNVL_PATH)')
    parser.add_argument('--params', type=str, default='./params')
    parser.add_argument('--boxes', type=str, default=None, help='labelme-style JSON with ordered character boxes')
    parser.add_argument('--max_new_tokens', type=int, default=1024, help='generation_config.max_new_tokens (inference.py:92-96: 1024)')
    parser.add_argument('--batch_pages', type=int, default=DEFAULT_BATCH_PAGES,
                        help='folder mode: pages per batch through chat_ocr_stream (two batches in flight; same responses as the serial loop, which 1 selects)')
    parser.add_argument('--fp8_decode', action='store_true', help='e4m3 weights for the decode (cr_enable_fp8_decode; off by default: the reference computes in bf16)')
    parser.add_argument('--fp8_mfma', type=int, nargs='?', const=1, default=0, choices=(0, 1, 2),
                        help='e4m3 x e4m3 matrix-core linears (cr_enable_fp8_mfma; off by default; a throughput option, not parity-preserving): 1 = the norm-fed linears of the ViT / projector / prefill, 2 = also ViT fc2 and the prefill\'s wo / w2')
    args = parser.parse_args(argv)
    if not isinstance(args.tgt, str):
        raise ValueError(f'The target should a string, not a instance of {type(args.tgt)}!')
    from .tokenization_internlm2 import InternLM2Tokenizer
    folder = os.path.isdir(args.tgt) and args.batch_pages > 1
    model = InternVLChatModel.from_pretrained(args.model, params_dir=args.params, torch_dtype=torch.bfloat16,
                                              **(dict(max_pages=args.batch_pages) if folder else {})).eval().cuda()
    if args.fp8_mfma:
        model.engine.enable_fp8_mfma(True, level=args.fp8_mfma)
    if args.fp8_decode:
        model.engine.enable_fp8_decode(True)
    # the engine's own reader of tokenizer.model (+ tokenizer_config.json / added_tokens.json): the reference's
    # AutoTokenizer path needs sentencepiece==0.2.0; any HF-style tokenizer object works with chat_ocr as well
    tokenizer = InternLM2Tokenizer.from_pretrained(args.model)
    # the reference hard-codes the pseudo-token placeholder's id (92537, modeling_internvl_chat.py:1100); read from the tokenizer it is the same number on the
    # reference's files and the right one on any other vocabulary
    tid = tokenizer.convert_tokens_to_ids('[UNUSED_TOKEN_140]')
    if isinstance(tid, int) and 0 < tid < model.dims.vocab and tid != tokenizer.sp_model.unk_id:
        model.aligned_token_id = tid
    generation_config = dict(num_beams=1, max_new_tokens=args.max_new_tokens, do_sample=False)
    detect_model = load_detector(args.params)
    if is_image(args.tgt):
        print('Single image recognition mode.')
        single_rec(model, tokenizer, detect_model, generation_config, args.tgt, args.prompt, args.use_p, args.hard_vq, args.drop_zero,
                   args.repetition_penalty, args.verbose, args.boxes)
    elif os.path.isdir(args.tgt):
        print('Multiple images recognition mode')
        os.makedirs('results', exist_ok=True)
        folder_rec(model, tokenizer, detect_model, generation_config, args.tgt, args.prompt, os.path.join('results', args.save_name),
                   args.use_p, args.hard_vq, args.drop_zero, args.repetition_penalty, args.verbose, batch_pages=args.batch_pages)
    else:
        raise ValueError('The target should be either a image path or a folder that contain images!')


if __name__ == '__main__':
    main()
