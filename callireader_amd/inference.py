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


def is_image(p):
    return os.path.isfile(p) and p.lower().endswith(IMG_EXT)


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
               boxes=None):
    bx = None
    if use_p and (boxes is not None or detect_model is None):
        bx = boxes_for(image_path, boxes)
    response, history = model.chat_ocr(tokenizer, detect_model, image_path, prompt, generation_config, use_p=use_p, hard_vq=hard_vq,
                                       drop_zero=drop_zero, repetition_penalty=repetition_penalty, return_history=True,
                                       verbose=verbose, boxes=bx)
    print(f'User: {prompt}\nAssistant: {response}')
    return response


def folder_rec(model, tokenizer, detect_model, generation_config, folder_path, prompt, save_name, use_p, hard_vq, drop_zero, repetition_penalty,
               verbose):
    results = []
    for pic in sorted(f for f in os.listdir(folder_path) if f.lower().endswith(IMG_EXT)):
        pic_path = os.path.join(folder_path, pic)
        try:
            response = single_rec(model, tokenizer, detect_model, generation_config, pic_path, prompt, use_p, hard_vq, drop_zero,
                                  repetition_penalty, verbose)
        except Exception as e:                               # inference.py:55-57
            print(f'An error has occured:\n{e}')
            response = 'ERROR!'
        results.append({'imagePath': pic_path, 'prompt': prompt, 'response': response})
    if not save_name.endswith('json'):
        save_name += '_result.json'
    with open(save_name, 'w', encoding='utf-8') as f:
        json.dump(results, f, ensure_ascii=False, indent=2)


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
    parser.add_argument('--fp8_decode', action='store_true', help='e4m3 weights for the decode (cr_enable_fp8_decode; off by default: the reference computes in bf16)')
    parser.add_argument('--fp8_mfma', type=int, nargs='?', const=1, default=0, choices=(0, 1, 2),
                        help='e4m3 x e4m3 matrix-core linears (cr_enable_fp8_mfma; off by default; a throughput option, not parity-preserving): 1 = the norm-fed linears of the ViT / projector / prefill, 2 = also ViT fc2 and the prefill\'s wo / w2')
    args = parser.parse_args(argv)
    if not isinstance(args.tgt, str):
        raise ValueError(f'The target should a string, not a instance of {type(args.tgt)}!')
    from .tokenization_internlm2 import InternLM2Tokenizer
    model = InternVLChatModel.from_pretrained(args.model, params_dir=args.params, torch_dtype=torch.bfloat16).eval().cuda()
    if args.fp8_mfma:
        model.engine.enable_fp8_mfma(True, level=args.fp8_mfma)
    if args.fp8_decode:
        model.engine.enable_fp8_decode(True)
    # the engine's own reader of tokenizer.model (+ tokenizer_config.json / added_tokens.json): the reference's
    # AutoTokenizer path needs sentencepiece==0.2.0; any HF-style tokenizer object works with chat_ocr as well
    tokenizer = InternLM2Tokenizer.from_pretrained(args.model)
    generation_config = dict(num_beams=1, max_new_tokens=1024, do_sample=False)
    detect_model = load_detector(args.params)
    if is_image(args.tgt):
        print('Single image recognition mode.')
        single_rec(model, tokenizer, detect_model, generation_config, args.tgt, args.prompt, args.use_p, args.hard_vq, args.drop_zero,
                   args.repetition_penalty, args.verbose, args.boxes)
    elif os.path.isdir(args.tgt):
        print('Multiple images recognition mode')
        os.makedirs('results', exist_ok=True)
        folder_rec(model, tokenizer, detect_model, generation_config, args.tgt, args.prompt, os.path.join('results', args.save_name),
                   args.use_p, args.hard_vq, args.drop_zero, args.repetition_penalty, args.verbose)
    else:
        raise ValueError('The target should be either a image path or a folder that contain images!')


if __name__ == '__main__':
    main()
