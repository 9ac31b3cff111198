#!/usr/bin/env python3
"""tests/golden/eval_vectors.json: the reference's own scoring functions on fixed strings (build container only).
`calculate_metrics` is utils/utils.py:516-542, `get_clean_string` evaluate.py:42-53 (evaluate.py is imported with stub
modules for opencc / ultralytics / Levenshtein, which it needs at import time only).  The normalised edit distance of
evaluate.py:145-147 calls the third-party `Levenshtein` wheel, absent here: its cases are produced by a textbook
dynamic programme in this script and marked as such."""
import json
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'scripts'))
from make_golden import install_stubs  # noqa: E402

install_stubs()
import types  # noqa: E402
sys.modules['opencc'].OpenCC = lambda *a, **k: types.SimpleNamespace(convert=lambda s: s)
tvt = sys.modules['torchvision.transforms']
tvt.Compose = lambda x: x
tvt.Lambda = tvt.Resize = tvt.ToTensor = tvt.Normalize = lambda *a, **k: None
os.chdir('/root/reference')
import evaluate as ref_eval  # noqa: E402
from utils.utils import calculate_metrics  # noqa: E402


def textbook_edit_distance(a, b):
    d = [[i + j if i * j == 0 else 0 for j in range(len(b) + 1)] for i in range(len(a) + 1)]
    for i in range(1, len(a) + 1):
        for j in range(1, len(b) + 1):
            d[i][j] = min(d[i - 1][j] + 1, d[i][j - 1] + 1, d[i - 1][j - 1] + (a[i - 1] != b[j - 1]))
    return d[-1][-1]


rng = random.Random(0)
alphabet = '君不见黄河之水天上来奔流到海不复回高堂明镜悲白发朝如青丝暮成雪abc'
pairs = [('', ''), ('君不见', ''), ('', '黄河'), ('君不见黄河之水', '君不见黄河之水'), ('水水水天', '水天天'), ('abcabc', 'cba')]
for _ in range(40):
    n, m = rng.randint(0, 30), rng.randint(1, 30)
    pairs.append((''.join(rng.choice(alphabet) for _ in range(n)), ''.join(rng.choice(alphabet) for _ in range(m))))
metrics = []
for p, g in pairs:
    pr, rc, f1 = calculate_metrics(list(p), list(g))
    metrics.append({'pred': p, 'gt': g, 'precision': pr, 'recall': rc, 'f1': f1, 'edit_distance_textbook': textbook_edit_distance(list(p), list(g))})
# evaluate_accuracy (evaluate.py:78-123) on one response at a time: letters named, option texts quoted, both, neither
answers = [('A', '王羲之', '颜真卿', '柳公权'), ('B', '楷书', '行书', '草书'), ('C', '竖排', '横排', '扇面')]
frags = ['A', 'B', 'C', 'A。', '答案：B', '选C', 'A和B', 'ABC', '', '王羲之', '颜真卿', '楷书', '行书', '竖排', 'A 王羲之', 'B 王羲之', '王羲之或颜真卿', 'C：竖排', 'B：行书', '不知道',
         'A: 颜真卿', '楷书，不是草书', 'C 扇面', '答案是 B：楷书']
choice = []
for ans in answers:
    for r in frags:
        choice.append({'response': r, 'answer': list(ans), 'accuracy': ref_eval.evaluate_accuracy([r], [ans])})
for _ in range(30):
    ans = rng.choice(answers)
    r = ''.join(rng.choice(frags + ['，', ' ']) for _ in range(rng.randint(1, 3)))
    choice.append({'response': r, 'answer': list(ans), 'accuracy': ref_eval.evaluate_accuracy([r], [ans])})
clean = ['君不见，黄河之水天上来！', 'Hello, world. (123) [x]-{y}*\n“引号”《书名》：；…—', '无标点', '', '1234567890', "it's; a: \"test\"?"]
out = {'_how': 'scripts/make_golden_eval.py', 'metrics': metrics, 'choice': choice, 'clean': [{'in': c, 'out': ref_eval.get_clean_string(c)} for c in clean]}
json.dump(out, open(os.path.join(ROOT, 'tests', 'golden', 'eval_vectors.json'), 'w'), ensure_ascii=False, indent=1)
print('ok', len(metrics), 'metric cases')
