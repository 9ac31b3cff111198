"""World-size-2 run of the sharded path on the GPU box: every page's ids equal the single-process result.
On a single-GPU box both ranks share GPU 0 and the all-gather goes over gloo (host buffers); on a multi-GPU node set
CR_DIST_BACKEND=nccl to exercise RCCL."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_sharded_pages_equal_single_process():
    env = dict(os.environ)
    env.setdefault('CR_DIST_BACKEND', 'nccl' if torch.cuda.device_count() >= 2 else 'gloo')
    env['HSA_ENABLE_IPC_MODE_LEGACY'] = '0'
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', '29533', os.path.join(ROOT, 'scripts', 'dist_check.py')]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = r.stdout.decode()
    assert r.returncode == 0, out[-3000:]
    assert 'DIST_CHECK OK' in out, out[-3000:]


def test_rccl_all_gather_runs_on_the_visible_gpus():
    """backend 'nccl' = RCCL.  With >= 2 GPUs: two ranks, one GPU each.  On a one-GPU box: world size 1 -- communicator set-up and
    the collective kernels still run (RCCL cannot put two ranks on one GPU, so the world-2 equality test above uses gloo there)."""
    n = min(torch.cuda.device_count(), 2)
    env = dict(os.environ)
    env['HSA_ENABLE_IPC_MODE_LEGACY'] = '0'
    script = os.path.join(ROOT, 'scripts', 'rccl_check.py')
    if n >= 2:
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
               '--master-port', '29537', script]
    else:
        cmd = [sys.executable, script]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = r.stdout.decode()
    assert r.returncode == 0 and 'RCCL_CHECK OK' in out, out[-3000:]


def test_sharded_generate_without_a_process_group_is_generate_pages():
    """parallel.sharded_generate with no process group (one rank): the same ids as the hand-written single-process flow, ragged pages included
    (a page without tiles of its own, different character counts)."""
    from callireader_amd.config import ModelDims
    from callireader_amd import synthetic
    from callireader_amd.modeling_internvl_chat import InternVLChatModel
    from callireader_amd.parallel import sharded_generate
    IMG, REF, NEW = 8990, 8991, 5
    dims = ModelDims.reduced(vit_layers=1, llm_layers=1, rs_depth=1, vocab=9000)
    m = InternVLChatModel.from_state_dict(synthetic.make_state_dict(dims, seed=0), dims, device=0, max_tokens=1024, max_pages=4)
    m.img_context_token_id, m.aligned_token_id = IMG, REF
    pts, cts = [2, 0, 1], [3, 5, 1]
    page_tiles = [synthetic.make_pixels(n, seed=30 + i) if n else torch.empty((0, 3, 448, 448), dtype=torch.bfloat16) for i, n in enumerate(pts)]
    char_tiles = [synthetic.make_pixels(n, seed=40 + i) for i, n in enumerate(cts)]
    ids = [torch.cat([torch.arange(50 + p, 58 + p), torch.full((pts[p] * 256,), IMG), torch.full((cts[p] * 3,), REF), torch.arange(5)]) for p in range(3)]
    got = sharded_generate(m, page_tiles, char_tiles, ids, img_id=IMG, ref_id=REF, max_new_tokens=NEW, eos_token_id=None)
    embeds = []
    for p in range(3):
        v = m.extract_feature(page_tiles[p].cuda()) if pts[p] else None
        ps, _ = m.align_tiles(char_tiles[p].cuda())
        embeds.append(m.engine.embed_splice(ids[p].cuda(), v, ps.reshape(-1, 3, dims.llm_hidden), img_id=IMG, ref_id=REF))
    want = m.generate_pages(embeds, max_new_tokens=NEW, eos_token_id=None)
    assert got == dict(enumerate(want))
