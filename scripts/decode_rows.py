#!/usr/bin/env python3
"""Decode step time against the number of rows: 1 / 8 / 16 / 64 pages of the bench's prompt length, plain launches
and hipGraph replay (CR_DECODE_GRAPH is read when the context is created, so the parent starts one child per setting).
   python scripts/decode_rows.py            -> gpurun_out/decode_rows.json
   python scripts/decode_rows.py child      (one setting, rows swept in-process; under rocprofv3 --kernel-trace --stats with ROWS=1 for the per-kernel view)"""
import json, os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

ROWS = [int(x) for x in os.environ.get('ROWS', '1,8,16,64').split(',')]
STEPS = int(os.environ.get('STEPS', '64'))

def child():
    import torch
    import bench
    from callireader_amd.config import ModelDims
    from callireader_amd.modeling_internvl_chat import InternVLChatModel
    dev = torch.device('cuda', 0)
    dims = ModelDims.full()
    S = bench.PAGE_TILES * 256 + bench.CHAR_TILES * 3 + bench.TEXT_TOKENS
    P = max(ROWS)
    model = InternVLChatModel.from_synthetic(dims, seed=0, device=0, max_tokens=S + 2048, max_pages=P)
    eng = model.engine
    g = torch.Generator(device='cuda').manual_seed(1)
    kv = eng.kv_alloc(P, S + 2048)
    kv.reset()
    for i0 in range(0, P, 16):
        idx = list(range(i0, min(P, i0 + 16)))
        eng.prefill_batch(kv, idx, [(torch.randn(S, dims.llm_hidden, device=dev, generator=g) * 0.02).bfloat16() for _ in idx], penalty=1.0)
    res = {}
    for n in ROWS:
        live = list(range(n))
        for _ in range(6): eng.decode(kv, live, penalty=1.0)
        torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(STEPS): eng.decode(kv, live, penalty=1.0)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / STEPS * 1e3)
        res[n] = min(ts)
        ctx = kv.length(0) if hasattr(kv, 'length') else -1
        print(f'rows {n:3d}: {min(ts):.3f} ms per step (runs {", ".join(f"{t:.3f}" for t in ts)}), graph {os.environ.get("CR_DECODE_GRAPH", "0")} fused {os.environ.get("CR_DECODE_FUSED", "1")}', flush=True)
    print('RESULT ' + json.dumps(res), flush=True)

if len(sys.argv) > 1 and sys.argv[1] == 'child':
    child()
else:
    out = {}
    for name, graph, fused in (('separate kernels (CR_DECODE_FUSED=0), plain launches', '0', '0'), ('fused small-batch path, plain launches', '0', '1'),
                               ('fused small-batch path, hipGraph replay', '1', '1')):
        env = dict(os.environ, CR_DECODE_GRAPH=graph, CR_DECODE_FUSED=fused)
        p = subprocess.run([sys.executable, os.path.abspath(__file__), 'child'], env=env, capture_output=True, text=True)
        sys.stdout.write(p.stdout); sys.stderr.write(p.stderr[-2000:])
        for line in p.stdout.splitlines():
            if line.startswith('RESULT '): out[name] = json.loads(line[7:])
    os.makedirs('gpurun_out', exist_ok=True)
    json.dump(out, open('gpurun_out/decode_rows.json', 'w'), indent=1)
    print(json.dumps(out))
