#!/usr/bin/env python3
"""Where do gemm256's spilled VGPRs live (round-4 verdict, item 6b)?  Compiles gemm256.hip to assembly and reports, per kernel instance that spills, every
scratch_store / scratch_load with its position relative to the K loop (the innermost loops that contain the main loop's MFMAs).
    python scripts/spill_sites.py [instance regex]      (no GPU needed)"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, 'callireader_amd', 'csrc', 'gemm256.hip')
asm = os.path.join(tempfile.gettempdir(), 'gemm256_spills.s')
subprocess.check_call(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-S', '--cuda-device-only', src, '-o', asm])
txt = open(asm).read()
pat = re.compile(sys.argv[1]) if len(sys.argv) > 1 else None
for m in re.finditer(r'\.globl\s+(_ZN12_GLOBAL__N_114gemm256_kernelILi(\d+)ELb(\d)ELb(\d)E\S*)[^\n]*\n(.*?)\.Lfunc_end\d+:', txt, flags=re.S):
    name = f'gemm256_kernel<{m.group(2)}, {"true" if m.group(3) == "1" else "false"}, {"true" if m.group(4) == "1" else "false"}>'
    if pat and not pat.search(name):
        continue
    lines = m.group(5).split('\n')
    scr = [(i, l.strip()) for i, l in enumerate(lines) if re.match(r'\s*scratch_(load|store)', l)]
    if not scr:
        continue
    labels = {mm.group(1): i for i, l in enumerate(lines) for mm in [re.match(r'(\.LBB\d+_\d+):', l)] if mm}
    loops = []
    for i, l in enumerate(lines):
        mm = re.search(r's_c?branch\w* (\.LBB\d+_\d+)', l)
        if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
            loops.append((labels[mm.group(1)], i))
    mf = [i for i, l in enumerate(lines) if 'v_mfma' in l]
    # the K loop(s): innermost loops holding >= 64 MFMAs
    kloops = [lp for lp in loops if sum(1 for x in mf if lp[0] <= x <= lp[1]) >= 64]
    kloops = [lp for lp in kloops if not any(o != lp and lp[0] <= o[0] and o[1] <= lp[1] for o in kloops)]
    print(f'{name}: {len(scr)} scratch instructions; K loop(s) at lines {kloops} of {len(lines)} ({[sum(1 for x in mf if a <= x <= b) for a, b in kloops]} MFMAs)')
    for i, l in scr:
        where = 'INSIDE the K loop' if any(a <= i <= b for a, b in kloops) else ('inside the persistent tile loop, outside the K loop' if any(a <= i <= b for a, b in loops) else 'outside every loop (prologue / final epilogue)')
        print(f'    line {i:5d}  {l.split(";")[0].strip():48s} {where}')
