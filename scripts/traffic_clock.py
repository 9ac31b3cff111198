#!/usr/bin/env python3
"""Round-4 verdict, item 6: does the tiled GEMM's fabric traffic (2.3 x its algorithmic bytes) cost it CLOCK?  One table: for the ViT QKV shape
(M = 64 575, N = 3 072, K = 1 024) and 8192^3, the 256x256 kernel (i) as it is -- tile order = 8-row super-groups after the XCD remap, i.e. an XCD's 32
concurrent tiles are 8 row panels x 4 column panels = 12 panels, against the 2 sqrt(32) = 11.3 minimum; (ii) with the weight panel read as contiguous KiB
blocks (knock-out -DCR_KO_WCONTIG: what a pre-tiled copy would give; wrong results); (iii) with WORSE tile orders, -DCR_TILE_GM=1 (row-major: 33 panels per
32 tiles at 8192^3) and -DCR_TILE_GM=32 -- the lever pulled the other way, since (i) already sits at the minimum the verdict's (iii) asks for.  Per variant:
launch time by HIP events, the in-kernel clock (d s_memtime / d s_memrealtime, -DCR_DIAG_STAMPS), FETCH_SIZE by `rocprofv3 --pmc` (doubled, as the
micro-architecture guide prescribes for gfx950).  All variants are built by scripts/build_variant.py (csrc/diag.hpp) into ab/.
    python scripts/traffic_clock.py  -> profiles/round5/traffic_clock.json"""
import csv, glob, json, os, re, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VARIANTS = [('as it is (8-row super-groups: 12 panels per 32 tiles)', 'tc_base', []),
            ('weight panel as contiguous KiB blocks (knock-out, wrong results)', 'tc_wcontig', ['-DCR_KO_WCONTIG=1']),
            ('row-major tile order (-DCR_TILE_GM=1)', 'tc_gm1', ['-DCR_TILE_GM=1']),
            ('32-row super-groups (-DCR_TILE_GM=32)', 'tc_gm32', ['-DCR_TILE_GM=32'])]
SHAPES = [('ViT QKV', 64575, 3072, 1024, 0), ('8192^3', 8192, 8192, 8192, 0)]
prof = shutil.which('rocprofv3') or '/opt/rocm/bin/rocprofv3'
out = {'how': __doc__.split('\n')[0], 'rows': []}
env0 = dict(os.environ, TMPDIR='/tmp')
for label, name, flags in VARIANTS:
    subprocess.check_call([sys.executable, os.path.join(ROOT, 'scripts', 'build_variant.py'), name, 'gemm256.hip', '-DCR_DIAG_STAMPS=1'] + flags, stdout=subprocess.DEVNULL)
    lib = os.path.join(ROOT, 'ab', f'lib{name}.so')
    for sname, M, N, K, epi in SHAPES:
        env = dict(env0, CR_HIP_LIB=lib)
        r = subprocess.run([sys.executable, os.path.join(ROOT, 'scripts', 'gemm_stamps.py'), str(M), str(N), str(K), str(epi)], env=env, capture_output=True, text=True, cwd=ROOT)
        txt = r.stdout
        ms = float(re.search(r': ([0-9.]+) ms by events', txt).group(1)) if re.search(r': ([0-9.]+) ms by events', txt) else None
        clk = re.search(r'in-kernel clock .*?: median (\d+) MHz \(min (\d+), max (\d+)\)', txt)
        work = tempfile.mkdtemp(prefix='tc_', dir='/tmp')
        fetch = None
        try:
            r2 = subprocess.run([prof, '--pmc', 'FETCH_SIZE', '--output-format', 'csv', '-d', work, '--', sys.executable, os.path.join(ROOT, 'scripts', 'gemm_one.py'), str(M), str(N), str(K), str(epi)],
                                env=env, cwd='/tmp', capture_output=True, text=True, timeout=300)
            files = glob.glob(os.path.join(work, '**', '*counter_collection.csv'), recursive=True)
            vals = [float(row['Counter_Value']) for f in files for row in csv.DictReader(open(f)) if row['Counter_Name'] == 'FETCH_SIZE' and 'gemm256_kernel' in row['Kernel_Name']]
            if vals:
                fetch = 2 * 1024 * sum(vals[1:]) / max(len(vals) - 1, 1)          # skip the first (cold) launch
        finally:
            shutil.rmtree(work, ignore_errors=True)
        alg = 2.0 * (M * K + N * K)
        row = {'variant': label, 'flags': flags, 'shape': sname, 'M': M, 'N': N, 'K': K, 'ms': ms, 'tflops': round(2.0 * M * N * K / ms / 1e9, 1) if ms else None,
               'in_kernel_clock_mhz': {'median': int(clk.group(1)), 'min': int(clk.group(2)), 'max': int(clk.group(3))} if clk else None,
               'fetch_bytes_per_launch': fetch, 'fetch_over_algorithmic_reads': round(fetch / alg, 2) if fetch else None}
        out['rows'].append(row)
        print(json.dumps(row), flush=True)
        if not ms:
            print(txt[-600:], r.stderr[-600:])
os.makedirs(os.path.join(ROOT, 'profiles', 'round5'), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, 'profiles', 'round5', 'traffic_clock.json'), 'w'), indent=1)
