"""Static VALU cost of one kernel by source line: compiles a kernels_fast/ or kernels/ file to gfx950 assembly with line tables and
sums VALU instructions per .loc (quarter-rate ops weighted 4). Usage: python tools/valu_by_line.py <source.hip> <mangled-name-substring> [top]"""
import os, re, subprocess, sys, tempfile
from collections import Counter
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from plainrenderer_amd import build as b

def main(src, pat, top=40):
    flags = b.FLAGS
    if os.sep + "kernels_fast" + os.sep in os.path.abspath(src):
        flags = [b.FAST_FLAGS_REPLACE.get(f, f) for f in b.FLAGS] + b.FAST_FLAGS_EXTRA
    out = os.path.join(tempfile.gettempdir(), "valu_by_line.s")
    subprocess.run([b.HIPCC, "-x", "hip"] + flags + ["-gline-tables-only", "--cuda-device-only", "-S", "-o", out, src], check=True, capture_output=True)
    s = open(out).read()
    names = [m.group(1) for m in re.finditer(r'^(_Z\S+):', s, re.M) if pat in m.group(1)]
    if not names:
        raise SystemExit("no kernel matches " + pat)
    n = names[0]
    a = s.index(n + ':'); e = s.index('.Lfunc_end', a)
    files = {int(m.group(1)): (m.group(3) or m.group(2)) for m in re.finditer(r'\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', s)}
    Q = ('v_rcp', 'v_rsq', 'v_sqrt', 'v_log', 'v_exp', 'v_sin', 'v_cos', 'v_mul_lo', 'v_mul_hi', 'v_mad_u64', 'v_mad_i64')
    cur = (0, 0); cnt = Counter(); vmem = 0; nins = 0
    for l in s[a:e].split('\n'):
        t = l.strip()
        m = re.match(r'\.loc\s+(\d+)\s+(\d+)', t)
        if m:
            cur = (int(m.group(1)), int(m.group(2))); continue
        if l.startswith('\t') and t and not t.startswith(('.', ';')):
            op = t.split()[0]; nins += 1
            if op.startswith('v_'):
                cnt[cur] += 4 if op.startswith(Q) else 1
            if op.startswith(('global_load', 'global_store', 'buffer_', 'scratch_')):
                vmem += 1
    meta = re.findall(r'; (NumVgprs|Occupancy|ScratchSize): (\d+)', s[e:e + 3000])
    print(n[:100]); print("instructions", nins, "weighted VALU", sum(cnt.values()), "VMEM", vmem, meta)
    cache = {}
    for (f, ln), c in cnt.most_common(int(top)):
        path = files.get(f, '?')
        if path not in cache:
            try: cache[path] = open(path if os.path.isabs(path) else os.path.join(os.path.dirname(src), path)).read().split('\n')
            except Exception: cache[path] = None
        lines = cache[path]
        text = lines[ln - 1].strip()[:120] if lines and 0 < ln <= len(lines) else ''
        print("%5d %-20s:%4d  %s" % (c, os.path.basename(path), ln, text))

if __name__ == "__main__":
    main(*sys.argv[1:])
