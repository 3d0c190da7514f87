"""Branch instructions of one kernel by source line (with the innermost inlined-at lines), from the assembly tools/valu_by_line.py left
in /tmp/valu_by_line.s. Usage: python tools/valu_by_line.py <src> <mangled-substring> 3 && python tools/branch_sites.py <mangled-substring>"""
import os, re, sys, tempfile
from collections import Counter
s = open(os.path.join(tempfile.gettempdir(), "valu_by_line.s")).read()
names = [m.group(1) for m in re.finditer(r'^(_Z\S+):', s, re.M) if sys.argv[1] in m.group(1)]
a = s.index(names[0] + ':'); e = s.index('.Lfunc_end', a)
files = {int(m.group(1)): (m.group(3) or m.group(2)) for m in re.finditer(r'\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', s)}
cur = None; c = Counter()
for l in s[a:e].split('\n'):
    t = l.strip()
    m = re.match(r'\.loc\s+(\d+)\s+(\d+)\s+(\d+)(.*)', t)
    if m:
        chain = re.findall(r'([\w\.]+):(\d+):\d+', m.group(4))
        cur = "%s:%d %s" % (files.get(int(m.group(1)), '?').split('/')[-1], int(m.group(2)), ' <- '.join(x[0].split('/')[-1] + ':' + x[1] for x in chain[-2:]))
        continue
    if t.startswith('s_cbranch'): c[(t.split()[0], cur)] += 1
print(names[0][:90], "branches:", sum(c.values()))
for k, v in sorted(c.items(), key=lambda kv: -kv[1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 25]: print("%3d %-16s %s" % (v, k[0], k[1]))
