"""Instruction mix of the loops of one kernel in the assembly tools/valu_by_line.py left in /tmp/valu_by_line.s.
Usage: python tools/valu_by_line.py <src> <mangled-substring> 3 && python tools/loop_isa.py <mangled-substring>"""
import re, sys, tempfile, os
from collections import Counter
s = open(os.path.join(tempfile.gettempdir(), "valu_by_line.s")).read()
names = [m.group(1) for m in re.finditer(r'^(_Z\S+):', s, re.M) if sys.argv[1] in m.group(1)]
a = s.index(names[0] + ':'); a = s.index('\n', a); e = s.index('.Lfunc_end', a)
labels = {}; ins = []
for l in s[a:e].split('\n'):
    t = l.strip()
    if re.match(r'\.LBB\d+_\d+:', t): labels[t.split(':')[0]] = len(ins); continue
    if l.startswith('\t') and t and not t.startswith(('.', ';')): ins.append(t)
for i, t in enumerate(ins):
    m = re.match(r's_cbranch_\w+\s+(\.LBB\d+_\d+)', t)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        loop = ins[labels[m.group(1)]:i + 1]
        c = Counter(x.split()[0] for x in loop)
        print("loop", m.group(1), "instrs", len(loop), "valu", sum(v for k, v in c.items() if k.startswith('v_')), "vmem",
              sum(v for k, v in c.items() if k.startswith(('global_', 'buffer_'))), "lds", sum(v for k, v in c.items() if k.startswith('ds_')),
              "branches", sum(v for k, v in c.items() if k.startswith('s_cbranch')))
        if len(sys.argv) > 2:
            for k, v in c.most_common(int(sys.argv[2])): print("  ", v, k)
