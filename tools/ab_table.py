#!/usr/bin/env python3
"""Condense the A/B log of tools/pc_bench.py runs (sections headed `## <variant> <method> rep<N>`) into one line per
(method, shape, variant): backward us, GB/s and forward us of every repetition."""
import re,sys,collections
d=collections.defaultdict(list)
h=None
for l in open(sys.argv[1]):
    if l.startswith('##'): h=tuple(l.split()[1:3]); continue
    m=re.match(r"\[\s*(\d+) x\s*(\d+)\].*fwd\s+([\d.]+) us.*bwd\s+([\d.]+) us\s+(\d+) GB/s",l)
    if m: d[(h[1],m.group(1)+'x'+m.group(2),h[0])].append((float(m.group(4)),int(m.group(5)),float(m.group(3))))
for k in sorted(d):
    v=d[k]; print(k, ' bwd us', [x[0] for x in v], 'GB/s', [x[1] for x in v], 'fwd us',[x[2] for x in v])
