#!/usr/bin/env python3
"""Table of tools/probes/tile_sweep_b1.sh results: python tools/probes/sweep_table.py gpurun_out/<tag>"""
import re
import sys
import os
d0 = sys.argv[1]
names = ['base', 'tile1', 'tile2', 'tile3', 'tile4', 'pre1', 'pre2', 'pre3', 'gemm1', 'gemm2', 'gemm3', 'gemm4']
data, layers = {}, []
for n in names:
    d = {}
    for l in open(os.path.join(d0, 'sweep_layers_%s.txt' % n)):
        m = re.match(r'(\S+)\s+M=\s*(\d+) N=\s*(\d+) K=\s*(\d+)\s+([\d.]+) us\s+([\d.]+) TF\s+tile (.*)', l)
        if m:
            d[m.group(1)] = (float(m.group(5)), m.group(7))
            if n == 'base':
                layers.append((m.group(1), m.group(2), m.group(3), m.group(4)))
        elif l.startswith('total'):
            d['total'] = (float(l.split()[1]), '')
    data[n] = d
print('%-10s %7s %5s %5s ' % ('layer', 'M', 'N', 'K') + ' '.join('%7s' % n for n in names))
for (L, M, N, K) in layers:
    print('%-10s %7s %5s %5s ' % (L, M, N, K) + ' '.join('%7.1f' % data[n][L][0] for n in names) + '  ' + data['base'][L][1][:24])
print('%-30s' % 'total' + ' '.join('%7.0f' % data[n]['total'][0] for n in names))
