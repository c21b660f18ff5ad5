#!/usr/bin/env python3
"""Dev aid: list the loops (backward branches) of one kernel in a hipcc -S listing with their instruction mix.
usage: asm_loops.py build/render_wave.s _Z15skr_leaf_kernelILb0EEv12RenderParams"""
import re
import sys

src, name = sys.argv[1], sys.argv[2]
text = open(src).read().split('\n')
start = next(i for i, l in enumerate(text) if l.startswith(name + ':'))
end = next(i for i in range(start, len(text)) if 's_endpgm' in text[i])
lines = text[start:end + 1]
labels = {}
for i, l in enumerate(lines):
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m:
        labels[m.group(1)] = i
loops = []
for i, l in enumerate(lines):
    m = re.search(r's_cbranch_\w+\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)', l)
    if m:
        t = m.group(1) or m.group(2)
        if t in labels and labels[t] < i:
            loops.append((labels[t], i, t))
print('kernel lines', len(lines))
for a, b, t in sorted(loops):
    ins = [x for x in lines[a:b + 1] if x.startswith('\t') and not x.strip().startswith(('.', ';'))]
    cnt = lambda p: sum(1 for x in ins if re.match(r'\s+' + p, x))
    print('%-12s %5d-%5d n=%4d valu=%4d pk=%3d f64=%3d trans=%3d ds=%3d salu=%4d scratch=%2d global=%2d' % (
        t, a, b, len(ins), cnt('v_'), cnt('v_pk'), sum(1 for x in ins if '_f64' in x),
        cnt(r'v_(sqrt|rcp|rsq|exp|log|sin|cos)_'), cnt('ds_'), cnt('s_'), cnt('scratch'), cnt('global')))
