#!/usr/bin/env python3
"""Instruction census of every innermost loop of one kernel in a gfx950 .s file (the Keccak round loops):
loop_census.py file.s kernel-name-substring"""
import collections
import re
import sys

text = open(sys.argv[1]).read()
flt = sys.argv[2]
for name in re.findall(r'^(_Z\w+):', text, re.M):
    if flt not in name:
        continue
    m = re.search(r'^' + re.escape(name) + r':(.*?)\.Lfunc_end', text, re.S | re.M)
    lines = [l.strip() for l in m.group(1).split('\n')]
    label_at = {}
    for i, l in enumerate(lines):
        mm = re.match(r'(\.LBB\d+_\d+):', l)
        if mm:
            label_at[mm.group(1)] = i
    loops = []
    for i, l in enumerate(lines):
        mm = re.match(r's_cbranch_\w+ (\.LBB\d+_\d+)', l) or re.match(r's_branch (\.LBB\d+_\d+)', l)
        if mm and mm.group(1) in label_at and label_at[mm.group(1)] < i:
            loops.append((label_at[mm.group(1)], i))
    inner = [lp for lp in loops if not any(o != lp and lp[0] <= o[0] and o[1] <= lp[1] for o in loops)]
    print(name)
    for a, b in inner:
        c = collections.Counter(l.split()[0] for l in lines[a:b + 1] if l and not l.startswith(('.', ';')) and not l.endswith(':'))
        valu = sum(v for k, v in c.items() if k.startswith('v_'))
        print(f"  loop lines {a}-{b}: VALU {valu}  bitop3 {c['v_bitop3_b32']} alignbit {c['v_alignbit_b32']} mov {c['v_mov_b32_e32']} "
              f"xor {c['v_xor_b32_e32']} scratch {sum(v for k, v in c.items() if k.startswith('scratch'))} salu {sum(v for k, v in c.items() if k.startswith('s_'))} "
              f"ds {sum(v for k, v in c.items() if k.startswith('ds_'))} vmem {sum(v for k, v in c.items() if k.startswith('global'))}")
