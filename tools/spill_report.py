#!/usr/bin/env python3
"""For every kernel of a gfx950 .s file: VGPRs, LDS, spill count, and how many scratch (spill) instructions sit INSIDE
loops (label ... backward branch to that label) as opposed to straight-line code.  Usage: spill_report.py file.s [filter]"""
import re
import sys


def report(text, flt=None):
    meta = {}
    for blk in re.findall(r'- \.agpr_count.*?\.wavefront_size', text, re.S):
        name = re.search(r'\.name:\s+(\S+)', blk).group(1)
        meta[name] = tuple(int(re.search(r'\.%s:\s+(\d+)' % k, blk).group(1))
                           for k in ('vgpr_count', 'group_segment_fixed_size', 'vgpr_spill_count', 'private_segment_fixed_size'))
    for name, (vgpr, lds, spill, scratch) in meta.items():
        if flt and flt not in name:
            continue
        m = re.search(r'^' + re.escape(name) + r':(.*?)\.Lfunc_end', text, re.S | re.M)
        if not m:
            continue
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
        in_loop = out_loop = 0
        per_loop = {}
        for i, l in enumerate(lines):
            if l.startswith('scratch_'):
                hit = [lp for lp in loops if lp[0] <= i <= lp[1]]
                if hit:
                    in_loop += 1
                    per_loop[min(hit, key=lambda lp: lp[1] - lp[0])] = per_loop.get(min(hit, key=lambda lp: lp[1] - lp[0]), 0) + 1
                else:
                    out_loop += 1
        print(f"{name[:66]:66s} vgpr {vgpr:3d} lds {lds:5d} spills {spill:3d} scratch {scratch:4d}B  scratch-ops in loops {in_loop:3d} / straight-line {out_loop:3d}"
              + ("  loops: " + ", ".join(f"[{a}-{b}]x{c}" for (a, b), c in sorted(per_loop.items())) if per_loop else ""))


if __name__ == "__main__":
    report(open(sys.argv[1]).read(), sys.argv[2] if len(sys.argv) > 2 else None)
