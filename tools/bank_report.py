#!/usr/bin/env python3
"""tools/bank_report.py file.s [filter] -- per kernel: three-VGPR-source VALU instructions whose sources all have the same register
parity (tools/bank_ubench.hip: those issue at half rate on gfx950), total and inside loops.  v_fmac/v_pk_* excluded/included per
the rule table below."""
import re
import sys

THREE = re.compile(r'^(v_bitop3_b32|v_fma_f32|v_mad_u32_u24|v_mad_i32_i24|v_add3_u32|v_lshl_or_b32|v_and_or_b32|v_or3_b32|v_xad_u32|v_perm_b32|v_lshl_add_u32|v_add_lshl_u32|v_med3_f32|v_max3_f32|v_min3_f32|v_bfe_u32|v_bfi_b32|v_alignbit_b32|v_mad_u64_u32)\s+(.*)$')
FMAC = re.compile(r'^(v_fmac_f32(?:_e32|_e64|_dpp)?)\s+(.*)$')


def vregs(ops):
    return [int(m) for m in re.findall(r'(?<![\w\[])v(\d+)\b', ops)]


def analyse(body):
    lines = [l.strip() for l in body.split('\n')]
    label_at = {}
    for i, l in enumerate(lines):
        m = re.match(r'(\.LBB\d+_\d+):', l)
        if m:
            label_at[m.group(1)] = i
    inloop = [0] * len(lines)
    for i, l in enumerate(lines):
        m = re.match(r's_cbranch_\w+ (\.LBB\d+_\d+)', l) or re.match(r's_branch (\.LBB\d+_\d+)', l)
        if m and m.group(1) in label_at and label_at[m.group(1)] < i:
            for j in range(label_at[m.group(1)], i + 1):
                inloop[j] += 1
    stat = {}
    for i, l in enumerate(lines):
        l = l.split(';')[0].strip()
        m = THREE.match(l)
        srcs = None
        if m:
            ops = [o.strip() for o in m.group(2).split(',')]
            srcs = [o.split()[0] for o in ops[1:4] if o]
            name = m.group(1)
        else:
            m = FMAC.match(l)
            if m:
                ops = [o.strip() for o in m.group(2).split(',')]
                srcs = [ops[1], ops[2].split()[0], ops[0]]
                name = 'v_fmac_f32'
        if not srcs:
            continue
        regs = []
        for o in srcs:
            mm = re.match(r'^v(\d+)$', o)
            if mm:
                regs.append(int(mm.group(1)))
        key = (name, inloop[i] > 0)
        t = stat.setdefault(key, [0, 0, 0])
        t[0] += 1
        if len(regs) == 3:
            t[1] += 1
            if len({r & 1 for r in regs}) == 1:
                t[2] += 1
    return stat


def main():
    text = open(sys.argv[1]).read()
    flt = sys.argv[2] if len(sys.argv) > 2 else None
    for m in re.finditer(r'^(_Z\w+):(.*?)\.Lfunc_end', text, re.S | re.M):
        name = m.group(1)
        if flt and flt not in name:
            continue
        st = analyse(m.group(2))
        tot3 = sum(v[1] for k, v in st.items() if k[1])
        bad = sum(v[2] for k, v in st.items() if k[1])
        print("%-60s in loops: %5d three-VGPR-source, %5d same parity (%.0f %%)" % (name[:60], tot3, bad, 100.0 * bad / max(tot3, 1)))
        for (n, lp), v in sorted(st.items()):
            if v[1]:
                print("      %-16s %-8s total %5d  3-vgpr %5d  same-parity %5d" % (n, "loop" if lp else "straight", v[0], v[1], v[2]))


if __name__ == '__main__':
    main()
