#!/usr/bin/env python3
"""Static per-kernel instruction census of a gfx950 .s file, priced with the per-class issue costs measured by
tools/valu_ubench*.hip (profiles/r01_valu_ubench.txt): full-rate VALU 1.1, slow-class VALU 1.73, packed 2.0
ns per wave-instruction per SIMD at 8 waves.  Usage: isa_count.py file.s [name-filter]"""
import collections
import re
import sys

FAST = set("""v_add_f32 v_sub_f32 v_subrev_f32 v_mul_f32 v_fma_f32 v_fmac_f32 v_fmamk_f32 v_fmaak_f32 v_add_u32 v_sub_u32
v_subrev_u32 v_and_b32 v_or_b32 v_xor_b32 v_mov_b32 v_bitop3_b32 v_add_co_u32 v_addc_co_u32 v_not_b32""".split())


def census(text, flt=None):
    rows = []
    for name in re.findall(r'^(_Z\w+):', text, re.M):
        if flt and flt not in name:
            continue
        m = re.search(r'^' + re.escape(name) + r':(.*?)\.Lfunc_end', text, re.S | re.M)
        if not m:
            continue
        ins = [l.split()[0] for l in m.group(1).split('\n')
               if l.startswith('\t') and not l.strip().startswith(('.', ';'))]
        c = collections.Counter(ins)
        units = 0.0
        for k, v in c.items():
            if not k.startswith('v_'):
                continue
            base = re.sub(r'_(e32|e64|sdwa|dpp)$', '', k)
            units += (1.1 if base in FAST else 2.0 if base.startswith('v_pk_') else 1.73) * v
        rows.append((name, sum(v for k, v in c.items() if k.startswith('v_')), int(units),
                     sum(v for k, v in c.items() if k.startswith('ds_')),
                     sum(v for k, v in c.items() if k.startswith(('global_', 'buffer_', 'flat_'))),
                     sum(v for k, v in c.items() if k.startswith('s_')), c))
    return rows


if __name__ == "__main__":
    text = open(sys.argv[1]).read()
    flt = sys.argv[2] if len(sys.argv) > 2 else None
    for name, valu, units, ds, vmem, salu, c in census(text, flt):
        print(f"{name[:70]:70s} VALU {valu:6d} units {units:6d} ds {ds:4d} vmem {vmem:4d} salu {salu:5d}")
        if flt:
            for k, v in c.most_common(25):
                print(f"    {k:28s} {v}")
