#!/usr/bin/env python3
"""tools/gen_keccak_asm.py -- EXPERIMENT (measured, not adopted: profiles/r03_vgpr_bank_ubench.txt).  Generator of
crystals-kyber_amd/csrc/mlkem_keccak_gfx950.inc: Keccak-f[1600] (sha3.c:15-216 of the reference) as ONE hand-allocated gfx950
assembly block, 24 rounds = 6 iterations of a 4-round body.  To try it again: run this script, then in mlkem_device.hpp wrap the
block as keccak_f1600() -- MLKEM_KECCAK_DECL(s); asm volatile("s_mov_b32 " S_OFF ", 0; loop: s_load_dwordx8 " S_RC ", %[rc], "
S_OFF "; " MLKEM_KECCAK_BODY "s_add_u32 ...32; s_cmp_lg_u32 ..., 0xc0; s_cbranch_scc1 loop" : MLKEM_KECCAK_OPERANDS : [rc] "s"(KECCAK_RC)
: the two clobber lists, "scc"); MLKEM_KECCAK_STORE(s).  Outcome on MI355X: bit-exact (73 GPU parity tests), -3 % on the bare
permutation at 6-8 waves per SIMD, but +1.7 % on the ML-KEM-768 step (70 pinned registers under the 80-register budget).

Why hand allocation: a full-rate VALU instruction with three VGPR sources issues at half rate on MI355X when its three
register numbers have the same parity (tools/bank_ubench.hip, profiles/r03_vgpr_bank_ubench.txt); the compiler's allocation of the
C++ round leaves 27-41 % of the 120 v_bitop3_b32 of a round in that state.  Here every register is chosen:

  state   50 registers, slot (x, y) -> pair P = V_STATE + 2 (x + 5 y); rows with even y hold (lo, hi) = (P, P + 1), rows with
          odd y hold (lo, hi) = (P + 1, P): every column then has registers of both parities in both halves;
  C[x]    (column parities) even registers, R[x] = rotl(C[x], 1) odd registers: A ^ C[x-1] ^ R[x+1] never has equal parities;
  B row   (rho/pi outputs of one chi row) aliases C/R registers with the parity pattern (e, o, e, o, o) for the low halves and
          (o, e, o, e, e) for the high halves: no three cyclically consecutive B registers share a parity.
  theta   C[x] = (s0 ^ s1 ^ s2) then (C ^ s3 ^ s4): parities (e, o, e) / (C = e, o, e) for lo, (o, e, o) / (e, e, o) for hi.

In-place rounds: the chi outputs of row y' go into the slots of the five lanes rho/pi just consumed, the output with column X
into the freed slot of column X; lane (X, y) of round p+1 then sits in the slot where lane (X, X + 2 y) sat in round p -- a map
of period 4, hence the 4-round body (Keccak implementation overview, "in-place processing").  70 VGPRs in all.
180 VALU per round as before: 120 v_bitop3_b32 + 58 v_alignbit_b32 + 2 v_xor_b32 (iota, constants through s_load_dwordx8).

  gen_keccak_asm.py            write the .inc
  gen_keccak_asm.py --check    simulate the generated instruction list on random states against hashlib-independent Python Keccak
"""
import os
import sys

V_STATE = 10      # v10..v59
V_TMP = 60        # v60..v79: C even, R odd; B row aliases the first ten
S_OFF = 87        # s87: byte offset into the round-constant table
S_RC = 88         # s[88:95]: (lo, hi) of the four round constants of one iteration

RHO = [0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14]   # index x + 5 y
RC = [0x0000000000000001, 0x0000000000008082, 0x800000000000808A, 0x8000000080008000, 0x000000000000808B, 0x0000000080000001,
      0x8000000080008081, 0x8000000000008009, 0x000000000000008A, 0x0000000000000088, 0x0000000080008009, 0x000000008000000A,
      0x000000008000808B, 0x800000000000008B, 0x8000000000008089, 0x8000000000008003, 0x8000000000008002, 0x8000000000000080,
      0x000000000000800A, 0x800000008000000A, 0x8000000080008081, 0x8000000000008080, 0x0000000080000001, 0x8000000080008008]


def slot_regs(x, y):
    p = V_STATE + 2 * (x + 5 * y)
    return (p, p + 1) if y % 2 == 0 else (p + 1, p)       # (lo, hi)


def c_regs(x):
    return (V_TMP + 4 * x, V_TMP + 4 * x + 2)


def r_regs(x):
    return (V_TMP + 4 * x + 1, V_TMP + 4 * x + 3)


B_REGS = [(60, 61), (63, 62), (64, 65), (67, 66), (69, 68)]   # (lo, hi): lo parities e o e o o, hi parities o e o e e


def slot_of(p, x, y):
    """slot holding lane (x, y) at the start of round phase p"""
    for _ in range(p):
        y = (x + 2 * y) % 5
    return x, y


def rot_ops(dst, src, r):
    """instructions for (dlo, dhi) = rotl64((slo, shi), r), 0 < r < 64, r != 32"""
    (dlo, dhi), (slo, shi) = dst, src
    if r < 32:
        return [("alignbit", dlo, slo, shi, 32 - r), ("alignbit", dhi, shi, slo, 32 - r)]
    return [("alignbit", dlo, shi, slo, 64 - r), ("alignbit", dhi, slo, shi, 64 - r)]


def gen_round(p):
    ins = []
    # theta: column parities of the five slots of each column (which lane sits in which slot does not matter here)
    for x in range(5):
        for h in (0, 1):
            c = c_regs(x)[h]
            ins.append(("xor3", c, slot_regs(x, 0)[h], slot_regs(x, 1)[h], slot_regs(x, 2)[h]))
            ins.append(("xor3", c, c, slot_regs(x, 3)[h], slot_regs(x, 4)[h]))
    for x in range(5):
        ins += rot_ops(r_regs(x), c_regs(x), 1)
    for x in range(5):
        for y in range(5):
            for h in (0, 1):
                a = slot_regs(x, y)[h]
                ins.append(("xor3", a, a, c_regs((x + 4) % 5)[h], r_regs((x + 1) % 5)[h]))
    # rho + pi + chi, one output row at a time
    for yo in range(5):
        b = []
        for xo in range(5):
            lx, ly = (xo + 3 * yo) % 5, xo                  # source lane of B[xo, yo]
            src = slot_regs(*slot_of(p, lx, ly))
            r = RHO[lx + 5 * ly]
            if r == 0:
                b.append(src)
            else:
                ins += rot_ops(B_REGS[xo], src, r)
                b.append(B_REGS[xo])
        order = [1, 2, 3, 4, 0]
        for X in order:
            dst = slot_regs(*slot_of(p, X, (X + 2 * yo) % 5))     # = slot of lane (X, yo) in phase p + 1
            for h in (0, 1):
                ins.append(("chi", dst[h], b[X][h], b[(X + 1) % 5][h], b[(X + 2) % 5][h]))
        if yo == 0:
            d = slot_regs(0, 0)
            ins.append(("iota", d[0], S_RC + 2 * p))
            ins.append(("iota", d[1], S_RC + 2 * p + 1))
    return ins


def check_parities(ins):
    bad = 0
    for i in ins:
        if i[0] in ("xor3", "chi"):
            if len({i[2] & 1, i[3] & 1, i[4] & 1}) == 1:
                bad += 1
    return bad


def emit(ins_by_phase):
    lines = []
    for p, ins in enumerate(ins_by_phase):
        first_iota = True
        for i in ins:
            if i[0] == "xor3":
                lines.append("v_bitop3_b32 v%d, v%d, v%d, v%d bitop3:0x96" % i[1:])
            elif i[0] == "chi":
                lines.append("v_bitop3_b32 v%d, v%d, v%d, v%d bitop3:0xd2" % i[1:])
            elif i[0] == "alignbit":
                lines.append("v_alignbit_b32 v%d, v%d, v%d, %d" % i[1:])
            elif i[0] == "iota":
                if p == 0 and first_iota:
                    lines.append("s_waitcnt lgkmcnt(0)")
                first_iota = False
                lines.append("v_xor_b32 v%d, s%d, v%d" % (i[1], i[2], i[1]))
    return lines


# ------------------------------------------------------------------------------------------------
# reference permutation + simulator of the generated list (for --check and the CPU test)
# ------------------------------------------------------------------------------------------------
def keccak_f_ref(A):
    M = (1 << 64) - 1
    rol = lambda v, n: ((v << n) | (v >> (64 - n))) & M if n else v
    A = list(A)
    for rnd in range(24):
        C = [A[x] ^ A[x + 5] ^ A[x + 10] ^ A[x + 15] ^ A[x + 20] for x in range(5)]
        D = [C[(x + 4) % 5] ^ rol(C[(x + 1) % 5], 1) for x in range(5)]
        A = [A[i] ^ D[i % 5] for i in range(25)]
        B = [0] * 25
        for x in range(5):
            for y in range(5):
                B[y + 5 * ((2 * x + 3 * y) % 5)] = rol(A[x + 5 * y], RHO[x + 5 * y])
        A = [B[i] ^ (~B[(i % 5 + 1) % 5 + 5 * (i // 5)] & B[(i % 5 + 2) % 5 + 5 * (i // 5)]) & M for i in range(25)]
        A[0] ^= RC[rnd]
    return A


def simulate(A):
    v = {}
    for x in range(5):
        for y in range(5):
            lo, hi = slot_regs(x, y)
            v[lo], v[hi] = A[x + 5 * y] & 0xffffffff, A[x + 5 * y] >> 32
    body = [gen_round(p) for p in range(4)]
    for it in range(6):
        s = {}
        for j in range(4):
            s[S_RC + 2 * j], s[S_RC + 2 * j + 1] = RC[4 * it + j] & 0xffffffff, RC[4 * it + j] >> 32
        for ins in body:
            for i in ins:
                if i[0] == "xor3":
                    v[i[1]] = v[i[2]] ^ v[i[3]] ^ v[i[4]]
                elif i[0] == "chi":
                    v[i[1]] = (v[i[2]] ^ (~v[i[3]] & v[i[4]])) & 0xffffffff
                elif i[0] == "alignbit":
                    v[i[1]] = (((v[i[2]] << 32) | v[i[3]]) >> i[4]) & 0xffffffff
                elif i[0] == "iota":
                    v[i[1]] ^= s[i[2]]
    out = []
    for y in range(5):
        for x in range(5):
            lo, hi = slot_regs(x, y)
            out.append(v[lo] | (v[hi] << 32))
    return out


def self_check(n=20, seed=7):
    import random
    rng = random.Random(seed)
    for _ in range(n):
        A = [rng.getrandbits(64) for _ in range(25)]
        if simulate(A) != keccak_f_ref(A):
            return False
    return True


def main():
    body = [gen_round(p) for p in range(4)]
    counts = {}
    for ins in body:
        for i in ins:
            counts[i[0]] = counts.get(i[0], 0) + 1
    bad = sum(check_parities(ins) for ins in body)
    if "--check" in sys.argv:
        print("instructions per 4 rounds:", counts, " same-parity three-source:", bad, " simulation == reference:", self_check())
        return
    assert bad == 0 and self_check(4)
    lines = emit(body)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = os.path.join(root, "crystals-kyber_amd", "csrc", "mlkem_keccak_gfx950.inc")
    with open(out, "w") as f:
        f.write("// GENERATED by tools/gen_keccak_asm.py -- do not edit.  Keccak-f[1600], 4-round body of the hand-allocated gfx950 form\n")
        f.write("// (register plan and the reason for it: see the generator's header).  %d instructions per 4 rounds: %s\n" % (
            sum(counts.values()), ", ".join("%d %s" % (v, k) for k, v in sorted(counts.items()))))
        f.write("// the 50 state words pinned to their registers (register asm variables are honoured for inline-asm operands)\n")
        f.write("#define MLKEM_KECCAK_DECL(s) \\\n")
        rows = []
        for i in range(25):
            lo, hi = slot_regs(i % 5, i // 5)
            rows.append('    register uint32_t kl%d asm("v%d") = (s).lo[%d]; register uint32_t kh%d asm("v%d") = (s).hi[%d];' % (i, lo, i, i, hi, i))
        f.write(" \\\n".join(rows) + "\n")
        f.write("#define MLKEM_KECCAK_OPERANDS %s\n" % ", ".join('"+v"(kl%d), "+v"(kh%d)' % (i, i) for i in range(25)))
        f.write("#define MLKEM_KECCAK_STORE(s) \\\n")
        f.write(" \\\n".join("    (s).lo[%d] = kl%d; (s).hi[%d] = kh%d;" % (i, i, i, i) for i in range(25)) + "\n")
        f.write('#define MLKEM_KECCAK_TMP_CLOBBERS %s\n' % ", ".join('"v%d"' % r for r in range(V_TMP, V_TMP + 20)))
        f.write('#define MLKEM_KECCAK_SGPR_CLOBBERS %s\n' % ", ".join('"s%d"' % r for r in range(S_OFF, S_RC + 8)))
        f.write('#define MLKEM_KECCAK_S_OFF "s%d"\n#define MLKEM_KECCAK_S_RC "s[%d:%d]"\n' % (S_OFF, S_RC, S_RC + 7))
        f.write("#define MLKEM_KECCAK_BODY \\\n")
        f.write(" \\\n".join('    "%s\\n\\t"' % l for l in lines) + "\n")
    print("wrote", out, counts)


if __name__ == "__main__":
    main()
