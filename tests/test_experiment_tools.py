"""The measurement tools whose outcomes DESIGN.md quotes must keep working (CPU tier): the hand-allocated Keccak generator's
instruction list is simulated against a plain Python Keccak-f[1600], and the VGPR parity pass is run on a small kernel."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "tools", name + ".py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_hand_allocated_keccak_plan_is_keccak_and_has_no_same_parity_sources():
    g = _load("gen_keccak_asm")
    body = [g.gen_round(p) for p in range(4)]
    assert sum(len(b) for b in body) == 4 * 180
    assert sum(g.check_parities(b) for b in body) == 0
    regs = {r for b in body for i in b for r in i[1:4] if i[0] != "iota"} | {i[1] for b in body for i in b if i[0] == "iota"}
    assert min(regs) >= g.V_STATE and max(regs) < g.V_TMP + 20 and len(regs) <= 70
    assert g.self_check(n=3)
    # the reference permutation itself: Keccak-f[1600] of the zero state (first lane of the well-known test vector)
    assert g.keccak_f_ref([0] * 25)[0] == 0xF1258F7940E1DDE7


def test_vgpr_parity_pass_renumbers_only_free_singles_and_lowers_the_count():
    vp = _load("vgpr_parity")
    asm = """	.amdhsa_kernel k_demo
k_demo:
	global_load_dwordx2 v[4:5], v[0:1], off
.LBB0_1:
	v_bitop3_b32 v6, v8, v10, v12 bitop3:0x96
	v_bitop3_b32 v7, v8, v10, v4 bitop3:0x96
	v_fmac_f32_e32 v9, v11, v13
	s_cbranch_scc1 .LBB0_1
	global_store_dword v[0:1], v6, off
.Lfunc_end0:
"""
    new, rep = vp.process(asm)
    st = rep["k_demo"]
    assert st["before"] > st["after"]
    # tuple members and v0-v2 keep their numbers; the instruction count and opcodes are unchanged
    assert "v[4:5]" in new and "v[0:1]" in new
    assert [l.split()[0] for l in new.splitlines() if l.startswith("\t")] == [l.split()[0] for l in asm.splitlines() if l.startswith("\t")]
    # the renaming is a permutation: same number of distinct single registers
    import re
    singles = lambda t: set(re.findall(r"(?<![\w.\[])v(\d+)\b", t))
    assert len(singles(new)) == len(singles(asm))
