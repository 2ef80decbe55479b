"""CPU tier: the LDS layouts of mlkem_kpke2.hpp restated in Python — every layout covers the 128 pairs exactly once, and the
exchange buffer's swizzle is free of bank conflicts for the access groups the LDS uses on MI355X (MI355X_MICROARCH.md, LDS):
ds_read_b64 = 2 groups of 32 lanes over 64 banks; ds_write_b64 (and the ds_read2 forms) = 4 groups of 16 contiguous lanes over
32 banks.  The functions below mirror k2_blk / k2_lc_hi / k2_pair / k2_slot line by line."""


def k2_blk(t):
    return (((t >> 2) & 3) << 3) | ((t >> 4) << 2) | (t & 3)


def k2_lc_hi(t):
    return (((t >> 2) & 3) << 2) | ((t >> 4) << 1) | ((t >> 1) & 1)


def k2_pair(layout, t, j):
    if layout == "LA":
        return (j << 5) | t
    if layout == "LB":
        return ((t >> 3) << 5) | (j << 3) | (t & 7)
    if layout == "LC":
        return (k2_lc_hi(t) << 3) | (j << 1) | (t & 1)
    return (k2_blk(t) << 2) | j


def k2_slot(pair):
    return pair ^ (0x15 if pair & 64 else 0) ^ (0x0A if pair & 32 else 0)


LAYOUTS = ("LA", "LB", "LC", "NAT")


def test_every_layout_is_a_permutation_of_the_pairs_and_slots():
    for lay in LAYOUTS:
        pairs = sorted(k2_pair(lay, t, j) for t in range(32) for j in range(4))
        assert pairs == list(range(128)), lay
    assert sorted(k2_slot(p) for p in range(128)) == list(range(128))
    assert sorted(k2_blk(t) for t in range(32)) == list(range(32))
    # lanes 4m .. 4m+3 own consecutive blocks (the codecs' lane pairs and quads are lane-adjacent)
    assert all(k2_blk(4 * m + i) == k2_blk(4 * m) + i for m in range(8) for i in range(4))


def test_register_bits_are_the_two_butterfly_bits_of_the_layout():
    """pair = idx7..idx1: the register index j must be (idx7, idx6) / (idx5, idx4) / (idx3, idx2) / (idx2, idx1)"""
    shift = {"LA": 5, "LB": 3, "LC": 1, "NAT": 0}
    for lay in LAYOUTS:
        for t in range(32):
            for j in range(4):
                assert (k2_pair(lay, t, j) >> shift[lay]) & 3 == j


def test_exchange_buffer_is_bank_conflict_free_for_reads_and_writes():
    for lay in LAYOUTS:
        for j in range(4):
            # ds_read_b64: 32 lanes per group, banks = dword address mod 64, a lane touches two consecutive dwords
            banks = [b for t in range(32) for b in ((2 * k2_slot(k2_pair(lay, t, j))) % 64, (2 * k2_slot(k2_pair(lay, t, j)) + 1) % 64)]
            assert len(set(banks)) == 64, (lay, j, "read")
            # ds_write_b64 / ds_read2_b64: 16 contiguous lanes per group, banks mod 32
            for g in range(2):
                banks = [b for t in range(16 * g, 16 * g + 16)
                         for b in ((2 * k2_slot(k2_pair(lay, t, j))) % 32, (2 * k2_slot(k2_pair(lay, t, j)) + 1) % 32)]
                assert len(set(banks)) == 32, (lay, j, g, "write")
