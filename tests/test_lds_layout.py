"""CPU property test of conv3's LDS operand image (livetalking_amd/csrc/conv3_mfma.hip): a bank simulator of `ds_read_b128` over the lane
groups and bank rule of MI355X_MICROARCH.md (4 groups of 16 lanes - {0-3,12-15,20-27}, {4-11,16-19,28-31} and the same + 32 -, bank of byte
address a = (a / 4) mod 64, one LDS cycle per group when no bank is asked for two different addresses) applied to the addresses the kernel's
A-fragment reads form, restated here from the kernel (`aj[j][dx]`, `arow`) and the host's tile geometry (`conv3_launch` geom()).

What it pins: every A-fragment read of every tap is conflict-free (4 LDS cycles) on tile rows of 32, 16 and 8 pixels for the 3x3, 1x1, merged
transposed and four-phase upsample convolutions - with the COLUMN key (halves swap where bit 3 of the patch column is set) on 32-pixel rows and
the ROW key (halves swap on odd patch rows; 8-pixel rows padded to a 12-pixel pitch) below - and what the column key alone cost on the narrow
maps before round 4 (8 / 14.7 / 16 cycles).  The stride-2 image is pinned for 32-pixel rows.
"""
GROUP0 = list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28))
GROUP1 = list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32))
GROUPS = [GROUP0, GROUP1, [32 + x for x in GROUP0], [32 + x for x in GROUP1]]
EXT = {"c3": 2, "c1": 0, "ct": 1, "u4": 2}
TAPS = {"c3": [(dy, dx) for dy in range(3) for dx in range(3)], "c1": [(0, 0)], "ct": [(0, 0), (0, 1), (1, 0), (1, 1)],
        "u4": [(dy, dx) for dy in range(3) for dx in range(3)]}


def lds_cycles(addrs):
    """LDS cycles of one wave-wide ds_read_b128 (ideal: 4)."""
    total = 0
    for g in GROUPS:
        per_bank = {}
        for lane in g:
            a = addrs[lane]
            for k in range(4):
                per_bank.setdefault(((a >> 2) + k) & 63, set()).add(a)
        total += max(len(v) for v in per_bank.values())
    return total


def geom(kind, Wo, N, pxw, S=1, new_rule=True):
    """conv3_launch's geom(): tile = 2^l2w x 2^l2h pixels x NB images, patch PH x PW, and the key the host passes."""
    M = 128 * pxw
    lm = M.bit_length() - 1
    cl = lambda v: (v - 1).bit_length()       # noqa: E731
    l2w = min(5, cl(Wo))
    l2h = min(lm - l2w, cl(Wo))
    NB = max(1, min(M >> (l2w + l2h), N))
    ext = EXT[kind] if S == 1 else 2
    PH = ((1 << l2h) - 1) * S + 1 + ext
    PW = ((1 << l2w) - 1) * S + 1 + ext
    swz_x, swz_row = 1, 0
    if S == 2:
        PW = (PW + 1) & ~1
    elif l2w <= 4 and new_rule:
        swz_x, swz_row = 0, 1
        if l2w == 3 and ext > 0:
            PW = 12
    return l2w, l2h, NB, PW, PH, swz_x, swz_row


def a_read_addrs(S, g, wave, j, pxw, dy, dx):
    l2w, l2h, NB, PW, PH, swz_x, swz_row = g
    out = []
    for lane in range(64):
        l31, hh = lane & 31, lane >> 5
        m = (wave * pxw + j) * 32 + l31
        tx, ty, b = m & ((1 << l2w) - 1), (m >> l2w) & ((1 << l2h) - 1), m >> (l2w + l2h)
        if b >= NB:
            tx = ty = b = 0
        prow = (b * PH + ty * S) * PW
        pc = tx * S + dx
        if S == 2:      # unit q = 2 * (column & 1) + half sits at q ^ ((column >> 3) & 3) inside its aligned pixel pair
            q = (2 * (pc & 1) + hh) ^ ((pc >> 3) & 3)
            off = (prow + ((pc & ~1) | (q >> 1))) * 32 + ((q & 1) << 4)
        else:           # aj for tap row 0, flipped on odd tap rows under the row key (arow)
            key = (swz_x & (pc >> 3) & 1) ^ (swz_row & (b * PH + ty) & 1)
            off = (prow + pc) * 32 + ((key ^ hh) << 4)
            if dy & 1:
                off ^= swz_row << 4
        out.append(dy * PW * 32 + off)
    return out


def avg_cycles(kind, Wo, pxw, S=1, new_rule=True, N=16):
    g = geom(kind, Wo, N, pxw, S, new_rule)
    taps = TAPS["c3"] if S == 2 else TAPS[kind]
    c = [lds_cycles(a_read_addrs(S, g, wave, j, pxw, dy, dx)) for wave in range(4) for j in range(pxw) for dy, dx in taps]
    return sum(c) / len(c), max(c)


def test_simulator_on_the_plain_cases():
    assert lds_cycles([lane * 16 for lane in range(64)]) == 4                 # consecutive 16-byte items
    assert lds_cycles([lane * 32 for lane in range(64)]) == 8                 # every other item: 2-way
    assert lds_cycles([lane * 256 for lane in range(64)]) == 64               # one bank set: fully serialised
    assert lds_cycles([0] * 64) == 4                                          # identical addresses broadcast


def test_wide_tiles_conflict_free_with_the_column_key():
    for kind, pxws in (("c3", (1, 2, 4)), ("c1", (2,)), ("ct", (2,)), ("u4", (2,))):
        for Wo in (32, 64, 256):
            for pxw in pxws:
                assert avg_cycles(kind, Wo, pxw) == (4.0, 4), (kind, Wo, pxw)


def test_narrow_tiles_conflict_free_with_the_row_key():
    for kind, pxws in (("c3", (1, 2, 4)), ("c1", (2,)), ("ct", (2,)), ("u4", (2,))):
        for Wo in (16, 8):
            for pxw in pxws:
                assert avg_cycles(kind, Wo, pxw) == (4.0, 4), (kind, Wo, pxw, avg_cycles(kind, Wo, pxw))
    for kind in ("c3", "ct", "u4"):                                           # 4-pixel rows: halved, not gone
        assert avg_cycles(kind, 4, 2)[1] <= 8


def test_what_the_column_key_cost_on_narrow_maps():
    """Rounds 1-3: 2x / 3.7x / 4x the LDS cycles per A-fragment read on 16- / 8- / 4-pixel-wide maps (the U-Net levels of MuseTalk, the small maps of
    Wav2Lip); round 3's PMC showed it as 30 % conflict cycles of the 128-pixel-tile instantiation."""
    assert avg_cycles("c3", 16, 1, new_rule=False)[0] == 8.0
    assert avg_cycles("c3", 8, 1, new_rule=False)[0] > 14.0
    assert avg_cycles("c3", 4, 1, new_rule=False)[0] == 16.0


def test_stride2_image_on_wide_rows():
    for Wo in (32, 64, 128):
        assert avg_cycles("c3", Wo, 2, S=2) == (4.0, 4)


def writer_unit_source(S, g, slot):
    """conv3_item's staging descriptor for the 16-byte unit `slot` of a channel-block image: (image b, patch row, SOURCE patch column, source half)."""
    l2w, l2h, NB, PW, PH, swz_x, swz_row = g
    pix = slot >> 1
    b, rem = divmod(pix, PH * PW)
    py, px = divmod(rem, PW)
    if S == 2:
        q = (2 * (px & 1) + (slot & 1)) ^ ((px >> 3) & 3)
        return b, py, (px & ~1) | (q >> 1), q & 1
    return b, py, px, (slot & 1) ^ (swz_x & (px >> 3) & 1) ^ (swz_row & (b * PH + py) & 1)


def test_reader_meets_writer():
    checked = 0
    for S, kinds in ((1, ("c3", "c1", "ct", "u4")), (2, ("c3",))):
        for kind in kinds:
            taps = TAPS["c3"] if S == 2 else TAPS[kind]
            for Wo in (4, 8, 16, 32, 64):
                for N in (1, 3, 16):
                    for pxw in ((1, 2, 4) if (kind == "c3" and S == 1) else (2,)):
                        for new_rule in (True, False):
                            g = geom(kind, Wo, N, pxw, S, new_rule)
                            l2w, l2h, NB, PW, PH, _, _ = g
                            for wave in range(4):
                                for j in range(pxw):
                                    for dy, dx in taps:
                                        addrs = a_read_addrs(S, g, wave, j, pxw, dy, dx)
                                        for lane, a in enumerate(addrs):
                                            m = (wave * pxw + j) * 32 + (lane & 31)
                                            tx, ty, b = m & ((1 << l2w) - 1), (m >> l2w) & ((1 << l2h) - 1), m >> (l2w + l2h)
                                            if b >= NB:
                                                continue                     # a pixel slot beyond the tile's images: never stored
                                            assert a % 16 == 0
                                            assert writer_unit_source(S, g, a // 16) == (b, ty * S + dy, tx * S + dx, lane >> 5), (S, kind, Wo, N, pxw, new_rule, lane, dy, dx)
                                            checked += 1
    assert checked > 100000
