"""The LDS layouts of the split LeNet kernels against the gfx950 bank rules (MI355X_MICROARCH.md §LDS), on the CPU:
a wave's access is served in fixed lane groups, one LDS cycle per group when its lanes hit distinct banks (identical addresses
broadcast) —
  ds_read_b128: four 16-lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, {32-35, 44-47, 52-59}, {36-43, 48-51, 60-63}, bank =
                (addr / 4) mod 64, i.e. sixteen 16-byte slots per group;
  ds_read_b64:  two 32-lane halves, thirty-two 8-byte slots.
The address formulas below are the kernels' own (gpd_amd/csrc/lenet_fast.hip), restated lane by lane; the constants they use are
checked against the source text.  What this pins: conv1's pixel fragments are conflict free in all seven k-steps (the 18 % conflict share its SQ counters show come
from the pixel-major turn of the next image, not from the MFMA operands); every conv2 fragment read is conflict free except the one entry whose lane groups differ by a channel
group; every ip1 fragment read is conflict free under the (4 - (row >> 2)) & 3 swizzle, and is 2-way without it."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = open(os.path.join(ROOT, "gpd_amd", "csrc", "lenet_fast.hip")).read()

B128_GROUPS = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)), list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
               list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)), list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))]


def _cycles(addrs, groups, width):
    """LDS cycles of one wave access: per lane group the largest number of DISTINCT addresses on one slot of `width` bytes"""
    slots = 256 // width
    total = 0
    for g in groups:
        by_slot = {}
        for l in g:
            by_slot.setdefault((addrs[l] // width) % slots, set()).add(addrs[l])
        total += max(len(v) for v in by_slot.values())
    return total


def test_conv1_pixel_fragments():
    assert "constexpr int F1_PITCH = 72;" in SRC and "(g == 0 ? 0 : g == 1 ? 2 : g == 2 ? 1 : 3) * 5 + ks" in SRC
    P = 72
    kyg = [0, 2, 1, 3]
    for trow, tcol in ((0, 0), (13, 3), (27, 6)):
        for ks in range(7):
            addrs = []
            for l in range(64):
                j, q = l & 15, l >> 4
                base = ((2 * trow + ((j >> 1) & 1)) * P + 8 * tcol + 2 * (j >> 2) + (j & 1)) * 16
                a = base + kyg[q] * P * 16 + ks * 16 if ks < 5 else base + 4 * P * 16 + q * 16 if ks == 5 else base + 4 * P * 16 + 64
                addrs.append(a)
            c = _cycles(addrs, B128_GROUPS, 16)
            assert c == 4, (ks, c)  # (step 5 pairs taps one column apart: the lanes' overlapping pixels are identical addresses — broadcast)
    # a pitch of 64 or 60 pixels (no padding / power of two) would conflict in every step
    for pitch in (60, 64):
        addrs = [(((l >> 1) & 1) * pitch + 2 * ((l & 15) >> 2) + (l & 1)) * 16 + kyg[l >> 4] * pitch * 16 for l in range(64)]
        assert _cycles(addrs, B128_GROUPS, 16) > 4


def test_conv2_activation_fragments():
    m = re.search(r"constexpr int F2_PP = 28 \* 40;", SRC)
    assert m and "e < 25 ? (e / 5) * 5 + g : e < 30 ? g * 5 + 4" in SRC
    PP, RS = 28 * 40, 3 * 28 * 40
    halves = [list(range(32)), list(range(32, 64))]
    worst = {}
    for e in range(32):
        addrs = []
        for l in range(64):
            j, q = l & 15, l >> 4
            lane_off = ((j >> 1) & 1) * RS + (2 * (j >> 2) + (j & 1)) * 40
            if e < 25:
                a = lane_off + 40 * q + (e // 5) * RS + (e % 5) * 8
            elif e < 30:
                a = lane_off + q * RS + 160 + (e - 25) * 8
            elif e == 30:
                a = lane_off + 4 * RS + 160 + 8 * q
            else:
                a = lane_off + 4 * RS + 160 + 32
            addrs.append(a)
        worst[e] = _cycles(addrs, halves, 8)
    assert all(worst[e] == 2 for e in range(32) if e != 30), worst  # neighbouring columns / rows of one channel group: broadcast + distinct
    assert worst[30] == 4                                            # the one entry whose lane groups are channel groups (8 bytes apart)
    # the straightforward table (lane group = channel group for every tap) is 2-way everywhere: why the slots are arranged by column
    addrs = [((l >> 1) & 1) * RS + (2 * ((l & 15) >> 2) + (l & 1)) * 40 + 8 * (l >> 4) for l in range(64)]
    assert _cycles(addrs, halves, 8) == 4


def test_ip1_fragments_and_the_swizzle():
    # the reader's swizzle, and (round 6: the tiles arrive by LDS-DMA from blocked operands) the same one where the blocks are built
    assert "((g ^ ((4 - (j >> 2)) & 3)) << 4)" in SRC and "((((k >> 3) & 3) ^ ((4 - (rr >> 2)) & 3)) << 4)" in SRC
    for swz, want in ((True, 4), (False, 8)):
        addrs = []
        for l in range(64):
            j, g = l & 15, l >> 4
            ch = g ^ ((4 - (j >> 2)) & 3) if swz else g
            addrs.append(j * 64 + ch * 16)
        assert _cycles(addrs, B128_GROUPS, 16) == want
    # the loader writes the chunk where the reader looks for it
    for row in range(64):
        seen = set()
        for ch in range(4):
            seen.add(row * 64 + ((ch ^ ((4 - (row >> 2)) & 3)) << 4))
        assert len(seen) == 4
