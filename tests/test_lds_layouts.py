"""LDS layouts of the Op C kernel (hs_patch_irc.hip) against the gfx950 bank model of tools/lds_conflicts.py (MI355X_MICROARCH.md, LDS
section): the chunk loop's hot accesses -- pw1's h1 stores, the depthwise stage's h1 reads and h2 stores, pw3's transpose reads, the
B-fragment build's skip-vector reads -- are conflict-free by construction (round 2's kernel lost 29 % of its LDS cycles to conflicts,
profiles/round2_pmc_ir_split_M_level4.txt).  The geometry constants are restated in the tool; the kernel static_asserts its own."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))


@pytest.mark.parametrize('rh', [8, 16])
def test_irc_layouts_are_conflict_free(rh):
    import lds_conflicts as L
    g = L.irc_geometry(rh)
    assert g['ps1'] == {8: 208, 16: 336}[rh] and g['ps1'] % 64 == 16 and g['ps1'] > g['npos']      # IrcGeom::PS1
    assert sorted(L.h1_slot(c) for c in range(16)) == list(range(16))
    for p in range(16):                                       # the h2 row swizzle is a permutation of the region's rows for every plane
        assert sorted(t ^ L.h2_swz(p) for t in range(rh)) == list(range(rh))
    pats = L.irc_patterns(rh, verbose=False)
    assert len(pats) > 40
    for name, op, c, ideal in pats:
        if name.startswith('pw3'):
            assert c <= 2 * ideal, (name, op, c, ideal)       # the transpose read has conflict classes the model does not know: bound only
        else:
            assert c == ideal, (name, op, c, ideal)


def test_bank_model_basics():
    import lds_conflicts as L
    # 64 consecutive dwords: conflict-free for every width; a stride of 32 dwords: every lane of a group on one bank
    assert L.cycles('ds_read_b32', [4 * i for i in range(64)]) == (2, 2)
    assert L.cycles('ds_read_b32', [128 * i for i in range(64)]) == (64, 2)
    assert L.cycles('ds_read_b32', [0] * 64) == (2, 2)                      # identical addresses broadcast
    assert L.cycles('ds_read_b128', [16 * i for i in range(64)]) == (4, 4)
    assert L.cycles('ds_write_b128', [16 * i for i in range(64)]) == (8, 8)
