"""CPU check of the fused inverted-residual kernel's matrix-core tile maps (hyperseg_amd/csrc/hs_ir_tiles.h, read back
through the host-only entry point hs_ir_tile_map): every halo position of a region is produced by exactly one live
tile column, and all live columns of a tile are filtered with ONE patch's weights -- for Op D (hyperseg_v0_1.py:205-237:
image-level patch convolutions) that is the patch that OWNS the (reflected) position, which is what the oracle's
patch_inverted_residual_v0 applies; for Op C the region lies inside one patch."""
import numpy as np
import pytest


def reflect(i, n):
    return -i if i < 0 else (2 * (n - 1) - i if i >= n else i)


@pytest.mark.parametrize('reg,mode,pwr', [(16, 0, 16), (8, 0, 8), (8, 1, 4), (16, 1, 8), (16, 1, 16), (16, 1, 4), (8, 1, 8)])
def test_tile_map_covers_the_halo_once(reg, mode, pwr):
    import hyperseg_amd.functional as HF
    nt3, tm = HF.ir_tile_map(reg, mode, pwr)
    hw = reg + 2
    count = np.zeros((hw, hw), dtype=int)
    for t in range(tm.shape[0]):
        for n in range(16):
            u, v, live = tm[t, n]
            assert 0 <= u < hw and 0 <= v < hw
            if live:
                count[u, v] += 1
    assert (count == 1).all()
    assert nt3 == reg * reg // 16
    if mode == 1:
        # the first nt3 tiles are the pixel tiles of pw3: together they cover the interior once, all columns live
        inner = np.zeros((reg, reg), dtype=int)
        for t in range(nt3):
            assert tm[t, :, 2].all()
            for n in range(16):
                inner[tm[t, n, 0] - 1, tm[t, n, 1] - 1] += 1
        assert (inner == 1).all()


@pytest.mark.parametrize('reg,pwr,patch', [(8, 4, 4), (16, 8, 8), (16, 16, 16), (16, 16, 32)])
def test_op_d_tiles_have_one_owner(reg, pwr, patch):
    """Every region of a (3 x 3 regions) image, image-border regions included: the live columns of a tile, mapped through
    reflect padding, lie in one patch."""
    import hyperseg_amd.functional as HF
    _, tm = HF.ir_tile_map(reg, 1, pwr)
    H = W = 3 * reg if patch <= reg else 2 * patch
    for y0 in range(0, H, reg):
        for x0 in range(0, W, reg):
            for t in range(tm.shape[0]):
                owners = set()
                for n in range(16):
                    u, v, live = tm[t, n]
                    if live:
                        yy, xx = reflect(y0 + u - 1, H), reflect(x0 + v - 1, W)
                        owners.add((yy // patch, xx // patch))
                assert len(owners) == 1, (y0, x0, t, owners)
                # column 0 is always live (the kernel reads the tile's owner from it)
                assert tm[t, 0, 2] == 1
