"""Integer model of index maps in msm.hip that the device code relies on and no GPU test looks at directly.  No GPU.

  * k_msm_fold: which column block a workgroup takes.  The launch is 64 ns row blocks followed by 64 ns column blocks
    (ns slots of 128 rows x 256 columns, a column block = 4 columns); consecutive workgroups go to consecutive XCDs
    (blockIdx % 8), and the eight column blocks that share a 128-byte line of the bucket sums (32 columns x 4 bytes) must
    sit on one XCD, next to each other in its dispatch order:  cb = (ci % 8) * 8 ns + ci / 8.
  * fold_member_slot: member m of bit plane b of a weighted sum is the m-th weight with bit b set.
"""
import pytest


def column_block(ci, ns):
    return (ci & 7) * (8 * ns) + (ci >> 3)          # k_msm_fold, the else branch


@pytest.mark.parametrize("ns", [1, 2, 3, 5, 7, 9])   # 9: the whole key range; fewer: the two-part flow's launches
def test_fold_column_blocks_are_a_permutation_grouped_by_xcd(ns):
    total = 64 * ns
    assert (64 * ns) % 8 == 0                        # the row blocks in front do not shift the XCD of a column block
    cbs = [column_block(ci, ns) for ci in range(total)]
    assert sorted(cbs) == list(range(total))
    for xcd in range(8):
        mine = [column_block(ci, ns) for ci in range(xcd, total, 8)]      # dispatch order on this XCD
        assert mine == list(range(8 * ns * xcd, 8 * ns * (xcd + 1)))       # a contiguous run, neighbours back to back
    # blocks sharing a 128-byte line: same slot, lb // 8 equal - all on one XCD
    xcd_of = {column_block(ci, ns): ci & 7 for ci in range(total)}
    for blk in range(ns):
        for line in range(8):
            assert len({xcd_of[blk * 64 + line * 8 + j] for j in range(8)}) == 1


def member_weight(b, m):
    """fold_member_slot's w for b < 8: insert a 1 at bit b of m."""
    return ((m >> b) << (b + 1)) | (1 << b) | (m & ((1 << b) - 1))


@pytest.mark.parametrize("H", [128, 256])
def test_plane_members_enumerate_the_weights_with_bit_b(H):
    for b in range(8):
        if (1 << b) >= H:
            continue
        want = [w for w in range(H) if (w >> b) & 1]
        got = [member_weight(b, m) for m in range(H // 2)]
        assert got == want
