"""The statement behind gpd_amd/csrc/preprocess.hip, checked on the CPU: what libstdc++'s std::set does under the
reference's "differs" comparator (cloud.h:105-122, cloud.cpp:286-348) is decided by the tree's LEFT SPINE alone —

  * a point is dropped iff its voxel equals the voxel of a spine node,
  * a kept point becomes the new leftmost node (iteration order = reverse insertion order),
  * the spine after _Rb_tree_insert_and_rebalance depends only on (voxel, colour, colour of the right child) of the
    spine nodes, and which node leaves the spine at the m-th insertion depends on m alone.

The model below is the one the kernel runs (depth-ordered spine, two colour masks); the oracle's voxelize IS std::set."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def spine_model(keys):
    """keys: int array [n, 3] in insertion order -> (kept indices in std::set iteration order, ops table)."""
    spine = []            # voxel keys by depth, root first
    C = R = 0             # bit d: node d is red / its right child is red
    kept, ops = [], []
    for i, k in enumerate(map(tuple, keys)):
        if k in spine:
            continue
        kept.append(i)
        L = len(spine)
        spine.append(k)
        C |= 1 << L
        R &= ~(1 << L)
        x, gone = L, -1
        while x > 0 and (C >> (x - 1)) & 1:
            g = x - 2     # a red parent is not the root
            if (R >> g) & 1:            # red uncle: recolour, continue from the grandparent
                C = (C & ~(1 << (x - 1))) | (1 << g)
                R &= ~(1 << g)
                x = g
            else:                        # black uncle: right rotation at the grandparent, which leaves the spine
                C &= ~(1 << (x - 1))
                R |= 1 << (x - 1)
                low = (1 << g) - 1
                C = (C & low) | ((C >> 1) & ~low)
                R = (R & low) | ((R >> 1) & ~low)
                del spine[g]
                gone = g
                break
        C &= ~1           # the root is black
        ops.append(gone)
    return kept[::-1], ops


def _keys(xyz, cell):
    xyz = np.ascontiguousarray(xyz, np.float32)
    return np.floor((xyz - xyz.min(0)) / np.float32(cell)).astype(np.int64)


def test_krylon_known_answer(oracle_mod):
    xyz = np.load(os.path.join(GOLD, "krylon_xyz.npz"))["xyz"]
    _, src = oracle_mod.voxelize(xyz, 0.003)
    kept, _ = spine_model(_keys(xyz, 0.003))
    assert len(kept) == 3366 and kept[:5] == [4466, 4464, 4459, 4458, 4457]
    assert kept == src.tolist()


@pytest.mark.parametrize("kind", ["scan", "random", "few_voxels"])
def test_model_equals_std_set(oracle_mod, kind):
    rng = np.random.RandomState({"scan": 1, "random": 2, "few_voxels": 3}[kind])
    n = 20000
    if kind == "scan":
        t = np.arange(n)
        xyz = np.stack([(t % 317) * 0.0011, (t // 317) * 0.0013, 0.4 + 0.01 * np.sin(t * 0.01)], 1) + rng.randn(n, 3) * 2e-4
    elif kind == "random":
        xyz = rng.rand(n, 3) * [0.05, 0.04, 0.03]
    else:
        xyz = rng.randint(0, 3, (n, 3)) * 0.003 + 1e-4
    xyz = np.ascontiguousarray(xyz, np.float32)
    _, src = oracle_mod.voxelize(xyz, 0.003)
    kept, _ = spine_model(_keys(xyz, 0.003))
    assert kept == src.tolist()


def test_the_rebalancing_is_a_function_of_the_count_alone(oracle_mod):
    """Two clouds with nothing in common but the number of kept points rebalance identically — which is why the host can
    tabulate the spine operations once (preprocess.hip spine_ops) and the kernel only appends and closes a gap."""
    rng = np.random.RandomState(4)
    a = np.arange(3000)[:, None] * np.array([[1, 0, 0]])          # all voxels distinct: every point kept
    b = rng.permutation(3000)[:, None] * np.array([[0, 7, 1]])
    _, ops_a = spine_model(a)
    _, ops_b = spine_model(b)
    assert ops_a == ops_b and len(ops_a) == 3000
    # the spine stays logarithmic: 2 log2(m + 1) bounds a red-black tree's height
    depth = 0
    for m, g in enumerate(ops_a):
        depth += 1 if g < 0 else 0
        assert depth <= 2 * np.log2(m + 2) + 1
