"""CPU: the numpy restatement of the LaneConv work-item plan covers every edge of the lane graph exactly once,
for any grouping and any source-row capacity (the GPU suite then pins the device plan to it bit for bit)."""
import numpy as np
import pytest

import lc_plan_ref as R


def random_graph(rng, n, sizes, hot=0):
    us, vs = [], []
    for m in sizes:
        u = rng.integers(0, n, m)
        if hot and m:
            u = np.where(rng.random(m) < 0.5, rng.integers(0, hot, m), u)
        us.append(u)
        vs.append(rng.integers(0, n, m))
    return us, vs


@pytest.mark.parametrize("M,cap,groups", [(192, 304, 4), (192, 192, 1), (128, 200, 15), (128, 130, 3)])
def test_reference_plan_is_the_edge_multiset(M, cap, groups):
    rng = np.random.default_rng(M + cap + groups)
    n = 16 * 27 + 5
    us, vs = random_graph(rng, n, (700, 500, 0, 300, 40, 500, 400, 0, 600, 300, 3, 450, 260, 0), hot=12)
    rowptr, col = R.csr_from_coo(us, vs, n)
    n_units = len(us) + 1
    gstart = [(g * n_units + groups - 1) // groups for g in range(groups)] + [n_units]
    plan = R.lc_plan_ref(rowptr, col, n, len(us), M, cap, gstart)
    assert R.plan_edges(plan, col, n, M) == R.csr_edges(rowptr, col, n, len(us))
    n_src = plan["hdr"][:, 1]
    assert n_src.max() <= cap
    # items tile each group's unit range: spans of the items of a group add up to the group's length
    nb = (n + M - 1) // M
    for b in range(nb):
        for g in range(groups):
            u = gstart[g]
            while u < gstart[g + 1]:
                span = int(plan["hdr"][b * R.LC_UNITS + u, 2])
                assert span >= 1
                u += span
            assert u == gstart[g + 1]


def test_chain_graph_dedupes_sources():
    """A chain with dilations: 15 units of a 192-row block name ~300 distinct source rows, not 15 x 192."""
    n = 192 * 3
    us, vs = [], []
    for s in range(6):
        d = 1 << s
        idx = np.arange(n - d)
        us += [idx + d, idx]          # pre: (u = i + d, v = i), suc: (u = i, v = i + d)
        vs += [idx, idx + d]
    half = n // 2
    us += [np.arange(half), np.arange(half, n)]      # left / right: the other half, node-wise
    vs += [np.arange(half, n), np.arange(half)]
    rowptr, col = R.csr_from_coo(us, vs, n)
    plan = R.lc_plan_ref(rowptr, col, n, 14, 192, 304, [0, 15])
    live = plan["hdr"][:, 0] > 0
    # 192 own rows + 32 on each side + up to 192 left/right partners: more than one item only where that exceeds 304
    assert plan["hdr"][live, 1].max() <= 304
    assert R.plan_edges(plan, col, n, 192) == R.csr_edges(rowptr, col, n, 14)
