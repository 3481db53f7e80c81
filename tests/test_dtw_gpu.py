"""GPU parity tests for the DTW path (-m gpu).

Oracle = oracle/dtw.py (restatement of fastdtw; PARITY UNPINNED against the
reference, see its header).  Bar: alignment indices bit-exact vs that oracle on
continuous random data, accumulated cost within 1e-12 relative; tie-heavy
inputs are held only to the reference's own assertions (shapes, norm).
"""
import numpy as np
import pytest
import torch

from oracle import dtw as OD

pytestmark = pytest.mark.gpu


def _tracks(rng, t, D):
    return np.cumsum(rng.randn(t, D), 0) * 0.1


def _run_pairs(pairs, radius=1):
    from nnmnkwii_amd import _hip
    N = len(pairs)
    D = pairs[0][0].shape[1]
    Tx = max(len(x) for x, _ in pairs)
    Ty = max(len(y) for _, y in pairs)
    X = np.zeros((N, Tx, D))
    Y = np.zeros((N, Ty, D))
    for n, (x, y) in enumerate(pairs):
        X[n, :len(x)] = x
        Y[n, :len(y)] = y
    lenx = torch.tensor([len(x) for x, _ in pairs], dtype=torch.int32, device="cuda")
    leny = torch.tensor([len(y) for _, y in pairs], dtype=torch.int32, device="cuda")
    pi, pj, pl, cost = _hip.fastdtw_l2(torch.from_numpy(X).cuda(), torch.from_numpy(Y).cuda(), lenx, leny, radius)
    return pi.cpu().numpy(), pj.cpu().numpy(), pl.cpu().numpy(), cost.cpu().numpy()


@pytest.mark.parametrize("radius", [1, 2, 3])
def test_paths_bit_exact_vs_oracle(radius):
    rng = np.random.RandomState(10 + radius)
    sizes = [(1, 1), (1, 7), (2, 2), (3, 3), (2, 40), (40, 3), (5, 7), (20, 31), (64, 50), (33, 17), (100, 120),
             (129, 64), (65, 257), (200, 200), (301, 299), (7, 300)]
    pairs = [(_tracks(rng, tx, 4), _tracks(rng, ty, 4)) for tx, ty in sizes]
    pi, pj, pl, cost = _run_pairs(pairs, radius)
    for n, (x, y) in enumerate(pairs):
        d, path = OD.fastdtw(x, y, radius)
        assert pl[n] == len(path), (n, sizes[n])
        assert np.array_equal(pi[n, :pl[n]], path[:, 0]), (n, sizes[n])
        assert np.array_equal(pj[n, :pl[n]], path[:, 1]), (n, sizes[n])
        assert abs(cost[n] - d) <= 1e-12 * max(d, 1e-300), (n, sizes[n])


@pytest.mark.parametrize("radius,sizes", [
    (30, [(300, 280), (257, 300), (64, 300)]),       # windows too wide for the candidate tables: sequential back-trace
    (8, [(500, 9), (9, 500), (12, 13), (700, 650)]),  # one side shorter than radius + 2: full DTW at level 0 / 1
    (1, [(900, 450), (450, 900), (1000, 1000)]),      # slope 2 windows, 16+ chunks, 63 segments
    (1, [(280, 1), (324, 2), (1, 300), (511, 3)]),    # full DTW of a tall thin matrix: > 64 back-trace segments of 4 rows
    (3, [(324, 1), (300, 4), (420, 2)]),
])
def test_paths_bit_exact_wide_and_lopsided(radius, sizes):
    rng = np.random.RandomState(100 + radius)
    pairs = [(_tracks(rng, tx, 3), _tracks(rng, ty, 3)) for tx, ty in sizes]
    pi, pj, pl, cost = _run_pairs(pairs, radius)
    for n, (x, y) in enumerate(pairs):
        d, path = OD.fastdtw(x, y, radius)
        assert pl[n] == len(path), (n, sizes[n])
        assert np.array_equal(pi[n, :pl[n]], path[:, 0]) and np.array_equal(pj[n, :pl[n]], path[:, 1]), (n, sizes[n])
        assert abs(cost[n] - d) <= 1e-12 * max(d, 1e-300)


def test_paths_bit_exact_on_tie_heavy_data():
    """Quantised tracks: many exactly equal candidate costs; the kernel must break ties like the oracle
    (compare after the add, first minimum of up / left / diagonal)."""
    rng = np.random.RandomState(77)
    pairs = []
    for tx, ty in [(60, 60), (130, 97), (200, 230), (64, 64)]:
        pairs.append((np.round(_tracks(rng, tx, 2) * 4) / 4, np.round(_tracks(rng, ty, 2) * 4) / 4))
    pairs.append((np.zeros((50, 2)) + 1.0, np.zeros((70, 2)) + 1.0))     # all costs zero
    for radius in (1, 2):
        pi, pj, pl, cost = _run_pairs(pairs, radius)
        for n, (x, y) in enumerate(pairs):
            d, path = OD.fastdtw(x, y, radius)
            assert pl[n] == len(path), (radius, n)
            assert np.array_equal(pi[n, :pl[n]], path[:, 0]) and np.array_equal(pj[n, :pl[n]], path[:, 1]), (radius, n)
            assert cost[n] == d


def test_baseline_config4_shape():
    """BASELINE configs[3] shape at reduced batch: T in [700, 900], 25-dim, radius 1."""
    rng = np.random.RandomState(1234)
    pairs = []
    for _ in range(24):
        tx, ty = rng.randint(700, 901, size=2)
        pairs.append((_tracks(rng, tx, 25), _tracks(rng, ty, 25)))
    pi, pj, pl, cost = _run_pairs(pairs, 1)
    for n, (x, y) in enumerate(pairs):
        d, path = OD.fastdtw(x, y, 1)
        assert pl[n] == len(path)
        assert np.array_equal(pi[n, :pl[n]], path[:, 0]) and np.array_equal(pj[n, :pl[n]], path[:, 1])
        assert abs(cost[n] - d) <= 1e-12 * d
        # path validity properties (size independent)
        steps_i, steps_j = np.diff(pi[n, :pl[n]]), np.diff(pj[n, :pl[n]])
        assert pi[n, 0] == 0 and pj[n, 0] == 0 and pi[n, pl[n] - 1] == len(x) - 1 and pj[n, pl[n] - 1] == len(y) - 1
        assert ((steps_i >= 0) & (steps_i <= 1) & (steps_j >= 0) & (steps_j <= 1) & (steps_i + steps_j >= 1)).all()


def test_aligner_matches_oracle_transform():
    from nnmnkwii_amd.preprocessing.alignment import DTWAligner
    rng = np.random.RandomState(3)
    N, D = 6, 5
    for (Tx, Ty, dt) in ((60, 75, np.float64), (75, 60, np.float32), (50, 50, np.float64)):
        X = np.zeros((N, Tx, D), dtype=dt)
        Y = np.zeros((N, Ty, D), dtype=dt)
        for n in range(N):
            a, b = rng.randint(Tx // 2, Tx + 1), rng.randint(Ty // 2, Ty + 1)
            X[n, :a] = _tracks(rng, a, D)
            Y[n, :b] = _tracks(rng, b, D)
        Xa, Ya = DTWAligner().transform((X, Y))
        Xo, Yo, paths, dists = OD.dtw_align(X, Y)
        assert Xa.shape == Ya.shape == Xo.shape
        assert Xa.dtype == Xo.dtype
        assert np.array_equal(Xa, Xo) and np.array_equal(Ya, Yo)


def test_reference_assertions_on_shifted_copy():
    # reference tests/test_preprocessing.py:441-457: zero-shifted copy triggers the frame
    # length adjustment; only shapes are asserted (ties are everywhere in this input)
    from nnmnkwii_amd.preprocessing.alignment import DTWAligner
    rng = np.random.RandomState(0)
    X = np.zeros((3, 40, 5))
    for n, t in enumerate((40, 33, 21)):
        X[n, :t] = rng.rand(t, 5) + 0.1
    Y = np.pad(X, [(0, 0), (5, 0), (0, 0)], mode="constant", constant_values=0)[:, :-5, :]
    Xa, Ya = DTWAligner().transform((X, Y))
    assert Xa.shape == Ya.shape
    # and a time-stretched copy aligns closer than the raw pair (:488-489)
    t = np.linspace(0, 1, 200)[:, None]
    x = np.sin(2 * np.pi * (1 + np.arange(4)) * t)
    y = np.sin(2 * np.pi * (1 + np.arange(4)) * t ** 1.5)
    Xa, Ya = DTWAligner().transform((x[None], y[None]))
    assert Xa.shape == Ya.shape
    assert np.linalg.norm(Xa - Ya) < np.linalg.norm(x - y)


def test_trim_zeros_frames():
    from nnmnkwii_amd.preprocessing import trim_zeros_frames
    rng = np.random.RandomState(0)
    for dt in (np.float32, np.float64):
        x = rng.rand(100, 10).astype(dt)
        assert trim_zeros_frames(x) is x
        x[70:] = 0
        x[:5] = 0
        assert np.array_equal(trim_zeros_frames(x), OD.trim_zeros_frames(x))
        assert trim_zeros_frames(x).shape == (70, 10)
        assert trim_zeros_frames(x, trim="f").shape == (95, 10)
        assert trim_zeros_frames(x, trim="fb").shape == (65, 10)
        x[60] = 1e-9   # below eps -> still "zero" but interior
        assert trim_zeros_frames(x).shape == (70, 10)
    assert trim_zeros_frames(np.zeros((8, 3))).shape == (0, 3)


def test_against_the_real_fastdtw_package_when_installed():
    """The DTW oracle restates slaypni/fastdtw from its published algorithm (parity unpinned: the package is not in
    this image).  Wherever it IS installed this test pins oracle and kernel against it: random, tie-heavy
    (quantised) and constant inputs, radius 1 and 2, the aligner's default distance."""
    fastdtw_mod = pytest.importorskip("fastdtw")
    import torch
    from numpy.linalg import norm
    from nnmnkwii_amd import _hip
    from oracle import dtw as OD
    rng = np.random.RandomState(12)
    cases = []
    for k in range(6):
        a, b = rng.randint(20, 120, size=2)
        x, y = np.cumsum(rng.randn(a, 5), 0), np.cumsum(rng.randn(b, 5), 0)
        if k % 3 == 1:
            x, y = np.round(x), np.round(y)                  # ties
        if k % 3 == 2:
            x[:] = 1.0
            y[:] = 1.0                                       # every path costs the same
        cases.append((x, y))
    # which tie rule the installed build follows (pure Python: first minimum; the compiled extension is recalled to
    # use a strict-less chain): one rule must reproduce EVERY case
    rules = [t for t in (0, 1) if all(
        [tuple(p) for p in fastdtw_mod.fastdtw(x, y, radius=r, dist=lambda u, v: norm(u - v))[1]]
        == [tuple(p) for p in OD.fastdtw(x, y, r, tie=t)[1]] for r in (1, 2) for x, y in cases)]
    assert rules, "neither tie rule reproduces the installed fastdtw"
    tie = rules[0]
    for radius in (1, 2):
        for x, y in cases:
            d_ref, p_ref = fastdtw_mod.fastdtw(x, y, radius=radius, dist=lambda u, v: norm(u - v))
            d_o, p_o = OD.fastdtw(x, y, radius, tie=tie)
            assert [tuple(p) for p in p_ref] == [tuple(p) for p in p_o]
            assert abs(d_ref - d_o) <= 1e-12 * max(1.0, abs(d_ref))
            X = torch.from_numpy(x[None].copy()).cuda()
            Y = torch.from_numpy(y[None].copy()).cuda()
            lx = torch.tensor([len(x)], dtype=torch.int32, device="cuda")
            ly = torch.tensor([len(y)], dtype=torch.int32, device="cuda")
            pi, pj, pl, cost = _hip.fastdtw_l2(X, Y, lx, ly, radius, tie_rule=tie)
            n = int(pl[0])
            assert list(zip(pi[0, :n].tolist(), pj[0, :n].tolist())) == [tuple(p) for p in p_ref]


@pytest.mark.parametrize("radius", [1, 3])   # radius 3: rows of ~26 cells, no pair fits the optimistic back-pointer capacity
def test_many_pairs_take_the_two_launch_form_and_wide_windows_are_retried(radius):
    """More than 512 pairs: the first launch runs 256 threads per pair with optimistic LDS capacities (four pairs per
    CU), pairs whose windows do not fit are marked and run again by a second launch with the proven bounds.  The
    batch mixes ordinary pairs with pairs whose warping path has long horizontal and vertical runs (rows hundreds of
    cells wide at every level): every path equals the oracle's, whichever launch produced it."""
    rng = np.random.RandomState(77)
    N, D = 640, 3
    pairs = []
    for n in range(N):
        if n % 40 == 7:
            # a step function against a ramp: the path runs along the axes
            tx, ty = int(rng.randint(300, 420)), int(rng.randint(300, 420))
            x = np.zeros((tx, D)); x[tx // 2:] = 5.0
            x += 1e-3 * rng.randn(tx, D)
            y = np.linspace(0.0, 5.0, ty)[:, None] * np.ones((1, D)) + 1e-3 * rng.randn(ty, D)
        else:
            tx, ty = int(rng.randint(40, 160)), int(rng.randint(40, 160))
            x, y = _tracks(rng, tx, D), _tracks(rng, ty, D)
        pairs.append((x, y))
    pi, pj, pl, cost = _run_pairs(pairs, radius)
    assert (pl > 0).all()
    for n in list(range(0, N, 11)) + [n for n in range(N) if n % 40 == 7]:
        x, y = pairs[n]
        d, path = OD.fastdtw(x, y, radius)
        assert pl[n] == len(path), n
        assert np.array_equal(pi[n, :pl[n]], path[:, 0]) and np.array_equal(pj[n, :pl[n]], path[:, 1]), n
        assert abs(cost[n] - d) <= 1e-12 * max(d, 1e-300), n


def _quantised_pairs(seed=77):
    rng = np.random.RandomState(seed)
    pairs = []
    for tx, ty in [(60, 60), (130, 97), (200, 230), (64, 64), (33, 90)]:
        pairs.append((np.round(_tracks(rng, tx, 2) * 4) / 4, np.round(_tracks(rng, ty, 2) * 4) / 4))
    pairs.append((np.zeros((50, 2)) + 1.0, np.zeros((70, 2)) + 1.0))     # all costs zero
    x = np.round(_tracks(rng, 120, 3) * 2) / 2
    pairs.append((x, np.concatenate([x[:1].repeat(7, 0), x])[:120]))     # shifted copy
    return pairs


@pytest.mark.parametrize("radius", [1, 2])
def test_second_tie_rule_against_the_oracle(radius):
    """tie_rule = MLPG_HIP_TIE_DIAG_LAST (the strict-less chain recalled for upstream's compiled extension) on inputs
    full of exactly equal candidates: the kernel's paths equal the oracle's under the SAME rule, for both rules, and the
    two rules really differ on these inputs."""
    from nnmnkwii_amd import _hip
    pairs = _quantised_pairs()
    N = len(pairs)
    Tx = max(len(x) for x, _ in pairs)
    Ty = max(len(y) for _, y in pairs)
    X = np.zeros((N, Tx, 3))
    Y = np.zeros((N, Ty, 3))
    for n, (x, y) in enumerate(pairs):
        X[n, :len(x), :x.shape[1]] = x
        Y[n, :len(y), :y.shape[1]] = y
    lenx = torch.tensor([len(x) for x, _ in pairs], dtype=torch.int32, device="cuda")
    leny = torch.tensor([len(y) for _, y in pairs], dtype=torch.int32, device="cuda")
    got = {}
    for tie in (_hip.TIE_FIRST_MIN, _hip.TIE_DIAG_LAST):
        pi, pj, pl, cost = _hip.fastdtw_l2(torch.from_numpy(X).cuda(), torch.from_numpy(Y).cuda(), lenx, leny, radius,
                                           tie_rule=tie)
        pi, pj, pl, cost = pi.cpu().numpy(), pj.cpu().numpy(), pl.cpu().numpy(), cost.cpu().numpy()
        got[tie] = []
        for n in range(N):
            xs, ys = X[n, :len(pairs[n][0])], Y[n, :len(pairs[n][1])]
            d, path = OD.fastdtw(xs, ys, radius, tie=tie)
            assert pl[n] == len(path), (tie, n)
            assert np.array_equal(pi[n, :pl[n]], path[:, 0]) and np.array_equal(pj[n, :pl[n]], path[:, 1]), (tie, n)
            assert cost[n] == d
            got[tie].append(list(zip(pi[n, :pl[n]].tolist(), pj[n, :pl[n]].tolist())))
    assert sum(a != b for a, b in zip(got[0], got[1])) >= 3


def test_second_tie_rule_through_the_host_entry_point_and_the_aligner():
    from nnmnkwii_amd import _hip
    from nnmnkwii_amd.preprocessing.alignment import DTWAligner
    pairs = _quantised_pairs(5)
    N = len(pairs)
    Tx = max(max(len(x), len(y)) for x, y in pairs)
    X = np.zeros((N, Tx, 3))
    Y = np.zeros((N, Tx, 3))
    for n, (x, y) in enumerate(pairs):
        X[n, :len(x), :x.shape[1]] = x + 3.0        # no all-zero frames inside
        Y[n, :len(y), :y.shape[1]] = y + 3.0
    for tie, name in ((_hip.TIE_FIRST_MIN, "first"), (_hip.TIE_DIAG_LAST, "diag_last")):
        pi, pj, pl, cost, lenx, leny = _hip.fastdtw_host(X, Y, 1, tie_rule=tie)
        Xa, Ya = DTWAligner(tie_rule=name).transform((X, Y))
        for n in range(N):
            xs, ys = X[n, :lenx[n]], Y[n, :leny[n]]
            d, path = OD.fastdtw(xs, ys, 1, tie=tie)
            assert pl[n] == len(path)
            assert np.array_equal(pi[n, :pl[n]], path[:, 0]) and np.array_equal(pj[n, :pl[n]], path[:, 1]), (tie, n)
            assert np.array_equal(Xa[n, :len(path)], xs[path[:, 0]]) and np.array_equal(Ya[n, :len(path)], ys[path[:, 1]])
            assert not Xa[n, len(path):].any()


def _cosine(u, v):
    return 1.0 - float(u @ v) / (float(np.sqrt(u @ u)) * float(np.sqrt(v @ v)) + 1e-12)


@pytest.mark.parametrize("radius", [1, 2, 4])
def test_arbitrary_python_dist_runs_with_host_costs_and_gives_the_literal_restatements_paths(radius):
    """A cosine distance (nothing the kernel's cost table knows): the aligner evaluates it on the host, cell by cell, and
    the GPU does DP + back-trace + window expansion from the cost buffer.  Paths and costs equal the pure-Python
    restatement of fastdtw called with the SAME callable; ragged lengths, pairs whose recursion depths differ, pairs
    below the recursion threshold (one full-window level)."""
    from nnmnkwii_amd import _hip
    from nnmnkwii_amd.preprocessing.alignment import DTWAligner
    rng = np.random.RandomState(21 + radius)
    sizes = [(90, 70), (33, 64), (5, 4), (2, 9), (1, 1), (140, 141), (17, 3)]
    xs = [np.abs(_tracks(rng, a, 6)) + 0.1 for a, _ in sizes]
    ys = [np.abs(_tracks(rng, b, 6)) + 0.1 for _, b in sizes]
    for tie in (_hip.TIE_FIRST_MIN, _hip.TIE_DIAG_LAST):
        pi, pj, pl, cost = _hip.fastdtw_callable(xs, ys, radius, _cosine, tie)
        for n in range(len(xs)):
            d, path = OD.fastdtw_py(xs[n], ys[n], radius, _cosine, tie=tie)
            path = np.asarray(path)
            assert pl[n] == len(path), (radius, tie, n)
            assert np.array_equal(pi[n, :pl[n]], path[:, 0]) and np.array_equal(pj[n, :pl[n]], path[:, 1]), (radius, tie, n)
            assert abs(cost[n] - d) <= 1e-12 * max(1.0, abs(d))
    # through the aligner (zero padded batch, float32 input)
    N = len(xs)
    Tm = 150
    X = np.zeros((N, Tm, 6), dtype=np.float32)
    Y = np.zeros((N, Tm - 5, 6), dtype=np.float32)
    for n in range(N):
        X[n, :len(xs[n])] = xs[n]
        Y[n, :len(ys[n])] = ys[n]
    Xa, Ya = DTWAligner(dist=_cosine, radius=radius).transform((X, Y))
    assert Xa.dtype == np.float32 and Xa.shape == Ya.shape and Xa.shape[1] >= Tm
    for n in range(N):
        x, y = X[n, :len(xs[n])], Y[n, :len(ys[n])]
        _, path = OD.fastdtw_py(x.astype(np.float64), y.astype(np.float64), radius, _cosine)
        path = np.asarray(path)
        assert np.array_equal(Xa[n, :len(path)], x[path[:, 0]]) and np.array_equal(Ya[n, :len(path)], y[path[:, 1]]), n
        assert not Xa[n, len(path):].any() and not Ya[n, len(path):].any()


def test_host_cost_route_equals_the_in_kernel_route_for_the_euclidean_distance():
    """The same pairs through both routes (costs in the kernel / costs from the host callable): identical paths."""
    from numpy.linalg import norm
    from nnmnkwii_amd import _hip
    rng = np.random.RandomState(8)
    pairs = [(_tracks(rng, a, 4), _tracks(rng, b, 4)) for a, b in [(200, 180), (64, 300), (31, 31)]]
    pi, pj, pl, cost = _run_pairs(pairs, 1)
    qi, qj, ql, qcost = _hip.fastdtw_callable([p[0] for p in pairs], [p[1] for p in pairs], 1, lambda u, v: norm(u - v))
    for n in range(len(pairs)):
        assert pl[n] == ql[n]
        assert np.array_equal(pi[n, :pl[n]], qi[n, :pl[n]]) and np.array_equal(pj[n, :pl[n]], qj[n, :pl[n]])
        assert abs(cost[n] - qcost[n]) <= 1e-12 * cost[n]


def test_kernel_against_textbook_dtw_and_its_own_path_cost():
    """Independent of the fastdtw restatement (the package is absent: parity unpinned): with a radius that covers the
    whole matrix the kernel's distance must be the optimum of the textbook DTW recurrence; at radius 1 on config-4 sized
    pairs the reported distance must be the local costs accumulated along the reported path, never below the optimum."""
    rng = np.random.RandomState(91)

    def textbook(x, y):
        d = np.sqrt(((x[:, None, :] - y[None, :, :]) ** 2).sum(-1))
        D = np.full((len(x) + 1, len(y) + 1), np.inf)
        D[0, 0] = 0.0
        for i in range(1, len(x) + 1):
            for j in range(1, len(y) + 1):
                D[i, j] = d[i - 1, j - 1] + min(D[i - 1, j], D[i, j - 1], D[i - 1, j - 1])
        return float(D[-1, -1]), d

    small = [(_tracks(rng, tx, 4), _tracks(rng, ty, 4)) for tx, ty in ((30, 41), (64, 50), (17, 17), (70, 23))]
    pi, pj, pl, cost = _run_pairs(small, radius=80)
    for n, (x, y) in enumerate(small):
        opt, d = textbook(x, y)
        assert abs(cost[n] - opt) <= 1e-12 * opt, (n, cost[n], opt)
        k = int(pl[n])
        assert abs(float(d[pi[n, :k], pj[n, :k]].sum()) - opt) <= 1e-12 * opt
    big = [(_tracks(rng, int(rng.randint(700, 901)), 25), _tracks(rng, int(rng.randint(700, 901)), 25)) for _ in range(6)]
    pi, pj, pl, cost = _run_pairs(big, radius=1)
    for n, (x, y) in enumerate(big):
        k = int(pl[n])
        i, j = pi[n, :k], pj[n, :k]
        assert (i[0], j[0]) == (0, 0) and (i[-1], j[-1]) == (len(x) - 1, len(y) - 1)
        si, sj = np.diff(i), np.diff(j)
        assert ((si >= 0) & (si <= 1) & (sj >= 0) & (sj <= 1) & (si + sj >= 1)).all()
        acc = float(np.sqrt(((x[i] - y[j]) ** 2).sum(-1)).sum())
        assert abs(acc - cost[n]) <= 1e-11 * acc, (n, acc, cost[n])
    opt, _ = textbook(big[0][0][:200], big[0][1][:180])
    _, _, _, c1 = _run_pairs([(big[0][0][:200], big[0][1][:180])], radius=1)
    assert c1[0] >= opt * (1 - 1e-12)
