"""Pin the CPU oracle against golden vectors produced by the reference itself.

(-m "not gpu".)  MLPG: oracle/mlpg_oracle.c must reproduce the reference
bit-for-bit on every golden case (it restates the same summation order with FP
contraction off).  DTW: parity is unpinned (no reference binary exists); the C
and literal-Python restatements must agree with each other exactly.
"""
import numpy as np
import pytest

from cases import WINDOW_SETS, c2_utterance, rand_case
from oracle import dtw as OD
from oracle import mlpg as O


def _mlpg_cases(golden):
    for k in golden.files:
        if k.startswith("mlpg/") and k.endswith("/y") and "-T" in k:
            wname, dt, T, sd = k.split("/")[1].split("-")
            yield k, wname, dt, int(T[1:]), int(sd[2:])


def test_kat_stages(golden):
    b, P, chol, bad = O.mlpg_stages(golden["kat/means"], golden["kat/var"], WINDOW_SETS["std3"])
    assert bad == 0
    assert np.array_equal(b, golden["kat/b"])
    assert np.array_equal(P, golden["kat/P"])
    assert np.array_equal(chol, golden["kat/chol"])
    # SURVEY.md 8(c) printed values
    assert np.allclose(b, [0.05, 0.15, 1.6, -0.25, 1.35, 0.5], atol=1e-15)
    y = O.mlpg(golden["kat/means"], golden["kat/var"], WINDOW_SETS["std3"])
    assert np.array_equal(y, golden["kat/y"])
    assert np.allclose(y.ravel(), [0.056395946541, 0.214084300191, 0.435981788809,
                                   0.245836393009, 0.431370245264, 0.316331326186], atol=1e-11)


def test_mlpg_bit_exact_vs_reference(golden):
    n = 0
    for k, wname, dt, T, sd in _mlpg_cases(golden):
        m, v, vg = rand_case(wname, dt, T, sd)
        y = O.mlpg(m, v, WINDOW_SETS[wname])
        yg = O.mlpg(m, vg, WINDOW_SETS[wname])
        assert y.dtype == golden[k].dtype == m.dtype
        assert np.array_equal(y, golden[k]), k
        assert np.array_equal(yg, golden[k[:-1] + "yg"]), k
        n += 1
    assert n == 168


def test_mlpg_baseline_configs(golden):
    m, v, vg = rand_case("std3", "f64", 100, 2)
    assert np.array_equal(O.mlpg(m, v, WINDOW_SETS["std3"]), golden["mlpg/c1/y"])
    assert np.array_equal(O.mlpg(m, vg, WINDOW_SETS["std3"]), golden["mlpg/c1/yg"])
    for b in range(2):
        m, v = c2_utterance(b)
        assert np.array_equal(O.mlpg(m, v, WINDOW_SETS["std3"]), golden["mlpg/c2-utt%d/y" % b])
    m, v = c2_utterance(0)
    y32 = O.mlpg(m.astype(np.float32), v.astype(np.float32), WINDOW_SETS["std3"])
    assert y32.dtype == np.float32
    assert np.array_equal(y32, golden["mlpg/c2-utt0-f32/y"])


def test_mlpg_batch_lengths():
    ms, vs = zip(*(c2_utterance(b, T=40, sd=3) for b in range(3)))
    lengths = np.array([40, 17, 1], dtype=np.int32)
    M = np.stack(ms)
    V = np.stack(vs)
    out, status, rc = O.mlpg_batch(M, V, WINDOW_SETS["std3"], lengths)
    assert rc == 0 and not status.any()
    for b, T in enumerate(lengths):
        assert np.array_equal(out[b, :T], O.mlpg(M[b, :T], V[b, :T], WINDOW_SETS["std3"]))
        assert not out[b, T:].any()


def test_edge_cases():
    # T=1, T=2 with the std windows: every dynamic precision is zeroed -> y == static means
    for T in (1, 2):
        m, v, _ = rand_case("std3", "f64", T, 2)
        assert np.allclose(O.mlpg(m, v, WINDOW_SETS["std3"]), m[:, :2], rtol=1e-14, atol=0)
    # all-zero-extent dynamic windows: the [-0:] slice zeroes the whole column
    m, v, _ = rand_case("zero2", "f64", 9, 2)
    assert np.allclose(O.mlpg(m, v, WINDOW_SETS["zero2"]), m[:, :2], rtol=1e-15)


def test_error_message(golden):
    m, _, _ = rand_case("std3", "f64", 10, 1)
    with pytest.raises(np.linalg.LinAlgError) as ei:
        O.mlpg(m, golden["err/negvar-v"], WINDOW_SETS["std3"])
    assert str(ei.value) == str(golden["err/negvar-msg"]) == "5-th leading minor not positive definite"


def test_grad_and_matrix(golden):
    for k in golden.files:
        if k.startswith("grad/") and k.endswith("/g"):
            wname, T, sd = k.split("/")[1].split("-")
            T, sd = int(T[1:]), int(sd[2:])
            m, v, _ = rand_case(wname, "f32", T, sd, salt=7)
            go = np.random.RandomState(99 + T + sd).randn(T, sd).astype(np.float32)
            g = O.mlpg_grad(m, v, WINDOW_SETS[wname], go)
            assert g.dtype == np.float32
            scale = np.abs(golden[k]).max() + 1e-30
            assert np.abs(g - golden[k]).max() <= 1e-6 * scale, k
        if k.startswith("uvmat/"):
            wname, T = k.split("/")[1].split("-")
            R = O.unit_variance_mlpg_matrix(WINDOW_SETS[wname], int(T[1:]))
            assert R.dtype == np.float32 and R.shape == golden[k].shape
            assert np.abs(R - golden[k]).max() <= 1e-7, k


def test_reshape_means():
    m = np.random.RandomState(0).rand(5, 6)
    r = O.reshape_means(m, 2)
    assert r.shape == (15, 2)
    assert O.reshape_means(r, 2) is r
    assert np.array_equal(r[5:10], m[:, 2:4])


def test_fastdtw_c_matches_literal_python():
    rng = np.random.RandomState(0)
    for (tx, ty, D, r) in [(5, 7, 3, 1), (20, 31, 4, 1), (64, 50, 2, 2), (3, 3, 1, 1), (2, 9, 2, 1),
                           (33, 17, 5, 1), (100, 120, 25, 1), (57, 57, 1, 3), (1, 1, 2, 1), (1, 6, 2, 1)]:
        x = np.cumsum(rng.randn(tx, D), 0) * 0.1
        y = np.cumsum(rng.randn(ty, D), 0) * 0.1
        d1, p1 = OD.fastdtw_py(x, y, r)
        d2, p2 = OD.fastdtw(x, y, r)
        assert d1 == d2
        assert np.array_equal(np.asarray(p1, dtype=np.int32), p2)
        assert p2[0].tolist() == [0, 0] and p2[-1].tolist() == [tx - 1, ty - 1]
        steps = np.diff(p2, axis=0)
        assert ((steps >= 0) & (steps <= 1)).all() and (steps.sum(1) >= 1).all()


def test_dtw_align_reference_properties():
    # the reference's own assertions (tests/test_preprocessing.py:441-457): shapes equal
    rng = np.random.RandomState(1)
    X = np.cumsum(rng.randn(4, 60, 5), 1) * 0.1
    Y = np.zeros((4, 75, 5))
    for i in range(4):
        n = 50 + 5 * i
        Y[i, :n] = np.cumsum(rng.randn(n, 5), 0) * 0.1
    Xa, Ya, paths, dists = OD.dtw_align(X, Y)
    assert Xa.shape == Ya.shape and Xa.shape[0] == 4 and Xa.shape[1] >= 75
    Xb, Yb, _, _ = OD.dtw_align(X, Y, use_c=False)
    assert np.array_equal(Xa, Xb) and np.array_equal(Ya, Yb)


def test_delta_features_oracle_vs_reference(golden):
    n = 0
    for k in golden.files:
        if k.startswith("delta/") and k.endswith("/y") and k != "delta/plain/y":
            wname = k.split("/")[1].split("-")[0]
            y = O.delta_features(golden[k[:-1] + "x"], WINDOW_SETS[wname])
            assert y.dtype == golden[k].dtype and np.array_equal(y, golden[k]), k
            n += 1
    assert n == 36
    y = O.delta_features(golden["delta/plain/x"], [np.array([1.0]), np.array([0.25, 0.5, -1.0, 2.0]),
                                                   np.array([-0.5, 0.0, 0.5])])
    assert np.array_equal(y, golden["delta/plain/y"])


def test_oracle_vs_compiled_reference():
    """Where oracle/_ref (the reference's own modules, compiled by oracle/build_ref_so.sh) is
    present, the C restatement must match it bit for bit on fresh random inputs."""
    from oracle import ref
    if not ref.available():
        pytest.skip("oracle/_ref not built")
    G = ref.load()
    rng = np.random.RandomState(2024)
    for wname in ("std3", "wide3", "std2"):
        windows = WINDOW_SETS[wname]
        nw = len(windows)
        for T in (7, 64, 301):
            m = rng.randn(T, nw * 3)
            v = rng.rand(T, nw * 3) + 0.05
            assert np.array_equal(O.mlpg(m, v, windows), G.mlpg(m, v, windows))
            m32, v32 = m.astype(np.float32), v.astype(np.float32)
            assert np.array_equal(O.mlpg(m32, v32, windows), G.mlpg(m32, v32, windows))
        R = G.unit_variance_mlpg_matrix(windows, 12)
        assert np.abs(R - O.unit_variance_mlpg_matrix(windows, 12)).max() <= 1e-7


def test_modspec_oracle_matches_reference_goldens():
    """oracle/modspec.py (numpy's FFT, as the reference) against every modulation-spectrum golden the reference
    produced (make_golden.py): spectrum, phase, inverse, smoothing in both domains and norms, analytic gradient."""
    import os
    from oracle import modspec as OM
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mlpg_golden.npz"))
    seen = 0
    for key in [k for k in g.files if k.startswith("modspec/T") and k.endswith("/x")]:
        case = key[:-2]
        n = int(case.split("-n")[1])
        x = g[key]
        for norm in (None, "ortho"):
            pre = "%s/%s/" % (case, norm or "none")
            ms, ph = OM.modspec(x, n=n, norm=norm, return_phase=True)
            np.testing.assert_allclose(ms, g[pre + "ms"], rtol=1e-12, atol=1e-12 * np.abs(g[pre + "ms"]).max())
            np.testing.assert_allclose(ph, g[pre + "phase"], rtol=0, atol=1e-9)
            np.testing.assert_allclose(OM.inv_modspec(g[pre + "ms"], g[pre + "phase"], norm=norm), g[pre + "inv"], rtol=0, atol=1e-12)
            for k in [k for k in g.files if k.startswith(pre + "smooth-")]:
                log_domain = bool(int(k.split("smooth-log")[1][0]))
                cutoff = int(k.split("-c")[1])
                y = OM.modspec_smoothing(x, 200, n=n, norm=norm, cutoff=cutoff, log_domain=log_domain)
                np.testing.assert_allclose(y, g[k], rtol=0, atol=1e-11)
                seen += 1
    for key in [k for k in g.files if k.startswith("modspec_grad/") and k.endswith("/y")]:
        case = key[:-2]
        n = int(case.split("-n")[1].split("-")[0])
        norm = None if case.endswith("-none") else "ortho"
        grad = OM.modspec_grad(g[key], g[case + "/w"], n, norm)
        np.testing.assert_allclose(grad, g[case + "/grad"], rtol=0, atol=2e-5 * np.abs(g[case + "/grad"]).max())   # float32 goldens
        seen += 1
    assert seen > 30


def test_dtw_oracle_matches_64_reference_pairs_with_numpy_norm():
    """Round 3: 64 BASELINE config-4 sized pairs (T in [700, 900], 25-dim) through the reference's unmodified
    DTWAligner with its own ``dist = lambda x, y: norm(x - y)`` (numpy BLAS order), paths recorded by
    tests/golden/make_golden3.py.  The C oracle sums the local cost sequentially; on these ~110 000 path cells (about a
    million window cells) the two orders never lead the DP to a different decision, and the distances agree to rounding.
    (fastdtw itself is bound to the literal restatement there: parity unpinned, as oracle/dtw.py says.)"""
    import os
    from cases import c4_pairs
    from oracle import dtw as OD
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dtw_paths64.npz"))
    X, Y = c4_pairs(64, seed=64)
    assert len(g["plen"]) == 64
    for n in range(64):
        x, y = OD.trim_zeros_frames(X[n]), OD.trim_zeros_frames(Y[n])
        assert (len(x), len(y)) == (int(g["lenx"][n]), int(g["leny"][n]))
        d, path = OD.fastdtw(x, y, 1)
        k = int(g["plen"][n])
        assert len(path) == k and np.array_equal(path, g["paths"][n, :k].astype(np.int32)), n
        assert abs(d - g["dist"][n]) <= 1e-12 * g["dist"][n]


def _textbook_dtw_cost(x, y):
    """Dynamic time warping from its definition (Sakoe & Chiba 1978: D[i,j] = d(i,j) + min(D[i-1,j], D[i,j-1], D[i-1,j-1]),
    Euclidean local cost, full matrix) -- written here, independent of oracle/dtw.py and of fastdtw's source."""
    tx, ty = len(x), len(y)
    d = np.sqrt(((x[:, None, :] - y[None, :, :]) ** 2).sum(-1))
    D = np.full((tx + 1, ty + 1), np.inf)
    D[0, 0] = 0.0
    for i in range(1, tx + 1):
        for j in range(1, ty + 1):
            D[i, j] = d[i - 1, j - 1] + min(D[i - 1, j], D[i, j - 1], D[i - 1, j - 1])
    return float(D[tx, ty]), d


def test_fastdtw_restatement_against_textbook_dtw():
    """What can be pinned without the fastdtw package: (1) with a radius that covers the whole matrix fastdtw IS exact DTW --
    its distance must equal the textbook recurrence's optimum; (2) for any radius its path is a valid warping path whose
    accumulated local cost is the distance it reports, and that distance is never below the optimum."""
    rng = np.random.RandomState(77)
    for (tx, ty, D) in [(9, 13, 2), (40, 31, 5), (64, 64, 3), (25, 70, 4)]:
        x = np.cumsum(rng.randn(tx, D), 0) * 0.1
        y = np.cumsum(rng.randn(ty, D), 0) * 0.1
        opt, d = _textbook_dtw_cost(x, y)
        for radius in (1, 2, 3, max(tx, ty)):
            for tie in (0, 1):
                dist, path = OD.fastdtw(x, y, radius, tie=tie)
                path = np.asarray(path)
                assert path[0].tolist() == [0, 0] and path[-1].tolist() == [tx - 1, ty - 1]
                steps = np.diff(path, axis=0)
                assert ((steps >= 0) & (steps <= 1)).all() and (steps.sum(1) >= 1).all()
                acc = float(d[path[:, 0], path[:, 1]].sum())
                assert abs(acc - dist) <= 1e-12 * max(1.0, abs(dist)), (tx, ty, radius, tie)
                assert dist >= opt - 1e-12 * opt
                if radius >= max(tx, ty):
                    assert abs(dist - opt) <= 1e-12 * opt, (tx, ty, tie, dist, opt)
        d1, p1 = OD.fastdtw_py(x, y, max(tx, ty))
        assert abs(d1 - opt) <= 1e-12 * opt
