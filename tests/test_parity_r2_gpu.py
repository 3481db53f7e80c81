"""Round-2 parity tests (-m gpu): the holes the round-1 review listed.

* backward (mlpg_grad) of every kernel -- generic, wave-per-system (M = 8 / 16 / 32), strip -- against the
  REFERENCE's own O(T^2) mlpg_grad at T = 300, 1000, 2000 (tests/golden/mlpg_golden2.npz, made by
  tests/golden/make_golden2.py from the reference itself), float32 and float64;
* float32 / global-variance / unit-variance forward of the wave and generic kernels against the ORACLE (not against
  each other) at the same T;
* BASELINE config 3 at its size (64 x 500 x 180 float32): unit_variance_mlpg forward and backward against the dense
  R @ means of the reference's definition evaluated in float64 on the host (R = the oracle's restatement of
  unit_variance_mlpg_matrix, itself checked against rows produced by the reference);
* BASELINE config 4 sized DTW pairs through DTWAligner against the reference's unmodified alignment.py run with its
  own dist = norm(x - y) (align_golden2.npz).
"""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from cases import WINDOW_SETS, c4_pairs, rand_case  # noqa: E402
from oracle import mlpg as O  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def golden2():
    return np.load(os.path.join(HERE, "golden", "mlpg_golden2.npz"))


def _algos(T):
    from nnmnkwii_amd import _hip
    a = [_hip.ALGO_GENERIC, _hip.ALGO_STRIP, _hip.ALGO_AUTO]
    if T <= 2048:
        a.append(_hip.ALGO_WAVE)
    return a


@pytest.mark.parametrize("key", ["std3-f32-T300", "std3-f64-T300", "std3-f32-T1000", "std3-f64-T1000",
                                 "std3-f32-T2000", "std3-f64-T2000", "std2-f64-T700", "asym2-f64-T700"])
def test_backward_vs_reference_mlpg_grad(golden2, key):
    import torch
    from nnmnkwii_amd import _hip
    wname, dt, T = key.split("-")
    T = int(T[1:])
    windows = WINDOW_SETS[wname]
    sd = 2 if wname == "std3" else 3
    m, v, _ = rand_case(wname, dt, T, sd, salt=11)
    go = np.random.RandomState(500 + T).randn(T, sd).astype(m.dtype)
    ref = golden2["grad2/%s/g" % key]                      # float32, as the reference returns it (_mlpg.py:248)
    scale = np.abs(ref).max()
    vt, gt = torch.from_numpy(v[None]).cuda(), torch.from_numpy(go[None]).cuda()
    for algo in _algos(T):
        g, st = _hip.backward(vt, gt, windows, v.shape[1], out_dtype=torch.float32, algo=algo)
        assert int(st.abs().max()) == 0
        g = g[0].cpu().numpy()
        assert g.dtype == np.float32 and g.shape == ref.shape
        assert np.abs(g - ref).max() <= 3e-6 * scale, (key, algo, np.abs(g - ref).max() / scale)
    # float64 output of the float64 case against the float64 oracle restatement as well
    if dt == "f64":
        g64, _ = _hip.backward(vt, gt, windows, v.shape[1], out_dtype=torch.float64, algo=_hip.ALGO_AUTO)
        assert np.abs(g64[0].cpu().numpy() - ref).max() <= 3e-6 * scale


@pytest.mark.parametrize("T", [200, 500, 1000, 2000])
def test_forward_f32_global_unit_vs_oracle(T):
    """Every kernel, float32 inputs with per-frame / global / unit variances and float64 with global / unit ones,
    ragged lengths: against the oracle (which is bit-exact against the reference)."""
    import torch
    from nnmnkwii_amd import _hip
    windows = WINDOW_SETS["std3"]
    B, sd = 3, 20
    rng = np.random.RandomState(T)
    M_ = rng.randn(B, T, 3 * sd)
    V_ = rng.rand(B, T, 3 * sd) + 0.1
    lengths = np.array([T, T - 1, T // 2], dtype=np.int32)
    L = torch.from_numpy(lengths).cuda()
    for dt, tol in ((np.float32, 5e-6), (np.float64, 1e-9)):
        Md = M_.astype(dt)
        for mode in ("frame", "global", "unit"):
            if mode == "frame":
                Vd = V_.astype(dt)
            elif mode == "global":
                Vd = V_[0, 0].astype(dt)
            else:
                Vd = None
            yo, _, rc = O.mlpg_batch(Md, Vd if Vd is not None else np.ones(3 * sd, dtype=dt), windows, lengths)
            assert rc == 0
            for algo in _algos(T):
                y, st = _hip.forward(torch.from_numpy(Md).cuda(), None if Vd is None else torch.from_numpy(Vd).cuda(),
                                     windows, L, algo=algo)
                assert int(st.abs().max()) == 0
                y = y.cpu().numpy()
                sc = np.abs(yo).max(axis=1, keepdims=True) + 1e-300
                assert (np.abs(y - yo) / sc).max() <= tol, (T, dt.__name__, mode, algo)


def test_config3_size_unit_variance_autograd(golden2):
    """BASELINE config 3: B = 64, T = 500, D = 180 float32 tensors on the GPU, forward + backward through
    autograd.unit_variance_mlpg, against the reference's definition (dense R @ means, R^T @ grad) in float64."""
    import torch
    from nnmnkwii_amd import autograd as AF
    from nnmnkwii_amd import paramgen as G
    windows = WINDOW_SETS["std3"]
    B, T, D = 64, 500, 180
    sd = D // 3
    Ro = O.unit_variance_mlpg_matrix(windows, T).astype(np.float64)          # (T, 3T) restatement of _mlpg.py:297-373
    assert np.abs(Ro[[0, 1, 2, 250, 497, 498, 499]] - golden2["uv3/rows"]).max() <= 1e-6   # the reference's rows (float32)
    assert np.abs(Ro.sum(axis=1) - golden2["uv3/rowsum"]).max() <= 1e-5
    R = torch.from_numpy(G.unit_variance_mlpg_matrix(windows, T)).cuda()
    assert np.abs(R.cpu().numpy() - Ro).max() <= 2e-7
    torch.manual_seed(1234)
    means = torch.rand(B, T, D, device="cuda", requires_grad=True)
    target = torch.rand(B, T, sd, device="cuda")
    y = AF.unit_variance_mlpg(R, means)
    loss = torch.nn.MSELoss()(y, target)
    loss.backward()
    m64 = means.detach().cpu().numpy().astype(np.float64)
    rm = m64.reshape(B, T, 3, sd).transpose(0, 2, 1, 3).reshape(B, 3 * T, sd)   # reshape_means per utterance
    y_ref = np.einsum("tk,bkd->btd", Ro, rm)
    assert y.dtype == torch.float32 and tuple(y.shape) == (B, T, sd)
    assert np.abs(y.detach().cpu().numpy() - y_ref).max() <= 5e-6
    gy = 2.0 * (y_ref - target.cpu().numpy().astype(np.float64)) / (B * T * sd)
    gr = np.einsum("tk,btd->bkd", Ro, gy).reshape(B, 3, T, sd).transpose(0, 2, 1, 3).reshape(B, T, D)
    assert np.abs(means.grad.cpu().numpy() - gr).max() <= 1e-6 * np.abs(gr).max() + 1e-12


def test_config4_size_dtw_aligner_vs_reference_alignment():
    from nnmnkwii_amd.preprocessing.alignment import DTWAligner
    g = np.load(os.path.join(HERE, "golden", "align_golden2.npz"))
    X, Y = c4_pairs(6)
    Xa, Ya = DTWAligner().transform((X, Y))
    assert Xa.shape == g["dtw4/Xa"].shape and Ya.shape == g["dtw4/Ya"].shape
    np.testing.assert_array_equal(Xa, g["dtw4/Xa"])
    np.testing.assert_array_equal(Ya, g["dtw4/Ya"])


@pytest.mark.parametrize("name", ["small", "grow", "c4"])
def test_dtw_aligner_custom_dist_melcd_vs_reference(name):
    """DTWAligner(dist=melcd): the reference's own custom-dist case (tests/test_preprocessing.py:496-501), goldens from
    the reference's unmodified alignment.py + metrics.melcd."""
    from cases import align_batch
    from nnmnkwii_amd.metrics import melcd
    from nnmnkwii_amd.preprocessing.alignment import DTWAligner
    g = np.load(os.path.join(HERE, "golden", "align_golden2.npz"))
    X, Y = c4_pairs(2, seed=99) if name == "c4" else align_batch(name)
    Xa, Ya = DTWAligner(dist=melcd).transform((X, Y))
    np.testing.assert_array_equal(Xa, g["dtw-melcd/%s/Xa" % name])
    np.testing.assert_array_equal(Ya, g["dtw-melcd/%s/Ya" % name])


def test_dtw_aligner_dist_resolution():
    from numpy.linalg import norm
    from nnmnkwii_amd import _hip
    from nnmnkwii_amd.metrics import melcd
    from nnmnkwii_amd.preprocessing.alignment import DTWAligner, _resolve_dist
    assert _resolve_dist(lambda x, y: norm(x - y)) == (_hip.DIST_L2, 1.0)          # the reference's default, passed explicitly
    assert _resolve_dist(lambda x, y: np.sqrt(((x - y) ** 2).sum())) == (_hip.DIST_L2, 1.0)
    k, c = _resolve_dist(melcd)
    assert k == _hip.DIST_SCALED_L2_NP and abs(c - 10.0 / np.log(10.0) * np.sqrt(2.0)) < 1e-15
    k, c = _resolve_dist(lambda x, y: 3.0 * norm(x - y))
    assert k == _hip.DIST_SCALED_L2_NP and abs(c - 3.0) < 1e-12
    k, c = _resolve_dist(lambda x, y: np.abs(x - y).sum())
    assert k == _hip.DIST_SCALED_L1_NP and abs(c - 1.0) < 1e-12
    k, c = _resolve_dist(lambda x, y: 0.5 * ((x - y) ** 2).sum())
    assert k == _hip.DIST_SCALED_SQL2_NP and abs(c - 0.5) < 1e-12
    assert _resolve_dist(lambda x, y: np.abs(x - y).max()) is None      # evaluated on the host, cell by cell
    X, Y = c4_pairs(1, seed=5)
    a = DTWAligner(dist=lambda x, y: norm(x - y)).transform((X, Y))
    b = DTWAligner().transform((X, Y))
    np.testing.assert_array_equal(a[0], b[0])


def test_host_entry_point_numpy_in_numpy_out():
    """mlpg_hip_forward_host (chunked, overlapped staging; no torch on the path): several chunks, ragged lengths,
    every variance mode, float32 and float64, pageable and pinned host arrays -- against the oracle."""
    from nnmnkwii_amd import _hip
    from nnmnkwii_amd import paramgen as G
    windows = WINDOW_SETS["std3"]
    rng = np.random.RandomState(3)
    B, T, sd = 37, 300, 60           # 37 utterances: chunks of 10 (the last one short)
    M_ = rng.randn(B, T, 3 * sd)
    V_ = rng.rand(B, T, 3 * sd) + 0.1
    lengths = rng.randint(1, T + 1, size=B).astype(np.int32)
    lengths[5] = T
    yo, _, rc = O.mlpg_batch(M_, V_, windows, lengths)
    assert rc == 0
    y, st = _hip.forward_host(M_, V_, windows, lengths)
    assert not st.any() and y.dtype == np.float64
    sc = np.abs(yo).max(axis=1, keepdims=True) + 1e-300
    assert (np.abs(y - yo) / sc).max() <= 1e-9
    # pinned arrays are transferred in place
    Mp, Vp = _hip.pinned_empty(M_.shape), _hip.pinned_empty(V_.shape)
    Mp[...] = M_
    Vp[...] = V_
    y2, _ = _hip.forward_host(Mp, Vp, windows, lengths)
    assert np.array_equal(y2, y)
    # the drop-in batch call takes this path for numpy inputs; global / unit variances; float32
    y3 = G.mlpg_batch(M_, V_, windows, lengths)
    assert np.array_equal(y3, y)
    vg = V_[0, 0].copy()
    yg = G.mlpg_batch(M_, vg, windows, lengths)
    ygo, _, _ = O.mlpg_batch(M_, vg, windows, lengths)
    assert (np.abs(yg - ygo) / (np.abs(ygo).max(axis=1, keepdims=True) + 1e-300)).max() <= 1e-9
    M32 = M_.astype(np.float32)
    yu = G.mlpg_batch(M32, None, windows, lengths)
    yuo, _, _ = O.mlpg_batch(M32, np.ones(3 * sd, dtype=np.float32), windows, lengths)
    assert yu.dtype == np.float32
    assert (np.abs(yu - yuo) / (np.abs(yuo).max(axis=1, keepdims=True) + 1e-300)).max() <= 5e-6
    # a non-positive-definite system is reported like the reference does
    Vb = V_.copy()
    Vb[20, 100, 7] = -1e-3
    with pytest.raises(np.linalg.LinAlgError):
        G.mlpg_batch(M_, Vb, windows, lengths=None)


def test_fastdtw_host_entry_equals_device_entry():
    """mlpg_hip_fastdtw_host (chunks of pairs on two internal streams, device-side trim, pinned staging) returns
    exactly what the device entry point returns on resident tensors: several chunks with a short last one,
    float64 and float32 inputs, given and device-trimmed lengths, the melcd distance, pinned input arrays."""
    import torch
    from nnmnkwii_amd import _hip
    X, Y = c4_pairs(11, seed=9)
    Xd, Yd = torch.from_numpy(X).cuda(), torch.from_numpy(Y).cuda()
    lx, ly = _hip.trim_lengths(Xd), _hip.trim_lengths(Yd)
    for kind, scale in ((_hip.DIST_L2, 1.0), (_hip.DIST_SCALED_L2_NP, 6.14185)):
        pi, pj, pl, cost = _hip.fastdtw_l2(Xd, Yd, lx, ly, 1, kind, scale)
        hi, hj, hl, hc, hx, hy = _hip.fastdtw_host(X, Y, 1, kind, scale)
        assert np.array_equal(hx, lx.cpu().numpy()) and np.array_equal(hy, ly.cpu().numpy())
        assert np.array_equal(hl, pl.cpu().numpy()) and np.array_equal(hc, cost.cpu().numpy())
        for n in range(len(X)):
            assert np.array_equal(hi[n, :hl[n]], pi[n, :hl[n]].cpu().numpy())
            assert np.array_equal(hj[n, :hl[n]], pj[n, :hl[n]].cpu().numpy())
    # explicit lengths (shorter than the trim would give), radius 2
    lx2 = (lx.cpu().numpy() - 50).astype(np.int32)
    ly2 = (ly.cpu().numpy() - 20).astype(np.int32)
    pi, pj, pl, cost = _hip.fastdtw_l2(Xd, Yd, torch.from_numpy(lx2).cuda(), torch.from_numpy(ly2).cuda(), 2)
    hi, hj, hl, hc, hx, hy = _hip.fastdtw_host(X, Y, 2, lenx=lx2, leny=ly2)
    assert np.array_equal(hl, pl.cpu().numpy()) and np.array_equal(hc, cost.cpu().numpy()) and np.array_equal(hx, lx2)
    # float32 arrays are widened on the device; pinned arrays are transferred in place
    X32, Y32 = X.astype(np.float32), Y.astype(np.float32)
    pi, pj, pl, cost = _hip.fastdtw_l2(torch.from_numpy(X32).cuda().double(), torch.from_numpy(Y32).cuda().double(), lx, ly, 1)
    Xp, Yp = _hip.pinned_empty(X32.shape, np.float32), _hip.pinned_empty(Y32.shape, np.float32)
    Xp[...] = X32
    Yp[...] = Y32
    for a, b in ((X32, Y32), (Xp, Yp)):
        hi, hj, hl, hc, _, _ = _hip.fastdtw_host(a, b, 1)
        assert np.array_equal(hl, pl.cpu().numpy()) and np.array_equal(hc, cost.cpu().numpy())
        assert np.array_equal(hi[3, :hl[3]], pi[3, :hl[3]].cpu().numpy())


def test_dtw_aligner_host_entry_route_equals_device_route(monkeypatch):
    """DTWAligner takes the host-pointer entry point for large batches and device tensors for small ones: the two
    routes return the same arrays (forced here on one small batch), float64 and float32."""
    from nnmnkwii_amd.preprocessing.alignment import DTWAligner, IterativeDTWAligner
    X, Y = c4_pairs(5, seed=21)
    for dt in (np.float64, np.float32):
        Xc, Yc = X.astype(dt), Y.astype(dt)
        a = DTWAligner().transform((Xc, Yc))
        monkeypatch.setattr(DTWAligner, "_HOST_ENTRY_BYTES", 0)
        b = DTWAligner().transform((Xc, Yc))
        monkeypatch.undo()
        assert a[0].dtype == b[0].dtype == dt and a[0].shape == b[0].shape
        np.testing.assert_array_equal(a[0], b[0])
        np.testing.assert_array_equal(a[1], b[1])
    np.random.seed(0)
    a = IterativeDTWAligner(n_iter=1, n_components_gmm=2, max_iter_gmm=5).transform((X[:3, :200], Y[:3, :200]))
    monkeypatch.setattr(DTWAligner, "_HOST_ENTRY_BYTES", 0)
    np.random.seed(0)
    b = IterativeDTWAligner(n_iter=1, n_components_gmm=2, max_iter_gmm=5).transform((X[:3, :200], Y[:3, :200]))
    np.testing.assert_array_equal(a[0], b[0])
    np.testing.assert_array_equal(a[1], b[1])


def test_shutdown_releases_everything_and_the_library_keeps_working():
    """mlpg_hip_shutdown frees the per-stream scratch, the side streams of the multi-stream entry and the staging
    buffers / streams of the host entry points; every path must come up again afterwards with the same results."""
    import torch
    from nnmnkwii_amd import _hip
    from nnmnkwii_amd import paramgen as G
    windows = WINDOW_SETS["std3"]
    rng = np.random.RandomState(8)
    M_ = rng.randn(9, 200, 180)
    V_ = rng.rand(9, 200, 180) + 0.1

    def run():
        y_host = G.mlpg_batch(M_, V_, windows)                                        # host entry
        m, v = torch.from_numpy(M_).cuda(), torch.from_numpy(V_).cuda()
        y_strip, _ = _hip.forward(m, v, windows, algo=_hip.ALGO_STRIP)
        y_gen, _ = _hip.forward(m, v, windows, algo=_hip.ALGO_GENERIC)
        ys = G.multi_stream_mlpg(M_[:, :, :129], V_[0, 0, :129], windows, [120, 3, 6], [True, True, True])
        X, Y = c4_pairs(3, seed=2)
        paths = _hip.fastdtw_host(X, Y, 1)
        torch.cuda.synchronize()
        return y_host, y_strip.cpu().numpy(), y_gen.cpu().numpy(), ys, paths[0], paths[3]
    a = run()
    _hip.lib().mlpg_hip_shutdown()
    b = run()
    _hip.lib().mlpg_hip_shutdown()
    _hip.lib().mlpg_hip_shutdown()           # idempotent
    c = run()
    for u, v, w in zip(a, b, c):
        assert np.array_equal(u, v) and np.array_equal(u, w)


def test_config4_64_pairs_paths_vs_reference_with_numpy_norm():
    """64 config-4 pairs: the HIP fastdtw paths against the paths the reference's DTWAligner took with its own
    ``norm(x - y)`` dist (tests/golden/dtw_paths64.npz, make_golden3.py), index for index, and the aligned arrays the
    aligner returns against ``X[path_i]`` / ``Y[path_j]`` built from those paths."""
    import torch
    from nnmnkwii_amd import _hip
    from nnmnkwii_amd.preprocessing.alignment import DTWAligner
    g = np.load(os.path.join(HERE, "golden", "dtw_paths64.npz"))
    X, Y = c4_pairs(64, seed=64)
    Xd, Yd = torch.from_numpy(X).cuda(), torch.from_numpy(Y).cuda()
    lenx, leny = _hip.trim_lengths(Xd), _hip.trim_lengths(Yd)
    assert np.array_equal(lenx.cpu().numpy(), g["lenx"]) and np.array_equal(leny.cpu().numpy(), g["leny"])
    pi, pj, pl, cost = _hip.fastdtw_l2(Xd, Yd, lenx, leny, 1)
    pi, pj, pl, cost = pi.cpu().numpy(), pj.cpu().numpy(), pl.cpu().numpy(), cost.cpu().numpy()
    assert np.array_equal(pl, g["plen"])
    for n in range(64):
        k = int(pl[n])
        assert np.array_equal(pi[n, :k], g["paths"][n, :k, 0]) and np.array_equal(pj[n, :k], g["paths"][n, :k, 1]), n
    assert np.abs(cost - g["dist"]).max() <= 1e-12 * g["dist"].max()
    Xa, Ya = DTWAligner().transform((X, Y))
    assert Xa.shape[1] == int(g["T_out"][0])
    for n in (0, 17, 63):
        k = int(pl[n])
        assert np.array_equal(Xa[n, :k], X[n][g["paths"][n, :k, 0]]) and np.array_equal(Ya[n, :k], Y[n][g["paths"][n, :k, 1]])
        assert not Xa[n, k:].any() and not Ya[n, k:].any()


@pytest.mark.parametrize("form", ["l1", "sql2"])
def test_dtw_aligner_cityblock_and_squared_euclidean_callables(form):
    """dist callables of the city-block and squared-Euclidean forms run on the device (MLPG_HIP_DIST_SCALED_L1_NP /
    _SQL2_NP, numpy's summation order): paths and aligned arrays against the literal restatement of fastdtw driven by
    the very same Python callable (oracle/dtw.py::fastdtw_py; small pairs, it calls the callable per DP cell)."""
    from cases import align_batch
    from nnmnkwii_amd.preprocessing.alignment import DTWAligner
    from oracle import dtw as OD
    dist = (lambda x, y: np.abs(x - y).sum()) if form == "l1" else (lambda x, y: ((x - y) ** 2).sum())
    for X, Y in (align_batch("grow"), c4_pairs(2, seed=7)):     # 4-dim (sequential sum) and 25-dim (pairwise sum) frames
        Xa, Ya = DTWAligner(dist=dist).transform((X, Y))
        for n in range(X.shape[0]):
            x, y = OD.trim_zeros_frames(X[n]), OD.trim_zeros_frames(Y[n])
            _, path = OD.fastdtw_py(x, y, radius=1, dist=dist)
            path = np.asarray(path)
            k = len(path)
            assert np.array_equal(Xa[n, :k], x[path[:, 0]]) and np.array_equal(Ya[n, :k], y[path[:, 1]]), (form, n)
            assert not Xa[n, k:].any() and not Ya[n, k:].any()
