"""GPU parity tests (-m gpu) of the strip MLPG kernel (algo = MLPG_HIP_ALGO_STRIP: lane per static dim,
wavefront per 16-frame chunk, strips of one utterance solved across workgroups), through the C ABI,
against the CPU oracle.  Same tolerances as tests/test_mlpg_gpu.py."""
import numpy as np
import pytest

from cases import WINDOW_SETS, c2_utterance
from oracle import mlpg as O

pytestmark = pytest.mark.gpu

TOL64 = 1e-9
TOL32 = 5e-6
STRIP_WINDOWS = ("std3", "std2", "asym2", "static", "zero2")   # extents <= 1


def rel_err(y, ref):
    scale = np.abs(ref).max(axis=0, keepdims=True)
    scale = np.where(scale == 0, 1.0, scale)
    return float((np.abs(y.astype(np.float64) - ref.astype(np.float64)) / scale).max())


@pytest.mark.parametrize("T", [1, 2, 3, 15, 16, 17, 33, 63, 64, 65, 66, 127, 128, 129, 200, 257, 513, 1000, 1025, 2048, 4100])
def test_strip_all_lengths(T):
    """Every strip count from 1 to 65 (several level-3 batches), ragged lengths, 70 static dims in two dim groups."""
    import torch
    from nnmnkwii_amd import _hip
    for wname in ("std3", "asym2") if T > 600 else STRIP_WINDOWS:
        windows = WINDOW_SETS[wname]
        nw = len(windows)
        B, sd = 3, 70 if T <= 300 else 20
        rng = np.random.RandomState(T + nw)
        M_ = rng.randn(B, T, nw * sd)
        V_ = rng.rand(B, T, nw * sd) + 0.1
        lengths = np.array([T, max(1, T - 1), max(1, T // 2)], dtype=np.int32)
        m, v, L = torch.from_numpy(M_).cuda(), torch.from_numpy(V_).cuda(), torch.from_numpy(lengths).cuda()
        ys, sts = _hip.forward(m, v, windows, L, algo=_hip.ALGO_STRIP)
        assert int(sts.abs().max()) == 0
        yo, _, rc = O.mlpg_batch(M_, V_, windows, lengths)
        assert rc == 0
        ys = ys.cpu().numpy()
        assert rel_err(ys.reshape(-1, sd), yo.reshape(-1, sd)) <= TOL64, (wname, T)
        for b in range(B):
            assert not ys[b, lengths[b]:].any()
        # backward: strip == generic (and == oracle in test_strip_backward_vs_reference_goldens)
        go = torch.from_numpy(rng.randn(B, T, sd)).cuda()
        gs, _ = _hip.backward(v, go, windows, nw * sd, L, out_dtype=torch.float64, algo=_hip.ALGO_STRIP)
        gg, _ = _hip.backward(v, go, windows, nw * sd, L, out_dtype=torch.float64, algo=_hip.ALGO_GENERIC)
        scale = float(gg.abs().max()) + 1e-300
        assert float((gs - gg).abs().max()) <= 1e-10 * scale, (wname, T)
        # float32 inputs; global and unit variances vs the oracle
        M32 = M_.astype(np.float32)
        m32 = torch.from_numpy(M32).cuda()
        vg = V_[0, 0].astype(np.float32)
        a, _ = _hip.forward(m32, torch.from_numpy(vg).cuda(), windows, L, algo=_hip.ALGO_STRIP)
        ao, _, _ = O.mlpg_batch(M32, vg, windows, lengths)
        assert rel_err(a.cpu().numpy().reshape(-1, sd), ao.reshape(-1, sd)) <= TOL32, (wname, T)
        a, _ = _hip.forward(m32, None, windows, L, algo=_hip.ALGO_STRIP)
        ao, _, _ = O.mlpg_batch(M32, np.ones(nw * sd, dtype=np.float32), windows, lengths)
        assert rel_err(a.cpu().numpy().reshape(-1, sd), ao.reshape(-1, sd)) <= TOL32, (wname, T)


@pytest.mark.parametrize("sd", [1, 5, 16, 25, 60, 63, 64, 65, 128, 130])
def test_strip_static_dims(sd):
    import torch
    from nnmnkwii_amd import _hip
    windows = WINDOW_SETS["std3"]
    rng = np.random.RandomState(sd)
    B, T = 2, 150
    M_ = rng.randn(B, T, 3 * sd)
    V_ = rng.rand(B, T, 3 * sd) + 0.1
    y, st = _hip.forward(torch.from_numpy(M_).cuda(), torch.from_numpy(V_).cuda(), windows, algo=_hip.ALGO_STRIP)
    yo, _, rc = O.mlpg_batch(M_, V_, windows)
    assert rc == 0 and int(st.abs().max()) == 0
    assert rel_err(y.cpu().numpy().reshape(-1, sd), yo.reshape(-1, sd)) <= TOL64


def test_strip_config2_utterances(golden):
    """BASELINE config-2 utterances against the reference's own outputs (goldens), whole batch via AUTO."""
    import torch
    from nnmnkwii_amd import _hip
    ms, vs = zip(*(c2_utterance(b) for b in range(2)))
    m, v = torch.from_numpy(np.stack(ms)).cuda(), torch.from_numpy(np.stack(vs)).cuda()
    for algo in (_hip.ALGO_STRIP, _hip.ALGO_AUTO):
        y, st = _hip.forward(m, v, WINDOW_SETS["std3"], algo=algo)
        assert int(st.abs().max()) == 0
        for b in range(2):
            assert rel_err(y[b].cpu().numpy(), golden["mlpg/c2-utt%d/y" % b]) <= TOL64
    y32, _ = _hip.forward(m[:1].float().contiguous(), v[:1].float().contiguous(), WINDOW_SETS["std3"], algo=_hip.ALGO_STRIP)
    assert rel_err(y32[0].cpu().numpy(), golden["mlpg/c2-utt0-f32/y"]) <= 2e-6


def test_strip_not_pd_status():
    """The strip kernel reports the reference's first failing natural-order pivot and zero-fills the system."""
    import torch
    from nnmnkwii_amd import _hip
    windows = WINDOW_SETS["std3"]
    rng = np.random.RandomState(0)
    B, T, sd = 2, 300, 20
    M_ = rng.randn(B, T, 3 * sd)
    V_ = rng.rand(B, T, 3 * sd) + 0.1
    V_[1, 137, 4] = -1e-3       # static variance of dim 4 -> pivot 138 fails (third strip)
    V_[0, 10, sd + 2] = -1e-4   # delta variance of dim 2 (first strip)
    V_[0, 290, 7] = -1e-4       # last strip
    exp = np.zeros((B, sd), dtype=np.int32)
    for b, d in ((1, 4), (0, 2), (0, 7)):
        cols = [d, sd + d, 2 * sd + d]
        _, s1, _ = O.mlpg_batch(M_[b:b + 1][:, :, cols], V_[b:b + 1][:, :, cols], windows)
        exp[b, d] = s1[0, 0]
    assert exp[1, 4] == 138 and exp[0, 2] > 0 and exp[0, 7] > 0
    y, st = _hip.forward(torch.from_numpy(M_).cuda(), torch.from_numpy(V_).cuda(), windows, algo=_hip.ALGO_STRIP)
    assert np.array_equal(st.cpu().numpy().reshape(B, sd), exp)
    y = y.cpu().numpy()
    assert not y[1, :, 4].any() and not y[0, :, 2].any() and not y[0, :, 7].any()
    ok = np.ones((B, sd), dtype=bool)
    ok[1, 4] = ok[0, 2] = ok[0, 7] = False
    V_ok = np.abs(V_)
    yo, _, _ = O.mlpg_batch(M_, V_ok, windows)
    for b in range(B):
        for d in range(sd):
            if ok[b, d]:
                assert np.abs(y[b, :, d] - yo[b, :, d]).max() <= TOL64 * np.abs(yo[b, :, d]).max()


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_strip_not_pd_far_from_strip0(dtype):
    """Level 3 is windowed: a strip never sees a failing pivot 3+ strips away.  The verdict (status = the reference's
    first failing natural-order pivot, all-zero column) must not depend on where the pivot fails: first, middle and
    last strips of long utterances, two failures in one system, ragged lengths, forward and backward."""
    import torch
    from nnmnkwii_amd import _hip
    STD3 = WINDOW_SETS["std3"]
    npdt = np.float64 if dtype == "f64" else np.float32
    rng = np.random.RandomState(3)
    B, T, sd = 4, 1500, 6
    m = rng.randn(B, T, 3 * sd).astype(npdt)
    v = (rng.rand(B, T, 3 * sd) + 0.1).astype(npdt)
    lengths = np.array([T, T - 100, T, 700], dtype=np.int32)
    v[0, 1333, 1] = -1e-3             # strip 20 of 24
    v[0, 5, 4] = -1e-3                # strip 0
    v[1, 700, 0] = -1e-3              # middle
    v[1, 1390, 2] = -1e-3             # last live strip of a ragged utterance
    v[2, 640, 3] = -1e-3; v[2, 1200, 3] = -1e-3   # two failures in one system: the first one is reported
    v[2, 1450, sd + 5] = -1e-4        # a delta variance near the end
    v[3, 690, 5] = -1e-3              # just before the end of a short utterance
    v[3, 900, 2] = -1e-3              # in the padding: not a failure
    exp = np.zeros((B, sd), dtype=np.int32)
    for b in range(B):
        for d in range(sd):
            cols = [d, sd + d, 2 * sd + d]
            _, s1, _ = O.mlpg_batch(m[b:b + 1][:, :, cols], v[b:b + 1][:, :, cols], STD3, lengths[b:b + 1])
            exp[b, d] = s1[0, 0]
    assert exp[0, 1] == 1334 and exp[0, 4] == 6 and exp[2, 3] == 641 and exp[3, 2] == 0 and exp[2, 5] > 0
    mg, vg, L = torch.from_numpy(m).cuda(), torch.from_numpy(v).cuda(), torch.from_numpy(lengths).cuda()
    y, st = _hip.forward(mg, vg, STD3, L, algo=_hip.ALGO_STRIP)
    yg, stg = _hip.forward(mg, vg, STD3, L, algo=_hip.ALGO_GENERIC)
    assert np.array_equal(st.cpu().numpy().reshape(B, sd), exp)
    assert np.array_equal(stg.cpu().numpy().reshape(B, sd), exp)
    y, yg = y.cpu().numpy(), yg.cpu().numpy()
    for b in range(B):
        for d in range(sd):
            if exp[b, d]:
                assert not y[b, :, d].any(), (b, d)
            else:
                assert np.abs(y[b, :, d] - yg[b, :, d]).max() <= (1e-9 if dtype == "f64" else 2e-4) * np.abs(yg[b, :, d]).max(), (b, d)
    go = torch.from_numpy(rng.randn(B, T, sd).astype(npdt)).cuda()
    gs, sts = _hip.backward(vg, go, STD3, 3 * sd, L, out_dtype=torch.float64, algo=_hip.ALGO_STRIP)
    gg, _ = _hip.backward(vg, go, STD3, 3 * sd, L, out_dtype=torch.float64, algo=_hip.ALGO_GENERIC)
    assert np.array_equal(sts.cpu().numpy().reshape(B, sd), exp)
    gs, gg = gs.cpu().numpy().reshape(B, T, 3, sd), gg.cpu().numpy().reshape(B, T, 3, sd)
    for b in range(B):
        for d in range(sd):
            if exp[b, d]:
                assert not gs[b, :, :, d].any(), (b, d)
            else:
                assert np.abs(gs[b, :, :, d] - gg[b, :, :, d]).max() <= 1e-6 * np.abs(gg[b, :, :, d]).max(), (b, d)


@pytest.mark.parametrize("sigma,tol", [(2.0, 1e-9), (4.0, 1e-7), (6.0, 1e-5)])
def test_strip_ill_conditioned(sigma, tol):
    import torch
    from nnmnkwii_amd import _hip
    STD3 = WINDOW_SETS["std3"]
    rng = np.random.RandomState(int(sigma * 10))
    B, T, sd = 3, 1000, 16
    m = rng.randn(B, T, 3 * sd)
    v = np.exp(sigma * rng.randn(B, T, 3 * sd))
    ref, st, rc = O.mlpg_batch(m, v, STD3)
    assert rc == 0
    out, status = _hip.forward(torch.from_numpy(m).cuda(), torch.from_numpy(v).cuda(), STD3, algo=_hip.ALGO_STRIP)
    assert int(status.abs().max().item()) == 0
    err = np.abs(out.cpu().numpy() - ref).max(axis=1) / np.abs(ref).max(axis=1)
    print("sigma", sigma, "strip max rel err", err.max())
    assert err.max() <= tol


@pytest.mark.parametrize("T", [1000, 2500])
def test_strip_long_range_coupling_falls_back_to_full_sweep(T):
    """Level 3 first tries the records of strips r-2 .. r+2 and accepts them when its damping bound is far below
    rounding.  Static variances of 1e14 over most of the utterance leave only the dynamic features to tie the
    trajectory down there, strips far apart stay coupled, the bound says so and those strips sweep the whole
    utterance: the result still equals the oracle's / the generic kernel's.  (Mixed batch: utterance 1 is
    ordinary and takes the window everywhere.)"""
    import torch
    from nnmnkwii_amd import _hip
    STD3 = WINDOW_SETS["std3"]
    rng = np.random.RandomState(T)
    B, sd = 3, 20
    m = rng.randn(B, T, 3 * sd)
    v = rng.rand(B, T, 3 * sd) + 0.1
    v[0, 200:T - 200, :sd] = 1e14
    v[2, 100:T - 300, :sd:2] = 1e12       # every other system only: lanes of one strip disagree about the window
    lengths = np.array([T, T - 7, T], dtype=np.int32)
    ref, _, rc = O.mlpg_batch(m, v, STD3, lengths)
    assert rc == 0
    mg, vg, L = torch.from_numpy(m).cuda(), torch.from_numpy(v).cuda(), torch.from_numpy(lengths).cuda()
    out, status = _hip.forward(mg, vg, STD3, L, algo=_hip.ALGO_STRIP)
    gen, _ = _hip.forward(mg, vg, STD3, L, algo=_hip.ALGO_GENERIC)
    assert int(status.abs().max().item()) == 0
    out, gen = out.cpu().numpy(), gen.cpu().numpy()
    scale = np.abs(ref).max(axis=1, keepdims=True)
    e_ref = (np.abs(out - ref) / scale).max()
    e_gen = (np.abs(gen - ref) / scale).max()
    print("T", T, "strip vs oracle", e_ref, "generic vs oracle", e_gen)
    assert e_ref <= max(1e-9, 10 * e_gen)
    # backward through the same operators
    go = torch.from_numpy(rng.randn(B, T, sd)).cuda()
    gs, _ = _hip.backward(vg, go, STD3, 3 * sd, L, out_dtype=torch.float64, algo=_hip.ALGO_STRIP)
    gg, _ = _hip.backward(vg, go, STD3, 3 * sd, L, out_dtype=torch.float64, algo=_hip.ALGO_GENERIC)
    assert float((gs - gg).abs().max()) <= 1e-7 * float(gg.abs().max())


def test_strip_full_size_and_repeatability():
    """Config-2 size: linearity, batch-permutation equivariance (bitwise), repeated launches bitwise equal
    (the inter-workgroup protocol leaves no timing dependence in the numbers), spot parity vs the oracle."""
    import torch
    from nnmnkwii_amd import _hip
    windows = WINDOW_SETS["std3"]
    g = torch.Generator(device="cuda").manual_seed(7)
    B, T, D = 256, 1000, 180
    m1 = torch.randn(B, T, D, dtype=torch.float64, device="cuda", generator=g)
    m2 = torch.randn(B, T, D, dtype=torch.float64, device="cuda", generator=g)
    v = torch.rand(B, T, D, dtype=torch.float64, device="cuda", generator=g) + 0.1
    f = lambda m: _hip.forward(m, v, windows, algo=_hip.ALGO_STRIP)
    y1, s1 = f(m1)
    assert int(s1.abs().max()) == 0
    for _ in range(5):
        ya, _ = f(m1)
        assert torch.equal(ya, y1)
    y2, _ = f(m2)
    y3, _ = f(2.5 * m1 + m2)
    scale = float(y3.abs().max())
    assert float((y3 - (2.5 * y1 + y2)).abs().max()) <= 1e-11 * scale
    perm = torch.randperm(B, device="cuda", generator=g)
    yp, _ = _hip.forward(m1[perm].contiguous(), v[perm].contiguous(), windows, algo=_hip.ALGO_STRIP)
    assert torch.equal(yp, y1[perm])
    for b in (0, 100, 255):
        yo = O.mlpg(m1[b].cpu().numpy(), v[b].cpu().numpy(), windows)
        assert rel_err(y1[b].cpu().numpy(), yo) <= TOL64
    # forward/backward adjointness at full size
    go = torch.randn(B, T, D // 3, dtype=torch.float64, device="cuda", generator=g)
    gm, _ = _hip.backward(v, go, windows, D, out_dtype=torch.float64, algo=_hip.ALGO_STRIP)
    lhs, rhs = float((go * y1).sum()), float((gm * m1).sum())
    assert abs(lhs - rhs) <= 1e-9 * max(abs(lhs), 1.0)


def test_strip_two_streams_concurrently():
    """Launches on two HIP streams of one device use separate scratch (records, arrival counters)."""
    import torch
    from nnmnkwii_amd import _hip
    windows = WINDOW_SETS["std3"]
    g = torch.Generator(device="cuda").manual_seed(11)
    B, T, D = 64, 700, 90
    ms = [torch.randn(B, T, D, dtype=torch.float64, device="cuda", generator=g) for _ in range(2)]
    vs = [torch.rand(B, T, D, dtype=torch.float64, device="cuda", generator=g) + 0.1 for _ in range(2)]
    ref = [_hip.forward(ms[k], vs[k], windows, algo=_hip.ALGO_STRIP)[0] for k in range(2)]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(2)]
    outs = [None, None]
    for rep in range(8):
        for k in range(2):
            with torch.cuda.stream(streams[k]):
                outs[k] = _hip.forward(ms[k], vs[k], windows, algo=_hip.ALGO_STRIP)[0]
    torch.cuda.synchronize()
    for k in range(2):
        assert torch.equal(outs[k], ref[k])


@pytest.mark.parametrize("T", [1100, 3000])
def test_strip_tight_dynamic_variances_every_window_rejected(T):
    """Variances as acoustic models have them (delta / delta-delta 100 x / 1000 x tighter than static): the coupling
    between strips decays by only ~1e-2 per strip, a 5-strip window would be rejected by the damping bound, and the
    strips route themselves to the full-utterance sweep from their own transfer factor; with 10 x / 100 x they take
    the 9-strip window.  Same numbers as the generic kernel and the oracle; mixed in one batch with an utterance of
    ordinary variances (5-strip window); bitwise repeatable."""
    import torch
    from nnmnkwii_amd import _hip
    STD3 = WINDOW_SETS["std3"]
    rng = np.random.RandomState(77)
    B, sd = 6, 60                    # T = 3000 (47 strips): the tight utterances take windows of 8 / 16 strips per side
    m = rng.randn(B, T, 3 * sd)
    v = rng.rand(B, T, 3 * sd) + 0.1
    v[:3, :, sd:2 * sd] *= 1e-2
    v[:3, :, 2 * sd:] *= 1e-3
    v[3:5, :, sd:2 * sd] *= 1e-1          # moderately tight: the 9-strip window (route 2)
    v[3:5, :, 2 * sd:] *= 1e-2
    lengths = np.array([T, T - 3, 700, T, 65, T], dtype=np.int32)
    mg, vg, L = torch.from_numpy(m).cuda(), torch.from_numpy(v).cuda(), torch.from_numpy(lengths).cuda()
    out, status = _hip.forward(mg, vg, STD3, L, algo=_hip.ALGO_STRIP)
    gen, _ = _hip.forward(mg, vg, STD3, L, algo=_hip.ALGO_GENERIC)
    assert int(status.abs().max().item()) == 0
    out, gen = out.cpu().numpy(), gen.cpu().numpy()
    # oracle on a subset of the systems (3 of the 60 dims)
    cols = [0, 31, 59]
    sel = [c + w * sd for w in range(3) for c in cols]
    ref, _, rc = O.mlpg_batch(m[:, :, sel], v[:, :, sel], STD3, lengths)
    assert rc == 0
    scale = np.abs(ref).max(axis=1, keepdims=True)
    e_gen = (np.abs(gen[:, :, cols] - ref) / scale).max()
    e_ref = (np.abs(out[:, :, cols] - ref) / scale).max()
    print("tight dynamic variances: strip vs oracle", e_ref, "generic vs oracle", e_gen)
    assert e_ref <= max(1e-9, 10 * e_gen)
    assert (np.abs(out - gen) / (np.abs(gen).max(axis=1, keepdims=True) + 1e-300)).max() <= 1e-8
    a, _ = _hip.forward(mg, vg, STD3, L, algo=_hip.ALGO_STRIP)
    assert np.array_equal(a.cpu().numpy(), out)       # the route depends on the data only


def test_strip_kernel_inside_a_hip_graph():
    """The strip launch (control-word memset, main kernel, verdict kernel) captured into a HIP graph and replayed on
    new data, interleaved with eager launches of another shape on the same scratch: every replay must start from
    zeroed control words (the memset is always part of a captured launch)."""
    import torch
    from nnmnkwii_amd import _hip
    STD3 = WINDOW_SETS["std3"]
    gen = torch.Generator(device="cuda").manual_seed(3)
    B, T, sd = 8, 700, 60
    m = torch.randn(B, T, 3 * sd, dtype=torch.float64, device="cuda", generator=gen)
    v = torch.rand(B, T, 3 * sd, dtype=torch.float64, device="cuda", generator=gen) + 0.1
    pw = _hip.prepack_windows(STD3)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        _hip.forward(m, v, pw, algo=_hip.ALGO_STRIP)                 # warm the scratch of this stream
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            y, st = _hip.forward(m, v, pw, algo=_hip.ALGO_STRIP)
        for k in range(3):
            m.copy_(torch.randn(B, T, 3 * sd, dtype=torch.float64, device="cuda", generator=gen))
            if k == 1:
                v[2, 300, 7] = -1e-3                                  # a failing system in this replay only
            ref, sref = _hip.forward(m, v, pw, algo=_hip.ALGO_GENERIC)
            # an eager strip launch of another shape in between (same stream, same scratch, larger control area)
            _hip.forward(m[:, :100].contiguous().repeat(3, 1, 1), v[:, :100].contiguous().repeat(3, 1, 1), pw, algo=_hip.ALGO_STRIP)
            g.replay()
            torch.cuda.synchronize()
            assert torch.equal(st, sref)
            assert float((y - ref).abs().max()) <= 1e-9 * float(ref.abs().max())
            if k == 1:
                assert int(st.reshape(B, sd)[2, 7]) == 301 and not y[2, :, 7].any()
                v[2, 300, 7] = 0.5


def test_strip_whole_utterance_route_many_groups_long_utterances():
    """Round-2 advisor scenario: >= 8 system groups, T = 8000 (125 strips > the 64 workgroups of one XCD) and
    variances whose dynamic features are 1e3 / 1e4 x tighter than the static ones, so that every strip takes the
    whole-utterance route.  With a whole utterance per XCD list each XCD's workgroups would hold the first strips of
    their own utterance and starve; the launcher deals such an utterance to the lists in blocks of at most 32
    consecutive strips (every list running through the utterances in the same order), so the oldest unfinished
    utterance always has all its strips drawn.  No time-out, same numbers as the natural-order kernel."""
    import torch
    from nnmnkwii_amd import _hip
    STD3 = WINDOW_SETS["std3"]
    g = torch.Generator(device="cuda").manual_seed(5)
    B, T, sd = 10, 8000, 60
    m = torch.randn(B, T, 3 * sd, dtype=torch.float64, device="cuda", generator=g)
    v = torch.rand(B, T, 3 * sd, dtype=torch.float64, device="cuda", generator=g) + 0.1
    v[:, :, sd:2 * sd] *= 1e-3
    v[:, :, 2 * sd:] *= 1e-4
    out, status = _hip.forward(m, v, STD3, algo=_hip.ALGO_STRIP)
    assert int(status.abs().max().item()) == 0           # -1 would be a timed-out inter-workgroup wait
    gen, _ = _hip.forward(m, v, STD3, algo=_hip.ALGO_GENERIC)
    scale = gen.abs().amax(dim=1, keepdim=True)
    assert float(((out - gen).abs() / scale).max()) <= 1e-7
    out2, _ = _hip.forward(m, v, STD3, algo=_hip.ALGO_STRIP)
    assert torch.equal(out, out2)


@pytest.mark.parametrize("tight", [False, True])
def test_strip_long_ragged_utterances_dealt_in_blocks(tight):
    """T = 4000: 63 strips = two blocks of 32 per utterance (the second one a strip short: a ticket that is nobody's),
    ragged lengths down to one frame, 20 utterances in two dim groups; ordinary variances (narrow level-3 windows that
    cross the block ends) and tight dynamic ones (whole-utterance route across the lists)."""
    import torch
    from nnmnkwii_amd import _hip
    STD3 = WINDOW_SETS["std3"]
    g = torch.Generator(device="cuda").manual_seed(15)
    B, T, sd = 20, 4000, 70
    m = torch.randn(B, T, 3 * sd, dtype=torch.float64, device="cuda", generator=g)
    v = torch.rand(B, T, 3 * sd, dtype=torch.float64, device="cuda", generator=g) + 0.1
    if tight:
        v[:, :, sd:2 * sd] *= 1e-3
        v[:, :, 2 * sd:] *= 1e-4
    lengths = torch.tensor([4000, 3999, 2049, 2048, 2047, 1, 65, 3000] + [4000 - 97 * k for k in range(12)],
                           dtype=torch.int32, device="cuda")
    out, status = _hip.forward(m, v, STD3, lengths, algo=_hip.ALGO_STRIP)
    assert int(status.abs().max().item()) == 0
    gen, _ = _hip.forward(m, v, STD3, lengths, algo=_hip.ALGO_GENERIC)
    scale = gen.abs().amax(dim=1, keepdim=True).clamp_min(1e-300)
    assert float(((out - gen).abs() / scale).max()) <= 1e-7
    for b_ in range(B):
        assert not bool(out[b_, int(lengths[b_]):].any())
    out2, _ = _hip.forward(m, v, STD3, lengths, algo=_hip.ALGO_STRIP)
    assert torch.equal(out, out2)


def test_strip_whole_utterance_route_while_another_stream_holds_cus():
    """The whole-utterance route at the largest strip count the kernel accepts (T = 16 384) while a second stream keeps
    the GPU busy with large elementwise kernels: the persistent grid is sized from the runtime's occupancy answer and
    every wait is on strips that resident workgroups hold, so the result arrives (status 0) however the two kernels
    share the CUs."""
    import torch
    from nnmnkwii_amd import _hip
    STD3 = WINDOW_SETS["std3"]
    g = torch.Generator(device="cuda").manual_seed(6)
    B, T, sd = 4, 16384, 60
    m = torch.randn(B, T, 3 * sd, dtype=torch.float64, device="cuda", generator=g)
    v = torch.rand(B, T, 3 * sd, dtype=torch.float64, device="cuda", generator=g) + 0.1
    v[:, :, sd:2 * sd] *= 1e-3
    v[:, :, 2 * sd:] *= 1e-4
    ref, _ = _hip.forward(m, v, STD3, algo=_hip.ALGO_GENERIC)
    torch.cuda.synchronize()
    hog = torch.randn(64 << 20, device="cuda")
    side = torch.cuda.Stream()
    outs = []
    for rep in range(4):
        with torch.cuda.stream(side):
            for _ in range(20):
                hog = torch.sin(hog) * 1.0001
        o, st = _hip.forward(m, v, STD3, algo=_hip.ALGO_STRIP)
        outs.append((o, st))
    torch.cuda.synchronize()
    scale = ref.abs().amax(dim=1, keepdim=True)
    for o, st in outs:
        assert int(st.abs().max().item()) == 0
        assert float(((o - ref).abs() / scale).max()) <= 1e-7


def test_retired_pipe_algo_selects_the_strip_kernel():
    """MLPG_HIP_ALGO_PIPE (= 4; the software-pipelined kernel of round 3, removed from the tree in round 6; git history: tools/experimental/pipe) is still accepted and
    runs the strip kernel: same bits, one strip launch."""
    import torch
    from nnmnkwii_amd import _hip
    STD3 = WINDOW_SETS["std3"]
    g = torch.Generator(device="cuda").manual_seed(21)
    m = torch.randn(16, 700, 180, dtype=torch.float64, device="cuda", generator=g)
    v = torch.rand(16, 700, 180, dtype=torch.float64, device="cuda", generator=g) + 0.1
    n0 = _hip.lib().mlpg_hip_launch_count(2)
    a, _ = _hip.forward(m, v, STD3, algo=_hip.ALGO_PIPE)
    assert _hip.lib().mlpg_hip_launch_count(2) == n0 + 1
    b, _ = _hip.forward(m, v, STD3, algo=_hip.ALGO_STRIP)
    assert torch.equal(a, b)


@pytest.mark.parametrize("dt", [np.float64, np.float32])
def test_strip_ladder_rungs_with_variance_regimes_changing_along_the_utterance(dt):
    """Round 5: level 3 tries the 3-strip window r-1 .. r+1 first (a strip routes itself there when its own transfer factor
    is below 2^-66), then the 5-strip one, then the whole utterance.  Here the variance regime changes every few strips
    along the utterance -- ordinary (one order of magnitude), dynamic features 10 x / 100 x tighter, 100 x / 1000 x tighter
    -- so that strips with tiny transfer factors sit next to slowly decaying ones: their 3-strip attempt is rejected by the
    bound and the ladder is climbed.  Every utterance against the oracle and the natural-order kernel; ragged lengths;
    repeated launches bitwise equal (the rung a strip ends on depends on the data only)."""
    import torch
    from nnmnkwii_amd import _hip
    STD3 = WINDOW_SETS["std3"]
    rng = np.random.RandomState(2025)
    B, T, sd = 6, 1536, 60
    m = rng.randn(B, T, 3 * sd).astype(dt)
    v = (rng.rand(B, T, 3 * sd) + 0.1)
    scale_d = np.ones((B, T, 1))
    for b in range(B):
        for s0 in range(0, T, 128):                       # two strips per regime
            reg = rng.randint(3) if b else (s0 // 128) % 3
            f1, f2 = ((1.0, 1.0), (0.1, 0.01), (0.01, 0.001))[reg]
            v[b, s0:s0 + 128, sd:2 * sd] *= f1
            v[b, s0:s0 + 128, 2 * sd:] *= f2
    v = v.astype(dt)
    lengths = np.array([T, T - 1, 1000, 65, 700, T], dtype=np.int32)
    for b in range(B):
        m[b, lengths[b]:] = 0
    ref, _, rc = O.mlpg_batch(m, v, STD3, lengths)
    assert rc == 0
    mg, vg, L = torch.from_numpy(m).cuda(), torch.from_numpy(v).cuda(), torch.from_numpy(lengths).cuda()
    out, status = _hip.forward(mg, vg, STD3, L, algo=_hip.ALGO_STRIP)
    out2, _ = _hip.forward(mg, vg, STD3, L, algo=_hip.ALGO_STRIP)
    gen, _ = _hip.forward(mg, vg, STD3, L, algo=_hip.ALGO_GENERIC)
    assert int(status.abs().max().item()) == 0 and torch.equal(out, out2)
    out, gen = out.cpu().numpy().astype(np.float64), gen.cpu().numpy().astype(np.float64)
    scale = np.abs(ref).max(axis=1, keepdims=True)
    tol = 1e-9 if dt == np.float64 else 5e-6
    e_ref = (np.abs(out - ref) / scale).max()
    e_gen = (np.abs(gen - ref) / scale).max()
    assert e_ref <= max(tol, 10 * e_gen), (e_ref, e_gen)
    go = torch.from_numpy(rng.randn(B, T, sd).astype(dt)).cuda()
    gs, _ = _hip.backward(vg, go, STD3, 3 * sd, L, out_dtype=torch.float64, algo=_hip.ALGO_STRIP)
    gg, _ = _hip.backward(vg, go, STD3, 3 * sd, L, out_dtype=torch.float64, algo=_hip.ALGO_GENERIC)
    assert float((gs - gg).abs().max()) <= (1e-7 if dt == np.float64 else 1e-5) * float(gg.abs().max())


def _scatter_failures(rng, v, B, T, sd, lengths, n):
    """n negative variances at random live places of a (B, T, 3 sd) tensor (torch, on the GPU)."""
    import torch
    bs = rng.randint(0, B, size=n)
    ds = rng.randint(0, sd, size=n)
    ws = rng.randint(0, 3, size=n)
    ts = np.array([rng.randint(2 if w else 0, max(lengths[b] - (2 if w else 0), 3)) for b, w in zip(bs, ws)])
    v[torch.from_numpy(bs).cuda(), torch.from_numpy(ts).cuda(), torch.from_numpy(ws * sd + ds).cuda()] = -1e-3
    return bs, ds


@pytest.mark.parametrize("direction", ["fwd", "bwd"])
def test_strip_failures_settled_at_full_size(direction):
    """Failing pivots at the config-2 size: 4096 strips over 512 workgroups on all eight XCDs (whose L2s are not coherent with each
    other), ~60 failures scattered over utterances, dims, windows and strips.  Status equal to the natural-order kernel's, failed
    columns exactly zero (no strip's rows land on top of the zeros the verdict writes), the other columns untouched by it; twenty
    launches in a row, a clean launch after each (it starts from the control words a launch with marks left behind)."""
    import torch
    from nnmnkwii_amd import _hip
    STD3 = WINDOW_SETS["std3"]
    rng = np.random.RandomState(77)
    B, T, sd = 256, 1000, 60
    g = torch.Generator(device="cuda").manual_seed(5)
    m = torch.randn(B, T, 3 * sd, dtype=torch.float64, device="cuda", generator=g)
    v_ok = torch.rand(B, T, 3 * sd, dtype=torch.float64, device="cuda", generator=g) + 0.1
    go = torch.randn(B, T, sd, dtype=torch.float64, device="cuda", generator=g)
    lengths = np.full(B, T, dtype=np.int32)

    def run(v, algo):
        if direction == "fwd":
            return _hip.forward(m, v, STD3, algo=algo)
        o, st = _hip.backward(v, go, STD3, 3 * sd, out_dtype=torch.float64, algo=algo)
        return o.view(B, T, 3, sd), st

    clean, st0 = run(v_ok, _hip.ALGO_STRIP)
    assert int(st0.abs().max()) == 0
    n0 = _hip.lib().mlpg_hip_launch_count(2)
    for it in range(20):
        v = v_ok.clone()
        bs, ds = _scatter_failures(rng, v, B, T, sd, lengths, 60)
        out, st = run(v, _hip.ALGO_STRIP)
        again, st_again = run(v_ok, _hip.ALGO_STRIP)            # starts from the control words the failing launch left
        ref, st_ref = run(v, _hip.ALGO_GENERIC) if it < 2 else (None, None)
        st = st.view(B, sd)
        failed = torch.zeros(B, sd, dtype=torch.bool, device="cuda")
        failed[torch.from_numpy(bs).cuda(), torch.from_numpy(ds).cuda()] = True
        assert torch.equal(st != 0, failed), it
        assert int(st_again.abs().max()) == 0 and torch.equal(again, clean), it
        bad = failed[:, None, :] if direction == "fwd" else failed[:, None, None, :]
        assert not bool((out != 0)[bad.expand_as(out)].any()), it
        if ref is not None:
            assert torch.equal(st, st_ref.view(B, sd))
            good = ~bad.expand_as(out)
            assert float((out - ref)[good].abs().max()) <= 1e-9 * float(ref[good].abs().max())
    assert _hip.lib().mlpg_hip_launch_count(2) == n0 + 40


def test_strip_failures_settled_over_many_groups():
    """The same with many system groups (4500 short utterances, a control area over 1 MB): failures scattered over them, then a
    clean launch."""
    import torch
    from nnmnkwii_amd import _hip
    STD3 = WINDOW_SETS["std3"]
    rng = np.random.RandomState(78)
    B, T, sd = 4500, 100, 8
    g = torch.Generator(device="cuda").manual_seed(6)
    m = torch.randn(B, T, 3 * sd, dtype=torch.float64, device="cuda", generator=g)
    v_ok = torch.rand(B, T, 3 * sd, dtype=torch.float64, device="cuda", generator=g) + 0.1
    lengths = np.full(B, T, dtype=np.int32)
    clean, st0 = _hip.forward(m, v_ok, STD3, algo=_hip.ALGO_STRIP)
    assert int(st0.abs().max()) == 0
    v = v_ok.clone()
    bs, ds = _scatter_failures(rng, v, B, T, sd, lengths, 80)
    out, st = _hip.forward(m, v, STD3, algo=_hip.ALGO_STRIP)
    again, st_again = _hip.forward(m, v_ok, STD3, algo=_hip.ALGO_STRIP)
    ref, st_ref = _hip.forward(m, v, STD3, algo=_hip.ALGO_GENERIC)
    assert torch.equal(st, st_ref) and int((st != 0).sum()) == len(set(zip(bs.tolist(), ds.tolist())))
    assert int(st_again.abs().max()) == 0 and torch.equal(again, clean)
    bad = (st.view(B, sd) != 0)[:, None, :].expand_as(out)
    assert not bool((out != 0)[bad].any())
    assert float((out - ref)[~bad].abs().max()) <= 1e-9 * float(ref[~bad].abs().max())
