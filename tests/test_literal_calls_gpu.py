"""The LITERAL drop-in calls (-m gpu): one numpy -> numpy ``paramgen.mlpg(mean_frames (T, D), variance_frames, windows)``
per utterance, as the reference's users write it (paramgen/_mlpg.py:92; the loop: util/__init__.py:56-66).

Such a call is small -- 9.6 KB at BASELINE config 1, 2.9 MB at one config-2 utterance -- and takes the SHORT PATH of
mlpg_hip_forward_host (csrc/host_api.hip host_small: one stream, one pinned staging buffer, no thread, the kernel
writing into pinned host memory, a polled sequence number).  Checked here: WHICH route a call takes (the library's call
counters 10 / 11), parity of that route with the oracle in every variance mode / dtype / with lengths / at the edge
lengths, the reference's exception for a failing pivot, that the per-list window cache follows in-place edits, growth of
the cached buffers, and calls from two threads.
"""
import threading

import numpy as np
import pytest

from cases import WINDOW_SETS
from oracle import mlpg as O

pytestmark = pytest.mark.gpu

W = WINDOW_SETS["std3"]
ROUTE_COPIED, ROUTE_DIRECT = 10, 11


def _routes():
    from nnmnkwii_amd import _hip
    L = _hip.lib()
    return int(L.mlpg_hip_launch_count(ROUTE_COPIED)), int(L.mlpg_hip_launch_count(ROUTE_DIRECT))


def _rel(y, yo):
    return float(np.abs(y - yo).max() / max(np.abs(yo).max(), 1e-300))


def test_config1_call_takes_the_direct_route_and_matches_the_oracle():
    """BASELINE config 1 as SURVEY 8(d) states it: means = rng.rand(100, 6), variances = rng.rand(6) tiled."""
    from nnmnkwii_amd import paramgen as G
    rng = np.random.RandomState(1234)
    m = rng.rand(100, 6)
    vg = rng.rand(6)
    v = np.tile(vg, (100, 1))
    c0, d0 = _routes()
    y = G.mlpg(m, v, W)
    c1, d1 = _routes()
    assert (c1 - c0, d1 - d0) == (0, 1)              # 9.6 KB: the kernel reads the pinned staging buffer itself
    assert y.shape == (100, 2) and y.dtype == np.float64
    assert _rel(y, O.mlpg(m, v, W)) <= 1e-12
    yg = G.mlpg(m, vg, W)                            # global (D,) variances: the same trajectory
    assert _rel(yg, O.mlpg(m, vg, W)) <= 1e-12
    assert _routes() == (c1, d1 + 1)
    # float32 in -> float32 out (the dtype of the means, _mlpg.py:166,183)
    y32 = G.mlpg(m.astype(np.float32), v.astype(np.float32), W)
    assert y32.dtype == np.float32 and _rel(y32, O.mlpg(m.astype(np.float32), v.astype(np.float32), W)) <= 2e-6


def test_config2_utterance_takes_the_copied_route_and_matches_the_oracle():
    from nnmnkwii_amd import paramgen as G
    rng = np.random.RandomState(7)
    m = rng.randn(1000, 180)
    v = rng.rand(1000, 180) + 0.1
    c0, d0 = _routes()
    y = G.mlpg(m, v, W)
    assert _routes() == (c0 + 1, d0)                 # 2.9 MB: staged, copied to the device array by array
    assert _rel(y, O.mlpg(m, v, W)) <= 1e-12
    # the same utterance as a batch of one through mlpg_batch, and against the device-tensor entry point
    import torch
    from nnmnkwii_amd import _hip
    yb = G.mlpg_batch(m[None], v[None], W)
    assert np.array_equal(yb[0], y)
    yd, st = _hip.forward(torch.from_numpy(m[None]).cuda(), torch.from_numpy(v[None]).cuda(), W)
    assert int(st.abs().max()) == 0 and np.array_equal(yd[0].cpu().numpy(), y)
    y32 = G.mlpg(m.astype(np.float32), v.astype(np.float32), W)
    assert y32.dtype == np.float32 and _rel(y32, O.mlpg(m.astype(np.float32), v.astype(np.float32), W)) <= 2e-6


def test_a_buffer_handed_in_again_is_copied_in_place_with_the_same_result():
    """A config-2 utterance (two arrays of 1.44 MB): staged through the pinned buffer the first time, copied by the runtime straight from
    the caller's arrays from the second call with the same arrays on -- the same bits either way; an edit of the arrays in between is seen."""
    from nnmnkwii_amd import paramgen as G
    rng = np.random.RandomState(41)
    m = rng.randn(1000, 180)
    v = rng.rand(1000, 180) + 0.1
    y1 = G.mlpg(m, v, W)                              # first sight: staged
    y2 = G.mlpg(m, v, W)                              # seen before: direct
    y3 = G.mlpg(m.copy(), v.copy(), W)                # other arrays, same numbers
    assert np.array_equal(y1, y2) and np.array_equal(y1, y3)
    assert _rel(y1, O.mlpg(m, v, W)) <= 1e-12
    m[500:] *= 2.0
    v[:500] *= 0.5
    y4 = G.mlpg(m, v, W)
    assert _rel(y4, O.mlpg(m, v, W)) <= 1e-12 and not np.array_equal(y4, y1)
    g1 = G.mlpg_grad(m, v, W, y4)
    g2 = G.mlpg_grad(m, v, W, y4)
    assert np.array_equal(g1, g2)


def test_a_preallocated_result_array_is_filled_in_place_from_the_second_call_on():
    """The C entry point with the caller's own result array (1.9 MB: between the two direct-copy thresholds): staged through pinned
    memory at first sight, written by the runtime's device -> host copy straight into the array when it is handed in again -- same bits,
    and the bytes around the array are left alone."""
    import ctypes
    from nnmnkwii_amd import _hip
    L = _hip.lib()
    rng = np.random.RandomState(43)
    B, T, sd = 4, 1000, 60
    M_ = rng.randn(B, T, 3 * sd)
    V_ = rng.rand(B, T, 3 * sd) + 0.1
    pw = _hip.cached_windows(W)
    pl, pu, pc = pw.ptrs()
    guard = 4096
    buf = np.full(B * T * sd + 2 * guard, 7.25)
    out = buf[guard:guard + B * T * sd].reshape(B, T, sd)
    status = np.zeros((B, sd), dtype=np.int32)
    results = []
    for k in range(3):
        out[...] = -1.0
        rc = L.mlpg_hip_forward_host(0, _hip.F64, 0, M_.ctypes.data, V_.ctypes.data, _hip.VAR_FRAME, None, B, T, 3 * sd, 3, pl, pu, pc,
                                     out.ctypes.data, status.ctypes.data)
        assert rc == 0 and not status.any()
        assert (buf[:guard] == 7.25).all() and (buf[-guard:] == 7.25).all()
        results.append(out.copy())
    assert np.array_equal(results[0], results[1]) and np.array_equal(results[0], results[2])
    yo, _, rc = O.mlpg_batch(M_, V_, W)
    assert rc == 0 and _rel(results[0], yo) <= 1e-12


def test_memmapped_inputs_and_a_file_backed_result(tmp_path):
    """Arrays that are not ordinary anonymous memory at the sizes the runtime copies directly (>= 4 MB): read-only np.memmap inputs
    (np.load(..., mmap_mode="r")) and a writable file-backed result array handed to the C entry point -- the same bits as on copies."""
    from nnmnkwii_amd import _hip
    from nnmnkwii_amd import paramgen as G
    rng = np.random.RandomState(47)
    B, T, sd = 6, 1000, 60
    M_ = rng.randn(B, T, 3 * sd)
    V_ = rng.rand(B, T, 3 * sd) + 0.1
    want = G.mlpg_batch(M_.copy(), V_.copy(), W)
    np.save(tmp_path / "m.npy", M_)
    np.save(tmp_path / "v.npy", V_)
    Mr = np.load(tmp_path / "m.npy", mmap_mode="r")
    Vr = np.load(tmp_path / "v.npy", mmap_mode="r")
    assert not Mr.flags.writeable
    for _ in range(2):
        assert np.array_equal(G.mlpg_batch(Mr, Vr, W), want)
    out = np.lib.format.open_memmap(tmp_path / "y.npy", mode="w+", dtype=np.float64, shape=(B, T, sd))
    st = np.zeros((B, sd), dtype=np.int32)
    pl, pu, pc = _hip.cached_windows(W).ptrs()
    for _ in range(2):
        out[...] = 0
        rc = _hip.lib().mlpg_hip_forward_host(0, _hip.F64, 0, Mr.ctypes.data, Vr.ctypes.data, _hip.VAR_FRAME, None, B, T, 3 * sd, 3, pl, pu, pc,
                                              out.ctypes.data, st.ctypes.data)
        assert rc == 0 and np.array_equal(np.asarray(out), want)


def test_a_config2_batch_does_not_take_the_short_path():
    from nnmnkwii_amd import paramgen as G
    rng = np.random.RandomState(3)
    M_ = rng.randn(50, 500, 180)                      # 72 MB of means and variances (the limit is 64 MiB): the chunked path
    V_ = rng.rand(50, 500, 180) + 0.1
    r0 = _routes()
    y = G.mlpg_batch(M_, V_, W)
    assert _routes() == r0
    yo, _, rc = O.mlpg_batch(M_, V_, W)
    assert rc == 0 and _rel(y, yo) <= 1e-12


@pytest.mark.parametrize("dt", [np.float64, np.float32])
@pytest.mark.parametrize("mode", ["frame", "global", "unit"])
def test_a_batch_of_a_few_utterances_goes_straight_from_and_to_the_callers_arrays(mode, dt):
    """Arrays of 4 MB and more -- and arrays of 1.2 MB and more that the library has been handed before -- are not staged: the runtime
    copies them straight from / to the caller's pageable memory (MLPG_HIP_HOST_DIRECT_KB / _ALWAYS_KB).  8 utterances of T = 1000 x 60 dims (23 MB of means and variances in float64, 3.8 MB out), ragged with
    junk in the padding: one call on the one-stream path, equal to the device entry point bit for bit, the oracle within its bound;
    the caller's arrays are left untouched."""
    import torch
    from nnmnkwii_amd import _hip
    from nnmnkwii_amd import paramgen as G
    rng = np.random.RandomState(17)
    B, T, sd = 8, 1000, 60
    M_ = rng.randn(B, T, 3 * sd).astype(dt)
    V_ = (rng.rand(B, T, 3 * sd) + 0.1).astype(dt)
    lengths = np.array([1000, 999, 3, 0, 517, 1000, 64, 2], dtype=np.int32)
    for b in range(B):
        M_[b, lengths[b]:] = 1e30
    var = {"frame": V_, "global": V_[0, 0].copy(), "unit": None}[mode]
    M0, V0 = M_.copy(), V_.copy()
    r0 = _routes()
    y = G.mlpg_batch(M_, var, W, lengths)
    assert sum(_routes()) == sum(r0) + 1
    assert np.array_equal(M_, M0) and np.array_equal(V_, V0)
    yd, st = _hip.forward(torch.from_numpy(M_).cuda(), None if var is None else torch.from_numpy(var).cuda(), W, torch.from_numpy(lengths).cuda())
    assert int(st.abs().max()) == 0 and np.array_equal(y, yd.cpu().numpy())
    yo, _, rc = O.mlpg_batch(M_, var if var is not None else np.ones(3 * sd, dtype=dt), W, lengths)
    assert rc == 0
    for b in range(B):
        if lengths[b]:
            assert _rel(y[b, :lengths[b]].astype(np.float64), yo[b, :lengths[b]]) <= (1e-12 if dt == np.float64 else 2e-6)
        assert not y[b, lengths[b]:].any()
    # the backward of the same batch: the gradient (B, T, D) goes back the same way
    go = rng.randn(B, T, sd).astype(dt)
    g, stb = _hip.backward_host(var, go, W, 3 * sd, out_dtype=np.float32, lengths=lengths)
    gd, _ = _hip.backward(None if var is None else torch.from_numpy(var).cuda(), torch.from_numpy(go).cuda(), W, 3 * sd,
                          lengths=torch.from_numpy(lengths).cuda(), out_dtype=torch.float32)
    assert not stb.any() and np.array_equal(g, gd.cpu().numpy())


@pytest.mark.parametrize("T", [1, 2, 3, 4, 17, 64, 100, 333])
@pytest.mark.parametrize("sd", [1, 2, 25, 60])
def test_short_path_edge_lengths_and_widths(T, sd):
    """T = 1, 2: every dynamic precision is zeroed (_mlpg.py:191-193) and y equals the static means."""
    from nnmnkwii_amd import paramgen as G
    rng = np.random.RandomState(100 * T + sd)
    m = rng.randn(T, 3 * sd)
    v = rng.rand(T, 3 * sd) + 0.05
    r0 = _routes()
    y = G.mlpg(m, v, W)
    assert sum(_routes()) == sum(r0) + 1
    yo = O.mlpg(m, v, W)
    assert _rel(y, yo) <= 1e-12
    if T <= 2:
        assert np.allclose(y, m[:, :sd], rtol=1e-14, atol=0.0)      # (tau mu) / tau: the static means to the last bit or two


@pytest.mark.parametrize("wname", ["std3", "wide3", "asym2"])
@pytest.mark.parametrize("mode", ["frame", "global", "unit"])
@pytest.mark.parametrize("dt", [np.float64, np.float32])
def test_short_path_small_batches_with_lengths(wname, mode, dt):
    """mlpg_batch on a small padded batch with ragged lengths (incl. 1 and the full length) and junk in the padding."""
    from nnmnkwii_amd import paramgen as G
    if wname not in WINDOW_SETS:
        pytest.skip("no such window set in tests/golden/cases.py")
    w = WINDOW_SETS[wname]
    nw = len(w)
    rng = np.random.RandomState(11)
    B, T, sd = 5, 90, 7
    M_ = rng.randn(B, T, nw * sd).astype(dt)
    V_ = (rng.rand(B, T, nw * sd) + 0.1).astype(dt)
    lengths = np.array([T, 1, 37, 2, 64], dtype=np.int32)
    for b in range(B):
        M_[b, lengths[b]:] = 1e30                     # the padding may hold anything
    var = {"frame": V_, "global": V_[0, 0].copy(), "unit": None}[mode]
    r0 = _routes()
    y = G.mlpg_batch(M_, var, w, lengths)
    assert sum(_routes()) == sum(r0) + 1
    vo = var if var is not None else np.ones(nw * sd, dtype=dt)
    yo, _, rc = O.mlpg_batch(M_, vo, w, lengths)
    assert rc == 0 and y.dtype == dt
    tol = 1e-10 if dt == np.float64 else 5e-6
    for b in range(B):
        n = lengths[b]
        assert _rel(y[b, :n], yo[b, :n]) <= tol
        assert not y[b, n:].any()                     # output frames beyond the length are zero


def test_short_path_failing_pivot_raises_the_references_error():
    from nnmnkwii_amd import paramgen as G
    rng = np.random.RandomState(5)
    m = rng.randn(200, 6)
    v = rng.rand(200, 6) + 0.1
    v[57, 1] = -1e-3
    with pytest.raises(np.linalg.LinAlgError) as e:
        G.mlpg(m, v, W)
    assert str(e.value) == "58-th leading minor not positive definite"   # linalg.pyx:79-82: k = first failing frame + 1
    # ... and the next call is fine
    v[57, 1] = 0.3
    assert _rel(G.mlpg(m, v, W), O.mlpg(m, v, W)) <= 1e-12
    with pytest.raises(AssertionError):
        G.mlpg(m, v[:, :5], W)                        # shape mismatch: the reference's assert (_mlpg.py:171)


def test_window_cache_follows_the_window_list():
    """The packed window tables are remembered per list object; an edit of the list or of a coefficient array in place
    must be seen by the next call."""
    from nnmnkwii_amd import paramgen as G
    rng = np.random.RandomState(9)
    m = rng.randn(120, 6)
    v = rng.rand(120, 6) + 0.1
    w = [(0, 0, np.array([1.0])), (1, 1, np.array([-0.5, 0.0, 0.5])), (1, 1, np.array([1.0, -2.0, 1.0]))]
    y0 = G.mlpg(m, v, w)
    assert _rel(y0, O.mlpg(m, v, w)) <= 1e-12
    assert np.array_equal(G.mlpg(m, v, w), y0)
    w[1][2][:] = [-0.25, 0.0, 0.25]                  # a coefficient array edited in place
    y1 = G.mlpg(m, v, w)
    assert _rel(y1, O.mlpg(m, v, w)) <= 1e-12 and not np.array_equal(y1, y0)
    w[2] = (1, 1, np.array([0.5, -1.0, 0.5]))         # an entry replaced
    y2 = G.mlpg(m, v, w)
    assert _rel(y2, O.mlpg(m, v, w)) <= 1e-12 and not np.array_equal(y2, y1)
    w.pop()                                          # two windows; D = 4
    y3 = G.mlpg(m[:, :4], v[:, :4], w)
    assert _rel(y3, O.mlpg(np.ascontiguousarray(m[:, :4]), np.ascontiguousarray(v[:, :4]), w)) <= 1e-12
    # windows given as plain lists / tuples of numbers
    wl = [(0, 0, [1.0]), (1, 1, (-0.5, 0.0, 0.5))]
    assert np.array_equal(G.mlpg(m[:, :4], v[:, :4], wl), G.mlpg(m[:, :4], v[:, :4], [(0, 0, np.array([1.0])), (1, 1, np.array([-0.5, 0.0, 0.5]))]))


def test_short_path_buffers_grow_and_shrinking_calls_reuse_them():
    from nnmnkwii_amd import paramgen as G
    rng = np.random.RandomState(21)
    for T, sd in ((10, 1), (900, 60), (40, 3), (1900, 60), (100, 2), (2000, 30), (5, 60)):
        m = rng.randn(T, 3 * sd)
        v = rng.rand(T, 3 * sd) + 0.1
        r0 = _routes()
        y = G.mlpg(m, v, W)
        assert sum(_routes()) == sum(r0) + 1
        assert _rel(y, O.mlpg(m, v, W)) <= 1e-12
    # non-contiguous and non-float inputs behave as in the reference
    m = rng.randn(64, 12)
    v = rng.rand(64, 12) + 0.1
    assert np.array_equal(G.mlpg(m[:, ::2], v[:, ::2], W), G.mlpg(np.ascontiguousarray(m[:, ::2]), np.ascontiguousarray(v[:, ::2]), W))
    mi = (m * 10).astype(np.int64)
    yi = G.mlpg(mi, v, W)
    assert yi.dtype == np.int64 or yi.dtype == np.float64     # (the reference casts the result to the means' dtype)


def test_the_loop_over_utterances_and_two_threads():
    """[paramgen.mlpg(m, v, windows) for m, v in utterances] (util/__init__.py:56-66), then the same from two threads."""
    from nnmnkwii_amd import paramgen as G
    rng = np.random.RandomState(31)
    utts = [(rng.randn(T, 18), rng.rand(T, 18) + 0.1) for T in rng.randint(1, 400, size=40)]
    want = [O.mlpg(m, v, W) for m, v in utts]
    got = [G.mlpg(m, v, W) for m, v in utts]
    assert max(_rel(a, b) for a, b in zip(got, want)) <= 1e-12
    out = [None, None]

    def run(k):
        out[k] = [G.mlpg(m, v, W) for m, v in utts[k::2]]

    th = [threading.Thread(target=run, args=(k,)) for k in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for k in range(2):
        assert all(np.array_equal(a, b) for a, b in zip(out[k], got[k::2]))


def test_a_config2_utterance_call_is_faster_than_the_reference_by_a_wide_margin():
    """Not a benchmark (tools/bench_paths.py --only lit is): a guard that the short path has not fallen back to something
    slow -- one config-2 utterance takes the reference 2.9 ms on the box's host; the call must stay below 1 ms."""
    import time
    from nnmnkwii_amd import paramgen as G
    rng = np.random.RandomState(2)
    m = rng.randn(1000, 180)
    v = rng.rand(1000, 180) + 0.1
    for _ in range(10):
        G.mlpg(m, v, W)
    ts = []
    for _ in range(50):
        t0 = time.perf_counter()
        G.mlpg(m, v, W)
        ts.append(time.perf_counter() - t0)
    assert float(np.median(ts)) < 1e-3, np.median(ts)


# ---- the literal backward call: paramgen.mlpg_grad on numpy arrays, autograd.MLPG on CPU tensors (mlpg_hip_backward_host) ----

def _grad_rel(g, gr):
    gr = gr.astype(np.float64)                        # (the oracle's gradient is float32, as the reference's)
    scale = np.abs(gr).max(axis=0) + 1e-300
    return float((np.abs(g.astype(np.float64) - gr) / scale).max())


def test_mlpg_grad_config1_takes_the_direct_route_and_matches_the_oracle():
    """paramgen.mlpg_grad(mean_frames, variance_frames, windows, grad_output) at BASELINE config 1 (T = 100, 2 static dims):
    the short path with the kernel reading the pinned staging buffer, float32 (T, D) out as the reference (_mlpg.py:248)."""
    from nnmnkwii_amd import paramgen as G
    rng = np.random.RandomState(11)
    m = rng.rand(100, 6)
    v = rng.rand(100, 6) + 0.1
    go = rng.randn(100, 2)
    c0, d0 = _routes()
    g = G.mlpg_grad(m, v, W, go)
    assert _routes() == (c0, d0 + 1)
    assert g.shape == (100, 6) and g.dtype == np.float32
    assert _grad_rel(g, O.mlpg_grad(m, v, W, go)) <= 1e-6            # (float32 result)
    # float32 variances, float64 grad_output: the arithmetic's inputs are float32, as through the device entry point
    g32 = G.mlpg_grad(m, v.astype(np.float32), W, go)
    assert g32.dtype == np.float32 and _grad_rel(g32, O.mlpg_grad(m, v.astype(np.float32).astype(np.float64), W, go.astype(np.float32))) <= 2e-5
    # a global (D,) variance vector
    gg = G.mlpg_grad(m, v[0], W, go)
    assert _grad_rel(gg, O.mlpg_grad(m, np.tile(v[0], (100, 1)), W, go)) <= 1e-6


@pytest.mark.parametrize("wname", ["std3", "std2", "wide3", "asym2"])
@pytest.mark.parametrize("T,sd", [(1, 3), (2, 1), (3, 2), (5, 7), (17, 1), (64, 60), (333, 25), (1000, 60), (2049, 5)])
def test_mlpg_grad_short_path_against_the_device_entry_point(wname, T, sd):
    """Every length / width / window set: the host-memory call returns exactly what mlpg_hip_backward returns for the same arrays on
    the device (same kernels, same routing; only where the arrays live differs), and the oracle's dense gradient for the small ones."""
    import torch
    from nnmnkwii_amd import _hip
    from nnmnkwii_amd import paramgen as G
    w = WINDOW_SETS[wname]
    nw = len(w)
    rng = np.random.RandomState(T * 131 + sd)
    m = np.zeros((T, nw * sd))
    v = rng.rand(T, nw * sd) + 0.1
    go = rng.randn(T, sd)
    g = G.mlpg_grad(m, v, w, go)
    gd, st = _hip.backward(torch.from_numpy(v[None]).cuda(), torch.from_numpy(go[None]).cuda(), w, nw * sd, out_dtype=torch.float32)
    assert int(st.abs().max()) == 0
    assert np.array_equal(g, gd[0].cpu().numpy())
    if T <= 333:
        assert _grad_rel(g, O.mlpg_grad(m, v, w, go)) <= 2e-6


def test_mlpg_grad_batches_are_cut_into_pieces_of_whole_utterances():
    """mlpg_hip_backward_host on a batch larger than the short path's limit (64 MiB of input): pieces of whole utterances, one after
    the other; lengths and the status rows follow their utterances; global and unit variances."""
    import torch
    from nnmnkwii_amd import _hip
    rng = np.random.RandomState(5)
    B, T, sd = 60, 700, 60
    V_ = rng.rand(B, T, 3 * sd) + 0.1                   # 60 x (1.0 MB of variances + 0.34 MB of grad_out): 49 utterances fit the limit
    go = rng.randn(B, T, sd)
    lengths = rng.randint(0, T + 1, size=B).astype(np.int32)
    lengths[[0, 1, 2, 3, 48, 49, 59]] = [700, 1, 350, 0, 699, 700, 2]
    per = max(1, min(B, (64 << 20) // (T * sd * 8 + T * 3 * sd * 8)))
    c0, d0 = _routes()
    g, st = _hip.backward_host(V_, go, W, 3 * sd, out_dtype=np.float64, lengths=lengths)
    c1, d1 = _routes()
    assert per == 49 and (c1 - c0) + (d1 - d0) == 2 and not st.any()
    # piece by piece the same bits as mlpg_hip_backward on device copies of that piece (AUTO picks its kernel by the batch's size, so
    # the whole batch in one device call may differ in the last bit), the whole batch within rounding
    for b0 in range(0, B, per):
        sl = slice(b0, min(B, b0 + per))
        gp, _ = _hip.backward(torch.from_numpy(V_[sl]).cuda(), torch.from_numpy(go[sl]).cuda(), W, 3 * sd,
                              lengths=torch.from_numpy(lengths[sl]).cuda(), out_dtype=torch.float64)
        assert np.array_equal(g[sl], gp.cpu().numpy())
    gd, _ = _hip.backward(torch.from_numpy(V_).cuda(), torch.from_numpy(go).cuda(), W, 3 * sd, lengths=torch.from_numpy(lengths).cuda(),
                          out_dtype=torch.float64)
    assert np.abs(g - gd.cpu().numpy()).max() <= 1e-12 * np.abs(g).max()
    for b in range(B):
        assert not g[b, lengths[b]:].any()
    del gd, gp
    for var in (V_[0, 0].copy(), None):
        g, st = _hip.backward_host(var, go, W, 3 * sd, out_dtype=np.float32)
        gd, _ = _hip.backward(None if var is None else torch.from_numpy(var).cuda(), torch.from_numpy(go).cuda(), W, 3 * sd, out_dtype=torch.float32)
        assert np.array_equal(g, gd.cpu().numpy()) and not st.any()
    # a failing pivot in utterance 55 (the second piece): the status row of that utterance, the reference's exception with its k
    V_[55, 41, 7] = -1e-9
    g, st = _hip.backward_host(V_, go, W, 3 * sd)
    assert st[55, 7] == 42 and not np.delete(st, 55, axis=0).any()
    from nnmnkwii_amd import paramgen as G
    with pytest.raises(np.linalg.LinAlgError, match="42-th leading minor not positive definite"):
        G.mlpg_grad(np.zeros((T, 3 * sd)), V_[55], W, go[55])


def test_autograd_mlpg_on_cpu_tensors_runs_both_passes_on_the_short_path():
    """autograd.mlpg on CPU tensors (the reference's tensors, autograd/_impl/mlpg.py:50-67): forward and backward are one host-memory
    call each (no torch device tensor), float32 out, the gradient equal to paramgen.mlpg_grad's and in the means' dtype."""
    import torch
    from nnmnkwii_amd import autograd as AF
    from nnmnkwii_amd import paramgen as G
    rng = np.random.RandomState(2)
    for T, sd, dt in ((100, 2, torch.float32), (1000, 60, torch.float32), (100, 2, torch.float64)):
        m = torch.from_numpy(rng.rand(T, 3 * sd)).to(dt).requires_grad_()
        v = torch.from_numpy(rng.rand(T, 3 * sd) + 0.1).to(dt)
        wgt = torch.from_numpy(rng.randn(T, sd)).to(torch.float32)
        r0 = sum(_routes())
        y = AF.mlpg(m, v, W)
        (y * wgt).sum().backward()
        assert sum(_routes()) == r0 + 2
        assert y.dtype == torch.float32 and m.grad.dtype == dt and not m.grad.is_cuda
        want = G.mlpg_grad(m.detach().numpy(), v.numpy(), W, wgt.numpy())
        assert np.array_equal(m.grad.numpy(), want.astype(m.grad.numpy().dtype))
        # the same through CUDA tensors
        mc = m.detach().cuda().requires_grad_()
        (AF.mlpg(mc, v.cuda(), W) * wgt.cuda()).sum().backward()
        assert np.allclose(mc.grad.cpu().numpy(), m.grad.numpy(), rtol=2e-5, atol=1e-7)
