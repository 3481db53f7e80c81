"""Executable specification (numpy) of the constant-coefficient MLPG kernel (csrc/mlpg_const_impl.h, algo CONST).

Global (D,) or unit variances: the precision matrix  P_d = sum_w W_w^T diag(tau_w) W_w  of a static dim depends on
(d, T) only (/root/reference/nnmnkwii/paramgen/_mlpg.py:169-170 tiles the variances over the frames), not on the
utterance.  The scheme:

  setup (once per set of variances):
    natural-order LDL^T of P_d for T = infinity, row by row, until the multipliers have converged to their steady
    state (row i_s); table[i] = (l1_i, l2_i, 1/d_i, d_i) for i <= i_s.  Rows T-2 and T-1 (the reference zeroes the
    dynamic precisions on the last frame, _mlpg.py:191-193) are re-derived per utterance from the table's state.
  solve (one workgroup walks one utterance; lane = static dim, wavefront = chunk of M frames, W chunks = a super-step):
    forward  z_i = b_i - l1_i z_{i-1} - l2_i z_{i-2}                  (L z = b)
    backward y_i = z_i / d_i - l1_{i+1} y_{i+1} - l2_{i+2} y_{i+2}    (L^T y = D^-1 z)
    are second-order linear recurrences with data-independent coefficients.  See solve_utterance.

The model mirrors the kernel's decomposition and is pinned against the oracle on the CPU (tests/test_const_model.py).
Test infrastructure only.
"""
import numpy as np

TOL = 2.0 ** -56          # a chunk waits for the next super-step unless the transfer from the super-step's bottom is below this
CONV = 2.0 ** -50    # relative change below which the factor rows count as converged


def win_coefs(windows):
    """(cm, c0, cp) per window: W[t, t-1], W[t, t], W[t, t+1]; extents <= 1."""
    out = []
    for l, u, c in windows:
        assert l <= 1 and u <= 1
        c = np.asarray(c, dtype=np.float64)
        out.append((c[0] if l else 0.0, c[l], c[l + 1] if u else 0.0))
    return np.asarray(out)


def live(w, t, T, mw):
    """1 where frame t carries precision for window w (_mlpg.py:191-193; T = None: no lower edge)."""
    if t < 0 or (T is not None and t >= T):
        return 0.0
    if w == 0:
        return 1.0
    if mw == 0 or t < mw or (T is not None and t >= T - mw):
        return 0.0
    return 1.0


def p_entries(i, T, tau, wc, mw):
    """a = P[i,i], c = P[i,i-1], e = P[i,i-2] of one static dim (tau: (nw, sd))."""
    nw = len(wc)
    a = c = e = 0.0
    for w in range(nw):
        cm, c0, cp = wc[w]
        tm, t0, tp = (live(w, i - 1, T, mw) * tau[w], live(w, i, T, mw) * tau[w], live(w, i + 1, T, mw) * tau[w])
        a = a + tm * cp * cp + t0 * c0 * c0 + tp * cm * cm
        c = c + tm * cp * c0 + t0 * c0 * cm
        e = e + tm * cp * cm
    if i < 1:
        c = c * 0.0
    if i < 2:
        e = e * 0.0
    return a, c, e


def ldl_row(a, c, e, d1, d2, l1p, i):
    """One row of the pentadiagonal LDL^T from the state (d_{i-1}, d_{i-2}, l1_{i-1})."""
    l2 = e / d2 if i >= 2 else np.zeros_like(a)
    l1 = (c - e * l1p) / d1 if i >= 1 else np.zeros_like(a)
    d = a - l1 * l1 * (d1 if i >= 1 else 0.0) - l2 * l2 * (d2 if i >= 2 else 0.0)
    return l1, l2, d


def factor_table(tau, wc, mw, Tcap):
    """Rows of the T = infinity factor until steady: list of (l1, l2, dinv, d) arrays (sd,), i_s, kfail (sd,)."""
    sd = tau.shape[1]
    rows = []
    d1 = d2 = np.ones(sd)
    l1p = np.zeros(sd)
    kfail = np.zeros(sd, dtype=np.int64)
    steady_run = 0
    for i in range(Tcap):
        a, c, e = p_entries(i, None, tau, wc, mw)
        l1, l2, d = ldl_row(a, c, e, d1, d2, l1p, i)
        bad = ~(d > 0)
        kfail = np.where((kfail == 0) & bad, i + 1, kfail)
        rows.append((l1, l2, 1.0 / d, d))
        if i >= 3:
            pl1, pl2, _, pd = rows[-2]
            same = (np.abs(d - pd) <= CONV * np.abs(d)) & (np.abs(l1 - pl1) <= CONV * np.abs(l1) + 1e-300) & \
                   (np.abs(l2 - pl2) <= CONV * np.abs(l2) + 1e-300)
            steady_run = steady_run + 1 if np.all(same | (kfail > 0)) else 0
            if steady_run >= 2:
                break
        d2, d1, l1p = d1, d, l1
    return rows, len(rows) - 1, kfail


class Coefs:
    """Per-row (l1, l2, dinv) of one utterance of T frames: table rows, steady constants, the two tail rows."""

    def __init__(self, rows, i_s, T, tau, wc, mw):
        self.rows, self.i_s, self.T = rows, i_s, T
        sd = tau.shape[1]
        self.tail = {}
        self.tail_bad = np.zeros(sd, dtype=bool)
        for i in (T - 2, T - 1):
            if i < 0:
                continue
            def state(j):
                if j < 0:
                    return None
                if j in self.tail:
                    return self.tail[j]
                return self.rows[min(j, i_s)]
            r1, r2 = state(i - 1), state(i - 2)
            a, c, e = p_entries(i, T, tau, wc, mw)
            d1 = r1[3] if r1 else np.ones(sd)
            d2 = r2[3] if r2 else np.ones(sd)
            l1p = r1[0] if r1 else np.zeros(sd)
            l1, l2, d = ldl_row(a, c, e, d1, d2, l1p, i)
            self.tail_bad |= ~(d > 0)
            self.tail[i] = (l1, l2, 1.0 / d, d)

    def __call__(self, i):
        """(l1_i, l2_i, dinv_i); identity rows behind the utterance's end."""
        if i >= self.T or i < 0:
            z = np.zeros_like(self.rows[0][0])
            return z, z, z + 1.0
        if i in self.tail:
            return self.tail[i][:3]
        return self.rows[min(i, self.i_s)][:3]


def rhs_rows(mu, tau, wc, mw, T, a, M):
    """b_i for rows a .. a+M-1 of one utterance (mu: (T, nw, sd)); rows >= T give 0."""
    nw, sd = tau.shape
    b = np.zeros((M, sd))
    for k in range(M):
        i = a + k
        if i >= T or i < 0:
            continue
        for w in range(nw):
            cm, c0, cp = wc[w]
            for t, cf in ((i - 1, cp), (i, c0), (i + 1, cm)):
                lv = live(w, t, T, mw)
                if lv:
                    b[k] += cf * tau[w] * mu[t, w]
    return b


def mm(A, B):
    """2x2 products per lane: A, B = (a, b, c, d) tuples of (sd,) arrays."""
    return (A[0] * B[0] + A[1] * B[2], A[0] * B[1] + A[1] * B[3], A[2] * B[0] + A[3] * B[2], A[2] * B[1] + A[3] * B[3])


def mv(A, v):
    return (A[0] * v[0] + A[1] * v[1], A[2] * v[0] + A[3] * v[1])


def amax(A):
    return max(float(np.max(np.abs(x))) for x in A)


def solve_utterance(rhs_fn, T, co, M, W, tol=TOL, stats=None, slots=None, lag_ok=True):
    """y (T, sd) for one utterance, the way stream_kernel walks it.  rhs_fn(a, M) -> b rows (M, sd) of rows a .. a+M-1.

    Chunks of M rows are aligned to the utterance's END (chunk j of NC covers rows T - (NC - j) M ..: the two tail rows
    always are rows M-2, M-1 of the last chunk; chunk 0 may start above row 0, rows < 0 have a zero right-hand side and
    stay zero).  W chunks side by side form a super-step.  The forward state is carried from super-step to super-step
    (exact).  A chunk goes out at once if the chunks between it and its super-step's bottom damp whatever comes up from
    below under `tol`; otherwise it is parked (at most `slots` chunks per super-step, the lowest ones) and goes out one
    super-step later, with the upward state of the NEXT super-step solved with zero input from below.  A chunk that would
    need a slot it does not have sends the utterance to the exact two-sweep path (forward sweep, then super-steps in
    reverse order with the state from below carried exactly).  What the lag leaves out is the transfer matrix of a whole
    steady super-step times the state below it: `lag_ok` (setup: that matrix is below `tol` for every dim and the
    transient ends inside the first super-step) says whether the lag may be tried at all.
    """
    sd = co(0)[0].shape[0]
    slots = W if slots is None else slots
    NC = (T + M - 1) // M
    K = (NC + W - 1) // W
    ident = (np.ones(sd), np.zeros(sd), np.zeros(sd), np.ones(sd))
    zero2 = (np.zeros(sd), np.zeros(sd))
    out = np.zeros((NC * M, sd))     # row r of the utterance at index r - a00
    a00 = T - NC * M

    def coef(i):
        if i < 0:
            z = np.zeros(sd)
            return z, z, z + 1.0
        return co(i)

    def local_forward(a, s):
        b = rhs_fn(a, M)
        z = np.zeros((M, sd))
        zm1, zm2 = s
        h1 = [np.ones(sd), np.zeros(sd)]
        h2 = [np.zeros(sd), np.ones(sd)]
        g1 = g2 = np.zeros(sd)
        for k in range(M):
            l1, l2, _ = coef(a + k)
            z[k] = b[k] - l1 * zm1 - l2 * zm2
            zm2, zm1 = zm1, z[k]
            h1 = [-l1 * h1[0] - l2 * h1[1], h1[0]]
            h2 = [-l1 * h2[0] - l2 * h2[1], h2[0]]
        return z, (zm1, zm2), (h1[0], h2[0], h1[1], h2[1])

    def local_backward(a, z, t):
        y = np.zeros((M, sd))
        yp1, yp2 = t
        k1 = [np.ones(sd), np.zeros(sd)]
        k2 = [np.zeros(sd), np.ones(sd)]
        for k in range(M - 1, -1, -1):
            i = a + k
            dinv = coef(i)[2]
            l1n = coef(i + 1)[0] if i + 1 < T else np.zeros(sd)
            l2n = coef(i + 2)[1] if i + 2 < T else np.zeros(sd)
            y[k] = dinv * z[k] - l1n * yp1 - l2n * yp2
            yp2, yp1 = yp1, y[k]
            k1 = [-l1n * k1[0] - l2n * k1[1], k1[0]]
            k2 = [-l1n * k2[0] - l2n * k2[1], k2[0]]
        return y, (yp1, yp2), (k1[0], k2[0], k1[1], k2[1])

    def homog_up(a, t):
        """response of the rows a .. a+M-1 to the state t below them"""
        e = np.zeros((M, sd))
        ep1, ep2 = t
        for k in range(M - 1, -1, -1):
            i = a + k
            l1n = coef(i + 1)[0] if i + 1 < T else np.zeros(sd)
            l2n = coef(i + 2)[1] if i + 2 < T else np.zeros(sd)
            e[k] = -l1n * ep1 - l2n * ep2
            ep2, ep1 = ep1, e[k]
        return e

    def sweep(lag):
        """lag=True: the one-step lag with parking; returns False if some chunk found no slot.  lag=False: two sweeps."""
        S = zero2
        parked = []          # (j, y-hat, tl, Bs) of the previous super-step
        zs = {}
        for k in range(K):
            js = [j for j in range(k * W, min(NC, (k + 1) * W))]
            # local forward with the true incoming state (prefix through the super-step's chunks)
            yh, e, B = {}, {}, {}
            s = S
            for j in js:
                a = T - (NC - j) * M
                z, s, _ = local_forward(a, s)
                zs[j] = z
                yh[j], e[j], B[j] = local_backward(a, z, zero2)
            S = s
            # suffixes: the state below chunk j = tl_j + Bs_j T for the state T below the super-step
            tl, Bs = {}, {}
            t, P = zero2, ident
            for j in reversed(js):
                tl[j], Bs[j] = t, P
                t = tuple(x + y for x, y in zip(e[j], mv(B[j], t)))
                P = mm(B[j], P)
            Ehat = t
            if lag:
                for (j, y, tlj, Bsj) in parked:
                    a = T - (NC - j) * M
                    tt = tuple(x + y_ for x, y_ in zip(tlj, mv(Bsj, Ehat)))
                    out[a - a00:a - a00 + M] = y + homog_up(a, tt)
                parked = []
                for j in js:
                    a = T - (NC - j) * M
                    need = k < K - 1 and not amax(Bs[j]) < tol
                    if not need:
                        out[a - a00:a - a00 + M] = yh[j] + homog_up(a, tl[j])
                    elif (k + 1) * W - 1 - j < slots:
                        parked.append((j, yh[j], tl[j], Bs[j]))
                        if stats is not None:
                            stats["parked"] = stats.get("parked", 0) + 1
                    else:
                        return False
        if lag:
            return True
        # two sweeps: the second in reverse order, the state from below carried exactly
        t = zero2
        for k in range(K - 1, -1, -1):
            for j in reversed(range(k * W, min(NC, (k + 1) * W))):
                a = T - (NC - j) * M
                y, t, _ = local_backward(a, zs[j], t)
                out[a - a00:a - a00 + M] = y
        return True

    ok = (lag_ok or K <= 1) and sweep(True)
    if stats is not None:
        stats.setdefault("two_sweep", []).append(not ok)
    if not ok:
        sweep(False)
    return out[-a00:-a00 + T]


def mlpg_const(means, variances, windows, lengths=None, M=16, W=8, tol=TOL, stats=None, slots=None):
    """Forward MLPG for a (B, Tmax, D) batch with global (D,) variances (None: unit). Returns (out, status)."""
    means = np.asarray(means, dtype=np.float64)
    B, Tmax, D = means.shape
    nw = len(windows)
    sd = D // nw
    wc = win_coefs(windows)
    mw = max(max(l, u) for l, u, _ in windows)
    assert mw == 1
    var = np.ones(D) if variances is None else np.asarray(variances, dtype=np.float64)
    tau = (1.0 / var).reshape(nw, sd)
    rows, i_s, kfail = factor_table(tau, wc, mw, max(Tmax, 4))
    if stats is not None:
        stats["i_s"] = i_s
    # the transfer matrix of a steady super-step (M W rows)
    l1, l2 = rows[i_s][0], rows[i_s][1]
    P = (np.ones(sd), np.zeros(sd), np.zeros(sd), np.ones(sd))
    for _ in range(M * W):
        P = (-l1 * P[0] - l2 * P[2], -l1 * P[1] - l2 * P[3], P[0], P[1])
    lag_ok = amax(P) < tol and i_s + 2 < M * W and not (kfail > 0).any()
    out = np.zeros((B, Tmax, sd))
    status = np.zeros((B, sd), dtype=np.int64)
    for b in range(B):
        T = Tmax if lengths is None else int(lengths[b])
        if T <= 0:
            continue
        co = Coefs(rows, i_s, T, tau, wc, mw)
        mu = means[b, :T].reshape(T, nw, sd)
        y = solve_utterance(lambda a, m_: rhs_rows(mu, tau, wc, mw, T, a, m_), T, co, M, W, tol, stats, slots, lag_ok)
        failed = ((kfail > 0) & (kfail - 1 < T - 2)) | co.tail_bad
        out[b, :T] = np.where(failed[None, :], 0.0, y)
        status[b] = np.where(failed, -1, 0)
    return out, status


def mlpg_const_backward(variances, windows, grad_out, lengths=None, M=16, W=8, tol=TOL):
    """grad wrt the means: grad[t, w*sd+d] = tau_w(t) (W_w z)[t], z = P^-1 grad_out[:, d]."""
    grad_out = np.asarray(grad_out, dtype=np.float64)
    B, Tmax, sd = grad_out.shape
    nw = len(windows)
    D = nw * sd
    wc = win_coefs(windows)
    mw = max(max(l, u) for l, u, _ in windows)
    var = np.ones(D) if variances is None else np.asarray(variances, dtype=np.float64)
    tau = (1.0 / var).reshape(nw, sd)
    rows, i_s, _ = factor_table(tau, wc, mw, max(Tmax, 4))
    out = np.zeros((B, Tmax, D))
    for b in range(B):
        T = Tmax if lengths is None else int(lengths[b])
        if T <= 0:
            continue
        co = Coefs(rows, i_s, T, tau, wc, mw)
        go = grad_out[b]

        def rhs(a, m_):
            r = np.zeros((m_, sd))
            lo, hi = max(0, a), min(T, a + m_)
            if hi > lo:
                r[lo - a:hi - a] = go[lo:hi]
            return r
        z = solve_utterance(rhs, T, co, M, W, tol)
        zp = np.zeros((T + 2, sd))
        zp[1:T + 1] = z
        for w in range(nw):
            cm, c0, cp = wc[w]
            for t in range(T):
                out[b, t, w * sd:(w + 1) * sd] = live(w, t, T, mw) * tau[w] * (cm * zp[t] + c0 * zp[t + 1] + cp * zp[t + 2])
    return out
