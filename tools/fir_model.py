"""Executable specification (numpy) of the unit-variance FIR kernel (csrc/mlpg_fir.hip, float32 tensors: autograd.unit_variance_mlpg,
BASELINE config 3).  NOT product code and not the oracle.

With unit variances P = sum_w W~_w^T W_w is the same for every system (paramgen/_mlpg.py:297-373; autograd/_impl/mlpg.py:108-172 is a dense
float32 GEMM with R = P^-1 [W~_w^T]) and, away from the utterance's ends, Toeplitz; its inverse decays by about a bit per frame.  So
    y = P^-1 b,   b[i] = sum_w sum_t W_w[t, i] m_w[t] mu_w[t]      (m_w: the edge mask of the dynamic windows)
is, to 2^-26 of the largest tap, a FIR filter of 2 H + 1 taps on b in the interior and a small table of rows near the two ends:
    TAP[0]            the interior row:  y[t] = sum_k TAP[0][k] b[t - H + k]
    TAP[1 .. E]       rows t = 0 .. E-1 of P^-1 (columns 0 .. t + H, stored at offset k = s - t + H)
    TAP[E+1 .. 2E]    rows T-1 .. T-E (by distance from the end; the matrix near the end does not depend on T)
The table comes from ONE reference solve of 1 + 2 E one-hot right-hand sides on a system of T_ref frames (the product does this on the
GPU with its own exact kernel); the backward pass applies the same table to grad_out (P^-1 is symmetric) and then W_w."""
import numpy as np

H = 24      # taps per side
E = 24      # rows at either end that have their own taps
T_REF = 160


def window_matrix(l, u, c, T):
    W = np.zeros((T, T))
    for k in range(-l, u + 1):
        for t in range(max(0, -k), min(T, T - k)):
            W[t, t + k] = c[l + k]
    return W


def edge_mask(T, mw):
    m = np.ones(T)
    if mw == 0:
        m[:] = 0
    else:
        m[:mw] = 0
        m[T - mw:] = 0
    return m


def build_taps(windows, T_ref=T_REF):
    """(TAP (1 + 2E, 2H + 1), ok): the table and whether the inverse has decayed below 2^-26 at H taps."""
    mw = max(max(l, u) for l, u, _ in windows)
    Ws = [window_matrix(l, u, np.asarray(c, dtype=np.float64), T_ref) for l, u, c in windows]
    masks = [np.ones(T_ref)] + [edge_mask(T_ref, mw) for _ in windows[1:]]
    P = sum(W.T @ (m[:, None] * W) for W, m in zip(Ws, masks))
    Pi = np.linalg.inv(P)     # (the product: 1 + 2 E one-hot solves with its exact kernel)
    c = T_ref // 2
    TAP = np.zeros((1 + 2 * E, 2 * H + 1))
    TAP[0] = Pi[c, c - H:c + H + 1]
    for t in range(E):
        for k in range(2 * H + 1):
            s = t - H + k
            if 0 <= s < T_ref:
                TAP[1 + t, k] = Pi[t, s]
            sb = (T_ref - 1 - t) - H + k
            if 0 <= sb < T_ref:
                TAP[1 + E + t, k] = Pi[T_ref - 1 - t, sb]
    g0 = abs(Pi[c, c])
    ok = abs(Pi[c, c + H + 1]) <= 2.0 ** -26 * g0 and abs(Pi[c, c - H - 1]) <= 2.0 ** -26 * g0
    # the rows just inside the edge tables must already be the interior row
    ok = ok and np.abs(Pi[E, max(0, E - H):E + H + 1] - Pi[c, c - (E - max(0, E - H)):c + H + 1]).max() <= 2.0 ** -26 * g0
    ok = ok and np.abs(np.abs(Pi[c, c + H + 1:c + 2 * H]).sum()) <= 2.0 ** -24 * g0
    return TAP, bool(ok)


def rhs(means, windows, T):
    """b (T, sd) from frame-major means (T, nw*sd)."""
    nw = len(windows)
    sd = means.shape[1] // nw
    mw = max(max(l, u) for l, u, _ in windows)
    b = np.zeros((T, sd))
    for w, (l, u, c) in enumerate(windows):
        m = np.ones(T) if w == 0 else edge_mask(T, mw)
        mu = means[:T, w * sd:(w + 1) * sd] * m[:, None]
        for k in range(-l, u + 1):              # W_w[t, t + k] = c[l + k]  =>  b[t + k] += c[l + k] mu[t]
            lo, hi = max(0, -k), min(T, T - k)
            b[lo + k:hi + k] += c[l + k] * mu[lo:hi]
    return b


def apply_taps(TAP, b, T):
    """y[t] = sum_k tap(t)[k] b[t - H + k], tap(t) = the interior row, or the row of its end of the utterance."""
    assert T >= 2 * E
    y = np.zeros_like(b[:T])
    bp = np.zeros((T + 2 * H, b.shape[1]))
    bp[H:H + T] = b[:T]
    for t in range(T):
        row = TAP[0]
        if t < E:
            row = TAP[1 + t]
        elif T - 1 - t < E:
            row = TAP[1 + E + (T - 1 - t)]
        y[t] = row @ bp[t:t + 2 * H + 1]
    return y


def forward(means, windows, T):
    TAP, ok = build_taps(windows)
    assert ok
    return apply_taps(TAP, rhs(means, windows, T), T)


def backward(grad_out, windows, T):
    """grad[t, w*sd + d] = m_w[t] sum_k c_w[l + k] z[t + k],  z = P^-1 grad_out."""
    TAP, ok = build_taps(windows)
    assert ok
    z = apply_taps(TAP, grad_out[:T], T)
    nw = len(windows)
    sd = grad_out.shape[1]
    mw = max(max(l, u) for l, u, _ in windows)
    g = np.zeros((T, nw * sd))
    for w, (l, u, c) in enumerate(windows):
        m = np.ones(T) if w == 0 else edge_mask(T, mw)
        for k in range(-l, u + 1):
            lo, hi = max(0, -k), min(T, T - k)
            g[lo:hi, w * sd:(w + 1) * sd] += c[l + k] * z[lo + k:hi + k]
        g[:, w * sd:(w + 1) * sd] *= m[:, None]
    return g
