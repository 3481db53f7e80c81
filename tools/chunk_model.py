"""Executable specification (numpy) of the chunked MLPG kernel for window extents up to 2 (csrc/mlpg_chunk_impl.h,
MLPG_HIP_ALGO_CHUNK): P = sum_w W_w^T diag(tau_w) W_w has half-bandwidth q = 2 * extent <= 4.

NOT product code and not the oracle.  Three passes, none of which waits for another workgroup:

  pass 1  one wavefront per (utterance, dim group, chunk), lane = static dim.  A chunk is I interior frames followed by a
          separator of q frames.  Its rows of P (lower band, q + 1 entries per row) are eliminated in natural order with
          the INTERIOR rows as the only pivots; the q columns that couple the first interior rows to the previous chunk's
          separator ride along as right-hand sides.  Nothing is kept but the chunk's record (q = 4: 44 doubles):
            S_LL (q x q, symmetric) and g_L (q)   what the chunk adds to the PREVIOUS separator's block / right-hand side,
            S_RL (q x q)                           the coupling of its own separator to the previous one,
            S_RR (q x q, symmetric) and g_R (q)    its own separator's block / right-hand side after the elimination.
  pass 2  one wavefront per (utterance, dim group): the separators form a block-tridiagonal system with q x q blocks;
          forward elimination over the chunks, back-substitution: the separators' solutions.
  pass 3  as pass 1, with the neighbouring separators' solutions known: the interior rows are eliminated AGAIN (the
          factor rows of a chunk stay in registers this time) and back-substituted; the trajectory is stored.

The price of not waiting is reading the inputs twice; the chunk length is what pass 3's registers hold.
Rows behind the utterance's end are identity rows (the last chunk is padded)."""
import numpy as np


def lower_band(P, q):
    """A[i, k] = P[i, i-k], k = 0..q (zero where i-k < 0)."""
    T = P.shape[0]
    A = np.zeros((T, q + 1))
    for k in range(q + 1):
        A[k:, k] = np.diagonal(P, -k)
    return A


def _chunk_forward(A, b, c, I, q, xl=None):
    """The streaming elimination of chunk c (rows c*C .. c*C+C-1), pivots = interior rows only.
    xl None: pass 1 (the q left-coupling columns ride along, returns the record);
    xl given: pass 3 (the left coupling is folded into the right-hand side, returns the factor rows and u)."""
    C = I + q
    Lm = np.zeros((C, q + 1))       # Lm[r, m] = multiplier of row r for column r-m (an interior pivot), else 0
    dd = np.zeros(C)                # pivot of interior row r, 0 for separator rows (they are no pivots)
    ub = np.zeros(C)                # forward-substituted right-hand side
    us = np.zeros((C, q))           # forward-substituted left-coupling columns (pass 1)
    srr = np.zeros((q, q))
    for r in range(C):
        i = c * C + r
        row = np.zeros(q + 1)       # what will be this row's multipliers / Schur entries
        yb = b[i]
        ys = np.zeros(q)
        for k in range(q, 0, -1):   # column r-k
            if r - k < 0:           # a column of the previous separator: coupling, not a pivot
                s = q + (r - k)
                if xl is None:
                    ys[s] = A[i, k]
                else:
                    yb -= A[i, k] * xl[s]
                continue
            num = A[i, k]
            for m in range(k + 1, q + 1):
                if r - m >= 0:
                    num -= row[m] * dd[r - m] * Lm[r - k, m - k]
            if r - k < I:
                row[k] = num / dd[r - k]
            else:                   # separator row against an earlier separator row: a Schur entry
                srr[r - I, r - k - I] = num
                row[k] = 0.0
        diag = A[i, 0]
        for m in range(1, q + 1):
            if r - m >= 0:
                diag -= row[m] * row[m] * dd[r - m]
        # forward substitution (interior pivots only: row[m] is zero for the others)
        for m in range(1, q + 1):
            if r - m >= 0:
                yb -= row[m] * ub[r - m]
                ys -= row[m] * us[r - m]
        Lm[r] = row
        ub[r] = yb
        us[r] = ys
        if r < I:
            if not diag > 0:
                raise np.linalg.LinAlgError("pivot")
            dd[r] = diag
        else:
            srr[r - I, r - I] = diag
    if xl is not None:
        return Lm, dd, ub
    sll = np.zeros((q, q))
    gl = np.zeros(q)
    for r in range(I):
        sll -= np.outer(us[r], us[r]) / dd[r]
        gl -= us[r] * ub[r] / dd[r]
    srl = us[I:].copy()             # [own separator row][previous separator column]
    gr = ub[I:].copy()
    srr = srr + np.tril(srr, -1).T
    return sll, gl, srl, srr, gr


def solve_chunks(P, b, q, I=16):
    """x = P^-1 b for an SPD matrix of half-bandwidth <= q by the three passes (one system; the kernel runs one per lane)."""
    T = len(b)
    assert I >= q
    C = I + q
    K = (T + C - 1) // C
    Tp = K * C
    Ap = np.zeros((Tp, q + 1))
    Ap[:, 0] = 1.0                              # identity rows behind the end
    Ap[:T] = lower_band(P, q)
    bp = np.zeros(Tp)
    bp[:T] = b
    rec = [_chunk_forward(Ap, bp, c, I, q) for c in range(K)]                # pass 1
    # pass 2: block tridiagonal over the separators
    Dk = [rec[k][3] + (rec[k + 1][0] if k + 1 < K else 0.0) for k in range(K)]
    rk = [rec[k][4] + (rec[k + 1][1] if k + 1 < K else 0.0) for k in range(K)]
    Dp, rp = [], []
    for k in range(K):
        D, r = Dk[k].copy(), rk[k].copy()
        if k:
            E = rec[k][2]                       # rows: separator k, columns: separator k-1
            G = np.linalg.solve(Dp[k - 1], E.T).T
            D = D - G @ E.T
            r = r - G @ rp[k - 1]
        np.linalg.cholesky(D)                   # (raises on a non-positive pivot)
        Dp.append(D)
        rp.append(r)
    xs = [None] * K
    for k in range(K - 1, -1, -1):
        r = rp[k].copy()
        if k + 1 < K:
            r = r - rec[k + 1][2].T @ xs[k + 1]
        xs[k] = np.linalg.solve(Dp[k], r)
    # pass 3
    x = np.zeros(Tp)
    for c in range(K):
        xl = xs[c - 1] if c else np.zeros(q)
        Lm, dd, ub = _chunk_forward(Ap, bp, c, I, q, xl=xl)
        xc = np.zeros(C)
        xc[I:] = xs[c]
        for r in range(I - 1, -1, -1):
            v = ub[r] / dd[r]
            for m in range(1, q + 1):
                if r + m < C:
                    v -= Lm[r + m, m] * xc[r + m]
            xc[r] = v
        x[c * C:(c + 1) * C] = xc
    return x[:T]


def mlpg_model(mean_frames, variance_frames, windows, I=16):
    """paramgen.mlpg through the chunk scheme (float64), for the CPU tests."""
    from band_model import band_of
    mean_frames = np.asarray(mean_frames, dtype=np.float64)
    T, D = mean_frames.shape
    nw = len(windows)
    sd = D // nw
    var = np.asarray(variance_frames, dtype=np.float64)
    if var.ndim == 1:
        var = np.tile(var, (T, 1))
    mw = max(max(l, u) for l, u, _ in windows)
    q = 2 * mw
    out = np.zeros((T, sd))
    for d in range(sd):
        tau = np.zeros((T, nw))
        for w in range(nw):
            tau[:, w] = 1.0 / var[:, w * sd + d]
            if w and mw:
                tau[:mw, w] = 0
                tau[T - mw:, w] = 0
            elif w:
                tau[:, w] = 0
        P, Wt = band_of(windows, tau, T)
        b = sum(Wt[w] @ mean_frames[:, w * sd + d] for w in range(nw))
        out[:, d] = solve_chunks(P, b, max(q, 1), I=I) if T > 0 else 0
    return out
