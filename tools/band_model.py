"""Executable blueprint (numpy) of the strip scheme for window extents up to 2 (half-bandwidth q = 2 * extent <= 4).

NOT product code and not the oracle.  The HIP kernels (mlpg_strip_impl.h, mlpg_wave_impl.h) are written for the
pentadiagonal case q = 2: separators of 2 frames, 2 x 2 blocks in levels 2 and 3.  The reference's own tests use 5-tap
windows (tests/test_paramgen.py:22-26), whose P = sum_w W_w^T diag(tau_w) W_w has half-bandwidth 4; those run on the
natural-order kernel only.  This file pins the algebra a q = 4 strip kernel would implement -- the same
three levels with separators of q frames and q x q blocks -- against the oracle on the CPU (tests/test_band_model.py), and
counts what it would have to hold per lane, so that the next round starts from a checked design:

  * chunk of M frames per wavefront = M - q interior frames + a separator of q frames;
  * level 1: assemble rows of P (q + 1 band entries per frame) and b, eliminate the interior carrying q "left spike"
    columns towards the previous separator: per chunk the Schur data D (q x q, symmetric), L (q x q coupling to the
    previous separator), F (q) and what the chunk adds to the PREVIOUS separator's block (T, q x q symmetric; h, q);
  * level 2: the W separators of a strip, block-tridiagonal with q x q blocks, eliminated in order with a spike block
    towards the previous strip's last separator;
  * level 3: the strips' last separators (block-tridiagonal again);
  * back-substitution in reverse.

Per-lane state of level 1 at M = 16: (q + 2) x M doubles = 64 (q = 2: the shipped kernel, 128 VGPRs) or 96 (q = 4: 192
VGPRs -- does not fit beside the rest; M = 12 gives 72 doubles = 144 VGPRs with 8 interior frames per chunk).  Record per
chunk / strip: q(q+1)/2 + q^2 + q + q(q+1)/2 + q = 14 doubles (q = 2, as shipped) or 44 (q = 4).
"""
import numpy as np


def band_of(windows, tau, T):
    """Dense P (T x T) and the map means -> b for one system: tau (T, nw) precisions (already zeroed where the reference
    zeroes them), windows [(l, u, coeff)].  Returns P, Wt (list of W_w^T diag(tau_w)) so that b = sum_w Wt[w] @ mean_w."""
    P = np.zeros((T, T))
    Wt = []
    for w, (l, u, c) in enumerate(windows):
        W = np.zeros((T, T))
        for k in range(-l, u + 1):
            for t in range(max(0, -k), min(T, T - k)):
                W[t, t + k] = c[l + k]
        P += W.T @ (tau[:, w][:, None] * W)
        Wt.append(W.T * tau[:, w][None, :])
    return P, Wt


def record_doubles(q):
    return q * (q + 1) // 2 + q * q + q + q * (q + 1) // 2 + q


def solve_strips(P, b, q, M=16, W=4):
    """x = P^-1 b for an SPD matrix of half-bandwidth <= q by the three-level scheme (one system; the kernel runs one per
    lane).  Rows beyond T are identity rows, as the kernel pads the last chunk.  Returns x (T,)."""
    T = len(b)
    assert M > 2 * q, "a chunk must hold an interior that separates its two separators"
    nC = (T + M - 1) // M
    nS = (nC + W - 1) // W
    nC = nS * W                       # whole strips: chunks of identity rows behind the utterance's end
    Tp = nC * M
    A = np.eye(Tp)
    A[:T, :T] = P
    r = np.zeros(Tp)
    r[:T] = b
    n = M - q
    D = np.zeros((nC, q, q)); L = np.zeros((nC, q, q)); F = np.zeros((nC, q))
    Tb = np.zeros((nC, q, q)); hb = np.zeros((nC, q))          # what chunk c adds to separator c-1
    G = [None] * nC; Va = [None] * nC; Vs = [None] * nC        # interior solves kept for the back-substitution
    # ---- level 1 ----
    for c in range(nC):
        I = slice(c * M, c * M + n)
        S = slice(c * M + n, (c + 1) * M)
        Aii = A[I, I]
        gi = np.linalg.solve(Aii, r[I])
        vs = np.linalg.solve(Aii, A[I, S])                     # towards the chunk's own separator
        D[c] = A[S, S] - A[S, I] @ vs
        F[c] = r[S] - A[S, I] @ gi
        G[c], Vs[c] = gi, vs
        if c > 0:
            Sp = slice(c * M - q, c * M)
            va = np.linalg.solve(Aii, A[I, Sp])                # the q left spikes
            Va[c] = va
            L[c] = -A[S, I] @ va                               # coupling block (rows S_c, columns S_{c-1})
            Tb[c] = A[Sp, I] @ va
            hb[c] = A[Sp, I] @ gi
    for c in range(nC - 1):                                     # the next chunk's contribution lands on this separator
        D[c] -= Tb[c + 1]
        F[c] -= hb[c + 1]
    # ---- level 2: per strip, eliminate separators 0 .. W-2 with a spike towards the previous strip's last separator ----
    E = np.zeros((nS, q, q)); g3 = np.zeros((nS, q)); V3 = np.zeros((nS, q, q))   # level-3 system: last separators
    fac = [None] * nC
    for s in range(nS):
        c0 = s * W
        Dj = D[c0].copy(); Fj = F[c0].copy()
        Sj = L[c0].copy() if s > 0 else np.zeros((q, q))        # spike: coupling of separator j to the previous strip
        for j in range(W - 1):
            c = c0 + j
            Di = np.linalg.inv(Dj)
            Ln = L[c + 1]                                       # couples separator j+1 to j
            K = Ln @ Di
            fac[c] = (Di, Fj.copy(), Sj.copy(), Ln.copy())
            Dj, Fj, Sj = D[c + 1] - K @ Ln.T, F[c + 1] - K @ Fj, -K @ Sj
        E[s], g3[s], V3[s] = Dj, Fj, Sj
        if s > 0:
            Tsum = np.zeros((q, q)); hsum = np.zeros(q)
            for j in range(W - 1):
                Di, Fj_, Sj_, _ = fac[c0 + j]
                Tsum += Sj_.T @ Di @ Sj_
                hsum += Sj_.T @ Di @ Fj_
            E[s - 1] -= Tsum
            g3[s - 1] -= hsum
    # ---- level 3: block-tridiagonal over the strips' last separators (block Thomas) ----
    u3 = np.zeros((nS, q))
    Dm = [None] * nS; Fm = [None] * nS
    for s in range(nS):
        Ds, Fs = E[s].copy(), g3[s].copy()
        if s > 0:
            K = V3[s] @ np.linalg.inv(Dm[s - 1])
            Ds -= K @ V3[s].T
            Fs -= K @ Fm[s - 1]
        Dm[s], Fm[s] = Ds, Fs
    for s in range(nS - 1, -1, -1):
        rhs_ = Fm[s].copy()
        if s + 1 < nS:
            rhs_ -= V3[s + 1].T @ u3[s + 1]
        u3[s] = np.linalg.solve(Dm[s], rhs_)
    # ---- back-substitution: level 2, then level 1 ----
    x = np.zeros(Tp)
    usep = np.zeros((nC, q))
    for s in range(nS):
        c0 = s * W
        usep[c0 + W - 1] = u3[s]
        uprev = u3[s - 1] if s > 0 else np.zeros(q)
        for j in range(W - 2, -1, -1):
            Di, Fj_, Sj_, Ln = fac[c0 + j]
            usep[c0 + j] = Di @ (Fj_ - Sj_ @ uprev - Ln.T @ usep[c0 + j + 1])
    for c in range(nC):
        I = slice(c * M, c * M + n)
        S = slice(c * M + n, (c + 1) * M)
        xi = G[c] - Vs[c] @ usep[c]
        if c > 0:
            xi = xi - Va[c] @ usep[c - 1]
        x[I] = xi
        x[S] = usep[c]
    return x[:T]


def mlpg_model(mean_frames, variance_frames, windows, M=16, W=4):
    """paramgen.mlpg through solve_strips, one static dim at a time (numpy, float64)."""
    mean_frames = np.asarray(mean_frames, dtype=np.float64)
    variance_frames = np.asarray(variance_frames, dtype=np.float64)
    T, Dm = mean_frames.shape
    nw = len(windows)
    sd = Dm // nw
    mw = max(max(l, u) for l, u, _ in windows)
    q = 2 * mw
    out = np.zeros((T, sd))
    for d in range(sd):
        tau = 1.0 / variance_frames[:, d::sd][:, :nw]
        if mw > 0:
            tau[:mw, 1:] = 0.0
            tau[-mw:, 1:] = 0.0                               # (as the reference's precisions[-mw:] = 0)
        else:
            tau[:, 1:] = 0.0                                  # python's "-0:" slice zeroes the whole column
        P, Wt = band_of(windows, tau, T)
        b = sum(Wt[w] @ mean_frames[:, w * sd + d] for w in range(nw))
        out[:, d] = solve_strips(P, b, max(q, 1), M=M, W=W) if T > 0 else 0.0
    return out
