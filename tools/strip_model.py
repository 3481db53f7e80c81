"""Executable specification (numpy) of the strip MLPG kernel (nnmnkwii_amd/csrc/mlpg_strip.hip).

NOT product code and not the oracle: a model of the kernel's three-level elimination, vectorised
over static dims the way the kernel is vectorised over lanes, used by tests/test_strip_model.py to
pin the algebra against the oracle on the CPU before the HIP transliteration runs on a GPU.

Layout of the kernel this mirrors (std windows, extents <= 1: pentadiagonal P):
  * lane      = static dim d
  * wavefront = chunk of M = 16 consecutive frames
  * workgroup = strip of W consecutive chunks (W * M frames) of one utterance
  * level 1 (per wavefront, registers): assemble the chunk's rows of P and b, eliminate the M-2
    interior frames carrying the two "left spike" columns that couple the chunk to the previous
    chunk's last two frames (its separator), run the elimination on into the chunk's own separator
  * level 2 (per workgroup, LDS): block-tridiagonal system of the strip's W separators (2x2 blocks),
    eliminated sequentially with a left spike block towards the previous strip's last separator
  * level 3 (per utterance, HBM): block-tridiagonal system over the strips' last separators
  * back-substitution in the reverse order.
"""
import numpy as np

M = 16


def assemble_chunk(mean, tau, windows, f0, T):
    """Rows f0 .. f0+M-1 of P (Pd = P[f,f], P1 = P[f+1,f], P2 = P[f+2,f]) and b, plus the coupling
    ca = P[f0, f0-2], cb = P[f0, f0-1], cc = P[f0+1, f0-1] to the previous chunk.
    mean, tau: (T, nw, sd) (tau already zeroed on the dynamic windows' edge frames)."""
    sd = mean.shape[2]
    Pd = np.zeros((M, sd)); P1 = np.zeros((M, sd)); P2 = np.zeros((M, sd)); rhs = np.zeros((M, sd))
    ca = np.zeros(sd); cb = np.zeros(sd); cc = np.zeros(sd)
    for w, (l, u, c) in enumerate(windows):
        cm = c[0] if l else 0.0
        c0 = c[l]
        cp = c[l + 1] if u else 0.0
        for i in range(-1, M + 1):
            t = f0 + i
            if t < 0 or t >= T:
                continue
            ta = tau[t, w]
            tm = ta * mean[t, w]
            if 0 <= i < M:
                Pd[i] += c0 * c0 * ta
                P1[i] += cp * c0 * ta
                rhs[i] += c0 * tm
            if 0 <= i + 1 < M:
                Pd[i + 1] += cp * cp * ta
                rhs[i + 1] += cp * tm
            if 0 <= i - 1 < M:
                Pd[i - 1] += cm * cm * ta
                P1[i - 1] += c0 * cm * ta
                P2[i - 1] += cp * cm * ta
                rhs[i - 1] += cm * tm
            # coupling of this chunk's first two rows to the previous separator
            if i == -1:
                ca += cp * cm * ta          # P[f0, f0-2]   = P2[f0-2]  gets cp*cm*tau[f0-1]
                cb += cp * c0 * ta          # P[f0, f0-1]   = P1[f0-1]  gets cp*c0*tau[f0-1] ...
            if i == 0:
                cb += c0 * cm * ta          #                            ... + c0*cm*tau[f0]
                cc += cp * cm * ta          # P[f0+1, f0-1] = P2[f0-1]  gets cp*cm*tau[f0]
    for i in range(M):
        f = f0 + i
        if f >= T:
            Pd[i] = 1.0
            P1[i] = P2[i] = rhs[i] = 0.0
        else:
            if f + 1 >= T:
                P1[i] = 0.0
            if f + 2 >= T:
                P2[i] = 0.0
    if f0 == 0 or f0 >= T:
        ca[:] = cb[:] = cc[:] = 0.0
    elif f0 + 1 >= T:
        cc[:] = 0.0
    return Pd, P1, P2, rhs, ca, cb, cc


def eliminate_chunk(Pd, P1, P2, rhs, ca, cb, cc):
    """Level 1. Overwrites Pd/P1/P2/rhs rows 0..n-1 with 1/d, l1, l2, g. Returns the Schur data."""
    n = M - 2
    sd = Pd.shape[1]
    z = lambda: np.zeros(sd)
    t00, t01, t11, h0, h1 = z(), z(), z(), z(), z()
    g1, g2, va1, va2, vb1, vb2 = z(), z(), z(), z(), z(), z()
    l1p, l2p, l2pp = z(), z(), z()
    bad = np.zeros(sd, dtype=bool)
    for i in range(n):
        dd = Pd[i].copy()
        bad |= dd <= 0.0
        dinv = 1.0 / dd
        e1, e2 = P1[i].copy(), P2[i].copy()
        l1, l2 = e1 * dinv, e2 * dinv
        Pd[i + 1] -= l1 * e1
        P1[i + 1] -= l2 * e1
        Pd[i + 2] -= l2 * e2
        gi = rhs[i] - l1p * g1 - l2pp * g2
        ba = ca if i == 0 else 0.0
        bb = cb if i == 0 else (cc if i == 1 else 0.0)
        va = ba - l1p * va1 - l2pp * va2
        vb = bb - l1p * vb1 - l2pp * vb2
        wa, wb = va * dinv, vb * dinv
        t00 += wa * va; t01 += wa * vb; t11 += wb * vb
        h0 += wa * gi; h1 += wb * gi
        Pd[i] = dinv; P1[i] = l1; P2[i] = l2; rhs[i] = gi
        g2, g1 = g1, gi
        va2, va1 = va1, va
        vb2, vb1 = vb1, vb
        l2pp, l2p, l1p = l2p, l2, l1
    F1 = rhs[n] - (l1p * g1 + l2pp * g2)
    F2 = rhs[n + 1] - l2p * g1
    L11 = -(l1p * va1 + l2pp * va2); L12 = -(l1p * vb1 + l2pp * vb2)
    L21 = -(l2p * va1); L22 = -(l2p * vb1)
    D11, D12, D22 = Pd[n].copy(), P1[n].copy(), Pd[n + 1].copy()
    return dict(T=(t00, t01, t11), h=(h0, h1), D=(D11, D12, D22), F=(F1, F2), L=(L11, L12, L21, L22), bad=bad)


def backsub_chunk(Pd, P1, P2, rhs, ca, cb, cc, ul, u):
    """Level-1 back-substitution: ul = solution on the previous separator, u = on this chunk's."""
    n = M - 2
    sd = Pd.shape[1]
    a1 = np.zeros(sd); a2 = np.zeros(sd); b1 = np.zeros(sd); b2 = np.zeros(sd)
    q1 = np.zeros(sd); q2 = np.zeros(sd); q3 = np.zeros(sd)
    x = np.zeros((M, sd))
    z = rhs.copy()
    for i in range(n):
        ba = ca if i == 0 else 0.0
        bb = cb if i == 0 else (cc if i == 1 else 0.0)
        va = ba - q1 * a1 - q3 * a2
        vb = bb - q1 * b1 - q3 * b2
        z[i] = rhs[i] - (va * ul[0] + vb * ul[1])
        a2, a1 = a1, va
        b2, b1 = b1, vb
        q3, q2, q1 = q2, P2[i], P1[i]
    x1, x2 = u[0], u[1]
    for i in range(n - 1, -1, -1):
        xi = z[i] * Pd[i] - P1[i] * x1 - P2[i] * x2
        x[i] = xi
        x2, x1 = x1, xi
    x[n] = u[0]
    x[n + 1] = u[1]
    return x


# ---- 2x2 block helpers, every entry an array over dims ----
def sym_inv(E):
    E11, E12, E22 = E
    det = E11 * E22 - E12 * E12
    bad = (E11 <= 0.0) | (det <= 0.0)
    idet = 1.0 / det
    return (E22 * idet, -E12 * idet, E11 * idet), bad


def mat_sym(Lm, S):          # L (2x2 full) @ S (sym)
    L11, L12, L21, L22 = Lm
    S11, S12, S22 = S
    return (L11 * S11 + L12 * S12, L11 * S12 + L12 * S22, L21 * S11 + L22 * S12, L21 * S12 + L22 * S22)


def mat_mat(A, B):
    A11, A12, A21, A22 = A
    B11, B12, B21, B22 = B
    return (A11 * B11 + A12 * B21, A11 * B12 + A12 * B22, A21 * B11 + A22 * B21, A21 * B12 + A22 * B22)


def mat_matT_sym(A, B):      # A @ B^T, result symmetric by construction: (11, 12, 22)
    A11, A12, A21, A22 = A
    B11, B12, B21, B22 = B
    return (A11 * B11 + A12 * B12, A11 * B21 + A12 * B22, A21 * B21 + A22 * B22)


def mat_vec(A, v):
    A11, A12, A21, A22 = A
    return (A11 * v[0] + A12 * v[1], A21 * v[0] + A22 * v[1])


def matT_vec(A, v):
    A11, A12, A21, A22 = A
    return (A11 * v[0] + A21 * v[1], A12 * v[0] + A22 * v[1])


def sym_vec(S, v):
    S11, S12, S22 = S
    return (S11 * v[0] + S12 * v[1], S12 * v[0] + S22 * v[1])


def matT_sym_mat(V, S):      # V^T S V, symmetric (11, 12, 22)
    SV = (S[0] * V[0] + S[1] * V[2], S[0] * V[1] + S[1] * V[3], S[1] * V[0] + S[2] * V[2], S[1] * V[1] + S[2] * V[3])
    return (V[0] * SV[0] + V[2] * SV[2], V[0] * SV[1] + V[2] * SV[3], V[1] * SV[1] + V[3] * SV[3])


def neg(A):
    return tuple(-a for a in A)


def sub(A, B):
    return tuple(a - b for a, b in zip(A, B))


def add(A, B):
    return tuple(a + b for a, b in zip(A, B))


def strip_eliminate(seps, first_strip):
    """Level 2: the W separators of one strip, sequential block elimination with the left spike block.
    seps[j] = level-1 Schur data of chunk j.  Returns (record for level 3, per-separator factor)."""
    W = len(seps)
    sd = seps[0]["D"][0].shape[0]
    zero2, zero3 = (np.zeros(sd),) * 2, (np.zeros(sd),) * 3
    bad = np.zeros(sd, dtype=bool)
    for s in seps:
        bad |= s["bad"]
    # D~_j = D_j - T_{j+1}, F~_j = F_j - h_{j+1} for the separators whose successor chunk is in the strip
    Dt = [s["D"] for s in seps]
    Ft = [s["F"] for s in seps]
    for j in range(W - 1):
        Dt[j] = sub(Dt[j], seps[j + 1]["T"])
        Ft[j] = sub(Ft[j], seps[j + 1]["h"])
    E = Dt[0]
    g = Ft[0]
    V = seps[0]["L"] if not first_strip else (np.zeros(sd),) * 4
    Ts, hs = seps[0]["T"], seps[0]["h"]            # Schur contribution onto the previous strip's last separator
    fac = []
    for j in range(W - 1):
        Einv, b = sym_inv(E)
        bad |= b
        Lm = seps[j + 1]["L"]
        Mn = mat_sym(Lm, Einv)                     # M_{j+1} = L_{j+1} E_j^-1
        EV = (Einv[0] * V[0] + Einv[1] * V[2], Einv[0] * V[1] + Einv[1] * V[3],
              Einv[1] * V[0] + Einv[2] * V[2], Einv[1] * V[1] + Einv[2] * V[3])   # E^-1 V
        Ts = add(Ts, (V[0] * EV[0] + V[2] * EV[2], V[0] * EV[1] + V[2] * EV[3], V[1] * EV[1] + V[3] * EV[3]))
        Eg = sym_vec(Einv, g)
        hs = add(hs, matT_vec(V, Eg))
        fac.append(dict(Einv=Einv, g=g, V=V, Mn=Mn))
        E = sub(Dt[j + 1], mat_matT_sym(Mn, Lm))
        g = sub(Ft[j + 1], mat_vec(Mn, g))
        V = neg(mat_mat(Mn, V))
    rec = dict(E=E, g=g, V=V, T=Ts, h=hs, bad=bad)
    return rec, fac


def strip_backsub(fac, s, u_last):
    """Level-2 back-substitution: s = solution on the previous strip's last separator, u_last = on this
    strip's.  Returns the W separator solutions."""
    W = len(fac) + 1
    us = [None] * W
    us[W - 1] = u_last
    for j in range(W - 2, -1, -1):
        f = fac[j]
        r = sub(f["g"], mat_vec(f["V"], s))
        us[j] = sub(sym_vec(f["Einv"], r), matT_vec(f["Mn"], us[j + 1]))
    return us


def utterance_solve(recs):
    """Level 3: block-tridiagonal system over the strips' last separators (sequential)."""
    R = len(recs)
    sd = recs[0]["E"][0].shape[0]
    bad = np.zeros(sd, dtype=bool)
    for r in recs:
        bad |= r["bad"]
    Es, gs = [], []
    for r in range(R):
        E, g = recs[r]["E"], recs[r]["g"]
        if r + 1 < R:
            E = sub(E, recs[r + 1]["T"])
            g = sub(g, recs[r + 1]["h"])
        Es.append(E)
        gs.append(g)
    Einvs, Ms = [], []
    for r in range(R):
        if r > 0:
            Mn = mat_sym(recs[r]["V"], Einvs[r - 1])
            Es[r] = sub(Es[r], mat_matT_sym(Mn, recs[r]["V"]))
            gs[r] = sub(gs[r], mat_vec(Mn, gs[r - 1]))
            Ms.append(Mn)
        Einv, b = sym_inv(Es[r])
        bad |= b
        Einvs.append(Einv)
    sig = [None] * R
    sig[R - 1] = sym_vec(Einvs[R - 1], gs[R - 1])
    for r in range(R - 2, -1, -1):
        sig[r] = sub(sym_vec(Einvs[r], gs[r]), matT_vec(Ms[r], sig[r + 1]))
    return sig, bad


def utterance_solve_two_sided(recs, r, lo=None, hi=None, edge=False):
    """Level 3 as the kernel runs it in strip r: a top-down elimination of rows lo .. r-1, a bottom-up elimination of
    rows hi .. r+1, and the 2-block system of rows r-1, r in the middle.  Returns (sigma_{r-1}, sigma_r, bad, damp);
    no factor of either sweep is stored.

    lo = 0, hi = R-1: the exact solve.  A narrower window solves the rows lo .. hi with the separators just outside
    it clamped to zero: separator lo-1 (record lo holds strip lo's interior and its coupling V_lo to that separator)
    and, with ``edge``, separator hi+1 (record hi+1 is read for T, h -- strip hi+1's interior -- and its coupling
    V_{hi+1} only).  What that ignores is exactly the terms V_lo u_{lo-1} and V_{hi+1}^T u_{hi+1} on the window's
    first / last row, and they reach rows r-1, r through the transfer matrices of the two eliminations:

        top:     |du_{r-1}| <= prod_{j=lo}^{r-1} |A_j^-1 V_j| |u_{lo-1}|,     du_r one more factor |S_r^-1 V_r|
        bottom:  |du_r| <= |S_r^-1 V_{r+1}^T| prod_{j=r+1}^{hi} |B_j^-1 V_{j+1}^T| |u_{hi+1}|,
                 du_{r-1} one more factor |A_{r-1}^-1 V_r^T|

    (|.| of a 2x2 block bounded by twice its largest entry).  ``damp`` is the larger of the two products per system;
    the kernel accepts the windowed result only if damp is far below the rounding level, else it repeats with the
    full range."""
    R = len(recs)
    lo = 0 if lo is None else lo
    hi = R - 1 if hi is None else hi
    sd = recs[0]["E"][0].shape[0]
    zero2 = (np.zeros(sd), np.zeros(sd))
    bad = np.zeros(sd, dtype=bool)
    for rec in recs:
        bad |= rec["bad"]
    amax = lambda M4: np.max(np.abs(np.stack(M4)), axis=0)   # noqa: E731
    sym_mat = lambda S_, V: (S_[0] * V[0] + S_[1] * V[2], S_[0] * V[1] + S_[1] * V[3],    # noqa: E731
                             S_[1] * V[0] + S_[2] * V[2], S_[1] * V[1] + S_[2] * V[3])   # S V
    sym_matT = lambda S_, V: (S_[0] * V[0] + S_[1] * V[1], S_[0] * V[2] + S_[1] * V[3],   # noqa: E731
                              S_[1] * V[0] + S_[2] * V[1], S_[1] * V[2] + S_[2] * V[3])  # S V^T
    # top-down: row j is finalised when row j+1 is at hand (its T, h and V)
    Ainv, av, Mn = None, None, None
    damp_t = np.ones(sd) if lo > 0 else np.zeros(sd)
    for j in range(lo, r):
        nxt = recs[j + 1]
        A = sub(recs[j]["E"], nxt["T"])
        aa = sub(recs[j]["g"], nxt["h"])
        if j > lo:
            A = sub(A, mat_matT_sym(Mn, recs[j]["V"]))
            aa = sub(aa, mat_vec(Mn, av))
        Ainv, b = sym_inv(A)
        bad |= b
        av = aa
        Mn = mat_sym(nxt["V"], Ainv)            # V_{j+1} A_j^-1
        damp_t = damp_t * 2.0 * amax(sym_mat(Ainv, recs[j]["V"]))
    # bottom-up: Schur complement (S, s) of the rows below onto row j
    S, sv = (np.zeros(sd),) * 3, zero2
    Tn, hn = (np.zeros(sd),) * 3, zero2          # T, h of the row below
    Vn = (np.zeros(sd),) * 4                     # its coupling to row j
    damp_b = np.zeros(sd)
    if edge:
        Tn, hn, Vn = recs[hi + 1]["T"], recs[hi + 1]["h"], recs[hi + 1]["V"]
        damp_b = np.ones(sd)
    for j in range(hi, r, -1):
        B = sub(sub(recs[j]["E"], Tn), S)
        bv = sub(sub(recs[j]["g"], hn), sv)
        Binv, b = sym_inv(B)
        bad |= b
        V = recs[j]["V"]
        damp_b = damp_b * 2.0 * amax(sym_matT(Binv, Vn))
        W = sym_mat(Binv, V)     # Binv V
        S = (V[0] * W[0] + V[2] * W[2], V[0] * W[1] + V[2] * W[3], V[1] * W[1] + V[3] * W[3])   # V^T Binv V
        sv = matT_vec(V, sym_vec(Binv, bv))
        Tn, hn, Vn = recs[j]["T"], recs[j]["h"], V
    # middle: rows r-1 and r
    B = sub(sub(recs[r]["E"], Tn), S)
    bv = sub(sub(recs[r]["g"], hn), sv)
    Vr = recs[r]["V"]
    if r > lo:
        B = sub(B, mat_matT_sym(Mn, Vr))         # Mn = V_r A_{r-1}^-1
        bv = sub(bv, mat_vec(Mn, av))
    Binv, b = sym_inv(B)
    bad |= b
    sig = sym_vec(Binv, bv)
    sprev = zero2
    damp_b = damp_b * 2.0 * amax(sym_matT(Binv, Vn))
    if r > lo:
        sprev = sub(sym_vec(Ainv, av), matT_vec(Mn, sig))   # A^-1 (a - V_r^T sigma_r)
        damp_t = damp_t * np.maximum(1.0, 2.0 * amax(sym_mat(Binv, Vr)))
        damp_b = damp_b * np.maximum(1.0, 2.0 * amax(sym_matT(Ainv, Vr)))
    return sprev, sig, bad, np.maximum(damp_t, damp_b)


def local_window(r, R, k):
    """The records strip r reads first: rows max(0, r-k) .. hi, plus record hi+1 as the clamped edge if ``edge``
    (hi + 1 = r + k when that is not the utterance's last strip; otherwise the rows down to the last one)."""
    lo = max(0, r - k)
    if r + k < R - 1:
        return lo, r + k - 1, True
    return lo, R - 1, False


def mlpg_strip(mean_frames, variance_frames, windows, W=4, T=None, two_sided=True, local_k=0, local_tol=1e-22, stats=None):
    """One utterance through the strip algorithm. (Tmax, D) -> (Tmax, sd); float64."""
    mean_frames = np.asarray(mean_frames, dtype=np.float64)
    Tmax, D = mean_frames.shape
    T = Tmax if T is None else T
    nw = len(windows)
    sd = D // nw
    var = np.asarray(variance_frames, dtype=np.float64)
    if var.ndim == 1:
        var = np.tile(var, (Tmax, 1))
    mean = mean_frames.reshape(Tmax, nw, sd)
    tau = 1.0 / var.reshape(Tmax, nw, sd)
    mw = max(max(l, u) for l, u, _ in windows)
    for w in range(1, nw):
        if mw == 0:
            tau[:, w] = 0.0
        else:
            tau[:mw, w] = 0.0
            tau[T - mw:, w] = 0.0
    stats = [] if stats is None else stats
    out = np.zeros((Tmax, sd))
    if T == 0:
        return out, np.zeros(sd, dtype=bool)
    nchunks = -(-T // M)
    R = -(-nchunks // W)
    chunks = []
    for c in range(R * W):
        a = assemble_chunk(mean, tau, windows, c * M, T)
        s = eliminate_chunk(*a)
        chunks.append((a, s))
    recs, facs = [], []
    for r in range(R):
        rec, fac = strip_eliminate([chunks[r * W + j][1] for j in range(W)], r == 0)
        recs.append(rec)
        facs.append(fac)
    sig, bad = utterance_solve(recs)
    zero2 = (np.zeros(sd), np.zeros(sd))
    for r in range(R):
        s = sig[r - 1] if r > 0 else zero2
        if two_sided:
            s, sr, bad2, _ = utterance_solve_two_sided(recs, r)
            bad = bad | bad2
            if local_k:
                # the kernel's ladder (round 5): local_k / local_tol may be sequences -- the windows tried in turn, each with its
                # own acceptance bound (kernel: 1 strip per side at 2^-66, then 2 at 1e-22); none accepted = the exact solve
                ks = local_k if isinstance(local_k, (tuple, list)) else (local_k,)
                tols = local_tol if isinstance(local_tol, (tuple, list)) else (local_tol,) * len(ks)
                for k_, tol_ in zip(ks, tols):
                    lo_, hi_, edge_ = local_window(r, R, k_)
                    s2, sr2, _, damp = utterance_solve_two_sided(recs, r, lo_, hi_, edge_)
                    rec_ = (float(damp.max()), float(max(np.abs(np.stack(sr2) - np.stack(sr)).max(), np.abs(np.stack(s2) - np.stack(s)).max())))
                    stats.append(rec_ + (k_,) if len(ks) > 1 else rec_)
                    if damp.max() < tol_:
                        s, sr = s2, sr2
                        break
        else:
            sr = sig[r]
        us = strip_backsub(facs[r], s, sr)
        for j in range(W):
            a, _ = chunks[r * W + j]
            ul = us[j - 1] if j > 0 else s
            x = backsub_chunk(*a, ul, us[j])
            f0 = (r * W + j) * M
            hi = min(f0 + M, T)
            if hi > f0:
                out[f0:hi] = x[:hi - f0]
    return out, bad
