"""Executable specification (numpy) of the WALK form of the strip MLPG kernel (round 6; csrc/mlpg_walk_impl.h).

NOT product code and not the oracle.  Same three-level substructured LDL^T as tools/strip_model.py -- levels 1 and 2 are
taken from there unchanged -- but ONE workgroup walks the strips of an utterance in order, so level 3 needs no exchange
between workgroups:

  * the top-down elimination over the strips' last separators is CARRIED exactly from strip to strip
    (row j is finalised when record j + 1 is at hand:  A_j = E_j - T_{j+1} - M_j V_j^T,  a_j = g_j - h_{j+1} - M_j a_{j-1},
    M_{j+1} = V_{j+1} A_j^-1);
  * strip r is finished one step late, when record r + 1 exists: sigma_r = A_r^-1 a_r with the separator r + 1 clamped to
    zero -- exactly strip_model's window (lo = 0, hi = r, edge = True) -- and sigma_{r-1} = A_{r-1}^-1 (a_{r-1} - V_r^T sigma_r);
    what the clamp ignores is V_{r+1}^T u_{r+1}, which reaches rows r, r - 1 through A_r^-1 V_{r+1}^T (and A_{r-1}^-1 V_r^T):
    the result is accepted if  2 max|A_r^-1 V_{r+1}^T| * max(1, 2 max|A_{r-1}^-1 V_r^T|)  is below the tolerance (2^-66, the
    strip kernel's 3-strip bound), else the utterance is REJECTED (the caller runs the strip kernel's general route on it);
  * the last strip is exact.

mlpg_walk returns (trajectory, bad pivots, accepted).
"""
import numpy as np

try:
    from tools import strip_model as SM
except ImportError:  # run from tools/
    import strip_model as SM

M = SM.M
TOL = 2.0 ** -66


def _amax(M4):
    return np.max(np.abs(np.stack(M4)), axis=0)


def _sym_matT(S_, V):      # S V^T
    return (S_[0] * V[0] + S_[1] * V[1], S_[0] * V[2] + S_[1] * V[3], S_[1] * V[0] + S_[2] * V[1], S_[1] * V[2] + S_[2] * V[3])


def mlpg_walk(mean_frames, variance_frames, windows, W=4, T=None, tol=TOL, stats=None):
    """One utterance through the walk form. (Tmax, D) -> ((Tmax, sd), bad (sd,), accepted bool)."""
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
    out = np.zeros((Tmax, sd))
    bad = np.zeros(sd, dtype=bool)
    if T == 0:
        return out, bad, True
    nchunks = -(-T // M)
    R = -(-nchunks // W)
    zero2 = (np.zeros(sd), np.zeros(sd))
    accepted = True

    def level12(r):
        chunks = []
        for j in range(W):
            a = SM.assemble_chunk(mean, tau, windows, (r * W + j) * M, T)
            s = SM.eliminate_chunk(*a)
            chunks.append((a, s))
        rec, fac = SM.strip_eliminate([c[1] for c in chunks], r == 0)
        return chunks, rec, fac

    def finish(r, chunks, fac, s, sr):
        us = SM.strip_backsub(fac, s, sr)
        for j in range(W):
            a, _ = chunks[j]
            ul = us[j - 1] if j > 0 else s
            x = SM.backsub_chunk(*a, ul, us[j])
            f0 = (r * W + j) * M
            hi = min(f0 + M, T)
            if hi > f0:
                out[f0:hi] = x[:hi - f0]

    # carried state of the top-down elimination: A_{j}^-1, a_j of the last finalised row, M_{j+1} = V_{j+1} A_j^-1
    Ainv_p = av_p = None        # row r - 2 (finalised two steps ago), kept for sigma_{r-2}'s correction -- see below
    held = None                 # (r, chunks, rec, fac) of the strip that waits for its successor's record
    Mn = None                   # M_r for the held strip r (None for r = 0)
    Ainv_pp = av_pp = None
    for r in range(R):
        chunks, rec, fac = level12(r)
        bad |= rec["bad"]
        if held is not None:
            rh, ch, rech, fach = held
            # finalise row rh now that record rh + 1 (= rec) is at hand
            A = SM.sub(rech["E"], rec["T"])
            aa = SM.sub(rech["g"], rec["h"])
            if rh > 0:
                A = SM.sub(A, SM.mat_matT_sym(Mn, rech["V"]))
                aa = SM.sub(aa, SM.mat_vec(Mn, av_p))
            Ainv, b = SM.sym_inv(A)
            bad |= b
            # windowed solution on separator rh (separator rh + 1 clamped) and the bound on what the clamp ignores
            sig = SM.sym_vec(Ainv, aa)
            damp = 2.0 * _amax(_sym_matT(Ainv, rec["V"]))
            sprev = zero2
            if rh > 0:
                sprev = SM.sub(SM.sym_vec(Ainv_p, av_p), SM.matT_vec(Mn, sig))
                damp = damp * np.maximum(1.0, 2.0 * _amax(_sym_matT(Ainv_p, rech["V"])))
            if stats is not None:
                stats.append(float(damp.max()))
            if not (damp.max() < tol):
                accepted = False
            finish(rh, ch, fach, sprev, sig)
            Mn = SM.mat_sym(rec["V"], Ainv)          # M_{rh+1} = V_{rh+1} A_rh^-1
            Ainv_p, av_p = Ainv, aa
        held = (r, chunks, rec, fac)
    # the last strip: exact
    rh, ch, rech, fach = held
    A, aa = rech["E"], rech["g"]
    if rh > 0:
        A = SM.sub(A, SM.mat_matT_sym(Mn, rech["V"]))
        aa = SM.sub(aa, SM.mat_vec(Mn, av_p))
    Ainv, b = SM.sym_inv(A)
    bad |= b
    sig = SM.sym_vec(Ainv, aa)
    sprev = zero2
    if rh > 0:
        sprev = SM.sub(SM.sym_vec(Ainv_p, av_p), SM.matT_vec(Mn, sig))
    finish(rh, ch, fach, sprev, sig)
    return out, bad, accepted
