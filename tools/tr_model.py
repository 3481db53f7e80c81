"""Executable specification (numpy) of the strip kernel's TRANSPOSED form (csrc/mlpg_strip_impl.h: strip_kernel<..., TR>,
lane_stream_tr, assemble_eliminate<..., LT>; csrc/mlpg_strip.hip: launch_strip_tr) -- what a LANE is there, not how the strip
kernel solves (that is tools/strip_model.py).

A narrow stream (sd = 1 .. 32 static dims; paramgen/_mlpg.py:92-199 per (utterance, dim)) leaves most of the kernel's 64 lanes idle.
The transposed form lets the lanes of a system group run over u = 64 // sd consecutive utterances x the sd dims:

    lane l of group g  <->  utterance b = g u + l // sd,  dim d = l % sd           (lanes >= (min(u, B - g u)) sd are idle)

and addresses everything from the group's FIRST utterance (the buffer descriptor's base) with a per-lane element offset:

    input  (mean / per-frame variance) of window w, frame t:  base_in(g u) + t ld_in + [ (l // sd) Tmax ld_in + in_col + d + w pitch ]
    output frame t:                                           base_out(g u) + t ld_out + [ (l // sd) Tmax ld_out + out_col + d ]
    status:                                                   (g u) ld_status + [ (l // sd) ld_status + stat_col + d ]
    global (D,) variance of window w:                         var[ in_col + d + w pitch ]        (no utterance offset)

(`pitch` = the stream's window pitch: its static dim, also for a PIECE of a stream that holds only some of its dims.)

With a lengths vector the wavefronts of a group run to the group's longest utterance T_g; lane l's own frames >= T_b enter with
precision 0 and mean 0 (per-lane selects: the padding may hold anything) and its rows >= T_b are identity rows with a zero right-hand
side.  Claim: lane l's solution on frames < T_b is the reference's for utterance b alone, and exactly 0 on T_b .. T_g - 1.  The model
builds the lane's T_g x T_g system that way, solves it densely, and is pinned against the oracle on the CPU (tests/test_tr_model.py).
Test infrastructure only.
"""
import numpy as np


def lane_map(B, Tmax, sd, pitch, ld_in, ld_out, ld_status, in_col, out_col, stat_col):
    """Per group: (b0, lanes) with lanes = list of (lane, b, d, din, dout, dstat, dvar) -- the integers the kernel computes
    (lane_stream_tr), relative to the group's first utterance b0."""
    u = 64 // sd
    groups = []
    for g in range((B + u - 1) // u):
        b0 = g * u
        nd = min(u, B - b0) * sd
        lanes = []
        for lane in range(nd):
            uu, d = lane // sd, lane % sd
            lanes.append((lane, b0 + uu, d,
                          uu * Tmax * ld_in + in_col + d,        # din: window-0 input column, from the group's first utterance
                          uu * Tmax * ld_out + out_col + d,      # dout
                          uu * ld_status + stat_col + d,         # dstat
                          in_col + d))                           # dvar: column of a global variance vector
        groups.append((b0, lanes))
    return groups


def fits(B, Tmax, sd, ld_in, ld_out):
    """launch_strip_tr's condition: every byte offset inside the 2 GB window of the group's first utterance."""
    u = 64 // sd
    return 1 <= sd <= 32 and B >= 2 and float(u) * Tmax * max(ld_in, ld_out) * 8.0 < 2147483647.0


def window_matrix(l, u, coeff, T):
    W = np.zeros((T, T))
    for t in range(T):
        for k in range(-l, u + 1):
            if 0 <= t + k < T:
                W[t, t + k] = coeff[l + k]
    return W


def solve_lane(mean_flat, var_flat, var_mode, base_in, din, dvar, pitch, ld_in, windows, Tg, Tu):
    """The lane's system as the masked assembly builds it for a group that runs to Tg frames: per-lane live frames [0, Tu) for
    the static window, [mw, Tu - mw) for the dynamic ones (_mlpg.py:177,191-193), identity rows beyond Tu."""
    nw = len(windows)
    mw = max(max(l, u) for l, u, _ in windows)
    P = np.zeros((Tg, Tg))
    b = np.zeros(Tg)
    for w, (l, u, coeff) in enumerate(windows):
        W = window_matrix(l, u, np.asarray(coeff, dtype=np.float64), Tg)
        tau = np.zeros(Tg)
        mu = np.zeros(Tg)
        for t in range(Tg):
            lv = t < Tu if w == 0 else (mw != 0 and mw <= t < Tu - mw)
            if not lv:
                continue                                       # a select: the values at dead frames are never used
            idx = base_in + t * ld_in + din + w * pitch
            if var_mode == "frame":
                tau[t] = 1.0 / var_flat[idx]
            elif var_mode == "global":
                tau[t] = 1.0 / var_flat[dvar + w * pitch]
            else:
                tau[t] = 1.0
            mu[t] = mean_flat[idx]
        P += W.T @ (tau[:, None] * W)
        b += W.T @ (tau * mu)
    for t in range(Tu, Tg):                                    # identity rows (fix_row)
        P[t, :] = 0.0
        P[:, t] = 0.0
        P[t, t] = 1.0
        b[t] = 0.0
    return np.linalg.solve(P, b)


def forward(mean, var, windows, lengths=None, in_col=0, sd=None, pitch=None, out_col=0, ld_out=None):
    """The transposed form on a (B, Tmax, ld_in) batch: stream columns [in_col, in_col + nw pitch), of which dims [0, sd) are solved
    (sd < pitch: a piece).  Returns (out (B, Tmax, ld_out) with the stream's trajectory in columns [out_col, out_col + sd), zeros in
    the padding) -- every element fetched through the FLAT arrays with the kernel's offsets."""
    mean = np.ascontiguousarray(mean, dtype=np.float64)
    B, Tmax, ld_in = mean.shape
    nw = len(windows)
    if pitch is None:
        pitch = (ld_in - in_col) // nw if sd is None else sd
    if sd is None:
        sd = pitch
    if ld_out is None:
        ld_out = out_col + sd
    assert fits(B, Tmax, sd, ld_in, ld_out)
    var_mode = "unit" if var is None else ("global" if np.ndim(var) == 1 else "frame")
    mean_flat = mean.reshape(-1)
    var_flat = None if var is None else np.ascontiguousarray(var, dtype=np.float64).reshape(-1)
    L = np.full(B, Tmax, dtype=np.int64) if lengths is None else np.clip(np.asarray(lengths, dtype=np.int64), 0, Tmax)
    out_flat = np.full(B * Tmax * ld_out, np.nan)
    touched = np.zeros(B * Tmax * ld_out, dtype=bool)
    for b0, lanes in lane_map(B, Tmax, sd, pitch, ld_in, ld_out, sd, in_col, out_col, 0):
        Tg = int(max(L[b] for _, b, *_ in lanes))               # the wave reduction over the lanes' lengths
        base_in, base_out = b0 * Tmax * ld_in, b0 * Tmax * ld_out
        for lane, b, d, din, dout, dstat, dvar in lanes:
            y = solve_lane(mean_flat, var_flat, var_mode, base_in, din, dvar, pitch, ld_in, windows, Tg, int(L[b])) if Tg else np.zeros(0)
            for t in range(Tmax):
                idx = base_out + t * ld_out + dout
                assert not touched[idx]
                touched[idx] = True
                out_flat[idx] = y[t] if t < Tg else 0.0         # strips past the group's last live frame are zero-filled
    out = out_flat.reshape(B, Tmax, ld_out)
    return out, touched.reshape(B, Tmax, ld_out)
