"""Generate tests/golden/*.npz from the REFERENCE ITSELF.

Run in the build container only (needs /root/reference):

    bash oracle/build_reference.sh            # scratch build under /tmp/oracle_ref
    PYTHONPATH=/tmp/oracle_ref python tests/golden/make_golden.py

The script imports the unmodified reference package (``nnmnkwii.paramgen``,
``nnmnkwii.autograd``), feeds it seeded inputs and stores inputs + outputs as
data.  The .npz files are what travels to the GPU box; the reference does not.
Case names are ``<family>/<id>/<field>``; ``tests/golden/cases.py`` rebuilds
the (deterministic) inputs that are too large to store.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from cases import WINDOW_SETS, c2_utterance, rand_case  # noqa: E402


def _seed_for(*parts):
    import zlib
    return zlib.crc32("/".join(str(p) for p in parts).encode()) & 0x7FFFFFFF


def main():
    import nnmnkwii
    from nnmnkwii import paramgen as G
    assert "oracle_ref" in nnmnkwii.__file__ or "reference" in nnmnkwii.__file__, nnmnkwii.__file__

    out = {}

    # --- SURVEY 8(c) known-answer vector, regenerated from the reference
    windows = WINDOW_SETS["std3"]
    means = (np.arange(18).reshape(6, 3) % 7) / 10
    var = np.array([0.5, 1.0, 2.0])
    out["kat/means"] = means
    out["kat/var"] = var
    out["kat/y"] = G.mlpg(means, var, windows)
    # stage products through the reference's own helpers (_mlpg.py:175-197)
    from nnmnkwii.paramgen._mlpg import build_poe, build_win_mats
    from nnmnkwii.paramgen._bandmat import linalg as bla
    win_mats = build_win_mats(windows, 6)
    prec = 1.0 / np.tile(var, (6, 1))
    prec[:1, 1:] = 0
    prec[-1:, 1:] = 0
    b, P = build_poe(prec * means, prec, win_mats)
    out["kat/b"] = b
    out["kat/P"] = P.data.copy()
    out["kat/chol"] = bla.cholesky(P, lower=True).data.copy()

    # --- mlpg over window sets x dtypes x lengths, per-frame and global variance
    for wname, windows in WINDOW_SETS.items():
        for dt in ("f32", "f64"):
            for T in (1, 2, 3, 4, 5, 10, 37):
                for sd in (1, 2):
                    m, v, vg = rand_case(wname, dt, T, sd)
                    key = "mlpg/%s-%s-T%d-sd%d" % (wname, dt, T, sd)
                    y = G.mlpg(m, v, windows)
                    yg = G.mlpg(m, vg, windows)
                    assert y.dtype == m.dtype
                    out[key + "/y"] = y
                    out[key + "/yg"] = yg

    # --- BASELINE config 1 (T=100, sd=2) and a slice of config 2 (T=1000, sd=60)
    m, v, vg = rand_case("std3", "f64", 100, 2)
    out["mlpg/c1/y"] = G.mlpg(m, v, WINDOW_SETS["std3"])
    out["mlpg/c1/yg"] = G.mlpg(m, vg, WINDOW_SETS["std3"])
    for b_ in range(2):
        m, v = c2_utterance(b_)
        out["mlpg/c2-utt%d/y" % b_] = G.mlpg(m, v, WINDOW_SETS["std3"])
    # float32 variant of the same utterance 0 (reciprocal evaluated in float32)
    m, v = c2_utterance(0)
    out["mlpg/c2-utt0-f32/y"] = G.mlpg(m.astype(np.float32), v.astype(np.float32), WINDOW_SETS["std3"])

    # --- mlpg_grad (float32 inputs, as autograd.MLPG.backward passes them)
    for wname, windows in WINDOW_SETS.items():
        for T in (3, 5, 10, 50):
            for sd in (1, 3):
                m, v, _ = rand_case(wname, "f32", T, sd, salt=7)
                go = np.random.RandomState(99 + T + sd).randn(T, sd).astype(np.float32)
                key = "grad/%s-T%d-sd%d" % (wname, T, sd)
                out[key + "/g"] = G.mlpg_grad(m, v, windows, go)
                m64, v64, _ = rand_case(wname, "f64", T, sd, salt=7)
                out[key + "/g64"] = G.mlpg_grad(m64, v64, windows, go.astype(np.float64))

    # --- unit_variance_mlpg_matrix
    for wname, windows in WINDOW_SETS.items():
        for T in (3, 5, 10, 50):
            out["uvmat/%s-T%d" % (wname, T)] = G.unit_variance_mlpg_matrix(windows, T)

    # --- autograd forward/backward (torch CPU) at small sizes
    import torch
    from nnmnkwii import autograd as AF
    for wname in ("std3", "wide3"):
        windows = WINDOW_SETS[wname]
        nw = len(windows)
        for (B, T, sd) in ((1, 10, 2), (3, 25, 4)):
            torch.manual_seed(1234)
            means_t = torch.rand(B, T, sd * nw, requires_grad=True)
            target = torch.rand(B, T, sd)
            R = torch.from_numpy(G.unit_variance_mlpg_matrix(windows, T))
            y = AF.unit_variance_mlpg(R, means_t)
            loss = torch.nn.MSELoss()(y, target)
            loss.backward()
            key = "autograd_uv/%s-B%d-T%d-sd%d" % (wname, B, T, sd)
            out[key + "/means"] = means_t.detach().numpy().copy()
            out[key + "/target"] = target.numpy().copy()
            out[key + "/y"] = y.detach().numpy().copy()
            out[key + "/grad"] = means_t.grad.numpy().copy()

            # generic MLPG, one utterance, random variances
            torch.manual_seed(4321)
            m2 = torch.rand(T, sd * nw, requires_grad=True)
            v2 = torch.rand(T, sd * nw) + 0.1
            tg2 = torch.rand(T, sd)
            y2 = AF.mlpg(m2, v2, windows)
            torch.nn.MSELoss()(y2, tg2).backward()
            key = "autograd_mlpg/%s-T%d-sd%d" % (wname, T, sd)
            out[key + "/means"] = m2.detach().numpy().copy()
            out[key + "/vars"] = v2.numpy().copy()
            out[key + "/target"] = tg2.numpy().copy()
            out[key + "/y"] = y2.detach().numpy().copy()
            out[key + "/grad"] = m2.grad.numpy().copy()

    # --- delta_features (preprocessing/generic.py:250-288), tuple windows and plain arrays
    from nnmnkwii.preprocessing import delta_features
    for wname, windows in WINDOW_SETS.items():
        for dt in ("f32", "f64"):
            for T in (5, 12, 40):
                x = np.random.RandomState(_seed_for(wname, dt, T)).randn(T, 3).astype(np.float32 if dt == "f32" else np.float64)
                out["delta/%s-%s-T%d/x" % (wname, dt, T)] = x
                out["delta/%s-%s-T%d/y" % (wname, dt, T)] = delta_features(x, windows)
    x = np.random.RandomState(5).randn(9, 2)
    out["delta/plain/x"] = x
    out["delta/plain/y"] = delta_features(x, [np.array([1.0]), np.array([0.25, 0.5, -1.0, 2.0]), np.array([-0.5, 0.0, 0.5])])

    # --- Merlin-style multi-stream utterance from the reference's own example data (util/files.py:90-115:
    #     mgc 180 | lf0 3 | vuv 1 | bap 3, float32), cropped to 160 frames; per-stream paramgen.mlpg with the
    #     global variance of the file, exactly the loop a user of the reference writes
    from nnmnkwii.util import example_file_data_sources_for_acoustic_model
    _, Ysrc = example_file_data_sources_for_acoustic_model()
    feats = Ysrc.collect_features(Ysrc.collect_files()[0])[100:260].astype(np.float32)
    gvar = feats.var(axis=0).astype(np.float32) + np.float32(1e-3)
    out["merlin/feats"] = feats
    out["merlin/var"] = gvar
    sizes, dyn = [180, 3, 1, 3], [True, True, False, True]
    cols, c0 = [], 0
    for size, d_ in zip(sizes, dyn):
        blk = feats[:, c0:c0 + size]
        cols.append(G.mlpg(blk, gvar[c0:c0 + size], WINDOW_SETS["std3"]) if d_ else blk)
        c0 += size
    out["merlin/y"] = np.concatenate(cols, axis=1)
    # the same with per-frame variances (float64)
    f64 = feats.astype(np.float64)
    v64 = np.tile(gvar.astype(np.float64), (len(f64), 1)) * (1.0 + 0.5 * np.random.RandomState(3).rand(*f64.shape))
    out["merlin/v64"] = v64
    cols, c0 = [], 0
    for size, d_ in zip(sizes, dyn):
        cols.append(G.mlpg(f64[:, c0:c0 + size], v64[:, c0:c0 + size], WINDOW_SETS["std3"]) if d_ else f64[:, c0:c0 + size])
        c0 += size
    out["merlin/y64"] = np.concatenate(cols, axis=1)

    # --- modulation spectrum (preprocessing/modspec.py, autograd/_impl/modspec.py)
    from nnmnkwii.preprocessing import inv_modspec, modspec, modspec_smoothing
    for T, n in ((10, 16), (64, 64), (50, 128), (300, 1024), (1000, 4096)):
        x = np.random.RandomState(_seed_for("modspec", T, n)).rand(T, 3)
        out["modspec/T%d-n%d/x" % (T, n)] = x
        for norm in (None, "ortho"):
            key = "modspec/T%d-n%d/%s" % (T, n, norm or "none")
            ms, ph = modspec(x, n=n, norm=norm, return_phase=True)
            out[key + "/ms"] = ms
            out[key + "/phase"] = ph
            out[key + "/inv"] = inv_modspec(ms, ph, norm=norm)
            for log_domain in (True, False):
                for cutoff in (100, 25, 60):
                    out[key + "/smooth-log%d-c%s" % (log_domain, cutoff)] = modspec_smoothing(
                        x, 200, n=n, norm=norm, cutoff=cutoff, log_domain=log_domain)
    x32 = np.random.RandomState(8).rand(40, 2).astype(np.float32)
    out["modspec/f32/x"] = x32
    out["modspec/f32/ms"] = modspec(x32, n=64)
    out["modspec/f32/smooth"] = modspec_smoothing(x32, 200, n=64, cutoff=30)
    import torch
    from nnmnkwii.autograd import modspec as modspec_t
    for T, n in ((16, 16), (12, 32), (40, 256)):
        for norm in (None, "ortho"):
            torch.manual_seed(T + n)
            y = torch.rand(T, 4, requires_grad=True)
            w = torch.rand(n // 2 + 1, 4)
            ms_t = modspec_t(y, n=n, norm=norm)
            (ms_t * w).sum().backward()
            key = "modspec_grad/T%d-n%d-%s" % (T, n, norm or "none")
            out[key + "/y"] = y.detach().numpy().copy()
            out[key + "/w"] = w.numpy().copy()
            out[key + "/ms"] = ms_t.detach().numpy().copy()
            out[key + "/grad"] = y.grad.numpy().copy()

    # --- error behaviour: negative variance -> LinAlgError text
    m, v, _ = rand_case("std3", "f64", 10, 1)
    v = v.copy()
    v[4, 0] = -1e-3
    try:
        G.mlpg(m, v, WINDOW_SETS["std3"])
        msg = ""
    except np.linalg.LinAlgError as e:
        msg = str(e)
    out["err/negvar-msg"] = np.array(msg)
    out["err/negvar-v"] = v

    path = os.path.join(HERE, "mlpg_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, len(out), "arrays,", os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
