"""GMM-based feature conversion with MLPG on MI355X.

Host-side mirror of /root/reference/nnmnkwii/baseline/gmm.py:46-247 (``MLPGBase``,
``MLPG``).  The mixture model itself stays with scikit-learn, as in the reference; what
the reference does frame by frame in Python (one ``np.linalg.solve`` per frame and
mixture, gmm.py:111-113,225-229) is ONE launch of ``mlpg_hip_gmm_convert`` over all frames
(the M regression matrices ``S_yx S_xx^-1`` are formed once per model on the host), and the
trajectory generation -- the expensive part -- goes to the HIP MLPG kernels through
:func:`nnmnkwii_amd.paramgen.mlpg`.  ``transform_batch`` converts a whole list of utterances
with one launch (what ``IterativeDTWAligner`` calls every iteration).
"""
import numpy as np
from scipy import linalg as _sla
from sklearn.mixture import GaussianMixture

from .. import _hip
from ..paramgen import mlpg as _mlpg


def _precisions_cholesky_full(covariances):
    """Upper factors U_k with U_k U_k^T = covariances[k]^-1 (what sklearn stores in
    ``precisions_cholesky_`` for covariance_type="full"; gmm.py:8-41 restates sklearn 0.24)."""
    K, F, _ = covariances.shape
    out = np.empty((K, F, F))
    eye = np.eye(F)
    for k in range(K):
        try:
            c = _sla.cholesky(covariances[k], lower=True)
        except _sla.LinAlgError:
            raise ValueError(
                "Fitting the mixture model failed because some components have ill-defined empirical "
                "covariance (for instance caused by singleton or collapsed samples). Try to decrease the "
                "number of components, or increase reg_covar.")
        out[k] = _sla.solve_triangular(c, eye, lower=True).T
    return out


class MLPGBase(object):
    """Frame-wise conversion E[y | x] under a joint source/target GMM (gmm.py:46-120).

    Attributes (same names as the reference): ``num_mixtures, weights, src_means, tgt_means,
    covarXX, covarXY, covarYX, covarYY, px``.
    """

    def __init__(self, gmm, swap=False, diff=False):
        assert gmm.covariance_type == "full"
        half = gmm.means_.shape[1] // 2
        self.num_mixtures = gmm.means_.shape[0]
        self.weights = gmm.weights_
        mu, cov = gmm.means_, gmm.covariances_
        self.src_means, self.tgt_means = mu[:, :half], mu[:, half:]
        self.covarXX, self.covarXY = cov[:, :half, :half], cov[:, :half, half:]
        self.covarYX, self.covarYY = cov[:, half:, :half], cov[:, half:, half:]

        if diff:  # differential GMM: target := target - source (gmm.py:62-66)
            self.tgt_means = self.tgt_means - self.src_means
            self.covarYY = self.covarXX + self.covarYY - self.covarXY - self.covarYX
            self.covarXY = self.covarXY - self.covarXX
            self.covarYX = self.covarXY.transpose(0, 2, 1)
        if swap:  # gmm.py:69-72
            self.src_means, self.tgt_means = self.tgt_means, self.src_means
            self.covarXX, self.covarYY = self.covarYY, self.covarXX
            self.covarXY, self.covarYX = self.covarYX, self.covarXY

        # p(x): marginal source model, used for the mixture posteriors (gmm.py:76-85)
        px = GaussianMixture(n_components=self.num_mixtures, covariance_type="full")
        px.means_, px.covariances_, px.weights_ = self.src_means, self.covarXX, self.weights
        px.precisions_cholesky_ = _precisions_cholesky_full(px.covariances_)
        self.px = px

    def _regression(self):
        """A[m] = S_yx[m] S_xx[m]^-1, (M, Dy, D), formed once per model (host LAPACK, M small systems)."""
        A = getattr(self, "_A", None)
        if A is None:
            # A^T = S_xx^-1 S_xy  (S_xx symmetric)
            A = np.ascontiguousarray(np.linalg.solve(self.covarXX, self.covarYX.transpose(0, 2, 1)).transpose(0, 2, 1))
            self._A = A
        return A

    def _convert(self, src, posterior=None, mix=None):
        """sum_m posterior[n, m] (mu_y[m] + A[m] (x_n - mu_x[m])) for all rows of ``src`` in one GPU launch
        (``mix``: one mixture per row instead of posterior weights).  Returns a float64 ndarray."""
        torch = _hip.torch_mod()
        dev = _hip.require_gpu()
        f64 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(dev)  # noqa: E731
        x = f64(src)
        post = f64(posterior) if posterior is not None else None
        mx = torch.from_numpy(np.ascontiguousarray(mix, dtype=np.int32)).to(dev) if mix is not None else None
        out = _hip.gmm_convert(x, post, mx, f64(self.src_means), f64(self.tgt_means), f64(self._regression()))
        return out.cpu().numpy()

    def _transform_frame(self, src):
        """One frame (D,) -> E[p(y|x)] (gmm.py:97-120)."""
        src = np.asarray(src)
        posterior = self.px.predict_proba(np.atleast_2d(src))             # (1, M)
        return self._convert(np.atleast_2d(src), posterior)[0]

    def transform(self, src):
        if src.ndim != 2:
            return self._transform_frame(src)
        posterior = self.px.predict_proba(src)                            # (T, M): scikit-learn, as the reference
        tgt = np.zeros_like(src)                                          # dtype of src (gmm.py:89)
        tgt[...] = self._convert(src, posterior)
        return tgt

    def transform_batch(self, utterances):
        """``[transform(x) for x in utterances]`` with one posterior evaluation and one GPU launch over all frames."""
        utterances = [np.asarray(u) for u in utterances]
        if not utterances:
            return []
        allx = np.concatenate(utterances, axis=0)
        y = self._convert(allx, self.px.predict_proba(allx))
        out, o = [], 0
        for u in utterances:
            t = np.zeros_like(u)
            t[...] = y[o:o + len(u)]
            out.append(t)
            o += len(u)
        return out


class MLPG(MLPGBase):
    """Maximum-likelihood trajectory conversion (Toda 2007) with a joint GMM (gmm.py:123-247).

    ``transform(src)``: per frame the most likely mixture m_t under p(x); E_t and a diagonal
    approximation D_t of the conditional covariance; then ``paramgen.mlpg(E, D, windows)`` on the
    GPU.  Static-only inputs (feature dim == static dim) take the frame-wise path of ``MLPGBase``.
    """

    def __init__(self, gmm, windows=None, swap=False, diff=False):
        super(MLPG, self).__init__(gmm, swap, diff)
        if windows is None:
            windows = [(0, 0, np.array([1.0])), (1, 1, np.array([-0.5, 0.0, 0.5]))]
        self.windows = windows
        self.static_dim = gmm.means_.shape[-1] // 2 // len(windows)

    def transform(self, src):
        if src.shape[1] == self.static_dim:
            return super(MLPG, self).transform(src)
        mix = self.px.predict(src)                                        # sub-optimum mixture sequence, eq. 37
        E = self._convert(src, mix=mix)                                   # eq. 22 / 40
        dg = lambda a: np.diagonal(a, axis1=1, axis2=2)                   # noqa: E731
        Dm = dg(self.covarYY) - dg(self.covarYX) / dg(self.covarXX) * dg(self.covarXY)   # eq. 23, diagonal approx.
        return _mlpg(E, np.ascontiguousarray(Dm[mix]), self.windows)
