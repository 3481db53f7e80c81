"""GMM-based feature conversion with MLPG on MI355X.

Host-side mirror of /root/reference/nnmnkwii/baseline/gmm.py:46-247 (``MLPGBase``,
``MLPG``).  The mixture model itself stays with scikit-learn, as in the reference; what
the reference does frame by frame in Python (one ``np.linalg.solve`` per frame and
mixture, gmm.py:111-113,225-229) is evaluated here for the whole utterance at once with
stacked LAPACK solves, and the trajectory generation -- the expensive part -- goes to the
HIP MLPG kernels through :func:`nnmnkwii_amd.paramgen.mlpg`.
"""
import numpy as np
from scipy import linalg as _sla
from sklearn.mixture import GaussianMixture

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

    def _conditional_means(self, src, mix=None):
        """E_m[y | x_t] = mu_y[m] + S_yx[m] S_xx[m]^-1 (x_t - mu_x[m]).

        ``mix`` None: all mixtures -> (T, M, D); else per-frame mixture indices -> (T, D).
        One stacked LAPACK solve replaces the reference's Python double loop.
        """
        src = np.asarray(src, dtype=np.float64)
        if mix is None:
            dev = src[:, None, :] - self.src_means[None]                   # (T, M, D)
            z = np.linalg.solve(self.covarXX[None], dev[..., None])        # (T, M, D, 1)
            return self.tgt_means[None] + np.matmul(self.covarYX[None], z)[..., 0]
        dev = src - self.src_means[mix]
        z = np.linalg.solve(self.covarXX[mix], dev[..., None])
        return self.tgt_means[mix] + np.matmul(self.covarYX[mix], z)[..., 0]

    def _transform_frame(self, src):
        """One frame (D,) -> E[p(y|x)] (gmm.py:97-120)."""
        src = np.asarray(src)
        E = self._conditional_means(src[None])[0]                         # (M, D)
        posterior = self.px.predict_proba(np.atleast_2d(src))             # (1, M)
        return posterior.dot(E).flatten()

    def transform(self, src):
        if src.ndim != 2:
            return self._transform_frame(src)
        E = self._conditional_means(src)                                  # (T, M, D)
        posterior = self.px.predict_proba(src)                            # (T, M)
        tgt = np.zeros_like(src)                                          # dtype of src (gmm.py:89)
        tgt[...] = np.einsum("tm,tmd->td", posterior, E)
        return tgt


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
        E = self._conditional_means(src, mix)                             # eq. 22 / 40
        dg = lambda a: np.diagonal(a, axis1=1, axis2=2)                   # noqa: E731
        Dm = dg(self.covarYY) - dg(self.covarYX) / dg(self.covarXX) * dg(self.covarXY)   # eq. 23, diagonal approx.
        return _mlpg(E, np.ascontiguousarray(Dm[mix]), self.windows)
