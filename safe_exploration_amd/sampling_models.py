"""Monte-Carlo verification of the ellipsoidal state bounds (reference: sampling_models.py:14-107).

Every propagation step is ONE batched GP evaluation of all n_samples particles on the GPU
(`SimpleGPModel.sample_device`: K* -> variance contraction -> `sr_gp_sample`), the particles never
leave HBM between steps; the containment test is the batched `sr_distance_to_center` kernel.
"""
import numpy as np

from . import _buffers as B
from .utils_ellipsoid import distance_to_center_batch


class MonteCarloSafetyVerification(object):
    """Verify probabilistic state bounds of a GP dynamic system through sampling."""

    def __init__(self, GP):
        self.GP = GP
        self.n_s = getattr(GP, "n_s", None) or GP.n_s_out    # the reference reads GP.n_s (:30); SimpleGPModel has n_s_out
        self.n_u = GP.n_u

    def sample_n_step(self, x0, K, k, n=1, n_samples=1000, eps=None, generator=None, as_tensor=False):
        """Sample from the n-step-ahead distribution of the closed loop u_i = K[i] x_i + k[i].

        x0 (n_s, 1) deterministic start; K (n, n_u, n_s); k (n, n_u)  (sampling_models.py:33-80).
        eps (n, n_samples, n_s): optional standard-normal draws (default: the device generator).
        Returns S (n_samples, n_s) and S_all (n, n_samples, n_s)."""
        n_s, n_u = self.n_s, self.n_u
        assert n > 0, "The time horizon n for the multi-step sampling must be positive!"
        assert np.shape(K) == (n, n_u, n_s), "Required shape of K is ({},{},{})".format(n, n_u, n_s)
        assert np.shape(k) == (n, n_u), "Required shape of k is ({},{})".format(n, n_u)
        K = np.asarray(K, dtype=np.float64)
        k = np.asarray(k, dtype=np.float64)
        x0 = np.asarray(x0, dtype=np.float64).reshape(n_s, 1)
        u0 = K[0].dot(x0) + k[0, :, None]
        inp = np.vstack((x0, u0)).T
        dev = self.GP.device
        S_all = B.empty((n, n_samples, n_s), dev)
        for i in range(n):
            e = None
            if eps is not None:
                e = B.as_dev(eps[i], dev, (n_samples, n_s))
                e = e[None] if i == 0 else e[:, None]
            size = n_samples if i == 0 else 1
            if i + 1 < n:
                S, z = self.GP.sample_device(inp, size, e, generator, K[i + 1], k[i + 1])
                inp = z.reshape(n_samples, n_s + n_u)
            else:
                S = self.GP.sample_device(inp, size, e, generator)
            S_all[i].copy_(S.reshape(n_samples, n_s))
        if as_tensor:
            return S_all[n - 1], S_all
        out = B.to_numpy(S_all)
        return out[n - 1].squeeze(), out

    def inside_ellipsoid_ratio(self, S, Q, p):
        """Ratio of samples inside the ellipsoid of each time step  (sampling_models.py:82-107).

        S (n, n_samples, n_s); Q (n, n_s, n_s); p (n, n_s) -> Ratio (n,), R_bool (n, n_samples)."""
        as_t = B.is_tensor(S)
        dev = S.device if as_t else self.GP.device
        n = np.shape(S)[0]
        inside = distance_to_center_batch(B.as_dev(S, dev), B.as_dev(p, dev, (n, self.n_s)),
                                          B.as_dev(Q, dev, (n, self.n_s, self.n_s))) < 1.0
        ratio = inside.double().mean(dim=1)
        if as_t:
            return ratio, inside
        return B.to_numpy(ratio), inside.cpu().numpy()
