"""ORACLE (test infrastructure only) -- ARD-RBF kernel, psi-statistics and their gradients in NumPy.

This file is the CPU restatement of the arithmetic that the reference reaches through GPy
(``kern.K / psi0 / psi1 / psi2 / update_gradients_expectations / gradients_Z_expectations /
gradients_qX_expectations / update_gradients_full / gradients_X``), called from
/root/reference/autoreg/inference/vardtc.py:54-66,110, svi_vardtc.py:43-52,88 and
/root/reference/autoreg/layers.py:98-164,1452-1458.

GPy is a third-party dependency that is NOT vendored under /root/reference and is not pinned
there (no setup.py / requirements); the closed forms restated here are GPy's published
``GPy/kern/src/psi_comp/rbf_psi_comp.py`` formulas for ``RBF(ARD=True, inv_l=True)``.  Absolute psi values are therefore "parity unpinned" by the reference
itself; this module pins them against the *definition* (Gaussian expectation of the kernel) by
Gauss-Hermite quadrature in tests/test_oracle_psi.py, and every gradient against central finite
differences.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
Two formulations are kept on purpose:
  * ``*_direct``  -- broadcasting over (N, M, M, Q); the readable definition, tiny shapes only;
  * ``PsiStatsFast`` -- chunked GEMM formulation (GPy's own structure), used as the timed CPU baseline.
"""
import numpy as np

# ----------------------------------------------------------------------------------------------
# kernel on inducing points
# ----------------------------------------------------------------------------------------------

def rbf_K(variance, inv_l, X, X2=None):
    """k(x,x') = variance * exp(-0.5 * sum_q inv_l_q (x_q - x'_q)^2);  inv_l = 1/lengthscale^2."""
    X2 = X if X2 is None else X2
    Xs = X * np.sqrt(inv_l)
    X2s = X2 * np.sqrt(inv_l)
    r2 = (np.square(Xs).sum(1)[:, None] + np.square(X2s).sum(1)[None, :] - 2.0 * Xs.dot(X2s.T))
    r2 = np.maximum(r2, 0.0)
    if X2 is X:
        np.fill_diagonal(r2, 0.0)
    return variance * np.exp(-0.5 * r2)


def rbf_K_exact(variance, inv_l, X, X2=None):
    """Same kernel without the GEMM expansion (no cancellation) -- used for parity references."""
    X2 = X if X2 is None else X2
    diff = X[:, None, :] - X2[None, :, :]
    return variance * np.exp(-0.5 * (np.square(diff) * inv_l).sum(-1))


def rbf_K_grads(variance, inv_l, Z, dL_dK):
    """Gradients of sum(dL_dK * K(Z,Z)) w.r.t. variance, inv_l (Q,), Z (M,Q).

    Mirrors kern.update_gradients_full(dL_dKmm, Z, None) + kern.gradients_X(dL_dKmm, Z)
    (/root/reference/autoreg/layers.py:105,134)."""
    K = rbf_K_exact(variance, inv_l, Z)
    G = dL_dK * K
    diff = Z[:, None, :] - Z[None, :, :]                     # (M,M,Q)
    dvar = G.sum() / variance
    dinv_l = -0.5 * np.einsum('ij,ijq->q', G, np.square(diff))
    Gs = G + G.T
    dZ = -np.einsum('ij,ijq->iq', Gs, diff) * inv_l
    return dvar, dinv_l, dZ


# ----------------------------------------------------------------------------------------------
# psi statistics, direct (definition) form
# ----------------------------------------------------------------------------------------------

def psi0_direct(variance, mu):
    return np.full(mu.shape[0], float(variance))


def psi1_direct(variance, inv_l, Z, mu, S):
    ls2 = 1.0 / inv_l
    logdenom = -0.5 * np.log(S / ls2 + 1.0).sum(1)                       # (N,)
    dist = np.square(mu[:, None, :] - Z[None, :, :]) / (S[:, None, :] + ls2)
    return variance * np.exp(logdenom[:, None] - 0.5 * dist.sum(-1))      # (N,M)


def psi2n_direct(variance, inv_l, Z, mu, S):
    ls2 = 1.0 / inv_l
    logdenom = -0.5 * np.log(2.0 * S / ls2 + 1.0).sum(1)                 # (N,)
    Zdist = -0.25 * (np.square(Z[:, None, :] - Z[None, :, :]) / ls2).sum(-1)   # (M,M)
    Zhat = 0.5 * (Z[:, None, :] + Z[None, :, :])                          # (M,M,Q)
    mudist = (np.square(mu[:, None, None, :] - Zhat[None]) / (2.0 * S[:, None, None, :] + ls2)).sum(-1)
    return variance * variance * np.exp(logdenom[:, None, None] + Zdist[None] - mudist)   # (N,M,M)


def psi2_direct(variance, inv_l, Z, mu, S):
    return psi2n_direct(variance, inv_l, Z, mu, S).sum(0)


def psi_grads_direct(variance, inv_l, Z, mu, S, dL_dpsi0, dL_dpsi1, dL_dpsi2):
    """All gradients through psi0/psi1/psi2 (dL_dpsi2 is (M,M) or (N,M,M)).

    Returns dict(dvar, dinv_l (Q,), dZ (M,Q), dmu (N,Q), dS (N,Q)).
    Restates kern.update_gradients_expectations / gradients_Z_expectations /
    gradients_qX_expectations as used at /root/reference/autoreg/layers.py:98-132,1452-1458."""
    ls2 = 1.0 / inv_l
    ls = np.sqrt(ls2)
    N, Q = mu.shape
    M = Z.shape[0]
    # psi0
    dvar = float(np.sum(dL_dpsi0))
    dl = np.zeros(Q)
    dZ = np.zeros((M, Q))
    dmu = np.zeros((N, Q))
    dS = np.zeros((N, Q))
    # psi1
    if dL_dpsi1 is not None:
        p1 = psi1_direct(variance, inv_l, Z, mu, S)
        Lp = dL_dpsi1 * p1
        e = 1.0 / (S + ls2)                                                # (N,Q)
        diff = mu[:, None, :] - Z[None, :, :]                              # (N,M,Q)
        dvar += Lp.sum() / variance
        dmu += -np.einsum('nm,nq,nmq->nq', Lp, e, diff)
        dS += 0.5 * (np.einsum('nm,nmq->nq', Lp, np.square(diff)) * np.square(e) - Lp.sum(1)[:, None] * e)
        dZ += np.einsum('nm,nq,nmq->mq', Lp, e, diff)
        dl += (np.einsum('nm,nq->q', Lp, S * e) / ls
               + ls * np.einsum('nm,nq,nmq->q', Lp, np.square(e), np.square(diff)))
    # psi2
    if dL_dpsi2 is not None:
        p2n = psi2n_direct(variance, inv_l, Z, mu, S)
        Lp = (dL_dpsi2[None] if dL_dpsi2.ndim == 2 else dL_dpsi2) * p2n     # (N,M,M)
        d = 1.0 / (2.0 * S + ls2)                                          # (N,Q)
        Zhat = 0.5 * (Z[:, None, :] + Z[None, :, :])
        Zdiff = Z[:, None, :] - Z[None, :, :]
        dd = mu[:, None, None, :] - Zhat[None]                             # (N,M,M,Q)
        dvar += 2.0 * Lp.sum() / variance
        dmu += -2.0 * np.einsum('nij,nq,nijq->nq', Lp, d, dd)
        dS += (2.0 * np.einsum('nij,nijq->nq', Lp, np.square(dd)) * np.square(d)
               - Lp.sum((1, 2))[:, None] * d)
        Lps = Lp + Lp.transpose(0, 2, 1)
        dZ += (-0.5 * np.einsum('nij,ijq->iq', Lps, Zdiff) / ls2
               + np.einsum('nij,nq,nijq->iq', Lps, d, dd))
        dl += (2.0 * np.einsum('nij,nq->q', Lp, S * d) / ls
               + 0.5 * np.einsum('nij,ijq->q', Lp, np.square(Zdiff)) / (ls ** 3)
               + 2.0 * ls * np.einsum('nij,nq,nijq->q', Lp, np.square(d), np.square(dd)))
    dinv_l = dl * (-(ls ** 3) / 2.0)       # GPy: inv_l.gradient = lengthscale.gradient * (l^3 / -2)
    return dict(dvar=dvar, dinv_l=dinv_l, dZ=dZ, dmu=dmu, dS=dS)


# ----------------------------------------------------------------------------------------------
# psi statistics, chunked GEMM form (CPU baseline; GPy's own structure: the data-dependent part of
# the psi2 exponent is [N x 2Q] . [2Q x M^2])
# ----------------------------------------------------------------------------------------------

class PsiStatsFast(object):
    def __init__(self, chunk=512):
        self.chunk = chunk

    def psi1(self, variance, inv_l, Z, mu, S):
        ls2 = 1.0 / inv_l
        e = 1.0 / (S + ls2)
        logdenom = -0.5 * np.log(S * inv_l + 1.0).sum(1)
        ex = ((np.square(mu) * e).sum(1)[:, None] - 2.0 * (mu * e).dot(Z.T) + e.dot(np.square(Z).T))
        return variance * np.exp(logdenom[:, None] - 0.5 * ex)

    def _psi2n_chunk(self, variance, inv_l, Z, mu, S, Zdist, Zhat2, ZhatSq2):
        ls2 = 1.0 / inv_l
        M = Z.shape[0]
        d = 1.0 / (2.0 * S + ls2)
        logdenom = -0.5 * np.log(2.0 * S * inv_l + 1.0).sum(1)
        ex = (-(np.square(mu) * d).sum(1)[:, None] + 2.0 * (mu * d).dot(Zhat2.T) - d.dot(ZhatSq2.T))   # (n, M*M)
        ex += logdenom[:, None]
        ex += Zdist.reshape(1, -1)
        np.exp(ex, out=ex)
        ex *= variance * variance
        return ex.reshape(-1, M, M), d

    def _zterms(self, inv_l, Z):
        M, Q = Z.shape
        Zdiff = Z[:, None, :] - Z[None, :, :]
        Zdist = -0.25 * (np.square(Zdiff) * inv_l).sum(-1)
        Zhat = 0.5 * (Z[:, None, :] + Z[None, :, :])
        return Zdist, Zdiff, Zhat.reshape(M * M, Q), np.square(Zhat).reshape(M * M, Q)

    def psi2(self, variance, inv_l, Z, mu, S):
        M = Z.shape[0]
        Zdist, _, Zhat2, ZhatSq2 = self._zterms(inv_l, Z)
        out = np.zeros((M, M))
        for s in range(0, mu.shape[0], self.chunk):
            p2n, _ = self._psi2n_chunk(variance, inv_l, Z, mu[s:s + self.chunk], S[s:s + self.chunk], Zdist, Zhat2, ZhatSq2)
            out += p2n.sum(0)
        return out

    def psi2n(self, variance, inv_l, Z, mu, S):
        Zdist, _, Zhat2, ZhatSq2 = self._zterms(inv_l, Z)
        return self._psi2n_chunk(variance, inv_l, Z, mu, S, Zdist, Zhat2, ZhatSq2)[0]

    def grads(self, variance, inv_l, Z, mu, S, dL_dpsi0, dL_dpsi1, dL_dpsi2):
        """Same outputs as psi_grads_direct, O(chunk*M*M) memory, dL_dpsi2 (M,M)."""
        ls2 = 1.0 / inv_l
        ls = np.sqrt(ls2)
        N, Q = mu.shape
        M = Z.shape[0]
        dvar = float(np.sum(dL_dpsi0))
        dl = np.zeros(Q)
        dZ = np.zeros((M, Q))
        dmu = np.zeros((N, Q))
        dS = np.zeros((N, Q))
        Z2 = np.square(Z)
        if dL_dpsi1 is not None:
            p1 = self.psi1(variance, inv_l, Z, mu, S)
            Lp = dL_dpsi1 * p1                                   # (N,M)
            e = 1.0 / (S + ls2)
            L1 = Lp.sum(1)[:, None]                              # sum_m Lp
            LZ = Lp.dot(Z)                                       # sum_m Lp Z
            LZ2 = Lp.dot(Z2)
            # sum_m Lp (mu - Z)   and   sum_m Lp (mu - Z)^2
            s1 = mu * L1 - LZ
            s2 = np.square(mu) * L1 - 2.0 * mu * LZ + LZ2
            dvar += Lp.sum() / variance
            dmu += -e * s1
            dS += 0.5 * (np.square(e) * s2 - L1 * e)
            # dZ[m,q] = sum_n Lp[n,m] e[n,q] (mu[n,q] - Z[m,q])
            dZ += Lp.T.dot(e * mu) - Z * Lp.T.dot(e)
            dl += (L1 * S * e).sum(0) / ls + ls * (np.square(e) * s2).sum(0)
        if dL_dpsi2 is not None:
            Zdist, Zdiff, Zhat2, ZhatSq2 = self._zterms(inv_l, Z)
            dLs = dL_dpsi2 + dL_dpsi2.T
            Lsum = np.zeros((M, M))
            for s in range(0, N, self.chunk):
                mu_c, S_c = mu[s:s + self.chunk], S[s:s + self.chunk]
                p2n, d = self._psi2n_chunk(variance, inv_l, Z, mu_c, S_c, Zdist, Zhat2, ZhatSq2)
                Lp = p2n * dL_dpsi2[None]                         # (n,M,M)
                Lpf = Lp.reshape(-1, M * M)
                L0 = Lpf.sum(1)[:, None]                          # (n,1)
                LZh = Lpf.dot(Zhat2)                              # sum Lp Zhat       (n,Q)
                LZh2 = Lpf.dot(ZhatSq2)                           # sum Lp Zhat^2     (n,Q)
                s1 = mu_c * L0 - LZh
                s2 = np.square(mu_c) * L0 - 2.0 * mu_c * LZh + LZh2
                dvar += 2.0 * Lp.sum() / variance
                dmu[s:s + self.chunk] += -2.0 * d * s1
                dS[s:s + self.chunk] += 2.0 * np.square(d) * s2 - L0 * d
                Lsum += Lp.sum(0)
                # dZ: sum_n sum_j Lps[n,i,j] d[n,q] (mu[n,q] - Zhat[i,j,q])
                Lps = (p2n * dLs[None])                           # symmetrised weight
                R = Lps.sum(2)                                    # (n,M)
                dZ += R.T.dot(d * mu_c)
                #   - sum_n sum_j Lps[n,i,j] d[n,q] (Z_i + Z_j)/2
                dZ += -0.5 * Z * R.T.dot(d)
                T = np.tensordot(Lps, d, axes=([0], [0]))          # (M,M,Q): sum_n Lps[n,i,j] d[n,q]
                dZ += -0.5 * np.einsum('ijq,jq->iq', T, Z)
                dl += 2.0 * (L0 * S_c * d).sum(0) / ls + 2.0 * ls * (np.square(d) * s2).sum(0)
            Lss = Lsum + Lsum.T
            dZ += -0.5 * np.einsum('ij,ijq->iq', Lss, Zdiff) / ls2
            dl += 0.5 * np.einsum('ij,ijq->q', Lsum, np.square(Zdiff)) / (ls ** 3)
        dinv_l = dl * (-(ls ** 3) / 2.0)
        return dict(dvar=dvar, dinv_l=dinv_l, dZ=dZ, dmu=dmu, dS=dS)


class RBF(object):
    """Minimal object with the GPy kernel protocol the reference's inference classes call
    (K, Kdiag, psi0, psi1, psi2, psi2n) -- /root/reference/autoreg/inference/vardtc.py:59-64,110."""

    def __init__(self, input_dim, variance=1.0, inv_l=None, fast=True, chunk=512):
        self.input_dim = input_dim
        self.variance = np.array([float(variance)])
        self.inv_l = np.ones(input_dim) if inv_l is None else np.array(inv_l, dtype=float).reshape(input_dim)
        self._fast = PsiStatsFast(chunk) if fast else None

    def K(self, X, X2=None):
        return rbf_K_exact(self.variance[0], self.inv_l, X, X2)

    def Kdiag(self, X):
        return np.full(X.shape[0], self.variance[0])

    def psi0(self, Z, q):
        return psi0_direct(self.variance[0], q.mean)

    def psi1(self, Z, q):
        if self._fast is not None:
            return self._fast.psi1(self.variance[0], self.inv_l, Z, q.mean, q.variance)
        return psi1_direct(self.variance[0], self.inv_l, Z, q.mean, q.variance)

    def psi2(self, Z, q):
        if self._fast is not None:
            return self._fast.psi2(self.variance[0], self.inv_l, Z, q.mean, q.variance)
        return psi2_direct(self.variance[0], self.inv_l, Z, q.mean, q.variance)

    def psi2n(self, Z, q):
        return psi2n_direct(self.variance[0], self.inv_l, Z, q.mean, q.variance)

    def psi_grads(self, Z, q, dL_dpsi0, dL_dpsi1, dL_dpsi2):
        if self._fast is not None:
            return self._fast.grads(self.variance[0], self.inv_l, Z, q.mean, q.variance, dL_dpsi0, dL_dpsi1, dL_dpsi2)
        return psi_grads_direct(self.variance[0], self.inv_l, Z, q.mean, q.variance, dL_dpsi0, dL_dpsi1, dL_dpsi2)

    def K_grads(self, Z, dL_dK):
        return rbf_K_grads(self.variance[0], self.inv_l, Z, dL_dK)
