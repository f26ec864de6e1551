"""Which side loses digits when K_uu is ill conditioned?  The SVI chain (svi_vardtc.py:70-195 + comp_KL_qU) for M = 600
inducing points in 6 input dimensions, evaluated three ways on the same inputs: the oracle's SciPy solves in fp64, the device
chain (blocked Cholesky + inverse factor + matrix products), and the oracle's own formulas in x87 extended precision (64-bit
mantissa: hand-written Cholesky / triangular solves substituted for SciPy's).  Prints each fp64 result's distance from the
extended one.   python tools/chain_accuracy.py [M] [Q] [inverse-lengthscale factor]"""
import sys
import numpy as np

sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
LD = np.longdouble


def chol_ld(A):
    A = np.array(A, dtype=LD)
    M = A.shape[0]
    L = np.zeros_like(A)
    for j in range(M):
        d = A[j, j] - L[j, :j].dot(L[j, :j])
        L[j, j] = np.sqrt(d)
        if j + 1 < M:
            L[j + 1:, j] = (A[j + 1:, j] - L[j + 1:, :j].dot(L[j, :j])) / L[j, j]
    return L


def fsolve_ld(L, B):
    L = np.asarray(L, dtype=LD); B = np.array(B, dtype=LD)
    X = np.zeros_like(B)
    for i in range(L.shape[0]):
        X[i] = (B[i] - L[i, :i].dot(X[:i])) / L[i, i]
    return X


def bsolve_ld(L, B):
    L = np.asarray(L, dtype=LD); B = np.array(B, dtype=LD)
    X = np.zeros_like(B)
    for i in range(L.shape[0] - 1, -1, -1):
        X[i] = (B[i] - L[i + 1:, i].dot(X[i + 1:])) / L[i, i]
    return X


def main():
    import torch
    from oracle import inference as inf, psi_stats as ps
    from rgp_amd.capi import Handle
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 600
    Q = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    D, N = 3, 26
    rng = np.random.default_rng(M + Q)
    Z = rng.standard_normal((M, Q)) * 1.2
    ils = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0            # < 1: smoother kernel, worse conditioned K_uu
    il, var, noise = ils * (0.3 + rng.random(Q)), 1.1, 0.2
    mu = rng.standard_normal((N, Q)); S = 0.05 + 0.1 * rng.random((N, Q))
    P = ps.PsiStatsFast(16)
    psi1, psi2 = P.psi1(var, il, Z, mu, S), P.psi2(var, il, Z, mu, S)
    Kuu = ps.rbf_K_exact(var, il, Z)
    ev = np.linalg.eigvalsh(Kuu + 1e-6 * np.eye(M))
    print('M = %d, Q = %d: cond(K_uu + 1e-6 I) = %.3g' % (M, Q, ev[-1] / ev[0]))
    Y = rng.standard_normal((N, D))
    qm, W, a = rng.standard_normal((M, D)), 0.2 * rng.standard_normal((M, M)), 0.3
    qS = W.dot(W.T) + a * np.eye(M)
    ratio = 0.5

    def run(ext):
        keep = inf._chol, inf._fsolve, inf._bsolve
        c = (lambda x: np.asarray(x, dtype=LD)) if ext else (lambda x: x)
        if ext:
            inf._chol, inf._fsolve, inf._bsolve = chol_ld, fsolve_ld, bsolve_ld
        try:
            st = inf.Stats(N * var, c(psi1), c(psi2), c(Y))
            logL, g, winv, wvec, mid = inf.svi(c(Kuu), st, noise, c(qm), c(W).dot(c(W).T) + a * np.eye(M) if ext else qS)
            KL, dKLm, dKLv, dKLK = inf.kl_qU(c(qm), c(W).dot(c(W).T) + a * np.eye(M) if ext else qS, mid)
        finally:
            inf._chol, inf._fsolve, inf._bsolve = keep
        Gv = g['dL_dqU_var'] - ratio * dKLv
        return dict(bound=np.asarray(logL - ratio * KL).reshape(1), dL_dKmm=g['dL_dKmm'] - ratio * dKLK, dL_dpsi2=g['dL_dpsi2'],
                    Kinv_mu=wvec, woodbury_inv=winv, gqU_m=g['dL_dqU_mean'] - ratio * dKLm, gqU_W=(Gv + Gv.T).dot(c(W)))

    o64 = run(False)
    ext = run(True)
    H = Handle()
    dev = lambda x: torch.tensor(np.ascontiguousarray(x), dtype=torch.float64, device=H.device)
    o = H.svi_chain(N, dev(Kuu + 1e-6 * np.eye(M)), dev([noise]), dev(qm), dev(W), dev([a]), dev([N * var]), dev(Y.T.dot(psi1)),
                    dev(psi2), dev([(Y ** 2).sum()]), qU_ratio=ratio)
    H.sync()
    # dL/dZ through K_uu (layers.py:105,134) from each dL_dKmm: the sum over m' that cancels
    dZ = {}
    for name, dK in (('oracle', o64['dL_dKmm']), ('device', o['dL_dKmm'].cpu().numpy()), ('extended', ext['dL_dKmm'])):
        Gm = np.asarray(dK, dtype=LD) * Kuu
        Gs = Gm + Gm.T
        dZ[name] = -(il * (Z * Gs.sum(1)[:, None] - Gs.dot(Z)))
    print('%-14s %12s %12s   (largest difference from the extended-precision value / largest entry)' % ('', 'oracle fp64', 'device'))
    for k in ('bound', 'dL_dKmm', 'dL_dpsi2', 'Kinv_mu', 'woodbury_inv', 'gqU_m', 'gqU_W'):
        e = np.asarray(ext[k], dtype=LD)
        sc = np.abs(e).max()
        d = o[k].cpu().numpy().reshape(e.shape)
        print('%-14s %12.2e %12.2e' % (k, float(np.abs(np.asarray(o64[k]) - e).max() / sc), float(np.abs(d - e).max() / sc)))
    sc = np.abs(dZ['extended']).max()
    print('%-14s %12.2e %12.2e' % ('dZ via K_uu', float(np.abs(dZ['oracle'] - dZ['extended']).max() / sc),
                                   float(np.abs(dZ['device'] - dZ['extended']).max() / sc)))


if __name__ == '__main__':
    main()
