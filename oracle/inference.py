"""ORACLE (test infrastructure only) -- the M x M algebra of the two sparse-GP bounds, in NumPy/SciPy.

Restates (in this repo's own words, function by function):
  * collapsed bound  : /root/reference/autoreg/inference/vardtc.py:88-208  (``VarDTC.inference``)
  * SVI bound        : /root/reference/autoreg/inference/svi_vardtc.py:70-195 (``SVI_VarDTC.inference``)
  * KL(q(U)||p(U))   : /root/reference/autoreg/inference/svi_vardtc.py:197-215 (``comp_KL_qU``)
  * latent terms     : /root/reference/autoreg/variational.py:4-25 (``NormalEntropy`` / ``NormalPrior``)

Pinned: tests/golden/*.npz hold outputs of the reference's own ``autoreg.inference`` /
``autoreg.variational`` modules imported unmodified in the build container
(tests/golden/make_golden.py); tests/test_oracle_golden.py compares every output of the functions
below against them.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import numpy as np
from scipy.linalg import cholesky, solve_triangular

LOG_2PI = np.log(2.0 * np.pi)
JITTER = 1e-6          # vardtc.py:28, svi_vardtc.py:26 (const_jitter)


def _chol(A):
    return cholesky(A, lower=True, check_finite=False)


def _fsolve(L, B):      # L^{-1} B
    return solve_triangular(L, B, lower=True, trans=0, check_finite=False)


def _bsolve(L, B):      # L^{-T} B
    return solve_triangular(L, B, lower=True, trans=1, check_finite=False)


def _sandwich_right(L, X):   # L^{-1} X L^{-T}     (GPy backsub_both_sides(L, X, 'right'))
    return _fsolve(L, _fsolve(L, X).T).T


def _sandwich_left(L, X):    # L^{-T} X L^{-1}     (GPy backsub_both_sides(L, X, 'left'))
    return _bsolve(L, _bsolve(L, X).T).T


class Stats(object):
    """Sufficient statistics of one layer's data (unscaled by beta).

    psi0sum = sum_n psi0_n; psi1 (N,M); psi2 (M,M); Ymean (N,D); Yvar (N,D) or None;"""

    def __init__(self, psi0sum, psi1, psi2, Ymean, Yvar=None):
        self.psi0sum, self.psi1, self.psi2, self.Ymean, self.Yvar = psi0sum, psi1, psi2, Ymean, Yvar


def vardtc(Kuu, st, noise_var):
    """Collapsed (Titsias) bound with uncertain inputs and optionally uncertain outputs.

    Returns (logL, grad dict, woodbury_inv, woodbury_vector)."""
    N, D = st.Ymean.shape
    M = Kuu.shape[0]
    beta = 1.0 / max(float(noise_var), 1e-6)                      # vardtc.py:102
    K = Kuu + JITTER * np.eye(M)                                   # :110-114
    Lm = _chol(K)
    psi1b = st.psi1 * beta
    psi2b = st.psi2 * beta
    psi0b = st.psi0sum * beta
    A = _sandwich_right(Lm, psi2b)                                 # :120
    LL = _chol(np.eye(M) + A)                                      # :124-125
    L = Lm.dot(LL)
    psi1Y = st.Ymean.T.dot(psi1b)                                  # (D,M) :73/:83
    b = _fsolve(L, psi1Y.T).T                                      # (D,M) :130
    v = _bsolve(L, b.T).T                                          # (D,M) :132
    C = b.T.dot(b)
    bbt = np.square(b).sum()
    uncertain_y = st.Yvar is not None
    if uncertain_y:
        r = np.sqrt(st.Yvar.sum(1))                                # Shalf :74
        c = _fsolve(L, (r[:, None] * psi1b).T).T                   # (N,M) :136
        bbt += np.square(c).sum()
        C += c.T.dot(c)
        cP = _bsolve(L, c.T).T                                     # psi1SP :139
        YRY = (np.square(st.Ymean).sum() + st.Yvar.sum()) * beta
    else:
        YRY = np.square(st.Ymean).sum() * beta
    I = np.eye(M)
    R = 0.5 * _sandwich_left(Lm, D * I - _sandwich_left(LL, C + D * I))       # dL_dpsi2R :140-141
    logdetLL = 2.0 * np.log(np.diag(LL)).sum()
    logL = (-0.5 * (D * (N * LOG_2PI - N * np.log(beta) + psi0b - np.trace(A)) + YRY - bbt)
            - 0.5 * D * logdetLL)                                  # :148-149
    dL_dKmm = R - 0.5 * D * _sandwich_left(Lm, A)                  # :155
    winv = _sandwich_left(Lm, I - _sandwich_left(LL, I))           # :161
    dL_dthetaL = (0.5 * (YRY * beta + beta * D * psi0b - N * D * beta)
                  - beta * (R * psi2b).sum() - beta * np.trace(C))  # :168
    g = dict(dL_dKmm=dL_dKmm, dL_dpsi0=np.full(N, -0.5 * D * beta), dL_dpsi2=beta * R,
             dL_dthetaL=dL_dthetaL)
    if uncertain_y:
        g['dL_dpsi1'] = beta * (st.Ymean.dot(v) + r[:, None] * cP)   # :179
        P = _fsolve(L, psi1b.T).T                                   # psi1LmiLLi :202
        g['dL_dYmean'] = -beta * st.Ymean + P.dot(b.T)              # :204
        g['dL_dYvar'] = -0.5 * beta + 0.5 * np.square(P).sum(1)     # (N,) :205
    else:
        g['dL_dpsi1'] = beta * st.Ymean.dot(v)                      # :181
    return logL, g, winv, v.T


def svi(Kuu, st, noise_var, qU_mean, qU_var):
    """Uncollapsed (Hensman) bound with explicit q(U)=N(qU_mean, qU_var) shared by the D outputs.

    Returns (logL, grad dict, woodbury_inv, woodbury_vector, mid) where ``mid`` carries the
    factors KL needs (the reference stashes them in ``self.mid``, svi_vardtc.py:100-104)."""
    N, D = st.Ymean.shape
    M = Kuu.shape[0]
    beta = 1.0 / float(noise_var)                                  # svi_vardtc.py:80
    K = Kuu + JITTER * np.eye(M)
    Lm = _chol(K)
    Ls = _chol(qU_var)                                             # :95
    E = _fsolve(Lm, Ls)                                            # LinvLs :96
    gmu = _fsolve(Lm, qU_mean)                                     # Linvmu (M,D) :97
    psi1b = st.psi1 * beta
    psi1Y = st.Ymean.T.dot(psi1b)                                  # (D,M)
    Ybar = _fsolve(Lm, psi1Y.T).T                                  # psi1YLinvT (D,M) :98
    A = _sandwich_right(Lm, st.psi2 * beta)                        # :107
    B = D * E.dot(E.T) + gmu.dot(gmu.T)                            # LmInvSmuLmInvT :111
    psi0b = st.psi0sum * beta
    if st.Yvar is not None:
        YRY = (np.square(st.Ymean).sum() + st.Yvar.sum()) * beta
    else:
        YRY = np.square(st.Ymean).sum() * beta
    data_fit = (-0.5 * D * psi0b - 0.5 * YRY - 0.5 * (B * A).sum() + 0.5 * D * np.trace(A)
                + (gmu * Ybar.T).sum())
    logL = -0.5 * N * D * LOG_2PI + 0.5 * D * N * np.log(beta) + data_fit      # :121-123
    t1 = _sandwich_left(Lm, B.dot(A))                              # :129
    t2 = gmu.dot(Ybar)                                             # :130
    t3 = 0.5 * _sandwich_left(Lm, -D * A - t2 - t2.T)              # :131
    dL_dKmm = 0.5 * (t1 + t1.T) + t3                               # :133
    dL_dthetaL = -0.5 * D * N * beta - data_fit * beta             # :139
    mA = _sandwich_left(Lm, -A)                                    # :145
    Kinv_mu = _bsolve(Lm, gmu)                                     # (M,D) :153
    I = np.eye(M)
    g = dict(dL_dKmm=dL_dKmm,
             dL_dpsi0=np.full(N, -0.5 * D * beta),
             dL_dpsi1=beta * st.Ymean.dot(Kinv_mu.T),               # :164-167
             dL_dpsi2=0.5 * beta * _sandwich_left(Lm, D * I - B),   # :169
             dL_dthetaL=dL_dthetaL,
             dL_dqU_mean=mA.dot(qU_mean) + _bsolve(Lm, Ybar.T),     # :146
             dL_dqU_var=0.5 * D * mA)                               # :147
    if st.Yvar is not None:
        g['dL_dYmean'] = -beta * st.Ymean + _fsolve(Lm, psi1b.T).T.dot(gmu)     # :192
        g['dL_dYvar'] = np.full((N, D), -0.5 * beta)                # :193
    winv = _sandwich_left(Lm, I - E.dot(E.T))                      # :154
    mid = dict(Lm=Lm, Ls=Ls, E=E, gmu=gmu)
    return logL, g, winv, Kinv_mu, mid


def kl_qU(qU_mean, qU_var, mid):
    """KL(q(U)||p(U)) and its gradients w.r.t. qU_mean, qU_var, Kuu (svi_vardtc.py:197-215)."""
    M, D = qU_mean.shape
    Lm, Ls, E, gmu = mid['Lm'], mid['Ls'], mid['E'], mid['gmu']
    I = np.eye(M)
    Kinv = _bsolve(Lm, _fsolve(Lm, I))
    Sinv = _bsolve(Ls, _fsolve(Ls, I))
    KL = (-0.5 * D * M - D * np.log(np.diag(Ls)).sum() + D * np.log(np.diag(Lm)).sum()
          + 0.5 * D * np.square(E).sum() + 0.5 * np.square(gmu).sum())           # :209
    dKL_dmean = _bsolve(Lm, gmu)                                                  # :211
    dKL_dvar = 0.5 * D * (Kinv - Sinv)                                            # :212
    dKL_dK = 0.5 * D * Kinv - 0.5 * Kinv.dot(qU_mean.dot(qU_mean.T) + D * qU_var).dot(Kinv)   # :213
    return float(KL), dKL_dmean, dKL_dvar, dKL_dK


def normal_entropy(var):
    """Entropy term as the reference returns it: ``comp_value`` is MINUS the entropy
    (variational.py:7-9); the layer adds ``-comp_value`` (layers.py:1478)."""
    return -0.5 * (1.0 + LOG_2PI + np.log(var)).sum()


def normal_entropy_grad_var(var):
    return 1.0 / (2.0 * var)                       # variational.py:12, added to variance.gradient


def normal_prior_kl(mean, var):
    """KL(N(mean,var)||N(0,I)) as ``NormalPrior.comp_value`` (variational.py:16-19)."""
    return 0.5 * (np.square(mean).sum() + (var - np.log(var)).sum()) - 0.5 * mean.size


def normal_prior_grads(mean, var):
    """Increments applied by ``NormalPrior.update_gradients`` (variational.py:21-24)."""
    return -mean, -0.5 * (1.0 - 1.0 / var)
