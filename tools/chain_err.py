"""Relative errors of the device SVI chain against the reference's golden fixtures: python tools/chain_err.py"""
import glob, os, sys, numpy as np, torch
sys.path.insert(0, '.')
from rgp_amd.capi import Handle
H = Handle()
dev = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64), device=H.device)
rel = lambda a, b: float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))
for path in sorted(glob.glob('tests/golden/inference_*.npz')):
    g = np.load(path)
    M, D = g['in_qU_mean'].shape
    N = g['in_mu'].shape[0]
    ratio = 0.37
    yry = (g['in_Ymean'] ** 2).sum() + (g['in_Yvar'].sum() if 'in_Yvar' in g.files else 0.0)
    K = g['Kuu'] + 1e-6 * np.eye(M)
    o = H.svi_chain(N, dev(K), dev([float(g['in_noise'])]), dev(g['in_qU_mean']),
                    dev(g['in_qU_W']), dev([float(g['in_qU_a'])]), dev([g['psi0'].sum()]),
                    dev(g['in_Ymean'].T.dot(g['psi1'])), dev(g['psi2']), dev([yry]), qU_ratio=ratio)
    H.sync()
    c = lambda k: o[k].cpu().numpy()
    Gv = g['svi_dL_dqU_var'] - ratio * g['kl_dvar']
    print(os.path.basename(path), 'M', M, 'D', D, 'N', N, 'cond(K) %.2e' % np.linalg.cond(K))
    print('  bound', abs(c('bound')[0] - (g['svi_logL'] - ratio * g['kl'])) / abs(g['svi_logL']))
    print('  dKmm', rel(c('dL_dKmm'), g['svi_dL_dKmm'] - ratio * g['kl_dKuu']), 'parts:', np.abs(g['svi_dL_dKmm']).max(), np.abs(g['kl_dKuu']).max())
    print('  dpsi2', rel(c('dL_dpsi2'), g['svi_dL_dpsi2']))
    print('  winv', rel(c('woodbury_inv'), g['svi_woodbury_inv']))
    print('  wvec', rel(c('Kinv_mu'), g['svi_woodbury_vector']))
    print('  dthetaL', rel(c('dL_dthetaL'), g['svi_dL_dthetaL'].reshape(1)))
    print('  gqU_m', rel(c('gqU_m'), g['svi_dL_dqU_mean'] - ratio * g['kl_dmean']))
    print('  gqU_W', rel(c('gqU_W'), (Gv + Gv.T).dot(g['in_qU_W'])))
    print('  gqU_a', rel(c('gqU_a'), np.trace(Gv).reshape(1)))

# ---- conditioning stress: M = 100 RBF Gram matrices with clustered inducing points, statistics of N = 4000 rows ----
from oracle import inference as inf, psi_stats as ps
for spread, dup in ((1.0, 0), (0.3, 0), (1.0, 10), (0.1, 20)):
    rng = np.random.default_rng(7)
    M, Q, D, N = 100, 12, 16, 4000
    Z = rng.standard_normal((M, Q)) * spread
    if dup:
        Z[-dup:] = Z[:dup] + 1e-3 * rng.standard_normal((dup, Q))         # near-duplicate inducing points
    il, var, noise = np.full(Q, 0.3), 1.3, 0.05
    mu = rng.standard_normal((N, Q)) * spread; S = 0.05 + 0.1 * rng.random((N, Q))
    P = ps.PsiStatsFast(64)
    psi0, psi1, psi2 = np.full(N, var), P.psi1(var, il, Z, mu, S), P.psi2(var, il, Z, mu, S)
    Kuu = var * np.exp(-0.5 * (((Z[:, None] - Z[None]) * il) ** 2).sum(-1))
    Y = rng.standard_normal((N, D))
    qm = rng.standard_normal((M, D)); W = 0.3 * rng.standard_normal((M, M)); a = 0.5
    qS = W.dot(W.T) + a * np.eye(M)
    st = inf.Stats(psi0.sum(), psi1, psi2, Y)
    logL, g, winv, wvec, mid = inf.svi(Kuu, st, noise, qm, qS)
    KL, dKLm, dKLv, dKLK = inf.kl_qU(qm, qS, mid)
    ratio = 0.37
    o = H.svi_chain(N, dev(Kuu + 1e-6 * np.eye(M)), dev([noise]), dev(qm), dev(W), dev([a]), dev([st.psi0sum]),
                    dev(Y.T.dot(psi1)), dev(st.psi2), dev([(Y ** 2).sum()]), qU_ratio=ratio)
    H.sync()
    c = lambda k: o[k].cpu().numpy()
    Gv = g['dL_dqU_var'] - ratio * dKLv
    print('spread %.1f dup %d cond(K) %.2e:  bound %.1e dKmm %.1e dpsi2 %.1e winv %.1e gqU_m %.1e gqU_W %.1e gqU_a %.1e dthetaL %.1e' % (
        spread, dup, np.linalg.cond(Kuu + 1e-6 * np.eye(M)), abs(c('bound')[0] - (logL - ratio * KL)) / abs(logL),
        rel(c('dL_dKmm'), g['dL_dKmm'] - ratio * dKLK), rel(c('dL_dpsi2'), g['dL_dpsi2']), rel(c('woodbury_inv'), winv),
        rel(c('gqU_m'), g['dL_dqU_mean'] - ratio * dKLm), rel(c('gqU_W'), (Gv + Gv.T).dot(W)),
        rel(c('gqU_a'), np.trace(Gv).reshape(1)), rel(c('dL_dthetaL'), np.asarray(g['dL_dthetaL']).reshape(1))))
