"""The exponent products of the psi1 forward pass at config 4's size (N = 130 432 rows, M = 100, Q = 40) on the fp64 MFMA GEMM
(rgp_matmul), next to the time of the LDS-broadcast VALU kernel that the fused path uses (psi1_svi_fwd2_kernel: 153 us per
launch in profiles/r2_bench_kernel_stats.csv).  Four products N x M x Q as in the wide-window path (e (Z^2)^T, (e mu) Z^T,
(d2 mu) Z^T, d2 (Z^2)^T) plus one element-wise pass for exp / the h offset.

    python tools/psi1_mfma_ab.py > profiles/rX_psi1_mfma_ab.txt        (on the GPU box)
"""
import torch
from rgp_amd.capi import Handle

h = Handle()
N, M, Q = 130432, 100, 40
g = torch.Generator(device='cuda').manual_seed(0)
rows = [torch.randn(N, Q, dtype=torch.float64, device='cuda', generator=g) for _ in range(4)]
ZZ = torch.randn(M, 2 * Q + 8, dtype=torch.float64, device='cuda', generator=g)
Z1, Z2 = ZZ[:, :Q], ZZ[:, Q:2 * Q]
P = torch.empty(N, M, dtype=torch.float64, device='cuda')
H = torch.empty(N, M, dtype=torch.float64, device='cuda')
g1 = torch.randn(N, 1, dtype=torch.float64, device='cuda', generator=g)


def products():
    h.matmul(rows[0], Z2, tb=True, out=P)
    h.matmul(rows[1], Z1, tb=True, alpha=-2.0, gamma=1.0, out=P)
    h.matmul(rows[2], Z1, tb=True, out=H)
    h.matmul(rows[3], Z2, tb=True, alpha=-0.25, gamma=1.0, out=H)


def elementwise():
    torch.exp_(P.mul_(-0.5).add_(g1))


for _ in range(3):
    products(); elementwise()
h.sync()
for name, fn in (('four N x M x Q products (MFMA GEMM, 64 x 64 tiles)', products), ('element-wise exp pass (torch)', elementwise)):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        fn()
    b.record()
    torch.cuda.synchronize()
    print('%-55s %7.1f us per psi1 forward' % (name, 1e3 * a.elapsed_time(b) / 20))
print('psi1_svi_fwd2_kernel<40,1> (the fused path, also forms Lp and the psi1^T Y partials): 153 us (profiles/r2_bench_kernel_stats.csv)')
