"""rgp_matmul (the fp64 MFMA product behind the blocked and the wide-window paths) against torch's fp64 matmul: both tile
sizes (64 x 64 for small outputs, 128 x 128 with the XCD-aware tile order for large ones), every transpose combination, ragged
edges in all three dimensions, strided rows, split-k and accumulation.  Tolerance: 1e-13 of the largest |A||B| row-column
product (different summation orders of k)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def handle():
    from rgp_amd.capi import Handle
    return Handle()


def _ops(m, n, k, ta, tb, pad, seed):
    g = torch.Generator(device='cuda').manual_seed(seed)
    ra = (k, m) if ta else (m, k)
    rb = (n, k) if tb else (k, n)
    A = torch.randn(ra[0], ra[1] + pad, dtype=torch.float64, device='cuda', generator=g)[:, :ra[1]]
    B = torch.randn(rb[0], rb[1] + pad, dtype=torch.float64, device='cuda', generator=g)[:, :rb[1]]
    return A, B


def _check(out, A, B, ta, tb, alpha=1.0, gamma=0.0, C0=None):
    opA = A.t() if ta else A
    opB = B.t() if tb else B
    ref = alpha * (opA @ opB)
    if C0 is not None:
        ref = ref + gamma * C0
    bound = abs(alpha) * (opA.abs() @ opB.abs()).max().item() + 1.0
    assert (out - ref).abs().max().item() <= 1e-13 * bound


@pytest.mark.parametrize('ta', [False, True])
@pytest.mark.parametrize('tb', [False, True])
@pytest.mark.parametrize('shape', [
    (70, 45, 33),            # small tiles, ragged everywhere
    (5050, 344, 700),        # 40 x 3 large tiles would not fill the chip: small tiles, many workgroups
    (2900, 1100, 137),       # likewise (23 x 9), ragged k
    (4200, 2100, 90),        # large tiles (33 x 17 = 561 tiles), k not a multiple of 16, ragged last tiles
])
def test_matmul_matches_torch(handle, shape, ta, tb):
    m, n, k = shape
    A, B = _ops(m, n, k, ta, tb, pad=3, seed=m + 2 * n + (5 if ta else 0) + (7 if tb else 0))
    out = handle.matmul(A, B, ta=ta, tb=tb)
    handle.sync()
    _check(out, A, B, ta, tb)


@pytest.mark.parametrize('ta,tb', [(True, False), (False, True)])
def test_split_k_and_strided_output(handle, ta, tb):
    m, n, k = 5050, 344, 6000
    A, B = _ops(m, n, k, ta, tb, pad=1, seed=11)
    wide = torch.zeros(m, n + 5, dtype=torch.float64, device='cuda')
    out = wide[:, :n]
    handle.matmul(A, B, ta=ta, tb=tb, out=out, splits=7, alpha=-0.5)
    handle.sync()
    _check(out, A, B, ta, tb, alpha=-0.5)
    assert wide[:, n:].abs().max().item() == 0.0
    dense = handle.matmul(A, B, ta=ta, tb=tb, splits=16)
    handle.sync()
    _check(dense, A, B, ta, tb)


def test_accumulate_large_and_small(handle):
    for (m, n, k) in [(4200, 2100, 64), (100, 100, 340)]:
        A, B = _ops(m, n, k, False, True, pad=0, seed=3)
        C0 = torch.randn(m, n, dtype=torch.float64, device='cuda')
        out = C0.clone()
        handle.matmul(A, B, tb=True, alpha=2.0, gamma=-0.25, out=out)
        handle.sync()
        _check(out, A, B, False, True, alpha=2.0, gamma=-0.25, C0=C0)


def test_bad_arguments_fail_loudly(handle):
    from rgp_amd.capi import RgpError
    A = torch.zeros(8, 8, dtype=torch.float64, device='cuda')
    with pytest.raises(RgpError):
        handle.matmul(A, A, gamma=1.0, out=torch.zeros(8, 8, dtype=torch.float64, device='cuda'), splits=2)
