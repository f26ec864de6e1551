"""rgp_psi_forward_static (the entry point that allocates nothing: its scratch is a static device array) against
rgp_psi_forward through the C ABI: two calls in a row on the same GPU, the second with a smaller input, so stale scratch
contents of the first call would show; the device's free memory is the same before and after a call.  fp64, same kernels, partial sums combined over fewer workgroups: 1e-13 relative."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_static_scratch_entry_point_matches_the_allocating_one():
    import torch
    from rgp_amd.capi import Handle
    from rgp_amd._lib import RgpError
    H = Handle()
    dev = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device=H.device)
    rng = np.random.default_rng(42)
    for (N, M, Q) in [(3000, 100, 40), (257, 33, 20)]:            # the second call is smaller in every dimension
        mu = dev(rng.standard_normal((N, Q)) + 0.5)
        S = dev(0.02 + 0.3 * rng.random((N, Q)))
        Z = dev(rng.standard_normal((M, Q)) * 1.2 + 0.5)
        var, il = dev([0.5 + rng.random()]), dev(0.3 + rng.random(Q))
        want1, want2 = H.psi_forward(mu, S, Z, var, il)
        got1 = torch.full((N, M), float('nan'), dtype=torch.float64, device=H.device)
        got2 = torch.full((M, M), float('nan'), dtype=torch.float64, device=H.device)
        H.psi_forward_static(mu, S, Z, var, il, got1, got2)       # the first call at a shape may load code objects
        H.sync()
        torch.cuda.synchronize()
        free0, reserved0 = torch.cuda.mem_get_info()[0], torch.cuda.memory_reserved()
        got1.fill_(float('nan')); got2.fill_(float('nan'))
        H.psi_forward_static(mu, S, Z, var, il, got1, got2)
        H.sync()
        torch.cuda.synchronize()
        # what the entry point is for: the call took no device memory, neither through the library nor through torch
        assert torch.cuda.mem_get_info()[0] == free0 and torch.cuda.memory_reserved() == reserved0
        assert torch.equal(got1, want1)                           # psi1: same kernel, no reduction
        err = float((got2 - want2).abs().max() / want2.abs().max())
        assert err <= 1e-13, (N, M, Q, err)
        assert torch.equal(got2, got2.T)
    # beyond the static limits: a clear error, nothing launched
    N, M, Q = 5000, 16, 8
    with pytest.raises(RgpError, match='static scratch'):
        H.psi_forward_static(dev(np.zeros((N, Q))), dev(np.ones((N, Q))), dev(np.zeros((M, Q))), dev([1.0]), dev(np.ones(Q)),
                             dev(np.zeros((N, M))), dev(np.zeros((M, M))))
    H.close()
