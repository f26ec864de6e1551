"""The fused psi2 pass over MANY iterations per workgroup, against the oracle.

The packed-team psi2 kernels walk their rows in iterations of NT rows per workgroup and publish row operands, h values and
partial row sums through rings of five, two and three slots; the short sequences of the other model tests give every
workgroup a single iteration.  Here one long sequence makes every workgroup run six or more (232 workgroups x NT rows x 6),
with a ragged last iteration, and the ELBO and every gradient are compared with the oracle as in test_gpu_model
(1e-11 on the bound, 1e-9 of the largest entry on each gradient group)."""
import pytest

from test_gpu_model import build_pair, compare

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('cfg', [
    # config-4 geometry (M = 100: five rows in flight, the reduction scratch of the two-role kernel in its narrow form)
    dict(wins=(0, 20, 20), nDims=(16, 1, 1), n_seq=1, T=7303, M=100, U_win=20, u_dim=1, y_dim=16),
    # small M (eight rows in flight, one partly idle wave per team), three hidden layers with different windows
    dict(wins=(0, 5, 4, 3), nDims=(2, 1, 2, 1), n_seq=1, T=12011, M=6, y_dim=2),
    # M = 50: teams of 64 lanes, Q = 20 + 20 (the wide form of the reduction scratch)
    dict(wins=(0, 20, 20), nDims=(1, 1, 1), n_seq=2, T=6007, M=50, U_win=20, u_dim=1, y_dim=1),
    # the widest windows: Q = 60 (padded to 64: a z row no longer fits the scalar registers at once) and Q = 45 (padded to 48)
    dict(wins=(0, 30, 30), nDims=(2, 1, 1), n_seq=1, T=12040, M=20, U_win=30, u_dim=1, y_dim=2),
    dict(wins=(0, 15, 15), nDims=(2, 2, 1), n_seq=1, T=12017, M=33, U_win=15, u_dim=1, y_dim=2),
    # row counts just above a whole number of iterations: 6 x 232 iterations of five rows + 10 more (most workgroups' last
    # iteration is empty or ragged); and M = 50 (eight rows in flight): 9 x 232 iterations + 20 more
    dict(wins=(0, 20, 20), nDims=(16, 1, 1), n_seq=1, T=7028, M=100, U_win=20, u_dim=1, y_dim=16),
    dict(wins=(0, 20, 20), nDims=(1, 1, 1), n_seq=2, T=8450, M=50, U_win=20, u_dim=1, y_dim=1),
])
def test_many_iterations_per_workgroup_match_oracle(cfg):
    o, g = build_pair(fast_psi=True, chunk=512, **cfg)
    compare(o, g)


def test_minibatch_of_long_sequences_matches_oracle():
    """Two of three long ragged sequences (qU_ratio = 2/3): the table slice and the rings together."""
    o, g = build_pair(wins=(0, 6, 6), nDims=(3, 2, 2), n_seq=3, T=4001, M=12, U_win=6, ragged=True, n_total=3, fast_psi=True,
                      chunk=512)
    compare(o, g, seqs=[0, 2])
