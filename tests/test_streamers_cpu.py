"""Data-streamer semantics of autoreg/data_streamers.py restated in rgp_amd/data_streamers.py."""
import numpy as np

from rgp_amd.data_streamers import RandomPermutationDataStreamer, StdMemoryDataStreamer, TrivialDataStreamer


def _data(n):
    return [np.full((3, 1), float(i)) for i in range(n)], [np.full((3, 2), float(i)) for i in range(n)]


def test_trivial_returns_everything_each_call():
    Y, X = _data(4)
    ds = TrivialDataStreamer(Y, X)
    assert ds.total_size == 4 and ds.minibatch_size == 4
    for _ in range(3):
        k, idx, Yo, Xo = ds.next_minibatch()
        assert list(idx) == [0, 1, 2, 3] and len(Yo) == 4 and k == 1          # every call is "last in epoch"


def test_random_permutation_is_a_permutation_of_all():
    Y, X = _data(5)
    ds = RandomPermutationDataStreamer(Y, X, seed=1)
    seen = set()
    for _ in range(4):
        k, idx, Yo, Xo = ds.next_minibatch()
        assert sorted(idx) == [0, 1, 2, 3, 4]
        assert [float(y[0, 0]) for y in Yo] == [float(i) for i in idx]        # data follow the indices
        seen.add(tuple(idx))
        assert ds.get_cur_index()[1] == list(idx)
    assert len(seen) > 1


def test_std_memory_walks_and_wraps():
    Y, X = _data(7)
    ds = StdMemoryDataStreamer(Y, X, 3)
    out = [ds.next_minibatch() for _ in range(4)]
    assert [list(o[1]) for o in out] == [[0, 1, 2], [3, 4, 5], [6], [0, 1, 2]]
    assert [o[0] for o in out] == [1, 2, 3, 1]                                # minibatch index restarts with the epoch
    assert ds.minibatch_last_in_epoch() is False
    Y, X = _data(6)
    ds = StdMemoryDataStreamer(Y, X, 3)
    out = [ds.next_minibatch() for _ in range(3)]
    assert [list(o[1]) for o in out] == [[0, 1, 2], [3, 4, 5], [0, 1, 2]]
