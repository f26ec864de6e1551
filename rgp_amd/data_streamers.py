"""Minibatch iterators over lists of sequences, with the semantics of /root/reference/autoreg/data_streamers.py.
Because the model keeps every sequence resident on the device, a streamer's job here is
to say WHICH sequences form the next minibatch; the data lists are still returned so the call signature
``next_minibatch() -> (minibatch_index, sample_indexes, Y_out, X_out)`` matches the reference (data_streamers.py:61,143,237).

Reference behaviour kept on purpose: ``TrivialDataStreamer`` and
``RandomPermutationDataStreamer`` use minibatch_size == total_size, i.e. they always return the whole set (the latter in
a fresh random order); only ``StdMemoryDataStreamer`` walks through the data in contiguous chunks and wraps around."""
import random
import warnings

import numpy as np


def _as_list(a):
    return [a] if isinstance(a, np.ndarray) else list(a)


class _Streamer(object):
    def __init__(self, Y, X, minibatch_size):
        self.Y, self.X = _as_list(Y), _as_list(X)
        assert len(self.Y) == len(self.X), "Input and output size must match"
        self.total_size = len(self.Y)
        assert 0 < minibatch_size <= self.total_size, "Minibatch size must be less than the data size."
        self.minibatch_size = minibatch_size
        self.next_minibatch_start_idx = 0
        self.minibatch_index = 0
        self.iterations_started = False
        self.last_in_epoch = False
        self.previous_indexes_out = None

    def _window(self):
        """Advance the epoch position; returns (start, stop, minibatch_index_of_this_batch)."""
        self.iterations_started = True
        self.minibatch_index += 1
        st = self.next_minibatch_start_idx
        en = min(st + self.minibatch_size, self.total_size)
        self.last_in_epoch = (st + self.minibatch_size) >= self.total_size
        idx_out = self.minibatch_index
        if self.last_in_epoch:
            self.next_minibatch_start_idx, self.minibatch_index = 0, 0
        else:
            self.next_minibatch_start_idx = st + self.minibatch_size
        return st, en, idx_out

    def get_cur_index(self):
        return self.minibatch_index, self.previous_indexes_out

    def minibatch_last_in_epoch(self):
        return self.last_in_epoch


class TrivialDataStreamer(_Streamer):
    """Returns all the data at every call (data_streamers.py:34-86)."""

    def __init__(self, Y, X):
        Y = _as_list(Y)
        _Streamer.__init__(self, Y, X, len(Y))

    def next_minibatch(self):
        st, en, k = self._window()
        self.previous_indexes_out = list(range(self.total_size))
        return k, list(range(self.total_size)), self.Y[st:en], self.X[st:en]


class RandomPermutationDataStreamer(_Streamer):
    """Returns all the data in a fresh random order at every call (data_streamers.py:108-177)."""

    def __init__(self, Y, X, seed=None):
        Y = _as_list(Y)
        if len(Y) == 1:
            warnings.warn("Input has only one sequence. No permutation functionality will be used.", RuntimeWarning)
        _Streamer.__init__(self, Y, X, len(Y))
        self._rng = random.Random(seed)

    def next_minibatch(self):
        st, en, k = self._window()
        inds = self._rng.sample(range(en - st), en - st)
        self.previous_indexes_out = inds[:]
        return k, inds, [self.Y[st + i] for i in inds], [self.X[st + i] for i in inds]


class StdMemoryDataStreamer(_Streamer):
    """Contiguous minibatches of ``minibatch_size`` sequences, wrapping at the end of the data (data_streamers.py:203-267)."""

    def __init__(self, Y, X, minibatch_size):
        Y = _as_list(Y)
        if len(Y) == 1:
            warnings.warn("Input has only one sequence. No permutation functionality will be used.", RuntimeWarning)
        _Streamer.__init__(self, Y, X, minibatch_size)

    def next_minibatch(self):
        st, en, k = self._window()
        inds = list(range(st, en))
        self.previous_indexes_out = inds[:]
        return k, inds, self.Y[st:en], self.X[st:en]
