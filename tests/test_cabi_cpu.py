"""CPU checks of the boundary: the C-ABI library loads, exports every symbol include/rgp_hip.h declares, the ctypes
structs match the header's field order, and the product path refuses to run without a GPU (no CPU fallback)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header():
    return open(os.path.join(ROOT, 'include', 'rgp_hip.h')).read()


def test_library_exports_every_declared_symbol():
    from rgp_amd import _lib
    lib = _lib.load()
    code = re.sub(r'/\*.*?\*/', '', _header(), flags=re.S)            # drop comments
    declared = set(re.findall(r'\b(rgp_[a-z0-9_]+)\s*\(', code))
    assert declared, 'no declarations found'
    for name in sorted(declared):
        assert hasattr(lib, name), 'librgp_hip.so does not export %s' % name
        assert name in _lib.SYMBOLS, 'ctypes signature missing for %s' % name
    assert lib.rgp_abi_version() == 1


def test_struct_fields_follow_the_header():
    from rgp_amd import _lib
    h = _header()
    body_in = h[h.index('typedef struct {'):h.index('} rgp_layer_in;')]
    order = [f for f, _ in _lib.LayerIn._fields_]
    pos = [body_in.index(f) for f in ('M, D', 'X_win', 'U_win', 'n_seq', 'mode', 'y_uncertain', 'int64_t N', 'own_rows',
                                      'up_rows', 'qU_ratio', 'up_var_const', '*own_off', '*own_mean', '*up_mean', '*Z',
                                      '*variance', '*inv_l', '*noise', '*qU_m', '*qU_W', '*qU_a')]
    assert pos == sorted(pos)
    assert order[:2] == ['M', 'D'] and order[-3:] == ['qU_m', 'qU_W', 'qU_a']
    import ctypes as C
    assert C.sizeof(_lib.LayerIn) == 10 * 4 + 3 * 8 + 2 * 8 + 15 * 8
    assert C.sizeof(_lib.LayerOut) == 14 * 8
    # rgp_mlp_scan: 8 + 5 + 3 int32, then 21 pointers; rgp_rnn_scan: 4 int32, then 8 pointers (field order as in the header)
    assert C.sizeof(_lib.MlpScan) == 16 * 4 + 21 * 8
    assert C.sizeof(_lib.RnnScan) == 4 * 4 + 8 * 8
    body = h[h.index('typedef struct rgp_mlp_scan {'):h.index('} rgp_mlp_scan;')]
    pos = [body.index(f) for f in ('n_seq', 'dims[', 'reserved[3]', '* W[', '* b[', 'init_mean', '* up;', '* mean;', '* act[', 'g_mean', 'g_init',
                                   'g_up', 'delta[')]
    assert pos == sorted(pos)
    assert [f for f, _ in _lib.MlpScan._fields_][8:] == ['dims', 'reserved', 'W', 'b', 'init_mean', 'up', 'mean', 'act', 'g_mean', 'g_init',
                                                         'g_up', 'delta']
    body = h[h.index('typedef struct rgp_rnn_scan {'):h.index('} rgp_rnn_scan;')]
    pos = [body.index(f) for f in ('T, B, H, cell', '* gi;', 'W_hh', 'b_hh', 'hseq', 'gates', 'g_out', 'dgi', 'dgh')]
    assert pos == sorted(pos)
    assert [f for f, _ in _lib.RnnScan._fields_] == ['T', 'B', 'H', 'cell', 'gi', 'W_hh', 'b_hh', 'hseq', 'gates', 'g_out', 'dgi', 'dgh']


def test_workspace_estimate_needs_no_gpu():
    from rgp_amd import _lib
    lib = _lib.load()
    b = lib.rgp_workspace_bytes(130432, 100, 40, 16)
    assert 3e8 < b < 3e9


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from rgp_amd.capi import Handle
    with pytest.raises(RuntimeError):
        Handle()


def test_product_code_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, 'rgp_amd')):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dirpath, f)).read()
                assert 'import oracle' not in src and 'from oracle' not in src, f
