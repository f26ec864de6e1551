"""ctypes binding of librgp_hip.so (include/rgp_hip.h).  No torch types cross this boundary: only raw device
pointers (tensor.data_ptr()), sizes and POD structs.  Import fails loudly if the library is missing -- there is no
CPU fallback in the product path."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('RGP_HIP_LIB') or os.path.join(_HERE, 'librgp_hip.so')

RGP_OK, RGP_ERR_BAD_ARG, RGP_ERR_HIP, RGP_ERR_NOT_PD, RGP_ERR_WORKSPACE = 0, 1, 2, 3, 4
RGP_MODE_VARDTC, RGP_MODE_SVI = 0, 1

c_dp = C.c_void_p          # device pointer to double
c_ip = C.c_void_p          # device pointer to int64


class LayerIn(C.Structure):
    _fields_ = [('M', C.c_int32), ('D', C.c_int32), ('X_win', C.c_int32), ('X_dim', C.c_int32),
                ('U_win', C.c_int32), ('U_dim', C.c_int32), ('n_seq', C.c_int32), ('mode', C.c_int32),
                ('y_uncertain', C.c_int32), ('reserved', C.c_int32),
                ('N', C.c_int64), ('own_rows', C.c_int64), ('up_rows', C.c_int64),
                ('qU_ratio', C.c_double), ('up_var_const', C.c_double),
                ('own_off', c_ip), ('own_len', c_ip), ('up_off', c_ip), ('up_len', c_ip),
                ('own_mean', c_dp), ('own_var', c_dp), ('up_mean', c_dp), ('up_var', c_dp),
                ('Z', c_dp), ('variance', c_dp), ('inv_l', c_dp), ('noise', c_dp),
                ('qU_m', c_dp), ('qU_W', c_dp), ('qU_a', c_dp)]


class LayerOut(C.Structure):
    _fields_ = [('elbo', c_dp), ('gZ', c_dp), ('gvariance', c_dp), ('ginv_l', c_dp), ('gnoise', c_dp),
                ('gqU_m', c_dp), ('gqU_W', c_dp), ('gqU_a', c_dp),
                ('g_own_mean', c_dp), ('g_own_var', c_dp), ('g_up_mean', c_dp), ('g_up_var', c_dp),
                ('woodbury_inv', c_dp), ('woodbury_vector', c_dp)]


MLP_MAX_LAYERS = 4


class MlpScan(C.Structure):
    _fields_ = [('n_seq', C.c_int32), ('X_win', C.c_int32), ('X_dim', C.c_int32), ('U_win', C.c_int32), ('U_dim', C.c_int32),
                ('N', C.c_int32), ('up_len', C.c_int32), ('n_layers', C.c_int32),
                ('dims', C.c_int32 * (MLP_MAX_LAYERS + 1)), ('reserved', C.c_int32 * 3),
                ('W', c_dp * MLP_MAX_LAYERS), ('b', c_dp * MLP_MAX_LAYERS),
                ('init_mean', c_dp), ('up', c_dp), ('mean', c_dp), ('act', c_dp * (MLP_MAX_LAYERS - 1)),
                ('g_mean', c_dp), ('g_init', c_dp), ('g_up', c_dp), ('delta', c_dp * MLP_MAX_LAYERS)]


CELL_GRU, CELL_RNN_TANH, CELL_LSTM = 0, 1, 2


class RnnScan(C.Structure):
    _fields_ = [('T', C.c_int32), ('B', C.c_int32), ('H', C.c_int32), ('cell', C.c_int32),
                ('gi', c_dp), ('W_hh', c_dp), ('b_hh', c_dp), ('hseq', c_dp), ('gates', c_dp),
                ('g_out', c_dp), ('dgi', c_dp), ('dgh', c_dp)]


# every symbol include/rgp_hip.h declares: name -> (restype, argtypes)
SYMBOLS = {
    'rgp_create': (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_void_p]),
    'rgp_destroy': (None, [C.c_void_p]),
    'rgp_last_error': (C.c_char_p, [C.c_void_p]),
    'rgp_abi_version': (C.c_int, []),
    'rgp_sync_status': (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    'rgp_join': (C.c_int, [C.c_void_p]),
    'rgp_workspace_bytes': (C.c_int64, [C.c_int64, C.c_int, C.c_int, C.c_int]),
    'rgp_set_profiling': (C.c_int, [C.c_void_p, C.c_int]),
    'rgp_set_deterministic': (C.c_int, [C.c_void_p, C.c_int]),
    'rgp_set_psi2_pair': (C.c_int, [C.c_void_p, C.c_int]),
    'rgp_psi2_pair_would_run': (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_int]),
    'rgp_side_streams_independent': (C.c_int, [C.c_void_p]),
    'rgp_get_profile': (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double), C.c_int]),
    'rgp_vardtc_stats': (C.c_int, [C.c_void_p, C.POINTER(LayerIn), c_dp]),
    'rgp_vardtc_finish': (C.c_int, [C.c_void_p, C.POINTER(LayerIn), C.POINTER(LayerOut), c_dp, C.c_int64, C.c_double]),
    'rgp_layer_prepare': (C.c_int, [C.c_void_p, C.POINTER(LayerIn)]),
    'rgp_layer_step': (C.c_int, [C.c_void_p, C.POINTER(LayerIn), C.POINTER(LayerOut)]),
    'rgp_kuu': (C.c_int, [C.c_void_p, C.c_int, C.c_int, c_dp, c_dp, c_dp, C.c_double, c_dp]),
    'rgp_matmul': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, c_dp, C.c_int, C.c_int, c_dp, C.c_int, C.c_int,
                             C.c_double, c_dp, C.c_int, C.c_int]),
    'rgp_psi_forward': (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_int, c_dp, c_dp, c_dp, c_dp, c_dp, c_dp, c_dp]),
    'rgp_psi_forward_static': (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_int, c_dp, c_dp, c_dp, c_dp, c_dp, c_dp, c_dp]),
    'rgp_psi_backward': (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_int, c_dp, c_dp, c_dp, c_dp, c_dp, C.c_double,
                                   c_dp, c_dp, c_dp, c_dp, c_dp, c_dp, c_dp]),
    'rgp_svi_chain': (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_double] + [c_dp] * 18),
    'rgp_freerun': (C.c_int, [C.c_void_p] + [C.c_int] * 7 + [c_dp] * 9 + [C.c_double, C.c_int64, C.c_int64, c_dp, c_dp]),
    'rgp_adam_step': (C.c_int, [C.c_void_p, C.c_int64, c_dp, c_dp, c_dp, c_dp, c_dp, C.c_void_p, C.c_double, C.c_double,
                                C.c_double, C.c_double, C.c_int64]),
    'rgp_adam_step_post': (C.c_int, [C.c_void_p, C.c_int64, c_dp, c_dp, c_dp, c_dp, c_dp, C.c_void_p, C.c_double, C.c_double,
                                     C.c_double, C.c_double, C.c_int64]),
    'rgp_discard_prepared': (C.c_int, [C.c_void_p]),
    'rgp_get_stream': (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]),
    'rgp_mlp_scan_forward': (C.c_int, [C.c_void_p, C.POINTER(MlpScan)]),
    'rgp_mlp_scan_backward': (C.c_int, [C.c_void_p, C.POINTER(MlpScan)]),
    'rgp_rnn_scan_forward': (C.c_int, [C.c_void_p, C.POINTER(RnnScan)]),
    'rgp_rnn_scan_backward': (C.c_int, [C.c_void_p, C.POINTER(RnnScan)]),
}
PREPARE_CHAIN_ONLY, PREPARE_ROWS_ONLY, PREPARE_ROWS_EARLY = 1, 2, 4          # rgp_layer_in.reserved for rgp_layer_prepare

_lib = None


def load():
    """Load the shared library (once).  Raises if it has not been built (``__graft_entry__.build()``)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError('%s is missing: the HIP library has not been built (run __graft_entry__.build()); '
                              'rgp_amd has no CPU fallback' % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


class RgpError(RuntimeError):
    def __init__(self, code, msg):
        RuntimeError.__init__(self, 'librgp_hip status %d: %s' % (code, msg))
        self.code = code
