"""rgp_amd -- MI355X-native recurrent-GP variational inference path (drop-in for the hot path of zhenwendai/RGP's
``autoreg.model.DeepAutoreg*`` / ``autoreg.layers.Layer*``).  The numerical work lives in librgp_hip.so
(include/rgp_hip.h); importing the model needs the built library and a GPU -- there is no CPU fallback."""
from . import _lib  # noqa: F401  (ctypes signatures; loading is deferred to first use)

__all__ = ['DeepAutoreg', 'DeepAutoreg_new', 'DeepAutoreg_rnn', 'DeepAutoreg_mlp', 'Handle']


def __getattr__(name):
    if name in ('DeepAutoreg', 'DeepAutoreg_new'):
        from .model import DeepAutoreg
        return DeepAutoreg
    if name in ('DeepAutoreg_rnn', 'DeepAutoreg_mlp'):
        from . import model
        return getattr(model, name)
    if name == 'Handle':
        from .capi import Handle
        return Handle
    raise AttributeError(name)
