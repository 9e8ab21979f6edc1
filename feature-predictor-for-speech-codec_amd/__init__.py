"""fpcodec_amd -- MI355X-native hot path of haiciyang/Feature-predictor-for-speech-codec.

Python host side that mirrors the reference's call surface (``Wavernn.forward`` /
``Wavernn.encoder``, ``vq_quantize``, ``scl_quantize``, ``ceps2lpc_v``, the
``synthesis_qtz`` driver and the LPCNet ``test_lpcnet.py`` vocoder CLI) on top of the
C ABI of ``libfpcodec.so`` (include/fpcodec.h) whose kernels are hand-written HIP for
gfx950.  There is no CPU fallback: using any compute entry point without the built
library and a HIP device raises.
"""
from . import synth  # noqa: F401
from .config import default_cfg  # noqa: F401

__all__ = ["synth", "default_cfg"]


def __getattr__(name):
    # heavy modules (torch, ctypes library) are loaded on first use
    import importlib
    lazy = {
        "Wavernn": ".wavernn", "vq_quantize": ".vq_func", "scl_quantize": ".vq_func",
        "ceps2lpc_v": ".ceps2lpc", "LPCNet": ".lpcnet", "synthesis": ".synthesis_qtz",
        "cal_entropy": ".vq_func", "lib": "._lib", "cb_func": ".cb_func",
    }
    if name in lazy:
        mod = importlib.import_module(lazy[name], __name__)
        return mod if name in ("lib", "cb_func") else getattr(mod, name)
    raise AttributeError(name)
