"""`ceps2lpc_v` with the reference's signature (src/ceps2lpc/ceps2lpc_vct.py:122-162)."""
import torch

from . import _lib

NB_BANDS = 18
LPC_ORDER = 16


def ceps2lpc_v(cepstrum, all_rows=False):
    """cepstrum (N, C>=18) float32 (un-normalised) -> (e, lpc (N,16), rc).  Like the
    reference, `e` and `rc` are those of the LAST row (set all_rows=True for every row).
    Tensors come back on the device the kernel ran on."""
    _lib.require_gpu()
    c = cepstrum.to("cuda", torch.float32).contiguous()
    N, stride = c.shape
    lpc = torch.empty(N, LPC_ORDER, device="cuda")
    e = torch.empty(N, device="cuda")
    rc = torch.empty(N, LPC_ORDER, device="cuda")
    _lib.check(_lib.lib().fpc_ceps2lpc(c.data_ptr(), N, stride, lpc.data_ptr(), e.data_ptr(), rc.data_ptr(),
                                       _lib.stream_ptr()), "fpc_ceps2lpc")
    if all_rows:
        return e, lpc, rc
    return e[-1], lpc, rc[-1].to(torch.float64)
