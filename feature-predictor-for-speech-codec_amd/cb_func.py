"""Codebook training (splitting LBG / k-means) on the GPU: the call surface of the reference's
src/quantization/cb_func.py (vq_train :28-54, find_nearest :56-68, update :71-100, quantize :103-112).

The nearest-entry search, the in-order float64 cell sums and the float32 column mean run as HIP
kernels behind the C ABI (include/fpcodec.h, fpc_cb_*); the splitting schedule stays here because it
draws its perturbations from numpy's global RNG exactly like the reference, so a seeded run yields the
reference's codebook bit for bit.  Training vectors are float32 rows for the first stage (what
train_cb.py:170-178 hands over) and float64 for later stages (the residual `qr - r`); pass a CUDA
tensor to keep them resident across calls (vq_train uploads once)."""
import numpy as np
import torch

from . import _lib


def _dev_data(data):
    """(rows on the device, data_f64): float32 rows stay float32 (first stage), anything else becomes
    float64 (later stages train on the residual `qr - r`); a suitable CUDA tensor is used in place"""
    if isinstance(data, torch.Tensor):
        d = data
        want = torch.float32 if d.dtype == torch.float32 else torch.float64
        if d.dtype != want or not d.is_cuda or not d.is_contiguous():
            d = d.to(device="cuda", dtype=want).contiguous()
        return d, int(want == torch.float64)
    a = np.asarray(data)
    a = np.ascontiguousarray(a if a.dtype == np.float32 else a.astype(np.float64))
    return torch.from_numpy(a).cuda(), int(a.dtype == np.float64)


def find_nearest(data, codebook):
    """(nb_vectors,) index of the nearest entry of every vector (float64 distances, first minimum)"""
    _lib.require_gpu()
    d, f64 = _dev_data(data)
    nv, nd = d.shape
    cb = torch.from_numpy(np.ascontiguousarray(np.asarray(codebook, dtype=np.float64))).cuda()
    idx = torch.empty(nv, dtype=torch.int32, device="cuda")
    _lib.check(_lib.lib().fpc_cb_find_nearest(d.data_ptr(), f64, nv, nd, cb.data_ptr(), cb.shape[0], idx.data_ptr(),
                                              _lib.stream_ptr()), "fpc_cb_find_nearest")
    return idx.cpu().numpy().astype(np.int64)


def update(data, codebook, nb_entries_tmp, return_count=False, verbose=False):
    """one k-means iteration: every entry becomes the mean of the vectors nearest to it (0 for an empty
    cell, the reference's count + 1e-20 division); float64 (nb_entries_tmp, ndims)"""
    _lib.require_gpu()
    d, f64 = _dev_data(data)
    nv, nd = d.shape
    e = int(nb_entries_tmp)
    cb = torch.from_numpy(np.ascontiguousarray(np.asarray(codebook, dtype=np.float64)[:e])).cuda()
    out = torch.empty(e, nd, dtype=torch.float64, device="cuda")
    count = torch.empty(e, dtype=torch.float64, device="cuda")
    L = _lib.lib()
    ws = torch.empty(max(1, int(L.fpc_cb_workspace_bytes(nv, e))), dtype=torch.uint8, device="cuda")
    _lib.check(L.fpc_cb_update(d.data_ptr(), f64, nv, nd, cb.data_ptr(), e, out.data_ptr(), count.data_ptr(),
                               ws.data_ptr(), _lib.stream_ptr()), "fpc_cb_update")
    res = out.cpu().numpy()
    if verbose or return_count:
        cnt = count.cpu().numpy()
        if verbose:  # the line cb_func.py:93-94 prints
            print('{} - min: {}, max: {}, small: {}, error: {}'.format(e, cnt.min(), cnt.max(), int((cnt == 0).sum()),
                                                                        float(np.sum((cnt / nv) ** 2))))
        if return_count:
            return res, cnt
    return res


def quantize(codebook, data):
    """(nb_vectors, ndims) float64: every vector replaced by its nearest entry"""
    return np.asarray(codebook, dtype=np.float64)[find_nearest(data, codebook)]


def mean0(data):
    """np.mean(data, 0) as vq_train takes it (float32 accumulation), float64 (ndims,)"""
    _lib.require_gpu()
    d, f64 = _dev_data(data)
    out = torch.empty(d.shape[1], dtype=torch.float64, device="cuda")
    _lib.check(_lib.lib().fpc_cb_mean0(d.data_ptr(), f64, d.shape[0], d.shape[1], out.data_ptr(), _lib.stream_ptr()),
               "fpc_cb_mean0")
    return out.cpu().numpy()


def vq_train(data, codebook, nb_entries, verbose=False):
    """cb_func.py:28-54: start from the mean, add one entry at a time (a copy of entry 0, all older
    entries nudged by .001 * rand / 2 from numpy's global RNG), 4 updates per split, 10 at the end"""
    d, _ = _dev_data(data)
    ndims = d.shape[1]
    codebook = np.array(codebook, dtype=np.float64, copy=True)
    codebook[0] = mean0(d)
    e = 1
    while e < nb_entries:
        codebook[e, :] = codebook[0, :]
        delta = .001 * (np.random.rand(e, ndims) / 2)
        codebook[:e, :] += delta
        e += 1
        for _ in range(4):
            codebook[:e, :] = update(d, codebook[:e, :], e, verbose=verbose)
    for _ in range(10):
        codebook = update(d, codebook, nb_entries, verbose=verbose)
    return codebook
