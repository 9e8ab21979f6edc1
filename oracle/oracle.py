"""ctypes front-end of the CPU ORACLE (oracle/libfpc_oracle.so).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py; the product package never imports this module.
Parity status: predictor/VQ/ceps2lpc side pinned by tests/golden (reference Python
outputs); LPCNet vocoder side PARITY UNPINNED (see fpc_oracle.c header).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def _san():
    # FPC_ORACLE_SAN=1: the sanitizer build (make SAN=1); the process must have been started with the ASan runtime
    # preloaded (tests/test_oracle_sanitized.py does that for a child pytest)
    return os.environ.get("FPC_ORACLE_SAN") == "1"


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE] + (["SAN=1"] if _san() else []))


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libfpc_oracle_san.so" if _san() else "libfpc_oracle.so")
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
        _LIB.orc_cal_entropy.restype = C.c_double
        _LIB.orc_ulaw2lin.restype = C.c_float
        _LIB.orc_ulaw2lin.argtypes = [C.c_int]
        _LIB.orc_lin2ulaw.argtypes = [C.c_float]
        for n in ("orc_tanh", "orc_sigmoid", "orc_exp", "orc_log"):
            getattr(_LIB, n).restype = C.c_float
            getattr(_LIB, n).argtypes = [C.c_float]
        _LIB.orc_philox_uniform.restype = C.c_float
        _LIB.orc_philox_uniform.argtypes = [C.c_uint64, C.c_uint32]
        _LIB.orc_period_index.argtypes = [C.c_float]
        _LIB.orc_lpcnet_create.restype = C.c_void_p
        _LIB.orc_lpcnet_destroy.argtypes = [C.c_void_p]
        _LIB.orc_lpcnet_nblocks.argtypes = [C.c_void_p]
    return _LIB


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class _Pred(C.Structure):
    _fields_ = [("in_", C.c_int), ("h1", C.c_int), ("h2", C.c_int), ("fc", C.c_int)] + [
        (n, C.c_void_p)
        for n in ("w1_ih", "w1_hh", "b1_ih", "b1_hh", "w2_ih", "w2_hh", "b2_ih", "b2_hh", "fc_w", "fc_b")
    ]


_SD_KEYS = ["rnn1.weight_ih_l0", "rnn1.weight_hh_l0", "rnn1.bias_ih_l0", "rnn1.bias_hh_l0",
            "rnn2.weight_ih_l0", "rnn2.weight_hh_l0", "rnn2.bias_ih_l0", "rnn2.bias_hh_l0",
            "dual_fc.0.weight", "dual_fc.0.bias"]


class Predictor:
    """Wavernn.forward / Wavernn.encoder restatement (src/models/wavernn.py:63-102,165-256)."""

    def __init__(self, state_dict):
        self.keep = [_f32(np.asarray(state_dict[k])) for k in _SD_KEYS]
        h1 = self.keep[1].shape[1]
        h2 = self.keep[5].shape[1]
        self.s = _Pred(self.keep[0].shape[1], h1, h2, self.keep[8].shape[0], *[a.ctypes.data for a in self.keep])
        self.h1, self.h2, self.fc, self.inf = h1, h2, self.keep[8].shape[0], self.keep[0].shape[1]

    def forward(self, x, h1=None, h2=None):
        x = _f32(x)
        B, L, _ = x.shape
        h1 = np.zeros((B, self.h1), np.float32) if h1 is None else _f32(h1).reshape(B, self.h1).copy()
        h2 = np.zeros((B, self.h2), np.float32) if h2 is None else _f32(h2).reshape(B, self.h2).copy()
        y = np.zeros((B, L, self.fc), np.float32)
        lib().orc_predictor_forward(C.byref(self.s), _p(x), B, L, _p(h1), _p(h2), _p(y))
        return y, h1, h2

    def encode(self, feat, cb, l1, l2, qtz=True):
        feat = _f32(feat)
        B, L, Cc = feat.shape
        out = dict(
            c_in=np.zeros((B, L, Cc), np.float32), r=np.zeros((B, L, 18), np.float32),
            r_qtz=np.zeros((B, L, 18), np.float32), r_under=np.zeros((B, L, 18), np.float32),
            ind1=np.zeros((B, L), np.float32), ind2=np.zeros((B, L), np.float32),
            idx=np.zeros((B, L, 4), np.int32),
        )
        hist = np.zeros(cb.hist_size if cb is not None else 1, np.float64)
        lib().orc_encode(C.byref(self.s), C.byref(cb.s) if cb is not None else None, _p(feat), B, L,
                         C.c_float(l1), C.c_float(l2), int(bool(qtz)), _p(out["c_in"]), _p(out["r"]),
                         _p(out["r_qtz"]), _p(out["r_under"]), _p(out["ind1"]), _p(out["ind2"]),
                         _p(out["idx"]), _p(hist))
        out["hist"] = hist
        return out


    def decode(self, cb, idx, pitch):
        """receiver side: c_in[:,1:,:] from the symbols of encode() and the pitch columns"""
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        pitch = _f32(pitch)
        B, L, _ = idx.shape
        out = np.zeros((B, L, 20), np.float32)
        rc = lib().orc_decode_features(C.byref(self.s), C.byref(cb.s), _p(pitch), _p(idx), B, L, _p(out))
        assert rc == 0, "oracle: symbol outside its codebook"
        return out


class _CB(C.Structure):
    _fields_ = [("S_hi", C.c_int), ("N_hi", C.c_int * 2), ("vq_hi", C.c_void_p), ("N_lo", C.c_int),
                ("vq_lo", C.c_void_p), ("n_hi", C.c_int), ("scl_hi", C.c_void_p), ("n_lo", C.c_int),
                ("scl_lo", C.c_void_p)]


def _stages(cb):
    """codebook file content -> list of (N_s,17) float64 stages (vq_func.py:141-146)."""
    if isinstance(cb, np.ndarray) and cb.dtype != object:
        assert cb.ndim == 3
        return [_f64(cb[s]) for s in range(cb.shape[0])]
    return [_f64(np.asarray(s)) for s in cb]


class Codebooks:
    def __init__(self, vq_hi, scl_hi, vq_lo=None, scl_lo=None):
        st = _stages(vq_hi)
        assert len(st) in (1, 2)
        self.vq_hi = _f64(np.concatenate(st, 0))
        self.N_hi = [s.shape[0] for s in st]
        self.vq_lo = _f64(_stages(vq_lo)[-1]) if vq_lo is not None else None
        self.scl_hi = _f64(np.asarray(scl_hi).reshape(-1))
        self.scl_lo = _f64(np.asarray(scl_lo).reshape(-1)) if scl_lo is not None else None
        n2 = (C.c_int * 2)(self.N_hi[0], self.N_hi[1] if len(st) == 2 else 0)
        self.s = _CB(len(st), n2, self.vq_hi.ctypes.data,
                     self.vq_lo.shape[0] if self.vq_lo is not None else 0,
                     self.vq_lo.ctypes.data if self.vq_lo is not None else None,
                     self.scl_hi.size, self.scl_hi.ctypes.data,
                     self.scl_lo.size if self.scl_lo is not None else 0,
                     self.scl_lo.ctypes.data if self.scl_lo is not None else None)
        self.sizes = [self.scl_hi.size, self.scl_lo.size if self.scl_lo is not None else 0,
                      self.N_hi[0], self.N_hi[1] if len(st) == 2 else 0,
                      self.vq_lo.shape[0] if self.vq_lo is not None else 0]
        self.hist_size = int(sum(self.sizes))

    def split_hist(self, hist):
        out, o = [], 0
        for n in self.sizes:
            out.append(np.array(hist[o:o + n]))
            o += n
        return out


def vq_mbest(codebook, x):
    """vq_quantize_mbest (vq_func.py:10-24)"""
    cb = _f64(codebook)
    xx = _f64(x)
    idx = np.zeros(5, np.int32)
    dist = np.zeros(5, np.float64)
    lib().orc_vq_mbest(_p(cb), cb.shape[0], _p(xx), _p(idx), _p(dist))
    return idx, dist


def vq_quantize(r, codebook):
    """vq_quantize (vq_func.py:134-164): r (n,17) float32 -> qr (n,17) f64, idx (n,2), hists"""
    st = _stages(codebook)
    cb = _f64(np.concatenate(st, 0))
    ne = np.array([s.shape[0] for s in st], np.int32)
    r = _f32(r)
    n = r.shape[0]
    qr = np.zeros((n, 17), np.float64)
    idx = np.zeros((n, 2), np.int32)
    hist = np.zeros(int(ne.sum()), np.float64)
    lib().orc_vq_quantize(_p(r), n, len(st), _p(ne), _p(cb), _p(qr), _p(idx), _p(hist))
    hs, o = [], 0
    for k in ne:
        hs.append(hist[o:o + k].copy())
        o += k
    return qr, idx, hs


def scl_quantize(x, codes):
    """scl_quantize (vq_func.py:167-185): x (n,1) -> q (n,1) f64, idx, hist"""
    codes = _f64(np.asarray(codes).reshape(-1))
    x = _f32(np.asarray(x).reshape(-1))
    q = np.zeros(x.size, np.float64)
    idx = np.zeros(x.size, np.int32)
    hist = np.zeros(codes.size, np.float64)
    lib().orc_scl_quantize(_p(x), x.size, _p(codes), codes.size, _p(q), _p(idx), _p(hist))
    return q[:, None], idx, hist


def cal_entropy(hist):
    h = _f64(hist)
    return float(lib().orc_cal_entropy(_p(h), h.size))


def shape_cut(p, pitch_corr):
    """pdf shaping + tail cut of one 256-level pdf as the vocoder applies it (un-normalised result)"""
    p = _f32(p)
    out = np.zeros(256, np.float32)
    lib().orc_shape_cut(_p(p), C.c_float(pitch_corr), _p(out))
    return out


def period_index(x):
    return int(lib().orc_period_index(C.c_float(x)))


def ceps2lpc(ceps):
    """ceps2lpc_v (ceps2lpc_vct.py:122-162): (N,>=18) -> lpc (N,16), e (N,), rc (N,16)"""
    c = _f32(ceps)
    N, stride = c.shape
    lpc = np.zeros((N, 16), np.float32)
    e = np.zeros(N, np.float32)
    rc = np.zeros((N, 16), np.float32)
    lib().orc_ceps2lpc(_p(c), N, stride, _p(lpc), _p(e), _p(rc))
    return lpc, e, rc


def l2u_ref(x):
    x = _f32(x)
    u = np.zeros_like(x)
    lib().orc_l2u_ref(_p(x), x.size, _p(u))
    return u


def u2l_ref(u):
    u = _f32(u)
    x = np.zeros_like(u)
    lib().orc_u2l_ref(_p(u), u.size, _p(x))
    return x


def lpc_pred_ref(x, lpc, frame=160):
    x = _f32(x)
    lpc = _f32(lpc)
    B, F, _ = lpc.shape
    pred = np.zeros((B, F * frame), np.float32)
    lib().orc_lpc_pred_ref(_p(x.reshape(B, -1)), _p(lpc), B, F, frame, _p(pred))
    return pred


class _LW(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "embed_pitch", "conv1_k", "conv1_b", "conv2_k", "conv2_b", "d1_k", "d1_b", "d2_k", "d2_b",
        "embed_sig", "ga_k", "ga_r", "ga_b", "gb_k", "gb_r", "gb_b", "md_k", "md_b", "md_f")]


LPCNET_KEYS = ["embed_pitch", "conv1_kernel", "conv1_bias", "conv2_kernel", "conv2_bias",
               "dense1_kernel", "dense1_bias", "dense2_kernel", "dense2_bias", "embed_sig",
               "gru_a_kernel", "gru_a_recurrent", "gru_a_bias", "gru_b_kernel", "gru_b_recurrent",
               "gru_b_bias", "md_kernel", "md_bias", "md_factor"]


class LPCNet:
    """test_lpcnet.py restatement -- PARITY UNPINNED (xiph/LPCNet is not in the reference)."""

    def __init__(self, weights):
        self.keep = [_f32(weights[k]) for k in LPCNET_KEYS]
        self.s = _LW(*[a.ctypes.data for a in self.keep])
        self.h = lib().orc_lpcnet_create(C.byref(self.s))
        self.nblocks = lib().orc_lpcnet_nblocks(self.h)

    def __del__(self):
        try:  # (module globals may already be gone at interpreter shutdown)
            if getattr(self, "h", None):
                lib().orc_lpcnet_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def condition(self, feat):
        f = _f32(feat)
        T = f.shape[0]
        cf = np.zeros((T, 128), np.float32)
        lib().orc_lpcnet_condition(C.c_void_p(self.h), _p(f), T, _p(cf))
        return cf

    def synthesize(self, feat, seed, trace=False):
        f = _f32(feat)
        T = f.shape[0]
        pcm = np.zeros(T * 160, np.int16)
        exc = np.zeros(T * 160, np.uint8) if trace else None
        pf = np.zeros(T * 160, np.float32) if trace else None
        lib().orc_lpcnet_synthesize(C.c_void_p(self.h), _p(f), T, C.c_uint64(int(seed)), _p(pcm),
                                    _p(exc) if trace else None, _p(pf) if trace else None)
        return (pcm, exc, pf) if trace else pcm


# ---- codebook training (cb_func.py): update / find_nearest in C, the splitting schedule here ----
def _cb_data(data):
    """float32 rows stay float32, anything else is taken as float64 (later stages train on `qr - r`)"""
    data = np.asarray(data)
    if data.dtype == np.float32:
        return np.ascontiguousarray(data), 0
    return _f64(data), 1


def cb_find_nearest(data, codebook):
    data, f64 = _cb_data(data)
    cb = _f64(codebook)
    idx = np.zeros(data.shape[0], np.int32)
    rc = lib().orc_cb_find_nearest(_p(data), f64, data.shape[0], data.shape[1], _p(cb), cb.shape[0], _p(idx))
    assert rc == 0, "oracle: 17 dimensions only"
    return idx.astype(np.int64)


def cb_update(data, codebook, nb_entries_tmp, return_count=False):
    data, f64 = _cb_data(data)
    cb = _f64(codebook[:nb_entries_tmp])
    out = np.zeros_like(cb)
    count = np.zeros(nb_entries_tmp, np.float64)
    rc = lib().orc_cb_update(_p(data), f64, data.shape[0], data.shape[1], _p(cb), nb_entries_tmp, _p(out), _p(count))
    assert rc == 0, "oracle: 17 dimensions only"
    return (out, count) if return_count else out


def cb_mean0(data):
    data, f64 = _cb_data(data)
    out = np.zeros(data.shape[1], np.float64)
    lib().orc_cb_mean0(_p(data), f64, data.shape[0], data.shape[1], _p(out))
    return out


def cb_quantize(codebook, data):
    return _f64(codebook)[cb_find_nearest(data, codebook)]


def cb_vq_train(data, codebook, nb_entries):
    """cb_func.py:28-54: LBG splitting; draws its perturbations from numpy's global RNG like the reference"""
    ndims = data.shape[1]
    codebook = _f64(codebook).copy()
    codebook[0] = cb_mean0(data)
    e = 1
    while e < nb_entries:
        codebook[e, :] = codebook[0, :]
        delta = .001 * (np.random.rand(e, ndims) / 2)
        codebook[:e, :] += delta
        e += 1
        for _ in range(4):
            codebook[:e, :] = cb_update(data, codebook[:e, :], e)
    for _ in range(10):
        codebook = cb_update(data, codebook, nb_entries)
    return codebook


# ---- predictor training step (train_frame.py:53-120, live branch) ----
TRAIN_KEYS = ["rnn1.weight_ih_l0", "rnn1.weight_hh_l0", "rnn1.bias_ih_l0", "rnn1.bias_hh_l0", "rnn2.weight_ih_l0",
              "rnn2.weight_hh_l0", "rnn2.bias_ih_l0", "rnn2.bias_hh_l0", "dual_fc.0.weight", "dual_fc.0.bias"]


class _Params(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("w1_ih", "w1_hh", "b1_ih", "b1_hh", "w2_ih", "w2_hh", "b2_ih", "b2_hh",
                                          "fc_w", "fc_b")]


class Trainer:
    """state_dict (torch layouts) + Adam moments; step(feat) runs one restated training step in place"""

    def __init__(self, state_dict, lr=1e-4):
        self.p = {k: _f32(state_dict[k]).copy() for k in TRAIN_KEYS}
        self.m = {k: np.zeros_like(v) for k, v in self.p.items()}
        self.v = {k: np.zeros_like(v) for k, v in self.p.items()}
        self.g = {k: np.zeros_like(v) for k, v in self.p.items()}
        self.lr, self.t = float(lr), 0
        self.h1 = self.p["rnn1.weight_hh_l0"].shape[1]
        self.h2 = self.p["rnn2.weight_hh_l0"].shape[1]
        self.fc, self.inf = self.p["dual_fc.0.weight"].shape[0], self.p["rnn1.weight_ih_l0"].shape[1]

    def _s(self, d):
        return _Params(*[d[k].ctypes.data for k in TRAIN_KEYS])

    def step(self, feat):
        feat = _f32(feat)
        B, L, _ = feat.shape
        self.t += 1
        f = lib().orc_train_step
        f.restype = C.c_float
        P, M, V, G = self._s(self.p), self._s(self.m), self._s(self.v), self._s(self.g)
        return float(f(self.inf, self.h1, self.h2, self.fc, C.byref(P), C.byref(M), C.byref(V), C.byref(G), _p(feat),
                       B, L, C.c_double(self.lr), self.t))
