"""`Wavernn` with the reference's call surface (src/models/wavernn.py:22-256) on the
HIP kernels of libfpcodec.so.

Same constructor arguments, same state_dict keys/shapes (wavernn.py:24-52), same
`forward(x, h1, h2) -> (y, h1, h2)` (:63-102) and
`encoder(cfg, feat, mask, l1, l2, vq_quantize, scl_quantize, qtz)` ->
`(c_in[:,1:], r, r_qtz, r_under, ind1_mask, ind2_mask, cb_tot)` (:165-256).  The whole
closed loop of `encoder` runs in one persistent kernel per utterance when the quantizer
callables are this package's own `vq_func.vq_quantize` / `scl_quantize` (or None): the
codebooks named by cfg['scl_cb_path'] / ['cb_path'] / ['bl_scl_cb_path'] / ['bl_cb_path']
are then searched on the device with the same arithmetic.  Any OTHER callable is honoured
as the reference honours it (wavernn.py:219-240): the loop then runs frame by frame, the
predictor step on the device, the injected quantizers on the rows the reference hands them.
Dead/broken reference methods (mask_enc, decoder, loop_attention: SURVEY.md App. C) are not
reproduced.
"""
import ctypes as C
import weakref
from collections import OrderedDict

import numpy as np
import torch

from . import _lib
from .vq_func import load_codebooks

_KEYS = ["rnn1.weight_ih_l0", "rnn1.weight_hh_l0", "rnn1.bias_ih_l0", "rnn1.bias_hh_l0",
         "rnn2.weight_ih_l0", "rnn2.weight_hh_l0", "rnn2.bias_ih_l0", "rnn2.bias_hh_l0",
         "dual_fc.0.weight", "dual_fc.0.bias"]


class _ParameterList(list):
    """what `Wavernn.parameters()` returns: a list of tensors that remembers its model"""

    def __init__(self, model, tensors):
        super().__init__(tensors)
        self.model = model


class Wavernn:
    def __init__(self, in_features=20, gru_units1=384, gru_units2=16, fc_units=20, attn_units=20,
                 rnn_layers=2, bidirectional=False, packing=False):
        if bidirectional or packing:
            raise NotImplementedError("production path is unidirectional, unpacked (train_frame.py:195,203)")
        self.in_features, self.gru_units1, self.gru_units2, self.fc_units = (
            in_features, gru_units1, gru_units2, fc_units)
        self.scale = 1
        self._sd = None
        self._h = None
        self._trainer = None   # weakref to the live train_frame.Trainer built on this model's handle, if any
        self._dirty = False    # the trainer has stepped since state_dict() last matched the device weights
        self.training = False
        self.device = torch.device("cuda")

    # ---- torch.nn.Module-like surface used by synthesis_qtz.py:79-87 ----
    def shapes(self):
        i, h1, h2, f = self.in_features, self.gru_units1, self.gru_units2, self.fc_units
        return OrderedDict(zip(_KEYS, [(3 * h1, i), (3 * h1, h1), (3 * h1,), (3 * h1,), (3 * h2, h1),
                                       (3 * h2, h2), (3 * h2,), (3 * h2,), (f, h2), (f,)]))

    def load_state_dict(self, sd, strict=True):
        """`strict=False` (the transfer load of train_frame.py:244-248) ignores unexpected keys and keeps the
        current value of a missing one; a model that was never loaded has no current values, so there every key
        is required."""
        shp = self.shapes()
        missing = [k for k in _KEYS if k not in sd]
        extra = [k for k in sd if k not in shp]
        if (missing and (strict or self._sd is None)) or (strict and extra):
            raise KeyError(f"state_dict mismatch: missing {missing}, unexpected {extra}")
        if missing:
            cur = self.state_dict()
            sd = dict(sd)
            for k in missing:
                sd[k] = cur[k]
        arrs = []
        for k in _KEYS:
            v = sd[k]
            v = v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
            if tuple(v.shape) != shp[k]:
                raise ValueError(f"{k}: shape {tuple(v.shape)} != {shp[k]}")
            arrs.append(np.ascontiguousarray(v, dtype=np.float32))
        self._sd = OrderedDict(zip(_KEYS, arrs))
        self._release()
        return self

    def state_dict(self):
        """host copies in the reference's checkpoint format (utils.py:127-146).  After training steps the device
        holds newer weights than the host copy: they are pulled first, so a checkpoint never saves stale weights."""
        if self._sd is None:
            raise _lib.FpcError("Wavernn: no weights yet (load_state_dict first)")
        if self._dirty:
            t = self._trainer() if self._trainer is not None else None
            if t is None:
                raise _lib.FpcError("Wavernn.state_dict: the trainer that updated the device weights is gone and "
                                    "was never synced (call Trainer.sync() before dropping it)")
            t.sync()
        return OrderedDict((k, torch.from_numpy(v.copy())) for k, v in self._sd.items())

    def parameters(self):
        """the ten tensors of train_frame.py:250's `optim.Adam(model.parameters(), ...)`: host copies that carry a
        reference to this model, which `train_frame.Adam` uses to find the device weights"""
        return _ParameterList(self, [v for _, v in self.named_parameters()])

    def named_parameters(self):
        return list(self.state_dict().items())

    def to(self, device):
        return self

    def cuda(self, device=None):
        return self

    def train(self, mode=True):
        self.training = bool(mode)
        return self

    def eval(self):
        return self.train(False)

    def _release(self):
        if self._h is not None:
            _lib.lib().fpc_predictor_destroy(self._h)  # drops one reference; a live trainer keeps its own
            self._h = None
        self._trainer = None
        self._dirty = False

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    def _handle(self):
        if self._sd is None:
            raise _lib.FpcError("Wavernn: load_state_dict() first (the reference loads a checkpoint, synthesis_qtz.py:86)")
        if self._h is None:
            _lib.require_gpu()
            a = list(self._sd.values())
            w = _lib.PredictorWeights(self.in_features, self.gru_units1, self.gru_units2, self.fc_units,
                                      *[x.ctypes.data for x in a])
            h = C.c_void_p()
            _lib.check(_lib.lib().fpc_predictor_create(C.byref(w), C.byref(h)), "fpc_predictor_create")
            self._h = h
        return self._h

    # ---- Wavernn.forward (wavernn.py:63-102) ----
    def forward(self, x, h1=None, h2=None):
        h = self._handle()
        x = x.to(self.device, torch.float32).contiguous()
        B, L, Cc = x.shape
        assert Cc == self.in_features
        s1 = torch.zeros(B, self.gru_units1, device=self.device) if h1 is None else \
            h1.to(self.device, torch.float32).reshape(B, self.gru_units1).clone()
        s2 = torch.zeros(B, self.gru_units2, device=self.device) if h2 is None else \
            h2.to(self.device, torch.float32).reshape(B, self.gru_units2).clone()
        y = torch.empty(B, L, self.fc_units, device=self.device)
        self._pcheck(_lib.lib().fpc_predictor_forward(h, x.data_ptr(), B, L, s1.data_ptr(), s2.data_ptr(),
                                                      y.data_ptr(), _lib.stream_ptr()), "fpc_predictor_forward")
        return y, s1.unsqueeze(0), s2.unsqueeze(0)

    __call__ = forward

    def _pcheck(self, rc, what):
        """_lib.check for calls on this model's handle: a sticky failure (a timed-out exchange, a non-finite residual) is
        reported ONCE -- the status word is cleared as the exception is raised (fpc_predictor_status), so the next call
        on the model is accepted again instead of raising until someone remembers check()"""
        if rc in (-5, -6) and self._h is not None:
            msg = _lib.lib().fpc_last_error().decode()
            _lib.lib().fpc_predictor_status(self._h)
            raise _lib.FpcError(f"{what} failed ({rc}): {msg}")
        _lib.check(rc, what)

    def check(self):
        """synchronise and raise FpcError if a launch on this model's handle failed (a row-split exchange that
        timed out, a non-finite residual at a quantizer): fpc_predictor_status of include/fpcodec.h"""
        if self._h is not None:
            _lib.check(_lib.lib().fpc_predictor_status(self._h), "fpc_predictor_status")

    def fallback_groups(self):
        """groups of 16 utterances of the last weights-stationary launch that were served by the row-split fallback because
        their 32 workgroups could not all become resident (a busy or shared GPU); synchronises (fpcodec.h)"""
        n = _lib.lib().fpc_predictor_fallback_groups(self._handle())
        if n < 0:
            _lib.check(n, "fpc_predictor_fallback_groups")
        return n

    def set_split(self, n):
        """workgroups per utterance: 0 automatic (the process owns the GPU), 1 never split (shared GPU), 2 / 4 / 8"""
        _lib.check(_lib.lib().fpc_predictor_set_split(self._handle(), int(n)), "fpc_predictor_set_split")

    def _encoder_host_loop(self, cfg, feat, mask, l1, l2, vq_quantize, scl_quantize, qtz):
        """The reference's loop (wavernn.py:192-252) frame by frame for the cases the fused kernels do not cover:
        quantizer callables that are not this package's own (:165) and a below-threshold VQ file with more than one stage
        (:235-240), either with or without the input mask (:209-211, `mask` of shape (B, L, 2, 1): `mask[:, i, 0]` must
        index like the thresholds' (B, 1) indicator; with the package's own quantizers the mask rides the fused kernel).  The predictor step runs on the device, indicators and the callables on the
        host with exactly the rows, paths and accumulation of wavernn.py:217-252."""
        dev = self.device
        B, L, Cc = feat.shape
        c_in = torch.zeros(B, L + 1, Cc, device=dev)
        c_in[:, 1:, -2:] = feat[:, :, -2:]
        r = torch.zeros(B, L, 18, device=dev)
        r_qtz = torch.zeros(B, L, 18, device=dev)
        r_under = torch.zeros(B, L, 18, device=dev)
        ind1_mask = torch.zeros(B, L, 1, device=dev)
        ind2_mask = torch.zeros(B, L, 1, device=dev)
        cb_tot = [0, 0, 0, 0, 0]
        if mask is not None:
            mask = torch.as_tensor(mask).to(dev)
            if mask.dim() == 3:  # (B, L, 2): the shape the docstring suggests; the reference's indexing needs a 4th axis
                mask = mask.unsqueeze(-1)
        h1 = h2 = None
        for i in range(L):
            f_out, h1, h2 = self.forward(c_in[:, i:i + 1, :], h1, h2)
            f_out = f_out[:, -1, :]
            r_s = feat[:, i, :-2] - f_out
            r[:, i, :] = r_s
            if mask is None:  # thresholds (:201-207)
                ind1 = (abs(r_s[:, 0]) > l1).to(int).unsqueeze(1)
                ind1_mask[:, i, :] = ind1
                ind2 = (torch.sum(abs(r_s[:, 1:]), -1) > l2).to(int).unsqueeze(1)
                ind2_mask[:, i, :] = ind2
            else:  # the input mask (:209-211; the indicator outputs stay zero there)
                ind1 = mask[:, i, 0]
                ind2 = mask[:, i, 1]
            if qtz:
                r_host = r_s.cpu().numpy()
                i1, i2 = ind1.cpu().numpy(), ind2.cpu().numpy()
                for k in range(B):
                    if i1[k, 0]:
                        rq, cb_t = scl_quantize(r_host[k:k + 1, 0:1], cfg['scl_cb_path'])
                        r_qtz[k:k + 1, i, 0:1] = torch.as_tensor(np.asarray(rq), dtype=torch.float32).to(dev)
                        cb_tot[0] = cb_tot[0] + cb_t
                    elif cfg.get('bl_scl_cb_path'):
                        rq, cb_t = scl_quantize(r_host[k:k + 1, 0:1], cfg['bl_scl_cb_path'])
                        r_qtz[k:k + 1, i, 0:1] = torch.as_tensor(np.asarray(rq), dtype=torch.float32).to(dev)
                        cb_tot[1] = cb_tot[1] + cb_t
                for k in range(B):
                    if i2[k, 0]:
                        rq, cb_t = vq_quantize(r_host[k:k + 1, 1:], cfg['cb_path'])
                        r_qtz[k:k + 1, i, 1:] = torch.as_tensor(np.asarray(rq), dtype=torch.float32).to(dev)
                        cb_tot[2] = cb_tot[2] + cb_t[0]
                        cb_tot[3] = cb_tot[3] + cb_t[1]
                    elif cfg.get('bl_cb_path'):
                        rq, cb_t = vq_quantize(r_host[k:k + 1, 1:], cfg['bl_cb_path'])
                        r_qtz[k:k + 1, i, 1:] = torch.as_tensor(np.asarray(rq), dtype=torch.float32).to(dev)
                        cb_tot[4] = cb_tot[4] + cb_t[-1]
                c_in[:, i + 1, :-2] = f_out + r_qtz[:, i, :]
            else:  # (:244-252)
                r_under[:, i, 0:1] = r_s[:, 0:1] * (1 - ind1)
                r_under[:, i, 1:] = r_s[:, 1:] * (1 - ind2)
                r[:, i, 0:1] = r_s[:, 0:1] * ind1
                r[:, i, 1:] = r_s[:, 1:] * ind2
                c_in[:, i + 1, :-2] = f_out + r[:, i, :]
        self.check()
        return c_in[:, 1:, :], r, r_qtz, r_under, ind1_mask, ind2_mask, cb_tot

    # ---- Wavernn.encoder (wavernn.py:165-256) ----
    def encoder(self, cfg, feat, mask, l1, l2, vq_quantize=None, scl_quantize=None, qtz=True,
                return_indices=False):
        from . import vq_func as _vq
        own = (vq_quantize is None or vq_quantize is _vq.vq_quantize) and \
            (scl_quantize is None or scl_quantize is _vq.scl_quantize)
        # the fused kernels code the below-threshold residual with ONE stage (the production file, 1 x 512 x 17); a file with
        # more stages takes the reference's own route: quantize_mstage over all of them (wavernn.py:235-240)
        lo_stages = _vq.vq_file_stages(cfg["bl_cb_path"]) if (qtz and cfg.get("bl_cb_path")) else 1
        if (qtz and not own) or lo_stages > 1:
            # served frame by frame (wavernn.py:165: the reference calls whatever it is handed)
            if return_indices:
                raise _lib.FpcError("Wavernn.encoder: return_indices needs the fused path (the built-in quantizers, a "
                                    "1-stage below-threshold book)")
            if qtz and not own and (vq_quantize is None or scl_quantize is None):
                raise _lib.FpcError("Wavernn.encoder: pass both vq_quantize and scl_quantize or neither "
                                    "(wavernn.py:219,230 call both)")
            return self._encoder_host_loop(cfg, feat.to(self.device, torch.float32).contiguous(), mask, l1, l2,
                                           vq_quantize or _vq.vq_quantize, scl_quantize or _vq.scl_quantize, qtz)
        h = self._handle()
        feat = feat.to(self.device, torch.float32).contiguous()
        B, L, Cc = feat.shape
        dev = self.device
        c_in = torch.empty(B, L, Cc, device=dev)
        r = torch.empty(B, L, 18, device=dev)
        r_qtz = torch.empty(B, L, 18, device=dev)
        r_under = torch.empty(B, L, 18, device=dev)
        ind1 = torch.empty(B, L, 1, device=dev)
        ind2 = torch.empty(B, L, 1, device=dev)
        idx = torch.empty(B, L, 4, device=dev, dtype=torch.int32)
        mk = None
        if mask is not None:  # the input-mask mode (wavernn.py:209-211) rides the fused kernel: fpc_encode's mask_dev
            mk = torch.as_tensor(mask).to(dev, torch.float32)
            if mk.dim() == 4:  # (B, L, 2, 1): the shape the reference's indexing needs (mask[:, i, 0] must be (B, 1))
                mk = mk.squeeze(-1)
            if tuple(mk.shape) != (B, L, 2):
                raise _lib.FpcError(f"Wavernn.encoder: mask of shape {tuple(mask.shape)}; (B, L, 2) or (B, L, 2, 1) expected")
            mk = mk.contiguous()
        cb = None
        hist = None
        if qtz:
            cb = load_codebooks(cfg["cb_path"], cfg["scl_cb_path"], cfg.get("bl_cb_path") or None,
                                cfg.get("bl_scl_cb_path") or None)
            hist = torch.zeros(cb.hist_size, device=dev, dtype=torch.int64)
        self._pcheck(_lib.lib().fpc_encode(
            h, cb.handle if cb else None, feat.data_ptr(), B, L, float(l1), float(l2), int(bool(qtz)),
            c_in.data_ptr(), r.data_ptr(), r_qtz.data_ptr(), r_under.data_ptr(), ind1.data_ptr(),
            ind2.data_ptr(), idx.data_ptr(), hist.data_ptr() if hist is not None else None,
            mk.data_ptr() if mk is not None else None, _lib.stream_ptr()), "fpc_encode")
        if qtz:
            cb_tot = cb.split_hist(hist.cpu().numpy().astype(np.float64))  # (synchronises: the launch has reported)
            self.check()
        else:
            cb_tot = [0, 0, 0, 0, 0]  # wavernn.py:189
        out = (c_in, r, r_qtz, r_under, ind1, ind2, cb_tot)
        return out + (idx,) if return_indices else out

    def decode_indices(self, cfg, idx, pitch):
        """Receiver side of `encoder(..., qtz=True, return_indices=True)` (the reference's
        own `decoder`, wavernn.py:367-379, is dead code): c_in (B, L, 20) rebuilt from the symbols `idx`
        (B, L, 4) and the pitch columns (B, L, 2) alone -- bit-identical to the encoder's c_in."""
        h = self._handle()
        dev = self.device
        idx = torch.as_tensor(idx).to(dev, torch.int32).contiguous()
        pitch = torch.as_tensor(pitch).to(dev, torch.float32).contiguous()
        B, L, _ = idx.shape
        cb = load_codebooks(cfg["cb_path"], cfg["scl_cb_path"], cfg.get("bl_cb_path") or None,
                            cfg.get("bl_scl_cb_path") or None)
        c_out = torch.empty(B, L, 20, device=dev)
        self._pcheck(_lib.lib().fpc_decode_features(h, cb.handle, pitch.data_ptr(), idx.data_ptr(), B, L,
                                                    c_out.data_ptr(), _lib.stream_ptr()), "fpc_decode_features")
        return c_out
