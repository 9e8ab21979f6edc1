"""Predictor training step on the GPU: the live branch of the reference's
`train(...)` loop, src/train_frame.py:53-120 (`batch_idx <= 10`):

    feat_out, _, _ = model(feat)                                  # teacher-forced forward, (B, L, 18)
    loss = mseloss(feat_out[:, :-1, :], feat[:, 1:, :fc_units])   # predict the next frame
    optimizer.zero_grad(); loss.backward(); optimizer.step()      # optim.Adam(model.parameters(), lr)

as one call into libfpcodec.so (`fpc_trainer_step`: forward with kept activations, BPTT, weight gradients
on the matrix cores, Adam).  The trainer updates the device weights of the `Wavernn` it was built on, so
`model.forward` / `model.encoder` see the new weights at once; `model.state_dict()` is refreshed by
`Trainer.sync()` (checkpoints keep the reference's format, utils.py:127-146).  The reference's later
branch (`mask_enc`, batch_idx > 10) is dead code and not reproduced."""
import ctypes as C
import weakref

import numpy as np
import torch

from . import _lib
from .wavernn import _KEYS, _ParameterList


class Trainer:
    def __init__(self, model, lr=1e-4, max_batch=100, max_frames=150):
        """defaults = train_frame.py:188-192 (batch_size 100, chunks 10 x 15 frames, learning_rate 1e-4)"""
        self.model, self.lr = model, float(lr)
        self._t = C.c_void_p()
        self._h = model._handle()  # the trainer works on this predictor handle's device weights
        _lib.check(_lib.lib().fpc_trainer_create(self._h, int(max_batch), int(max_frames), C.byref(self._t)),
                   "fpc_trainer_create")
        model._trainer = weakref.ref(self)

    def __del__(self):
        try:
            if self._t:
                _lib.lib().fpc_trainer_destroy(self._t)
                self._t = None
        except Exception:
            pass

    def _check_current(self):
        # the C side keeps the predictor alive (reference count), so a stale trainer cannot touch freed memory;
        # it would, however, train or export weights the model no longer uses: refuse
        if self.model._h is not self._h:
            raise _lib.FpcError("Trainer: the model's weights were reloaded (load_state_dict) after this trainer was "
                                "built; create a new Trainer")

    def step(self, feat):
        """one optimisation step on feat (B, L, 20) normalised frames; returns the loss of this step"""
        self._check_current()
        f = torch.as_tensor(feat).to("cuda", torch.float32).contiguous()
        B, L, Cc = f.shape
        assert Cc == self.model.in_features
        loss = C.c_float()
        _lib.check(_lib.lib().fpc_trainer_step(self._t, f.data_ptr(), B, L, self.lr, C.byref(loss), _lib.stream_ptr()),
                   "fpc_trainer_step")
        self.model._dirty = True
        return float(loss.value)

    def _export(self, what):
        self._check_current()
        shp = self.model.shapes()
        arrs = [np.zeros(shp[k], np.float32) for k in _KEYS]
        w = _lib.PredictorWeights(self.model.in_features, self.model.gru_units1, self.model.gru_units2,
                                  self.model.fc_units, *[a.ctypes.data for a in arrs])
        _lib.check(_lib.lib().fpc_trainer_export(self._t, what, C.byref(w)), "fpc_trainer_export")
        return dict(zip(_KEYS, arrs))

    def gradients(self):
        """gradients of the last step, torch layouts"""
        return self._export(1)

    def sync(self):
        """pull the updated weights into model.state_dict() (e.g. before utils.checkpoint)"""
        new = self._export(0)
        for k in _KEYS:
            self.model._sd[k] = new[k]
        self.model._dirty = False
        return self.model


class Adam:
    """`optimizer = optim.Adam(model.parameters(), lr=cfg['learning_rate'])` (train_frame.py:250) for this
    package's `Wavernn`: the update itself runs inside `fpc_trainer_step` (torch's single-tensor Adam, defaults
    betas (0.9, 0.999), eps 1e-8), so `zero_grad()` / `step()` are the no-ops that keep the reference's loop shape."""

    def __init__(self, params, lr=1e-3, max_batch=100, max_frames=150):
        if not isinstance(params, _ParameterList):
            raise TypeError("Adam: pass model.parameters() of a fpcodec_amd Wavernn")
        self.trainer = Trainer(params.model, lr=lr, max_batch=max_batch, max_frames=max_frames)

    def zero_grad(self):
        pass

    def step(self):
        pass


def train(model, optimizer, train_loader, epoch, model_label=None, padding=False, packing=False, fc_units=18,
          normalize=True, keep_rate=0.3, debugging=False):
    """The reference's `train(...)` (train_frame.py:53-120), live branch only: every batch takes the teacher-forced
    step of `batch_idx <= 10` (the later `mask_enc` branch raises AttributeError in the reference).
    `train_loader` yields (sample_name, x, c, nm_c) with c / nm_c (B, 19-frame-window layout flattened to frames, 36);
    frames [2:-2] and the first 20 columns are used (:66-69).  Returns the summed loss of the epoch."""
    if padding or packing:
        raise NotImplementedError("production path is unpadded, unpacked (train_frame.py:193-195)")
    model.train()
    epoch_loss = 0.0
    for batch_idx, (sample_name, x, c, nm_c) in enumerate(train_loader):
        src = nm_c if normalize else c
        feat = torch.as_tensor(src)[:, 2:-2, :-16].to(torch.float)
        epoch_loss += optimizer.trainer.step(feat)
        optimizer.zero_grad()
        optimizer.step()
        if debugging:
            break
    return epoch_loss


def evaluate(model, test_loader, padding=False, packing=False, fc_units=18, normalize=True, keep_rate=0.3,
             debugging=False):
    """`evaluate(...)` (train_frame.py:122-160), live branch: teacher-forced MSE of the next frame, no update"""
    model.eval()
    epoch_loss = 0.0
    for batch_idx, (sample_name, x, c, nm_c) in enumerate(test_loader):
        src = nm_c if normalize else c
        feat = torch.as_tensor(src)[:, 2:-2, :-16].to("cuda", torch.float)
        out, _, _ = model(feat)
        d = out[:, :-1, :] - feat[:, 1:, :fc_units]
        epoch_loss += float((d * d).mean().item())
        if debugging:
            break
    return epoch_loss
