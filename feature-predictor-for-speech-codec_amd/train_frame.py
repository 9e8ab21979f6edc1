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

import numpy as np
import torch

from . import _lib
from .wavernn import _KEYS


class Trainer:
    def __init__(self, model, lr=1e-4, max_batch=100, max_frames=150):
        """defaults = train_frame.py:188-192 (batch_size 100, chunks 10 x 15 frames, learning_rate 1e-4)"""
        self.model, self.lr = model, float(lr)
        self._t = C.c_void_p()
        self._h = model._handle()  # the trainer works on this predictor handle's device weights
        _lib.check(_lib.lib().fpc_trainer_create(self._h, int(max_batch), int(max_frames), C.byref(self._t)),
                   "fpc_trainer_create")

    def __del__(self):
        try:
            if self._t:
                _lib.lib().fpc_trainer_destroy(self._t)
                self._t = None
        except Exception:
            pass

    def step(self, feat):
        """one optimisation step on feat (B, L, 20) normalised frames; returns the loss of this step"""
        if self.model._h is not self._h:
            raise _lib.FpcError("Trainer: the model's weights were reloaded (load_state_dict) after this trainer was "
                                "built; create a new Trainer")
        f = torch.as_tensor(feat).to("cuda", torch.float32).contiguous()
        B, L, Cc = f.shape
        assert Cc == self.model.in_features
        loss = C.c_float()
        _lib.check(_lib.lib().fpc_trainer_step(self._t, f.data_ptr(), B, L, self.lr, C.byref(loss), _lib.stream_ptr()),
                   "fpc_trainer_step")
        return float(loss.value)

    def _export(self, what):
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
        return self.model
