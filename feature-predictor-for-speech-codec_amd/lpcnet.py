"""LPCNet-style vocoder front-end.  The reference calls xiph/LPCNet's
`training_tf2/test_lpcnet.py [model] [features] [out.pcm]` (README.md:47); the same
three-argument CLI is `python -m fpcodec_amd.lpcnet` (see main()).  Model file = `.npz` with
the Keras-layout arrays named in _lib.LPCNET_KEYS; feature file = raw float32 (LPCNet side)
or `.npy` (this repo's side) of (frames, 36); output = raw little-endian int16, 16 kHz."""
import ctypes as C
import sys

import numpy as np
import torch

from . import _lib

FRAME = 160
NB_FEATURES = 36


class LPCNet:
    def __init__(self, weights):
        _lib.require_gpu()
        self.w = {}
        for k in _lib.LPCNET_KEYS:
            a = np.ascontiguousarray(np.asarray(weights[k], dtype=np.float32))
            if tuple(a.shape) != _lib.LPCNET_SHAPES[k]:
                raise ValueError(f"{k}: shape {a.shape} != {_lib.LPCNET_SHAPES[k]}")
            self.w[k] = a
        s = _lib.LpcnetWeights(*[self.w[k].ctypes.data for k in _lib.LPCNET_KEYS])
        h = C.c_void_p()
        _lib.check(_lib.lib().fpc_lpcnet_create(C.byref(s), C.byref(h)), "fpc_lpcnet_create")
        self.handle = h
        self._ws = None

    def __del__(self):
        try:
            if self.handle is not None:
                _lib.lib().fpc_lpcnet_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    @classmethod
    def load(cls, path):
        with np.load(path) as z:
            return cls({k: z[k] for k in _lib.LPCNET_KEYS})

    def save(self, path):
        np.savez(path, **self.w)

    def _workspace(self, B, T):
        need = int(_lib.lib().fpc_lpcnet_workspace_bytes(self.handle, B, T))
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device="cuda")
        return self._ws

    def synthesize(self, features, seeds, out=None):
        """features (B,T,36) float32 cuda/cpu, seeds (B,) uint64/int64 -> pcm (B,T*160) int16 (cuda)"""
        f = torch.as_tensor(features).to("cuda", torch.float32).contiguous()
        B, T, nf = f.shape
        assert nf == NB_FEATURES
        sd = torch.as_tensor(np.asarray(seeds).astype(np.int64) if not torch.is_tensor(seeds) else seeds)
        sd = sd.to("cuda", torch.int64).contiguous()
        assert sd.numel() == B
        pcm = out if out is not None else torch.empty(B, T * FRAME, dtype=torch.int16, device="cuda")
        ws = self._workspace(B, T)
        _lib.check(_lib.lib().fpc_lpcnet_synthesize(self.handle, f.data_ptr(), B, T, sd.data_ptr(),
                                                    pcm.data_ptr(), ws.data_ptr(), _lib.stream_ptr()),
                   "fpc_lpcnet_synthesize")
        return pcm

    def condition(self, features):
        f = torch.as_tensor(features).to("cuda", torch.float32).contiguous()
        B, T, _ = f.shape
        cf = torch.empty(B, T, 128, device="cuda")
        ws = self._workspace(B, T)
        _lib.check(_lib.lib().fpc_lpcnet_condition(self.handle, f.data_ptr(), B, T, cf.data_ptr(),
                                                   ws.data_ptr(), _lib.stream_ptr()), "fpc_lpcnet_condition")
        return cf

    def set_chunk_frames(self, frames):
        """frames per pass of synthesize (0: the whole utterance at once); a chunked pass needs a workspace that does
        not grow with T and gives the same samples (fpc_lpcnet_set_chunk_frames)"""
        _lib.check(_lib.lib().fpc_lpcnet_set_chunk_frames(self.handle, int(frames)), "fpc_lpcnet_set_chunk_frames")

    def set_pairing(self, mode):
        """utterances per workgroup of synthesize: 0 two when B exceeds the device's compute units (default), 1 always
        two, -1 never (fpc_lpcnet_set_pairing); the PCM does not depend on it"""
        _lib.check(_lib.lib().fpc_lpcnet_set_pairing(self.handle, int(mode)), "fpc_lpcnet_set_pairing")

    def last_streams_per_workgroup(self):
        return int(_lib.lib().fpc_lpcnet_last_streams_per_workgroup(self.handle))

    def workspace_bytes(self, B, T):
        return int(_lib.lib().fpc_lpcnet_workspace_bytes(self.handle, B, T))

    def last_decode_ms(self):
        return float(_lib.lib().fpc_lpcnet_last_decode_ms(self.handle))

    def kernel_variant(self):
        """diagnostic: decode-kernel instance selected by the sparsity pattern (208, 408 or 1616)"""
        return int(_lib.lib().fpc_lpcnet_kernel_variant(self.handle))


def read_features(path):
    """(frames,36) float32 from raw .f32 (LPCNet dumps, data_preprocess/write_small_files.py:18-24)
    or .npy ((1,L,36) / (L,36), synthesis_qtz.py:142,160)"""
    if path.endswith(".npy"):
        a = np.load(path).astype(np.float32)
        return a.reshape(-1, NB_FEATURES)
    a = np.fromfile(path, dtype=np.float32)
    return a[: a.size // NB_FEATURES * NB_FEATURES].reshape(-1, NB_FEATURES)


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    if len(argv) != 3:
        print("usage: python -m fpcodec_amd.lpcnet [Saved_Model.npz] [Generated_Feature_Path] [Synthesized_Sample_Path]")
        return 2
    model = LPCNet.load(argv[0])
    feats = read_features(argv[1])
    pcm = model.synthesize(feats[None], np.array([0], np.uint64))[0].cpu().numpy()
    pcm[17:].astype("<i2").tofile(argv[2])  # test_lpcnet.py never writes the first order+1 samples
    return 0


if __name__ == "__main__":
    sys.exit(main())
