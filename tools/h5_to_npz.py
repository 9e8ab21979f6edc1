#!/usr/bin/env python3
"""Keras LPCNet checkpoint (.h5) -> the `.npz` model file of `python -m fpcodec_amd.lpcnet` (README.md:47's
`[Saved_Model]` argument of xiph/LPCNet `training_tf2/test_lpcnet.py`).

    python tools/h5_to_npz.py lpcnet_model.h5 lpcnet_model.npz

Needs h5py (absent from the build image: run it where the checkpoint was trained).  The mapping itself,
`from_keras_named`, is plain numpy and unit-tested on a synthetic Keras-named dict (tests/test_host_cpu.py).

Keras layer (xiph/LPCNet training_tf2/lpcnet.py, tree-pdf era)   ->  fpc_lpcnet_weights field (include/fpcodec.h)
    embed_pitch    Embedding(256, 64)         embeddings          ->  embed_pitch      (256, 64)
    feature_conv1  Conv1D(128, 3, 'same')     kernel, bias        ->  conv1_kernel (3, 84, 128), conv1_bias (128)
    feature_conv2  Conv1D(128, 3, 'same')     kernel, bias        ->  conv2_kernel (3, 128, 128), conv2_bias (128)
    feature_dense1 Dense(128, tanh)           kernel, bias        ->  dense1_kernel (128, 128), dense1_bias (128)
    feature_dense2 Dense(128, tanh)           kernel, bias        ->  dense2_kernel (128, 128), dense2_bias (128)
    embed_sig      Embedding(256, 128)        embeddings          ->  embed_sig        (256, 128)
    gru_a          GRU(384, reset_after)      kernel, recurrent_kernel, bias
                                                                  ->  gru_a_kernel (512, 1152) rows [sig|pred|exc|cfeat],
                                                                      gru_a_recurrent (384, 1152), gru_a_bias (2, 1152)
    gru_b          GRU(16, reset_after)       kernel, recurrent_kernel, bias
                                                                  ->  gru_b_kernel (512, 48) rows [gru_a state 384|cfeat 128],
                                                                      gru_b_recurrent (16, 48), gru_b_bias (2, 48)
    dual_fc        MDense(256, channels=2)    kernel, bias, factor ->  md_kernel (256, 16, 2), md_bias (256, 2), md_factor (256, 2)
Gate column order [z | r | h] and the `reset_after` bias rows (input, recurrent) are Keras's and are kept as they are.
A CuDNNGRU checkpoint stores the two bias rows as one (2*3*units,) vector: it is reshaped to (2, 3*units).
The recurrent matrix of gru_a must be block-sparse (8x4 blocks + diagonal, LPCNet's Sparsify) within the capacity of
the register-resident layout; `fpc_lpcnet_create` refuses a denser one with FPC_ERR_CAPACITY.
"""
import sys

import numpy as np

# (npz key, Keras layer, weight name, shape)
MAPPING = [
    ("embed_pitch", "embed_pitch", "embeddings", (256, 64)),
    ("conv1_kernel", "feature_conv1", "kernel", (3, 84, 128)), ("conv1_bias", "feature_conv1", "bias", (128,)),
    ("conv2_kernel", "feature_conv2", "kernel", (3, 128, 128)), ("conv2_bias", "feature_conv2", "bias", (128,)),
    ("dense1_kernel", "feature_dense1", "kernel", (128, 128)), ("dense1_bias", "feature_dense1", "bias", (128,)),
    ("dense2_kernel", "feature_dense2", "kernel", (128, 128)), ("dense2_bias", "feature_dense2", "bias", (128,)),
    ("embed_sig", "embed_sig", "embeddings", (256, 128)),
    ("gru_a_kernel", "gru_a", "kernel", (512, 1152)), ("gru_a_recurrent", "gru_a", "recurrent_kernel", (384, 1152)),
    ("gru_a_bias", "gru_a", "bias", (2, 1152)),
    ("gru_b_kernel", "gru_b", "kernel", (512, 48)), ("gru_b_recurrent", "gru_b", "recurrent_kernel", (16, 48)),
    ("gru_b_bias", "gru_b", "bias", (2, 48)),
    ("md_kernel", "dual_fc", "kernel", (256, 16, 2)), ("md_bias", "dual_fc", "bias", (256, 2)),
    ("md_factor", "dual_fc", "factor", (256, 2)),
]


def _find(named, layer, weight):
    """value of `<...>/<layer>/<...>/<weight>:0` in a dict of Keras weight paths (any nesting: `gru_a/gru_cell/kernel:0`,
    `model_weights/gru_a/gru_a/kernel:0`); `recurrent_kernel` must not match a search for `kernel`"""
    hits = []
    for name, val in named.items():
        parts = name.replace(":0", "").split("/")
        if layer in parts[:-1] and parts[-1] == weight:
            hits.append((name, val))
    if len(hits) != 1:
        raise KeyError(f"{layer}/{weight}: {len(hits)} matching entries {[h[0] for h in hits]}")
    return hits[0][1]


def from_keras_named(named):
    """dict of Keras weight paths -> dict of the 19 arrays `fpcodec_amd.lpcnet.LPCNet` takes (float32, checked shapes)"""
    out = {}
    for key, layer, weight, shape in MAPPING:
        a = np.asarray(_find(named, layer, weight), dtype=np.float32)
        if weight == "bias" and layer.startswith("gru") and a.ndim == 1:
            a = a.reshape(2, -1)  # CuDNNGRU: input and recurrent bias rows stored back to back
        if tuple(a.shape) != shape:
            raise ValueError(f"{layer}/{weight}: shape {tuple(a.shape)}, expected {shape} "
                             "(a different LPCNet generation? this build follows the 384/16-unit tree-pdf model)")
        out[key] = np.ascontiguousarray(a)
    return out


def read_h5(path):
    import h5py
    named = {}
    with h5py.File(path, "r") as f:
        root = f["model_weights"] if "model_weights" in f else f
        root.visititems(lambda n, o: named.__setitem__(n, np.array(o)) if isinstance(o, h5py.Dataset) else None)
    return named


def main(argv):
    if len(argv) != 2:
        print(__doc__)
        return 2
    np.savez(argv[1], **from_keras_named(read_h5(argv[0])))
    print("wrote", argv[1])
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
