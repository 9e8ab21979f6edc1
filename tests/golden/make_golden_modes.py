"""Golden G12: the two behaviours of the reference's Wavernn.encoder that the fused kernels do not cover and the host
loop of fpcodec_amd.wavernn serves frame by frame -- the input-mask mode (wavernn.py:209-211) and a multi-stage
below-threshold codebook (wavernn.py:235-240: quantize_mstage over all stages, cb_tot[4] += the LAST stage's histogram).
Run in the build container only (imports /root/reference through make_golden.import_reference):

    python tests/golden/make_golden_modes.py

Inputs are regenerated on the test side from these seeds: features synth.predictor_features(3, 30, utt0=40); mask
default_rng(1098).random((3, 30, 2, 1)) < 0.4 as float32 (the shape the reference's indexing needs: mask[:, i, 0] must
be (B, 1)); two-stage below-threshold book default_rng(1099).normal(0, (.02, .01), (512, 17)) per stage."""
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402

synth = mg.synth


def modes_inputs():
    feat = synth.predictor_features(3, 30, utt0=40)
    mask = (np.random.default_rng(1098).random((3, 30, 2, 1)) < 0.4).astype(np.float32)
    rng = np.random.default_rng(1099)
    lo2 = np.stack([rng.normal(0, 0.02, (512, 17)), rng.normal(0, 0.01, (512, 17))])
    return feat, mask, lo2


def main():
    torch.set_num_threads(1)
    wavernn, vq_func, _, _, _ = mg.import_reference()
    tmp = tempfile.mkdtemp()
    sd = synth.predictor_state_dict()
    model = wavernn.Wavernn(in_features=20, gru_units1=384, gru_units2=128, fc_units=18)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    model.eval()
    paths = {}
    for k, v in synth.codebooks().items():
        paths[k] = os.path.join(tmp, k + ".npy")
        np.save(paths[k], v)
    feat, mask, lo2 = modes_inputs()
    paths["vq_lo2"] = os.path.join(tmp, "vq_lo2.npy")
    np.save(paths["vq_lo2"], lo2)
    cfg_full = dict(scl_cb_path=paths["scl_hi"], cb_path=paths["vq_hi"], bl_scl_cb_path=paths["scl_lo"], bl_cb_path=paths["vq_lo"])
    cfg_lo2 = dict(cfg_full, bl_cb_path=paths["vq_lo2"])
    g = {}

    def run(tag, cfg, m, qtz):
        with torch.no_grad():
            out = model.encoder(cfg, torch.from_numpy(feat), None if m is None else torch.from_numpy(m), 0.09, 0.28,
                                vq_func.vq_quantize, vq_func.scl_quantize, qtz)
        for n, v in zip(["c_in", "r", "r_qtz", "r_under", "ind1", "ind2"], out[:6]):
            g[f"{tag}_{n}"] = v.numpy().copy()
        for i, h in enumerate(out[6]):
            g[f"{tag}_hist{i}"] = np.asarray(h, dtype=np.float64)
        print(tag, "done")

    run("mask_qtz", cfg_full, mask, True)
    run("mask_raw", cfg_full, mask, False)
    run("lo2", cfg_lo2, None, True)
    np.savez_compressed(os.path.join(HERE, "g12_encoder_modes.npz"), **g)


if __name__ == "__main__":
    main()
