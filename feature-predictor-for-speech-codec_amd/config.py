"""Configuration dict with the reference's key names (src/config.py:14-85).  The
reference drives it through sacred (`python X.py with cfg.key=value`); sacred is not a
dependency here, so `parse_overrides` accepts the same `cfg.key=value` tokens."""
import ast


def default_cfg():
    return {
        # data (src/config.py:16-32)
        "frame_size": 160, "lpcoeffs_N": 16, "sr": 16000, "n_sample_seg": 2400, "n_seg": 15,
        "qtz": True,
        "scl_cb_path": "../codebook/scalar_center_256.npy",
        "cb_path": "../codebook/ceps_vq_codebook_2_1024_large_17.npy",
        "bl_scl_cb_path": "", "bl_cb_path": "",
        "code_dim": 17, "l1": 0.0, "l2": 0.0,
        # predictor hyper-parameters (README.md:44; src/train_frame.py:198-200)
        "gru_units1": 384, "gru_units2": 128, "fc_units": 18, "attn_units": 20,
        "rnn_layers": 2, "bidirectional": False,
        # synthesis (src/config.py:76-83)
        "total_secs": 3, "num_samples": 2, "model_label_f": None, "epoch_f": None, "note": "",
    }


def parse_overrides(tokens, cfg=None):
    """`with cfg.l1=0.09 cfg.qtz=True ...` (README.md:44)"""
    cfg = dict(default_cfg() if cfg is None else cfg)
    for tok in tokens:
        if tok == "with":
            continue
        key, _, val = tok.partition("=")
        key = key[4:] if key.startswith("cfg.") else key
        try:
            cfg[key] = ast.literal_eval(val)
        except (ValueError, SyntaxError):
            cfg[key] = val
    return cfg
