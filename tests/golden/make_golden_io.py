"""Golden G11: the reference's own statements of the feature-file layouts and of the WAV normalisation
(SURVEY 8f row 2), each compiled from the reference's file ALONE -- the modules themselves cannot be imported
(sacred, soundfile, private data paths, module-level loops over /media/... files).

Run in the build container only (needs /root/reference):  python tests/golden/make_golden_io.py

* windows of a `.f32` dump: the `features = np.lib.stride_tricks.as_strided(...)` assignment of
  data_preprocess/write_small_files.py:58-64 with its `nb_frames` / `sizeof` lines (:56-57)
* windows of encoded frames: the as_strided assignment of src/generate_qtz_features.py:66-70 (hard-coded
  (10, 19, 36) = (L - 4) // 15 windows for the L = 154 frames that script handles)
* synthesis frames: the statements of `Libri_lpc_data_syn.__getitem__` from the `nb_frames = min(...)` line to
  `nm_feat = feat / self.maxi` (src/datasets/dataset_syn.py:66-97)
* `saveaudio` (src/synthesis_qtz.py:39-50) with `sf.write` replaced by a recorder and `ex.capture` by identity
Only outputs are stored; inputs are regenerated from the seeds by the *_inputs() functions below."""
import ast
import os
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def f32_inputs():
    return np.random.default_rng(9101).normal(size=(15 * 7 + 9) * 36).astype(np.float32)


def enc_inputs():
    return np.random.default_rng(9102).normal(size=(1, 154, 36)).astype(np.float32)


def syn_inputs():
    rng = np.random.default_rng(9103)
    return rng.normal(size=(9, 19, 36)).astype(np.float32), rng.normal(size=(8, 19, 36)).astype(np.float32)


def wave_inputs():
    return np.random.default_rng(9104).normal(0, 250.0, size=(1, 1, 4800)).astype(np.float32)


def _assign_to(tree, name, pred=lambda n: True):
    for node in ast.walk(tree):
        if isinstance(node, ast.Assign) and len(node.targets) == 1 and getattr(node.targets[0], "id", None) == name \
                and pred(node):
            return node
    raise KeyError(name)


def _run(stmts, env, tag):
    mod = ast.Module(body=list(stmts), type_ignores=[])
    ast.fix_missing_locations(mod)
    exec(compile(mod, f"<reference {tag}>", "exec"), env)
    return env


def main():
    import torch
    strided = lambda n: "as_strided" in ast.dump(n.value)
    # ---- write_small_files.py: nb_frames, sizeof, features = as_strided(...) ----
    t = ast.parse(open(f"{REF}/data_preprocess/write_small_files.py").read())
    env = {"np": np, "features": f32_inputs(), "feature_chunk_size": 15, "nb_features": 36}
    _run([_assign_to(t, "nb_frames"), _assign_to(t, "sizeof"), _assign_to(t, "features", strided)], env,
         "write_small_files.py:56-64")
    w_f32 = np.array(env["features"])
    # ---- generate_qtz_features.py: sizeof, all_features = as_strided(...) ----
    t = ast.parse(open(f"{REF}/src/generate_qtz_features.py").read())
    env = {"np": np, "all_features": enc_inputs()}
    _run([_assign_to(t, "sizeof"), _assign_to(t, "all_features", strided)], env, "generate_qtz_features.py:65-70")
    w_enc = np.array(env["all_features"])
    # ---- dataset_syn.py __getitem__, from `nb_frames = min(...)` to `nm_feat = ...` ----
    t = ast.parse(open(f"{REF}/src/datasets/dataset_syn.py").read())
    fn = [n for n in ast.walk(t) if isinstance(n, ast.FunctionDef) and n.name == "__getitem__"][0]
    first = _assign_to(fn, "nb_frames", lambda n: "min" in ast.dump(n.value)).lineno
    last = _assign_to(fn, "nm_feat").lineno
    body = [s for s in fn.body if first <= s.lineno <= last]
    outs = {}
    for chunks in (3, 0, 20):
        f, q = syn_inputs()
        env = {"np": np, "torch": torch, "self": types.SimpleNamespace(chunks=chunks, maxi=24.1),
               "features": torch.from_numpy(f), "qtz_features": torch.from_numpy(q), "nb_frames": 10 ** 9,
               "in_data": np.zeros(2400 * 9, np.float32)}
        _run(body, env, f"dataset_syn.py:{first}-{last}")
        outs[f"nm_{chunks}"] = env["nm_feat"].numpy()
        outs[f"qf_{chunks}"] = env["qtz_feat"].numpy()
    # ---- saveaudio ----
    t = ast.parse(open(f"{REF}/src/synthesis_qtz.py").read())
    fn = [n for n in t.body if isinstance(n, ast.FunctionDef) and n.name == "saveaudio"][0]
    fn.decorator_list = []
    rec = {}
    sf = types.SimpleNamespace(write=lambda name, data, sr, fmt: rec.update(name=name, data=np.array(data), sr=sr, fmt=fmt))
    env = _run([fn], {"np": np, "sf": sf}, "synthesis_qtz.py:39-50")
    env["saveaudio"](torch.from_numpy(wave_inputs()), "LBL", "utt-0001")
    assert rec["sr"] == 16000 and rec["fmt"] == "PCM_16"
    out = os.path.join(HERE, "g11_feature_io.npz")
    np.savez_compressed(out, w_f32=w_f32, w_enc=w_enc, wav=rec["data"], wav_name=np.array(rec["name"]), **outs)
    print("wrote", out, w_f32.shape, w_enc.shape, {k: v.shape for k, v in outs.items()}, rec["name"])


if __name__ == "__main__":
    main()
