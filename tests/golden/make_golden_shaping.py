"""Golden G9: the reference's own statement of the vocoder's pdf shaping and of the pitch-period index.

Run in the build container only (needs /root/reference, which never travels):

    python tests/golden/make_golden_shaping.py

* `sample_mu_prob(p, feat)` (src/train.py:79-92) is compiled from the reference's file ALONE (the module itself
  cannot be imported: sacred, private datasets, a broken `from modules import`), its source text located with
  `ast`; its `return` is widened at run time to hand back the normalised pdf next to the arg-max.  The function
  normalises by `np.sum(p)` over the WHOLE (256, L) array, so it states the per-sample shaping of the vocoder only
  when the array holds ONE pdf: every pdf is handed over on its own, in column 0 of a (256, 160) block of zeros
  (one frame, `feat` of one frame).
* the period index (src/synthesis.py:103): the right-hand side of that assignment, evaluated as the reference
  wrote it on a seeded tensor `c`.
Only outputs are stored; the inputs are regenerated from the seeds by `shaping_inputs()` / `period_inputs()`
below, which the tests import.
"""
import ast
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
NCOL = 160 * 64  # 10 240 pdfs: 64 frames of 160 samples


def shaping_inputs(seed=9001):
    """(p (256, NCOL) float32 pdfs of varying peakedness, feat (1, 20, 64) with row 19 = pitch correlation)"""
    rng = np.random.default_rng(seed)
    logits = rng.normal(size=(256, NCOL)) * rng.uniform(0.5, 6.0, size=(1, NCOL))
    centre = rng.integers(0, 256, size=NCOL)
    logits -= 0.002 * rng.uniform(0, 4, size=(1, NCOL)) * (np.arange(256)[:, None] - centre[None, :]) ** 2
    e = np.exp(logits - logits.max(0, keepdims=True))
    p = (e / e.sum(0, keepdims=True)).astype(np.float32)
    feat = np.zeros((1, 20, NCOL // 160), np.float32)
    corr = rng.uniform(-0.4, 0.95, size=NCOL // 160).astype(np.float32)
    corr[::7] = np.float32(1.0 / 3.0)  # exponent exactly at its knee (1.5 c - .5 = 0 up to rounding)
    corr[3::11] = np.float32(0.2)      # unvoiced: no sharpening
    feat[0, 19, :] = corr
    return p, feat


def period_inputs(seed=9002):
    rng = np.random.default_rng(seed)
    c = np.zeros((2, 19, 36), np.float32)
    P = rng.integers(40, 256, size=(2, 19))
    c[:, :, 18] = ((P - 100.1) / 50.0).astype(np.float32)  # the feature LPCNet stores for an integer period P
    c[0, :4, 18] = np.float32([-1.2, 3.1, 0.0, 2.0])      # arbitrary values as well
    return c


def _function_source(path, name):
    src = open(path).read()
    for node in ast.parse(src).body:
        if isinstance(node, ast.FunctionDef) and node.name == name:
            return node
    raise KeyError(name)


def main():
    import torch
    # ---- sample_mu_prob, widened to return (arg-max, normalised pdf) ----
    fn = _function_source("/root/reference/src/train.py", "sample_mu_prob")
    ret = fn.body[-1]
    assert isinstance(ret, ast.Return)
    fn.body[-1] = ast.Return(value=ast.Tuple(elts=[ret.value, ast.Name(id="p", ctx=ast.Load())], ctx=ast.Load()))
    mod = ast.Module(body=[fn], type_ignores=[])
    ast.fix_missing_locations(mod)
    ns = {"np": np}
    exec(compile(mod, "<reference src/train.py:sample_mu_prob>", "exec"), ns)
    p, feat = shaping_inputs()
    exc = np.zeros(NCOL, np.int64)
    pn = np.zeros((256, NCOL), np.float64)
    for k in range(NCOL):
        blk = np.zeros((256, 160), np.float32)
        blk[:, 0] = p[:, k]
        f1 = np.zeros((1, 20, 1), np.float32)
        f1[0, 19, 0] = feat[0, 19, k // 160]
        e_k, p_k = ns["sample_mu_prob"](blk, f1)
        exc[k], pn[:, k] = e_k[0], p_k[:, 0]
    # ---- period index: right-hand side of `periods = ...` in synthesis.py ----
    tree = ast.parse(open("/root/reference/src/synthesis.py").read())
    rhs = None
    for node in ast.walk(tree):
        if isinstance(node, ast.Assign) and len(node.targets) == 1 and getattr(node.targets[0], "id", "") == "periods":
            rhs = node.value
    assert rhs is not None
    expr = ast.Expression(body=rhs)
    ast.fix_missing_locations(expr)
    c = torch.from_numpy(period_inputs())
    periods = eval(compile(expr, "<reference src/synthesis.py:periods>", "eval"), {"c": c, "torch": torch, "device": "cpu"})
    out = os.path.join(HERE, "g9_shaping.npz")
    np.savez_compressed(out, exc=exc.astype(np.int16), pn_sub=pn[:, ::16].astype(np.float64),
                        pn_sum=pn.sum(0), periods=periods.numpy().astype(np.int32))
    print("wrote", out, os.path.getsize(out), "bytes; arg-max histogram head", np.bincount(exc)[:4])


if __name__ == "__main__":
    sys.exit(main())
