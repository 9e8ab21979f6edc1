#!/usr/bin/env python3
"""Long second opinion on the vocoder oracle (parity unpinned: xiph/LPCNet is not in the reference).

The float64 numpy restatement of the published algorithm (tests/test_vocoder_second_opinion.py: Keras layer
semantics, libm tanh/exp, no table activations, no canonical orders; here with the embedding x kernel products
and a CSR recurrent matrix only to make 48 000 steps affordable -- identical mathematics in float64) follows the
oracle's own trace of a whole 3-second utterance (teacher forcing: the chain is chaotic, one flipped draw would
end any sample-for-sample comparison) for each of the three kernel-variant weight sets, half of the frames voiced,
and counts the draws that differ.  A draw can only differ where u * S lands within float32 rounding of a CDF step.

    python tools/second_opinion_long.py [frames]      # CPU only; ~1 min per weight set at 300 frames
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def run(density, T, seed, utt0):
    import scipy.sparse as sp
    import fpcodec_amd
    from oracle import oracle as O
    import test_vocoder_second_opinion as so
    synth = fpcodec_amd.synth
    w = synth.lpcnet_weights(density=density)
    W = {k: np.asarray(v, np.float64) for k, v in w.items()}
    feat = synth.vocoder_features_raw(1, T, utt0=utt0)[0]
    rng = np.random.default_rng(utt0)
    voiced = rng.permutation(T) < (T + 1) // 2
    feat[:, 19] = np.where(voiced, rng.uniform(0.4, 0.95, T), rng.uniform(-0.4, 0.3, T)).astype(np.float32)
    feat[:, 20:] = O.ceps2lpc(feat[:, :20])[0]
    orc = O.LPCNet(w)
    pcm, exc_o, pcm_f = orc.synthesize(feat, seed, trace=True)
    cf = so._condition(w, feat)
    cond_err = float(np.abs(cf - orc.condition(feat)).max())
    EK = [W["embed_sig"] @ W["gru_a_kernel"][128 * s:128 * (s + 1)] for s in range(3)]  # float64 products
    cfa = cf @ W["gru_a_kernel"][384:] + W["gru_a_bias"][0]
    cfb = cf @ W["gru_b_kernel"][384:] + W["gru_b_bias"][0]
    Ra = sp.csr_matrix(W["gru_a_recurrent"].T)  # (1152, 384)
    Kb = W["gru_b_kernel"][:384].T.copy()
    Rb = W["gru_b_recurrent"].T.copy()
    Mk = W["md_kernel"]
    v = np.arange(256)
    nodes = np.zeros((256, 8), np.int64)
    bits = np.zeros((256, 8), bool)
    node = np.ones(256, np.int64)
    for l in range(8):
        b = (v >> (7 - l)) & 1
        nodes[:, l], bits[:, l] = node, b == 1
        node = 2 * node + b
    sig = so._sig
    s1, s2 = np.zeros(384), np.zeros(16)
    hist = np.zeros(T * 160 + 16)
    exc_prev, diff, far, n = 128, 0, 0, 0
    margins = []  # of the draws that differ: distance of u * S from the nearest CDF step, relative to S
    lib = O.lib()
    for t in range(17, T * 160):
        fr = t // 160
        a = feat[fr, 20:].astype(np.float64)
        pred = -float(a @ hist[16 + t - 1 - np.arange(16)])
        e_sig, e_pred = so._lin2ulaw(hist[16 + t - 1]), so._lin2ulaw(pred)
        gi = EK[0][e_sig] + EK[1][e_pred] + EK[2][exc_prev] + cfa[fr]
        gh = Ra @ s1 + W["gru_a_bias"][1]
        z = sig(gi[:384] + gh[:384])
        r = sig(gi[384:768] + gh[384:768])
        nn = np.tanh(gi[768:] + r * gh[768:])
        s1 = z * s1 + (1.0 - z) * nn
        gi = Kb @ s1 + cfb[fr]
        gh = Rb @ s2 + W["gru_b_bias"][1]
        z = sig(gi[:16] + gh[:16])
        r = sig(gi[16:32] + gh[16:32])
        nn = np.tanh(gi[32:] + r * gh[32:])
        s2 = z * s2 + (1.0 - z) * nn
        t2 = np.tanh(np.einsum("jic,i->jc", Mk, s2) + W["md_bias"])
        q = sig((W["md_factor"] * t2).sum(1))
        p = np.where(bits, q[nodes], 1.0 - q[nodes]).prod(1)
        p = p * p ** max(0.0, 1.5 * float(feat[fr, 19]) - 0.5)
        p = p / (1e-18 + p.sum())
        p = np.maximum(p - 0.002, 0.0)
        c = np.cumsum(p / (1e-8 + p.sum()))
        u = lib.orc_philox_uniform(seed, t)
        mine = min(int(np.sum(c <= u * c[-1])), 255)
        n += 1
        d = abs(mine - int(exc_o[t]))
        diff += d != 0
        if d:
            margins.append(float(np.abs(c - u * c[-1]).min() / c[-1]))
        far += d > 1
        exc_prev = int(exc_o[t])
        hist[16 + t] = float(pcm_f[t])
    return dict(density=density, frames=T, voiced_frames=int(voiced.sum()), samples=n, differ=diff, differ_by_more_than_one=far,
                margins_of_differing_draws=margins, condition_max_abs_err=cond_err)


RECORD = os.path.join(ROOT, "profiles", "second_opinion_long.txt")


def write_record(rows):
    """profiles/second_opinion_long.txt, regenerated (by `python tools/second_opinion_long.py` and by
    tests/test_vocoder_second_opinion.py::test_three_second_trace_all_kernel_variant_weight_sets)"""
    with open(RECORD, "w") as f:
        f.write("float64 numpy restatement of the published LPCNet algorithm, teacher-forced along the vocoder oracle's own trace,\n"
                "half of the frames voiced, per weight set (the three sets select the three decode-kernel instances 208 / 408 /\n"
                "1616); regenerated by tools/second_opinion_long.py and by the CPU test suite, never edited by hand:\n")
        for r in rows:
            r = dict(r)
            r["rate"] = r["differ"] / r["samples"]
            f.write(repr(r) + "\n")


if __name__ == "__main__":
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    rows = []
    for k, dens in enumerate([(0.02, 0.02, 0.10), (0.05, 0.05, 0.20), (0.05, 0.05, 0.26)]):
        t0 = time.time()
        r = run(dens, T, 777 + k, 60 + k)
        rows.append(r)
        print(dict(r, rate=r["differ"] / r["samples"]), f"{time.time() - t0:.0f} s", flush=True)
    if T == 300:
        write_record(rows)
