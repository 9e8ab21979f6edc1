"""Two utterances per workgroup (k_decode2): parity against the oracle and against k_decode, then timing.
    python tools/pair_probe.py [--no-parity] [--T frames] [--B list]
Prints one line per case; exits non-zero on any PCM difference."""
import argparse
import hashlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import fpcodec_amd
from fpcodec_amd.lpcnet import LPCNet
from oracle import oracle as O

ap = argparse.ArgumentParser()
ap.add_argument("--no-parity", action="store_true")
ap.add_argument("--T", type=int, default=100)
ap.add_argument("--B", type=str, default="256,512,1024")
ap.add_argument("--voiced", type=float, default=-1.0, help="fraction of voiced frames (default: the synthetic material's own)")
args = ap.parse_args()
synth = fpcodec_amd.synth
bad = 0


def feats(B, T, utt0=0):
    f = synth.vocoder_features_raw(B, T, utt0=utt0)
    f[:, :, 20:] = O.ceps2lpc(f.reshape(-1, 36)[:, :20])[0].reshape(B, T, 16)
    return f


if not args.no_parity:
    for density, variant in (((0.05, 0.05, 0.20), 408), ((0.02, 0.02, 0.10), 208)):
        w = synth.lpcnet_weights(density=density)
        voc = LPCNet(w)
        assert voc.kernel_variant() == variant
        orc = O.LPCNet(w)
        B, T = 5, 5
        f = feats(B, T, utt0=11)
        f[1, :, 19] = 0.9          # utterance 1 voiced throughout, 0 unvoiced: a pair with one voiced member
        f[2, 1:3, 19] = 0.8        # a pair whose members are voiced in different frames
        f[3, 2:4, 19] = 0.7
        f[0, :, 19] = -0.4
        sd = synth.seeds(B, utt0=11)
        ref = np.stack([orc.synthesize(f[b], int(sd[b])) for b in range(B)])
        for mode in (-1, 1):
            voc.set_pairing(mode)
            pcm = voc.synthesize(f, sd).cpu().numpy()
            spw = voc.last_streams_per_workgroup()
            ok = np.array_equal(pcm, ref)
            bad += not ok
            first = [int(np.nonzero(pcm[b] != ref[b])[0][0]) if (pcm[b] != ref[b]).any() else -1 for b in range(B)]
            print(f"variant {variant} pairing {mode:+d} ({spw}/wg) B={B} T={T}: {'bit-identical to the oracle' if ok else 'DIFFERS first ' + str(first)}", flush=True)
        # chunked pass, paired
        voc.set_pairing(1)
        voc.set_chunk_frames(2)
        pcm = voc.synthesize(f, sd).cpu().numpy()
        ok = np.array_equal(pcm, ref)
        bad += not ok
        print(f"variant {variant} pairing +1 chunked(2) : {'bit-identical' if ok else 'DIFFERS'}", flush=True)
        voc.set_chunk_frames(0)
    if bad:
        sys.exit(1)

# timing: the production model, B utterances of T frames
w = synth.lpcnet_weights()
voc = LPCNet(w)
T = args.T
nu = 16
raw = feats(nu, T, utt0=500)
if args.voiced >= 0:
    rng = np.random.default_rng(5)
    raw[:, :, 19] = np.where(rng.random((nu, T)) < args.voiced, 0.9, -0.4)
for B in [int(x) for x in args.B.split(",")]:
    f = torch.from_numpy(np.tile(raw, (B // nu + 1, 1, 1))[:B].copy()).cuda()
    sd = synth.seeds(B, utt0=500)
    res = {}
    for mode in (-1, 0):  # rounds of k_decode; the default policy (k_decode2, + a round of k_decode where that is cheaper)
        voc.set_pairing(mode)
        ms = []
        for _ in range(3):
            pcm = voc.synthesize(f, sd)
            torch.cuda.synchronize()
            ms.append(voc.last_decode_ms())
        res[mode] = (min(ms[1:]), hashlib.sha1(pcm.cpu().numpy().tobytes()).hexdigest()[:12])
    n = B * (T * 160 - 17)
    same = res[-1][1] == res[0][1]
    bad += not same
    print(f"B={B:5d} T={T}: one/wg {res[-1][0]:8.2f} ms ({n / res[-1][0] / 1e3:7.1f} M samples/s)   default policy {res[0][0]:8.2f} ms "
          f"({n / res[0][0] / 1e3:7.1f} M samples/s)   ratio {res[-1][0] / res[0][0]:.3f}   pcm {'same' if same else 'DIFFERENT'}", flush=True)
sys.exit(1 if bad else 0)
