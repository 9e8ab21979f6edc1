"""Stress of the two-track backward kernel: many shapes, repeated, every gradient bit for bit against the row-split kernel."""
import os, sys, time
sys.path.insert(0, '.')
import numpy as np, torch, fpcodec_amd
from fpcodec_amd.train_frame import Trainer
from fpcodec_amd.wavernn import Wavernn
synth = fpcodec_amd.synth
sd0 = synth.predictor_state_dict()

def run(feat, steps):
    m = Wavernn(20, 384, 128, 18); m.load_state_dict(sd0)
    tr = Trainer(m, lr=1e-3, max_batch=feat.shape[0], max_frames=feat.shape[1])
    losses = [tr.step(feat) for _ in range(steps)]
    g = tr.gradients(); tr.sync()
    sd = m.state_dict()
    return [np.float32(losses)] + [g[k] for k in sorted(g)] + [sd[k].numpy() for k in sorted(sd)], m.fallback_groups()

bad = 0
t0 = time.time()
for B, L, reps in ((1, 2, 3), (16, 3, 3), (17, 40, 3), (100, 150, 6), (128, 33, 4), (130, 7, 3), (200, 12, 3), (5, 300, 2)):
    feat = synth.predictor_features(B, L, utt0=7000 + B)
    os.environ["FPC_TRAIN_BWD_ROWSPLIT"] = "1"
    ref, _ = run(feat, 3)
    del os.environ["FPC_TRAIN_BWD_ROWSPLIT"]
    for r in range(reps):
        got, fb = run(feat, 3)
        ok = all(np.array_equal(a, b) and not np.isnan(b).any() for a, b in zip(ref, got))
        bad += not ok
        print(f"B={B} L={L} rep {r}: {'identical' if ok else 'DIFFERENT'} (fallback groups of the last launch: {fb})", flush=True)
print("ALL IDENTICAL" if bad == 0 else f"{bad} DIFFERENT", f"{time.time() - t0:.1f} s")
sys.exit(1 if bad else 0)
