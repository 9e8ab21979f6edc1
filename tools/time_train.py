"""Training-step timing at the reference's shapes (train_frame.py:188-192: batch 100 x 150 frames)."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch, fpcodec_amd
from fpcodec_amd.train_frame import Trainer
from fpcodec_amd.wavernn import Wavernn
from oracle import oracle as O
synth = fpcodec_amd.synth
m = Wavernn(20, 384, 128, 18); m.load_state_dict(synth.predictor_state_dict())
feat = torch.from_numpy(synth.predictor_features(100, 150, utt0=6000)).cuda()
tr = Trainer(m)
tr.step(feat); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    l = tr.step(feat)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / 5 * 1e3
print(f"train step 100 x 150 frames: {ms:.2f} ms ({100 * 150 / ms * 1e3:.0f} frames/s), loss {l:.6g}")
ref = O.Trainer(synth.predictor_state_dict())
f4 = synth.predictor_features(4, 150, utt0=6000)
t0 = time.time(); ref.step(f4); dt = time.time() - t0
print(f"CPU oracle (1 core) 4 x 150 frames: {dt:.2f} s -> {dt * 25:.1f} s scaled to 100 x 150; GPU/CPU = {dt * 25 * 1e3 / ms:.0f}x")
