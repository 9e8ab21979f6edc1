"""Codebook-training timings on the GPU box: one k-means update at production size (HIP events), the
CPU oracle on a bounded sample beside it, and a short full vq_train.
    python tools/time_cb.py [nv] [entries]"""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch, fpcodec_amd
from fpcodec_amd import cb_func
from oracle import oracle as O
synth = fpcodec_amd.synth
nv = int(sys.argv[1]) if len(sys.argv) > 1 else 400_000
e = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
data = synth.cb_training_vectors(nv, seed_offset=5)
cb = data[::nv // e][:e].astype(np.float64) + 1e-3
d = torch.from_numpy(data).cuda()
cb_func.update(d, cb, e)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record()
for _ in range(5):
    new = cb_func.update(d, cb, e)
ev[1].record()
torch.cuda.synchronize()
ms = ev[0].elapsed_time(ev[1]) / 5
flop = nv * e * 50.0
print(f"update nv={nv} entries={e}: {ms:.2f} ms per call (incl. host round trip of the codebook), "
      f"{flop / ms / 1e9:.2f} TFLOP/s f64 in the distance search (50 flop per vector-entry pair)")
ns = 20_000
t0 = time.time()
ref = O.cb_update(data[:ns], cb, e)
dt = time.time() - t0
print(f"CPU oracle (1 core) nv={ns} entries={e}: {dt * 1e3:.0f} ms -> {dt * nv / ns * 1e3:.0f} ms scaled to nv={nv}; "
      f"GPU/CPU = {dt * nv / ns * 1e3 / ms:.0f}x")
assert np.array_equal(cb_func.update(data[:ns], cb, e), ref)
np.random.seed(1)
t0 = time.time()
c = cb_func.vq_train(d, np.zeros((128, 17)), 128)
torch.cuda.synchronize()
print(f"vq_train to 128 entries on nv={nv}: {time.time() - t0:.2f} s ({127 * 4 + 10} updates)")
