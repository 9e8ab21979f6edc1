"""Receiver (Wavernn.decode_indices, fpc_decode_features) at 128 x 300: time and a hash of the output (FPC_LIB_PATH selects a variant)."""
import sys, os, tempfile, hashlib
sys.path.insert(0, '.')
import numpy as np, torch, fpcodec_amd
from fpcodec_amd.wavernn import Wavernn
synth = fpcodec_amd.synth
d = tempfile.mkdtemp(); c = synth.codebooks(); p = {}
for k, v in c.items():
    p[k] = os.path.join(d, k + '.npy'); np.save(p[k], v)
cfg = dict(scl_cb_path=p['scl_hi'], cb_path=p['vq_hi'], bl_scl_cb_path=p['scl_lo'], bl_cb_path=p['vq_lo'])
m = Wavernn(20, 384, 128, 18); m.load_state_dict(synth.predictor_state_dict())
f = torch.from_numpy(synth.predictor_features(128, 300, utt0=5000)).cuda()
enc = m.encoder(cfg, f, None, 0.09, 0.28, return_indices=True)
idx, pitch = enc[7], f[:, :, 18:].contiguous()
rec = m.decode_indices(cfg, idx, pitch); torch.cuda.synchronize()
best = 1e9
for _ in range(5):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); rec = m.decode_indices(cfg, idx, pitch); b.record(); b.synchronize()
    best = min(best, a.elapsed_time(b))
print(f"decode_indices 128 x 300: {best:.3f} ms  equal to the encoder's reconstruction: {bool(torch.equal(rec, enc[0]))}  out {hashlib.sha1(rec.cpu().numpy().tobytes()).hexdigest()[:10]}")
