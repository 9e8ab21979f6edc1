# 128 x 300 timings of the current library (FPC_LIB_PATH selects a variant)
import sys, os, time, tempfile; sys.path.insert(0, '.')
import numpy as np, torch, fpcodec_amd
from fpcodec_amd.wavernn import Wavernn
synth = fpcodec_amd.synth
d = tempfile.mkdtemp(); c = synth.codebooks(); p = {}
for k, v in c.items():
    p[k] = os.path.join(d, k + '.npy'); np.save(p[k], v)
cfg = dict(scl_cb_path=p['scl_hi'], cb_path=p['vq_hi'], bl_scl_cb_path=p['scl_lo'], bl_cb_path=p['vq_lo'])
m = Wavernn(20, 384, 128, 18); m.load_state_dict(synth.predictor_state_dict())
def tm(fn):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t = time.perf_counter(); fn(); torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t) * 1e3)
    return best
f = torch.from_numpy(synth.predictor_features(128, 300, utt0=5000)).cuda()
print(f"{os.environ.get('FPC_LIB_PATH', 'shipped').split('/')[-1]}: encode {tm(lambda: m.encoder(cfg, f, None, 0.09, 0.28)):.2f} ms  qtz=False {tm(lambda: m.encoder(cfg, f, None, 0.09, 0.28, qtz=False)):.2f}  forward {tm(lambda: m.forward(f)):.2f}", flush=True)
