# 128 x 300 timings of the group forms, repeated (is the 256-workgroup slow state reproducible?)
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
    fn(); torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize(); return (time.perf_counter() - t) * 1e3
B = int(os.environ.get("GT_B", "128"))
f = torch.from_numpy(synth.predictor_features(B, 300, utt0=5000)).cuda()
for rep in range(2):
    for U, n in ((1, 2), (2, 4), (2, 2), (4, 8), (4, 4), (4, 2), (1, 1)):
        if B // U * n > 256: continue
        os.environ["FPC_PRED_GROUP"] = str(U); os.environ["FPC_PRED_SPLIT"] = str(n if n > 1 else 0)
        print(f"B={B} x 300, U={U} n={n}: forward {tm(lambda: m.forward(f)):.2f} ms  qtz=False {tm(lambda: m.encoder(cfg, f, None, 0.09, 0.28, qtz=False)):.2f}  encode {tm(lambda: m.encoder(cfg, f, None, 0.09, 0.28)):.2f}", flush=True)
