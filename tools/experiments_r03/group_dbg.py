import sys, os, tempfile; sys.path.insert(0, '.')
import numpy as np, torch, fpcodec_amd
from fpcodec_amd.wavernn import Wavernn
synth = fpcodec_amd.synth
d = tempfile.mkdtemp(); c = synth.codebooks(); p = {}
for k, v in c.items():
    p[k] = os.path.join(d, k + '.npy'); np.save(p[k], v)
cfg = dict(scl_cb_path=p['scl_hi'], cb_path=p['vq_hi'], bl_scl_cb_path=p['scl_lo'], bl_cb_path=p['vq_lo'])
m = Wavernn(20, 384, 128, 18); m.load_state_dict(synth.predictor_state_dict())
feat = torch.from_numpy(synth.predictor_features(8, 10, utt0=7000)).cuda()
os.environ["FPC_PRED_SPLIT"] = "0"
y0, a0, b0 = m.forward(feat); torch.cuda.synchronize()
e0 = m.encoder(cfg, feat, None, 0.09, 0.28, qtz=False)
for U, n in ((2, 2), (2, 4), (4, 4), (4, 8)):
    os.environ["FPC_PRED_GROUP"] = str(U); os.environ["FPC_PRED_SPLIT"] = str(n)
    for name, fn in (("forward", lambda: m.forward(feat)), ("qtz=False", lambda: m.encoder(cfg, feat, None, 0.09, 0.28, qtz=False)),
                     ("encode", lambda: m.encoder(cfg, feat, None, 0.09, 0.28, qtz=True))):
        try:
            out = fn(); torch.cuda.synchronize(); m.check()
            ref = (y0, a0, b0) if name == "forward" else e0
            same = [bool(torch.equal(a, b)) for a, b in zip(out[:3], ref[:3])] if name != "encode" else "-"
            print(U, n, name, "ok", same, flush=True)
        except Exception as e:
            print(U, n, name, "FAILED", str(e)[:80], flush=True)
