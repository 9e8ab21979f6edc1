# stage profile of the weights-stationary predictor kernels: FPC_LIB_PATH=build_variants/lib_ws_prof.so (-DFPC_WS_PROF)
import sys, os, tempfile; sys.path.insert(0, '.')
import numpy as np, torch, fpcodec_amd
from fpcodec_amd.wavernn import Wavernn
synth = fpcodec_amd.synth
d = tempfile.mkdtemp(); c = synth.codebooks(); p = {}
for k, v in c.items():
    p[k] = os.path.join(d, k + '.npy'); np.save(p[k], v)
cfg = dict(scl_cb_path=p['scl_hi'], cb_path=p['vq_hi'], bl_scl_cb_path=p['scl_lo'], bl_cb_path=p['vq_lo'])
m = Wavernn(20, 384, 128, 18); m.load_state_dict(synth.predictor_state_dict())
for B in (128, 1):
    f = torch.from_numpy(synth.predictor_features(B, 300, utt0=5000)).cuda()
    for rep in range(2):
        m.forward(f); torch.cuda.synchronize()
    for rep in range(2):
        m.encoder(cfg, f, None, 0.09, 0.28); torch.cuda.synchronize()
    m.encoder(cfg, f, None, 0.09, 0.28, qtz=False); torch.cuda.synchronize()
f = torch.from_numpy(synth.predictor_features(128, 300, utt0=5000)).cuda()
print("all frames coded (l1 = l2 = 0):", file=sys.stderr, flush=True)
m.encoder(cfg, f, None, 0.0, 0.0); torch.cuda.synchronize()
print("no frame above (l1 = l2 = 1e9):", file=sys.stderr, flush=True)
m.encoder(cfg, f, None, 1e9, 1e9); torch.cuda.synchronize()
