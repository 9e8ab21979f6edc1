# stage profile of k_encode_df (FPC_LIB_PATH -> a -DFPC_PRED_PROF build)
import sys, os, tempfile; sys.path.insert(0, '.')
import numpy as np, torch, fpcodec_amd
from fpcodec_amd.wavernn import Wavernn
synth = fpcodec_amd.synth
d = tempfile.mkdtemp(); c = synth.codebooks(); p = {}
for k, v in c.items():
    p[k] = os.path.join(d, k + '.npy'); np.save(p[k], v)
cfg = dict(scl_cb_path=p['scl_hi'], cb_path=p['vq_hi'], bl_scl_cb_path=p['scl_lo'], bl_cb_path=p['vq_lo'])
m = Wavernn(20, 384, 128, 18); m.load_state_dict(synth.predictor_state_dict())
f = torch.from_numpy(synth.predictor_features(128, 300, utt0=5000)).cuda()
for kw in (dict(qtz=True), dict(qtz=False)):
    m.encoder(cfg, f, None, 0.09, 0.28, **kw); torch.cuda.synchronize()
for l1, l2 in ((0.0, 0.0), (1e9, 1e9)):
    m.encoder(cfg, f, None, l1, l2); torch.cuda.synchronize()
