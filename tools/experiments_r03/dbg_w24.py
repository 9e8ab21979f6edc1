import sys, os; sys.path.insert(0, '.')
import numpy as np, torch, fpcodec_amd
from fpcodec_amd.wavernn import Wavernn
synth = fpcodec_amd.synth
m = Wavernn(20, 384, 128, 18); m.load_state_dict(synth.predictor_state_dict())
feat = torch.from_numpy(synth.predictor_features(1, 4, utt0=7000)).cuda()
os.environ["FPC_PRED_SPLIT"] = "0"; os.environ["FPC_PRED_DF"] = "0"
y0, a0, b0 = m.forward(feat); torch.cuda.synchronize()
os.environ.pop("FPC_PRED_DF")
y1, a1, b1 = m.forward(feat); torch.cuda.synchronize()
y0, y1 = y0.cpu().numpy(), y1.cpu().numpy()
for t in range(4):
    d = np.nonzero(y0[0, t] != y1[0, t])[0]
    print("frame", t, "differing outputs:", d.tolist(), "max abs diff", float(np.abs(y0[0, t] - y1[0, t]).max()))
h = (a0.cpu().numpy() != a1.cpu().numpy())[0]
print("h1 units differing:", int(h.sum()), np.nonzero(h)[0][:20].tolist())
