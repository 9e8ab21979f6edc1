import sys, os; sys.path.insert(0, '.')
import numpy as np, torch, fpcodec_amd
from fpcodec_amd.wavernn import Wavernn
from oracle import oracle as O
O.build()
synth = fpcodec_amd.synth
rng = np.random.default_rng(77)
for dims in ((20, 384, 128, 18), (20, 256, 64, 18), (20, 128, 256, 18), (20, 512, 32, 18), (64, 192, 96, 24), (36, 64, 16, 12)):
    inf, h1, h2, fc = dims
    sd = synth.predictor_state_dict(inf, h1, h2, fc, seed=900 + h1 + h2)
    P = O.Predictor(sd)
    B, L = 5, 14
    feat = synth.predictor_features(B, L, utt0=8100) if inf == 20 else (rng.normal(size=(B, L, inf)) * 0.1).astype(np.float32)
    y0, a0, b0 = P.forward(feat)
    m = Wavernn(inf, h1, h2, fc); m.load_state_dict(sd)
    x = torch.from_numpy(feat).cuda()
    for df in ("1", "0"):
        for n in ("0", "2", "4"):
            if int(n) and (h1 % (4 * int(n)) or h2 % (4 * int(n))): continue
            os.environ["FPC_PRED_SPLIT"] = n
            if df == "0": os.environ["FPC_PRED_DF"] = "0"
            else: os.environ.pop("FPC_PRED_DF", None)
            y, a, b = m.forward(x); torch.cuda.synchronize()
            d = np.abs(y.cpu().numpy() - y0).max()
            print(dims, "two-role" if df == "1" else "phase", "n=" + n, "identical" if d == 0 else f"max abs diff {d:.3e}", "t0 diff %.3e" % np.abs(y.cpu().numpy()[:, 0] - y0[:, 0]).max(), flush=True)
