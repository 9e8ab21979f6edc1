# forward time of the group form against the number of workgroups (FPC_LIB_PATH may point at a FPC_PRED_PROF build)
import sys, os, time; sys.path.insert(0, '.')
import numpy as np, torch, fpcodec_amd
from fpcodec_amd.wavernn import Wavernn
synth = fpcodec_amd.synth
m = Wavernn(20, 384, 128, 18); m.load_state_dict(synth.predictor_state_dict())
def tm(fn):
    fn(); torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize(); return (time.perf_counter() - t) * 1e3
for U, n in ((2, 4), (4, 8), (1, 2)):
    for B in (32, 64, 96, 112, 120, 128):
        if B % U: continue
        f = torch.from_numpy(synth.predictor_features(B, 100, utt0=5000)).cuda()
        os.environ["FPC_PRED_GROUP"] = str(U); os.environ["FPC_PRED_SPLIT"] = str(n)
        print(f"U={U} n={n} B={B} workgroups={B // U * n}: forward(100 frames) {tm(lambda: m.forward(f)):.2f} ms", flush=True)
