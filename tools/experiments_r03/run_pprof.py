# phase profile of k_forward (build with -DFPC_PRED_PROF; FPC_LIB_PATH points at it)
import sys, os; sys.path.insert(0, '.')
import numpy as np, torch, fpcodec_amd
from fpcodec_amd.wavernn import Wavernn
synth = fpcodec_amd.synth
m = Wavernn(20, 384, 128, 18); m.load_state_dict(synth.predictor_state_dict())
for B in (128, 1):
    f = torch.from_numpy(synth.predictor_features(B, 300)).cuda()
    for n in ("0", "2", "4", "8"):
        if B == 128 and n in ("4", "8"): continue
        os.environ["FPC_PRED_SPLIT"] = n
        m.forward(f); torch.cuda.synchronize()
        m.forward(f); torch.cuda.synchronize()
