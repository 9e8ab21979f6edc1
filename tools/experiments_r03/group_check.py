# group form (U utterances share every weight load) against the single-workgroup form: bits and time
import sys, os, time, tempfile; sys.path.insert(0, '.')
import numpy as np, torch, fpcodec_amd
from fpcodec_amd.wavernn import Wavernn
from fpcodec_amd import bitstream
synth = fpcodec_amd.synth
d = tempfile.mkdtemp(); c = synth.codebooks(); p = {}
for k, v in c.items():
    p[k] = os.path.join(d, k + '.npy'); np.save(p[k], v)
cfg = dict(scl_cb_path=p['scl_hi'], cb_path=p['vq_hi'], bl_scl_cb_path=p['scl_lo'], bl_cb_path=p['vq_lo'])
m = Wavernn(20, 384, 128, 18); m.load_state_dict(synth.predictor_state_dict())

def run(feat):
    y, h1, h2 = m.forward(feat)
    y2, h1b, h2b = m.forward(feat[:, :5], h1, h2)
    enc = m.encoder(cfg, feat, None, 0.09, 0.28, qtz=True, return_indices=True)
    enc2 = m.encoder(cfg, feat, None, 0.09, 0.28, qtz=False)
    dec = m.decode_indices(cfg, enc[7], feat[:, :, 18:].contiguous())
    torch.cuda.synchronize()
    out = [t.cpu().numpy() for t in (y, h1, h2, y2, h1b, h2b)] + [t.cpu().numpy() for t in enc[:6]] + list(enc[6]) + [enc[7].cpu().numpy()]
    out += [t.cpu().numpy() for t in enc2[:6]]
    if dec is not None: out.append(dec.cpu().numpy())
    return out

def tm(fn):
    fn(); torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize(); return (time.perf_counter() - t) * 1e3

ok = True
for B, L in ((8, 40), (128, 60)):
    feat = torch.from_numpy(synth.predictor_features(B, L, utt0=7000)).cuda()
    os.environ["FPC_PRED_SPLIT"] = "0"; os.environ.pop("FPC_PRED_GROUP", None)
    ref = run(feat)
    for U, n in ((2, 2), (2, 4), (2, 8), (4, 4), (4, 8)):
        if B // U * n > 256: continue
        os.environ["FPC_PRED_GROUP"] = str(U); os.environ["FPC_PRED_SPLIT"] = str(n)
        got = run(feat)
        same = all(np.array_equal(a, b) for a, b in zip(ref, got)) and len(ref) == len(got)
        bad = [i for i, (a, b) in enumerate(zip(ref, got)) if not np.array_equal(a, b)]
        print(f"B={B} L={L} U={U} n={n}: {'identical' if same else 'DIFFERENT ' + str(bad)}", flush=True)
        ok &= same
f = torch.from_numpy(synth.predictor_features(128, 300, utt0=5000)).cuda()
for U, n in ((1, 2), (2, 4), (2, 2), (4, 8), (4, 4)):
    os.environ["FPC_PRED_GROUP"] = str(U); os.environ["FPC_PRED_SPLIT"] = str(n)
    print(f"128 x 300, U={U} n={n}: encode {tm(lambda: m.encoder(cfg, f, None, 0.09, 0.28)):.2f} ms  qtz=False {tm(lambda: m.encoder(cfg, f, None, 0.09, 0.28, qtz=False)):.2f}  forward {tm(lambda: m.forward(f)):.2f}", flush=True)
print("ALL IDENTICAL" if ok else "MISMATCH")
