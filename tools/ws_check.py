# weights-stationary predictor kernels (predictor_ws.h) against the row-split two-role kernels (FPC_PRED_WS=0): bits and time
import sys, os, time, tempfile; sys.path.insert(0, '.')
import numpy as np, torch, fpcodec_amd
from fpcodec_amd.wavernn import Wavernn
synth = fpcodec_amd.synth
d = tempfile.mkdtemp(); c = synth.codebooks(); p = {}
for k, v in c.items():
    p[k] = os.path.join(d, k + '.npy'); np.save(p[k], v)
cfg = dict(scl_cb_path=p['scl_hi'], cb_path=p['vq_hi'], bl_scl_cb_path=p['scl_lo'], bl_cb_path=p['vq_lo'])
m = Wavernn(20, 384, 128, 18); m.load_state_dict(synth.predictor_state_dict())
what = sys.argv[1] if len(sys.argv) > 1 else "all"
def run(feat):
    out = []
    if what in ("all", "fwd"):
        y, h1, h2 = m.forward(feat)
        y2, h1b, h2b = m.forward(feat[:, :5], h1, h2)
        out += [t.cpu().numpy() for t in (y, h1, h2, y2, h1b, h2b)]
    if what in ("all", "enc"):
        enc = m.encoder(cfg, feat, None, 0.09, 0.28, qtz=True, return_indices=True)
        enc2 = m.encoder(cfg, feat, None, 0.09, 0.28, qtz=False)
        dec = m.decode_indices(cfg, enc[7], feat[:, :, 18:].contiguous())
        out += [t.cpu().numpy() for t in enc[:6]] + list(enc[6]) + [enc[7].cpu().numpy()]
        out += [t.cpu().numpy() for t in enc2[:6]] + [dec.cpu().numpy()]
    torch.cuda.synchronize()
    return out
def tm(fn):
    fn(); torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize(); return (time.perf_counter() - t) * 1e3
ok = True
for B, L in ((1, 30), (7, 40), (16, 33), (128, 60), (200, 20)):
    feat = torch.from_numpy(synth.predictor_features(B, L, utt0=7000)).cuda()
    os.environ["FPC_PRED_WS"] = "0"
    ref = run(feat)
    for fast in ("1", "0"):
        os.environ["FPC_PRED_WS"] = "1"; os.environ["FPC_FAST_HOP"] = fast
        got = run(feat)
        bad = [i for i, (a, b) in enumerate(zip(ref, got)) if not np.array_equal(a, b)]
        print(f"B={B} L={L} fast_hop={fast}: {'identical' if not bad else 'DIFFERENT ' + str(bad)}", flush=True)
        if bad:
            a, b = ref[bad[0]], got[bad[0]]
            w = np.argwhere(a != b)
            print("   first output", bad[0], "shape", a.shape, "first diff at", w[0], a[tuple(w[0])], b[tuple(w[0])], "count", len(w), flush=True)
        ok &= not bad
os.environ.pop("FPC_FAST_HOP")
f = torch.from_numpy(synth.predictor_features(128, 300, utt0=5000)).cuda()
for ws in ("0", "1"):
    os.environ["FPC_PRED_WS"] = ws
    line = f"128 x 300, weights-stationary {ws}:"
    if what in ("all", "enc"):
        line += f" encode {tm(lambda: m.encoder(cfg, f, None, 0.09, 0.28)):.2f} ms  qtz=False {tm(lambda: m.encoder(cfg, f, None, 0.09, 0.28, qtz=False)):.2f}"
    if what in ("all", "fwd"):
        line += f"  forward {tm(lambda: m.forward(f)):.2f}"
    print(line, flush=True)
f1 = f[:1].contiguous()
line = "one utterance, weights-stationary:"
if what in ("all", "enc"):
    line += f" encode {tm(lambda: m.encoder(cfg, f1, None, 0.09, 0.28)):.2f} ms"
if what in ("all", "fwd"):
    line += f"  forward {tm(lambda: m.forward(f1)):.2f}"
print(line, flush=True)
print("ALL IDENTICAL" if ok else "MISMATCH")
