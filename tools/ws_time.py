# time of the weights-stationary predictor kernels at 128 x 300 for several library variants (FPC_LIB_PATH per child)
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import tempfile, time
    import numpy as np, torch, fpcodec_amd
    from fpcodec_amd.wavernn import Wavernn
    synth = fpcodec_amd.synth
    d = tempfile.mkdtemp(); c = synth.codebooks(); p = {}
    for k, v in c.items():
        p[k] = os.path.join(d, k + '.npy'); np.save(p[k], v)
    cfg = dict(scl_cb_path=p['scl_hi'], cb_path=p['vq_hi'], bl_scl_cb_path=p['scl_lo'], bl_cb_path=p['vq_lo'])
    m = Wavernn(20, 384, 128, 18); m.load_state_dict(synth.predictor_state_dict())
    f = torch.from_numpy(synth.predictor_features(128, 300, utt0=5000)).cuda()
    def tm(fn):
        best = 1e9
        fn(); torch.cuda.synchronize()
        for _ in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fn(); b.record(); b.synchronize()
            best = min(best, a.elapsed_time(b))
        return best
    import hashlib
    enc = m.encoder(cfg, f, None, 0.09, 0.28, return_indices=True)
    h = hashlib.sha1(enc[0].cpu().numpy().tobytes() + enc[7].cpu().numpy().tobytes()).hexdigest()[:10]
    print(f"encode {tm(lambda: m.encoder(cfg, f, None, 0.09, 0.28)):.3f} ms  qtz=False {tm(lambda: m.encoder(cfg, f, None, 0.09, 0.28, qtz=False)):.3f}  "
          f"forward {tm(lambda: m.forward(f)):.3f}  out {h}", flush=True)
    sys.exit(0)
for name in sys.argv[1:]:
    env = dict(os.environ)
    if name != "base":
        env["FPC_LIB_PATH"] = os.path.join(ROOT, "build_variants", f"lib_{name}.so")
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, capture_output=True, text=True)
    print(f"{name:>10s}: {r.stdout.strip() or r.stderr.strip()[-300:]}", flush=True)
