"""fpc_decode_features at 128 x 300 (the receiver's closed loop): shipped form, FPC_WS_TAIL=pair (owner rebuilds, third hop),
FPC_PRED_WS=0 (row split)."""
import os, sys, time, tempfile, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import numpy as np, torch, fpcodec_amd, hashlib
    from fpcodec_amd.wavernn import Wavernn
    synth = fpcodec_amd.synth
    d = tempfile.mkdtemp(); c = synth.codebooks(); p = {}
    for k, v in c.items():
        p[k] = os.path.join(d, k + '.npy'); np.save(p[k], v)
    cfg = dict(scl_cb_path=p['scl_hi'], cb_path=p['vq_hi'], bl_scl_cb_path=p['scl_lo'], bl_cb_path=p['vq_lo'])
    m = Wavernn(20, 384, 128, 18); m.load_state_dict(synth.predictor_state_dict())
    f = torch.from_numpy(synth.predictor_features(128, 300, utt0=5000)).cuda()
    os.environ.pop("FPC_WS_TAIL", None)
    enc = m.encoder(cfg, f, None, 0.09, 0.28, qtz=True, return_indices=True)
    if len(sys.argv) > 2 and sys.argv[2] == "pair":
        os.environ["FPC_WS_TAIL"] = "pair"
    pit = f[:, :, 18:20].contiguous()
    rec = m.decode_indices(cfg, enc[7], pit); torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter(); rec = m.decode_indices(cfg, enc[7], pit); torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) * 1e3)
    print(f"decode_features {best:.3f} ms  equal to the encoder's reconstruction: {bool(torch.equal(rec, enc[0]))}", flush=True)
    sys.exit(0)
for name, env, arg in (("shipped", {}, ""), ("pair", {}, "pair"), ("row split", {"FPC_PRED_WS": "0"}, "")):
    e = dict(os.environ); e.update(env)
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", arg], env=e, capture_output=True, text=True)
    print(f"{name:>10s}: {r.stdout.strip() or r.stderr.strip()[-300:]}", flush=True)
