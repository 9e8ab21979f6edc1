"""time the quantizer searches alone (one workgroup per vector, 40 vectors per CU back to back):
    python tools/time_vq.py [variant names ...]   (build_variants/lib_<name>.so; 'base' = the shipped library)"""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import numpy as np, torch, fpcodec_amd
    from fpcodec_amd import vq_func
    synth = fpcodec_amd.synth
    d = tempfile.mkdtemp(); c = synth.codebooks(); p = {}
    for k, v in c.items():
        p[k] = os.path.join(d, k + '.npy'); np.save(p[k], v)
    n = 256 * 40
    r = synth.cb_training_vectors(n, seed_offset=3) * np.float32(0.3)
    x = r[:, :1].copy()
    def tm(fn):
        fn(); torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize(); return (time.perf_counter() - t)
    rd = torch.from_numpy(r).cuda()
    for name, fn in (("2-stage 1024+1024", lambda: vq_func.vq_quantize(r, p['vq_hi'])), ("1-stage 512", lambda: vq_func.vq_quantize(r, p['vq_lo'])),
                     ("scalar 256", lambda: vq_func.scl_quantize(x, p['scl_hi']))):
        t = tm(fn)
        print(f"  {name:18s} {t * 1e6 / 40:8.2f} us per search (incl. host copies of the call)", flush=True)
    sys.exit(0)
for name in sys.argv[1:] or ["base"]:
    env = dict(os.environ)
    if name != "base":
        env["FPC_LIB_PATH"] = os.path.join(ROOT, "build_variants", f"lib_{name}.so")
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, capture_output=True, text=True)
    print(name + ":\n" + (r.stdout or r.stderr[-500:]), flush=True)
