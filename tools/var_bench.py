"""Time the decode kernel of several library variants on the same input and compare PCM hashes.
    python tools/var_bench.py [--voiced] name1 name2 ...   (build_variants/lib_<name>.so; 'base' = the shipped one)
Each variant runs in its own child process (one library per process)."""
import hashlib, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import numpy as np, torch, fpcodec_amd
    from fpcodec_amd.lpcnet import LPCNet
    from fpcodec_amd.ceps2lpc import ceps2lpc_v
    synth = fpcodec_amd.synth
    B, T = 256, 100
    raw = synth.vocoder_features_raw(16, T)
    if sys.argv[2] == "voiced":
        raw[:, :, 19] = 0.9
    f = torch.from_numpy(np.tile(raw, (16, 1, 1))).cuda()
    f[:, :, 20:] = ceps2lpc_v(f.reshape(-1, 36)[:, :20].contiguous())[1].reshape(B, T, 16)
    voc = LPCNet(synth.lpcnet_weights())
    sd = synth.seeds(B)
    ms = []
    for _ in range(4):
        pcm = voc.synthesize(f, sd)
        torch.cuda.synchronize()
        ms.append(voc.last_decode_ms())
    h = hashlib.sha1(pcm.cpu().numpy().tobytes()).hexdigest()[:12]
    m = min(ms[1:])
    print(f"{m:8.3f} ms  {m * 1e-3 * 2.4e9 / (T * 160 - 17):7.0f} cyc/sample  pcm {h}", flush=True)
    sys.exit(0)
args = sys.argv[1:]
mode = "unvoiced"
if args and args[0] == "--voiced":
    mode, args = "voiced", args[1:]
for name in args:
    env = dict(os.environ)
    if name != "base":
        env["FPC_LIB_PATH"] = os.path.join(ROOT, "build_variants", f"lib_{name}.so")
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", mode], env=env, capture_output=True, text=True)
    print(f"{name:>14s}: {r.stdout.strip() or r.stderr.strip()[-300:]}", flush=True)
