"""Time k_decode2 builds (build_variants/lib_<name>.so; 'base' = the shipped library) at B = 512: decode ms, cycles per
sample pair and the phase stamps.  python tools/pair_var.py name1 name2 ...   (timing only: ablation builds give wrong PCM)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for name in sys.argv[1:]:
    env = dict(os.environ, FPC_LPCNET_PAIRING="1")
    if name != "base":
        env["FPC_LIB_PATH"] = os.path.join(ROOT, "build_variants", f"lib_{name}.so")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "stamp_probe.py"), "512"], env=env, capture_output=True, text=True, timeout=300)
    ms = [l for l in r.stdout.splitlines() if l.startswith("decode ms")]
    env["FPC_DECODE_STAMPS"] = "1"
    r2 = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "stamp_probe.py"), "512"], env=env, capture_output=True, text=True, timeout=300)
    ph = [l for l in r2.stderr.splitlines() if "phase lengths" in l][-1:]
    w = [l for l in r2.stderr.splitlines() if "wave  0" in l or "wave  4" in l or "wave  8" in l or "wave 10" in l][-4:]
    print(f"{name:>10s}: {ms[-1] if ms else r.stderr[-300:]}", flush=True)
    for l in ph + w:
        print("            " + l.replace("[fpc stamps] ", ""), flush=True)
