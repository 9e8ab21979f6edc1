"""encode / forward timing of predictor library variants at 128 x 300:  python tools/time_encode_var.py base name ..."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for name in sys.argv[1:] or ["base"]:
    env = dict(os.environ)
    if name != "base":
        env["FPC_LIB_PATH"] = os.path.join(ROOT, "build_variants", f"lib_{name}.so")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "time_encode_split.py")], env=env, capture_output=True, text=True)
    keep = [l for l in r.stdout.splitlines() if l.startswith(("encode full", "encode qtz=False", "forward", "single utterance, 8"))]
    print(f"{name:>10s}: " + " | ".join(keep) + (r.stderr[-300:] if r.returncode else ""), flush=True)
