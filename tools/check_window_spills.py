"""Guard for the hand-placed weight-load windows of the phase-form and training predictor kernels (csrc/predictor.hip chain4;
the shipped two-role kernels of csrc/predictor_df.h use plain loads the compiler counts): the loads and waits are inline
assembly the compiler does not count, so a window register must never be spilled
(or reloaded) between its load and its wait -- the spill would read it before the load has landed.  Compiles predictor.hip
to gfx950 assembly and reports, per kernel, the scratch operations that lie within `near` lines of such a load.

    python tools/check_window_spills.py            # prints a table, exit code 1 if any kernel has a suspect spill
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "feature-predictor-for-speech-codec_amd", "csrc")


def assembly():
    out = os.path.join(tempfile.mkdtemp(), "predictor.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                           "-fvisibility=hidden", "-I" + os.path.join(ROOT, "include"), "-w", "--offload-device-only", "-S",
                           os.path.join(CSRC, "predictor.hip"), "-o", out])
    return open(out).read().split("\n")


def check(lines, near=60):
    report, name, body = [], None, []

    def close():
        if name is None:
            return
        loads = [i for i, l in enumerate(body) if "global_load_dwordx4" in l and i > 0 and "ASMSTART" in body[i - 1]]
        scratch = [i for i, l in enumerate(body) if "scratch_store" in l or "scratch_load" in l]
        bad = [i for i in scratch if any(abs(i - j) < near for j in loads)]
        if loads:
            report.append((name, len(loads), len(scratch), len(bad)))

    for l in lines:
        m = re.match(r"^(_ZN[^:]*k_[a-z_]+[A-Za-z0-9_]*):", l)
        if m:
            close()
            name, body = re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", m.group(1)).split("ENS")[0].split("EPK")[0], []
        elif name is not None:
            body.append(l)
            if "s_endpgm" in l:
                close()
                name = None
    return report


if __name__ == "__main__":
    rep = check(assembly())
    for name, loads, scratch, bad in rep:
        print(f"{name:24s} window loads {loads:4d}  scratch operations {scratch:4d}  within 60 lines of a window load {bad}")
    sys.exit(1 if any(r[3] for r in rep) or not rep else 0)
