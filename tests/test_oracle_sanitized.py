"""The CPU oracle under AddressSanitizer + UndefinedBehaviorSanitizer (`make -C oracle SAN=1`): the golden-vector and
known-answer suites run against the sanitizer build in a child process (the ASan runtime has to be the first library a
process loads, so it cannot be switched on inside this one).  Any report -- an out-of-bounds read, a misaligned or
overlapping access, signed overflow -- aborts the child (halt_on_error, -fno-sanitize-recover) and fails this test.  The
oracle is test infrastructure; for the vocoder half it is the only reference there is."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _runtime(name):
    p = subprocess.run(["gcc", f"-print-file-name={name}"], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


@pytest.mark.parametrize("suite", ["tests/test_oracle_golden.py", "tests/test_vocoder_second_opinion.py::test_numpy_restatement_draws_what_the_oracle_draws",
                                   "tests/test_host_cpu.py -k oracle_or_philox_or_tree_pdf_or_ulaw_or_lpc_only_or_bitstream"])
def test_oracle_suites_are_clean_under_asan_ubsan(suite):
    asan = _runtime("libasan.so")
    if asan is None:
        pytest.skip("no libasan next to gcc")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "SAN=1"])
    env = dict(os.environ, FPC_ORACLE_SAN="1", LD_PRELOAD=asan,
               # (python itself is not leak-clean; everything else is fatal)
               ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
               OMP_NUM_THREADS="1")
    probe = ("from oracle import oracle as O; O.lib(); m = open('/proc/self/maps').read(); "
             "print('libasan' in m and 'libfpc_oracle_san.so' in m and 'libfpc_oracle.so' not in m)")
    live = subprocess.run([sys.executable, "-c", probe], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert live.stdout.strip() == "True", (live.stdout, live.stderr[-2000:])  # the child really runs the sanitizer build
    args = suite.replace("_or_", " or ").split(" ", 2)
    cmd = [sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider"] + \
        ([args[0], "-k", args[2]] if len(args) == 3 else [args[0]])
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    tail = (p.stdout + p.stderr)[-3000:]
    assert p.returncode == 0, tail
    assert "ERROR: AddressSanitizer" not in p.stderr and "runtime error:" not in p.stderr, tail
    assert " passed" in p.stdout, tail
