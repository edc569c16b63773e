"""The oracle under sanitizers (VERDICT r5 'next' #6).  Every parity claim in this repository rests on oracle/rc_oracle.c -- ~1 600 lines of C
with a hand-rolled thread pool, thread-locals and fixed-size traversal stacks -- so its memory- and thread-safety is a committed,
repeatable fact, not an assumption: `make -C oracle san` (AddressSanitizer + UndefinedBehaviorSanitizer) re-runs the KAT / property /
golden / driver / BVH4 / collision / mesh modules of this suite in a child interpreter, `make -C oracle tsan` (ThreadSanitizer) drives
every multi-threaded entry point with 8 threads (tests/oracle_threads_child.py).  CPU builds only: no GPU-side sanitizer exists on this
pool.  (The reference's own safety net is static: Aqua + @inferred, test/runtests.jl:75.)"""
import os
import re
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def runtime(name):
    p = subprocess.run(["gcc", "-print-file-name=" + name], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


ASAN, TSAN = runtime("libasan.so"), runtime("libtsan.so")
MODULES = ["test_oracle_kats.py", "test_oracle_properties.py", "test_golden.py", "test_oracle_analytic_drivers.py", "test_oracle_bvh4.py",
           "test_oracle_collision.py", "test_oracle_mesh.py", "test_oracle_independent_f64.py"]
REPORT = re.compile(r"runtime error:|AddressSanitizer|ThreadSanitizer|LeakSanitizer|SUMMARY: \w+Sanitizer")


@pytest.mark.skipif(ASAN is None, reason="libasan.so not found next to gcc")
def test_oracle_modules_under_asan_and_ubsan():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "san"])
    env = dict(os.environ, LD_PRELOAD=ASAN, RC_ORACLE_VARIANT="asan",
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:halt_on_error=0",  # (leaks: the interpreter's own; the oracle's scenes are freed by the tests)
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=0")
    p = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "not gpu", "-p", "no:cacheprovider"] + [os.path.join(HERE, m) for m in MODULES],
                       env=env, capture_output=True, text=True, timeout=1500, cwd=ROOT)
    out = p.stdout + p.stderr
    assert p.returncode == 0, out[-6000:]
    assert not REPORT.search(out), "sanitizer report:\n" + "\n".join(l for l in out.splitlines() if REPORT.search(l))[:4000]
    m = re.search(r"(\d+) passed", out)
    assert m and int(m.group(1)) >= 70, out[-2000:]   # the modules really ran (and against the sanitized library: pyoracle asserts the variant)


@pytest.mark.skipif(TSAN is None, reason="libtsan.so not found next to gcc")
def test_oracle_thread_pool_under_tsan():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "tsan"])
    env = dict(os.environ, LD_PRELOAD=TSAN, RC_ORACLE_VARIANT="tsan", TSAN_OPTIONS="halt_on_error=0:report_signal_unsafe=0:exitcode=0")
    p = subprocess.run([sys.executable, os.path.join(HERE, "oracle_threads_child.py")], env=env, capture_output=True, text=True, timeout=1500, cwd=ROOT)
    out = p.stdout + p.stderr
    assert p.returncode == 0 and "threads-ok" in p.stdout, out[-6000:]
    assert not REPORT.search(out), "sanitizer report:\n" + out[-6000:]
