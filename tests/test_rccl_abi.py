"""The product calls RCCL through dlopen'd function pointers declared by hand (raycore.jl_amd/csrc/rc_rccl_abi.h: six prototypes, four enum
values).  This CPU test compiles tests/rccl_abi_check.cpp -- static_asserts of every one of them against the image's <rccl/rccl.h> -- and
proves the check has teeth by compiling it once more against a tampered copy of the header, which must fail (VERDICT r5 'next' #1d)."""
import os
import re
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
RCCL_H = "/opt/rocm/include/rccl/rccl.h"
ABI_H = os.path.join(ROOT, "raycore.jl_amd", "csrc", "rc_rccl_abi.h")

pytestmark = pytest.mark.skipif(not os.path.exists(RCCL_H) or shutil.which("g++") is None, reason="needs <rccl/rccl.h> and g++")


def compile_check(src, cwd):
    return subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", src], cwd=cwd, capture_output=True, text=True)


def test_hand_declared_abi_matches_rccl_h():
    p = compile_check(os.path.join(HERE, "rccl_abi_check.cpp"), HERE)
    assert p.returncode == 0, p.stderr[-3000:]


@pytest.mark.parametrize("what, pattern, replacement", [
    ("ncclUint64 value", r"kUint64 = 5", "kUint64 = 4"),
    ("ncclUint32 value", r"kUint32 = 3", "kUint32 = 2"),
    ("ncclReduce argument order", r"int datatype, int op, int root,", "int datatype, int root, size_t op,"),
    ("ncclCommInitAll arity", r"\(comm_t\* comms, int ndev, const int\* devlist\)", "(comm_t* comms, int ndev)"),
])
def test_the_check_has_teeth(tmp_path, what, pattern, replacement):
    text = open(ABI_H).read()
    tampered, n = re.subn(pattern, replacement, text)
    assert n == 1, f"pattern for {what} no longer matches rc_rccl_abi.h"
    csrc = tmp_path / "raycore.jl_amd" / "csrc"
    csrc.mkdir(parents=True)
    (csrc / "rc_rccl_abi.h").write_text(tampered)
    tests = tmp_path / "tests"
    tests.mkdir()
    shutil.copy(os.path.join(HERE, "rccl_abi_check.cpp"), tests / "rccl_abi_check.cpp")
    p = compile_check(str(tests / "rccl_abi_check.cpp"), str(tests))
    assert p.returncode != 0 and "static assertion failed" in p.stderr, f"tampering with {what} went unnoticed"


def test_the_product_never_includes_rccl_h():
    csrc = os.path.join(ROOT, "raycore.jl_amd", "csrc")
    for f in os.listdir(csrc):
        if f.endswith((".hip", ".h")):
            assert "#include <rccl" not in open(os.path.join(csrc, f)).read(), f
