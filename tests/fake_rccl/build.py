"""Builds tests/fake_rccl/libfake_rccl.so (test infrastructure: the stub communicator of tests/test_gpu_fake_rccl.py) with hipcc for gfx950.
Also called by __graft_entry__.build() so that the library travels to the GPU box prebuilt; the test rebuilds it when the source is newer."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "fake_rccl.hip")
SO = os.path.join(HERE, "libfake_rccl.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def build(force=False):
    if force or not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(SRC):
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", SO, SRC])
    return SO


if __name__ == "__main__":
    print(build(force=True))
