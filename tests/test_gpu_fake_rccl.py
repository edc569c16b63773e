"""The multi-RANK branches of view_factors, executed on ONE GPU (VERDICT r5 'next' #1).  BASELINE's north star shards view_factors' rays over
the 8 GPUs of a node and reduces the per-triangle accumulators with RCCL over xGMI (the reference's own shape is a
Threads.@threads loop, src/kernels.jl:74-104).  The builder's and the driver's GPU boxes have one device, so until this file the RCCL
branches of raycore.jl_amd/csrc/rc_multi.hip had only ever run as one-rank communicators.  Here:

* G scenes on device 0 count as G ranks (test hook RC_DEBUG_RANKS_SHARE_DEVICE=1, honoured only under RC_ENABLE_DEBUG_HOOKS=1);
* tests/fake_rccl/libfake_rccl.so -- built below with hipcc from fake_rccl.hip, defining the six RCCL entry points with rccl.h's own
  prototypes and the real call's stream semantics -- is what the product's dlopen finds (RC_RCCL_LIBRARY);
* every case runs in a child process (tests/fake_rccl/child.py): the product resolves RCCL once per process, and on a node with several
  GPUs tests/test_gpu_multi_device_hw.py must still get the real library.

What is checked: the RAYS partition's matrix (chunked in-place ncclReduce on the communication streams, copier thread, event fan-in),
the totals' single u64 reduce, rc_multi_prepare, all against the oracle; the number / size / rank count of the collectives the stub saw;
an error on rank g > 0 (status word) and a failing ncclReduce -- both must surface as errors, not hangs, and leave the scenes usable.
What this does NOT show: anything about xGMI, RCCL's own kernels, or time.  DESIGN.md 5 keeps 'projected' on every multi-GPU number."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(600, method="thread")]

HERE = os.path.dirname(os.path.abspath(__file__))
FAKE_DIR = os.path.join(HERE, "fake_rccl")
FAKE_SO = os.path.join(FAKE_DIR, "libfake_rccl.so")


@pytest.fixture(scope="module")
def fake_so():
    from fake_rccl.build import build
    return build()


def run_child(case, ranks, fake_so, share=True, library=None, timeout=420):
    env = dict(os.environ)
    env["RC_RCCL_LIBRARY"] = fake_so if library is None else library
    env["RC_ENABLE_DEBUG_HOOKS"] = "1"
    if share:
        env["RC_DEBUG_RANKS_SHARE_DEVICE"] = "1"
    else:
        env.pop("RC_DEBUG_RANKS_SHARE_DEVICE", None)
    env["RC_CHILD_DUMP_AFTER"] = str(timeout - 60)
    try:
        p = subprocess.run([sys.executable, os.path.join(FAKE_DIR, "child.py"), case, str(ranks)], env=env, capture_output=True, text=True, timeout=timeout)
    except subprocess.TimeoutExpired as e:  # the child's faulthandler has dumped its threads by now
        pytest.fail(f"{case} with {ranks} ranks did not return within {timeout} s (a hang in the multi-rank path):\n{e.stdout}\n{e.stderr}")
    assert p.returncode == 0, f"{case} with {ranks} ranks failed (rc {p.returncode}):\n{p.stdout[-4000:]}\n{p.stderr[-6000:]}"
    info = json.loads(p.stdout.strip().splitlines()[-1])
    assert info["ok"] and info["ranks"] == ranks
    return info


@pytest.mark.parametrize("ranks", [2, 3, 8])
def test_rays_partition_reduces_chunks_over_the_communicator(fake_so, ranks):
    info = run_child("rays", ranks, fake_so)
    assert info["n_chunks"] >= 4 and info["stats"]["max_ranks"] == ranks


@pytest.mark.parametrize("ranks", [2, 3, 8])
def test_totals_one_u64_reduce(fake_so, ranks):
    run_child("totals", ranks, fake_so)


@pytest.mark.parametrize("ranks", [2, 8])
def test_multi_prepare_builds_and_warms_the_communicator(fake_so, ranks):
    info = run_child("prepare", ranks, fake_so)
    assert info["prepare"]["total_ms"] >= info["prepare"]["warmup_collective_ms"] >= 0


@pytest.mark.parametrize("ranks", [2, 3, 8])
def test_error_on_a_later_rank_is_reported_not_hung(fake_so, ranks):
    run_child("status_word", ranks, fake_so)


@pytest.mark.parametrize("ranks", [2, 8])
def test_failing_collective_is_reported_and_the_communicator_replaced(fake_so, ranks):
    run_child("reduce_failure", ranks, fake_so)


def test_without_the_hook_scenes_on_one_device_are_replicas(fake_so):
    run_child("hook_off", 3, fake_so, share=False)


def test_a_named_library_that_is_missing_is_an_error(fake_so):
    run_child("bad_library", 2, fake_so, library="/nonexistent/librccl.so.1")
