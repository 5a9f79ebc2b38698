"""Two data-parallel ranks on ONE GPU, as real processes (runs FIRST in the session: the file name sorts
before every other test module, and nothing here — nor tests/conftest.py — touches the GPU in the parent
before the children have been started).  Each child (tools/dp_rank_check.py) trains a real RadLIF step
through the gradient all-reducer under both launch policies, checks the averaged gradients against the mean
of the shards' gradients, and checks SyncBN against a single process on the whole batch."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_gradients_and_syncbn():
    import socket

    with socket.socket() as sock:  # a port nobody is listening on right now
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), SPARCH_DIST_BACKEND="gloo", SPARCH_SHARE_GPU="1",
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "dp_rank_check.py")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=420)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    for rank, out in enumerate(outs):  # both ranks' complete output is kept (a copy goes under profiles/)
        with open(os.path.join(ROOT, "gpurun_out", f"dp_two_ranks_one_gpu_rank{rank}.log"), "w") as f:
            f.write(out)
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {rank} failed:\n{out[-4000:]}"
        assert "data-parallel checks passed" in out
        assert "forced overlap on a full grid finished" in out
        assert "a timeout on one rank skips the step on both" in out


@pytest.mark.gpu
def test_rccl_backend_with_one_rank():
    """The `nccl` (= RCCL) backend itself, which the gloo rehearsals never touch: a one-rank process group with the
    collectives forced on (tools/nccl_world1_check.py) — init, async all-reduces on RCCL's stream under every launch
    policy with gradients bit-identical to the reducer-less step, the window policy's launch order, the status word's
    MAX all-reduce, a captured step with the all-reduce inside.  In its own process: it owns the process group."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    env.pop("RANK", None)
    env.pop("WORLD_SIZE", None)
    with __import__("socket").socket() as sock:
        sock.bind(("127.0.0.1", 0))
        env["MASTER_PORT"] = str(sock.getsockname()[1])
    try:
        p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "nccl_world1_check.py")], env=env, cwd=ROOT,
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=420)
        out = p.stdout
    except subprocess.TimeoutExpired as e:
        so = e.stdout or ""
        out = (so.decode(errors="replace") if isinstance(so, bytes) else so) + "\n[timed out after 420 s]"
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "nccl_world1_check.log"), "w") as f:
        f.write(out)
    # The tool prints this line behind its last check and leaves through os._exit: the verdict is the line, not
    # whatever RCCL's or HIP's exit handlers do afterwards.
    if "nccl world-size-1 checks passed" in out:
        for policy in ("overlap", "window", "deferred"):
            assert f"policy {policy}" in out
        return
    # One of OUR checks failing is a failure; the RCCL runtime not coming up on this box (rendezvous, communicator
    # init, a crash outside our code) is reported as a skip with its output — it says nothing about this library.
    assert "AssertionError" not in out and "SparchHipError" not in out, out[-4000:]
    pytest.skip("the RCCL runtime did not complete the one-rank run on this box:\n" + out[-1500:])
