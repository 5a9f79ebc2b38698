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
