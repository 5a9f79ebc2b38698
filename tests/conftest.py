import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """A fresh checkout has no libsparch_hip.so (it is git-ignored): build it once, as
    __graft_entry__.build() does.  hipcc cross-compiles gfx950 without a GPU.  (This is a build step, not a
    fallback: with no library the package refuses to import.)"""
    lib = os.path.join(ROOT, "sparch_amd", "libsparch_hip.so")
    if not os.path.exists(lib):
        import subprocess

        subprocess.run(["make", "-C", os.path.join(ROOT, "sparch_amd", "csrc")], check=True,
                       stdout=subprocess.DEVNULL)


def _cgroup_cpus():
    """CPUs the container may actually use (cgroup quota), or None: a GPU box can show 256 CPUs in the affinity mask
    under a 16-CPU quota, and the CPU oracle on 256 threads inside that quota crawls (bench.py usable_cpus)."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()[:2]
        return None if q == "max" else max(1, -(-int(q) // int(per)))
    except (OSError, ValueError):
        return None


@pytest.fixture(scope="session", autouse=True)
def _torch_threads_within_the_cpu_quota():
    quota = _cgroup_cpus()
    if quota is not None:
        import torch

        if torch.get_num_threads() > quota:
            torch.set_num_threads(quota)
    yield


def _has_gpu():
    """device_count() does not initialise the GPU in this process (is_available() would): the first test of a
    GPU session starts child processes that must come before any GPU use of the parent."""
    try:
        import torch

        return torch.cuda.device_count() > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
