import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(autouse=True)
def _library_default_modes(request):
    """every GPU test starts and ends in the library's default arithmetic (ops.DEFAULT_CONV_DTYPE = fp32x3, fp32 tensors, automatic
    tiles): a test that switches the process-wide mode cannot leak it into the next one"""
    gpu = request.node.get_closest_marker("gpu") is not None
    if gpu:
        import torch
        import litemkd_amd  # noqa: F401
        from litemkd_amd import ops
        ops.reset_compute_dtypes()
        # the CPU references: as many threads as this process may use (a cgroup share of a 256-thread host oversubscribes 10x)
        n = len(os.sched_getaffinity(0))
        try:
            quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
            if quota != "max":
                n = min(n, max(1, int(int(quota) / int(period))))
        except Exception:
            pass
        torch.set_num_threads(max(1, min(n, 32)))
    yield
    if gpu:
        ops.reset_compute_dtypes()
        litemkd_amd.lib().call("lmkd_conv_set_tile", 0)
        litemkd_amd.lib().call("lmkd_conv_set_patch", 1)
        litemkd_amd.lib().call("lmkd_conv_set_patch16", 1)
        litemkd_amd.lib().call("lmkd_conv_set_s2_patch", 1)
        litemkd_amd.lib().call("lmkd_conv_set_wgrad_win16", 1)
        litemkd_amd.lib().call("lmkd_conv_set_wgrad_stem", 1)
