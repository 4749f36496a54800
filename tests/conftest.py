import os
import socket
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
for p in (ROOT, os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)

# multi-process GPU tests (marker `dp2`): the rank processes of tests/dp_worker.py, started at the END OF COLLECTION - i.e. before
# this process has made any GPU call (a process that has initialised the GPU must not start other programs on this pool; counting
# devices does not initialise it) - and only when such a test was selected on a box with a GPU
DP2 = {"proc": None, "dir": None, "log": None, "rc": None}


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "dp2: checks the output of the 2-rank worker processes (tests/dp_worker.py)")


def pytest_collection_finish(session):
    if DP2["proc"] is not None or not any(item.get_closest_marker("dp2") for item in session.items):
        return
    import torch
    if torch.cuda.device_count() < 1:
        return
    out = tempfile.mkdtemp(prefix="m2f_dp2_")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    log = open(os.path.join(out, "worker.log"), "w")
    DP2["proc"] = subprocess.Popen([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                                    "--master-addr", "127.0.0.1", "--master-port", str(port),
                                    os.path.join(ROOT, "tests", "dp_worker.py"), out], env=env, stdout=log, stderr=subprocess.STDOUT)
    DP2["dir"], DP2["log"] = out, log
    # ... and WAIT for it here, still before this process has touched the GPU: the two rank processes must not share the device
    # with this process's own GPU tests (a fault or an out-of-memory kill could not be attributed, timing tests would get noisy)
    try:
        DP2["rc"] = DP2["proc"].wait(timeout=900)
    except subprocess.TimeoutExpired:
        DP2["proc"].kill()
        DP2["rc"] = -9


def pytest_sessionfinish(session, exitstatus):
    p = DP2["proc"]
    if p is not None and p.poll() is None:
        p.kill()                                            # the exact process we started
    if DP2["log"] is not None:
        DP2["log"].close()


@pytest.fixture(scope="session")
def dp2_results():
    """The two ranks' result files (the worker processes have exited before the first test ran; skips when they were not started)."""
    import torch
    p = DP2["proc"]
    if p is None:
        pytest.skip("no GPU on this box: the 2-rank worker processes were not started")
    rc = DP2.get("rc")
    if rc is None:                                          # (not reached: collection_finish waits)
        rc = p.wait(timeout=600)
    DP2["log"].flush()
    log = open(os.path.join(DP2["dir"], "worker.log")).read()[-4000:]
    errs = "".join(open(os.path.join(DP2["dir"], f)).read() for f in sorted(os.listdir(DP2["dir"])) if f.startswith("error_rank"))
    assert rc == 0, f"2-rank worker exited with {rc}\n{errs}\n--- log tail ---\n{log}"
    return [torch.load(os.path.join(DP2["dir"], f"rank{r}.pt"), weights_only=False) for r in range(2)], DP2["dir"]


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
