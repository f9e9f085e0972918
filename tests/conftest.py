import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))  # the CPU checker is test infrastructure

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# Some GPU tests run a full-size configuration against the CPU checker and say nothing for a minute or two; pytest holds
# their output back.  A line per test start and one per minute of a running test go to a log file, so that whoever watches
# the run can tell a long test from a hung one (a test that really hangs is ended by the timeout in pytest.ini).
_progress = {"test": None, "since": 0.0, "path": None}


def _progress_line(text):
    import time
    if _progress["path"]:
        try:
            with open(_progress["path"], "a") as f:
                f.write("%s %s\n" % (time.strftime("%H:%M:%S"), text))
        except OSError:
            pass


@pytest.fixture(scope="session", autouse=True)
def _progress_log(request):
    import threading
    import time
    out_dir = os.path.join(ROOT, "gpurun_out")
    if "not gpu" in (request.config.getoption("-m") or ""):
        yield
        return
    os.makedirs(out_dir, exist_ok=True)
    _progress["path"] = os.path.join(out_dir, "pytest_progress.log")
    stop = threading.Event()

    def tick():
        while not stop.wait(60.0):
            if _progress["test"]:
                _progress_line("running %s for %.0f s" % (_progress["test"], time.time() - _progress["since"]))

    th = threading.Thread(target=tick, daemon=True)
    th.start()
    yield
    stop.set()


def pytest_runtest_logstart(nodeid, location):
    import time
    _progress["test"], _progress["since"] = nodeid, time.time()
    _progress_line("start " + nodeid)


@pytest.fixture(scope="session")
def scenarios():
    with open(os.path.join(GOLDEN, "scenarios.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def sparse_goldens():
    with open(os.path.join(GOLDEN, "sparse_graphs.json")) as f:
        return json.load(f)["graphs"]


@pytest.fixture(scope="session")
def threshold_kats():
    with open(os.path.join(GOLDEN, "threshold_kats.json")) as f:
        return json.load(f)["kats"]
