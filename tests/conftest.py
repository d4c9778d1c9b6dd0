import os
import sys

import numpy as np
import pytest

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")  # before the HIP runtime starts; see ocflow_amd/__init__.py
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")

if os.environ.get("OCF_ABRT_TRACE_FILE"):
    # diagnostics (tools/diag/abrt_trace.c is LD_PRELOADed): Python tracebacks of every thread into a file of their own at a fatal
    # signal; faulthandler then restores the handler it found -- the preloaded one -- and re-raises, so the native trace follows.
    # Run pytest with `-p no:faulthandler` (its plugin would write into the captured stderr).
    import faulthandler
    _FH = open(os.environ["OCF_ABRT_TRACE_FILE"] + ".python.txt", "w")
    faulthandler.enable(file=_FH, all_threads=True)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # environment only (no GPU touched): the conv solver db the repo ships, so full-size model tests do not spend a minute in
    # MIOpen's search on a fresh box
    from ocflow_amd.runtime import use_repo_miopen_db
    use_repo_miopen_db()


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False) as z:
        return {k: z[k] for k in z.files if k != "_meta"}


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = load_golden(name)
        return cache[name]

    return get
