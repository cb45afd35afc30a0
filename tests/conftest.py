import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# every communicator in these tests lives on one node: keep RCCL's bootstrap sockets on the loopback interface (inherited by the
# subprocesses the tests start) instead of whatever interfaces the box happens to have (RCCL initialisation was seen to stall for exactly 300 s on some boxes of the pool)
os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_manifest():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "manifest.json")) as f:
        return json.load(f)
