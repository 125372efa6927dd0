import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: CPU test taking more than ~30 s")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN



def free_port() -> int:
    """A TCP port nobody listens on right now (bind to 0, read it back): the rendezvous of the multi-process tests.  Ports derived
    from the process id collided (the module-scoped 1-rank RCCL group of tests/test_model_gpu.py and the two-rank subprocess test drew
    the same number: EADDRINUSE, once in a GPU run of round 3)."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def parity_log(*parts) -> None:
    """Every parity measurement the GPU tests make (max |hip - reference|, error / spread, gradient deviations, TSV flips) is printed
    AND, when MEMEHIP_PARITY_OUT names a file, appended to it: tools/publish_parity.py runs the reference-run tests that way and commits
    the result as profiles/rNN_parity.txt, so that every tolerance asserted in tests/ has its measured value on record next to it."""
    line = " ".join(str(x) for x in parts)
    print(line)
    out = os.environ.get("MEMEHIP_PARITY_OUT")
    if out:
        with open(out, "a", encoding="utf-8") as f:
            f.write(line + "\n")
