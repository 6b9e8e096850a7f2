import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


# Scratch blocks handed out by the library's device pool are filled with NaN patterns in the test processes: a kernel that
# reads memory the path never wrote fails deterministically (round 2: a masked value row of the decoder-step attention
# entered its sum as 0 * garbage and only showed once a recycled block held NaNs).  Costs a device synchronisation per
# allocation, so it is for tests only.
os.environ.setdefault("SC_DEBUG_FILL", "0xff")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def report_dir():
    d = ROOT / "gpurun_out"
    d.mkdir(exist_ok=True)
    return d
