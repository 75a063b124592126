import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)


def pytest_configure(config):
  config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def cuda():
  import torch
  if not torch.cuda.is_available():
    pytest.fail("this test is marked gpu and needs a HIP device; run CPU suites with -m 'not gpu'")
  return torch.device("cuda:0")


# Tests written while the builder had no GPU access carry this mark until they have passed once on an MI355X: the
# driver runs `pytest -m gpu -x`, and a first-run surprise in a NEW test must not hide the verified suite behind it.
# strict=False: a pass is reported as XPASS; the mark is removed as soon as scripts/run2.sh has been green on hardware.
unverified_on_gpu = pytest.mark.xfail(strict=False, reason="added in round 4 after the builder's GPU access ended: not yet run on an MI355X")
