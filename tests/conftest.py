"""pytest wiring: `gpu` marker, import paths, golden-fixture loaders."""
import json
import os
import sys

# The HIP runtime multiplexes every stream of a process onto GPU_MAX_HW_QUEUES hardware queues (default 4), and
# PyTorch alone makes 64 streams at its first torch.cuda.Stream(): streams that share a queue run in turn. The
# library's look-ahead stream and the lanes of the batched Hessian inverse want queues of their own
# (include/mi355q.h, mi355q_prepare_device); it must be set before the runtime is loaded (import torch).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (os.path.join(ROOT, "ai-edge-quantizer_amd"), ROOT):
  if p not in sys.path:
    sys.path.insert(0, p)


def pytest_configure(config):
  config.addinivalue_line("markers", "gpu: needs a real MI355X (run via gpurun)")


@pytest.fixture(scope="session")
def ref_cases():
  """(arrays, {name: case}) recorded from the real reference (tests/golden/gen)."""
  arrays = np.load(os.path.join(GOLDEN, "ref_cases.npz"))
  with open(os.path.join(GOLDEN, "ref_cases.json")) as f:
    meta = json.load(f)
  return arrays, {c["name"]: c for c in meta["cases"]}


@pytest.fixture(scope="session")
def ref_digests():
  with open(os.path.join(GOLDEN, "ref_digests.json")) as f:
    return json.load(f)["cases"]


@pytest.fixture(scope="session")
def known_answers():
  with open(os.path.join(GOLDEN, "ref_known_answers.json")) as f:
    return json.load(f)
