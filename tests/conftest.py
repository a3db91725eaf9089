import os
import sys
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))


@pytest.fixture
def golden():
    return load_golden


# U1 bounds live in oracle/bounds.py (shared with __graft_entry__.smoke())
from oracle.bounds import (U1_FP32_LINF, U1_FP32_L2, U1_ROUTE_LINF, U1_ROUTE_L2,   # noqa: E402,F401
                           U1_SMALL_FP32_LINF, U1_SMALL_FP32_L2)


def note_measured(**kw):
    """Append one measured figure to gpurun_out/r05_u1_measured.jsonl (how the bounds above were set)."""
    import json
    path = os.path.join(ROOT, 'gpurun_out', 'r05_u1_measured.jsonl')
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, 'a') as f:
        f.write(json.dumps(kw) + '\n')
