import json
import math
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _nan(o):
    if o is None:
        return math.nan
    if isinstance(o, list):
        return [_nan(x) for x in o]
    if isinstance(o, dict):
        return {k: _nan(v) for k, v in o.items()}
    return o


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(ROOT, "tests", "golden", "reference_known_answers.json")) as f:
        return _nan(json.load(f))
