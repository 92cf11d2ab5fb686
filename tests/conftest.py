import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def oracle():
    import oracle as O
    O.build()
    return O


def pytest_collection_modifyitems(config, items):
    """GPU runs: torch's bundled HIP runtime has to be the first one loaded in the process (the runtime that is loaded first
    serves the whole process; the engine library binds to whichever is there, torch only to its own) - whatever the order of
    the selected tests.  Done once, before the first test, and only when a test that needs the GPU was selected."""
    if not any(it.get_closest_marker("gpu") for it in items):
        return
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:       # noqa: BLE001 - a box without torch still runs the tests that do not need it
        pass
