import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

# Hardware queues of the test process (ROCm runtime variable, read when HIP initialises -- i.e. before the first torch / libvidc
# call): 8, so that a test can create a context whose kernel classes really run on 8 streams (VIDC_WIDE_STREAMS=1 when the
# context is created; the S2 repeated-decode stress runs in both modes).  Contexts stay in the default 4-stream mode unless a
# test asks otherwise: the automatic policies of the library are tested as a process without the variable would see them.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
os.environ.setdefault("VIDC_WIDE_STREAMS", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu():
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    from oracle.pyoracle import Oracle

    return Oracle()


@pytest.fixture(scope="session")
def golden():
    import json

    with open(os.path.join(ROOT, "tests", "golden", "roc_golden.json")) as f:
        return json.load(f)["cases"]
