import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "sp-gan_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "examples")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def pytest_sessionfinish(session, exitstatus):
    """Write the measured parity errors of this run (helpers.PARITY_LOG) next to the other GPU-run artefacts."""
    try:
        import json
        import helpers
        import torch
        if not helpers.PARITY_LOG:
            return
        out = os.path.join(ROOT, "gpurun_out")
        os.makedirs(out, exist_ok=True)
        kind = "gpu" if torch.cuda.is_available() else "cpu"
        with open(os.path.join(out, "parity_errors_%s.json" % kind), "w") as f:
            json.dump({"device": torch.cuda.get_device_name(0) if kind == "gpu" else "cpu", "entries": helpers.PARITY_LOG}, f, indent=0)
    except Exception as e:                                   # noqa: BLE001  (reporting must never fail a test run)
        print("parity report not written: %s" % e)
