import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _oracle_host_threads():
    """The CPU oracle (and every other host-side torch op of the tests) runs with the threads the box can actually schedule:
    torch's default capped by the cgroup CPU quota (oracle.host_threads: 16 on the pool's GPU boxes, where torch defaults to 128
    and the oracle then runs 3.9x slower)."""
    import torch
    import oracle
    before = torch.get_num_threads()
    torch.set_num_threads(oracle.host_threads())
    yield
    torch.set_num_threads(before)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


# ---- the headline parity tests cannot drop out of a green record (round-4 verdict, item 2).  Round 4 had a time-budget guard here
# that let the heaviest CPU-oracle tests skip themselves on a slow box; with oracle.host_threads() the oracle legs cost a quarter
# of what they did, so the guard is gone, and a `-m gpu` session that selected the tests carrying BASELINE configs[1..3] (and
# infer.py's own defaults) FAILS unless every one of them ran and passed -- a skip or a deselection inside the session counts as
# a failure.
MUST_PASS_ON_GPU = (
    "tests/test_engine_gpu.py::test_sd21_full_size_parity_base_unet_only",      # configs[1]: base UNet, B = 1
    "tests/test_engine_gpu.py::test_sd21_full_size_parity",                     # configs[2]: 1 -> 1 view, adapter + camera
    "tests/test_cfg4_shapes_gpu.py::test_sd21_full_size_parity_b32",            # configs[3]: 32 pairs
    "tests/test_engine_gpu.py::test_sd21_full_size_parity_768",                 # the reference's default image size
    "tests/test_engine_gpu.py::test_sd21_full_size_denoise_loop_cfg",           # chained forwards under CFG (Q4)
    "tests/test_engine_gpu.py::test_sd21_full_size_infer_defaults_loop",        # infer.py:181-187: 20 steps, guidance 1.0, B = 1
    "tests/test_engine_gpu.py::test_sd21_full_size_pipeline_defaults_loop_cfg", # pipeline.py defaults: 50 steps, guidance 7.5
)
_OUTCOMES = {}


def pytest_runtest_logreport(report):
    if report.nodeid not in MUST_PASS_ON_GPU:
        return
    if report.outcome != "passed":                       # a skip or a failure in any phase sticks
        _OUTCOMES[report.nodeid] = report.outcome
    elif report.when == "call":
        _OUTCOMES.setdefault(report.nodeid, "passed")


def missing_headline_tests(collected_ids, outcomes, have_gpu: bool):
    """The headline tests of a session that did not pass.  Enforced only for a session that has a GPU and collected ALL of them
    (the driver's `pytest tests -m gpu`); a run of one file or one -k expression is not held to it."""
    if not have_gpu or not all(t in collected_ids for t in MUST_PASS_ON_GPU):
        return []
    return [f"{t}: {outcomes.get(t, 'did not run')}" for t in MUST_PASS_ON_GPU if outcomes.get(t) != "passed"]


def pytest_collection_modifyitems(session, config, items):
    session.config._mvd_selected = {it.nodeid for it in items}


def pytest_sessionfinish(session, exitstatus):
    import torch
    sel = getattr(session.config, "_mvd_selected", set())
    if getattr(session, "shouldstop", False) or getattr(session, "shouldfail", False):
        return                                   # -x stopped the session at an earlier failure: that failure is the verdict
    # (MVD_ASSUME_GPU_SESSION=1: the CPU suite's own check of this hook -- a session in which every headline test skips must fail)
    have_gpu = torch.cuda.is_available() or os.environ.get("MVD_ASSUME_GPU_SESSION") == "1"
    bad = missing_headline_tests(sel, _OUTCOMES, have_gpu)
    if bad:
        tr = session.config.pluginmanager.get_plugin("terminalreporter")
        msg = "headline parity tests that did not pass in this GPU session:\n  " + "\n  ".join(bad)
        if tr is not None:
            tr.write_sep("=", "MVD: configs[1..3] parity is not optional", red=True)
            tr.write_line(msg)
        session.exitstatus = 1
