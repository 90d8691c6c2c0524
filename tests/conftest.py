import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# ---- the GPU suite's wall clock (verdict round 5, item 7) ------------------------------------------------------------------
# The driver gives `pytest -m gpu` 1 200 s.  The oracle side of the suite runs on the box's host cores and its duration moved
# between 444 and 710 s from box to box in round 5: a suite that passes in more than HPV_SUITE_LIMIT_S (default 900 s) FAILS here,
# with the durations in the log, instead of being killed by the driver's limit one slow box later.
import time as _time

_T0 = _time.time()


def pytest_report_header(config):
    seed = os.environ.get("HPV_FUZZ_SEED_USED") or os.environ.get("HPV_FUZZ_SEED")
    return ["hp-vpinns: fuzz seed of the rotating sweep cases HPV_FUZZ_SEED=%s (tests/test_gpu_fuzz.py)" % (seed or "<set at collection>")]


def pytest_collection_finish(session):
    seed = os.environ.get("HPV_FUZZ_SEED_USED")
    if seed:
        print("\nhp-vpinns: HPV_FUZZ_SEED=%s" % seed)


def pytest_sessionfinish(session, exitstatus):
    limit = float(os.environ.get("HPV_SUITE_LIMIT_S", "900"))
    took = _time.time() - _T0
    mark = session.config.getoption("-m") or ""
    if "gpu" in mark and "not gpu" not in mark and took > limit and exitstatus == 0:
        print("\nhp-vpinns: the GPU suite took %.0f s, more than HPV_SUITE_LIMIT_S = %.0f s (the driver's limit is 1 200 s): FAILING the "
              "session -- slow host cores on this box, or a test that grew; see --durations" % (took, limit))
        session.exitstatus = 1
