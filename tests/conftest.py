import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def _has_gpu():
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def pytest_sessionfinish(session, exitstatus):
    """Multi-rank tests on the one test GPU may re-run once after the exchange reported an expired bounded wait (tests/helpers.py:
    retry_once_if_stalled).  A session in which more than MAX_STALL_RETRIES tests needed that is failed: a stall that common is a defect."""
    from tests import helpers

    if len(helpers.RETRIES) > helpers.MAX_STALL_RETRIES and session.exitstatus == 0:
        sys.stderr.write(f"\n{len(helpers.RETRIES)} tests were re-run after an exchange stall (limit {helpers.MAX_STALL_RETRIES}): {helpers.RETRIES}\n")
        session.exitstatus = 1
