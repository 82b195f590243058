"""The suite's own plumbing (CPU): what ``retry_once_if_stalled`` retries and what it must never hide.

Eight worker processes time-sliced on ONE test GPU can let a bounded in-kernel wait of the exchange expire
(profiles/r05_world8_on_one_gpu.md); that is a property of the box, not of the arithmetic, so a multi-rank test is run once more when
-- and only when -- its failure carries the EXCHANGE'S OWN message for an expired bounded wait (tests/helpers.EXCHANGE_STALL_MESSAGES).  A
wrong sum, a crash, a skip, a second stall, or somebody else's "timed out" (gloo, a subprocess) go through unchanged.
"""
import warnings

import pytest

from tests.helpers import retry_once_if_stalled


class _Capfd:
    """Stand-in for pytest's capfd: hands back what the 'workers' wrote to stderr during the attempt."""

    def __init__(self):
        self.err = ""

    def readouterr(self):
        err, self.err = self.err, ""
        return type("Captured", (), {"out": "", "err": err})()


def _run(test, capfd, keep_count=False):
    from tests import helpers

    before = len(helpers.RETRIES)
    with warnings.catch_warnings(record=True) as seen:
        warnings.simplefilter("always")
        try:
            return test(capfd=capfd), seen
        except BaseException as exc:  # noqa: BLE001 (pytest.skip raises a BaseException subclass)
            return exc, seen
        finally:
            if not keep_count:
                del helpers.RETRIES[before:]  # (the session's count is about real multi-rank tests, not these stand-ins)


def test_a_stall_in_the_assertion_text_is_retried_once_with_a_warning():
    calls = []

    @retry_once_if_stalled
    def flaky():
        calls.append(1)
        assert len(calls) > 1, "a P2P exchange timed out waiting for another rank's partial sums"
        return "ok"

    out, seen = _run(flaky, _Capfd())
    assert out == "ok" and len(calls) == 2 and any("once more" in str(w.message) for w in seen)


def test_a_stall_reported_only_on_the_workers_stderr_is_retried():
    calls, capfd = [], _Capfd()

    @retry_once_if_stalled
    def flaky():
        calls.append(1)
        if len(calls) == 1:
            capfd.err = "rank 3: ExchangeError: a wait for another rank's partial sums timed out\n"
            raise RuntimeError("worker 3 exited with code 1")
        return "ok"

    out, _ = _run(flaky, capfd)
    assert out == "ok" and len(calls) == 2


def test_a_second_stall_fails():
    calls = []

    @retry_once_if_stalled
    def stalls():
        calls.append(1)
        raise AssertionError("a wait for another rank's partial sums timed out")

    out, _ = _run(stalls, _Capfd())
    assert isinstance(out, AssertionError) and len(calls) == 2


def test_somebody_elses_timeout_is_never_retried():
    """A gloo collective, a subprocess or pytest-timeout saying "timed out" is a failure of the test, not the exchange's bounded wait."""
    from tests import helpers

    calls, before = [], len(helpers.RETRIES)

    @retry_once_if_stalled
    def hangs():
        calls.append(1)
        raise RuntimeError("[gloo] Timed out waiting 1800000ms for recv operation to complete; subprocess timed out after 600 seconds")

    out, seen = _run(hangs, _Capfd())
    assert isinstance(out, RuntimeError) and len(calls) == 1 and not seen and len(helpers.RETRIES) == before


def test_retries_are_counted():
    from tests import helpers

    before = len(helpers.RETRIES)
    calls = []

    @retry_once_if_stalled
    def flaky_counted():
        calls.append(1)
        assert len(calls) > 1, helpers.EXCHANGE_STALL_MESSAGES[0]

    _run(flaky_counted, _Capfd(), keep_count=True)
    assert helpers.RETRIES[before:] == ["flaky_counted"]
    del helpers.RETRIES[before:]  # (this session's count is about real multi-rank tests)


def test_a_wrong_sum_is_never_retried():
    calls = []

    @retry_once_if_stalled
    def wrong():
        calls.append(1)
        assert 1.0 == 2.0, "rank 0: 17 wrong words with a healthy status"

    out, seen = _run(wrong, _Capfd())
    assert isinstance(out, AssertionError) and len(calls) == 1 and not seen


def test_a_skip_passes_through_and_arguments_are_forwarded():
    @retry_once_if_stalled
    def skips(world, mode="p2p"):
        if world == 8:
            pytest.skip("not runnable on this box")
        return world, mode

    out, _ = _run(lambda capfd: skips(8, capfd=capfd), _Capfd())
    assert isinstance(out, pytest.skip.Exception)
    assert skips(2, mode="fenced", capfd=_Capfd()) == (2, "fenced")
    import inspect

    assert list(inspect.signature(skips).parameters) == ["world", "mode", "capfd"]  # (pytest injects capfd from the signature)
