"""The suite's own plumbing (CPU): what ``retry_once_if_stalled`` retries and what it must never hide.

Eight worker processes time-sliced on ONE test GPU can let a bounded in-kernel wait of the exchange expire
(profiles/r05_world8_on_one_gpu.md); that is a property of the box, not of the arithmetic, so a multi-rank test is run once more when
-- and only when -- its failure says "timed out".  A wrong sum, a crash, a skip or a second stall go through unchanged.
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


def _run(test, capfd):
    with warnings.catch_warnings(record=True) as seen:
        warnings.simplefilter("always")
        try:
            return test(capfd=capfd), seen
        except BaseException as exc:  # noqa: BLE001 (pytest.skip raises a BaseException subclass)
            return exc, seen


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
            capfd.err = "rank 3: ExchangeError: timed out\n"
            raise RuntimeError("worker 3 exited with code 1")
        return "ok"

    out, _ = _run(flaky, capfd)
    assert out == "ok" and len(calls) == 2


def test_a_second_stall_fails():
    calls = []

    @retry_once_if_stalled
    def stalls():
        calls.append(1)
        raise AssertionError("timed out")

    out, _ = _run(stalls, _Capfd())
    assert isinstance(out, AssertionError) and len(calls) == 2


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
