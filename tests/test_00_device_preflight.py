"""Collected first (`test_00_`): is the NODE healthy, before anything of reagent_amd is loaded?

Round 4's driver record died in torch's own first host->device copy (`Memory access fault by GPU node-2`), before any rg_*
kernel could have been launched; with `-x` that voided every GPU test and read like a product crash.  This test touches the
device with torch alone, in a SUBPROCESS (a faulting first touch aborts the process it happens in), retried three times with
a 5 s back-off, and names the node when all three die.  `device_preflight()` is shared with `__graft_entry__.smoke()`."""
import pytest

from reagent_amd.device_preflight import NODE_FAULT, device_preflight, settle


@pytest.mark.gpu
def test_first_device_touch_in_a_subprocess():
    # (conftest.pytest_sessionstart already ran settle() and adopted a runtime workaround if the node needed one: this
    # repeats the touch under the environment the session now has)
    ok, log = device_preflight()
    assert ok, f"{NODE_FAULT}\n{log}"


def test_preflight_reports_a_dead_child_without_touching_a_device():
    # the retry / report logic itself (CPU): a child that aborts is a node fault with its stderr tail attached
    ok, log = device_preflight(code="import os; os.abort()", tries=2, backoff=0.0)
    assert not ok and "attempt 2" in log
    ok, log = device_preflight(code="print('fine')", tries=1)
    assert ok
    # settle(): a touch that only works under an alternative runtime setting is found, adopted and named
    import os

    code = "import os, sys; sys.exit(0 if os.environ.get('RG_TEST_ALT') == '1' else 134)"
    try:
        ok, log, adopted = settle(code=code, tries=2, backoff=0.0, alternatives=({"RG_TEST_OTHER": "1"}, {"RG_TEST_ALT": "1"}))
        assert ok and adopted == {"RG_TEST_ALT": "1"} and os.environ["RG_TEST_ALT"] == "1" and "NODE WORKAROUND" in log
        assert "RG_TEST_OTHER" not in os.environ
    finally:
        os.environ.pop("RG_TEST_ALT", None)
    ok, log, adopted = settle(code="import os; os.abort()", tries=1, backoff=0.0, alternatives=({"RG_TEST_OTHER": "1"},))
    assert not ok and adopted is None and "RG_TEST_OTHER" not in os.environ
