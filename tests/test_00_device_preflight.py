"""Collected first (`test_00_`): is the NODE healthy, before anything of reagent_amd is loaded?

Round 4's driver record died in torch's own first host->device copy (`Memory access fault by GPU node-2`), before any rg_*
kernel could have been launched; with `-x` that voided every GPU test and read like a product crash.  This test touches the
device with torch alone, in a SUBPROCESS (a faulting first touch aborts the process it happens in), retried three times with
a 5 s back-off, and names the node when all three die.  `device_preflight()` is shared with `__graft_entry__.smoke()`."""
import pytest

from reagent_amd.device_preflight import NODE_FAULT, device_preflight


@pytest.mark.gpu
def test_first_device_touch_in_a_subprocess():
    ok, log = device_preflight()
    assert ok, f"{NODE_FAULT}\n{log}"


def test_preflight_reports_a_dead_child_without_touching_a_device():
    # the retry / report logic itself (CPU): a child that aborts is a node fault with its stderr tail attached
    ok, log = device_preflight(code="import os; os.abort()", tries=2, backoff=0.0)
    assert not ok and "attempt 2" in log
    ok, log = device_preflight(code="print('fine')", tries=1)
    assert ok
