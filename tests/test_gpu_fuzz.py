"""GPU suite: a short, fixed-seed slice of the randomised sweeps in tests/fuzz_parity.py (C-ABI level, HIP vs oracle / the
oracle/_ref libraries) and tests/fuzz_system.py (System level, HIP vs the host classes routed to the oracle).  The tools
run the same checks for as long as one likes on fresh seeds; the suite pins a few dozen systems of every kind."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def test_c_abi_sweep_fixed_seeds():
    import fuzz_parity as F

    fails = []
    saved = F.T._cases  # the sweep swaps the parity module's case table for its own draw
    try:
        for seed in range(31000, 31025):
            for name, fn in F.checks(F.draw(seed)):
                try:
                    fn()
                except Exception as e:  # noqa: BLE001 - every kind of failure is a finding
                    fails.append((seed, name, type(e).__name__, str(e)[:120]))
    finally:
        F.T._cases = saved
    assert not fails, fails


def test_system_sweep_fixed_seeds():
    import fuzz_system as F

    fails, ran = [], 0
    for seed in range(32000, 32040):
        ran += F.run_seed(seed, fails)
    assert not fails, fails
    assert ran > 100


def test_system_sweep_on_the_cell_sorted_twin(monkeypatch):
    """the same sweep with the HIP side forced onto System's cell-sorted twin (mdapy_amd/system.py; csrc/order.hip): every
    analysis of every drawn system — triclinic, open, unwrapped, thin (no twin then), with string and numeric columns — must
    give what the oracle-routed classes give on the system as it is"""
    import fuzz_system as F

    monkeypatch.setenv("FUZZ_TWIN", "1")
    fails, ran = [], 0
    for seed in range(33000, 33040):
        ran += F.run_seed(seed, fails)
    assert not fails, fails
    assert ran > 100
