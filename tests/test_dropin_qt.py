"""The drop-in claim, end to end: host/qt/dropin_client.cpp drives `class vfo` through the PUBLIC
interface of the reference's unmodified vfo.h (setters, init, setVFOs, process, fftVFOSlot /
fftData) and a ZMQ subscriber records what the receiver publishes through the reference's own
ZmqPublisher.  Behind the header sit either the reference's sources (libdropin_ref.so) or
host/qt/vfo_adapter.cpp over libsdrx.so (libdropin_sdrx.so): the two message streams -- topic
bytes, sample-rate frame, payload bytes, order -- and the fftData emissions must be identical.
The reference stream is committed as tests/golden/dropin_*.json (tests/golden/make_golden.py)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
CASES = ["config1", "profile_25e", "config4_12"]


def _run(kind, name):
    want = json.load(open(os.path.join(GOLD, f"dropin_{name}.json")))
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "tools", "dropin_run.py"), kind, name, str(want["frames"]),
                                   want["fft_topic"]], text=True, timeout=600)
    return [json.loads(l) for l in out.splitlines()], want["lines"]


@pytest.mark.parametrize("name", CASES)
def test_fixture_is_what_the_reference_publishes(name):
    if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libdropin_ref.so")):
        pytest.skip("oracle/_ref/libdropin_ref.so not built (make -C host/qt needs /root/reference)")
    got, want = _run("ref", name)
    assert got == want


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_adapter_behind_the_unmodified_header_publishes_the_same_bytes(name):
    lib = os.path.join(ROOT, "oracle", "_ref", "libdropin_sdrx.so")
    assert os.path.exists(lib), "oracle/_ref/libdropin_sdrx.so must travel with the snapshot (make -C host/qt in the build container)"
    got, want = _run("sdrx", name)
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert g == w
