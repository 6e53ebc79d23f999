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


def _run(kind, name, copies=1, repeat=1, env=None):
    want = json.load(open(os.path.join(GOLD, f"dropin_{name}.json")))
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "tools", "dropin_run.py"), kind, name, str(want["frames"]),
                                   want["fft_topic"], str(copies), str(repeat)], text=True, timeout=600,
                                  env=dict(os.environ, **(env or {})))
    lines = want["lines"]
    if copies > 1 or repeat > 1:
        # `copies` receivers fed in turn: per frame the messages (and fftData emissions) of copy 1, then of
        # copy 2, ...; `repeat` build-run-delete cycles in one process: the whole stream again (zero start state)
        msgs = [l for l in lines if "topic" in l]
        ffts = [l for l in lines if "fft" in l]
        per = len(msgs) // want["frames"]
        assert per * want["frames"] == len(msgs)
        one = []
        for f in range(want["frames"]):
            one += msgs[f * per:(f + 1) * per] * copies
        for l in ffts:
            one += [l] * copies
        lines = one * repeat
    return [json.loads(l) for l in out.splitlines()], lines


def _probe(kind):
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "tools", "dropin_run.py"), "probe", kind], text=True, timeout=120)
    return json.loads(out.splitlines()[-1])


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


def test_reference_copes_with_two_receivers_and_restart():
    """What the GPU test below expects of the adapter, established on the reference build: two receivers
    built from one description and fed in turn publish every message twice per frame; a build-run-delete
    cycle repeated in one process repeats the stream (all state starts from zero)."""
    if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libdropin_ref.so")):
        pytest.skip("oracle/_ref/libdropin_ref.so not built (make -C host/qt needs /root/reference)")
    got, want = _run("ref", "config1", copies=2, repeat=2)
    assert got == want


def test_init_throws_where_the_reference_throws():
    """vfo::init of the reference throws std::out_of_range from firfilter::sanity_check_1f for a filter
    bandwidth above half the output rate (vfo.cpp:110-115 -> firfilter.cpp:122-134).  The adapter's init
    -- validated on the host through sdrx_check_vfo, no GPU involved -- throws the same exception type
    with the same what() text for the same descriptions, and nothing for the valid ones."""
    if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libdropin_ref.so")):
        pytest.skip("oracle/_ref/libdropin_ref.so not built (make -C host/qt needs /root/reference)")
    ref = _probe("ref")
    assert ref["ok"][0] == 0 and ref["bw_exactly_half_rate"][0] == 0 and ref["late_ok"][0] == 0
    assert ref["bw_above_half_rate"] == [1, "firdes check failed: 0 < fa <= sampling_freq / 2"]
    assert ref["late_bw_too_wide"][0] == 1 and ref["non_usb_with_a_bad_bandwidth"][0] == 1
    assert _probe("sdrx") == ref


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["profile_25e", "config4_12"])
def test_adapter_two_receivers_and_stop_start(name):
    """Trees are keyed by their root object, not by a process-wide registry: two receivers built from one
    description live side by side in one process (each main VFO runs its own context), and the whole
    build - run - delete cycle (MainWindow's stop / start, vfo.cpp:34-59) is done twice in that process.
    Every message and fftData emission of both receivers, both cycles, byte-identical to the reference's."""
    got, want = _run("sdrx", name, copies=2, repeat=2)
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert g == w


@pytest.mark.gpu
def test_adapter_over_a_device_list():
    """SDRX_DEVICES=0,0,0: behind the unmodified vfo.h every tree is sharded over a device list
    (sdrx_group_*; three shards on the one GPU of the test box) -- the subscriber still receives the
    reference's bytes in the reference's order, and fftData comes from whichever shard holds the VFO."""
    got, want = _run("sdrx", "profile_25e", env={"SDRX_DEVICES": "0,0,0"})
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert g == w


@pytest.mark.gpu
@pytest.mark.parametrize("env", [{"SDRX_PIPELINE": "1"}, {"SDRX_SHARE_UPLOAD": "1"}, {"SDRX_PIPELINE": "1", "SDRX_SHARE_UPLOAD": "1"},
                                 {"SDRX_PIPELINE": "1", "SDRX_DEVICES": "0,0"}])
def test_adapter_modes_publish_the_same_bytes(env):
    """The adapter's host-side modes change WHEN bytes move, never which: SDRX_PIPELINE=1 (process() submits its
    frame and delivers the previous one -- sdrx_submit* / sdrx_wait -- the last frame at the tree's deletion),
    with and without the shared upload (SDRX_SHARE_UPLOAD=1: the second main VFO of a receiver runs on the frame the first
    one uploaded once the library has compared the two byte for byte: sdrx_process_if_same / sdrx_submit_if_same), and over
    a device list.  profile_25e has two main
    VFOs and an fftData tap on one sub VFO: the subscriber's stream and the fftData log stay the reference's."""
    got, want = _run("sdrx", "profile_25e", env=env)
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert g == w


@pytest.mark.gpu
def test_adapter_pipelined_two_receivers_and_stop_start():
    got, want = _run("sdrx", "config4_12", copies=2, repeat=2, env={"SDRX_PIPELINE": "1"})
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert g == w


def _raw(kind, name, frames, fft, env):
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "tools", "dropin_run.py"), kind, name, str(frames), fft],
                                  text=True, timeout=600, env=dict(os.environ, **env))
    return [json.loads(l) for l in out.splitlines()]


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [{"SDRX_SHARE_UPLOAD": "1"}, {"SDRX_SHARE_UPLOAD": "1", "SDRX_PIPELINE": "1"}, {}])
@pytest.mark.parametrize("mutate", ["1", "2", "3"])
def test_adapter_processes_what_it_is_handed(mutate, mode):
    """sdrj::demodData reuses one `samples` vector for every frame and every main VFO, so the address and the length of
    what process() is handed never change.  Here the CONTENT changes between two main VFOs' calls (every 97th sample,
    sparing the 64 positions the round-3 adapter spot-checked), and on odd frames the first main VFO -- the usual uploader
    -- is skipped: the reference's `class vfo` processes what it is handed, and so must the adapter, whose second tree
    shares the first one's upload only when the library has compared the two frames byte for byte (sdrx_*_if_same).
    Both builds of the same client, same switch: byte-identical subscriber streams and fftData logs."""
    if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libdropin_ref.so")):
        pytest.skip("oracle/_ref/libdropin_ref.so not built (make -C host/qt needs /root/reference)")
    env = {"DROPIN_MUTATE": mutate}
    want = _raw("ref", "profile_25e", 4, "VFO19", env)
    plain = _raw("ref", "profile_25e", 4, "VFO19", {})
    assert want != plain  # (the switch does change what the reference publishes)
    got = _raw("sdrx", "profile_25e", 4, "VFO19", dict(env, **mode))
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert g == w


@pytest.mark.gpu
@pytest.mark.parametrize("env", [{}, {"SDRX_PIPELINE": "1"}, {"SDRX_DEVICES": "0,0"}])
def test_two_vfos_with_one_topic_are_two_taps(env):
    """vfo::fftVFOSlot sets emitFFT on EVERY VFO whose zmqTopic equals the selected string (vfo.cpp:492-509): an INI with one
    topic on two VFOs has two spectrum taps.  Here both are /5 leaves whose decimation is fused into the mix wave -- they
    keep decimate[0] only as taps (sdrx_add_tap gives each a buffer of its own) -- under different main VFOs, i.e. in two
    trees of the adapter: the fftData log (frame, topic, length, hash of the samples) and the subscriber's stream are the
    reference's, byte for byte."""
    if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libdropin_ref.so")):
        pytest.skip("oracle/_ref/libdropin_ref.so not built (make -C host/qt needs /root/reference)")
    want = _raw("ref", "config4_12_shared_topic", 3, "SHARE", {})
    assert sum(1 for l in want if "fft" in l) == 2 * 3  # two taps, three frames
    got = _raw("sdrx", "config4_12_shared_topic", 3, "SHARE", env)
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert g == w


@pytest.mark.gpu
def test_adapter_on_random_trees():
    """The seeded random trees of tests/test_gpu_parity.py (1-3 levels, depths 0-4, USB and IQ leaves, late decimation,
    partial last chunks) behind the unmodified vfo.h: the reference's sources and the adapter, the same client, three
    frames, an fftData tap on the tree's last leaf: byte-identical subscriber streams and fftData logs.  Trees the
    library refuses for a documented restriction (a rate below 1024 Hz ...) are skipped: the adapter reports them at the
    first process() and exits."""
    if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libdropin_ref.so")):
        pytest.skip("oracle/_ref/libdropin_ref.so not built (make -C host/qt needs /root/reference)")
    import numpy as np
    sys.path.insert(0, ROOT)
    from helpers import random_topology
    n = max(3, int(os.environ.get("SDRX_TEST_SEEDS", "60")) // 20)
    ran = 0
    for seed in range(n):
        topo = random_topology(np.random.default_rng(1000 + seed))
        if any(v.fs < 1024 or 0 < v.samples_per_buffer % 1024 < 256 for v in topo.vfos):
            continue
        fft = next((v.topic for v in reversed(topo.vfos) if v.topic), "")
        want = _raw("ref", f"random:{seed}", 3, fft, {})
        got = _raw("sdrx", f"random:{seed}", 3, fft, {"SDRX_PIPELINE": "1"} if seed % 2 else {})
        assert len(got) == len(want) and len(want) > 0, (seed, len(got), len(want))
        for g, w in zip(got, want):
            assert g == w, (seed, g, w)
        ran += 1
    assert ran >= 2


@pytest.mark.gpu
@pytest.mark.parametrize("pipelined", [True, False])
def test_adapter_survives_a_device_fault(pipelined):
    """a failed sdrx_wait must not leave the adapter's frame count ahead of the library's (every later
    submit failed with "2 frames in flight", the receiver published nothing forever).  SDRX_FAULT_WAIT=3 makes the third
    sdrx_wait of the process fail the way a HIP error does -- the frame stays queued, every later call of that context fails
    too.  The tree it hits (one of the three main VFOs of the 54W-style profile) loses the frames it had in flight, is given
    up, and is committed to a NEW context by its next process(): it publishes again from there (filter state restarted: a
    time gap, as after stop / start); the other two trees never notice -- their subscriber streams are the reference's, byte
    for byte."""
    if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libdropin_ref.so")):
        pytest.skip("oracle/_ref/libdropin_ref.so not built (make -C host/qt needs /root/reference)")
    frames = 6
    want = {"lines": _raw("ref", "config4_12", frames, "", {})}
    env = dict(os.environ, SDRX_FAULT_WAIT="3", **({"SDRX_PIPELINE": "1"} if pipelined else {}))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "dropin_run.py"), "sdrx", "config4_12", str(frames), ""],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "injected fault" in r.stderr and "given up and rebuilt" in r.stderr, r.stderr[-2000:]

    def by_topic(lines):
        d = {}
        for l in lines:
            m = json.loads(l) if isinstance(l, str) else l
            if "topic" in m:
                d.setdefault(m["topic"], []).append(m)
        return d
    ref, got = by_topic(want["lines"]), by_topic(r.stdout.splitlines())
    assert set(got) == set(ref) and all(len(v) == frames for v in ref.values())
    hit = sorted(t for t in ref if len(got[t]) != frames)
    assert len(hit) == 4, hit                        # the four /5 leaves under ONE main VFO
    lost = 2 if pipelined else 1                     # the frame whose wait failed (+ the one submitted behind it)
    for t in ref:
        if t in hit:
            assert len(got[t]) == frames - lost, (t, len(got[t]))
            assert all(m["len"] == ref[t][0]["len"] and m["rate"] == ref[t][0]["rate"] for m in got[t])
        else:
            assert got[t] == ref[t], t
