"""The plain-C oracle (oracle/vfo_oracle.c) against the committed golden fixtures, which are
outputs of the real reference build (tests/golden/make_golden.py).  Bit-exact everywhere."""
import numpy as np
import pytest

from helpers import GOLDEN_TREES, bits, golden, golden_topology, sha
from oracle import binding as ob
from sdrreceiver_amd import synth


def test_nco_tables_and_start_sequence():
    g = golden("primitives.npz")
    for k, (fs, f) in enumerate(g["nco_pairs"]):
        t = ob.osc_table("port", fs, f)
        assert len(t) == fs
        assert np.array_equal(bits(t[:512]), bits(g[f"nco{k}_head"]))
        assert np.array_equal(bits(t[-64:]), bits(g[f"nco{k}_tail"]))
        assert sha(t) == str(g[f"nco{k}_sha"])
        seq = ob.osc_sequence("port", fs, f, 8)
        assert np.array_equal(bits(seq), bits(g[f"nco{k}_seq"]))
        # oscillator.cpp:30,39-50: sample 0 sees the LAST table entry, sample k>=1 entry k
        assert bits(seq[:1])[0] == bits(t[-1:])[0] and np.array_equal(bits(seq[1:]), bits(t[1:8]))


def test_nco_known_answers_from_survey():
    # SURVEY.md 8a-2 (Fs=384000, f=110854)
    seq = ob.osc_sequence("port", 384000, 110854, 2)
    assert np.float32(seq[0].real) == np.float32(0.974677086) and np.float32(seq[0].imag) == np.float32(-0.00214868761)
    assert np.float32(seq[1].real) == np.float32(-0.879853129) and np.float32(seq[1].imag) == np.float32(-0.464900196)
    t = ob.osc_table("port", 384000, 110854)
    assert abs(abs(t[0]) ** 2 - 0.9025) < 1e-6
    assert abs(abs(t[5000]) ** 2 - 0.95) < 1e-6


def test_low_pass_designs():
    g = golden("primitives.npz")
    for k, a in enumerate(g["lp_args"]):
        taps = ob.low_pass("port", *a)
        assert np.array_equal(bits(taps), bits(g[f"lp{k}"])), a
    assert len(ob.low_pass("port", 2, 12000, 4000, 1000)) == 29
    assert len(ob.low_pass("port", 2, 48000, 10000, 2500)) == 47
    assert len(ob.low_pass("port", 2, 240000, 24000, 12000)) == 49
    with pytest.raises(ValueError):  # sanity_check_1f: cutoff above fs/2
        ob.low_pass("port", 2, 12000, 7000, 1750)


def test_hilbert_taps():
    g = golden("primitives.npz")
    for k, fs in enumerate(g["hilbert_fs"]):
        h = ob.hilbert_taps("port", 125, fs)
        assert np.array_equal(bits(h), bits(g[f"hilbert{k}"]))
        assert np.count_nonzero(h) == 62 and h[62] == 0 and np.all(h[0::2] == 0)
    h = ob.hilbert_taps("port", 125, 3000)
    assert np.float32(h[61]) == np.float32(0.638710558) and np.float32(h[1]) == np.float32(0.0104706651)


def test_halfband_frame_boundary_rule_shape():
    # ramp 1..48 in 3 frames of 16 -> frame 1 starts 11.006 12.9567 15.25 18.0433 ...
    g = golden("primitives.npz")
    out = g["hb_ramp_out"]
    assert np.allclose(out[1][0:8:2][:4], [11.006, 12.9567, 15.25, 18.0433], atol=2e-3)
    assert np.allclose(out[1][8::2], [19.994, 22, 24, 26][:len(out[1][8::2])], atol=2e-3)


@pytest.mark.parametrize("fixture", sorted(GOLDEN_TREES))
def test_tree_against_reference_fixture(fixture):
    key, frames = GOLDEN_TREES[fixture]
    topo = golden_topology(key)
    g = golden(fixture)
    nodes, roots = ob.build_tree("port", topo)
    lcg = synth.Lcg(1)
    for f in range(frames):
        ob.process_roots(roots, synth.lcg_frame(topo.frame, lcg))
        for i, (n, v) in enumerate(zip(nodes, topo.vfos)):
            z = n.stream()
            assert sha(z) == str(g[f"f{f}_v{i}_stream_sha"]), (fixture, f, i)
            if not topo.children(i):
                pay = n.usb() if v.demod_usb else n.iq()
                assert sha(pay) == str(g[f"f{f}_v{i}_pay_sha"]), (fixture, f, i)
                assert np.array_equal(pay[:256], g[f"f{f}_v{i}_pay_head"])
                if f"f{f}_v{i}_pay" in g:
                    assert np.array_equal(pay, g[f"f{f}_v{i}_pay"])


def test_publish_record_matches_zmq_framing():
    """vfo::transmitData -> ZmqPublisher::publish framing (zmqpublisher.cpp:82-96), against the
    three frames the real libzmq delivered for topic 'VFO07-extra' @ 24000."""
    g = golden("zmq_framing.npz")
    assert bytes(g["frame0"]) == b"VFO07"  # exactly 5 topic bytes
    assert bytes(g["frame1"]) == np.uint32(24000).tobytes()  # native-endian u32
    assert np.array_equal(g["frame2"], g["payload_in"])
    from sdrreceiver_amd import topology as tp
    topo = tp.config1()
    topo.vfos[1].topic = "VFO07-extra"
    nodes, roots = ob.build_tree("port", topo)
    ob.process_roots(roots, synth.lcg_frame(topo.frame, synth.Lcg(1)))
    topic, rate, payload = nodes[1].publish_record()
    assert topic == b"VFO07" and rate == 12000 and payload == nodes[1].usb().tobytes()
    assert nodes[0].publish_record() is None  # a VFO with children publishes nothing itself


def test_dc_correct_and_u8():
    x = np.arange(20, dtype=np.float32)
    st = np.zeros(2, np.float32)
    y = x.copy()
    ob.dc_correct(y, st)
    ar = np.float32(0)
    keep, k = np.float32(1.0) - np.float32(0.000001), np.float32(0.000001)
    exp = []
    for v in x[0::2]:
        ar = np.float32(np.float32(ar * keep) + np.float32(k * v))
        exp.append(np.float32(v - ar))
    assert np.array_equal(y[0::2], np.array(exp, np.float32)) and st[0] == ar
    assert np.array_equal(ob.u8_to_float(np.array([0, 127, 255], np.uint8)), np.array([-127, 0, 128], np.float32))


def test_double_to_short_wrap():
    L = ob.load("port")
    f = L.fn("double_to_short")
    assert f(1009.99) == 1009 and f(-1009.99) == -1009 and f(32768.0) == -32768 and f(65537.5) == 1
    assert f(3e9) == 0 and f(float("nan")) == 0


@pytest.mark.parametrize("fixture", ["ofast_config1.npz", "ofast_profile_25e.npz", "ofast_54w.npz"])
def test_oracle_against_the_shipped_ofast_build(fixture):
    """The plain-C oracle (= the -O2 reference, bit for bit) against the committed outputs of the reference as shipped
    (-Ofast): streams within 1e-5, int16 within +-1 LSB and equal outside the fixture's patch."""
    from helpers import OFAST_FIXTURES, check_against_ofast_fixture
    g = golden(fixture)
    topo = golden_topology(OFAST_FIXTURES[fixture])
    nodes, roots = ob.build_tree("port", topo)
    lcg = synth.Lcg(1)
    for f in range(int(g["frames"])):
        ob.process_roots(roots, synth.lcg_frame(topo.frame, lcg))
        worst, patched, total = check_against_ofast_fixture(g, topo, f, lambda i: nodes[i].stream(), lambda i: nodes[i].usb())
        assert worst < 1e-6 and patched * 1000 < total


def test_capture_like_stream_through_the_shipped_profile():
    """BASELINE.json north_star's "recorded IQ", as far as this repository can have it (the reference ships no recording): the
    seeded capture-like byte stream (tuner noise, carriers past +-100, ADC offset, BPSK / OQPSK bursts on sdr_25E VFO
    frequencies, int16 audio to 29 656 of 32 767) through the whole sdr_25E profile with correct_dc_bias=1 -- the plain-C oracle
    against what the REAL reference build produced (tests/golden/make_golden.py capture): every stream, every payload, 8
    frames, bit for bit; and against the -Ofast build's payloads, rebuilt from the patch, within 1 LSB."""
    from helpers import capture_frames, check_capture_frame_exact
    g, topo, frames = capture_frames()
    nodes, roots = ob.build_tree("port", topo)
    state = np.zeros(2, np.float32)
    peak = 0
    for f, b in enumerate(frames):
        iq = ob.u8_to_float(b)
        ob.dc_correct(iq, state)
        assert np.array_equal(bits(state), bits(g[f"f{f}_dc_state"])) and sha(iq) == str(g[f"f{f}_raw_sha"])
        ob.process_roots(roots, iq, threads=4)
        check_capture_frame_exact(g, topo, f, lambda i: nodes[i].stream(), lambda i: nodes[i].usb())
        for i, v in enumerate(topo.vfos):
            if not topo.children(i):
                a = nodes[i].usb()
                peak = max(peak, int(np.abs(a.astype(np.int32)).max()))
                shipped = a.copy()
                shipped[g[f"f{f}_v{i}_ofast_pay_idx"]] = g[f"f{f}_v{i}_ofast_pay_val"]
                assert sha(shipped) == str(g[f"f{f}_v{i}_ofast_pay_sha"]) and np.abs(shipped.astype(np.int32) - a).max() <= 1, (f, i)
    assert 26000 < peak < 32768  # near full scale, never past it (a wrapped sample would make "+-1 LSB" meaningless)
