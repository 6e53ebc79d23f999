"""Parity tests proper: the HIP path, called through the C ABI, against the CPU oracle on the
same seeded inputs and against the committed reference fixtures.  Run with -m gpu on an MI355X.

Bars (BASELINE.json north_star): bit-exact for bytes/integers; for floating point
``max|gpu - ref| <= 1e-5 * max|ref|`` per VFO-frame on the final complex stream and on the
pre-quantisation float ``usb*gain*32768``, int16 within +-1 LSB.  The library's default
("exact") arithmetic is held to the stricter bar of bit-identity with the -O2 oracle; the
"fast" (FMA) arithmetic to the 1e-5 bar.
"""
import os

import numpy as np
import pytest

from helpers import ADVERSARIAL_CARRIERS, GOLDEN_TREES, REFERENCE_BUILDS_DIFFER, adversarial_frames, bits, golden, golden_topology, random_topology, sha
from oracle import binding as ob
from sdrreceiver_amd import synth, topology as tp

pytestmark = pytest.mark.gpu

REL_TOL = 1e-5  # north_star: "within 1e-5 relative float tolerance"


@pytest.fixture(scope="module")
def Receiver():
    from sdrreceiver_amd.receiver import Receiver as R
    return R


def _frames(topo, n, seed=1, tones=None):
    lcg = synth.Lcg(seed)
    for f in range(n):
        iq = synth.lcg_frame(topo.frame, lcg)
        if tones:
            iq = iq + synth.tone_frame(topo.frame, topo.fs, tones, f * topo.frame)
        yield f, iq


def _check_exact(rx, nodes, topo, ctx):
    for i, v in enumerate(topo.vfos):
        got = rx.stream(i, missing_ok=True)  # (None: a fused late decimation keeps no decimate[0]; its payload is checked)
        assert got is None or np.array_equal(bits(got), bits(nodes[i].stream())), (ctx, i, "stream")
        if not topo.children(i):
            want = nodes[i].usb() if v.demod_usb else nodes[i].iq()
            assert np.array_equal(rx.output(i), want), (ctx, i, "payload")


def _check_tolerance(rx, nodes, topo, ctx):
    for i, v in enumerate(topo.vfos):
        ref = nodes[i].stream()
        got = rx.stream(i, missing_ok=True)
        scale = float(np.abs(ref).max())
        assert got is None or np.abs(got - ref).max() <= REL_TOL * scale, (ctx, i, "stream", np.abs(got - ref).max() / scale)
        if not topo.children(i) and v.demod_usb:
            pre_ref = nodes[i].usb_prequant()
            pre = rx.prequant(i).astype(np.float64)
            s = float(np.abs(pre_ref).max())
            assert np.abs(pre - pre_ref).max() <= REL_TOL * s, (ctx, i, "prequant", np.abs(pre - pre_ref).max() / s)
            assert np.abs(rx.output(i).astype(np.int32) - nodes[i].usb().astype(np.int32)).max() <= 1, (ctx, i)


# ------------------------------------------------------------------------------ init-time state
def test_nco_tables_bit_exact(Receiver):
    g = golden("primitives.npz")
    topo = tp.Topology(fs=1536000, frame=384000)
    pairs = [tuple(int(x) for x in p) for p in g["nco_pairs"]]
    for fs, f in pairs:
        topo.vfos.append(tp.VfoDesc(parent=-1, fs=fs, decimate_count=0, mixer_freq=float(f), demod_usb=False,
                                    cstyle=1, samples_per_buffer=384000 if fs >= 384000 else fs // 4 // 16 * 16))
    # parent-less VFOs must share the frame length: one receiver per distinct frame length
    by_frame = {}
    for k, v in enumerate(topo.vfos):
        by_frame.setdefault(v.samples_per_buffer, []).append(k)
    for frame, idxs in by_frame.items():
        t = tp.Topology(fs=0, frame=frame, vfos=[topo.vfos[k] for k in idxs])
        rx = Receiver.from_topology(t)
        for local, k in enumerate(idxs):
            fs, f = pairs[k]
            table = rx.nco(local, 0, fs)
            assert np.array_equal(bits(table[:512]), bits(g[f"nco{k}_head"])), (fs, f)
            assert np.array_equal(bits(table[-64:]), bits(g[f"nco{k}_tail"])), (fs, f)
            assert sha(table) == str(g[f"nco{k}_sha"]), (fs, f)
        rx.close()


def test_designed_taps_bit_exact(Receiver):
    topo = golden_topology("54w")
    rx = Receiver.from_topology(topo)
    nodes, _ = ob.build_tree("port", topo)
    for i, v in enumerate(topo.vfos):
        if v.demod_usb:
            for which in ("fir_usb", "fir_dec", "hilbert"):
                assert np.array_equal(bits(rx.taps(i, which)), bits(nodes[i].taps(which))), (i, which)
    g = golden("primitives.npz")
    assert np.array_equal(bits(rx.taps(3, "fir_dec")), bits(g["lp4"]))  # (2, 240000, 24000, 12000): 49 taps
    assert np.array_equal(bits(rx.taps(3, "fir_usb")), bits(g["lp1"]))  # (2, 48000, 10000, 2500): 47 taps
    rx.close()


# ------------------------------------------------------------------------------ whole chains
# How a /5 or /6 leaf with decimate_count 0 runs (options "fuse_late", "keep_streams"): the default -- the decimating
# low-pass inside the mix wave, decimate[0] never written --, the same keeping decimate[0] of every frame, and the
# two-kernel form of rounds 1-3.  Results must not differ by one bit.
# Option "fuse_demod" (off by default: measured slower, include/sdrx.h): a d = 2 USB leaf below a parent -- the reference's
# 48 kS/s sub VFOs -- demodulates inside its mix wave and writes only its int16 payload; with keep_streams also decimate[2].
LATE_MODES = {"fused": dict(), "fused+streams": dict(keep_streams=True), "two kernels": dict(fuse_late=False),
              "demodulation in the wave": dict(fuse_demod=True), "demodulation in the wave+streams": dict(fuse_demod=True, keep_streams=True)}


def _mode_applies(topo, late):
    """does launch form `late` differ from the default on this tree?"""
    if late == "fused":
        return True
    if late.startswith("demodulation"):
        return any(v.demod_usb and v.parent >= 0 and not v.late_decimate and v.decimate_count == 2 for v in topo.vfos)
    return any(v.late_decimate for v in topo.vfos)


@pytest.mark.parametrize("late", sorted(LATE_MODES))
@pytest.mark.parametrize("fixture", sorted(GOLDEN_TREES))
def test_exact_mode_against_reference_fixtures(Receiver, fixture, late):
    """Default arithmetic vs the committed outputs of the real reference build: bit-identical
    payload and stream on every VFO and frame (crosses the NCO table wrap, > 1 s of signal)."""
    key, frames = GOLDEN_TREES[fixture]
    topo = golden_topology(key)
    if not _mode_applies(topo, late):
        pytest.skip("this launch form is the default one on this tree")
    g = golden(fixture)
    rx = Receiver.from_topology(topo, exact=True, **LATE_MODES[late])
    for f, iq in _frames(topo, frames):
        rx.process(iq)
        for i, v in enumerate(topo.vfos):
            s = rx.stream(i, missing_ok="streams" not in late)
            assert s is None or sha(s) == str(g[f"f{f}_v{i}_stream_sha"]), (fixture, f, i)
            if not topo.children(i):
                assert sha(rx.output(i)) == str(g[f"f{f}_v{i}_pay_sha"]), (fixture, f, i)
    rx.close()


@pytest.mark.parametrize("late", sorted(LATE_MODES))
@pytest.mark.parametrize("key,frames", [("config1", 5), ("profile_25e", 5), ("54w", 3), ("288k", 6), ("compress", 2)])
def test_exact_mode_against_live_oracle(Receiver, key, frames, late):
    topo = golden_topology(key)
    if not _mode_applies(topo, late):
        pytest.skip("this launch form is the default one on this tree")
    rx = Receiver.from_topology(topo, exact=True, **LATE_MODES[late])
    nodes, roots = ob.build_tree("port", topo)
    for f, iq in _frames(topo, frames, seed=11, tones=[(-377000.0, 25.0), (251000.0, 11.0)]):
        rx.process(iq)
        ob.process_roots(roots, iq)
        _check_exact(rx, nodes, topo, (key, f))
    rx.close()


@pytest.mark.parametrize("exact", [True, False])
@pytest.mark.parametrize("fixture", ["ofast_config1.npz", "ofast_profile_25e.npz", "ofast_54w.npz"])
def test_against_the_reference_as_shipped(Receiver, fixture, exact):
    """The reference's release build is -Ofast (SDRReceiver.pro:74-75); tests/golden/ofast_*.npz are its outputs (the -O2
    build is the canonical oracle because -Ofast leaves the summation order to the compiler).  The HIP path, exact and
    fast arithmetic, against them: every stream within 1e-5 of max|ref|, int16 audio within +-1 LSB -- and for the exact
    arithmetic EQUAL to the shipped build's audio everywhere but at the samples where the two reference builds
    themselves differ (2 / 120 / 66 of 9 000 / 657 000 / 156 000), proven by the sha of the whole -Ofast payload."""
    from helpers import OFAST_FIXTURES, check_against_ofast_fixture
    g = golden(fixture)
    topo = golden_topology(OFAST_FIXTURES[fixture])
    rx = Receiver.from_topology(topo, exact=exact, keep_streams=True)
    nodes, roots = ob.build_tree("port", topo)  # (the fast arithmetic needs the -O2 payload to rebuild the shipped one from)
    for f, iq in _frames(topo, int(g["frames"])):
        rx.process(iq)
        ob.process_roots(roots, iq)
        worst, patched, total = check_against_ofast_fixture(g, topo, f, rx.stream, rx.output,
                                                            o2_payload_of=None if exact else (lambda i: nodes[i].usb()))
        assert worst < 2e-6 and patched * 1000 < total, (fixture, f, worst)
    rx.close()


@pytest.mark.parametrize("arith", ["exact", "tolerance"])
@pytest.mark.parametrize("entry", ["bytes, DC removal on the device", "floats, DC removal on the host"])
def test_capture_like_stream_through_the_shipped_profile(Receiver, arith, entry):
    """BASELINE.json north_star: "match the reference CPU path ... on recorded IQ".  The reference holds no recording, so: the
    seeded capture-like byte stream (tests/golden/capture_25e.npz was made from it by the REAL reference build, -O2 and
    -Ofast: tuner noise, carriers that take the bytes past +-100, an ADC offset, BPSK / OQPSK bursts on sdr_25E VFO
    frequencies, int16 audio up to 29 656) through the shipped sdr_25E profile with correct_dc_bias=1, 8 frames = 2 s.
    Exact arithmetic: every stream and payload sha-identical to the -O2 reference's.  Tolerance arithmetic: every stream and
    pre-quantisation float within 1e-5 of max|ref|, int16 within 1 LSB of the -O2 AND of the shipped -Ofast build's.
    Both ingest forms: dongle bytes with LUT + DC-bias removal on the device (sdrx_process_u8: sdrj.cpp:155-160,271-286),
    and floats after the host-side DC removal (sdrx_process: sdrj::demodData's own argument)."""
    from helpers import capture_frames, check_capture_frame_exact
    g, topo, frames = capture_frames()
    exact = arith == "exact"
    rx = Receiver.from_topology(topo, exact=exact, keep_prequant=not exact)
    nodes, roots = ob.build_tree("port", topo)
    state = np.zeros(2, np.float32)
    for f, b in enumerate(frames):
        iq = ob.u8_to_float(b)
        ob.dc_correct(iq, state)
        if entry.startswith("bytes"):
            rx.process_u8(b, correct_dc=True)
        else:
            rx.process(iq)
        if exact:
            check_capture_frame_exact(g, topo, f, rx.stream, rx.output)
            continue
        ob.process_roots(roots, iq, threads=4)
        _check_tolerance(rx, nodes, topo, ("capture", f))
        for i, v in enumerate(topo.vfos):
            z = rx.stream(i)
            e = float(np.abs(z[::128] - g[f"f{f}_v{i}_ofast_stream_every128"]).max()) / float(g[f"f{f}_v{i}_ofast_stream_absmax"])
            assert e <= REL_TOL, (f, i, "stream vs the -Ofast build", e)
            if not topo.children(i):
                shipped = nodes[i].usb().copy()
                shipped[g[f"f{f}_v{i}_ofast_pay_idx"]] = g[f"f{f}_v{i}_ofast_pay_val"]
                assert np.abs(rx.output(i).astype(np.int32) - shipped.astype(np.int32)).max() <= 1, (f, i, "int16 vs the -Ofast build")
    rx.close()


@pytest.mark.parametrize("arith", ["tolerance", "robust"])
@pytest.mark.parametrize("key,frames", [("config1", 5), ("profile_25e", 3), ("54w", 3), ("288k", 3)])
def test_fast_mode_within_tolerance(Receiver, key, frames, arith):
    topo = golden_topology(key)
    rx = Receiver.from_topology(topo, exact=arith, keep_prequant=True)
    nodes, roots = ob.build_tree("port", topo)
    for f, iq in _frames(topo, frames, seed=5, tones=[(-377000.0, 25.0)]):
        rx.process(iq)
        ob.process_roots(roots, iq)
        _check_tolerance(rx, nodes, topo, (key, f))
    rx.close()


def _relative_errors(rx, nodes, topo):
    """per VFO: max|gpu - ref| / max|ref| on the final complex stream and (USB leaves) on the pre-quantisation float, and the
    largest int16 difference"""
    out = {}
    for i, v in enumerate(topo.vfos):
        ref = nodes[i].stream()
        got = rx.stream(i, missing_ok=True)
        e_s = float(np.abs(got - ref).max() / max(float(np.abs(ref).max()), 1e-30)) if got is not None else 0.0
        e_p, lsb = 0.0, 0
        if not topo.children(i) and v.demod_usb:
            pre_ref = nodes[i].usb_prequant()
            e_p = float(np.abs(rx.prequant(i).astype(np.float64) - pre_ref).max() / max(float(np.abs(pre_ref).max()), 1e-30))
            lsb = int(np.abs(rx.output(i).astype(np.int32) - nodes[i].usb().astype(np.int32)).max())
        out[v.topic or f"main{i}"] = (e_s, e_p, lsb)
    return out


@pytest.mark.parametrize("arith", ["tolerance", "robust", "exact"])
@pytest.mark.parametrize("case", sorted(ADVERSARIAL_CARRIERS))
def test_strong_carrier_over_quiet_channels(Receiver, case, arith):
    """where does the 1e-5 bar of the non-exact arithmetics break?  The shipped sdr_25E tree under +-1 LSB
    of noise and ONE carrier of 100 LSB -- 40 dB over everything a quiet VFO delivers -- outside every band, and inside
    another VFO's passband.  Rounding noise of an fp32 mixer scales with the TOTAL input, the bar with each VFO's own output:
    on this input the reference's OWN two builds (-O2 / -Ofast) differ by 2.1e-6 / 4.0e-6 of max|stream|
    (helpers.REFERENCE_BUILDS_DIFFER, measured by tests/test_oracle_vs_reference.py).  Measured here (round 6): the
    TOLERANCE arithmetic (NCO as rotations + FMA mixer and filters) 2.9e-6 / 6.4e-6 on the streams, 5.9e-6 / 8.3e-6 on the
    pre-quantisation floats -- inside the bar with 1.2x to spare: a carrier of 127 LSB would leave it; the ROBUST arithmetic
    (exact NCO: option exact = 2) 1.7e-6 / 4.3e-6 and 5.1e-6 / 6.9e-6 -- as close to the -O2 build as the shipped -Ofast
    build is, which is what it is held to (1.5x); the exact arithmetic is bit-identical.  A figure beyond 1e-5 is an xfail
    that carries the measurement."""
    topo = tp.profile_25e()
    rx = Receiver.from_topology(topo, exact=arith, keep_prequant=True)
    nodes, roots = ob.build_tree("port", topo)
    worst = {}
    for f, iq in adversarial_frames(topo, case):  # (5 frames > 1 s: the sub VFOs' tables wrap)
        rx.process(iq)
        ob.process_roots(roots, iq)
        if arith == "exact":
            _check_exact(rx, nodes, topo, (case, f))
            continue
        for k, (es, ep, lsb) in _relative_errors(rx, nodes, topo).items():
            w = worst.get(k, (0.0, 0.0, 0))
            worst[k] = (max(w[0], es), max(w[1], ep), max(w[2], lsb))
    rx.close()
    if arith == "exact":
        return
    top = sorted(worst.items(), key=lambda kv: -max(kv[1][0], kv[1][1]))
    w_s, w_p, w_lsb = max(v[0] for v in worst.values()), max(v[1] for v in worst.values()), max(v[2] for v in worst.values())
    report = (f"{arith}, {case}: worst stream {w_s:.2e}, worst pre-quantisation {w_p:.2e} of max|ref|, int16 within {w_lsb} LSB "
              f"(the reference's -O2 and -Ofast builds differ by {REFERENCE_BUILDS_DIFFER[case]:.2e} on the streams); "
              + ", ".join(f"{k} {max(v[0], v[1]):.1e}" for k, v in top[:4]))
    print(report)
    assert max(w_s, w_p) < 1e-3 and w_lsb <= 1, report     # (sanity: an arithmetic error, not a different signal)
    if arith == "robust":
        assert w_s <= 1.5 * REFERENCE_BUILDS_DIFFER[case] and max(w_s, w_p) <= REL_TOL, report
    elif max(w_s, w_p) > REL_TOL:
        pytest.xfail("the tolerance arithmetic leaves the 1e-5 bar on this input (use exact = 2): " + report)


@pytest.mark.parametrize("late", sorted(LATE_MODES))
def test_config4_256_vfos_vs_cpu(Receiver, late):
    """BASELINE config 4: 1.92 MS/s, 3 mains, 256 late-decimate subs with the 10 kHz low-pass: every payload (and, where
    decimate[0] is kept, every stream) of all 256 bit-identical to the oracle."""
    topo = tp.config4(256)
    if not _mode_applies(topo, late):
        pytest.skip("this launch form is the default one on this tree")
    rx = Receiver.from_topology(topo, exact=True, **LATE_MODES[late])
    nodes, roots = ob.build_tree("port", topo)
    for f, iq in _frames(topo, 2, seed=2):
        rx.process(iq)
        ob.process_roots(roots, iq, threads=8)
        _check_exact(rx, nodes, topo, ("config4", late, f))
    rx.close()


def _demod_tree(frame_root=38400 * 8, n_extra=0):
    """one main VFO (d = 3) with d = 2 USB leaves under it whose audio low-passes have 0 / 31 / 47 / 61 / 155 taps (the last
    one too long for the wave: it keeps k_usb_demod), plus a d = 5 leaf for company"""
    fs = frame_root * 4
    t = tp.Topology(fs=fs, frame=frame_root, name="demod-in-wave")
    t.vfos.append(tp.VfoDesc(parent=-1, fs=fs, decimate_count=3, mixer_freq=float(fs // 5), demod_usb=False, cstyle=1, samples_per_buffer=frame_root))
    fs1, n1 = fs >> 3, frame_root >> 3
    out = fs1 >> 2
    for k, bw in enumerate([0, out // 3, int(out / 4.8), int(out / 6.3), out // 16] + [0, int(out / 4.8)] * n_extra):
        t.vfos.append(tp.VfoDesc(topic=f"W{k:03d}", parent=0, fs=fs1, decimate_count=2, mixer_freq=float((k * 7919) % (fs1 // 3) - fs1 // 6),
                                 filter_bw=bw, gain=tp._g(0.05), cstyle=1, samples_per_buffer=n1))
    t.vfos.append(tp.VfoDesc(topic="D5", parent=0, fs=fs1, decimate_count=5, mixer_freq=1234.0, gain=tp._g(0.05), cstyle=1, samples_per_buffer=n1))
    return t


@pytest.mark.parametrize("arith", ["exact", "tolerance", "robust"])
@pytest.mark.parametrize("frame_root,segments", [(38400 * 8, 0), (38400 * 8, 1), (38400 * 8, 5), (16 * 128 * 9, 0), (16 * 128 * 9, 2), (16 * 128 * 25 + 16 * 128, 3)])
def test_demodulation_inside_the_mix_wave(Receiver, frame_root, segments, arith):
    """Option fuse_demod (off by default -- measured slower): vfo::usb_demod (vfo.cpp:300-332) of a d = 2 leaf
    inside its mix wave.  Leaves without / with audio low-passes of several lengths (N mod 4 = 3, 1, ...; one too long for the
    wave), frames whose last chunk is partial, one / several time segments per VFO-frame (a segment that starts inside the
    frame warms the Hilbert and low-pass windows up from zero), synchronous frames and frames queued through the launch
    pipeline, the spectrum tap: payloads, pre-quantisation floats and (kept) streams against the oracle -- bit for bit in the
    exact arithmetic, within 1e-5 / 1 LSB otherwise -- and bit-identical to the k_usb_demod form in the exact and the robust
    arithmetic."""
    import torch
    from sdrreceiver_amd.receiver import SdrxError
    topo = _demod_tree(frame_root)
    lens = []
    rx = Receiver.from_topology(topo, exact=arith, fuse_demod=True, keep_prequant=True, segments=segments)
    other = Receiver.from_topology(topo, exact=arith, fuse_demod=False, keep_prequant=True, segments=segments)
    kept = Receiver.from_topology(topo, exact=arith, fuse_demod=True, keep_streams=True, keep_prequant=True, segments=max(0, segments - 1))
    for i in range(1, 6):
        lens.append(len(rx.taps(i, "fir_usb")) if topo.vfos[i].filter_bw else 0)
    assert lens[0] == 0 and lens[1] % 4 != lens[2] % 4 and max(lens[:4]) <= 64 < lens[4], lens
    nodes, roots = ob.build_tree("port", topo)
    check = _check_exact if arith == "exact" else _check_tolerance
    frames = [iq for _, iq in _frames(topo, 7, seed=9, tones=[(topo.fs / 5.0 + 900.0, 30.0), (-topo.fs / 3.1, 9.0)])]
    for f, iq in enumerate(frames[:4]):   # synchronous frames
        for r in (rx, other, kept):
            r.process(iq)
        ob.process_roots(roots, iq)
        check(rx, nodes, topo, ("in the wave", f))
        check(kept, nodes, topo, ("in the wave + streams", f))
        # (the tolerance arithmetic replays the NCO table exactly in the chunks that touch its first 512 entries or wrap, and the
        # two forms cut a VFO-frame into different segments, hence chunks: there the forms agree within the tolerance, not bit for bit)
        for i in range(1, len(topo.vfos) if arith != "tolerance" else 0):
            assert np.array_equal(rx.output(i), other.output(i)) and np.array_equal(bits(rx.prequant(i)), bits(other.prequant(i))), (f, i)
            assert np.array_equal(bits(kept.stream(i)), bits(other.stream(i))), (f, i)
        if f == 0:  # a leaf that demodulates in its wave keeps no decimate[2] unless asked
            with pytest.raises(SdrxError) as e:
                rx.stream(1)
            assert "demodulates inside the mix wave" in str(e.value) and rx.stream(5) is not None and rx.stream(6) is not None
        if f == 1:
            rx.set_tap(2)      # fftVFOSlot: from the next frame on
        if f >= 2:
            assert arith == "tolerance" or np.array_equal(bits(rx.stream(2)), bits(other.stream(2)))
            assert np.abs(rx.stream(2) - other.stream(2)).max() <= REL_TOL * np.abs(other.stream(2)).max()
    rx.set_tap(-1)
    dev = [torch.from_numpy(iq).cuda() for iq in frames[4:]]
    torch.cuda.synchronize()
    for d, iq in zip(dev, frames[4:]):     # ... and queued through the launch pipeline
        rx.process_device(d.data_ptr(), topo.frame)
        other.process_device(d.data_ptr(), topo.frame)
        ob.process_roots(roots, iq)
    rx.fetch()
    other.fetch()
    check(rx, nodes, topo, ("in the wave, queued", 6))
    for i in range(1, len(topo.vfos)):
        assert np.abs(rx.output(i).astype(np.int32) - other.output(i).astype(np.int32)).max() <= (1 if arith == "tolerance" else 0), ("queued", i)
    for r in (rx, other, kept):
        r.close()


def test_several_spectrum_taps_at_once(Receiver):
    """vfo::fftVFOSlot sets emitFFT on every VFO whose topic equals the selected string (vfo.cpp:492-509), so several VFOs
    can be taps at once: sdrx_add_tap adds to the selection, sdrx_set_tap replaces it.  All the /5 leaves of the 54W tree
    as taps together (each keeps decimate[0] in a buffer of its own), then a smaller selection, then none: the tapped
    streams are the oracle's bit for bit, the others answer SDRX_ENOSTREAM, payloads never change."""
    from sdrreceiver_amd import _lib
    from sdrreceiver_amd.receiver import SdrxError
    topo = golden_topology("54w")
    fused = [i for i, v in enumerate(topo.vfos) if v.late_decimate and v.decimate_count == 0 and v.parent >= 0]
    assert len(fused) >= 4
    rx = Receiver.from_topology(topo, exact=True)
    nodes, roots = ob.build_tree("port", topo)
    selections = [fused, fused[1:3], [fused[0], 0], [], fused[-2:]]
    for f, iq in _frames(topo, len(selections), seed=29, tones=[(700000.0, 30.0)]):
        rx.set_taps(selections[f])
        rx.process(iq)
        ob.process_roots(roots, iq)
        for i in fused:
            if i in selections[f]:
                assert np.array_equal(bits(rx.stream(i)), bits(nodes[i].stream())), (f, i)
            else:
                with pytest.raises(SdrxError) as e:
                    rx.stream(i)
                assert e.value.code == _lib.SDRX_ENOSTREAM and rx.stream(i, missing_ok=True) is None
        for i, v in enumerate(topo.vfos):
            if not topo.children(i):
                assert np.array_equal(rx.output(i), nodes[i].usb() if v.demod_usb else nodes[i].iq()), (f, i)
    rx.close()


def test_the_spectrum_tap_on_a_fused_late_decimation(Receiver):
    """fftVFOSlot on the 54W tree: a /5 leaf whose low-pass runs inside the mix wave keeps decimate[0] only while it is
    the tap (sdrx_set_tap), from the frame after the selection on; moving the tap moves the stream; every other stream
    of the tree stays available; the payloads do not care."""
    from sdrreceiver_amd.receiver import SdrxError
    topo = golden_topology("54w")
    fused = [i for i, v in enumerate(topo.vfos) if v.late_decimate and v.decimate_count == 0 and v.parent >= 0]
    kept = [i for i in range(len(topo.vfos)) if i not in fused]
    assert len(fused) >= 2 and kept
    rx = Receiver.from_topology(topo, exact=True)
    nodes, roots = ob.build_tree("port", topo)
    frames = [iq for _, iq in _frames(topo, 6, seed=23, tones=[(700000.0, 30.0)])]
    taps = [-1, fused[0], fused[0], fused[-1], kept[0], -1]  # selected BEFORE frame f
    for f, iq in enumerate(frames):
        rx.set_tap(taps[f])
        rx.process(iq)
        ob.process_roots(roots, iq)
        for i in fused:
            if i == taps[f]:
                assert np.array_equal(bits(rx.stream(i)), bits(nodes[i].stream())), (f, i)
            else:
                with pytest.raises(SdrxError):
                    rx.stream(i)
        for i in kept:
            assert np.array_equal(bits(rx.stream(i)), bits(nodes[i].stream())), (f, i)
        for i in topo.leaves_in_publish_order():
            want = nodes[i].usb() if topo.vfos[i].demod_usb else nodes[i].iq()
            assert np.array_equal(rx.output(i), want), (f, i)
    rx.close()


@pytest.mark.parametrize("key", ["54w", "288k"])
def test_time_segmentation_of_a_fused_late_decimation(Receiver, key):
    """The /5 (960-sample chunks) and /6 (1008) walks cut into 1, 2, 5 and 13 segments per VFO-frame, each segment but
    the first starting from zero history 80 / 96 samples early: not a bit changes, and it is the oracle's result."""
    topo = golden_topology(key)
    nodes, roots = ob.build_tree("port", topo)
    frames = [iq for _, iq in _frames(topo, 3, seed=6, tones=[(333000.0, 12.0)])]
    want = []
    for iq in frames:
        ob.process_roots(roots, iq)
        want.append([(nodes[i].usb() if topo.vfos[i].demod_usb else nodes[i].iq()) for i in topo.leaves_in_publish_order()])
    for seg in (1, 2, 5, 13):
        rx = Receiver.from_topology(topo, exact=True, segments=seg, keep_streams=seg == 5)
        for f, iq in enumerate(frames):
            rx.process(iq)
            for k, i in enumerate(topo.leaves_in_publish_order()):
                assert np.array_equal(rx.output(i), want[f][k]), (key, seg, f, i)
        rx.close()


# ------------------------------------------------------------------------------ structure
@pytest.mark.parametrize("fuse", [True, False])
def test_time_segmentation_is_invisible(Receiver, fuse):
    """Splitting a VFO-frame into time segments (with warm-up chunks) must not change a bit."""
    topo = golden_topology("profile_25e")
    outs = []
    for seg in (1, 3, 7):
        rx = Receiver.from_topology(topo, exact=True, segments=seg, fuse=fuse)
        frames = []
        for f, iq in _frames(topo, 3, seed=4):
            rx.process(iq)
            frames.append([rx.stream(i) for i in range(len(topo.vfos))] +
                          [rx.output(i) for i in topo.leaves_in_publish_order()])
        outs.append(frames)
        rx.close()
    for other in outs[1:]:
        for fa, fb in zip(outs[0], other):
            for a, b in zip(fa, fb):
                assert np.array_equal(bits(a), bits(b))


LAUNCH_MODES = {  # how a frame's kernels are launched (options of sdrx_set_option); results must not differ by one bit
    "one-launch, pipelined": dict(fuse=True, frame_pipeline=True),   # default: k_mix_levels, level l of frame k-l in one launch
    "one-launch per part": dict(fuse=True, frame_pipeline=False),    # k_mix_levels, one level per launch
    "separate kernels": dict(fuse=False),                            # k_mix_decimate per level
    "two streams": dict(fuse=False, pipeline=True),                  # ... with the leaf tail on a second stream
}


@pytest.mark.parametrize("mode", sorted(LAUNCH_MODES))
@pytest.mark.parametrize("key", ["profile_25e", "54w", "compress"])
def test_async_frames_back_to_back(Receiver, key, mode):
    """sdrx_process_device is asynchronous: 9 frames queued on a caller's stream without any
    synchronisation in between (level 0 of a frame may start while the previous frame's
    demodulation is still running), one sdrx_fetch at the end.  Streams and payloads equal the
    oracle's bit for bit."""
    import torch
    topo = golden_topology(key)
    nodes, roots = ob.build_tree("port", topo)
    frames = [iq for _, iq in _frames(topo, 9, seed=13)]
    for iq in frames:
        ob.process_roots(roots, iq)
    st = torch.cuda.Stream()
    rx = Receiver.from_topology(topo, exact=True, **LAUNCH_MODES[mode])
    rx.set_stream(st.cuda_stream)
    with torch.cuda.stream(st):
        dev = [torch.from_numpy(iq).cuda(non_blocking=True) for iq in frames]
        for d in dev:
            rx.process_device(d.data_ptr(), topo.frame)
    rx.fetch()
    _check_exact(rx, nodes, topo, ("async", key))
    rx.close()


@pytest.mark.parametrize("pipeline", [True, False])
def test_submit_wait_delivers_frames_in_order(Receiver, pipeline):
    """The pipelined interface (SURVEY 8b "async submit/wait pair"): submit(f+1); wait() -> f.  Every
    delivered frame carries exactly the messages ZmqPublisher::publish would get for THAT frame
    (topic, rate, payload; main order x sub order, vfo.cpp:426-453) while the next frame is already
    queued behind it; the input buffer is borrowed for the duration of the call only; the in-flight
    limit and the exclusion of the synchronous calls are enforced; afterwards the synchronous
    interface continues the same stream of frames."""
    from sdrreceiver_amd.receiver import SdrxError
    topo = golden_topology("profile_25e")
    rx = Receiver.from_topology(topo, pipeline=pipeline)
    nodes, roots = ob.build_tree("port", topo)
    order = topo.leaves_in_publish_order()
    frames = [iq for _, iq in _frames(topo, 9, seed=3, tones=[(-377000.0, 25.0)])]
    want = []
    for iq in frames:
        ob.process_roots(roots, iq)
        want.append([(topo.vfos[i].topic.encode()[:5].ljust(5, b"\0"), topo.vfos[i].output_rate, nodes[i].usb().tobytes()) for i in order])
    scratch = np.empty_like(frames[0])

    def submit(f):
        scratch[:] = frames[f]
        rx.submit(scratch)
        scratch[:] = -77.0  # the caller's buffer is free again as soon as the call returns

    submit(0)
    assert rx.in_flight() == 1
    for f in range(1, 7):
        submit(f)
        assert rx.in_flight() == 2
        if f == 1:
            for call in (lambda: rx.submit(frames[f]), lambda: rx.process(frames[f]), rx.fetch, lambda: rx.output(order[0])):
                with pytest.raises(SdrxError) as e:
                    call()
                assert e.value.code == -2
        rx.wait()
        assert rx.published == want[f - 1], (pipeline, f - 1)
        assert rx.in_flight() == 1
    rx.wait()
    assert rx.published == want[6] and rx.in_flight() == 0
    assert rx.output(order[-1]).tobytes() == want[6][-1][2]  # get_output serves the delivered frame
    with pytest.raises(SdrxError) as e:
        rx.wait()
    assert e.value.code == -2
    rx.process(frames[7])
    assert rx.published == want[7]
    rx.submit(frames[8])
    rx.wait()
    assert rx.published == want[8]
    rx.close()


def test_submit_u8_with_dc_correction(Receiver):
    """sdrx_submit_u8: dongle bytes through the pipelined interface, DC-bias IIR on the device with its
    state carried from frame to frame while two frames are in flight."""
    topo = tp.config2()
    rx = Receiver.from_topology(topo)
    nodes, roots = ob.build_tree("port", topo)
    order = topo.leaves_in_publish_order()
    rng = np.random.default_rng(9)
    state = np.zeros(2, np.float32)
    frames, want = [], []
    for f in range(4):
        b = rng.integers(0, 256, 2 * topo.frame, dtype=np.uint8)
        b[1::2] = np.clip(b[1::2].astype(int) // 8 + 120, 0, 255)
        iq = ob.u8_to_float(b)
        ob.dc_correct(iq, state)
        ob.process_roots(roots, iq)
        frames.append(b)
        want.append([nodes[i].usb().tobytes() for i in order])
    rx.submit_u8(frames[0], correct_dc=True)
    for f in range(1, 4):
        rx.submit_u8(frames[f], correct_dc=True)
        rx.wait()
        assert [p for _, _, p in rx.published] == want[f - 1], f - 1
    rx.wait()
    assert [p for _, _, p in rx.published] == want[3]
    rx.close()


def test_payload_copies_issued_by_wait(Receiver):
    """Frames that carry the DC recurrence get their payload copy issued by sdrx_wait, the others have
    it queued at submit time: both kinds in one stream of frames, two in flight, an sdrx_sync between submit and wait (the
    copy is still owed afterwards), a synchronous frame after the queue has drained, and a context destroyed with frames in
    flight -- payloads as the oracle's for every delivered frame."""
    topo = tp.config2()
    rx = Receiver.from_topology(topo)
    nodes, roots = ob.build_tree("port", topo)
    order = topo.leaves_in_publish_order()
    rng = np.random.default_rng(19)
    state = np.zeros(2, np.float32)
    kinds = [True, True, False, True, False, False, True, True, True]  # correct_dc per frame
    frames, want = [], []
    for dc in kinds:
        b = rng.integers(0, 256, 2 * topo.frame, dtype=np.uint8)
        b[0::2] = np.clip(b[0::2].astype(int) // 8 + 100, 0, 255)
        iq = ob.u8_to_float(b)
        if dc:
            ob.dc_correct(iq, state)
        ob.process_roots(roots, iq)
        frames.append(b)
        want.append([nodes[i].usb().tobytes() for i in order])
    rx.submit_u8(frames[0], correct_dc=kinds[0])
    for f in range(1, 6):
        rx.submit_u8(frames[f], correct_dc=kinds[f])
        if f % 2:
            rx.sync()  # everything queued has run; a copy that is sdrx_wait's to issue has not been issued
        rx.wait()
        assert [p for _, _, p in rx.published] == want[f - 1], f - 1
    rx.wait()
    assert [p for _, _, p in rx.published] == want[5]
    rx.process_u8(frames[6], correct_dc=kinds[6])
    assert [p for _, _, p in rx.published] == want[6]
    rx.submit_u8(frames[7], correct_dc=kinds[7])
    rx.submit_u8(frames[8], correct_dc=kinds[8])
    rx.close()  # two frames in flight, both copies still owed


def test_publish_order_and_framing(Receiver):
    """Callback order = main order x sub order (sdrj.cpp:288-294, vfo.cpp:257-263); topic is
    exactly 5 bytes, rate is outputRate, payload is the int16 audio (zmqpublisher.cpp:82-96)."""
    topo = tp.config2()
    rx = Receiver.from_topology(topo)
    rx.process(synth.lcg_frame(topo.frame, synth.Lcg(1)))
    order = topo.leaves_in_publish_order()
    assert len(rx.published) == len(order) == 32
    for (topic, rate, payload), i in zip(rx.published, order):
        v = topo.vfos[i]
        assert topic == v.topic.encode()[:5].ljust(5, b"\0") and rate == v.output_rate
        assert payload == rx.output(i).tobytes() and len(payload) == 2 * v.n_out
    rx.close()


def test_vfos_are_independent_and_shardable(Receiver):
    """A VFO's output does not depend on which other VFOs share the GPU (shard invariance):
    config-3 topology with 64 subs, whole vs the 4 shards of sdrreceiver_amd.topology.shard."""
    topo = tp.config3(64)
    rx = Receiver.from_topology(topo)
    shards = [(tp.shard(topo, r, 4), None) for r in range(4)]
    shards = [(t, Receiver.from_topology(t)) for t, _ in shards]
    for f, iq in _frames(topo, 2, seed=9):
        rx.process(iq)
        whole = {topo.vfos[i].topic: rx.output(i) for i in topo.leaves_in_publish_order()}
        seen = 0
        for t, r in shards:
            r.process(iq)
            for i in t.leaves_in_publish_order():
                assert np.array_equal(r.output(i), whole[t.vfos[i].topic])
                seen += 1
        assert seen == 64
    rx.close()
    for _, r in shards:
        r.close()


@pytest.mark.parametrize("members", [2, 3])
@pytest.mark.parametrize("key", ["profile_25e", "config3-64", "54w"])
def test_group_of_contexts_in_one_process(members, key):
    """The native multi-device host (sdrx_group_*): one process, one context per device entry -- here
    2 or 3 shards on the one GPU of the test box, so the fan-out runs as device copies instead of xGMI
    peer copies -- the tree partitioned in C++ like topology.shard, raw frames fanned out from the
    first member, payloads published in the reference's order over the WHOLE tree (main order x sub
    order, sdrj.cpp:288-294 / vfo.cpp:257-263) whichever member computed them.  Every message of every
    frame equals the oracle's; synchronous and pipelined (submit / wait) interface; byte input."""
    _group_body([0] * members, key)


def test_group_over_distinct_devices():
    """The same, over DISTINCT device ordinals: runs by itself wherever the box has two or more GPUs (the
    1-GPU test box skips it) -- the branches no same-device run reaches: hipMemcpyPeerAsync between two
    devices, peer-access enabling, the cross-device hipStreamWaitEvent on the first device's frame event.
    A host-staged fallback of the peer copies is not an error but must be visible (sdrx_group_peer_access)."""
    import torch
    n = torch.cuda.device_count()
    if n < 2:
        pytest.skip(f"{n} GPU(s) here: the distinct-device group needs >= 2")
    for key in ("profile_25e", "config3-64"):
        g = _group_body(list(range(min(n, 8))), key)
        print(f"group over {min(n, 8)} devices, {key}: peer access {'direct' if g else 'HOST-STAGED'}")


def _group_body(devices, key):
    from sdrreceiver_amd.receiver import Group, SdrxError
    members = len(devices)
    topo = tp.config3(64) if key == "config3-64" else golden_topology(key)
    g = Group.from_topology(topo, devices)
    peer = g.peer_access()
    assert peer or len(set(devices)) > 1  # members of ONE device always reach the frame directly
    nodes, roots = ob.build_tree("port", topo)
    order = topo.leaves_in_publish_order()

    def want():
        return [(topo.vfos[i].topic.encode()[:5].ljust(5, b"\0"), topo.vfos[i].output_rate,
                 (nodes[i].usb() if topo.vfos[i].demod_usb else nodes[i].iq()).tobytes()) for i in order
                if topo.vfos[i].demod_usb or topo.vfos[i].topic]

    # the partition: every leaf on exactly one member, blocks in order, replicated mains on none
    where = [g.locate(i)[0] for i in range(len(topo.vfos))]
    assert all(w >= 0 for w in where)
    for r in topo.roots():
        ch = topo.children(r)
        if ch:
            assert [where[c] for c in ch] == sorted(where[c] for c in ch)
            assert [where[c] for c in ch] == [next(k for k in range(members) if (len(ch) * k) // members <= q < (len(ch) * (k + 1)) // members)
                                              for q in range(len(ch))]
    frames = [iq for _, iq in _frames(topo, 6, seed=12, tones=[(-377000.0, 25.0)])]
    for f in range(2):  # synchronous
        g.process(frames[f])
        ob.process_roots(roots, frames[f])
        assert g.published == want(), (key, members, f)
        assert np.array_equal(g.output(order[0]), nodes[order[0]].usb() if topo.vfos[order[0]].demod_usb else nodes[order[0]].iq())
    wants = []
    for f in range(2, 6):
        ob.process_roots(roots, frames[f])
        wants.append(want())
    g.submit(frames[2])  # pipelined: submit(f+1); wait() -> f
    for f in range(3, 6):
        g.submit(frames[f])
        assert g.in_flight() == 2
        if f == 3:
            with pytest.raises(SdrxError) as e:
                g.submit(frames[f])
            assert e.value.code == -2
        g.wait()
        assert g.published == wants[f - 3], (key, members, f - 1)
    g.wait()
    assert g.published == wants[3]
    if key != "config3-64":  # dongle bytes: LUT on every member
        lcg = synth.Lcg(5)
        b = synth.lcg_frame_u8(topo.frame, lcg)
        g.submit_u8(b)
        g.wait()
        ob.process_roots(roots, ob.u8_to_float(b))
        assert g.published == want()
    import torch  # frames that are already on the first device, queued without payload copies
    dev = [torch.from_numpy(iq).to(f"cuda:{devices[0]}") for iq in frames[:3]]
    torch.cuda.synchronize(devices[0])
    for f, d in enumerate(dev):
        g.process_device(d.data_ptr(), topo.frame)
        ob.process_roots(roots, frames[f])
    g.sync()
    for i in order:
        ref = nodes[i].usb() if topo.vfos[i].demod_usb else nodes[i].iq()
        assert np.array_equal(g.output(i), ref), (key, members, "device", i)
    st = g.member_stats()
    assert sum(s["n_leaves"] for s in st if s) == len(order)
    g.close()
    return peer


def test_group_dongle_bytes_with_dc_bias_removal():
    """The shipped sdr_25E.ini sets correct_dc_bias=1 (sdrj.cpp:271-286): dongle bytes into a device list
    with the DC-bias IIR on.  Every member runs the recurrence itself on the fanned-out BYTES with an
    accumulator of its own; identical bytes and start state keep the members' estimates identical, so
    every leaf of every member equals the oracle fed with the host-side LUT + sequential fp32 IIR -- 1
    synchronous + 4 pipelined frames (the state carries across frames while two are in flight), then a
    frame WITHOUT the correction (the accumulators keep their state, the frame is not touched)."""
    from sdrreceiver_amd.receiver import Group
    topo = tp.profile_25e()
    g = Group.from_topology(topo, [0, 0, 0])
    nodes, roots = ob.build_tree("port", topo)
    order = topo.leaves_in_publish_order()
    lcg = synth.Lcg(21)
    frames = [np.clip(synth.lcg_frame_u8(topo.frame, lcg).astype(np.int32) + 9, 0, 255).astype(np.uint8) for _ in range(6)]  # a DC offset of ~9 LSB
    state = np.zeros(2, np.float32)

    def want(b, dc):
        x = ob.u8_to_float(b)
        if dc:
            ob.dc_correct(x, state)
        ob.process_roots(roots, x)
        return [(topo.vfos[i].topic.encode()[:5].ljust(5, b"\0"), topo.vfos[i].output_rate, nodes[i].usb().tobytes()) for i in order]

    g.process_u8(frames[0], correct_dc=True)
    assert g.published == want(frames[0], True)
    g.submit_u8(frames[1], correct_dc=True)
    for f in range(2, 5):
        g.submit_u8(frames[f], correct_dc=True)
        g.wait()
        assert g.published == want(frames[f - 1], True), f - 1
    g.wait()
    assert g.published == want(frames[4], True)
    assert abs(float(state[0])) > 1e-3  # the estimate moved: the correction did something
    g.process_u8(frames[5], correct_dc=False)
    assert g.published == want(frames[5], False)
    g.close()


def test_group_wide_level0_takes_bytes_too():
    """Parent-less leaves (the flat workloads; more than 4 of them: level 0 gets a layout pass) block-
    partitioned over a group, fed with dongle bytes with and without the DC-bias removal."""
    from sdrreceiver_amd.receiver import Group
    topo = tp.config3_flat(12)
    g = Group.from_topology(topo, [0, 0])
    nodes, roots = ob.build_tree("port", topo)
    order = topo.leaves_in_publish_order()
    lcg = synth.Lcg(4)
    state = np.zeros(2, np.float32)
    for f, dc in enumerate((False, True, True)):
        b = synth.lcg_frame_u8(topo.frame, lcg)
        x = ob.u8_to_float(b)
        if dc:
            ob.dc_correct(x, state)
        ob.process_roots(roots, x, threads=4)
        g.process_u8(b, correct_dc=dc)
        for i in order:
            assert np.array_equal(g.output(i), nodes[i].usb()), (f, dc, i)
    g.close()


def test_group_refuses_mains_with_different_frames_and_leaves_nothing_behind():
    """every parent-less VFO consumes the same raw frame (sdrj.cpp:288-294).  A tree whose
    roots disagree on samples_per_buffer is refused by sdrx_group_finalize -- whichever member a root
    lands on (a single context refuses it too) -- the failed finalize leaves no contexts or device
    memory behind, and the group can still only be destroyed or, after the failure, not be used."""
    import torch
    from sdrreceiver_amd.receiver import Group, SdrxError
    topo = tp.config3_flat(8)
    topo.vfos[5].samples_per_buffer = topo.frame // 2  # lands on the second member
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(3):
        with pytest.raises(SdrxError) as e:
            Group.from_topology(topo, [0, 0])
        assert e.value.code == -1 and "samples_per_buffer" in str(e.value)
    # a failure INSIDE a member's finalize (an audio filter the reference rejects, firfilter.cpp:122-134)
    # after an earlier member was already built
    t2 = tp.config3(16)
    last = t2.leaves_in_publish_order()[-1]
    t2.vfos[last].filter_bw = 40000
    with pytest.raises(SdrxError) as e:
        Group.from_topology(t2, [0, 0])
    assert e.value.code == -3
    import gc
    gc.collect()
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] >= free0 - (8 << 20), "a failed sdrx_group_finalize left device memory behind"


def test_group_with_more_devices_than_sub_vfos():
    """More members than sub VFOs: a member may hold one main VFO with a single sub, or nothing at all
    (even the first one, the ingest device) -- never a main VFO turned IQ-publishing leaf."""
    from sdrreceiver_amd.receiver import Group
    topo = tp.profile_25e()
    g = Group.from_topology(topo, [0] * 16)  # 12 + 15 subs over 16 members: some members hold one main only
    nodes, roots = ob.build_tree("port", topo)
    iq = synth.lcg_frame(topo.frame, synth.Lcg(4))
    g.process(iq)
    ob.process_roots(roots, iq)
    order = topo.leaves_in_publish_order()
    assert [p for _, _, p in g.published] == [nodes[i].usb().tobytes() for i in order]
    g.close()
    c1 = tp.config1()  # the single sub lands on member 1: member 0 (the ingest device) holds nothing
    g = Group.from_topology(c1, [0, 0])
    assert g.member_stats()[0] is None and g.locate(1) == (1, 1)
    nodes, roots = ob.build_tree("port", c1)
    for f, iq in _frames(c1, 3, seed=6):
        g.process(iq)
        ob.process_roots(roots, iq)
        assert g.published == [(b"VFO01", 12000, nodes[1].usb().tobytes())]
    g.close()


def test_full_size_properties_config3(Receiver):
    """BASELINE config 3 at full size (2 mains + 1 024 sub VFOs): EVERY VFO -- final complex stream
    and int16 payload -- bit for bit against the oracle for 3 frames (the oracle runs its sub VFOs
    on all host cores: ~0.1-0.5 s per frame), plus two size-independent properties: VFOs with
    identical parameters produce identical output, and an all-zero frame after start-up yields
    all-zero audio (zero state, linear chain)."""
    import os
    topo = tp.config3(1024)
    topo.vfos.append(tp.VfoDesc(**{**topo.vfos[5].__dict__, "topic": "DUP05"}))  # duplicate of a main0 sub
    rx = Receiver.from_topology(topo)
    nodes, roots = ob.build_tree("port", topo)
    threads = len(os.sched_getaffinity(0))
    for f, iq in _frames(topo, 3, seed=1):
        rx.process(iq)
        ob.process_roots(roots, iq, threads=threads)
        _check_exact(rx, nodes, topo, ("config3-full", f))
        assert np.array_equal(rx.output(5), rx.output(len(topo.vfos) - 1))
    for r in roots:
        r.free()
    rx.close()
    rx = Receiver.from_topology(topo)
    rx.process(np.zeros(2 * topo.frame, np.float32))
    for i in topo.leaves_in_publish_order():
        assert not rx.output(i).any()
    rx.close()


def test_u8_dc_bias_blocked_scan_option(Receiver):
    """Option dc_blocked_scan=1: the DC-bias IIR as a blocked parallel scan instead
    of the sequential rounded recurrence.  The scan evaluates the TRUE linear filter: fed through
    the oracle chain, the float64 IIR response reproduces the GPU result to the usual tolerance.
    The reference's fp32 recurrence itself wanders around the true response (its rounding errors
    are correlated from step to step: up to 8e-3 here at a 3-LSB offset), which is why the option
    is off by default; the deviation is measured and bounded here.  The scan must also be >= 15x
    faster than the exact recurrence (4.5 ms per frame)."""
    import scipy.signal as sg
    topo = tp.config2()
    rx = Receiver.from_topology(topo, exact=True, keep_prequant=True, dc_blocked_scan=True)
    nodes, roots = ob.build_tree("port", topo)          # fed with the true IIR response
    rnodes, rroots = ob.build_tree("port", topo)        # fed with the reference's recurrence
    rng = np.random.default_rng(5)
    A = float(np.float32(1.0) - np.float32(0.000001))
    B = float(np.float32(0.000001))
    z = [np.zeros(1), np.zeros(1)]
    state = np.zeros(2, np.float32)
    rx.enable_kernel_timing(True)
    worst = 0.0
    for f in range(6):
        b = rng.integers(0, 256, 2 * topo.frame, dtype=np.uint8)
        b[0::2] = np.clip(b[0::2].astype(int) + 3, 0, 255)
        b[1::2] = np.clip(b[1::2].astype(int) - 2, 0, 255)
        rx.process_u8(b, correct_dc=True)
        x = ob.u8_to_float(b)
        iq = x.copy()
        for comp in (0, 1):
            a, z[comp] = sg.lfilter([B], [1.0, -A], x[comp::2].astype(np.float64), zi=z[comp])
            iq[comp::2] = x[comp::2] - a.astype(np.float32)
        ob.process_roots(roots, iq)
        _check_tolerance(rx, nodes, topo, ("u8-blocked-dc", f))
        ref = x.copy()
        ob.dc_correct(ref, state)
        ob.process_roots(rroots, ref)
        for i in (0, 1):
            want = rnodes[i].stream()
            worst = max(worst, float(np.abs(rx.stream(i) - want).max() / np.abs(want).max()))
    kt = rx.kernel_times()
    rx.close()
    assert 1e-6 < worst < 2e-4, worst  # the reference's own wander, relative to the signal
    assert kt["k_ingest"]["ms"] / kt["k_ingest"]["launches"] < 0.1, kt["k_ingest"]


# ------------------------------------------------------------------------------ reference-style API
def test_reference_named_interface(Receiver):
    """The same chain driven through the vfo / sdrj mirror classes (vfo.h:16-49, sdrj.h)."""
    from sdrreceiver_amd.receiver import sdrj, vfo
    main = vfo()
    main.setFs(1536000); main.setDecimationCount(2); main.setMixerFreq(484000); main.setDemodUSB(False)
    main.setCompressonStyle(1); main.init(384000, False)
    sub = vfo()
    sub.setZmqTopic("VFO01"); sub.setDecimationCount(5); sub.setFilterBandwidth(4000); sub.setGain(5 / 100)
    sub.setMixerFreq(110854); sub.setFs(384000); sub.setCompressonStyle(1); sub.init(96000, True, 0)
    main.setVFOs([sub])
    radio = sdrj()
    radio.setVFOs([main])
    g = golden("config1.npz")
    lcg = synth.Lcg(1)
    for f in range(2):
        iq = synth.lcg_frame(384000, lcg)
        radio.demodData(iq, iq.size)
        assert np.array_equal(sub.transmit_usb, g[f"f{f}_v1_pay"])
        assert radio.published[0][:2] == (b"VFO01", 12000)


# ------------------------------------------------------------------------------ errors
def test_fft_taps_of_the_mirror_classes(Receiver):
    """vfo::fftVFOSlot / fftData (vfo.cpp:290-293,492-509) and sdrj::fftVFOSlot / fftData
    (sdrj.cpp:84-101,296-303) through the Python mirror, float and byte input."""
    from sdrreceiver_amd.receiver import sdrj, vfo
    topo = tp.config1()
    for use_bytes in (False, True):
        nodes = []
        for d in topo.vfos:
            v = vfo()
            v.setFs(d.fs); v.setDecimationCount(d.decimate_count); v.setMixerFreq(d.mixer_freq)
            v.setFilterBandwidth(d.filter_bw); v.setGain(d.gain); v.setDemodUSB(d.demod_usb)
            v.setCompressonStyle(d.cstyle); v.setScaleComp(d.scalecomp); v.setZmqTopic(d.topic)
            v.init(d.samples_per_buffer, False, d.late_decimate)
            nodes.append(v)
        nodes[0].setVFOs([nodes[1]])
        radio = sdrj()
        radio.setVFOs([nodes[0]])
        radio.setDCCorrection(True)
        got_vfo, got_raw = [], []
        nodes[1].fftData = lambda a: got_vfo.append(a)
        nodes[0].fftData = lambda a: got_vfo.append("main must stay silent")
        radio.fftData = lambda a: got_raw.append(a)
        for n in nodes:
            n.fftVFOSlot(topo.vfos[1].topic)
        radio.fftVFOSlot("Main")
        onodes, oroots = ob.build_tree("port", topo)
        lcg = synth.Lcg(3)
        state = np.zeros(2, np.float32)
        raws = []
        for f in range(9):
            if use_bytes:
                b = synth.lcg_frame_u8(topo.frame, lcg)
                radio.demodBytes(b)
                iq = ob.u8_to_float(b)
                ob.dc_correct(iq, state)
            else:
                iq = synth.lcg_frame(topo.frame, lcg)
                radio.demodData(iq, iq.size)
                ob.dc_correct(iq, state)
            ob.process_roots(oroots, iq)
            raws.append(iq.view(np.complex64).copy())
            assert len(got_vfo) == f + 1 and np.array_equal(bits(got_vfo[-1]), bits(onodes[1].stream()))
        assert len(got_raw) == 2  # calls 5 and 9
        assert np.array_equal(bits(got_raw[0]), bits(raws[4])) and np.array_equal(bits(got_raw[1]), bits(raws[8]))
        radio.rx.close()


def test_error_behaviour(Receiver):
    from sdrreceiver_amd.receiver import SdrxError
    t = tp.config1()
    t.vfos[1].filter_bw = 7000  # > 12000/2: the reference throws std::out_of_range in vfo::init
    with pytest.raises(SdrxError) as e:
        Receiver.from_topology(t)
    assert e.value.code == -3
    rx = Receiver.from_topology(tp.config1())
    with pytest.raises(SdrxError) as e:
        rx.process(np.zeros(2 * 1000, np.float32))  # wrong frame length
    assert e.value.code == -1
    rx.close()
    rx = Receiver()
    rx.add_vfo(tp.config1().vfos[0])
    with pytest.raises(SdrxError) as e:
        rx.process(np.zeros(8, np.float32))  # before finalize
    assert e.value.code == -2
    rx.close()


# ------------------------------------------------------------------------------ byte ingest
@pytest.mark.parametrize("correct_dc", [False, True])
def test_u8_ingest_and_dc_bias_on_device(Receiver, correct_dc):
    """sdrx_process_u8: the b-127 LUT (jonti/sdr.cpp:43-49) and the DC-bias IIR of
    sdrj::demodData (sdrj.cpp:271-286, state persisting across frames) done on the device, against
    the oracle's host-side restatement feeding the same chain.  Bit-exact."""
    topo = tp.config2()
    rx = Receiver.from_topology(topo, exact=True)
    nodes, roots = ob.build_tree("port", topo)
    rng = np.random.default_rng(3)
    state = np.zeros(2, np.float32)
    for f in range(3):
        b = rng.integers(0, 256, 2 * topo.frame, dtype=np.uint8)
        b[0::2] = np.clip(b[0::2].astype(int) // 8 + 130, 0, 255)  # a DC offset worth removing
        rx.process_u8(b, correct_dc=correct_dc)
        iq = ob.u8_to_float(b)
        if correct_dc:
            ob.dc_correct(iq, state)
        ob.process_roots(roots, iq)
        _check_exact(rx, nodes, topo, ("u8", correct_dc, f))
    rx.close()


@pytest.mark.parametrize("speculative", [True, False])
def test_dc_bias_removal_on_frames_of_every_shape(Receiver, speculative):
    """Both evaluations of the exact recurrence (option dc_speculative: verified 1024-sample blocks / every sample in turn).
    The sequential one walks the frame in groups of 32 samples with its scalar prefetch two groups ahead and a
    line prefetch 16 groups ahead of that: frames of 16 m samples, m odd and even, from 1 024 to ~70 000 (a last group of 16
    samples; frames shorter than the prefetch distance), the accumulator carried over five frames: bit-exact against the
    oracle's sequential restatement of sdrj.cpp:277-283."""
    cases = [1024, 1040, 1280 + 16, 2048, 4096 + 16 * 17, 8704, 16384 - 16, 20480 + 16, 69632 + 16 * 21]
    rng = np.random.default_rng(11)
    cases += [16 * int(m) for m in rng.integers(64, 4000, max(6, N_SEEDS // 10))]
    for n in cases:
        if 0 < n % 1024 < 256:
            n += 256
        t = tp.Topology(fs=4 * n, frame=n, name=f"dc{n}")
        t.vfos.append(tp.VfoDesc(topic="DC0", parent=-1, fs=4 * n, decimate_count=1, mixer_freq=float(n // 3), filter_bw=0,
                                 gain=tp._g(0.05), cstyle=1, samples_per_buffer=n))
        t.vfos.append(tp.VfoDesc(topic="IQ0", parent=-1, fs=4 * n, decimate_count=0, mixer_freq=-float(n // 5), demod_usb=False,
                                 cstyle=0, samples_per_buffer=n))
        rx = Receiver.from_topology(t, exact=True, dc_speculative=speculative)
        nodes, roots = ob.build_tree("port", t)
        state = np.zeros(2, np.float32)
        for f in range(5):
            b = rng.integers(0, 256, 2 * n, dtype=np.uint8)
            b[1::2] = np.clip(b[1::2].astype(int) // 4 + 60, 0, 255)
            rx.process_u8(b, correct_dc=True)
            iq = ob.u8_to_float(b)
            ob.dc_correct(iq, state)
            ob.process_roots(roots, iq)
            assert np.array_equal(bits(rx.raw()), bits(iq.view(np.complex64))), (n, f, "raw frame after the DC-bias removal")
            _check_exact(rx, nodes, t, ("dc", n, f))
        rx.close()


DC_STREAMS = {  # (offset I, offset Q, noise sigma) in LSB
    "offsets of the capture-like stream": (1.3, -0.7, 7.0),
    "next to a binade boundary and a rounding threshold": (0.25, 4.94, 7.0),
    "large, opposite signs (one estimate pinned to its threshold)": (100.0, -120.0, 10.0),
    "an offset ten times the noise (pinned)": (30.0, -2.0, 3.0),
    "no offset at all (the estimate wanders through zero)": (0.0, 0.02, 7.0),
    "quiet front end": (1.3, -0.7, 2.0),
    "strong carriers": (1.3, -0.7, 50.0),
}


@pytest.mark.parametrize("stream", sorted(DC_STREAMS))
@pytest.mark.parametrize("per_step", [1, 4, 8])
def test_speculative_dc_chain_is_the_sequential_recurrence(Receiver, stream, per_step):
    """k_dc_chain_spec evaluates avept = fl(fl(avept * (1 - 1e-6)) + fl(1e-6 * curr)) (sdrj.cpp:277-283) a block of 1 024 samples
    at a time as an integer recurrence of the mantissa (every lane its 16 samples from a speculated start value, repeated
    until the start values stand), verifies the block and falls back to the rounded operations where it does not converge
    or the verification fails.  14 frames of 384 000 samples from the zero start state (the estimate climbs through ~20
    binades: blocks that fall back and blocks that do not alternate; after ~3 s it has reached the threshold of the
    rounding it then hovers around), offsets next to a binade boundary, both signs, an estimate that crosses zero, estimates
    pinned to their threshold by steps of an ulp: the DC-corrected frame bit for bit the oracle's, and identical to the
    every-sample evaluation (dc_speculative=0).  The last frames of the dongle-like streams run entirely in verified
    blocks.  `per_step` blocks side by side (option dc_blocks_per_step: one wave each, totals exchanged through LDS; a step
    that does not verify as a whole is taken again block by block): the same bits for every value."""
    di, dq, sigma = DC_STREAMS[stream]
    n = 384000
    t = tp.Topology(fs=1536000, frame=n, name="dcspec")
    t.vfos.append(tp.VfoDesc(topic="M", parent=-1, fs=1536000, decimate_count=3, mixer_freq=-496000.0, demod_usb=False, cstyle=1,
                             samples_per_buffer=n))
    rx = Receiver.from_topology(t, exact=True, dc_blocks_per_step=per_step)
    rx0 = Receiver.from_topology(t, exact=True, dc_speculative=False) if per_step == 4 else None
    rng = np.random.default_rng(100 + sorted(DC_STREAMS).index(stream))
    state = np.zeros(2, np.float32)
    prev = (0, 0, 0)
    for f in range(14):
        z = rng.standard_normal(2 * n) * sigma
        z[0::2] += di
        z[1::2] += dq
        b = np.clip(np.rint(z) + 127, 0, 255).astype(np.uint8)
        rx.process_u8(b, correct_dc=True)
        iq = ob.u8_to_float(b)
        ob.dc_correct(iq, state)
        assert np.array_equal(bits(rx.raw()), bits(iq.view(np.complex64))), (stream, f, "vs the oracle")
        if rx0 is not None:
            rx0.process_u8(b, correct_dc=True)
            if f % 4 == 3:
                assert np.array_equal(bits(rx0.raw()), bits(iq.view(np.complex64))), (stream, f, "every-sample evaluation vs the oracle")
        st = rx.stats()
        blocks, fb, again = st["dc_blocks"] - prev[0], st["dc_fallback_blocks"] - prev[1], st["dc_retried_blocks"] - prev[2]
        prev = (st["dc_blocks"], st["dc_fallback_blocks"], st["dc_retried_blocks"])
        assert blocks == 2 * 375
        assert again == 0 or per_step > 1
        if f == 13:
            print(f"{stream}, {per_step} blocks per step: frame 13 redid {fb} of {blocks} blocks sequentially, took {again} again on their own; estimates {state}")
            if stream in ("offsets of the capture-like stream", "quiet front end", "strong carriers"):
                assert fb <= blocks // 10, (stream, fb)
    rx.close()
    if rx0 is not None:
        rx0.close()


def test_dc_soak_of_random_regimes(Receiver):
    """tests/dc_soak.py: random offsets (0 ... +-120 LSB), noise (sigma 0 ... 60 LSB; 0 = constant bytes), frame lengths with
    partial last steps, 1 / 2 / 4 / 8 blocks per step, frames from the zero state with an offset step half way -- every
    DC-corrected frame bit for bit the oracle's.  A dozen regimes here; 60 were run on the round's final build
    (profiles/r05_experiments/dc_soak.txt)."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), "dc_soak.py"), "12", "3"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "soak: passed" in r.stdout


def test_dc_blocks_per_step_takes_powers_of_two_up_to_eight(Receiver):
    """Option dc_blocks_per_step = waves of the recurrence's workgroup: 1, 2, 4 or 8; anything else is SDRX_EINVAL and names
    the choices; after sdrx_finalize it is SDRX_ESTATE like every option."""
    from sdrreceiver_amd.receiver import SdrxError
    for bad in (0, 3, 16, -1):
        with pytest.raises(SdrxError) as e:
            Receiver(dc_blocks_per_step=bad)
        assert e.value.code == -1 and "1, 2, 4 or 8" in str(e.value), (bad, str(e.value))
    for good in (1, 2, 4, 8):
        Receiver(dc_blocks_per_step=good).close()


@pytest.mark.parametrize("per_step", [1, 2, 4])
def test_speculative_dc_chain_on_constant_and_extreme_bytes(Receiver, per_step):
    """All-255, all-0, all-127 and alternating 0 / 255 bytes, frames that are not a whole number of blocks (480 000 = 468.75
    blocks, 57 600): the accumulator runs up to +-128 through every binade on the way, `p` is one value (every block either
    verifies or is an exact tie), the last block is partial."""
    for n in (480000, 57600):
        t = tp.Topology(fs=4 * n, frame=n, name=f"dcx{n}")
        t.vfos.append(tp.VfoDesc(topic="M", parent=-1, fs=4 * n, decimate_count=2, mixer_freq=float(n // 7), demod_usb=False, cstyle=1,
                                 samples_per_buffer=n))
        rx = Receiver.from_topology(t, exact=True, dc_blocks_per_step=per_step)
        state = np.zeros(2, np.float32)
        frames = []
        for val in (255, 0, 127):
            frames += [np.full(2 * n, val, np.uint8)] * 2
        alt = np.zeros(2 * n, np.uint8)
        alt[0::4] = 255
        alt[1::4] = 255
        frames += [alt, alt]
        for f, b in enumerate(frames):
            rx.process_u8(b, correct_dc=True)
            iq = ob.u8_to_float(b)
            ob.dc_correct(iq, state)
            assert np.array_equal(bits(rx.raw()), bits(iq.view(np.complex64))), (n, f)
        rx.close()


# ------------------------------------------------------------------------------ edge geometry
def _edge_topology():
    """Every decimation depth 0..8 (vfo.h:63 allows 8 stages) as parent-less USB leaves on a short
    frame whose last chunk is partial (8704 = 8*1024 + 512), with and without the audio low-pass:
    n_out runs from 8704 down to 34 samples per frame -- shorter than the 124-sample Hilbert
    history, so the previous-frame history spans several frames."""
    t = tp.Topology(fs=34816, frame=8704, name="edge")
    for d in range(9):
        rate = 34816 // (1 << d)
        t.vfos.append(tp.VfoDesc(topic=f"D{d}", parent=-1, fs=34816, decimate_count=d, mixer_freq=float(1000 + 377 * d),
                                 filter_bw=(rate // 5) if d % 2 else 0, gain=tp._g(0.05), cstyle=1, samples_per_buffer=8704))
    t.vfos.append(tp.VfoDesc(topic="IQ", parent=-1, fs=34816, decimate_count=7, mixer_freq=-5000.0, demod_usb=False,
                             cstyle=0, samples_per_buffer=8704))
    return t


@pytest.mark.parametrize("segments", [0, 1, 2])
def test_all_decimation_depths_and_tiny_frames(Receiver, segments):
    topo = _edge_topology()
    rx = Receiver.from_topology(topo, exact=True, segments=segments)
    nodes, roots = ob.build_tree("port", topo)
    for f, iq in _frames(topo, 6, seed=21, tones=[(3000.0, 30.0), (-7000.0, 12.0)]):
        rx.process(iq)
        ob.process_roots(roots, iq)
        _check_exact(rx, nodes, topo, ("edge", segments, f))
    rx.close()


def _random_topology(rng):
    return random_topology(rng)


N_SEEDS = int(__import__("os").environ.get("SDRX_TEST_SEEDS", "60"))  # (a one-off soak run: SDRX_TEST_SEEDS=600)


def test_random_trees_against_the_oracle(Receiver):
    """60 seeded random trees (shapes, depths, rates, frame lengths, filters, late decimation,
    compress styles), 3 frames each with tones: every stream and payload bit-identical to the
    oracle.  A tree the library refuses at sdrx_finalize must fall under a documented restriction
    (here: a node rate below 1024 Hz, or a frame whose last 1024-sample chunk holds fewer than 256 samples -- one seed in
    600); at least 55 of 60 must run."""
    from sdrreceiver_amd.receiver import SdrxError
    ran = 0
    for seed in range(N_SEEDS):
        rng = np.random.default_rng(1000 + seed)
        topo = _random_topology(rng)
        try:
            rx = Receiver.from_topology(topo, exact=True, segments=seed % 5, fuse=seed % 3 != 0,  # segments 0 = the library's own choice
                                        keep_streams=seed % 2 == 0, fuse_late=seed % 7 != 0, fuse_demod=seed % 4 >= 2)
        except SdrxError as e:
            assert "fs >= 1024" in str(e) or "last chunk shorter than 256" in str(e), (seed, str(e))
            continue
        nodes, roots = ob.build_tree("port", topo)
        for f, iq in _frames(topo, 3, seed=seed, tones=[(topo.fs / 7.3, 20.0), (-topo.fs / 3.1, 9.0)]):
            rx.process(iq)
            ob.process_roots(roots, iq)
            _check_exact(rx, nodes, topo, ("random", seed, f))
        rx.close()
        ran += 1
    assert ran >= N_SEEDS * 55 // 60, ran


@pytest.mark.parametrize("mode", ["one-launch, pipelined", "one-launch per part", "separate kernels"])
def test_frame_pipeline_on_random_trees(Receiver, mode):
    """The software pipeline of k_mix_levels (level l of frame k - l in one launch, the leaf tail of the
    frame that left the last level behind it) on the 60 random trees: 6 frames queued back to back with sdrx_process_device, for odd
    seeds with an sdrx_sync in the middle (the pipeline drains and fills again); streams and payloads of
    the last frame -- which depend on every earlier frame through the filter histories -- bit-identical
    to the oracle's, and a fetch in the middle serves the frame it should."""
    import torch
    from sdrreceiver_amd.receiver import SdrxError
    ran = 0
    for seed in range(N_SEEDS):
        rng = np.random.default_rng(1000 + seed)
        topo = _random_topology(rng)
        try:
            rx = Receiver.from_topology(topo, exact=True, segments=seed % 3, fuse_demod=seed % 2 == 1, **LAUNCH_MODES[mode])
        except SdrxError as e:
            assert "fs >= 1024" in str(e) or "last chunk shorter than 256" in str(e), (seed, str(e))
            continue
        nodes, roots = ob.build_tree("port", topo)
        frames = [iq for _, iq in _frames(topo, 6, seed=seed, tones=[(topo.fs / 7.3, 20.0)])]
        dev = [torch.from_numpy(iq).cuda() for iq in frames]
        torch.cuda.synchronize()
        for f, d in enumerate(dev):
            rx.process_device(d.data_ptr(), topo.frame)
            ob.process_roots(roots, frames[f])
            if f == 2 and seed % 2:
                rx.sync()
            if f == 3 and seed % 4 == 0:
                rx.fetch()
                _check_exact(rx, nodes, topo, ("pipeline-mid", mode, seed))
        rx.fetch()
        _check_exact(rx, nodes, topo, ("pipeline", mode, seed))
        rx.close()
        ran += 1
    assert ran >= N_SEEDS * 55 // 60, ran


def test_fast_mode_on_random_trees(Receiver):
    """The A/B arithmetic (exact=0: FMAs) on the random trees, three frames with tones: every stream, every pre-quantisation
    float within 1e-5 of max|ref|, int16 within one LSB -- the north-star tolerance, on trees nobody tuned it for."""
    from sdrreceiver_amd.receiver import SdrxError
    ran = 0
    for seed in range(max(20, N_SEEDS // 2)):
        topo = _random_topology(np.random.default_rng(1000 + seed))
        try:
            rx = Receiver.from_topology(topo, exact=(False, 2)[seed % 2], keep_prequant=True, keep_streams=True, segments=seed % 3, fuse_demod=seed % 4 >= 2)
        except SdrxError as e:
            assert "fs >= 1024" in str(e) or "last chunk shorter than 256" in str(e), (seed, str(e))
            continue
        nodes, roots = ob.build_tree("port", topo)
        for f, iq in _frames(topo, 3, seed=seed, tones=[(topo.fs / 7.3, 20.0), (-topo.fs / 3.1, 9.0)]):
            rx.process(iq)
            ob.process_roots(roots, iq)
            _check_tolerance(rx, nodes, topo, ("fast random", seed, f))
        rx.close()
        ran += 1
    assert ran >= max(20, N_SEEDS // 2) * 5 // 6, ran


def test_random_api_sequences_on_random_trees(Receiver):
    """The per-frame entry points mixed at random on the random trees: sdrx_process, sdrx_process_u8, the pipelined pair
    sdrx_submit / sdrx_wait (one or two frames in flight), sdrx_process_device queued 1-4 deep with one sdrx_fetch, an
    sdrx_sync or a stream read-back in between, the spectrum tap moved around.  Whatever the order, every DELIVERED frame's
    payloads (and the streams, where the call leaves them readable) are the oracle's for that frame: the state that
    connects the calls -- frame parity of every ping-pong buffer, the frame pipeline's in-flight levels, the filter
    histories -- is what this exercises."""
    import torch
    from sdrreceiver_amd.receiver import SdrxError
    ran = 0
    for seed in range(N_SEEDS):
        rng = np.random.default_rng(5000 + seed)
        topo = _random_topology(np.random.default_rng(1000 + seed))
        try:
            rx = Receiver.from_topology(topo, exact=True, segments=int(rng.choice([0, 0, 2, 3])), keep_streams=bool(seed % 3 == 0), fuse_demod=seed % 2 == 1)
        except SdrxError as e:
            assert "fs >= 1024" in str(e) or "last chunk shorter than 256" in str(e), (seed, str(e))
            continue
        nodes, roots = ob.build_tree("port", topo)
        leaves = topo.leaves_in_publish_order()
        lcg = synth.Lcg(100 + seed)
        fno = 0
        keepalive = []

        def next_frame(as_bytes=False):
            nonlocal fno
            iq = synth.lcg_frame(topo.frame, lcg)
            if as_bytes:
                b = np.clip(np.rint(iq) + 127, 0, 255).astype(np.uint8)
                iq = ob.u8_to_float(b)
            else:
                b = None
                iq = iq + synth.tone_frame(topo.frame, topo.fs, [(topo.fs / 7.3, 20.0)], fno * topo.frame)
            ob.process_roots(roots, iq)
            fno += 1
            want = [(nodes[i].usb() if topo.vfos[i].demod_usb else nodes[i].iq()).copy() for i in leaves]
            return iq, b, want

        def check(want, ctx, streams):
            for k, i in enumerate(leaves):
                assert np.array_equal(rx.output(i), want[k]), (seed, ctx, i, "payload")
            if streams:  # (the oracle's streams are those of the LAST frame fed: only right after a synchronous delivery of it)
                for i in range(len(topo.vfos)):
                    got = rx.stream(i, missing_ok=True)
                    assert got is None or np.array_equal(bits(got), bits(nodes[i].stream())), (seed, ctx, i, "stream")

        for step in range(int(rng.integers(5, 10))):
            kind = rng.choice(["process", "u8", "pipelined", "device", "tap"], p=[0.25, 0.15, 0.25, 0.3, 0.05])
            ctx = (step, str(kind))
            if kind == "process":
                iq, _, want = next_frame()
                rx.process(iq)
                check(want, ctx, True)
            elif kind == "u8":
                iq, b, want = next_frame(as_bytes=True)
                rx.process_u8(b)
                check(want, ctx, True)
            elif kind == "pipelined":
                k = int(rng.integers(2, 6))
                wants = []
                iq, _, w = next_frame()
                rx.submit(iq)
                wants.append(w)
                for _ in range(k - 1):
                    iq, _, w = next_frame()
                    rx.submit(iq)
                    wants.append(w)
                    if rng.random() < 0.7 or rx.in_flight() == 2:
                        rx.wait()
                        check(wants.pop(0), ctx, False)
                while wants:
                    rx.wait()
                    check(wants.pop(0), ctx, not wants)
            elif kind == "device":
                k = int(rng.integers(1, 5))
                for q in range(k):
                    iq, _, want = next_frame()
                    d = torch.from_numpy(iq).cuda()
                    torch.cuda.synchronize()
                    keepalive.append(d)
                    rx.process_device(d.data_ptr(), topo.frame)
                    if rng.random() < 0.25:
                        rx.sync()
                rx.fetch()
                check(want, ctx, True)
            else:
                rx.set_tap(int(rng.integers(-1, len(topo.vfos))))
        rx.close()
        ran += 1
    assert ran >= N_SEEDS * 55 // 60, ran


def test_random_api_sequences_on_a_device_list():
    """The same on sdrx_group_*: the random trees sharded over 2-5 members (on the one GPU of the test box), fed through
    sdrx_group_process, _process_u8 (with the DC-bias removal for every second tree: each member runs the recurrence itself),
    the pipelined _submit / _wait pair and _process_device + _sync in random order.  What the publish callback delivers --
    topic, rate, payload, in the reference's order over the WHOLE tree -- is the oracle's for the delivered frame."""
    import torch
    from sdrreceiver_amd.receiver import Group, SdrxError
    ran = 0
    for seed in range(max(20, N_SEEDS // 2)):
        rng = np.random.default_rng(9000 + seed)
        topo = _random_topology(np.random.default_rng(1000 + seed))
        members = int(rng.integers(2, 6))
        try:
            g = Group.from_topology(topo, [0] * members)
        except SdrxError as e:
            assert "fs >= 1024" in str(e) or "last chunk shorter than 256" in str(e), (seed, str(e))
            continue
        nodes, roots = ob.build_tree("port", topo)
        order = topo.leaves_in_publish_order()
        lcg = synth.Lcg(300 + seed)
        dc = bool(seed % 2)
        state = np.zeros(2, np.float32)
        keepalive = []

        def frame(as_bytes):
            iq = synth.lcg_frame(topo.frame, lcg)
            b = None
            if as_bytes:
                b = np.clip(np.rint(iq) + 130, 0, 255).astype(np.uint8)
                iq = ob.u8_to_float(b)
                if dc:
                    ob.dc_correct(iq, state)
            ob.process_roots(roots, iq)
            want = [(topo.vfos[i].topic.encode()[:5].ljust(5, b"\0"), topo.vfos[i].output_rate,
                     (nodes[i].usb() if topo.vfos[i].demod_usb else nodes[i].iq()).tobytes()) for i in order
                    if topo.vfos[i].demod_usb or topo.vfos[i].topic]
            return iq, b, want

        for step in range(int(rng.integers(4, 8))):
            kind = rng.choice(["process", "u8", "pipelined", "pipelined_u8", "device"])
            ctx = (seed, members, step, str(kind))
            if kind == "process":
                iq, _, want = frame(False)
                g.process(iq)
                assert g.published == want, ctx
            elif kind == "u8":
                _, b, want = frame(True)
                g.process_u8(b, correct_dc=dc)
                assert g.published == want, ctx
            elif kind in ("pipelined", "pipelined_u8"):
                wants = []
                for q in range(int(rng.integers(2, 5))):
                    iq, b, w = frame(kind == "pipelined_u8")
                    if b is None:
                        g.submit(iq)
                    else:
                        g.submit_u8(b, correct_dc=dc)
                    wants.append(w)
                    if g.in_flight() == 2 or rng.random() < 0.5:
                        g.wait()
                        assert g.published == wants.pop(0), ctx
                while wants:
                    g.wait()
                    assert g.published == wants.pop(0), ctx
            else:
                for q in range(int(rng.integers(1, 4))):
                    iq, _, want = frame(False)
                    d = torch.from_numpy(iq).cuda()
                    torch.cuda.synchronize()
                    keepalive.append(d)
                    g.process_device(d.data_ptr(), topo.frame)
                g.sync()
                for i in order:
                    ref = nodes[i].usb() if topo.vfos[i].demod_usb else nodes[i].iq()
                    assert np.array_equal(g.output(i), ref), (ctx, i)
        g.close()
        ran += 1
    assert ran >= max(20, N_SEEDS // 2) * 5 // 6, ran


def test_fused_late_decimation_random_geometries(Receiver):
    """The /5 and /6 walks on frames of 960 .. 11 520 samples that are NOT multiples of their 960 / 1008-sample chunks
    (partial last chunks of every length that is a multiple of 16 L), below mains of depth 0-2, with and without the audio
    low-pass, 0-5 forced segments, four frames (the 60 / 90-sample history crosses three frame boundaries, the NCO tables
    wrap): payloads, and decimate[0] where it is kept, bit-identical to the oracle."""
    n_cases = max(40, N_SEEDS * 40 // 60)
    ran = 0
    for case in range(n_cases):
        rng = np.random.default_rng(7000 + case)
        L = int(rng.choice([5, 6]))
        n = 16 * L * int(rng.integers(12 if L == 5 else 11, 121))
        if 0 < n % 1024 < 256:
            n += 16 * L * 4  # (a frame's last 1024-sample chunk must hold >= 256 samples: DESIGN.md section 8)
            if 0 < n % 1024 < 256:
                continue
        dp = int(rng.integers(0, 3))
        root_n = n << dp
        if 0 < root_n % 1024 < 256:
            continue
        fs_root = root_n * int(rng.choice([1, 2, 4]))
        t = tp.Topology(fs=fs_root, frame=root_n, name=f"late{case}")
        t.vfos.append(tp.VfoDesc(parent=-1, fs=fs_root, decimate_count=dp, mixer_freq=float(int(rng.integers(-fs_root // 2 + 1, fs_root // 2))),
                                 demod_usb=False, cstyle=1, samples_per_buffer=root_n))
        fs = fs_root >> dp
        for k in range(int(rng.integers(1, 4))):
            rate = fs // L
            bw = int(rate / rng.uniform(2.3, 10.0)) if rng.random() < 0.5 else 0
            t.vfos.append(tp.VfoDesc(topic=f"F{case % 100:02d}{k}", parent=0, fs=fs, decimate_count=0,
                                     mixer_freq=float(int(rng.integers(-fs // 2 + 1, fs // 2))), late_decimate=L, filter_bw=bw,
                                     gain=tp._g(float(rng.uniform(0.01, 0.08))), cstyle=1, samples_per_buffer=n))
        keep = bool(case % 2)
        rx = Receiver.from_topology(t, exact=True, segments=int(rng.choice([0, 1, 2, 3, 5])), keep_streams=keep)
        nodes, roots = ob.build_tree("port", t)
        for f, iq in _frames(t, 4, seed=case, tones=[(t.fs / 5.7, 15.0)]):
            rx.process(iq)
            ob.process_roots(roots, iq)
            _check_exact(rx, nodes, t, ("late geometry", case, L, n, dp, f))
            if not keep:
                assert rx.stream(1, missing_ok=True) is None  # (the fused path it is: decimate[0] of the leaves is not kept)
        rx.close()
        ran += 1
    assert ran >= n_cases * 3 // 4, ran


def test_shallow_leaf_beside_a_deep_tree_in_the_frame_pipeline(Receiver):
    """Found by a 600-seed soak run (seed 213): in the one-launch frame pipeline frame k gets its leaf tail behind launch
    k + n_levels - 1, and the streams are double buffered -- a parent-less leaf beside a THREE-level tree was overwritten by
    frame k + 2 before its demodulation had run (its 166-sample history spans three 64-output frames here, so every later
    frame was wrong).  Such a tree runs one launch per level now; the reference's two-level trees are not affected."""
    import torch
    rng = np.random.default_rng(1000 + 213)
    topo = _random_topology(rng)
    levels = {}
    for i, v in enumerate(topo.vfos):
        levels[i] = 0 if v.parent < 0 else levels[v.parent] + 1
    assert max(levels.values()) == 2 and any(levels[i] == 0 and not topo.children(i) for i in levels)
    rx = Receiver.from_topology(topo, exact=True)
    nodes, roots = ob.build_tree("port", topo)
    frames = [iq for _, iq in _frames(topo, 8, seed=213, tones=[(topo.fs / 7.3, 20.0)])]
    dev = [torch.from_numpy(iq).cuda() for iq in frames]
    torch.cuda.synchronize()
    for f, d in enumerate(dev):
        rx.process_device(d.data_ptr(), topo.frame)
        ob.process_roots(roots, frames[f])
    rx.fetch()
    _check_exact(rx, nodes, topo, "shallow leaf")
    rx.close()


def test_long_run_wraps_the_nco_tables_many_times(Receiver):
    """40 frames = 10 s of signal through config 1: every NCO table (1 s long) wraps ten times, the
    frame parity of every ping-pong buffer flips forty times; still bit-identical at the end and at
    every fifth frame on the way."""
    topo = tp.config1()
    rx = Receiver.from_topology(topo, exact=True)
    nodes, roots = ob.build_tree("port", topo)
    for f, iq in _frames(topo, 40, seed=31, tones=[(485000.0, 40.0)]):
        rx.process(iq)
        ob.process_roots(roots, iq)
        if f % 5 == 4:
            _check_exact(rx, nodes, topo, ("long", f))
    rx.close()


def test_tolerance_arithmetic_does_not_drift_over_a_long_run(Receiver):
    """The tolerance arithmetic rebuilds every run of 16 NCO table entries from an EXACT checkpoint, so its error cannot
    grow with time: 72 frames = 18 s of signal through the sdr_25E tree (the 1.536 M tables wrap 18 times, the 384 k /
    192 k ones 18 times, start-up entries replayed at every wrap), checked every 8th frame -- streams and pre-quantisation
    floats within 1e-5 of max|ref|, int16 within 1 LSB -- and the error at the end no larger than 4x the error after
    the first second."""
    topo = golden_topology("profile_25e")
    rx = Receiver.from_topology(topo, exact=False, keep_prequant=True)
    nodes, roots = ob.build_tree("port", topo)
    seen = []
    for f, iq in _frames(topo, 72, seed=77, tones=[(485000.0, 40.0), (-520000.0, 15.0)]):
        rx.process(iq)
        ob.process_roots(roots, iq)
        if f % 8 == 7:
            _check_tolerance(rx, nodes, topo, ("no drift", f))
            e = 0.0
            for i, v in enumerate(topo.vfos):
                ref = nodes[i].stream()
                e = max(e, float(np.abs(rx.stream(i) - ref).max()) / float(np.abs(ref).max()))
            seen.append(e)
    rx.close()
    assert seen[-1] <= 4 * max(seen[0], 1e-7), seen


def test_long_queue_of_frames_through_the_pipeline(Receiver):
    """64 frames = 16 s of signal queued back to back with sdrx_process_device on the sdr_25E tree (the
    software pipeline of k_mix_levels stays full for the whole run, every ping-pong buffer flips 64 times,
    the 384 k / 192 k NCO tables wrap 16 times), with an sdrx_fetch every 16th frame: streams and payloads
    of the fetched frames bit-identical to the oracle."""
    import torch
    topo = golden_topology("profile_25e")
    rx = Receiver.from_topology(topo, exact=True)
    nodes, roots = ob.build_tree("port", topo)
    lcg = synth.Lcg(41)
    for f in range(64):
        iq = synth.lcg_frame(topo.frame, lcg) + synth.tone_frame(topo.frame, topo.fs, [(485000.0, 40.0), (-520000.0, 15.0)], f * topo.frame)
        d = torch.from_numpy(iq).cuda()
        torch.cuda.synchronize()
        rx.process_device(d.data_ptr(), topo.frame)
        ob.process_roots(roots, iq)
        if f % 16 == 15:
            rx.fetch()
            _check_exact(rx, nodes, topo, ("queue", f))
        else:
            rx.sync() if f % 16 == 7 else None  # (a drain in the middle of a stretch, too)
    rx.close()


def test_longest_audio_filters(Receiver):
    """The audio low-pass at the top of the supported range: 255 taps (48 k, bw 1817), 251 (bw 1850),
    155 (bw 3000, the 3 kHz case SURVEY lists) and 31 (bw 15000), odd and even history lengths, also
    behind the /5 late decimation; 4 frames so the 124 + N samples of history span a frame boundary."""
    t = tp.Topology(fs=1536000, frame=384000, name="longfir")
    t.vfos.append(tp.VfoDesc(parent=-1, fs=1536000, decimate_count=3, mixer_freq=-496000.0, demod_usb=False, cstyle=1,
                             samples_per_buffer=384000))
    for k, bw in enumerate([1817, 1850, 3000, 15000, 2000]):
        t.vfos.append(tp.VfoDesc(topic=f"F{k}", parent=0, fs=192000, decimate_count=2, mixer_freq=float(-41300 + 7000 * k),
                                 filter_bw=bw, gain=tp._g(0.05), cstyle=1, samples_per_buffer=48000))
    t.vfos.append(tp.VfoDesc(parent=-1, fs=1536000, decimate_count=0, mixer_freq=100000.0, demod_usb=False, cstyle=1,
                             samples_per_buffer=384000))
    t.vfos.append(tp.VfoDesc(topic="L5", parent=len(t.vfos) - 1, fs=1536000, decimate_count=3, mixer_freq=-30000.0, late_decimate=6,
                             filter_bw=1300, gain=tp._g(0.04), cstyle=1, samples_per_buffer=384000))
    rx = Receiver.from_topology(t, exact=True)
    nodes, roots = ob.build_tree("port", t)
    lens = [len(rx.taps(i, "fir_usb")) for i in range(1, 6)]
    assert lens == [255, 251, 155, 31, 231], lens
    for f, iq in _frames(t, 4, seed=17, tones=[(-496000.0 - 40000.0, 30.0)]):
        rx.process(iq)
        ob.process_roots(roots, iq)
        _check_exact(rx, nodes, t, ("longfir", f))
    rx.close()


@pytest.mark.parametrize("fuse", [True, False])
def test_audio_filters_longer_than_256_taps(Receiver, fuse):
    """The reference accepts any filter_bandwidth (firfilter.cpp:108-119: ntaps = 53 fs / (22 bw/4)):
    500 Hz at 48 kS/s is 925 taps.  Filters above the 256 taps k_usb_demod applies itself go through
    k_lpf_long: 309, 463 and 925 taps at 48 kS/s, 771 taps at 12 kS/s, ~3 300 taps at 12 kS/s (longer than
    the 3 000 outputs of a frame: the history spans two frames), and ~440 taps behind the /6 late
    decimation -- all bit-identical to the oracle over 6 frames, queued frames included."""
    import torch
    t = tp.Topology(fs=1536000, frame=384000, name="longlpf")
    t.vfos.append(tp.VfoDesc(parent=-1, fs=1536000, decimate_count=3, mixer_freq=-496000.0, demod_usb=False, cstyle=1,
                             samples_per_buffer=384000))
    for k, bw in enumerate([1500, 1000, 500, 10000]):
        t.vfos.append(tp.VfoDesc(topic=f"W{k}", parent=0, fs=192000, decimate_count=2, mixer_freq=float(-41300 + 7000 * k),
                                 filter_bw=bw, gain=tp._g(0.05), cstyle=1, samples_per_buffer=48000))
    t.vfos.append(tp.VfoDesc(parent=-1, fs=1536000, decimate_count=2, mixer_freq=484000.0, demod_usb=False, cstyle=1,
                             samples_per_buffer=384000))
    for k, bw in enumerate([150, 35]):
        t.vfos.append(tp.VfoDesc(topic=f"N{k}", parent=5, fs=384000, decimate_count=5, mixer_freq=float(110854 - 9000 * k),
                                 filter_bw=bw, gain=tp._g(0.05), cstyle=1, samples_per_buffer=96000))
    t.vfos.append(tp.VfoDesc(parent=-1, fs=1536000, decimate_count=0, mixer_freq=100000.0, demod_usb=False, cstyle=1,
                             samples_per_buffer=384000))
    t.vfos.append(tp.VfoDesc(topic="L6", parent=len(t.vfos) - 1, fs=1536000, decimate_count=3, mixer_freq=-30000.0, late_decimate=6,
                             filter_bw=700, gain=tp._g(0.04), cstyle=1, samples_per_buffer=384000))
    rx = Receiver.from_topology(t, exact=True, fuse=fuse, keep_prequant=True)
    nodes, roots = ob.build_tree("port", t)
    usb = [i for i, v in enumerate(t.vfos) if v.demod_usb]
    lens = [len(rx.taps(i, "fir_usb")) for i in usb]
    assert lens[:5] == [309, 463, 925, 47, 771] and lens[5] > 3000 and lens[6] > 256, lens
    assert lens == [len(nodes[i].taps("fir_usb")) for i in usb]
    for i in usb:
        assert np.array_equal(bits(rx.taps(i, "fir_usb")), bits(nodes[i].taps("fir_usb"))), i
    frames = [iq for _, iq in _frames(t, 6, seed=19, tones=[(-496000.0 - 40000.0, 30.0), (484000.0 + 110000.0, 20.0)])]
    for f in range(3):
        rx.process(frames[f])
        ob.process_roots(roots, frames[f])
        _check_exact(rx, nodes, t, ("longlpf", f))
        for i in usb:  # exact: float * 2^15
            assert np.array_equal(rx.prequant(i).astype(np.float64), nodes[i].usb_prequant()), (f, i)
    dev = [torch.from_numpy(iq).cuda() for iq in frames[3:]]
    torch.cuda.synchronize()
    for f, d in enumerate(dev):
        rx.process_device(d.data_ptr(), t.frame)
        ob.process_roots(roots, frames[3 + f])
    rx.fetch()
    _check_exact(rx, nodes, t, ("longlpf", "queued"))
    rx.close()
    if fuse:  # the FMA arithmetic of the same kernels against the north-star tolerance
        rx = Receiver.from_topology(t, exact=False, keep_prequant=True)
        nodes, roots = ob.build_tree("port", t)
        for f in range(3):
            rx.process(frames[f])
            ob.process_roots(roots, frames[f])
            _check_tolerance(rx, nodes, t, ("longlpf-fast", f))
        rx.close()


def test_contexts_do_not_leak_device_memory(Receiver):
    """30 receivers created, run for a frame and destroyed (mainwindow's stop/start cycle): device
    memory in use returns to where it was (hipFree of every arena, staging buffer, stream, event)."""
    import torch
    topo = golden_topology("profile_25e")
    iq = synth.lcg_frame(topo.frame, synth.Lcg(2))

    def cycle():
        rx = Receiver.from_topology(topo)
        rx.process(iq)
        rx.process_u8((iq + 127).astype(np.uint8), correct_dc=True)
        rx.close()

    cycle()  # one-time allocations of the runtime (code objects, scratch) happen here
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for _ in range(30):
        cycle()
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < 8 << 20, (free0, free1)  # a context of this profile holds ~30 MB: 30 leaked ones would show


def _short_chunk_cases():
    fixed = [(5056, 2), (5760, 5), (5760, 7), (5072, 3), (8704, 4)]
    rng = np.random.default_rng(77)
    sweep = [(int(64 * rng.integers(70, 260)), int(rng.integers(2, 9))) for _ in range(35)]
    return fixed + [(f, sg) for f, sg in sweep if f % 1024 == 0 or f % 1024 >= 256]


@pytest.mark.parametrize("frame,segments", _short_chunk_cases())
def test_segments_whose_walk_ends_in_a_short_chunk(Receiver, frame, segments):
    """A segment that starts inside the frame walks from `first emitted sample - warm-up`, so its
    chunks are not aligned with the frame's: the frame's last samples may end up in a chunk of
    16 ... 240 samples.  The filter history saved there for the next frame (two lanes for stage 1)
    must still be complete: depths 1-6 on frames picked so that exactly this happens, 5 frames."""
    t = tp.Topology(fs=frame * 4, frame=frame, name="shortlast")
    for d in range(1, 7):
        if frame % (16 << max(0, d - 4)) or frame % (1 << d):
            continue
        t.vfos.append(tp.VfoDesc(topic=f"S{d}", parent=-1, fs=frame * 4, decimate_count=d, mixer_freq=float(900 + 211 * d),
                                 gain=tp._g(0.05), cstyle=1, samples_per_buffer=frame))
    rx = Receiver.from_topology(t, exact=True, segments=segments)
    nodes, roots = ob.build_tree("port", t)
    for f, iq in _frames(t, 4, seed=23, tones=[(2500.0, 25.0)]):
        rx.process(iq)
        ob.process_roots(roots, iq)
        _check_exact(rx, nodes, t, ("shortlast", frame, segments, f))
    rx.close()


def test_three_level_tree(Receiver):
    """vfo::process recurses (vfo.cpp:253-264); the reference only builds two levels, the library
    takes any depth: raw -> d=2 -> d=1 -> {d=2 USB leaf with low-pass, d=0 USB leaf, d=3 IQ leaf}."""
    t = tp.Topology(fs=1536000, frame=384000, name="3level")
    t.vfos.append(tp.VfoDesc(parent=-1, fs=1536000, decimate_count=2, mixer_freq=484000.0, demod_usb=False, cstyle=1,
                             samples_per_buffer=384000))
    t.vfos.append(tp.VfoDesc(parent=0, fs=384000, decimate_count=1, mixer_freq=-50000.0, demod_usb=False, cstyle=1,
                             samples_per_buffer=96000))
    t.vfos.append(tp.VfoDesc(topic="L2A", parent=1, fs=192000, decimate_count=2, mixer_freq=21000.0, filter_bw=10000,
                             gain=tp._g(0.05), cstyle=1, samples_per_buffer=48000))
    t.vfos.append(tp.VfoDesc(topic="L2B", parent=1, fs=192000, decimate_count=0, mixer_freq=-33000.0, gain=tp._g(0.02),
                             cstyle=1, samples_per_buffer=48000))
    t.vfos.append(tp.VfoDesc(topic="L2C", parent=1, fs=192000, decimate_count=3, mixer_freq=5000.0, demod_usb=False,
                             cstyle=1, scalecomp=4, samples_per_buffer=48000))
    rx = Receiver.from_topology(t, exact=True)
    nodes, roots = ob.build_tree("port", t)
    for f, iq in _frames(t, 3, seed=8, tones=[(-380000.0, 20.0)]):
        rx.process(iq)
        ob.process_roots(roots, iq)
        _check_exact(rx, nodes, t, ("3level", f))
    assert [p[0] for p in rx.published] == [b"L2A\0\0", b"L2B\0\0", b"L2C\0\0"]
    rx.close()


def test_five_level_tree_takes_the_per_level_launches(Receiver):
    """Deeper than k_mix_levels' four levels: raw -> d=1 -> d=1 -> d=1 -> d=1 -> {USB leaf, IQ leaf}; the
    library falls back to one launch per level, queued frames included."""
    import torch
    t = tp.Topology(fs=1536000, frame=384000, name="5level")
    fs, n, parent = 1536000, 384000, -1
    for lv in range(4):
        t.vfos.append(tp.VfoDesc(parent=parent, fs=fs, decimate_count=1, mixer_freq=float(40000 - 9000 * lv), demod_usb=False, cstyle=1,
                                 samples_per_buffer=n))
        parent, fs, n = len(t.vfos) - 1, fs // 2, n // 2
    t.vfos.append(tp.VfoDesc(topic="DEEP", parent=parent, fs=fs, decimate_count=1, mixer_freq=1234.0, filter_bw=10000, gain=tp._g(0.05),
                             cstyle=1, samples_per_buffer=n))
    t.vfos.append(tp.VfoDesc(topic="DEEPQ", parent=parent, fs=fs, decimate_count=2, mixer_freq=-4321.0, demod_usb=False, cstyle=1,
                             scalecomp=2, samples_per_buffer=n))
    rx = Receiver.from_topology(t, exact=True)
    assert rx.stats()["n_levels"] == 5
    nodes, roots = ob.build_tree("port", t)
    frames = [iq for _, iq in _frames(t, 5, seed=14, tones=[(30000.0, 25.0)])]
    for f in range(2):
        rx.process(frames[f])
        ob.process_roots(roots, frames[f])
        _check_exact(rx, nodes, t, ("5level", f))
    dev = [torch.from_numpy(iq).cuda() for iq in frames[2:]]
    torch.cuda.synchronize()
    for f, d in enumerate(dev):
        rx.process_device(d.data_ptr(), t.frame)
        ob.process_roots(roots, frames[2 + f])
    rx.fetch()
    _check_exact(rx, nodes, t, ("5level", "queued"))
    rx.close()


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_random_interleaving_of_the_frame_interfaces(Receiver, seed):
    """One receiver, 48 frames, fed through a seeded random interleaving of every way a frame can enter --
    sdrx_process (synchronous), sdrx_submit / sdrx_wait (pipelined egress, one or two frames in flight),
    sdrx_process_u8, sdrx_process_device (queued on the device, software-pipelined launches) with
    sdrx_fetch / sdrx_sync / getters at random points -- the transitions between the launch paths drain
    and refill the pipelines.  Every payload that comes out (callbacks of a wait / process / fetch) equals
    the oracle's for the frame it belongs to; the filter state carried across all transitions is one stream."""
    import torch
    rng = np.random.default_rng(900 + seed)
    topo = golden_topology("54w" if seed % 2 else "profile_25e")
    rx = Receiver.from_topology(topo, exact=True)
    nodes, roots = ob.build_tree("port", topo)
    order = topo.leaves_in_publish_order()
    lcg = synth.Lcg(seed)
    want, frames = [], []
    for f in range(48):
        iq = synth.lcg_frame(topo.frame, lcg)  # integer valued: also representable as dongle bytes
        ob.process_roots(roots, iq)
        frames.append(iq)
        want.append([(nodes[i].usb() if topo.vfos[i].demod_usb else nodes[i].iq()).tobytes() for i in order
                     if topo.vfos[i].demod_usb or topo.vfos[i].topic])
    pending = []  # frames submitted and not yet waited for
    f, checked, keep = 0, 0, []

    def delivered(idx):
        nonlocal checked
        assert [p for _, _, p in rx.published] == want[idx], (seed, idx)
        checked += 1

    while f < 48:
        op = rng.integers(0, 5)
        if op == 0 and not pending:
            rx.process(frames[f])
            delivered(f)
            f += 1
        elif op == 1 and not pending:
            rx.process_u8((frames[f] + 127).astype(np.uint8))
            delivered(f)
            f += 1
        elif op == 2:
            if len(pending) == 2:
                rx.wait()
                delivered(pending.pop(0))
            rx.submit(frames[f])
            pending.append(f)
            f += 1
        elif op == 3 and pending:
            rx.wait()
            delivered(pending.pop(0))
        elif op == 4 and not pending:
            k = int(rng.integers(1, 5))
            for _ in range(min(k, 48 - f)):
                d = torch.from_numpy(frames[f]).cuda()
                torch.cuda.synchronize()
                keep.append(d)
                rx.process_device(d.data_ptr(), topo.frame)
                f += 1
            what = rng.integers(0, 3)
            if what == 0:
                rx.fetch()
                delivered(f - 1)
            elif what == 1:
                rx.sync()
                assert rx.output(order[0]).tobytes() == want[f - 1][0]
            keep = keep[-8:]
    while pending:
        rx.wait()
        delivered(pending.pop(0))
    rx.close()
    assert checked >= 12, checked


@pytest.mark.parametrize("seed", [1, 2])
def test_random_interleaving_on_a_group(seed):
    """The same idea on a group of three contexts (sdrx_group_*): synchronous and pipelined calls, float and
    byte frames in random order; every delivered frame equals the oracle's, in the reference's publish order."""
    from sdrreceiver_amd.receiver import Group
    rng = np.random.default_rng(70 + seed)
    topo = tp.config3(48)
    g = Group.from_topology(topo, [0, 0, 0])
    nodes, roots = ob.build_tree("port", topo)
    order = topo.leaves_in_publish_order()
    lcg = synth.Lcg(seed)
    frames, want = [], []
    for f in range(24):
        iq = synth.lcg_frame(topo.frame, lcg)
        ob.process_roots(roots, iq)
        frames.append(iq)
        want.append([nodes[i].usb().tobytes() for i in order])
    pending, f, checked = [], 0, 0
    while f < 24 or pending:
        op = rng.integers(0, 4)
        if op == 0 and not pending and f < 24:
            g.process(frames[f])
            assert [p for _, _, p in g.published] == want[f], (seed, f)
            f += 1
            checked += 1
        elif op in (1, 2) and f < 24:
            if len(pending) == 2:
                g.wait()
                assert [p for _, _, p in g.published] == want[pending.pop(0)]
                checked += 1
            (g.submit_u8((frames[f] + 127).astype(np.uint8)) if op == 2 else g.submit(frames[f]))
            pending.append(f)
            f += 1
        elif pending:
            g.wait()
            assert [p for _, _, p in g.published] == want[pending.pop(0)]
            checked += 1
    g.close()
    assert checked == 24


def test_out_of_range_conversions_follow_the_x86_build(Receiver):
    """`short = double` and `signed char = float` beyond the target range are undefined in C; the reference's
    x86-64 binary does cvttsd2si / cvttss2si -- INT32_MIN for anything outside int32, then the low 16 / 8
    bits -- and so does the oracle.  A frame 200 000 times louder than the 8-bit front end can deliver drives
    every leaf far out of range (audio beyond int16 AND beyond int32, IQ bytes beyond int8): the payloads
    still equal the oracle's bit for bit, as do the pre-quantisation floats."""
    t = tp.Topology(fs=1536000, frame=384000, name="loud")
    t.vfos.append(tp.VfoDesc(parent=-1, fs=1536000, decimate_count=2, mixer_freq=484000.0, demod_usb=False, cstyle=1,
                             samples_per_buffer=384000))
    t.vfos.append(tp.VfoDesc(topic="LOUD1", parent=0, fs=384000, decimate_count=5, mixer_freq=110854.0, filter_bw=4000,
                             gain=tp._g(0.05), cstyle=1, samples_per_buffer=96000))
    t.vfos.append(tp.VfoDesc(topic="LOUD2", parent=0, fs=384000, decimate_count=3, mixer_freq=-20000.0, gain=tp._g(0.8), cstyle=1,
                             samples_per_buffer=96000))
    for cs, top in ((1, "IQ4"), (0, "IQ8")):
        t.vfos.append(tp.VfoDesc(topic=top, parent=-1, fs=1536000, decimate_count=3, mixer_freq=-496000.0, demod_usb=False,
                                 cstyle=cs, scalecomp=1, samples_per_buffer=384000))
    rx = Receiver.from_topology(t, exact=True, keep_prequant=True)
    nodes, roots = ob.build_tree("port", t)
    seen_wrap = False
    for f, iq in _frames(t, 3, seed=3, tones=[(484000.0 + 111000.0, 9.0), (-496000.0 + 3000.0, 5.0)]):
        iq = (iq * np.float32(200000.0)).astype(np.float32)
        rx.process(iq)
        ob.process_roots(roots, iq)
        _check_exact(rx, nodes, t, ("loud", f))
        for i in (1, 2):
            pre = nodes[i].usb_prequant()
            assert np.array_equal(rx.prequant(i).astype(np.float64), pre), (f, i)
            seen_wrap |= bool((np.abs(pre) >= 2.0 ** 31).any()) and bool((np.abs(pre) > 40000).any())
    rx.close()
    assert seen_wrap, "the test signal must drive the audio beyond int32"


def test_two_contexts_share_one_uploaded_frame(Receiver):
    """sdrx_process_shared / sdrx_submit_shared: sdrj::demodData hands every main VFO the same `samples`
    (sdrj.cpp:288-294); a binding that keeps one context per main VFO (host/qt/vfo_adapter.cpp) uploads the frame
    through the first context and runs the second one on the uploaded copy.  The two mains of sdr_25E as two
    contexts: synchronous, pipelined (the shared buffer is per frame parity: frame f stays valid while f+1 is
    staged), from floats and from dongle bytes; every leaf bit-identical to the oracle running the whole tree."""
    from sdrreceiver_amd.receiver import SdrxError
    full = tp.profile_25e()
    nodes, roots = ob.build_tree("port", full)
    parts = []
    for r in full.roots():
        keep = [r] + full.children(r)
        remap = {g: k for k, g in enumerate(keep)}
        vf = [tp.VfoDesc(**{**full.vfos[g].__dict__, "parent": remap.get(full.vfos[g].parent, -1)}) for g in keep]
        parts.append((keep, Receiver.from_topology(tp.Topology(fs=full.fs, frame=full.frame, vfos=vf))))
    (k0, a), (k1, b) = parts
    with pytest.raises(SdrxError) as e:
        b.process_shared(a)  # nothing staged yet
    assert e.value.code == -2

    def check(rx, keep, f):
        for local, g in enumerate(keep):
            if full.vfos[g].parent >= 0:
                assert np.array_equal(rx.output(local), nodes[g].usb()), (f, full.vfos[g].topic)

    frames = [iq for _, iq in _frames(full, 7, seed=31, tones=[(-377000.0, 25.0)])]
    for f in range(2):  # synchronous
        a.process(frames[f])
        b.process_shared(a)
        ob.process_roots(roots, frames[f])
        check(a, k0, f)
        check(b, k1, f)
    # pipelined: a.submit(f+1); a.wait() -> f; b.submit_shared(a) = f+1; b.wait() -> f
    want = []
    for f in range(2, 6):
        ob.process_roots(roots, frames[f])
        want.append({g: nodes[g].usb().copy() for g in k0 + k1 if full.vfos[g].parent >= 0})
    a.submit(frames[2])
    b.submit_shared(a)
    for f in range(3, 6):
        a.submit(frames[f])
        a.wait()
        b.submit_shared(a)
        b.wait()
        for rx, keep in ((a, k0), (b, k1)):
            for local, g in enumerate(keep):
                if full.vfos[g].parent >= 0:
                    assert np.array_equal(rx.output(local), want[f - 3][g]), ("pipelined", f - 1, full.vfos[g].topic)
    a.wait()
    b.wait()
    for rx, keep in ((a, k0), (b, k1)):
        for local, g in enumerate(keep):
            if full.vfos[g].parent >= 0:
                assert np.array_equal(rx.output(local), want[3][g])
    u8 = np.clip(np.rint(frames[6]) + 127, 0, 255).astype(np.uint8)  # dongle bytes: the LUT runs in each context's level 0
    a.process_u8(u8)
    b.process_shared(a)
    ob.process_roots(roots, ob.u8_to_float(u8))
    check(a, k0, 6)
    check(b, k1, 6)
    a.close()
    b.close()


def test_a_frame_is_shared_only_if_it_is_the_same_frame(Receiver):
    """sdrx_process_if_same / sdrx_submit_if_same: the second context runs on the first one's upload only when the frame
    it is handed equals that upload byte for byte (the library compares it with the uploader's pinned staging copy);
    one changed float anywhere -- here the last one, and one in the middle -- and the call answers SDRX_DIFFERENT without
    queueing anything, the caller uploads, and either way the results are the oracle's for the frame that was HANDED."""
    full = tp.profile_25e()
    parts = []
    for r in full.roots():
        keep = [r] + full.children(r)
        remap = {g: k for k, g in enumerate(keep)}
        vf = [tp.VfoDesc(**{**full.vfos[g].__dict__, "parent": remap.get(full.vfos[g].parent, -1)}) for g in keep]
        parts.append((keep, Receiver.from_topology(tp.Topology(fs=full.fs, frame=full.frame, vfos=vf))))
    (k0, a), (k1, b) = parts
    # the oracle runs the two mains on DIFFERENT inputs where the test hands them different frames
    nodes, roots = ob.build_tree("port", full)
    frames = [iq for _, iq in _frames(full, 6, seed=37, tones=[(200000.0, 18.0)])]
    shared = []
    for f, iq in enumerate(frames):
        mine = iq.copy()
        if f % 3 == 1:
            mine[-1] += 1.0            # the very last float
        elif f % 3 == 2:
            mine[len(mine) // 2 + 1] -= 2.0
        a.process(iq)
        if f < 3:
            ok = b.process_if_same(a, mine)
            if not ok:
                b.process(mine)
        else:  # pipelined
            ok = b.submit_if_same(a, mine)
            if not ok:
                b.submit(mine)
            b.wait()
        shared.append(ok)
        ob.process_roots([roots[0]], iq)
        ob.process_roots([roots[1]], mine)
        for rx, keep in ((a, k0), (b, k1)):
            for local, g in enumerate(keep):
                if full.vfos[g].parent >= 0:
                    assert np.array_equal(rx.output(local), nodes[g].usb()), (f, full.vfos[g].topic)
    assert shared == [True, False, False, True, False, False]
    a.close()
    b.close()
