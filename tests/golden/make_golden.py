#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REAL reference build (oracle/_ref/libsdrref.so).

Run in the build container only (needs /root/reference to have been compiled by
`make -C oracle/ref`):   python tests/golden/make_golden.py

The reference ships no tests or vectors of its own, so every fixture here
is an output of the reference's own code on a documented input.  Inputs are regenerated from the
LCG of sdrreceiver_amd/synth.py (seed 1), so only outputs are stored; large outputs are stored as
a SHA-256 of their bytes plus a short head/tail so both bit-exact (oracle) and toleranced (GPU)
checks have something to hold on to.  Fixtures are data: no reference source text is stored.
"""
from __future__ import annotations

import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import binding as ob  # noqa: E402
from sdrreceiver_amd import synth, topology as tp  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
KIND = "reference"


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def gen_primitives():
    d = {}
    # NCO tables: (Fs, f) pairs of the shipped profiles + a negative and a zero frequency.
    pairs = [(384000, 110854), (192000, -82430), (240000, 32065), (1536000, 484000), (1536000, -496000),
             (1920000, -911000), (60000, 14635), (384000, 0), (288000, 24000)]
    d["nco_pairs"] = np.array(pairs, np.int64)
    for k, (fs, f) in enumerate(pairs):
        t = ob.osc_table(KIND, fs, f)
        d[f"nco{k}_head"] = t[:512].copy()
        d[f"nco{k}_tail"] = t[-64:].copy()
        d[f"nco{k}_sha"] = np.array(sha(t))
        seq = ob.osc_sequence(KIND, fs, f, 8)  # multiplier of samples 0..7 (first is table[L-1])
        d[f"nco{k}_seq"] = seq
    # low-pass designs used by vfo::init for the shipped profiles (+155-tap and 73-tap cases)
    lps = [(2, 12000, 4000, 1000), (2, 48000, 10000, 2500), (2, 48000, 15000, 3750), (2, 48000, 3000, 750),
           (2, 240000, 24000, 12000), (2, 60000, 6000, 3000), (2, 288000, 24000, 9600)]
    d["lp_args"] = np.array(lps, np.float64)
    for k, a in enumerate(lps):
        d[f"lp{k}"] = ob.low_pass(KIND, *a)
    # Hilbert taps: "Fs" is samplesOut (vfo.cpp:137)
    hs = [3000, 6000, 12000, 15000, 750]
    d["hilbert_fs"] = np.array(hs, np.int64)
    for k, fs in enumerate(hs):
        d[f"hilbert{k}"] = ob.hilbert_taps(KIND, 125, fs)
    # single half-band stage, 3 frames of 16 of the ramp 1..48 (frame-boundary rule), and
    # 4 frames of 64 LCG samples
    R = ob.load(KIND)
    hb = R.fn("halfband_new")(11, 16)
    ramp = np.arange(1, 49, dtype=np.float32)
    outs = []
    for f in range(3):
        x = np.zeros(32, np.float32)
        x[0::2] = ramp[16 * f:16 * f + 16]
        x[1::2] = -ramp[16 * f:16 * f + 16]
        y = np.zeros(16, np.float32)
        R.fn("halfband_decimate")(hb, x.ctypes.data, 16, y.ctypes.data)
        outs.append(y.copy())
    R.fn("halfband_free")(hb)
    d["hb_ramp_out"] = np.stack(outs)
    hb = R.fn("halfband_new")(11, 64)
    lcg = synth.Lcg(7)
    outs = []
    for f in range(4):
        x = synth.lcg_frame(64, lcg)
        y = np.zeros(64, np.float32)
        R.fn("halfband_decimate")(hb, x.ctypes.data, 64, y.ctypes.data)
        outs.append(y.copy())
    R.fn("halfband_free")(hb)
    d["hb_lcg_out"] = np.stack(outs)
    np.savez_compressed(os.path.join(OUT, "primitives.npz"), **d)
    print("primitives.npz", len(d), "arrays")


def run_tree(topo, frames, seed=1, keep_full=(), head=256):
    """Process `frames` LCG frames; per VFO per frame store sha of int16/int8 payload and of
    the final complex stream, plus heads; for VFO indices in keep_full store everything."""
    nodes, roots = ob.build_tree(KIND, topo)
    lcg = synth.Lcg(seed)
    d = {}
    for f in range(frames):
        iq = synth.lcg_frame(topo.frame, lcg)
        ob.process_roots(roots, iq)
        for i, (n, v) in enumerate(zip(nodes, topo.vfos)):
            z = n.stream()
            d[f"f{f}_v{i}_stream_sha"] = np.array(sha(z))
            d[f"f{f}_v{i}_stream_head"] = z[:head].copy()
            d[f"f{f}_v{i}_stream_absmax"] = np.float32(np.abs(z).max())
            leaf = not topo.children(i)
            if leaf:
                pay = n.usb() if v.demod_usb else n.iq()
                d[f"f{f}_v{i}_pay_sha"] = np.array(sha(pay))
                d[f"f{f}_v{i}_pay_head"] = pay[:head].copy()
                if i in keep_full:
                    d[f"f{f}_v{i}_pay"] = pay.copy()
            if i in keep_full and len(z) <= 8192:
                d[f"f{f}_v{i}_stream"] = z.copy()
    return d


def gen_chains():
    # config 1: 1 main + 1 sub, 6 frames = 1.5 s: both NCO tables wrap (L = 4 frames)
    d = run_tree(tp.config1(), 6, keep_full=(1,))
    np.savez_compressed(os.path.join(OUT, "config1.npz"), **d)
    print("config1.npz")
    # the whole sdr_25E profile, 5 frames (hashes + heads; VFO07 = d=4 and VFO19 = 47-tap LPF in full
    # for frame 0 only would be big -- heads suffice, the oracle is checked bit-exact by hash)
    d = run_tree(tp.profile_25e(), 5)
    np.savez_compressed(os.path.join(OUT, "profile_25e.npz"), **d)
    print("profile_25e.npz")
    # 54W style: 3 mains, 6 late-decimate subs (d=0, L=5, 49+47 taps) + 2 subs d=2,L=5 (12 k out)
    t = tp.config4(6)
    t.vfos.append(tp.VfoDesc(topic="VFO41", parent=0, fs=240000, decimate_count=2, mixer_freq=105571.0,
                             late_decimate=5, filter_bw=0, gain=tp._gain_pct(4), cstyle=1,
                             samples_per_buffer=60000))
    t.vfos.append(tp.VfoDesc(topic="VFO44", parent=0, fs=240000, decimate_count=2, mixer_freq=-74731.0,
                             late_decimate=5, filter_bw=4000, gain=tp._gain_pct(4), cstyle=1,
                             samples_per_buffer=60000))
    d = run_tree(t, 5, keep_full=(3,))
    np.savez_compressed(os.path.join(OUT, "profile_54w.npz"), **d)
    print("profile_54w.npz")
    # childless main VFOs -> compress(): cstyle 1 with scalecomp 1 and 16, cstyle 0
    t = tp.Topology(fs=1536000, frame=384000, name="compress")
    for cs, sc, top in ((1, 1, "IQ4A"), (1, 16, "IQ4B"), (0, 1, "IQ8")):
        t.vfos.append(tp.VfoDesc(topic=top, parent=-1, fs=1536000, decimate_count=3, mixer_freq=-496000.0,
                                 demod_usb=False, cstyle=cs, scalecomp=sc, samples_per_buffer=384000))
    d = run_tree(t, 2, keep_full=())
    np.savez_compressed(os.path.join(OUT, "compress.npz"), **d)
    print("compress.npz")
    # 288 kS/s profile shape: bufsplit 5 (frame 57600), main d=0, sub late /6 (73 taps)
    t = tp.Topology(fs=288000, frame=57600, bufsplit=5, name="288k")
    t.vfos.append(tp.VfoDesc(parent=-1, fs=288000, decimate_count=0, mixer_freq=0.0, demod_usb=False, cstyle=1,
                             samples_per_buffer=57600))
    t.vfos.append(tp.VfoDesc(topic="VFO51", parent=0, fs=288000, decimate_count=0, mixer_freq=54578.0,
                             late_decimate=6, filter_bw=10000, gain=tp._gain_pct(4), cstyle=1,
                             samples_per_buffer=57600))
    d = run_tree(t, 6, keep_full=())
    np.savez_compressed(os.path.join(OUT, "profile_288k.npz"), **d)
    print("profile_288k.npz")


OFAST_CASES = [("config1", 3), ("profile_25e", 3), ("54w", 2)]  # (tests/helpers.golden_topology key, frames)


def gen_ofast():
    """Outputs of the reference AS SHIPPED (-Ofast, SDRReceiver.pro:74-75; oracle/_ref/libsdrref_ofast.so) on the same
    LCG frames as the -O2 fixtures.  -Ofast lets the compiler reassociate the filter sums, so these differ from the
    canonical -O2 results in the last bits: per leaf and frame the fixture holds the int16 payload as a PATCH against
    the -O2 payload (indices + values where the two builds differ, and the sha of the whole -Ofast payload: a checker
    that holds the -O2 payload bit-exactly can rebuild the -Ofast one and prove it), and of every final complex
    stream the first 256 samples, every 128th sample and max|z|."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import golden_topology
    for key, frames in OFAST_CASES:
        topo = golden_topology(key)
        ref, fast = ob.build_tree("reference", topo), ob.build_tree("reference_ofast", topo)
        lcg = synth.Lcg(1)
        d = {"frames": np.int64(frames)}
        ndiff = total = 0
        for f in range(frames):
            iq = synth.lcg_frame(topo.frame, lcg)
            ob.process_roots(ref[1], iq)
            ob.process_roots(fast[1], iq)
            for i, v in enumerate(topo.vfos):
                z = fast[0][i].stream()
                d[f"f{f}_v{i}_stream_head"] = z[:256].copy()
                d[f"f{f}_v{i}_stream_every128"] = z[::128].copy()
                d[f"f{f}_v{i}_stream_absmax"] = np.float32(np.abs(z).max())
                if not topo.children(i) and v.demod_usb:
                    a, b = ref[0][i].usb(), fast[0][i].usb()
                    idx = np.flatnonzero(a != b).astype(np.int32)
                    d[f"f{f}_v{i}_pay_idx"] = idx
                    d[f"f{f}_v{i}_pay_val"] = b[idx].copy()
                    d[f"f{f}_v{i}_pay_sha"] = np.array(sha(b))
                    ndiff += idx.size
                    total += a.size
        np.savez_compressed(os.path.join(OUT, f"ofast_{key}.npz"), **d)
        print(f"ofast_{key}.npz: {ndiff} of {total} int16 samples differ from the -O2 build")


CAPTURE_FRAMES = 8  # 2 s of signal at 1.536 MS/s


def gen_capture():
    """Row "recorded IQ" (BASELINE.json north_star; SURVEY.md 8c: the reference holds no recording): the capture-like byte
    stream of sdrreceiver_amd/synth.py (capture_like_u8: tuner noise, strong carriers past +-100, an ADC offset, 600 / 1200 Bd
    BPSK and 10 500 Bd OQPSK bursts on sdr_25E VFO frequencies) through the shipped sdr_25E profile as the reference runs
    it: bytes -> b - 127 (jonti/sdr.cpp:43-49) -> DC-bias removal (sdrj.cpp:271-286, correct_dc_bias=1; restated in
    oracle/vfo_oracle.c -- sdrj.cpp itself cannot be compiled here) -> the REAL reference's vfo tree, -O2 and, as a
    patch against it, the -Ofast build the project ships.  Per VFO and frame: sha256 + head + max|z| of the final complex
    stream, sha256 + head of the int16 payload; for -Ofast the stream's every 128th sample and the payload patch."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    topo = tp.profile_25e()
    u8 = synth.capture_like_u8(CAPTURE_FRAMES, topo.frame, topo.fs)
    d = {"frames": np.int64(CAPTURE_FRAMES), "input_sha256": np.array(hashlib.sha256(u8.tobytes()).hexdigest())}
    ref = ob.build_tree("reference", topo)
    fast = ob.build_tree("reference_ofast", topo) if ob.have_reference_ofast() else None
    state = np.zeros(2, np.float32)
    ndiff = total = 0
    peak = 0
    for f in range(CAPTURE_FRAMES):
        iq = ob.u8_to_float(u8[2 * topo.frame * f: 2 * topo.frame * (f + 1)])
        ob.dc_correct(iq, state)
        d[f"f{f}_dc_state"] = state.copy()
        d[f"f{f}_raw_sha"] = np.array(sha(iq))
        ob.process_roots(ref[1], iq)
        if fast:
            ob.process_roots(fast[1], iq)
        for i, v in enumerate(topo.vfos):
            z = ref[0][i].stream()
            d[f"f{f}_v{i}_stream_sha"] = np.array(sha(z))
            d[f"f{f}_v{i}_stream_head"] = z[:64].copy()
            d[f"f{f}_v{i}_stream_absmax"] = np.float32(np.abs(z).max())
            if not topo.children(i):
                a = ref[0][i].usb()
                peak = max(peak, int(np.abs(a.astype(np.int32)).max()))
                d[f"f{f}_v{i}_pay_sha"] = np.array(sha(a))
                d[f"f{f}_v{i}_pay_head"] = a[:64].copy()
                if fast:
                    b = fast[0][i].usb()
                    idx = np.flatnonzero(a != b).astype(np.int32)
                    d[f"f{f}_v{i}_ofast_pay_idx"] = idx
                    d[f"f{f}_v{i}_ofast_pay_val"] = b[idx].copy()
                    d[f"f{f}_v{i}_ofast_pay_sha"] = np.array(sha(b))
                    ndiff += idx.size
                    total += a.size
            if fast:
                zf = fast[0][i].stream()
                d[f"f{f}_v{i}_ofast_stream_every128"] = zf[::128].copy()
                d[f"f{f}_v{i}_ofast_stream_absmax"] = np.float32(np.abs(zf).max())
    np.savez_compressed(os.path.join(OUT, "capture_25e.npz"), **d)
    print(f"capture_25e.npz: {CAPTURE_FRAMES} frames, int16 peak {peak}; -Ofast: {ndiff} of {total} int16 samples differ from the -O2 build")


def gen_zmq():
    """ZmqPublisher::publish framing through the real libzmq (ipc transport)."""
    import ctypes as C
    R = ob.load(KIND)
    fn = R.fn("publish_roundtrip")
    payload = (np.arange(100, dtype=np.int16) - 50).tobytes()
    bufs = [C.create_string_buffer(4096) for _ in range(3)]
    lens = [C.c_int(0) for _ in range(3)]
    addr = f"ipc:///tmp/sdrref-golden-{os.getpid()}.ipc".encode()
    got = fn(addr, payload, len(payload), b"VFO07-extra", 24000, bufs[0], C.byref(lens[0]), bufs[1],
             C.byref(lens[1]), bufs[2], C.byref(lens[2]), 4096)
    assert got == 3, got
    frames = [bufs[k].raw[:lens[k].value] for k in range(3)]
    np.savez_compressed(os.path.join(OUT, "zmq_framing.npz"),
                        topic_in=np.array("VFO07-extra"), rate_in=np.uint32(24000),
                        payload_in=np.frombuffer(payload, np.uint8),
                        frame0=np.frombuffer(frames[0], np.uint8), frame1=np.frombuffer(frames[1], np.uint8),
                        frame2=np.frombuffer(frames[2], np.uint8))
    print("zmq_framing.npz", [len(f) for f in frames], frames[0], frames[1].hex())


DROPIN_CASES = [("config1", 3, "VFO01"), ("profile_25e", 3, "VFO19"), ("config4_12", 2, "C0003")]


def gen_dropin():
    """What a ZMQ subscriber receives from the REFERENCE's `class vfo` driven through its public
    interface by host/qt/dropin_client.cpp (oracle/_ref/libdropin_ref.so, `make -C host/qt`):
    per message topic bytes, rate, payload length and FNV-1a hash, plus the fftData emissions.
    The adapter build of the same client (libdropin_sdrx.so, GPU) must reproduce these lines."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(OUT))
    for name, frames, fft in DROPIN_CASES:
        out = subprocess.check_output([sys.executable, os.path.join(root, "tools", "dropin_run.py"), "ref", name, str(frames), fft], text=True)
        lines = [json.loads(l) for l in out.splitlines()]
        with open(os.path.join(OUT, f"dropin_{name}.json"), "w") as f:
            json.dump({"frames": frames, "fft_topic": fft, "lines": lines}, f, indent=0)
        print(f"dropin_{name}.json", len(lines), "lines")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "dropin":
        gen_dropin()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "capture":
        if not ob.have_reference():
            sys.exit("oracle/_ref/libsdrref.so missing: run `make -C oracle/ref` first (needs /root/reference)")
        gen_capture()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "ofast":
        if not (ob.have_reference() and ob.have_reference_ofast()):
            sys.exit("oracle/_ref/libsdrref.so / libsdrref_ofast.so missing: run `make -C oracle/ref` first (needs /root/reference)")
        gen_ofast()
        sys.exit(0)
    if not ob.have_reference():
        sys.exit("oracle/_ref/libsdrref.so missing: run `make -C oracle/ref` first (needs /root/reference)")
    gen_primitives()
    gen_chains()
    gen_capture()
    gen_zmq()
    gen_dropin()
    if ob.have_reference_ofast():
        gen_ofast()
