"""tools/dropin_run.py <ref|sdrx> <topology> <frames> [fft_topic] [copies] [repeat]

Runs host/qt/dropin_client.cpp (a client of the reference's unmodified vfo.h public interface) with
one of the two implementations of `class vfo` behind it -- oracle/_ref/libdropin_ref.so (the
reference's own sources) or oracle/_ref/libdropin_sdrx.so (host/qt/vfo_adapter.cpp over libsdrx.so,
needs the GPU) -- and prints what a ZMQ subscriber received: one JSON line per message
{"topic": hex of the 5 topic bytes, "rate": u32, "len": payload bytes, "fnv": fnv1a64 of payload},
then one line per fftData emission {"fft": [frame, topic, count, fnv1a64 hex]}.
`copies` receivers are built from the same description and fed in turn; the whole build-run-delete
cycle is done `repeat` times in this process (stop / start), output concatenated.
A separate process per invocation: the reference's bind publisher is a process-wide static."""
import ctypes as C
import json
import os
import struct
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def fnv1a(b: bytes) -> int:
    h = 1469598103934665603
    for x in b:
        h = ((h ^ x) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def topology(name):
    from sdrreceiver_amd import topology as tp
    if name == "config1":
        return tp.config1()
    if name == "profile_25e":
        return tp.profile_25e()
    if name == "config4_12":
        return tp.config4(12)
    if name == "config4_12_shared_topic":
        # two /5 leaves (under different main VFOs) and one main VFO's IQ publisher carry ONE topic: fftVFOSlot(topic) sets
        # emitFFT on every VFO whose zmqTopic equals the string (vfo.cpp:492-509), so all of them emit fftData
        t = tp.config4(12)
        t.vfos[4].topic = "SHARE"
        t.vfos[9].topic = "SHARE"
        return t
    if name.startswith("random:"):  # the seeded random trees of tests/test_gpu_parity.py
        import numpy as np
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from helpers import random_topology  # test infrastructure: the generator lives with the tests
        return random_topology(np.random.default_rng(1000 + int(name.split(":")[1])))
    raise SystemExit(f"unknown topology {name}")


def probe(kind):
    """tools/dropin_run.py probe <ref|sdrx>: does vfo::init throw, and what, for a few descriptions
    (firfilter::sanity_check_1f, firfilter.cpp:122-134, reached from vfo.cpp:82-87,110-115)."""
    C.CDLL("libstdc++.so.6", mode=C.RTLD_GLOBAL)
    from sdrreceiver_amd import topology as tp
    from sdrreceiver_amd._lib import desc_to_c
    lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", f"libdropin_{kind}.so"))
    base = tp.config1().vfos[1]  # 384 k -> 12 k, filter_bw 4000
    from dataclasses import replace
    cases = {
        "ok": base,
        "bw_above_half_rate": replace(base, filter_bw=7000),          # 7000 > 12000 / 2
        "bw_exactly_half_rate": replace(base, filter_bw=6000),        # allowed: fa <= fs / 2
        "late_ok": replace(base, fs=240000, decimate_count=0, late_decimate=5, filter_bw=10000, samples_per_buffer=60000),
        "late_bw_too_wide": replace(base, fs=240000, decimate_count=0, late_decimate=5, filter_bw=30000, samples_per_buffer=60000),
        "main_is_never_filtered": replace(base, demod_usb=False, filter_bw=0),
        "non_usb_with_a_bad_bandwidth": replace(base, demod_usb=False, filter_bw=7000),  # designed (and rejected) all the same
    }
    addr = f"ipc:///tmp/sdrx_dropin_probe_{os.getpid()}".encode()
    out = {}
    for name, d in cases.items():
        what = C.create_string_buffer(256)
        c = desc_to_c(d)
        rc = lib.dropin_init_probe(C.byref(c), addr, what, len(what))
        out[name] = [rc, what.value.decode()]
    print(json.dumps(out))
    try:
        os.unlink(addr.decode()[len("ipc://"):])
    except OSError:
        pass


def timing(kind, n_subs, frames):
    """tools/dropin_run.py time <ref|sdrx> <n_subs> <frames>: milliseconds per frame of the demodData loop over the
    main VFOs through the public interface of vfo.h (config-3 tree with n_subs sub VFOs), as JSON."""
    C.CDLL("libstdc++.so.6", mode=C.RTLD_GLOBAL)
    from sdrreceiver_amd import topology as tp
    from sdrreceiver_amd._lib import VfoDescC, desc_to_c
    topo = tp.config3(n_subs)
    lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", f"libdropin_{kind}.so"))
    lib.dropin_time.restype = C.c_double
    descs = (VfoDescC * len(topo.vfos))(*[desc_to_c(d) for d in topo.vfos])
    addr = f"ipc:///tmp/sdrx_dropin_time_{os.getpid()}".encode()
    ms = lib.dropin_time(descs, len(topo.vfos), addr, 3, frames)
    print(json.dumps({"kind": kind, "sub_vfos": n_subs, "frames": frames, "ms_per_frame": round(ms, 4),
                      "mode": {k: os.environ[k] for k in ("SDRX_PIPELINE", "SDRX_SHARE_UPLOAD", "SDRX_DEVICES") if k in os.environ}}))
    try:
        os.unlink(addr.decode()[len("ipc://"):])
    except OSError:
        pass


def main():
    if sys.argv[1] == "probe":
        return probe(sys.argv[2])
    if sys.argv[1] == "time":
        return timing(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]))
    kind, name, frames = sys.argv[1], sys.argv[2], int(sys.argv[3])
    fft_topic = sys.argv[4] if len(sys.argv) > 4 else ""
    copies = int(sys.argv[5]) if len(sys.argv) > 5 else 1
    repeat = int(sys.argv[6]) if len(sys.argv) > 6 else 1
    C.CDLL("libstdc++.so.6", mode=C.RTLD_GLOBAL)  # the system's, before /opt/conda's older one can be picked up
    from sdrreceiver_amd._lib import VfoDescC, desc_to_c
    topo = topology(name)
    lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", f"libdropin_{kind}.so"))
    lib.dropin_run.restype = C.c_int
    descs = (VfoDescC * len(topo.vfos))(*[desc_to_c(d) for d in topo.vfos])
    cap = 64 << 20
    out = C.create_string_buffer(cap)
    fft = C.create_string_buffer(1 << 16)
    addr = f"ipc:///tmp/sdrx_dropin_{os.getpid()}".encode()
    for _ in range(repeat):
        run_once(lib, descs, len(topo.vfos), addr, frames, fft_topic, out, cap, fft, copies)
    try:
        os.unlink(addr.decode()[len("ipc://"):])
    except OSError:
        pass


def run_once(lib, descs, nv, addr, frames, fft_topic, out, cap, fft, copies):
    n = lib.dropin_run(descs, nv, addr, frames, fft_topic.encode(), out, cap, fft, len(fft), copies)
    if n < 0:
        raise SystemExit(f"dropin_run failed: {n}")
    buf, pos = out.raw[:n], 0
    while pos < n:
        (nparts,) = struct.unpack_from("<I", buf, pos)
        pos += 4
        parts = []
        for _ in range(nparts):
            (ln,) = struct.unpack_from("<I", buf, pos)
            pos += 4
            parts.append(buf[pos:pos + ln])
            pos += ln
        assert nparts == 3 and len(parts[0]) == 5 and len(parts[1]) == 4, (nparts, [len(p) for p in parts])
        print(json.dumps({"topic": parts[0].hex(), "rate": struct.unpack("<I", parts[1])[0], "len": len(parts[2]),
                          "fnv": f"{fnv1a(parts[2]):016x}"}))
    for line in fft.value.decode().splitlines():
        f, t, cnt, h = line.split()
        print(json.dumps({"fft": [int(f), t, int(cnt), h]}))


if __name__ == "__main__":
    main()
