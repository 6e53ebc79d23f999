"""The C++ host layer (host/sdrx_host.hpp: the reference's vfo / sdrj interface and the INI front
door of mainwindow.cpp:27-233, Qt-free, over the C ABI), driven through host/sdrx_demo."""
import json
import os
import subprocess

import numpy as np
import pytest

from helpers import REFERENCE_ROOT
from sdrreceiver_amd import synth, topology as tp
from test_topology import INI_25E_LIKE

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEMO = os.path.join(ROOT, "host", "sdrx_demo")

INI_54W_LIKE = """
sample_rate=1920000
center_frequency=1545939000
zmq_address=tcp://*:6004
[main_vfos]
size=2
1\\frequency=1545120000
1\\out_rate=240000
2\\frequency=1546120000
2\\out_rate=240000
[vfos]
size=3
1\\frequency=1545014429
1\\gain=4
1\\data_rate=600
1\\topic=VFO41
2\\frequency=1546045422
2\\gain=4
2\\data_rate=10500
2\\topic=VFO51
3\\frequency=1546061717
3\\gain=4
3\\data_rate=10500
3\\filter_bandwidth=10000
3\\topic=VFO52
"""


def _build():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "host")], stdout=subprocess.DEVNULL)


def _dump(ini_text, tmp_path):
    _build()
    p = tmp_path / "profile.ini"
    p.write_text(ini_text)
    out = subprocess.check_output([DEMO, str(p), "--dump"], text=True)
    return [json.loads(l) for l in out.splitlines()]


def _check_dump(rows, topo):
    head, rest = rows[0], rows[1:]
    assert (head["fs"], head["frame"], head["bufsplit"], bool(head["correct_dc"])) == (topo.fs, topo.frame, topo.bufsplit, topo.correct_dc)
    mains = [v for v in topo.vfos if v.parent < 0]
    subs = [v for v in topo.vfos if v.parent >= 0]
    assert head["n"] == len(topo.vfos)
    for r, v in zip(rest[:len(mains)], mains):
        assert (r["fs"], r["d"], r["mixer"], bool(r["usb"]), r["spb"], r["scalecomp"]) == \
            (v.fs, v.decimate_count, v.mixer_freq, v.demod_usb, v.samples_per_buffer, v.scalecomp)
    # the dump lists subs grouped by main (VFOsub[i]); within a main the INI order is kept
    got = {(r["sub_of"], r["topic"]): r for r in rest[len(mains):]}
    assert len(got) == len(subs)
    for v in subs:
        r = got[(v.parent, v.topic)]
        assert (r["fs"], r["d"], r["late"], r["mixer"], r["bw"], r["spb"]) == \
            (v.fs, v.decimate_count, v.late_decimate, v.mixer_freq, v.filter_bw, v.samples_per_buffer)
        assert np.float32(r["gain"]) == np.float32(v.gain)


INI_25E_OFFSET = INI_25E_LIKE.replace("mix_offset=0", "mix_offset=-1750")   # mainwindow.cpp:65,151: added to every sub VFO's frequency
INI_25E_OFFSET_FAR = INI_25E_LIKE.replace("mix_offset=0", "mix_offset=-700000")  # ... far enough to move VFO19 under the other main


@pytest.mark.parametrize("ini", [INI_25E_LIKE, INI_54W_LIKE, INI_25E_OFFSET, INI_25E_OFFSET_FAR])
def test_cpp_ini_front_door_matches_python_rules(ini, tmp_path):
    _check_dump(_dump(ini, tmp_path), tp.topology_from_ini(ini))


@pytest.mark.skipif(not os.path.isdir(REFERENCE_ROOT), reason="needs the reference's sample INIs")
def test_cpp_ini_front_door_on_shipped_profiles(tmp_path):
    d = os.path.join(REFERENCE_ROOT, "sample_ini")
    for name in sorted(os.listdir(d)):
        text = open(os.path.join(d, name), encoding="utf-8", errors="replace").read()
        _check_dump(_dump(text, tmp_path), tp.topology_from_ini(text))


def test_cpp_ini_front_door_under_sanitizers(tmp_path):
    """host/sdrx_host.hpp's parser and tree builder under ASan + UBSan (leak check on), on the INIs of
    this file and -- where present -- every shipped sample profile."""
    exe = tmp_path / "demo_san"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-o", str(exe),
                           os.path.join(ROOT, "host", "demo.cpp"), "-L" + os.path.join(ROOT, "sdrreceiver_amd"), "-lsdrx",
                           "-Wl,-rpath," + os.path.join(ROOT, "sdrreceiver_amd"), "-Wl,-rpath,/opt/rocm/lib"])
    inis = []
    for k, text in enumerate([INI_25E_LIKE, INI_54W_LIKE]):
        p = tmp_path / f"p{k}.ini"
        p.write_text(text)
        inis.append(str(p))
    d = os.path.join(REFERENCE_ROOT, "sample_ini")
    if os.path.isdir(d):
        inis += [os.path.join(d, n) for n in sorted(os.listdir(d))]
    for ini in inis:
        r = subprocess.run([str(exe), ini, "--dump"], capture_output=True, text=True, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
        assert r.returncode == 0 and r.stderr == "", (ini, r.stderr[-1500:])


def test_cpp_missing_profile_is_an_error(tmp_path):
    _build()
    r = subprocess.run([DEMO, str(tmp_path / "nope.ini"), "--dump"], capture_output=True, text=True)
    assert r.returncode == 1 and "doesn't exist" in r.stderr


def test_header_is_plain_c_and_fails_loudly_without_a_gpu():
    """host/abi_check.c: gcc -std=c99 -pedantic -Werror over include/sdrx.h, linked to libsdrx.so.
    On a box without a GPU sdrx_create must refuse (no CPU fallback); on the GPU box the same
    program pushes a frame through the VFO01 chain (test below)."""
    import torch
    _build()
    r = subprocess.run([os.path.join(ROOT, "host", "abi_check")], capture_output=True, text=True)
    assert "sdrx ABI version 5, sizeof(sdrx_vfo_desc) = 56" in r.stdout
    if not torch.cuda.is_available():
        assert r.returncode == 3 and "no CPU fallback" in r.stderr


@pytest.mark.gpu
def test_c99_program_over_the_abi():
    _build()
    r = subprocess.run([os.path.join(ROOT, "host", "abi_check")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    # one frame through sdrx_process, then three through sdrx_group_submit / _wait on a two-entry device list
    assert r.stdout.count("published VFO01 rate 12000 bytes 6000") == 4


@pytest.mark.gpu
def test_c99_host_pipelined_interface():
    """host/abi_bench.c: a C99 host drives sdrx_process and the sdrx_submit / sdrx_wait pair on a 64-sub
    config-3 tree from a pageable buffer; every frame delivers every leaf (66 messages would be wrong: the
    two main VFOs have children and publish nothing), and the pipelined loop is not slower than the
    synchronous one."""
    _build()
    r = subprocess.run([os.path.join(ROOT, "host", "abi_bench"), "64", "8"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["messages_per_frame"] == 64 and out["payload_bytes_per_frame"] == 32 * 6000 + 32 * 24000
    assert out["sdrx_submit_wait_ms"] <= out["sdrx_process_ms"] * 1.1, out


def _fnv1a(b: bytes) -> int:
    h = 1469598103934665603
    for x in b:
        h = ((h ^ x) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


@pytest.mark.gpu
@pytest.mark.parametrize("u8", [False, True])
def test_cpp_host_end_to_end(u8, tmp_path):
    """INI -> vfo setters -> sdrj::demodData (host float path with host-side DC correction, or the
    byte path with LUT + DC on the device) -> publish hook, against the Python host path with the
    oracle's DC correction: same messages, same order, same bytes."""
    from oracle import binding as ob
    from sdrreceiver_amd.receiver import Receiver
    _build()
    p = tmp_path / "profile.ini"
    p.write_text(INI_25E_LIKE)
    out = subprocess.check_output([DEMO, str(p), "--frames", "2"] + (["--u8"] if u8 else []), text=True)
    got = [l.split() for l in out.splitlines()]
    topo = tp.topology_from_ini(INI_25E_LIKE)
    rx = Receiver.from_topology(topo)
    lcg = synth.Lcg(1)
    state = np.zeros(2, np.float32)
    want = []
    for f in range(2):
        iq = synth.lcg_frame(topo.frame, lcg)
        ob.dc_correct(iq, state)  # correct_dc_bias=1 in the profile
        rx.process(iq)
        for topic, rate, payload in rx.published:
            want.append([str(f), topic.rstrip(b"\0").decode(), str(rate), str(len(payload)), f"{_fnv1a(payload):016x}"])
    rx.close()
    assert got == want


@pytest.mark.gpu
@pytest.mark.parametrize("u8", [False, True])
def test_cpp_host_on_several_devices(u8, tmp_path):
    """host/sdrx_host.hpp's sdrj over a device LIST (sdrx_group_*: the native single-process multi-GPU
    host; here three shards on the one GPU of the test box): the same messages in the same order as the
    single-device run, and the VFO spectrum tap served from whichever member holds the VFO."""
    _build()
    p = tmp_path / "profile.ini"
    p.write_text(INI_25E_LIKE)
    for fft, frames in (("VFO19", "3"), ("Main", "9")):
        args = [DEMO, str(p), "--frames", frames] + (["--u8"] if u8 else []) + ["--fft", fft]
        one = subprocess.check_output(args, text=True).splitlines()
        many = subprocess.check_output(args + ["--devices", "0,0,0"], text=True).splitlines()
        assert len(one) >= 9 and any(l.startswith("fft ") for l in one) and one == many, fft


@pytest.mark.gpu
def test_cpp_host_fft_taps(tmp_path):
    """fftVFOSlot / fftData of vfo (vfo.cpp:290-293,492-509: the selected VFO's
    decimate[decimateCount] after every frame) and of sdrj (sdrj.cpp:84-101,296-303: the raw
    samples, on the 5th call after the selection and every 4th from then on)."""
    from oracle import binding as ob
    from sdrreceiver_amd.receiver import Receiver
    _build()
    p = tmp_path / "profile.ini"
    p.write_text(INI_25E_LIKE)
    topo = tp.topology_from_ini(INI_25E_LIKE)
    vid = [i for i, v in enumerate(topo.vfos) if v.topic == "VFO19"][0]

    def run(*extra):
        out = subprocess.check_output([DEMO, str(p), "--frames", "9", *extra], text=True)
        return [l.split() for l in out.splitlines() if l.startswith("fft ")]

    rx = Receiver.from_topology(topo)
    lcg = synth.Lcg(1)
    state = np.zeros(2, np.float32)
    want_vfo, want_raw = [], []
    for f in range(9):
        iq = synth.lcg_frame(topo.frame, lcg)
        ob.dc_correct(iq, state)
        rx.process(iq)
        st = rx.stream(vid)
        want_vfo.append(["fft", str(f), "VFO19", str(st.size), f"{_fnv1a(st.tobytes()):016x}"])
        if f in (4, 8):
            want_raw.append(["fft", str(f), "Main", str(topo.frame), f"{_fnv1a(iq.tobytes()):016x}"])
    rx.close()
    assert run("--fft", "VFO19") == want_vfo
    assert run("--fft", "Main") == want_raw
    assert run("--fft", "NOSUCH") == []


@pytest.mark.gpu
@pytest.mark.parametrize("devices", [None, "0,0"])
def test_cpp_host_fft_tap_on_a_fused_late_decimation(devices, tmp_path):
    """The 54W-like profile: VFO51 / VFO52 are /5 leaves with decimate_count 0 -- their low-pass runs inside the mix wave
    and decimate[0] exists only while the leaf is the tap.  host/sdrx_host.hpp names the selected VFO to the library before
    the frame (sdrx_set_tap), on one device and over a device list: the fftData log equals the stream an oracle-checked
    Receiver with that tap returns, the published messages do not change."""
    from sdrreceiver_amd.receiver import Receiver
    _build()
    p = tmp_path / "profile.ini"
    p.write_text(INI_54W_LIKE)
    topo = tp.topology_from_ini(INI_54W_LIKE)
    vid = [i for i, v in enumerate(topo.vfos) if v.topic == "VFO52"][0]
    assert topo.vfos[vid].late_decimate == 5 and topo.vfos[vid].decimate_count == 0
    rx = Receiver.from_topology(topo)
    rx.set_tap(vid)
    lcg = synth.Lcg(1)
    want = []
    for f in range(4):
        iq = synth.lcg_frame(topo.frame, lcg)
        rx.process(iq)
        st = rx.stream(vid)
        want.append(["fft", str(f), "VFO52", str(st.size), f"{_fnv1a(st.tobytes()):016x}"])
    rx.close()
    extra = ["--devices", devices] if devices else []
    out = subprocess.check_output([DEMO, str(p), "--frames", "4", "--fft", "VFO52", *extra], text=True).splitlines()
    plain = subprocess.check_output([DEMO, str(p), "--frames", "4", *extra], text=True).splitlines()
    assert [l.split() for l in out if l.startswith("fft ")] == want
    assert [l for l in out if not l.startswith("fft ")] == plain
