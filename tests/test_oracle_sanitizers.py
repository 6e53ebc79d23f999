"""The CPU restatement under AddressSanitizer + UndefinedBehaviorSanitizer (GPU sanitizers are not
available on the pool: the checker, at least, is clean), and -- same binary -- agreement of the
sanitized build with the regular oracle library."""
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_is_clean_under_asan_and_ubsan(tmp_path):
    exe = tmp_path / "san_driver"
    subprocess.check_call(["gcc", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-ffp-contract=off",
                           "-o", str(exe), os.path.join(ROOT, "oracle", "san_driver.c"), os.path.join(ROOT, "oracle", "vfo_oracle.c"), "-lm"])
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr
    got = dict(line.split(" ", 1) for line in r.stdout.splitlines())
    assert set(got) == {"VFO01", "VFO19", "IQ00", "VFO51", "VFO41"}
    assert not any(v.endswith("sum=0") for v in got.values())

    # the regular (-O2, OpenMP) library on the first tree gives the same int16 audio
    from oracle import binding as ob
    from sdrreceiver_amd import synth, topology as tp
    t = tp.Topology(fs=1536000, frame=384000, name="san")
    t.vfos.append(tp.VfoDesc(parent=-1, fs=1536000, decimate_count=2, mixer_freq=484000.0, demod_usb=False, cstyle=1, samples_per_buffer=384000))
    t.vfos.append(tp.VfoDesc(topic="VFO01", parent=0, fs=384000, decimate_count=5, mixer_freq=110854.0, filter_bw=4000, gain=tp._g(0.05),
                             cstyle=1, samples_per_buffer=96000))
    nodes, roots = ob.build_tree("port", t)
    lcg = synth.Lcg(1)
    state = np.zeros(2, np.float32)
    for _ in range(3):
        iq = synth.lcg_frame(480000, lcg)[: 2 * 384000].copy()  # the driver draws 54W-sized frames and uses their head
        ob.dc_correct(iq, state)
        ob.process_roots(roots, iq)
    s = 0
    for x in nodes[1].usb().tolist():
        s = (s * 31 + (x & 0xFFFFFFFFFFFFFFFF)) & 0xFFFFFFFFFFFFFFFF
    assert got["VFO01"] == f"n=3000 sum={s}"
