import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sdrreceiver_amd import topology as tp
from sdrreceiver_amd.receiver import Receiver
from oracle import binding as ob

t = tp.Topology(fs=1536000, frame=384000, name="dc")
t.vfos.append(tp.VfoDesc(topic="RAW", parent=-1, fs=1536000, decimate_count=0, mixer_freq=0.0, demod_usb=False, cstyle=0, samples_per_buffer=384000))
for exact in (True, False):
    rx = Receiver.from_topology(t, exact=exact)
    nodes, roots = ob.build_tree("port", t)
    rng = np.random.default_rng(5)
    st = np.zeros(2, np.float32)
    st64 = np.zeros(2)
    for f in range(8):
        b = rng.integers(0, 256, 2 * t.frame, dtype=np.uint8)
        b[0::2] = np.clip(b[0::2].astype(int) + 3, 0, 255)
        b[1::2] = np.clip(b[1::2].astype(int) - 2, 0, 255)
        rx.process_u8(b, correct_dc=True)
        iq = ob.u8_to_float(b)
        ob.dc_correct(iq, st)
        ob.process_roots(roots, iq)
        d = rx.stream(0) - nodes[0].stream()
        print(exact, f, "max|d| re %.3e im %.3e  mean d re %.3e im %.3e  state %s" % (np.abs(d.real).max(), np.abs(d.imag).max(), d.real.mean(), d.imag.mean(), st))
    rx.close()
