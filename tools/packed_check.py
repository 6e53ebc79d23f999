#!/usr/bin/env python3
"""tools/packed_check.py -- sha256 of every payload and pre-quantisation float of a few trees in the tolerance and the robust
arithmetic (run it under two builds, SDRX_LIB=..., and diff: round 6's packed-FMA demodulation must not change one bit)."""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
from helpers import golden_topology  # noqa: E402
from sdrreceiver_amd import synth, topology as tp  # noqa: E402
from sdrreceiver_amd.receiver import Receiver  # noqa: E402

for name, topo in (("profile_25e", golden_topology("profile_25e")), ("54w", golden_topology("54w")), ("288k", golden_topology("288k")),
                   ("config3-64", tp.config3(64)), ("config4-12", tp.config4(12))):
    for arith in (0, 2):
        rx = Receiver.from_topology(topo, exact=arith, keep_prequant=True)
        lcg = synth.Lcg(3)
        h = hashlib.sha256()
        for f in range(4):
            iq = synth.lcg_frame(topo.frame, lcg) + synth.tone_frame(topo.frame, topo.fs, [(-377000.0, 25.0)], f * topo.frame)
            rx.process(iq)
            for i in topo.leaves_in_publish_order():
                h.update(rx.output(i).tobytes())
                if topo.vfos[i].demod_usb:
                    h.update(rx.prequant(i).tobytes())
        rx.close()
        print(name, arith, h.hexdigest())
