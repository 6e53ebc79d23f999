#!/usr/bin/env python3
"""tests/dc_soak.py [regimes] [seed] -- soak of the exact DC-bias removal (k_dc_chain_spec) against the oracle's recurrence
(oracle/vfo_oracle.c: sdrj.cpp:277-283 restated): random (offset I, offset Q, noise sigma, frame length, blocks per step)
regimes, frames from the zero state and on, a step change of the offset in the middle, clipped bytes.  The checker is the
oracle; the thing checked is the HIP path.  Not collected by pytest (no test_ prefix): run it on the GPU box."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
from helpers import bits  # noqa: E402
from oracle import binding as ob  # noqa: E402
from sdrreceiver_amd import topology as tp  # noqa: E402
from sdrreceiver_amd.receiver import Receiver  # noqa: E402

n_reg = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
bad = 0
for k in range(n_reg):
    di, dq = rng.choice([0.0, 0.3, 1.3, -0.7, 5.0, -20.0, 60.0, -120.0], 2) + rng.normal(0, 0.2, 2)
    sigma = float(rng.choice([0.0, 0.5, 2.0, 7.0, 20.0, 60.0]))
    n = 16 * int(rng.choice([1024 * 8, 1024 * 8 + 1, 4096 + 37, 24000, 1024 * 24 - 1, 65536 + 64 * 5]))
    if 0 < n % 1024 < 256:  # (sdrx_finalize: a last chunk of a frame is at least 256 samples)
        n += 256
    per_step = int(rng.choice([1, 2, 4, 8]))
    t = tp.Topology(fs=4 * n, frame=n, name=f"soak{k}")
    t.vfos.append(tp.VfoDesc(topic="M", parent=-1, fs=4 * n, decimate_count=2, mixer_freq=float(n // 7), demod_usb=False, cstyle=1,
                             samples_per_buffer=n))
    rx = Receiver.from_topology(t, exact=True, dc_blocks_per_step=per_step)
    state = np.zeros(2, np.float32)
    frames = int(rng.integers(6, 14))
    st = None
    for f in range(frames):
        if f == frames // 2:  # the offset steps: the estimate leaves its threshold and travels
            di, dq = di + rng.normal(0, 3.0), dq - rng.normal(0, 3.0)
        z = rng.standard_normal(2 * n) * sigma
        z[0::2] += di
        z[1::2] += dq
        b = np.clip(np.rint(z) + 127, 0, 255).astype(np.uint8)
        rx.process_u8(b, correct_dc=True)
        iq = ob.u8_to_float(b)
        ob.dc_correct(iq, state)
        if not np.array_equal(bits(rx.raw()), bits(iq.view(np.complex64))):
            bad += 1
            print(f"MISMATCH regime {k} frame {f}: offsets {di:.2f} {dq:.2f} sigma {sigma} n {n} per_step {per_step}", flush=True)
            break
    st = rx.stats()
    print(f"regime {k}: offsets {di:7.2f} {dq:7.2f} sigma {sigma:5.1f} n {n:7d} x{frames:2d} per_step {per_step}: blocks {st['dc_blocks']} again {st['dc_retried_blocks']} "
          f"sequential {st['dc_fallback_blocks']}", flush=True)
    rx.close()
print("soak:", "FAILED" if bad else "passed", n_reg, "regimes, seed", seed)
sys.exit(1 if bad else 0)
