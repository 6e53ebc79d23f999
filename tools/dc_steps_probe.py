#!/usr/bin/env python3
"""tools/dc_steps_probe.py [frames] -- the exact DC-bias removal (k_dc_products + k_dc_chain_spec + k_dc_apply, event-timed as
one group) for every value of option dc_blocks_per_step, on the capture-like stream, a quiet front end and the zero-offset LCG
frames; with the counters: blocks walked / taken again on their own / redone with the sequential operations."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from sdrreceiver_amd import synth, topology as tp  # noqa: E402
from sdrreceiver_amd.receiver import Receiver  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 16
n = 384000
topo = tp.Topology(fs=1536000, frame=n, name="dcprobe")
topo.vfos.append(tp.VfoDesc(topic="M", parent=-1, fs=1536000, decimate_count=3, mixer_freq=-496000.0, demod_usb=False, cstyle=1, samples_per_buffer=n))
cap = synth.capture_like_u8(8, n, 1536000)
rng = np.random.default_rng(5)
streams = {
    "capture-like": [cap[2 * n * f: 2 * n * (f + 1)] for f in range(8)],
    "quiet front end": [np.clip(np.rint(rng.standard_normal(2 * n) * 2.0 + np.tile([1.3, -0.7], n)) + 127, 0, 255).astype(np.uint8) for _ in range(8)],
    "zero-offset LCG": [(synth.lcg_frame(n, synth.Lcg(1)) + 127).astype(np.uint8)] * 8,
    "pinned (offset 30, sigma 3)": [np.clip(np.rint(rng.standard_normal(2 * n) * 3.0 + np.tile([30.0, -2.0], n)) + 127, 0, 255).astype(np.uint8) for _ in range(8)],
}
PER_STEP = [int(x) for x in os.environ.get("PER_STEP", "1,2,4,8").split(",")]
out = {}
for name, fr in streams.items():
    for per_step in PER_STEP:
        rx = Receiver.from_topology(topo, dc_blocks_per_step=per_step)
        rx.set_publish(False)
        for k in range(16):
            rx.process_u8(fr[k % 8], correct_dc=True)
        st0 = rx.stats()
        rx.enable_kernel_timing(True)
        for k in range(frames):
            rx.process_u8(fr[k % 8], correct_dc=True)
        kt = rx.kernel_times()
        st1 = rx.stats()
        ing = [v for k, v in kt.items() if "ingest" in k.lower() or "dc" in k.lower()]
        out[f"{name}, {per_step}"] = {"group_ms": {k: round(v["ms"] / max(1, v["launches"]), 4) for k, v in kt.items() if v["launches"]},
                                      "walked": int(st1["dc_blocks"] - st0["dc_blocks"]), "again": int(st1["dc_retried_blocks"] - st0["dc_retried_blocks"]),
                                      "sequential": int(st1["dc_fallback_blocks"] - st0["dc_fallback_blocks"])}
        print(name, per_step, json.dumps(out[f"{name}, {per_step}"]), flush=True)
        rx.close()
