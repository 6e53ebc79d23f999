#!/usr/bin/env python3
"""tools/dc_overlap_probe.py [sync|pipelined|both] [frames] -- dongle bytes with the exact DC-bias removal (sdrj.cpp:271-286)
through config 3, frame after frame: ms per frame, and the event-timed ingest group (k_dc_products + k_dc_chain +
k_dc_apply) per frame, synchronous (sdrx_process_u8) and pipelined (sdrx_submit_u8 / sdrx_wait).  Run it under
`rocprofv3 --kernel-trace --stats` for k_dc_chain's own duration in each mode."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from sdrreceiver_amd import synth, topology as tp  # noqa: E402
from sdrreceiver_amd.receiver import Receiver  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "both"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 24
topo = tp.config3(1024)
cap = synth.capture_like_u8(8, topo.frame, topo.fs)  # an ADC offset of (+1.3, -0.7) LSB: what the correction exists for
cap = [cap[2 * topo.frame * f: 2 * topo.frame * (f + 1)] for f in range(8)]
out = {}
for m in (["sync", "pipelined", "sync, every sample in turn", "sync, zero-offset LCG frames"] if mode == "both" else [mode]):
    rx = Receiver.from_topology(topo, dc_speculative="every sample" not in m)
    rx.set_publish(False)
    if "LCG" in m:
        cap_m = [(synth.lcg_frame(topo.frame, synth.Lcg(1)) + 127).astype(np.uint8)] * 8
    else:
        cap_m = cap
    for k in range(16):
        rx.process_u8(cap_m[k % 8], correct_dc=True)
    rx.enable_kernel_timing(True)
    t0 = time.perf_counter()
    st0 = rx.stats()
    if m.startswith("sync"):
        for k in range(frames):
            rx.process_u8(cap_m[k % 8], correct_dc=True)
    else:
        rx.submit_u8(cap_m[0], correct_dc=True)
        for k in range(1, frames):
            rx.submit_u8(cap_m[k % 8], correct_dc=True)
            rx.wait()
        rx.wait()
    dt = (time.perf_counter() - t0) / frames
    kt = rx.kernel_times()
    st1 = rx.stats()
    out[m] = {"ms_per_frame": round(dt * 1e3, 4), "dc_blocks": int(st1["dc_blocks"] - st0["dc_blocks"]),
              "redone_sequentially": int(st1["dc_fallback_blocks"] - st0["dc_fallback_blocks"]), "kernels_ms_per_launch": {k: round(v["ms"] / v["launches"], 4) for k, v in kt.items()}}
    rx.close()
print(json.dumps(out))
