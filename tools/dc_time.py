"""tools/dc_time.py -- GPU time of the exact DC-bias removal (u8 ingest + sdrj.cpp:277-283 on the device) per frame:
HIP-event bracket around its launches (sdrx_enable_kernel_timing), config 1's tree, 12 frames of random dongle bytes."""
import json
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sdrreceiver_amd import topology as tp
from sdrreceiver_amd.receiver import Receiver

topo = tp.config1()
rx = Receiver.from_topology(topo)
rx.set_publish(False)
rng = np.random.default_rng(3)
frames = [rng.integers(0, 256, 2 * topo.frame, dtype=np.uint8) for _ in range(4)]
for f in range(4):
    rx.process_u8(frames[f % 4], correct_dc=True)
rx.enable_kernel_timing(True)
for f in range(12):
    rx.process_u8(frames[f % 4], correct_dc=True)
t = rx.kernel_times()
print(json.dumps({k: round(v["ms"] / v["launches"], 4) for k, v in t.items()}))
