"""k_usb_demod time by VFO kind: 1024 d=2 subs (192 k -> 48 k), all with / all without the 10 kHz low-pass."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdrreceiver_amd import topology as tp
from class_probe import run, only
full = tp.config3(2048)
for name, bw in (("no-lpf", 0), ("lpf", 10000)):
    t = only(full, lambda v: v.decimate_count == 2)
    for v in t.vfos:
        if v.parent >= 0:
            v.filter_bw = bw
    print(name, json.dumps(run(t)), flush=True)
