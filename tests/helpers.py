"""Shared helpers for the test-suite (test infrastructure)."""
from __future__ import annotations

import hashlib
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
REFERENCE_ROOT = "/root/reference"


def sha(a) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view({4: np.uint32, 8: np.uint64, 2: np.uint16, 1: np.uint8}[a.dtype.itemsize])


def topo_54w_golden():
    """The tree tests/golden/make_golden.py used for profile_54w.npz."""
    from sdrreceiver_amd import topology as tp
    t = tp.config4(6)
    t.vfos.append(tp.VfoDesc(topic="VFO41", parent=0, fs=240000, decimate_count=2, mixer_freq=105571.0,
                             late_decimate=5, filter_bw=0, gain=tp._gain_pct(4), cstyle=1,
                             samples_per_buffer=60000))
    t.vfos.append(tp.VfoDesc(topic="VFO44", parent=0, fs=240000, decimate_count=2, mixer_freq=-74731.0,
                             late_decimate=5, filter_bw=4000, gain=tp._gain_pct(4), cstyle=1,
                             samples_per_buffer=60000))
    return t


def topo_compress_golden():
    from sdrreceiver_amd import topology as tp
    t = tp.Topology(fs=1536000, frame=384000, name="compress")
    for cs, sc, top in ((1, 1, "IQ4A"), (1, 16, "IQ4B"), (0, 1, "IQ8")):
        t.vfos.append(tp.VfoDesc(topic=top, parent=-1, fs=1536000, decimate_count=3, mixer_freq=-496000.0,
                                 demod_usb=False, cstyle=cs, scalecomp=sc, samples_per_buffer=384000))
    return t


def topo_288k_golden():
    from sdrreceiver_amd import topology as tp
    t = tp.Topology(fs=288000, frame=57600, bufsplit=5, name="288k")
    t.vfos.append(tp.VfoDesc(parent=-1, fs=288000, decimate_count=0, mixer_freq=0.0, demod_usb=False, cstyle=1,
                             samples_per_buffer=57600))
    t.vfos.append(tp.VfoDesc(topic="VFO51", parent=0, fs=288000, decimate_count=0, mixer_freq=54578.0,
                             late_decimate=6, filter_bw=10000, gain=tp._gain_pct(4), cstyle=1,
                             samples_per_buffer=57600))
    return t


GOLDEN_TREES = {
    # fixture file -> (topology factory, frames)
    "config1.npz": ("config1", 6),
    "profile_25e.npz": ("profile_25e", 5),
    "profile_54w.npz": ("54w", 5),
    "compress.npz": ("compress", 2),
    "profile_288k.npz": ("288k", 6),
}


def golden_topology(key):
    from sdrreceiver_amd import topology as tp
    return {"config1": tp.config1, "profile_25e": tp.profile_25e, "54w": topo_54w_golden,
            "compress": topo_compress_golden, "288k": topo_288k_golden}[key]()
