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


# ---- the reference AS SHIPPED (-Ofast, SDRReceiver.pro:74-75): tests/golden/ofast_*.npz ------------------------------
OFAST_FIXTURES = {"ofast_config1.npz": "config1", "ofast_profile_25e.npz": "profile_25e", "ofast_54w.npz": "54w"}
OFAST_REL_TOL = 1e-5  # north_star: "within 1e-5 relative float tolerance"


def check_against_ofast_fixture(g, topo, f, stream_of, payload_of, o2_payload_of=None):
    """Frame `f` of an implementation (`stream_of(i)` -> complex64 or None, `payload_of(i)` -> int16) against the -Ofast
    fixture `g`: every final complex stream within 1e-5 of max|ref| at the stored positions (head + every 128th sample),
    every int16 payload within +-1 LSB of the shipped build's.  The fixture holds that payload as a patch against the -O2
    build's: `o2_payload_of(i)` must return the -O2 payload bit for bit (default: the implementation itself, i.e. it
    claims to BE the -O2 result) -- patched, its sha must be the -Ofast payload's, which proves the reconstruction and,
    for the default, that the implementation equals the shipped build everywhere outside the patch.
    Returns (worst relative stream error, patched samples, payload samples)."""
    worst, patched, total = 0.0, 0, 0
    for i, v in enumerate(topo.vfos):
        z = stream_of(i)
        if z is not None:
            scale = float(g[f"f{f}_v{i}_stream_absmax"])
            e = max(float(np.abs(z[:256] - g[f"f{f}_v{i}_stream_head"]).max()), float(np.abs(z[::128] - g[f"f{f}_v{i}_stream_every128"]).max()))
            assert e <= OFAST_REL_TOL * scale, (f, i, "stream", e / scale)
            worst = max(worst, e / scale)
        if not topo.children(i) and v.demod_usb:
            got = payload_of(i)
            shipped = (o2_payload_of(i) if o2_payload_of else got).copy()
            idx, val = g[f"f{f}_v{i}_pay_idx"], g[f"f{f}_v{i}_pay_val"]
            shipped[idx] = val
            assert sha(shipped) == str(g[f"f{f}_v{i}_pay_sha"]), (f, i, "not the -Ofast payload: the -O2 payload it was rebuilt from is not bit-exact")
            assert np.abs(got.astype(np.int32) - shipped.astype(np.int32)).max() <= 1, (f, i, "payload beyond 1 LSB of the shipped build")
            patched += idx.size
            total += got.size
    return worst, patched, total
