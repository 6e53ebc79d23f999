"""VFO-tree descriptors and the reference's configuration contract.

Host-side logic only (no arithmetic of the hot path).  Restates

* the INI -> VFO-tree rules of ``MainWindow::MainWindow`` (mainwindow.cpp:27-233: buffer
  split 65-81, main VFOs 98-138, sub VFOs 141-233), so that a shipped profile yields exactly
  the per-VFO parameters the reference would hand to ``vfo::setFs/setDecimationCount/
  setMixerFreq/.../init``;
* the BASELINE.json workload topologies.

A :class:`VfoDesc` is 1:1 with ``sdrx_vfo_desc`` in include/sdrx.h.
"""
from __future__ import annotations

import math
import re
from dataclasses import dataclass, field, replace

import numpy as np

SUPPORTED_RATES = (288000, 1536000, 1920000)  # mainwindow.h:29


@dataclass
class VfoDesc:
    """One VFO node.  Field <-> reference setter (vfo.h:21-38):

    fs <-> setFs, decimate_count <-> setDecimationCount, mixer_freq <-> setMixerFreq,
    demod_usb <-> setDemodUSB, filter_bw <-> setFilterBandwidth, gain <-> setGain,
    cstyle <-> setCompressonStyle, scalecomp <-> setScaleComp, topic <-> setZmqTopic,
    samples_per_buffer / late_decimate <-> init(samplesPerBuffer, bind, lateDecimate),
    parent <-> setVFOs on the parent (-1: fed by the raw stream, sdrj.cpp:288-294).
    """

    topic: str = ""
    parent: int = -1
    fs: int = 0
    decimate_count: int = 0
    mixer_freq: float = 0.0
    demod_usb: bool = True
    late_decimate: int = 0
    filter_bw: int = 0
    gain: float = 0.01
    cstyle: int = 0
    scalecomp: int = 1
    samples_per_buffer: int = 0

    @property
    def out_rate_stage(self) -> int:
        """vfo::getOutRate (vfo.cpp:212-217): rate after the half-band cascade."""
        return int(self.fs / (2 ** self.decimate_count))

    @property
    def output_rate(self) -> int:
        """outputRate (vfo.cpp:66-102): after the optional late /5 or /6."""
        r = self.out_rate_stage
        if self.demod_usb and self.late_decimate > 0:
            r //= self.late_decimate
        return r

    @property
    def n_stage_out(self) -> int:
        return int(self.samples_per_buffer / (2 ** self.decimate_count))

    @property
    def n_out(self) -> int:
        n = self.n_stage_out
        if self.demod_usb and self.late_decimate > 0:
            n //= self.late_decimate
        return n


@dataclass
class Topology:
    """A receiver profile: raw-stream parameters + the VFO nodes in creation order
    (mains first, then subs in INI order; publish order = main order x sub order,
    vfo.cpp:257-263)."""

    fs: int
    frame: int  # complex samples per raw frame = buflen/2
    bufsplit: int = 4
    center_frequency: int = 0
    correct_dc: bool = False
    zmq_address: str = ""
    vfos: list[VfoDesc] = field(default_factory=list)
    name: str = ""

    def _child_map(self) -> dict[int, list[int]]:
        """parent index -> child indices in creation order, rebuilt when the node list changed
        (one pass instead of one per query: config 5 has 65 538 nodes)."""
        key = (id(self.vfos), len(self.vfos))  # nodes are appended, never re-parented in place
        cached = self.__dict__.get("_cm")
        if cached is None or cached[0] != key:
            m: dict[int, list[int]] = {}
            for i, v in enumerate(self.vfos):
                m.setdefault(max(v.parent, -1), []).append(i)
            cached = (key, m)
            self.__dict__["_cm"] = cached
        return cached[1]

    def children(self, idx: int) -> list[int]:
        return list(self._child_map().get(idx, ()))

    def roots(self) -> list[int]:
        return [i for i, v in enumerate(self.vfos) if v.parent < 0]

    def is_leaf(self, idx: int) -> bool:
        return idx not in self._child_map()

    def leaves_in_publish_order(self) -> list[int]:
        out: list[int] = []

        def walk(i):
            ch = self.children(i)
            if ch:
                for c in ch:
                    walk(c)
            else:
                out.append(i)

        for r in self.roots():
            walk(r)
        return out

    def algorithmic_bytes_per_frame(self) -> int:
        """per VFO per frame 8*n_in (cf32 consumed) + W_out, with
        W_out = 2*n_out for an int16 leaf, 8*n_in/2^d for a VFO that writes its decimated
        cf32 stream for children, n (or 2n) bytes for a compress() leaf."""
        total = 0
        for i, v in enumerate(self.vfos):
            total += 8 * v.samples_per_buffer
            if self.children(i):
                total += 8 * v.n_stage_out
            elif v.demod_usb:
                total += 2 * v.n_out
            else:
                total += v.n_stage_out * (1 if v.cstyle == 1 else 2)
        return total

    def vfo_samples_per_frame(self) -> int:
        return sum(v.samples_per_buffer for v in self.vfos)


# ----------------------------------------------------------------------------- INI front door
def parse_ini(text: str) -> dict[str, str]:
    """The subset of QSettings::IniFormat the shipped profiles use: ``[section]`` headers,
    ``key=value`` with surrounding blanks trimmed, ``N\\key`` array members (the backslash is
    the group separator), keys before any section in the top level, ``;`` comment lines.  A
    leading ``#`` is NOT a comment for QSettings -- such a line just defines a key nobody
    reads (sdr_25E.ini:5-9) -- which is what happens here too."""
    out: dict[str, str] = {}
    section = ""
    for raw in text.splitlines():
        line = raw.strip()
        if not line or line.startswith(";"):
            continue
        m = re.match(r"^\[(.*)\]$", line)
        if m:
            section = m.group(1).strip()
            if section.lower() == "general":
                section = ""
            continue
        if "=" not in line:
            continue
        k, v = line.split("=", 1)
        k = k.strip().replace("\\", "/")
        v = v.strip()
        if len(v) >= 2 and v[0] == '"' and v[-1] == '"':
            v = v[1:-1]
        out[(section + "/" if section else "") + k] = v
    return out


def _to_int(s: str | None) -> int:
    """QVariant(QString).toInt(): 0 when missing or not an integer literal."""
    if s is None:
        return 0
    try:
        v = int(s.strip(), 10)
    except ValueError:
        return 0
    return v if -(2 ** 31) <= v < 2 ** 31 else 0


def _to_float32(s: str | None) -> np.float32:
    if s is None:
        return np.float32(0)
    try:
        return np.float32(float(s.strip()))
    except ValueError:
        return np.float32(0)


def _ilog2(x) -> int:
    return int(math.log2(x))


def topology_from_ini(text: str, name: str = "") -> Topology:
    """mainwindow.cpp:27-233 on an INI text."""
    kv = parse_ini(text)
    fs = _to_int(kv.get("sample_rate"))
    if fs == 0:
        raise ValueError("sample_rate ini file key not found or equal to zero")  # mainwindow.cpp:31-37
    if fs not in SUPPORTED_RATES:
        raise ValueError(f"sample_rate {fs} not supported, only {SUPPORTED_RATES}")  # 39-47
    center = _to_int(kv.get("center_frequency"))
    mix_offset = _to_int(kv.get("mix_offset"))
    # "usually 4 buffers per Fs but in some cases 5 due to multiple of 512", 65-80
    if ((2 * fs) // 4) % 512 > 0:
        buflen, bufsplit = (2 * fs) // 5, 5
    else:
        buflen, bufsplit = (2 * fs) // 4, 4
    topo = Topology(fs=fs, frame=buflen // 2, bufsplit=bufsplit, center_frequency=center,
                    correct_dc=kv.get("correct_dc_bias", "") == "1",
                    zmq_address=kv.get("zmq_address", ""), name=name)

    mains: list[int] = []
    for i in range(1, _to_int(kv.get("main_vfos/size")) + 1):  # 98-138
        p = f"main_vfos/{i}/"
        vfo_freq = _to_int(kv.get(p + "frequency"))
        out_rate = _to_int(kv.get(p + "out_rate"))
        if out_rate <= 0:
            raise ValueError(f"main_vfos/{i}: out_rate missing")
        d = 0 if fs // out_rate == 1 else _ilog2(fs // out_rate)
        desc = VfoDesc(parent=-1, fs=fs, decimate_count=d, mixer_freq=float(center - vfo_freq),
                       demod_usb=False, cstyle=1, samples_per_buffer=buflen // 2)
        compscale = _to_int(kv.get(p + "compress_scale"))
        if compscale > 0:
            desc.scalecomp = compscale
        addr, top = kv.get(p + "zmq_address", ""), kv.get(p + "zmq_topic", "")
        if addr != "" and top != "":
            desc.topic = top
        mains.append(len(topo.vfos))
        topo.vfos.append(desc)

    for i in range(1, _to_int(kv.get("vfos/size")) + 1):  # 141-233
        p = f"vfos/{i}/"
        vfo_freq = _to_int(kv.get(p + "frequency")) + mix_offset
        data_rate = _to_int(kv.get(p + "data_rate"))
        out_rate = _to_int(kv.get(p + "out_rate"))
        if out_rate == 0 and data_rate > 0:
            out_rate = {600: 12000, 1200: 24000}.get(data_rate, 48000)
        if out_rate <= 0:
            raise ValueError(f"vfos/{i}: neither out_rate nor data_rate given")
        filterbw = _to_int(kv.get(p + "filter_bandwidth"))
        main_idx, main_vfo_freq, main_out = 0, 0, fs
        for a, mi in enumerate(mains):  # first main whose band covers the VFO, 179-191
            m = topo.vfos[mi]
            diff = abs((center - int(m.mixer_freq)) - vfo_freq)
            if diff < m.out_rate_stage and not m.demod_usb:
                main_idx, main_vfo_freq, main_out = a, int(m.mixer_freq), m.out_rate_stage
                break
        late = 0
        if main_out // 48000 == 5:  # 196-216
            d, late = _ilog2(main_out // (5 * out_rate)), 5
        elif main_out // 48000 == 6:
            d, late = _ilog2(main_out // (6 * out_rate)), 6
        else:
            d = _ilog2(fs // out_rate) - _ilog2(fs // main_out)
        gain = float(_to_float32(kv.get(p + "gain")) / np.float32(100))
        if not mains:
            raise ValueError("profile has sub VFOs but no main VFO")
        topo.vfos.append(VfoDesc(
            topic=kv.get(p + "topic", ""), parent=mains[main_idx], fs=main_out, decimate_count=d,
            mixer_freq=float((center - main_vfo_freq) - vfo_freq), demod_usb=True, late_decimate=late,
            filter_bw=filterbw, gain=gain, cstyle=1, samples_per_buffer=main_out // bufsplit))
    return topo


# ----------------------------------------------------------------------------- BASELINE configs
# Derived parameters of sample_ini/sdr_25E.ini (reproduced from the INI by
# tests/test_topology.py where /root/reference is present).
_25E_MAIN = [(484000, 2), (-496000, 3)]
_25E_SUBS_MAIN0 = [  # (topic, mixer, d, filter_bw)
    ("VFO01", 110854, 5, 4000), ("VFO02", -98573, 5, 0), ("VFO03", -103706, 5, 0),
    ("VFO04", -108996, 5, 0), ("VFO05", 1866, 5, 0), ("VFO06", -3063, 5, 0), ("VFO07", -8261, 4, 0),
    ("VFO08", -13563, 5, 0), ("VFO09", -43288, 5, 0), ("VFO10", -48682, 5, 0),
    ("VFO11", -67905, 5, 0), ("VFO12", -73244, 5, 0)]
_25E_SUBS_MAIN1 = [  # (topic, mixer, d, filter_bw, gain)
    ("VFO13", 90700, 2, 0, 0.05), ("VFO14", 76200, 2, 0, 0.05), ("VFO15", 61300, 2, 0, 0.05),
    ("VFO16", 11400, 2, 0, 0.05), ("VFO17", -3900, 2, 0, 0.05), ("VFO18", -18200, 2, 0, 0.05),
    ("VFO19", -41300, 2, 10000, 0.05), ("VFO20", -46500, 2, 10000, 0.03),
    ("VFO21", -51700, 2, 10000, 0.03), ("VFO22", -56300, 2, 10000, 0.03),
    ("VFO23", -61500, 2, 10000, 0.03), ("VFO24", -66600, 2, 10000, 0.03),
    ("VFO25", -72300, 2, 10000, 0.03), ("VFO26", -77430, 2, 10000, 0.03),
    ("VFO27", -82430, 2, 10000, 0.03)]


def _g(x) -> float:
    """A gain as the reference holds it: a float32."""
    return float(np.float32(x))


def _gain_pct(pct) -> float:
    return float(np.float32(pct) / np.float32(100))  # mainwindow.cpp:219


def _spread_mixer(k: int, K: int, rate: int) -> int:
    """Config-3 rule for synthetic sub VFOs: spread over 80 % of the parent band."""
    return int(round((k + 0.5 - K / 2) * 0.8 * rate / K)) + 37


def _mains_25e(topo: Topology) -> list[int]:
    idx = []
    for mixer, d in _25E_MAIN:
        idx.append(len(topo.vfos))
        topo.vfos.append(VfoDesc(parent=-1, fs=1536000, decimate_count=d, mixer_freq=float(mixer),
                                 demod_usb=False, cstyle=1, samples_per_buffer=384000))
    return idx


def profile_25e(n_extra_main1: int = 0) -> Topology:
    """sample_ini/sdr_25E.ini: 2 mains + 27 subs (+ optional synthetic subs on main1)."""
    t = Topology(fs=1536000, frame=384000, bufsplit=4, center_frequency=1545600000, correct_dc=True,
                 zmq_address="tcp://*:6003", name="sdr_25E")
    m0, m1 = _mains_25e(t)
    for topic, mixer, d, bw in _25E_SUBS_MAIN0:
        t.vfos.append(VfoDesc(topic=topic, parent=m0, fs=384000, decimate_count=d, mixer_freq=float(mixer),
                              filter_bw=bw, gain=_gain_pct(5), cstyle=1, samples_per_buffer=96000))
    for topic, mixer, d, bw, g in _25E_SUBS_MAIN1:
        t.vfos.append(VfoDesc(topic=topic, parent=m1, fs=192000, decimate_count=d, mixer_freq=float(mixer),
                              filter_bw=bw, gain=_gain_pct(round(g * 100)), cstyle=1, samples_per_buffer=48000))
    for k in range(n_extra_main1):
        t.vfos.append(VfoDesc(topic=f"X{k:04d}"[:5], parent=m1, fs=192000, decimate_count=2,
                              mixer_freq=float(_spread_mixer(k, max(n_extra_main1, 1), 192000)),
                              filter_bw=10000 if k % 2 else 0, gain=_g(0.05), cstyle=1,
                              samples_per_buffer=48000))
    return t


def config1() -> Topology:
    """1 main VFO + 1 sub VFO of sdr_25E (BASELINE config 1)."""
    t = Topology(fs=1536000, frame=384000, bufsplit=4, center_frequency=1545600000, correct_dc=True,
                 name="config1")
    t.vfos.append(VfoDesc(parent=-1, fs=1536000, decimate_count=2, mixer_freq=484000.0, demod_usb=False,
                          cstyle=1, samples_per_buffer=384000))
    t.vfos.append(VfoDesc(topic="VFO01", parent=0, fs=384000, decimate_count=5, mixer_freq=110854.0,
                          filter_bw=4000, gain=_gain_pct(5), cstyle=1, samples_per_buffer=96000))
    return t


def config2() -> Topology:
    """32 sub VFOs across the two 25E mains (27 from the INI + 5 synthetic on main1)."""
    t = profile_25e(n_extra_main1=5)
    t.name = "config2"
    return t


def config3(n_subs: int = 1024) -> Topology:
    """n_subs sub VFOs, half per 25E main: main0 subs d=5 (12 k), main1 subs d=2 (48 k), every
    2nd main1 sub with the 10 kHz low-pass, gain 0.05."""
    t = Topology(fs=1536000, frame=384000, bufsplit=4, center_frequency=1545600000, name=f"config3-{n_subs}")
    m0, m1 = _mains_25e(t)
    K = n_subs // 2
    for k in range(K):
        t.vfos.append(VfoDesc(topic=f"A{k:04d}"[:5], parent=m0, fs=384000, decimate_count=5,
                              mixer_freq=float(_spread_mixer(k, K, 384000)), gain=_g(0.05), cstyle=1,
                              samples_per_buffer=96000))
    for k in range(n_subs - K):
        t.vfos.append(VfoDesc(topic=f"B{k:04d}"[:5], parent=m1, fs=192000, decimate_count=2,
                              mixer_freq=float(_spread_mixer(k, n_subs - K, 192000)),
                              filter_bw=10000 if k % 2 else 0, gain=_g(0.05), cstyle=1,
                              samples_per_buffer=48000))
    return t


def config3_flat(n_vfos: int = 1024) -> Topology:
    """Flat variant: leaf VFOs directly on the 1.536 MS/s stream, d=5 -> 48 kHz, no parent."""
    t = Topology(fs=1536000, frame=384000, bufsplit=4, name=f"flat-{n_vfos}")
    for k in range(n_vfos):
        t.vfos.append(VfoDesc(topic=f"F{k:04d}"[:5], parent=-1, fs=1536000, decimate_count=5,
                              mixer_freq=float(_spread_mixer(k, n_vfos, 1536000)), gain=_g(0.05), cstyle=1,
                              samples_per_buffer=384000))
    return t


def config4(n_subs: int = 256) -> Topology:
    """sdr_54W_all style: Fs 1.92 MS/s, 3 mains d=3 -> 240 k, subs d=0 with late /5 (49 taps) ->
    48 k and the 10 kHz low-pass (47 taps), gain 0.04."""
    t = Topology(fs=1920000, frame=480000, bufsplit=4, center_frequency=1545939000, name=f"config4-{n_subs}")
    mains = []
    for mixer in (819000, -181000, -911000):
        mains.append(len(t.vfos))
        t.vfos.append(VfoDesc(parent=-1, fs=1920000, decimate_count=3, mixer_freq=float(mixer),
                              demod_usb=False, cstyle=1, samples_per_buffer=480000))
    base, rem = divmod(n_subs, 3)
    for mi, m in enumerate(mains):
        K = base + (1 if mi < rem else 0)
        for k in range(K):
            t.vfos.append(VfoDesc(topic=f"{'CDE'[mi]}{k:04d}"[:5], parent=m, fs=240000, decimate_count=0,
                                  mixer_freq=float(_spread_mixer(k, K, 240000)), late_decimate=5,
                                  filter_bw=10000, gain=_g(0.04), cstyle=1, samples_per_buffer=60000))
    return t


def config5(n_subs: int = 65536) -> Topology:
    t = config3(n_subs)
    t.name = f"config5-{n_subs}"
    return t


def shard(topo: Topology, rank: int, world: int) -> Topology:
    """Static block partition of the sub VFOs of every main across `world` GPUs; a main is
    replicated on every rank that holds at least one of its subs.  The unit that
    moves is a SUBTREE: a sub VFO goes to its rank with everything below it.  A main whose block
    on this rank is empty (fewer subs than ranks) is NOT kept -- alone it would look like a
    childless main, i.e. a compress() leaf that publishes IQ, which the reference never does for
    a main that has subs (vfo.cpp:253-266).  Parent-less leaves are block-partitioned among
    themselves.  Node order, hence publish order within the shard, is preserved; a rank can end
    up with no VFOs at all (`len(shard.vfos) == 0`): it then only takes part in the broadcast."""
    if world <= 1:
        return topo
    cm = topo._child_map()
    keep: list[int] = []

    def subtree(i):
        stack = [i]
        while stack:
            j = stack.pop()
            keep.append(j)
            stack.extend(cm.get(j, ()))

    roots = cm.get(-1, [])
    for r in roots:
        ch = cm.get(r, [])
        if ch:
            lo, hi = (len(ch) * rank) // world, (len(ch) * (rank + 1)) // world
            if hi > lo:
                keep.append(r)
                for c in ch[lo:hi]:
                    subtree(c)
    flat = [r for r in roots if not cm.get(r)]
    lo, hi = (len(flat) * rank) // world, (len(flat) * (rank + 1)) // world
    keep.extend(flat[lo:hi])
    keep.sort()
    remap = {old: new for new, old in enumerate(keep)}
    vfos = [replace(topo.vfos[i], parent=remap.get(topo.vfos[i].parent, -1)) for i in keep]
    return Topology(fs=topo.fs, frame=topo.frame, bufsplit=topo.bufsplit,
                    center_frequency=topo.center_frequency, correct_dc=topo.correct_dc,
                    zmq_address=topo.zmq_address, vfos=vfos, name=f"{topo.name}[{rank}/{world}]")


def subset(topo: Topology, leaves) -> tuple[Topology, dict[int, int]]:
    """The VFOs `leaves` (indices into topo.vfos) with every VFO above them as a tree of its own, node order kept; returns
    it with the map old index -> new index.  A VFO's output depends on its ancestors only -- every child is handed the same
    read-only buffer (vfo.cpp:253-264) -- so the subset computes exactly what the same VFOs compute inside the full tree
    (what the sharding above rests on; bench.py and the full-size tests run the oracle on such subsets)."""
    keep = set()
    for i in leaves:
        j = int(i)
        while j >= 0 and j not in keep:
            keep.add(j)
            j = topo.vfos[j].parent
    order = sorted(keep)
    remap = {old: new for new, old in enumerate(order)}
    vfos = [replace(topo.vfos[i], parent=remap.get(topo.vfos[i].parent, -1)) for i in order]
    return (Topology(fs=topo.fs, frame=topo.frame, bufsplit=topo.bufsplit, center_frequency=topo.center_frequency,
                     correct_dc=topo.correct_dc, zmq_address=topo.zmq_address, vfos=vfos, name=f"{topo.name}[subset {len(order)}]"),
            remap)


def sample_leaves(topo: Topology, n_random: int, seed: int, world: int = 8) -> list[int]:
    """A seeded choice of publishing leaves for a parity check at sizes the CPU oracle cannot cover whole: for every group
    of siblings the first and the last leaf of each of the `world` blocks `shard` would make (where an off-by-one of a work
    list or of the partition shows first) plus `n_random` leaves drawn over all groups."""
    rng = np.random.default_rng(seed)
    cm = topo._child_map()
    groups = [[i for i in ch if i not in cm] for ch in cm.values()]
    groups = [g for g in groups if g]
    picked: set[int] = set()
    for g in groups:
        for r in range(world):
            lo, hi = (len(g) * r) // world, (len(g) * (r + 1)) // world
            if hi > lo:
                picked.update((g[lo], g[hi - 1]))
    total = sum(len(g) for g in groups)
    for g in groups:
        k = min(len(g), max(1, round(n_random * len(g) / max(total, 1))))
        picked.update(int(x) for x in rng.choice(g, size=k, replace=False))
    return sorted(picked)
