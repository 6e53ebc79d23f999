"""Host-side mirror of the reference interface for the VFO chain, over the C ABI.

Two layers:

* :class:`Receiver` -- thin object wrapper around ``sdrx_*`` (one context = one VFO tree).
* :class:`vfo` and :class:`sdrj` -- the reference's own class and method names (vfo.h:16-49,
  sdrj.h:29-48) so that host code and parity tests read like reference-side code:
  ``v = vfo(); v.setFs(...); v.setMixerFreq(...); v.init(n, True); main.setVFOs([...]);
  radio = sdrj(); radio.setVFOs([main]); radio.demodData(data, len)``.

Everything here is plumbing; the arithmetic runs in the HIP kernels of libsdrx.so.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .topology import Topology, VfoDesc


class SdrxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"sdrx error {code}: {msg}")
        self.code = code


def arithmetic(exact) -> int:
    """Option "exact" of include/sdrx.h: True / 1 / "exact" -> 1 (bit-identical to the -O2 reference), False / 0 / "tolerance" -> 0
    (NCO as rotations of its exact checkpoints, FMA mixer and filters), 2 / "robust" -> 2 (exact NCO, FMA mixer and filters)."""
    if isinstance(exact, str):
        return {"exact": 1, "tolerance": 0, "fast": 0, "robust": 2}[exact]
    return 2 if (exact is not True and exact == 2) else int(bool(exact))


class Receiver:
    """One libsdrx context: a VFO tree on one GPU."""

    def __init__(self, device: int = 0, exact: bool = True, keep_prequant: bool = False, segments: int = 0,
                 dc_blocked_scan: bool = False, pipeline: bool = False, fuse: bool = True, frame_pipeline: bool = True,
                 fuse_late: bool = True, keep_streams: bool = False, dc_speculative: bool = True,
                 dc_blocks_per_step: int | None = None, fuse_demod: bool = False):
        self.L = _lib.lib()
        h = C.c_void_p()
        rc = self.L.sdrx_create(C.byref(h), int(device))
        if rc != 0:
            raise SdrxError(rc, self.L.sdrx_last_error(None).decode())
        self.h = h
        self.descs: list[VfoDesc] = []
        self.published: list[tuple[bytes, int, bytes]] = []
        self._cb = _lib.PUBLISH_FN(self._on_publish)
        self._chk(self.L.sdrx_set_publish_callback(self.h, self._cb, None))
        self._chk(self.L.sdrx_set_option(self.h, b"exact", arithmetic(exact)))
        self._chk(self.L.sdrx_set_option(self.h, b"keep_prequant", int(bool(keep_prequant))))
        self._chk(self.L.sdrx_set_option(self.h, b"segments", int(segments)))
        self._chk(self.L.sdrx_set_option(self.h, b"dc_blocked_scan", int(bool(dc_blocked_scan))))
        self._chk(self.L.sdrx_set_option(self.h, b"pipeline", int(bool(pipeline))))
        self._chk(self.L.sdrx_set_option(self.h, b"fuse", int(bool(fuse))))
        self._chk(self.L.sdrx_set_option(self.h, b"frame_pipeline", int(bool(frame_pipeline))))
        self._chk(self.L.sdrx_set_option(self.h, b"fuse_late", int(bool(fuse_late))))
        self._chk(self.L.sdrx_set_option(self.h, b"keep_streams", int(bool(keep_streams))))
        self._chk(self.L.sdrx_set_option(self.h, b"fuse_demod", int(bool(fuse_demod))))
        self._chk(self.L.sdrx_set_option(self.h, b"dc_speculative", int(bool(dc_speculative))))
        if dc_blocks_per_step is not None:
            self._chk(self.L.sdrx_set_option(self.h, b"dc_blocks_per_step", int(dc_blocks_per_step)))
        self.finalized = False

    # -- plumbing -----------------------------------------------------------------
    def _chk(self, rc):
        if rc != 0:
            raise SdrxError(rc, self.L.sdrx_last_error(self.h).decode())

    def _on_publish(self, user, topic, rate, buf, length):
        self.published.append((C.string_at(topic, 5), int(rate), C.string_at(buf, length)))

    def set_publish(self, enabled: bool) -> None:
        """Install / remove the per-leaf publish callback (a ctypes callback costs microseconds per
        leaf: a C or C++ host pays nanoseconds)."""
        self._chk(self.L.sdrx_set_publish_callback(self.h, self._cb if enabled else _lib.PUBLISH_FN(0), None))

    def close(self):
        if getattr(self, "h", None):
            self.L.sdrx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- configuration --------------------------------------------------------------
    def add_vfo(self, d: VfoDesc) -> int:
        c = _lib.desc_to_c(d)
        out = C.c_int(-1)
        self._chk(self.L.sdrx_add_vfo(self.h, C.byref(c), C.byref(out)))
        self.descs.append(d)
        return out.value

    def finalize(self):
        self._chk(self.L.sdrx_finalize(self.h))
        self.finalized = True

    @classmethod
    def from_topology(cls, topo: Topology, **kw) -> "Receiver":
        r = cls(**kw)
        for d in topo.vfos:
            r.add_vfo(d)
        r.finalize()
        return r

    # -- per frame ----------------------------------------------------------------------
    def process(self, iq) -> None:
        """One frame of interleaved float32 I/Q on the host (sdrj::demodData's argument)."""
        iq = np.ascontiguousarray(iq, dtype=np.float32).reshape(-1)
        self.published.clear()
        self._chk(self.L.sdrx_process(self.h, iq.ctypes.data, iq.size // 2))

    def process_u8(self, iq_bytes, correct_dc: bool = False) -> None:
        """One frame of raw dongle bytes (I,Q unsigned, offset 127), LUT + optional DC-bias IIR on the
        device (sdrj::readyRead + demodData, sdrj.cpp:149-165,271-286)."""
        b = np.ascontiguousarray(iq_bytes, dtype=np.uint8).reshape(-1)
        self.published.clear()
        self._chk(self.L.sdrx_process_u8(self.h, b.ctypes.data, b.size // 2, int(bool(correct_dc))))

    def process_device(self, dev_ptr: int, n_complex: int) -> None:
        self.published.clear()  # the payloads of this frame arrive with the next (explicit or implicit) fetch
        self._chk(self.L.sdrx_process_device(self.h, C.c_void_p(dev_ptr), int(n_complex)))

    def fetch(self) -> None:
        self.published.clear()
        self._chk(self.L.sdrx_fetch(self.h))

    # -- pipelined interface: submit returns at once, wait delivers the oldest undelivered frame ------
    def submit(self, iq) -> None:
        iq = np.ascontiguousarray(iq, dtype=np.float32).reshape(-1)
        self._chk(self.L.sdrx_submit(self.h, iq.ctypes.data, iq.size // 2))

    def submit_u8(self, iq_bytes, correct_dc: bool = False) -> None:
        b = np.ascontiguousarray(iq_bytes, dtype=np.uint8).reshape(-1)
        self._chk(self.L.sdrx_submit_u8(self.h, b.ctypes.data, b.size // 2, int(bool(correct_dc))))

    def submit_device(self, dev_ptr: int, n_complex: int) -> None:
        self._chk(self.L.sdrx_submit_device(self.h, C.c_void_p(dev_ptr), int(n_complex)))

    def process_shared(self, src: "Receiver") -> None:
        """The frame `src` staged last (same device), once more through THIS tree, without a second upload."""
        self.published.clear()
        self._chk(self.L.sdrx_process_shared(self.h, src.h))

    def submit_shared(self, src: "Receiver") -> None:
        self._chk(self.L.sdrx_submit_shared(self.h, src.h))

    def process_if_same(self, src: "Receiver", iq) -> bool:
        """Like process_shared, but only if `iq` IS the frame `src` staged last (the library compares it byte for byte with
        src's pinned staging copy).  False -- and nothing processed -- when it is not."""
        iq = np.ascontiguousarray(iq, dtype=np.float32).reshape(-1)
        self.published.clear()
        rc = self.L.sdrx_process_if_same(self.h, src.h, iq.ctypes.data, iq.size // 2)
        if rc == _lib.SDRX_DIFFERENT:
            return False
        self._chk(rc)
        return True

    def submit_if_same(self, src: "Receiver", iq) -> bool:
        iq = np.ascontiguousarray(iq, dtype=np.float32).reshape(-1)
        rc = self.L.sdrx_submit_if_same(self.h, src.h, iq.ctypes.data, iq.size // 2)
        if rc == _lib.SDRX_DIFFERENT:
            return False
        self._chk(rc)
        return True

    def wait(self) -> None:
        """Blocks until the oldest undelivered frame's payloads are on the host; `published` then
        holds that frame's messages and output() serves it."""
        self.published.clear()
        self._chk(self.L.sdrx_wait(self.h))

    def in_flight(self) -> int:
        return int(self.L.sdrx_in_flight(self.h))

    def sync(self) -> None:
        self._chk(self.L.sdrx_sync(self.h))

    def set_stream(self, hip_stream: int | None) -> None:
        self._chk(self.L.sdrx_set_stream(self.h, C.c_void_p(hip_stream or 0)))

    # -- results ------------------------------------------------------------------------
    def output(self, vid: int) -> np.ndarray:
        buf, ln, rate = C.c_void_p(), C.c_uint32(), C.c_uint32()
        self._chk(self.L.sdrx_get_output(self.h, vid, C.byref(buf), C.byref(ln), C.byref(rate)))
        raw = C.string_at(buf.value, ln.value)
        return np.frombuffer(raw, dtype=np.int16 if self.descs[vid].demod_usb else np.int8).copy()

    def output_rate(self, vid: int) -> int:
        rate = C.c_uint32()
        self._chk(self.L.sdrx_get_output(self.h, vid, None, None, C.byref(rate)))
        return rate.value

    def stream(self, vid: int, missing_ok: bool = False):
        """decimate[decimateCount] of VFO `vid` after the last frame.  A leaf whose late decimation or whose demodulation runs
        inside the mix wave keeps it only while it is the tap (:meth:`set_tap`) or with ``keep_streams``: SdrxError otherwise,
        or None with `missing_ok`."""
        n = C.c_int()
        rc = self.L.sdrx_get_stream(self.h, vid, None, 0, C.byref(n))
        if rc == _lib.SDRX_ENOSTREAM and missing_ok:
            return None
        self._chk(rc)
        out = np.zeros(2 * max(n.value, 1), np.float32)
        self._chk(self.L.sdrx_get_stream(self.h, vid, out.ctypes.data, n.value, C.byref(n)))
        return out[: 2 * n.value].view(np.complex64).copy()

    def set_tap(self, vid: int) -> None:
        """fftVFOSlot: from the next frame on :meth:`stream` serves VFO `vid` whatever its kind; REPLACES the selection
        (-1: nothing selected)."""
        self._chk(self.L.sdrx_set_tap(self.h, int(vid)))

    def add_tap(self, vid: int) -> None:
        """One more tapped VFO (fftVFOSlot sets emitFFT on every VFO whose topic matches, vfo.cpp:492-509)."""
        self._chk(self.L.sdrx_add_tap(self.h, int(vid)))

    def set_taps(self, vids) -> None:
        """The selection becomes exactly `vids`."""
        self.set_tap(-1)
        for v in vids:
            self.add_tap(v)

    def raw(self) -> np.ndarray:
        """The raw frame as the main VFOs consumed it (after the byte LUT / DC-bias removal)."""
        out = np.zeros(2 * self.root_frame(), np.float32)
        n = C.c_int()
        self._chk(self.L.sdrx_get_raw(self.h, out.ctypes.data, out.size // 2, C.byref(n)))
        return out[: 2 * n.value].view(np.complex64).copy()

    def root_frame(self) -> int:
        return max(d.samples_per_buffer for d in self.descs if d.parent < 0)

    def prequant(self, vid: int) -> np.ndarray:
        n = C.c_int()
        self._chk(self.L.sdrx_get_prequant(self.h, vid, None, 0, C.byref(n)))
        out = np.zeros(max(n.value, 1), np.float32)
        self._chk(self.L.sdrx_get_prequant(self.h, vid, out.ctypes.data, n.value, C.byref(n)))
        return out[: n.value].copy()

    def taps(self, vid: int, which: str) -> np.ndarray:
        w = {"fir_usb": 0, "fir_dec": 1, "hilbert": 2}[which]
        n = C.c_int()
        self._chk(self.L.sdrx_get_taps(self.h, vid, w, None, 0, C.byref(n)))
        out = np.zeros(max(n.value, 1), np.float32)
        self._chk(self.L.sdrx_get_taps(self.h, vid, w, out.ctypes.data, n.value, C.byref(n)))
        return out[: n.value].copy()

    def nco(self, vid: int, first: int, count: int) -> np.ndarray:
        out = np.zeros(2 * max(count, 1), np.float32)
        self._chk(self.L.sdrx_get_nco(self.h, vid, first, count, out.ctypes.data))
        return out[: 2 * count].view(np.complex64).copy()

    # -- measurement -----------------------------------------------------------------------
    def stats(self) -> dict:
        s = _lib.StatsC()
        self._chk(self.L.sdrx_get_stats(self.h, C.byref(s)))
        return {k: getattr(s, k) for k, _ in s._fields_}

    def enable_kernel_timing(self, on: bool) -> None:
        self._chk(self.L.sdrx_enable_kernel_timing(self.h, int(on)))

    def kernel_times(self) -> dict:
        ms = (C.c_double * _lib.NKERNELS)()
        n = (C.c_int64 * _lib.NKERNELS)()
        b = (C.c_int64 * _lib.NKERNELS)()
        self._chk(self.L.sdrx_get_kernel_times(self.h, ms, n, b))
        return {self.L.sdrx_kernel_name(k).decode(): {"ms": ms[k], "launches": n[k], "alg_bytes": b[k]}
                for k in range(_lib.NKERNELS) if n[k]}


class Group:
    """One VFO tree on several GPUs from this one process (``sdrx_group_*``): one context per entry of
    `devices` (a device may be named twice: two shards on one GPU), sub VFOs block-partitioned per
    main VFO, the raw frame fanned out from the first device by peer-to-peer copies.  Ids are those of
    the whole tree; `published` holds the last delivered frame's messages in the reference's order."""

    def __init__(self, devices, exact: bool = True, **options):
        self.L = _lib.lib()
        h = C.c_void_p()
        arr = (C.c_int * len(devices))(*[int(d) for d in devices])
        rc = self.L.sdrx_group_create(C.byref(h), arr, len(devices))
        if rc != 0:
            raise SdrxError(rc, self.L.sdrx_group_last_error(None).decode())
        self.h = h
        self.devices = list(devices)
        self.descs: list[VfoDesc] = []
        self.published: list[tuple[bytes, int, bytes]] = []
        self._cb = _lib.PUBLISH_FN(lambda user, topic, rate, buf, length: self.published.append(
            (C.string_at(topic, 5), int(rate), C.string_at(buf, length))))
        self._chk(self.L.sdrx_group_set_publish_callback(self.h, self._cb, None))
        self._chk(self.L.sdrx_group_set_option(self.h, b"exact", arithmetic(exact)))
        for k, v in options.items():
            self._chk(self.L.sdrx_group_set_option(self.h, k.encode(), int(v)))

    def _chk(self, rc):
        if rc != 0:
            raise SdrxError(rc, self.L.sdrx_group_last_error(self.h).decode())

    @classmethod
    def from_topology(cls, topo: Topology, devices, **kw) -> "Group":
        g = cls(devices, **kw)
        for d in topo.vfos:
            c = _lib.desc_to_c(d)
            out = C.c_int(-1)
            g._chk(g.L.sdrx_group_add_vfo(g.h, C.byref(c), C.byref(out)))
            g.descs.append(d)
        g._chk(g.L.sdrx_group_finalize(g.h))
        return g

    def process(self, iq) -> None:
        iq = np.ascontiguousarray(iq, dtype=np.float32).reshape(-1)
        self.published.clear()
        self._chk(self.L.sdrx_group_process(self.h, iq.ctypes.data, iq.size // 2))

    def submit(self, iq) -> None:
        iq = np.ascontiguousarray(iq, dtype=np.float32).reshape(-1)
        self._chk(self.L.sdrx_group_submit(self.h, iq.ctypes.data, iq.size // 2))

    def submit_u8(self, iq_bytes, correct_dc: bool = False) -> None:
        b = np.ascontiguousarray(iq_bytes, dtype=np.uint8).reshape(-1)
        self._chk(self.L.sdrx_group_submit_u8(self.h, b.ctypes.data, b.size // 2, int(bool(correct_dc))))

    def process_u8(self, iq_bytes, correct_dc: bool = False) -> None:
        b = np.ascontiguousarray(iq_bytes, dtype=np.uint8).reshape(-1)
        self.published.clear()
        self._chk(self.L.sdrx_group_process_u8(self.h, b.ctypes.data, b.size // 2, int(bool(correct_dc))))

    def peer_access(self) -> bool:
        """True: every member reaches the first device's frame directly (same device or xGMI peer access)."""
        return bool(self.L.sdrx_group_peer_access(self.h))

    def member_context(self, k: int):
        ctx, dev = C.c_void_p(), C.c_int()
        self._chk(self.L.sdrx_group_member(self.h, k, C.byref(ctx), C.byref(dev)))
        return ctx, dev.value

    def stream(self, vid: int, missing_ok: bool = False):
        """decimate[decimateCount] of VFO `vid` (an id of the whole tree) from the member that holds it."""
        m, lid = self.locate(vid)
        ctx, _ = self.member_context(m)
        n = C.c_int()
        rc = self.L.sdrx_get_stream(ctx, lid, None, 0, C.byref(n))
        if rc == _lib.SDRX_ENOSTREAM and missing_ok:
            return None
        if rc != 0:
            raise SdrxError(rc, self.L.sdrx_last_error(ctx).decode())
        out = np.zeros(2 * max(n.value, 1), np.float32)
        rc = self.L.sdrx_get_stream(ctx, lid, out.ctypes.data, n.value, C.byref(n))
        if rc != 0:
            raise SdrxError(rc, self.L.sdrx_last_error(ctx).decode())
        return out[: 2 * n.value].view(np.complex64).copy()

    def submit_device(self, dev_ptr: int, n_complex: int, producer_stream: int | None = None) -> None:
        self._chk(self.L.sdrx_group_submit_device(self.h, C.c_void_p(dev_ptr), int(n_complex), C.c_void_p(producer_stream or 0)))

    def process_device(self, dev_ptr: int, n_complex: int, producer_stream: int | None = None) -> None:
        self._chk(self.L.sdrx_group_process_device(self.h, C.c_void_p(dev_ptr), int(n_complex), C.c_void_p(producer_stream or 0)))

    def wait(self) -> None:
        self.published.clear()
        self._chk(self.L.sdrx_group_wait(self.h))

    def sync(self) -> None:
        self._chk(self.L.sdrx_group_sync(self.h))

    def in_flight(self) -> int:
        return int(self.L.sdrx_group_in_flight(self.h))

    def output(self, vid: int) -> np.ndarray:
        buf, ln, rate = C.c_void_p(), C.c_uint32(), C.c_uint32()
        self._chk(self.L.sdrx_group_get_output(self.h, vid, C.byref(buf), C.byref(ln), C.byref(rate)))
        raw = C.string_at(buf.value, ln.value)
        return np.frombuffer(raw, dtype=np.int16 if self.descs[vid].demod_usb else np.int8).copy()

    def locate(self, vid: int) -> tuple[int, int]:
        m, l = C.c_int(), C.c_int()
        self._chk(self.L.sdrx_group_locate(self.h, vid, C.byref(m), C.byref(l)))
        return m.value, l.value

    def member_stats(self) -> list[dict | None]:
        out = []
        for k in range(len(self.devices)):
            ctx, dev = C.c_void_p(), C.c_int()
            self._chk(self.L.sdrx_group_member(self.h, k, C.byref(ctx), C.byref(dev)))
            if not ctx.value:
                out.append(None)
                continue
            s = _lib.StatsC()
            self.L.sdrx_get_stats(ctx, C.byref(s))
            out.append({k2: getattr(s, k2) for k2, _ in s._fields_})
        return out

    def close(self):
        if getattr(self, "h", None):
            self.L.sdrx_group_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# =============================================================================================
# The reference's own names
# =============================================================================================
class vfo:  # noqa: N801  (the reference's class name, vfo.h:11)
    """Mirror of ``class vfo`` (vfo.h:11-116): same setters, ``init`` and ``setVFOs``.  Nodes
    only collect parameters; the tree is instantiated on the GPU by the :class:`sdrj` that
    owns the main VFOs, at its first ``demodData`` (or an explicit ``start``)."""

    def __init__(self):
        self.desc = VfoDesc(gain=float(np.float32(0.01)), demod_usb=True, scalecomp=1)  # vfo.cpp:6-31
        self.children: list[vfo] = []
        self._radio: "sdrj | None" = None
        self._id = -1
        self._init = False
        self.emitFFT = False
        self.fftData = None  # callable(np.ndarray complex64): the Qt signal of vfo.h:46

    def setFs(self, samplerate): self.desc.fs = int(samplerate)
    def setDecimationCount(self, count): self.desc.decimate_count = int(count)
    def setMixerFreq(self, freq): self.desc.mixer_freq = float(freq)
    def getMixerFreq(self): return self.desc.mixer_freq
    def getOutRate(self): return self.desc.out_rate_stage
    def setFilterBandwidth(self, bw): self.desc.filter_bw = int(bw)
    def setGain(self, g): self.desc.gain = float(np.float32(g))
    def setDemodUSB(self, usb): self.desc.demod_usb = bool(usb)
    def getDemodUSB(self): return self.desc.demod_usb
    def setCompressonStyle(self, st): self.desc.cstyle = int(st)  # sic
    def setScaleComp(self, scale): self.desc.scalecomp = int(scale)
    def setZmqTopic(self, topic): self.desc.topic = str(topic)
    def setZmqAddress(self, address): self.zmqAddress = str(address)  # socket stays host-side

    def init(self, samplesPerBuffer, bind=True, lateDecimate=0):
        self.desc.samples_per_buffer = int(samplesPerBuffer)
        self.desc.late_decimate = int(lateDecimate)
        self._init = True

    def setVFOs(self, pVFOs):
        self.children = list(pVFOs)

    def fftVFOSlot(self, topic):
        """vfo.cpp:492-509: this VFO's decimate[decimateCount] goes to ``fftData`` after every
        frame while the selected topic is its own."""
        self.emitFFT = str(topic) == self.desc.topic
        if self._radio is not None and self._radio.rx is not None:
            self._radio._sync_taps()  # (a deselected fused leaf stops writing its decimate[0]; several VFOs may share a topic)

    # observation, available once the owning sdrj has started
    def _r(self) -> Receiver:
        if self._radio is None or self._radio.rx is None:
            raise RuntimeError("vfo is not attached to a started sdrj")
        return self._radio.rx

    @property
    def transmit_usb(self): return self._r().output(self._id)
    @property
    def transmit_iq(self): return self._r().output(self._id)
    @property
    def decimate_final(self): return self._r().stream(self._id)
    @property
    def outputRate(self): return self._r().output_rate(self._id)


class sdrj:  # noqa: N801  (the reference's class name, sdrj.h)
    """Mirror of the hot-path part of ``class sdrj``: ``setVFOs`` (sdrj.h) and
    ``demodData(const float*, int)`` (sdrj.cpp:266-305).  ``setDCCorrection`` keeps the DC-bias
    IIR on the host side for float input, exactly where the reference has it."""

    def __init__(self, device: int = 0, exact: bool = True, keep_prequant: bool = False):
        self.mains: list[vfo] = []
        self.rx: Receiver | None = None
        self.correctDC = False
        self._avept = np.zeros(2, np.float32)
        self._kw = dict(device=device, exact=exact, keep_prequant=keep_prequant)
        self.emitFFT = False
        self.count = 0
        self.fftData = None  # callable(np.ndarray complex64): the Qt signal of sdrj.h:40

    def setVFOs(self, vfos): self.mains = list(vfos)
    def setDCCorrection(self, dc): self.correctDC = bool(dc)

    def fftVFOSlot(self, topic):
        """sdrj.cpp:84-101: the raw spectrum is selected by the topic "Main"; any selection
        restarts the every-4th-call counter."""
        self.emitFFT = str(topic) == "Main"
        self.count = 0

    def _all_vfos(self):
        stack = list(self.mains)
        while stack:
            v = stack.pop(0)
            yield v
            stack[0:0] = v.children

    def _sync_taps(self):
        """The library's selection = the VFOs with emitFFT (every VFO whose topic equals the selected string, vfo.cpp:492-509)."""
        self.rx.set_taps([v._id for v in self._all_vfos() if v.emitFFT])

    def _after_frame(self):
        # vfo::process ends with `if (emitFFT) emit fftData(decimate[decimateCount])` (vfo.cpp:290-293),
        # demodData with the every-4th-call raw tap (sdrj.cpp:296-303)
        for v in self._all_vfos():
            if v.emitFFT and v.fftData is not None:
                v.fftData(self.rx.stream(v._id))
        if self.count == 4 and self.emitFFT:
            if self.fftData is not None:
                self.fftData(self.rx.raw())
            self.count = 0
        self.count += 1

    def start(self):
        self.rx = Receiver(**self._kw)

        def add(node: vfo, parent: int):
            if not node._init:
                raise RuntimeError("vfo::init was not called")
            node.desc.parent = parent
            node._id = self.rx.add_vfo(node.desc)
            node._radio = self
            for ch in node.children:
                add(ch, node._id)

        # ids must be assigned parents-first; children of main i come before main i+1's, which
        # is also the reference's publish order (sdrj.cpp:288-294, vfo.cpp:257-263)
        for m in self.mains:
            add(m, -1)
        self.rx.finalize()
        self._sync_taps()  # a selection made before the tree existed (fftVFOSlot, vfo.cpp:492-509)

    def demodData(self, data, length=None):
        if self.rx is None:
            self.start()
        data = np.ascontiguousarray(data, dtype=np.float32).reshape(-1)
        if length is not None:
            data = data[: int(length)]
        if self.correctDC:
            data = data.copy()
            _dc_correct(data, self._avept)
        self.rx.process(data)
        self._after_frame()

    def demodBytes(self, data):
        """The rtl_tcp byte stream (sdrj.cpp:155-160 feeds it through the LUT into demodData):
        LUT and DC-bias removal run on the device."""
        if self.rx is None:
            self.start()
        self.rx.process_u8(np.ascontiguousarray(data, dtype=np.uint8).reshape(-1), correct_dc=self.correctDC)
        self._after_frame()

    @property
    def published(self):
        return self.rx.published if self.rx else []


def _dc_correct(iq: np.ndarray, state: np.ndarray) -> None:
    """sdrj.cpp:277-283 on the host: avept = avept*(1-1e-6) + 1e-6*curr; curr -= avept, in fp32.
    A first-order recursion; evaluated blockwise in closed form is not bit-exact, so this is the
    plain sequential loop the reference runs (the device-side version lives in the library)."""
    keep = np.float32(1.0) - np.float32(0.000001)
    k = np.float32(0.000001)
    ar, ai = np.float32(state[0]), np.float32(state[1])
    re, im = iq[0::2], iq[1::2]
    for i in range(re.size):
        ar = np.float32(np.float32(ar * keep) + np.float32(k * re[i]))
        ai = np.float32(np.float32(ai * keep) + np.float32(k * im[i]))
        re[i] -= ar
        im[i] -= ai
    state[0], state[1] = ar, ai
