"""Synthetic IQ that emulates the 8-bit RTL-SDR front end.

The reference feeds its chain floats ``b - 127`` for dongle bytes ``b`` (jonti/sdr.cpp:43-49).
The measurement plan in BASELINE.md uses integer-valued cf32 drawn from the LCG
``x <- x*1664525 + 1013904223 (mod 2^32)``, seed 1, one draw per component (I then Q),
component = ``((x >> 24) mod 17) - 8``, i.e. uniform in -8..8.  Small amplitudes keep
``|usb*gain*32768|`` far inside int16.
"""
from __future__ import annotations

import numpy as np

_A = np.uint32(1664525)
_C = np.uint32(1013904223)


class Lcg:
    """Vectorised LCG stream: ``draw(n)`` returns the next n states (after stepping)."""

    def __init__(self, seed: int = 1):
        self.x = np.uint32(seed)

    def draw(self, n: int) -> np.ndarray:
        if n == 0:
            return np.zeros(0, np.uint32)
        with np.errstate(over="ignore"):
            apow = np.cumprod(np.full(n, _A, np.uint32), dtype=np.uint32)  # a^1 .. a^n
            geo = np.empty(n, np.uint32)  # 1 + a + ... + a^(k-1) for k = 1..n
            geo[0] = 1
            if n > 1:
                geo[1:] = np.uint32(1) + np.cumsum(apow[:-1], dtype=np.uint32)
            xs = apow * self.x + _C * geo
        self.x = xs[-1]
        return xs


def lcg_frame(n_complex: int, lcg: Lcg) -> np.ndarray:
    """One raw frame as interleaved float32 [I0,Q0,I1,Q1,...] with components in -8..8."""
    x = lcg.draw(2 * n_complex)
    return (((x >> np.uint32(24)) % np.uint32(17)).astype(np.int32) - 8).astype(np.float32)


def lcg_frame_u8(n_complex: int, lcg: Lcg) -> np.ndarray:
    """The same frame as dongle bytes (b = component + 127)."""
    return (lcg_frame(n_complex, lcg) + 127).astype(np.uint8)


def tone_frame(n_complex: int, fs: float, tones, start: int = 0, noise_lcg: Lcg | None = None) -> np.ndarray:
    """Parity-test signal: a few complex tones (freq Hz relative to the raw centre, amplitude)
    plus optional LCG noise, rounded to integers like the 8-bit front end; phase-continuous
    across frames through `start` (index of the first sample)."""
    k = np.arange(start, start + n_complex, dtype=np.float64)
    z = np.zeros(n_complex, np.complex128)
    for f, a in tones:
        z += a * np.exp(2j * np.pi * f * k / fs)
    out = np.empty(2 * n_complex, np.float32)
    out[0::2] = np.round(z.real)
    out[1::2] = np.round(z.imag)
    if noise_lcg is not None:
        out += lcg_frame(n_complex, noise_lcg)
    return out


# ---- a capture-like stream (the reference holds no recorded IQ; BASELINE.json north_star: "on recorded IQ") --------
# What an RTL-SDR hands sdr::rtlsdr_callback on an Inmarsat Aero downlink (jonti/sdr.cpp:100-145; README.md:1-11), rebuilt from
# a seed: the tuner's wide-band noise, strong carriers no VFO listens to that take the bytes past +-100 (what sets the front
# end's gain in practice -- and the hard case for a mixer that is only NEARLY the reference's: their leakage lands in-band), a
# DC offset of the ADC (what correct_dc_bias=1 is for, sdrj.cpp:271-286), and bursts of the signals JAERO decodes -- 600 /
# 1200 Bd BPSK and 10 500 Bd OQPSK -- on VFO frequencies of the shipped sdr_25E profile, 10-25 dB over the noise in their
# own bandwidth, switching on and off inside the span, the strongest taking the int16 audio (vfo.cpp:328) near full scale
# at the INI's gains.
CAPTURE_25E_CARRIERS = ((310000.0, 62.0), (-400000.0, 30.0), (455500.0, 2.5))  # (Hz from the centre frequency, amplitude LSB): the 1st outside both main VFOs' bands,
#   the 2nd 14.5 kHz from VFO02 inside main VFO 1's, the 3rd a weak CW inside VFO15's 48 kHz audio band
CAPTURE_25E_BURSTS = (
    # (VFO topic, RF Hz of the VFO in sdr_25E.ini, audio offset Hz, baud, modulation, amplitude LSB, start s, stop s)
    ("VFO01", 1545005146, 1500.0, 600, "bpsk", 3.6, 0.00, 1.40),
    ("VFO03", 1545219706, 1400.0, 600, "bpsk", 3.5, 0.30, 2.00),
    ("VFO05", 1545114134, 1600.0, 600, "bpsk", 6.5, 0.85, 1.15),
    ("VFO07", 1545124261, 2400.0, 1200, "bpsk", 6.0, 0.10, 1.90),
    ("VFO09", 1545159288, 1500.0, 600, "bpsk", 4.0, 1.20, 2.00),
    ("VFO12", 1545189244, 1500.0, 600, "bpsk", 7.5, 0.55, 0.80),
    ("VFO14", 1546019800, 9000.0, 10500, "oqpsk", 7.0, 0.00, 0.95),
    ("VFO19", 1546137300, 8500.0, 10500, "oqpsk", 2.8, 0.40, 2.00),
    ("VFO23", 1546157500, 9000.0, 10500, "oqpsk", 5.5, 1.05, 1.60),
    ("VFO27", 1546178430, 8000.0, 10500, "oqpsk", 6.0, 0.20, 1.75),
)


def capture_like_u8(n_frames: int = 8, frame: int = 384000, fs: int = 1536000, center: int = 1545600000, seed: int = 20261002,
                    noise_sigma: float = 7.0, dc=(1.3, -0.7), bursts=CAPTURE_25E_BURSTS, carriers=CAPTURE_25E_CARRIERS) -> np.ndarray:
    """`n_frames` x `frame` complex samples as dongle bytes [I0,Q0,I1,Q1,...] (b = value + 127, clipped to 0..255 like
    the ADC): Gaussian noise of `noise_sigma` LSB per component, the DC offset `dc` (LSB), the `carriers`, and the bursts -- each a
    linearly-interpolated NRZ symbol stream (BPSK: real symbols; OQPSK: I and Q streams half a symbol apart) with 2 ms
    ramps, at RF - `center` + audio offset Hz.  Deterministic for a given numpy: the fixtures made from it carry the
    sha256 of these bytes."""
    rng = np.random.default_rng(seed)
    n = n_frames * frame
    z = np.empty(n, np.complex128)
    z.real = rng.standard_normal(n) * noise_sigma + dc[0]
    z.imag = rng.standard_normal(n) * noise_sigma + dc[1]
    tt = np.arange(n, dtype=np.float64) / fs
    for f, amp in carriers:
        z += amp * np.exp(2j * np.pi * f * tt)
    del tt
    for k, (_topic, rf, audio, baud, mod, amp, t0, t1) in enumerate(bursts):
        a, b = max(0, int(t0 * fs)), min(n, int(t1 * fs))
        if b <= a:
            continue
        t = np.arange(a, b, dtype=np.float64)
        brng = np.random.default_rng(seed + 1000 + k)
        nsym = int((b - a) * baud / fs) + 4

        def nrz(offset_sym):
            sym = brng.integers(0, 2, nsym + 2).astype(np.float64) * 2.0 - 1.0
            pos = (t - a) * (baud / fs) + offset_sym + 1.0
            i0 = np.floor(pos).astype(np.int64)
            fr = pos - i0
            # raised-cosine transition between neighbouring symbols over the middle half of the symbol period
            w = np.clip((fr - 0.25) * 2.0, 0.0, 1.0)
            w = 0.5 - 0.5 * np.cos(np.pi * w)
            return sym[i0] * (1.0 - w) + sym[i0 + 1] * w

        base = nrz(0.0).astype(np.complex128) if mod == "bpsk" else (nrz(0.0) + 1j * nrz(0.5)) / np.sqrt(2.0)
        ramp = np.clip(np.minimum(t - a, b - 1 - t) / (0.002 * fs), 0.0, 1.0)
        f = (rf - center) + audio
        ph0 = brng.uniform(0.0, 2.0 * np.pi)
        z[a:b] += amp * ramp * base * np.exp(1j * (2.0 * np.pi * f * (t / fs) + ph0))
    out = np.empty(2 * n, np.float64)
    out[0::2] = np.rint(z.real)
    out[1::2] = np.rint(z.imag)
    return np.clip(out + 127.0, 0.0, 255.0).astype(np.uint8)
