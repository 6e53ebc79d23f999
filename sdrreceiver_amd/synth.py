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
