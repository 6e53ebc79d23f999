"""ctypes binding of libsdrx.so (include/sdrx.h).  There is no fallback: if the HIP library is
missing or cannot be loaded this module raises, and without a GPU ``sdrx_create`` fails."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SDRX_LIB") or os.path.join(_HERE, "libsdrx.so")  # SDRX_LIB: A/B builds of the SAME HIP library
CSRC = os.path.join(_HERE, "csrc")

NKERNELS = 8
SDRX_EINVAL, SDRX_ESTATE, SDRX_EFILTER, SDRX_EHIP, SDRX_EUNSUPPORTED, SDRX_ENOMEM, SDRX_ENOSTREAM = -1, -2, -3, -4, -5, -6, -7  # include/sdrx.h
SDRX_DIFFERENT = 1  # sdrx_*_if_same: the frame is not the one the source context staged


class VfoDescC(C.Structure):
    """struct sdrx_vfo_desc"""
    _fields_ = [
        ("fs", C.c_int32), ("decimate_count", C.c_int32), ("mixer_freq_hz", C.c_double),
        ("demod_usb", C.c_int32), ("late_decimate", C.c_int32), ("filter_bw_hz", C.c_int32),
        ("gain", C.c_float), ("cstyle", C.c_int32), ("scalecomp", C.c_int32), ("parent_id", C.c_int32),
        ("samples_per_buffer", C.c_int32), ("topic", C.c_char * 8),
    ]


def desc_to_c(d) -> VfoDescC:
    """topology.VfoDesc -> struct sdrx_vfo_desc"""
    return VfoDescC(fs=d.fs, decimate_count=d.decimate_count, mixer_freq_hz=float(d.mixer_freq),
                    demod_usb=int(d.demod_usb), late_decimate=d.late_decimate, filter_bw_hz=int(d.filter_bw),
                    gain=float(d.gain), cstyle=d.cstyle, scalecomp=d.scalecomp, parent_id=d.parent,
                    samples_per_buffer=d.samples_per_buffer, topic=d.topic.encode()[:7])


class StatsC(C.Structure):
    """struct sdrx_stats"""
    _fields_ = [
        ("n_vfos", C.c_int32), ("n_leaves", C.c_int32), ("n_levels", C.c_int32), ("exact", C.c_int32),
        ("algorithmic_bytes_per_frame", C.c_int64), ("vfo_samples_per_frame", C.c_int64),
        ("device_bytes", C.c_int64), ("frames", C.c_int64), ("mix_chunks_per_frame", C.c_int64),
        ("dc_blocks", C.c_int64), ("dc_fallback_blocks", C.c_int64), ("dc_retried_blocks", C.c_int64),
    ]


PUBLISH_FN = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_char), C.c_uint32, C.c_void_p, C.c_uint32)

# every symbol include/sdrx.h declares: (restype, argtypes)
_vp, _i = C.c_void_p, C.c_int
SYMBOLS = {
    "sdrx_abi_version": (_i, []),
    "sdrx_build_id": (C.c_char_p, []),
    "sdrx_create": (_i, [C.POINTER(_vp), _i]),
    "sdrx_destroy": (_i, [_vp]),
    "sdrx_last_error": (C.c_char_p, [_vp]),
    "sdrx_add_vfo": (_i, [_vp, C.POINTER(VfoDescC), C.POINTER(_i)]),
    "sdrx_set_option": (_i, [_vp, C.c_char_p, _i]),
    "sdrx_finalize": (_i, [_vp]),
    "sdrx_set_publish_callback": (_i, [_vp, PUBLISH_FN, _vp]),
    "sdrx_check_vfo": (_i, [C.POINTER(VfoDescC), C.c_char_p, C.c_size_t]),
    "sdrx_process": (_i, [_vp, _vp, _i]),
    "sdrx_process_u8": (_i, [_vp, _vp, _i, _i]),
    "sdrx_process_device": (_i, [_vp, _vp, _i]),
    "sdrx_submit": (_i, [_vp, _vp, _i]),
    "sdrx_submit_u8": (_i, [_vp, _vp, _i, _i]),
    "sdrx_submit_device": (_i, [_vp, _vp, _i]),
    "sdrx_wait": (_i, [_vp]),
    "sdrx_in_flight": (_i, [_vp]),
    "sdrx_fetch": (_i, [_vp]),
    "sdrx_sync": (_i, [_vp]),
    "sdrx_set_stream": (_i, [_vp, _vp]),
    "sdrx_get_output": (_i, [_vp, _i, C.POINTER(_vp), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "sdrx_get_stream": (_i, [_vp, _i, _vp, _i, C.POINTER(_i)]),
    "sdrx_set_tap": (_i, [_vp, _i]),
    "sdrx_add_tap": (_i, [_vp, _i]),
    "sdrx_get_raw": (_i, [_vp, _vp, _i, C.POINTER(_i)]),
    "sdrx_get_prequant": (_i, [_vp, _i, _vp, _i, C.POINTER(_i)]),
    "sdrx_get_taps": (_i, [_vp, _i, _i, _vp, _i, C.POINTER(_i)]),
    "sdrx_get_nco": (_i, [_vp, _i, C.c_long, C.c_long, _vp]),
    "sdrx_submit_shared": (_i, [_vp, _vp]),
    "sdrx_process_shared": (_i, [_vp, _vp]),
    "sdrx_submit_if_same": (_i, [_vp, _vp, _vp, _i]),
    "sdrx_process_if_same": (_i, [_vp, _vp, _vp, _i]),
    "sdrx_group_create": (_i, [C.POINTER(_vp), C.POINTER(_i), _i]),
    "sdrx_group_destroy": (_i, [_vp]),
    "sdrx_group_last_error": (C.c_char_p, [_vp]),
    "sdrx_group_size": (_i, [_vp]),
    "sdrx_group_add_vfo": (_i, [_vp, C.POINTER(VfoDescC), C.POINTER(_i)]),
    "sdrx_group_set_option": (_i, [_vp, C.c_char_p, _i]),
    "sdrx_group_set_publish_callback": (_i, [_vp, PUBLISH_FN, _vp]),
    "sdrx_group_finalize": (_i, [_vp]),
    "sdrx_group_process": (_i, [_vp, _vp, _i]),
    "sdrx_group_submit": (_i, [_vp, _vp, _i]),
    "sdrx_group_submit_u8": (_i, [_vp, _vp, _i, _i]),
    "sdrx_group_process_u8": (_i, [_vp, _vp, _i, _i]),
    "sdrx_group_peer_access": (_i, [_vp]),
    "sdrx_group_submit_device": (_i, [_vp, _vp, _i, _vp]),
    "sdrx_group_process_device": (_i, [_vp, _vp, _i, _vp]),
    "sdrx_group_wait": (_i, [_vp]),
    "sdrx_group_in_flight": (_i, [_vp]),
    "sdrx_group_sync": (_i, [_vp]),
    "sdrx_group_get_output": (_i, [_vp, _i, C.POINTER(_vp), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "sdrx_group_locate": (_i, [_vp, _i, C.POINTER(_i), C.POINTER(_i)]),
    "sdrx_group_member": (_i, [_vp, _i, C.POINTER(_vp), C.POINTER(_i)]),
    "sdrx_get_stats": (_i, [_vp, C.POINTER(StatsC)]),
    "sdrx_enable_kernel_timing": (_i, [_vp, _i]),
    "sdrx_get_kernel_times": (_i, [_vp, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "sdrx_kernel_name": (C.c_char_p, [_i]),
}

_lib = None


def build(force: bool = False) -> str:
    """hipcc --offload-arch=gfx950 build of libsdrx.so, in-tree (sdrreceiver_amd/csrc/Makefile)."""
    if force and os.path.exists(LIB_PATH):
        os.remove(LIB_PATH)
    subprocess.check_call(["make", "-C", CSRC], stdout=subprocess.DEVNULL)
    return LIB_PATH


def lib() -> C.CDLL:
    """The loaded library.  Raises if the HIP extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: the HIP extension must be built "
                              f"(python -c 'import __graft_entry__ as g; g.build()' or make -C {CSRC}); "
                              "there is no CPU fallback")
        # PyTorch wheels bundle their own libamdhip64.so.7 / libhsa-runtime64.so.1 -- the same
        # SONAMEs as /opt/rocm's, so whichever copy a process loads first serves everything in it.
        # libsdrx.so runs on either; torch cannot initialise on top of the newer system runtime
        # ("no ROCm-capable device is detected").  So when torch is installed it loads its copy
        # first, and a process may use the library and torch in any order afterwards.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            f = getattr(L, name)  # AttributeError if the header and the library ever diverge
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib
