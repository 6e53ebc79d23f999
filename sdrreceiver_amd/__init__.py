"""sdrreceiver_amd -- MI355X-native per-VFO IQ chain behind SDRReceiver's vfo / sdrj interface.

The product is ``libsdrx.so`` (C ABI in include/sdrx.h, HIP kernels in csrc/); this package is
the host-side mirror of the reference interface plus the configuration contract.
"""
from .topology import Topology, VfoDesc  # noqa: F401


def __getattr__(name):  # lazy: importing the package must not require the built library
    if name in ("Receiver", "vfo", "sdrj", "SdrxError"):
        from . import receiver
        return getattr(receiver, name)
    raise AttributeError(name)
