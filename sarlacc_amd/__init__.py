"""sarlacc_amd -- MI355X-native implementation of sarlacc's alignment-and-consensus
hot path (adaptorAlign/barcodeAlign -> umiGroup -> multiReadAlign -> consensusReadSeq).

Layers:
  include/sarlacc_amd.h + sarlacc_amd/csrc   C ABI + hand-written HIP kernels (gfx950)
  sarlacc_amd.calls                          `.Call`-level mirror (reference src/init.cpp)
  sarlacc_amd.generics                       counterparts of the R generics (reference R/*.R)
"""
from ._lib import SarlaccError, device_count, set_device, last_kernel_ms, stage_ms, stage_count  # noqa: F401
from .encoding import Encoding, phred_encoding  # noqa: F401
from .strset import StringSet  # noqa: F401
from . import calls  # noqa: F401

__version__ = "0.1.0"
