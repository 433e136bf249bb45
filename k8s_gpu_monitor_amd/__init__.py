"""MI355X-native GPU monitoring stack — Python package.

Layers (all backed by native code under ``core/`` and ``diag/``):
  engine    façade over libgpumon.so: host engine, exporter, pod attribution
  models    device data model with the shape of the reference's nvml package
  ops       active diagnostics: hand-written CDNA4 HIP kernels (HBM stream, memtest, MFMA GEMM)
  parallel  interconnect validation: RCCL collectives over xGMI, one process per GPU
  utils     exposition-format parser, amd-smi oracle, benchmark helpers
"""
from . import _abi as abi  # noqa: F401
from ._lib import GpumonError, build, lib  # noqa: F401
from . import engine  # noqa: F401

__version__ = "0.1.0"
