"""Multi-GPU side of the diagnostics: one process per GPU over torch.distributed
(RCCL on MI355X, gloo on CPU), collective/pairwise bandwidth sweeps over xGMI."""
from .collectives import OPS, SweepPoint, busbw_factor, default_sizes, p2p_matrix, peak_busbw, sweep  # noqa: F401
from .sweep import init_distributed  # noqa: F401
