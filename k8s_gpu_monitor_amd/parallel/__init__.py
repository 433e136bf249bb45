"""Interconnect validation with RCCL over xGMI (torch.distributed, one process per GPU)."""
