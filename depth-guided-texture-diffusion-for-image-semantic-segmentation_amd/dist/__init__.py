"""Data-parallel gradient reduction over RCCL (torch.distributed backend 'nccl' on ROCm)."""
