"""Data-parallel gradient reduction over RCCL (torch.distributed backend 'nccl' on ROCm; 'gloo' on CPU tests)."""
from .reducer import GradReducer, broadcast_parameters, init_process_group  # noqa: F401
