"""utils/distributed.py:7-124 of the reference (+ the bucketed gradient exchange that replaces its DDP wrappers)."""
from transeditor_amd.utils.distributed import (                                                     # noqa: F401
    GradSync, all_gather, broadcast_module, gather_grad, get_rank, get_world_size, reduce_loss_dict, reduce_sum, synchronize)
