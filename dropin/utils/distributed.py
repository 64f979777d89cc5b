"""utils/distributed.py:7-124 of the reference (+ the bucketed gradient exchange that replaces its DDP wrappers)."""
from transeditor_amd.utils.distributed import (                                                     # noqa: F401
    GradSync, all_gather, broadcast_module, gather_grad, get_rank, get_world_size, reduce_loss_dict, reduce_sum, synchronize)

# the reference's path-length step differentiates the OUTPUTS of the DDP-wrapped generator with respect to each other; torch >= 1.9
# returns them through an identity node that breaks that (see legacy_ddp_outputs): restore the behaviour of the torch the reference pins
import os as _os

from transeditor_amd.utils.distributed import legacy_ddp_outputs                                    # noqa: E402,F401

if _os.environ.get('TE_DROPIN_KEEP_DDP_SINK', '0') != '1':
    legacy_ddp_outputs()

