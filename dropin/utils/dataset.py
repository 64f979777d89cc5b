"""utils/dataset.py:9-45 of the reference."""
from transeditor_amd.utils.dataset import (                                                         # noqa: F401
    DevicePrefetcher, MultiResolutionDataset, data_loader, image_transform, sample_data)
