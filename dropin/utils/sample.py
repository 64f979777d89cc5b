"""utils/sample.py:3-21 of the reference."""
from transeditor_amd.utils.sample import prepare_noise_new, prepare_param                           # noqa: F401
