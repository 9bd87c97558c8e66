"""``from experiments.optimized_layers import EGConv`` -> the gfx950-native drop-in (egc_amd.optimized_layers)."""
from egc_amd.optimized_layers import EGConv  # noqa: F401
