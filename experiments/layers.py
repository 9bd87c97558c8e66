"""``from experiments.layers import EfficientGraphConv`` -> the gfx950-native drop-in (egc_amd.layers)."""
from egc_amd.layers import EfficientGraphConv, _AggLayer  # noqa: F401
