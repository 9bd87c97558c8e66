"""Import-path shim: ``from experiments.rmag.models import REGConv`` resolves to the gfx950 drop-in
(reference experiments/rmag/models.py:75-148).  The R-GCN baseline and the REGC wrapper are callers /
baselines and stay with the reference."""
from egc_amd.relational import EDGE_TYPES, NODE_TYPES, REGConv  # noqa: F401
