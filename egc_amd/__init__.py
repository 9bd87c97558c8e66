"""egc_amd -- the EGC message-passing layer's hot path, native to MI355X (gfx950).

Public surface (mirrors the reference's layer API):
  EfficientGraphConv   drop-in for experiments/layers.py:EfficientGraphConv
  EGConv               drop-in for experiments/optimized_layers.py:EGConv
  REGConv              drop-in for experiments/rmag/models.py:REGConv (relational EGC)
  FusedEGCBlock        conv -> BatchNorm1d -> ReLU (-> dropout) -> + identity: eval mode in the kernel's store, training
                       mode in two passes each way; global_mean_pool (differentiable segmented mean)
  SparseTensor         minimal adj_t container (torch_sparse is not required)
  CSRGraph             device CSR + degree statistics + long-row plan
  GraphBatch           a PyG-style batch of small graphs (edge_index + graph offsets) for the tile kernels: the CSR of
                       each tile of whole graphs is built in LDS by the workgroup that aggregates it
  egc_layer_forward    operator-level call into libegc_hip.so
  ops                  the same call as torch.library operators (torch.ops.egc_amd.layer_forward / _train / _backward)
  GraphedStep          a whole training / inference step (graph build included) recorded as one hipGraph
"""
from .graph import CSRGraph, GraphBatch, SparseTensor, GLOBAL_GRAPH_CACHE  # noqa: F401
from .functional import egc_layer_forward, make_spec, LayerSpec  # noqa: F401
from .layers import EfficientGraphConv  # noqa: F401
from .optimized_layers import EGConv  # noqa: F401
from .relational import REGConv  # noqa: F401
from .fusion import FusedEGCBlock, global_mean_pool  # noqa: F401
from .hipgraph import GraphedStep  # noqa: F401
from . import ops  # noqa: F401  (registers torch.ops.egc_amd.*)

__version__ = "0.1.0"
