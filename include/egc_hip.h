/*
 * egc_hip.h -- C ABI of libegc_hip.so: the MI355X (gfx950) native hot path of the EGC layer.
 *
 * This is the drop-in boundary.  The reference
 * (shyam196/egc) is pure Python; its layer classes reach native code only
 * through third-party torch extensions.  Each entry point below names the
 * reference call site(s) whose native work it replaces.  All pointers are
 * DEVICE pointers unless stated otherwise; all float tensors are fp32,
 * contiguous, row-major; indices handed IN are int64 (PyG edge_index), indices
 * held by the library's CSR are int32.  No torch types cross this boundary.
 * Nothing here allocates, frees or synchronises: every launch goes to the
 * caller's stream and is safe to capture into a hipGraph.
 *
 * Error convention: every function returns an egc_status (0 = ok).  The Python
 * host (egc_amd/_C.py) turns non-zero codes into RuntimeError, mirroring
 * TORCH_CHECK failures of the torch_scatter / torch_sparse ops it replaces.
 */
#ifndef EGC_HIP_H
#define EGC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* egc_stream_t; /* a hipStream_t */

typedef enum {
  EGC_OK = 0,
  EGC_ERR_INVALID = 1,     /* bad argument (null pointer, negative size, unknown enum, shape limits) */
  EGC_ERR_WORKSPACE = 2,   /* caller workspace too small */
  EGC_ERR_HIP = 3,         /* a HIP runtime call / launch failed; see egc_last_error() */
  EGC_ERR_UNSUPPORTED = 4  /* valid request outside the implemented envelope */
} egc_status;

/* Aggregators: union of layers.py:154-159 (add, mean, max, min, symadd, var, std) and
 * optimized_layers.py:92-94 (sum, mean, symnorm, min, max, var, std). */
typedef enum {
  EGC_AGGR_SUM = 0, EGC_AGGR_MEAN = 1, EGC_AGGR_MAX = 2, EGC_AGGR_MIN = 3,
  EGC_AGGR_VAR = 4, EGC_AGGR_STD = 5, EGC_AGGR_SYMNORM = 6
} egc_aggr;

/* Which edge set an aggregator reduces over (SURVEY.md 8a note 2):
 *   RAW    -- the CSR entries as given (duplicates and self-loops included);
 *   LOOPED -- the CSR entries minus every self-loop, plus exactly one self-loop per node
 *             (what gcn_norm / add_remaining_self_loops / fill_diag produce). */
typedef enum { EGC_SET_RAW = 0, EGC_SET_LOOPED = 1 } egc_edge_set;

/* Column order of the per-node weightings row:
 *   HBA: column = h*B*A + b*A + a   (EfficientGraphConv, layers.py:127-129)
 *   HAB: column = h*A*B + a*B + b   (EGConv, optimized_layers.py:195-202) */
typedef enum { EGC_LAYOUT_HBA = 0, EGC_LAYOUT_HAB = 1 } egc_weight_layout;

/* Nonlinearity on the weightings (layers.py:112-125, optimized_layers.py:183-184).
 * SOFTMAX is over the joint B*A axis per head. */
typedef enum { EGC_ACT_NONE = 0, EGC_ACT_SOFTMAX = 1, EGC_ACT_SIGMOID = 2, EGC_ACT_HARDTANH = 3 } egc_weight_act;

#define EGC_MAX_AGGRS 8
#define EGC_MAX_BASIS_WIDTH 1024   /* B * (F_out / H) */
#define EGC_MAX_WEIGHT_WIDTH 2048  /* H * B * A */
#define EGC_MAX_OUT_CHANNELS 2048

/* Rows with more than EGC_LONG_ROW_THRESHOLD entries are handled by whole wavefronts, in chunks of
 * EGC_LONG_ROW_CHUNK entries by separate wavefronts and merged (degree-skew handling). */
#define EGC_LONG_ROW_THRESHOLD 64
#define EGC_LONG_ROW_CHUNK 256

/* ------------------------------------------------------------------------------------------
 * Graph: CSR keyed by DESTINATION, entries of one row kept in input (edge_index) order.
 * Replaces torch_sparse.SparseTensor storage (rowptr / col; experiments/utils.py:107-113),
 * the gather index of MessagePassing.propagate and the degree pass of gcn_norm.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  int64_t n_nodes;
  int64_t n_edges;
  const int32_t* rowptr;      /* [n_nodes+1] */
  const int32_t* col;         /* [n_edges] source node of each entry */
  const int32_t* edge_id;     /* [n_edges] position of the entry in the input edge list; may be NULL */
  const float* dis_raw;       /* [n_nodes] indeg^-1/2 over RAW entries (0 where indeg==0); NULL if unused */
  const float* dis_looped;    /* [n_nodes] (non-self indeg + 1)^-1/2; NULL if unused */
  const int32_t* max_index;   /* device scalar: largest node id present in edge_index (-1 if none) */
  const int32_t* plan;        /* [egc_plan_ints(n_nodes,n_edges)] long-row work plan from egc_csr_prepare */
  int64_t n_chunks;           /* HOST copy of plan[1] (number of long-row chunks) if the caller has read it
                                 back, else -1: the launch is then sized for the plan's capacity */
  int64_t n_src_rows;         /* rows of the tables that `col` indexes (bases, dis_*): n_nodes on one GPU; in a
                                 vertex-partitioned run the owned rows come first and the halo rows received
                                 from other ranks follow (n_src_rows >= n_nodes).  0 means n_nodes.  A rectangular
                                 adjacency between two node types (relational EGC, rmag/models.py:131-134) may have
                                 any n_src_rows > 0, with the RAW edge set and without symnorm. */
  const float* edge_dis_raw;  /* [n_edges] dis_raw[col[p]] per CSR entry, or NULL: the source-side deg^-1/2 factor of the
                                 symnorm weight laid out in traversal order (egc_csr_edge_dis), so that the aggregate
                                 kernel streams it instead of gathering 4 bytes per entry; the weight itself,
                                 dis[src] * dis[dst], is still formed in the kernel */
  const float* edge_dis_looped; /* [n_edges] the same for dis_looped, or NULL */
} egc_graph;

/* int32 words the caller must allocate for egc_graph.plan. */
int64_t egc_plan_ints(int64_t n_nodes, int64_t n_edges);

/* Bytes of scratch egc_coo_to_csr needs (device). */
size_t egc_coo_to_csr_workspace_bytes(int64_t n_nodes, int64_t n_edges);

/* COO (edge_index[0]=src, edge_index[1]=dst, int64, any order, duplicates and self-loops allowed)
 * -> CSR by destination with a STABLE order inside each row.  Replaces the per-call
 * index_select/scatter index handling of MessagePassing.propagate (layers.py:191-193,
 * optimized_layers.py:191-193) and ToSparseTensor (experiments/utils.py:95-113).
 * Writes rowptr[n_nodes+1], col[n_edges], edge_id[n_edges], max_index[1].
 * Returns EGC_ERR_INVALID if n_nodes or n_edges >= 2^31.  Out-of-range node ids are NOT detected by this entry
 * (egc_coo_to_csr_checked is the one the host side of this repository calls). */
int egc_coo_to_csr(const int64_t* src, const int64_t* dst, int64_t n_edges, int64_t n_nodes,
                   int32_t* rowptr, int32_t* col, int32_t* edge_id, int32_t* max_index,
                   void* workspace, size_t workspace_bytes, egc_stream_t stream);

/* The same conversion with every node id RANGE-CHECKED, as egc_graph_build does it: an edge whose source is outside
 * [0, n_src_rows) (0 = n_nodes) or whose destination is outside [0, n_nodes) is dropped -- it sorts behind the last row,
 * rowptr[n_nodes] is the number of edges kept, col / edge_id behind it are defined (0 / undefined order) and never
 * referenced by a row -- and two flags are raised: *status (device int32, written by every call: 0 or 1) and, when
 * given, *host_flag = 1 (a STICKY word in host-visible memory, e.g. hipHostMalloc'ed: never cleared by the library, so
 * the host can poll it without synchronising the stream and report the error at its next call -- the reference's PyG
 * path raises at index_select, optimized_layers.py:191-193).  Same workspace as egc_coo_to_csr. */
int egc_coo_to_csr_checked(const int64_t* src, const int64_t* dst, int64_t n_edges, int64_t n_nodes, int64_t n_src_rows,
                           int32_t* rowptr, int32_t* col, int32_t* edge_id, int32_t* max_index, int32_t* status,
                           int32_t* host_flag, void* workspace, size_t workspace_bytes, egc_stream_t stream);

/* One call for a per-batch graph: COO -> CSR by destination (stable inside a row) + both deg^-1/2 tables + their
 * per-entry copies + the long-row plan, in five launches and without a library sort -- what egc_coo_to_csr +
 * egc_csr_prepare + egc_csr_edge_dis produce in a dozen (five launches: histogram, block sums, scan, scatter, rows).  Replaces the same reference call sites (per-batch
 * index handling of MessagePassing.propagate: zinc/models.py:60-74, mol/pna_style_models.py:64-79,
 * cifar/models.py:61-75; gcn_norm's degree pass).  Node ids are RANGE-CHECKED here: an edge whose source is outside
 * [0, n_src_rows) or whose destination is outside [0, n_nodes) is dropped and *status (device int32, written by
 * every call: 0 or 1) is set -- the PyG path behind optimized_layers.py:191-193 raises on such an index; the host side
 * reads the flag at its next synchronisation point (CSRGraph.check_indices); host_flag (may be NULL) is the sticky
 * host-visible word of egc_coo_to_csr_checked, polled without a synchronisation.  rowptr[n_nodes] is then the number
 * of edges kept; col / edge_id / edge_dis_* behind it are written with harmless values (0 / INT32_MAX / 0), and
 * egc_csr_transposed_coo emits (-1, -1) for those positions, which the transposed graph's build drops again.
 * n_src_rows = 0 means n_nodes.  dis_* / edge_dis_* may be NULL (skipped).
 * Workspace: egc_graph_build_workspace_bytes() bytes, zero-filled before its FIRST use; every call leaves ALL of it
 * zero again, so one buffer (of the largest size needed) serves graphs of any size on the same stream.  Scratch:
 * egc_graph_build_scratch_bytes() bytes, any content (sort area of rows longer than 4096 entries).
 * Limits: n_nodes, n_edges < 2^30. */
size_t egc_graph_build_workspace_bytes(int64_t n_nodes, int64_t n_edges);
size_t egc_graph_build_scratch_bytes(int64_t n_edges);
int egc_graph_build(const int64_t* src, const int64_t* dst, int64_t n_edges, int64_t n_nodes, int64_t n_src_rows,
                    int32_t* rowptr, int32_t* col, int32_t* edge_id, int32_t* max_index, float* dis_raw,
                    float* dis_looped, float* edge_dis_raw, float* edge_dis_looped, int32_t* plan, int32_t* status,
                    int32_t* host_flag, void* workspace, size_t workspace_bytes, void* scratch, size_t scratch_bytes,
                    egc_stream_t stream);

/* Degree statistics + long-row plan for an existing CSR.  Replaces gcn_norm's degree scatter and
 * pow(-0.5) (layers.py:173-178, optimized_layers.py:131-137) -- the per-edge weight
 * dis[src]*dis[dst] is formed inside the aggregate kernel, never materialised.
 * dis_raw / dis_looped may be NULL (skipped). */
int egc_csr_prepare(int64_t n_nodes, int64_t n_edges, const int32_t* rowptr, const int32_t* col,
                    float* dis_raw, float* dis_looped, int32_t* plan, egc_stream_t stream);

/* The CSR as the COO input of its TRANSPOSE's build: out_src[p] = the row that holds entry p, out_dst[p] = col[p]
 * (int64, as egc_graph_build / egc_coo_to_csr take them).  The backward's source-side pass walks the transposed graph
 * (autograd of the gather behind MessagePassing.propagate, optimized_layers.py:191-193); entry order is kept, so
 * the transposed graph's edge_id is the CSR position the arg comparisons need. */
int egc_csr_transposed_coo(int64_t n_nodes, int64_t n_edges, const int32_t* rowptr, const int32_t* col, int64_t* out_src,
                           int64_t* out_dst, egc_stream_t stream);

/* Per-entry copies of the source-side deg^-1/2 (egc_graph.edge_dis_*): out[p] = dis[col[p]].  Call after the
 * dis_* arrays are final (on a vertex partition: after their halo entries have arrived).  Either pair may be NULL. */
int egc_csr_edge_dis(int64_t n_edges, const int32_t* col, const float* dis_raw, const float* dis_looped,
                     float* edge_dis_raw, float* edge_dis_looped, egc_stream_t stream);

/* Vertex-partitioned runs (the reference is single-device, so there is no call site to cite beyond the
 * gather of MessagePassing.propagate this distributes): out[k][0..width) = table[idx[k]][0..width) -- the SEND PACK of
 * the halo all-to-all-v (the rows of `bases` other ranks read, grouped by destination rank), one launch.  width and ld
 * (row stride of table, floats) multiples of 4; table and out 16-byte aligned; idx int64 row ids. */
int egc_gather_rows_f32(const float* table, int64_t ld, const int64_t* idx, int64_t n_rows, int32_t width, float* out,
                        egc_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Layer description (the arguments of EfficientGraphConv.__init__ layers.py:13-27 /
 * EGConv.__init__ optimized_layers.py:74-86 that shape the computation).
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  int32_t in_channels;
  int32_t out_channels;
  int32_t num_heads;
  int32_t num_bases;
  int32_t num_aggrs;
  int32_t aggrs[EGC_MAX_AGGRS];   /* egc_aggr codes, in the layer's aggregator order */
  int32_t agg_set;                /* egc_edge_set for sum/mean/max/min/var/std */
  int32_t sym_set;                /* egc_edge_set for symnorm */
  int32_t loops_all_nodes;        /* LOOPED self-loop for every node (1) or only ids <= *max_index (0):
                                     add_remaining_self_loops without num_nodes, optimized_layers.py:164 */
  int32_t weight_layout;          /* egc_weight_layout */
  int32_t weight_act;             /* egc_weight_act */
  int32_t basis_stride;           /* floats from one basis to the next inside a `bases` row: 0 (or L = F_out / H)
                                     = contiguous, as torch.matmul(x, bases_weight) lays them out; a multiple of 4
                                     >= L = each basis padded with zero columns to 16-byte slots, which lets the
                                     register-resident kernels serve L that is not a multiple of 4 (the caller pads
                                     bases_weight with zero columns; egc_bases_ld() reports the row length) */
} egc_layer;

/* Leading dimension (floats) the library wants for the `bases` intermediate: B*L rounded up to 4. */
int32_t egc_bases_ld(const egc_layer* layer);

/* Step 1 -- basis transform + weightings Linear as ONE fp32 MFMA GEMM:
 *   [bases | weightings] = x[N,F_in] @ wcat[F_in, F_g+W]  (+ bcat on the W columns)
 * Replaces torch.matmul(x, bases_weight) (layers.py:97-101, optimized_layers.py:180) and
 * comb_weights(x) (layers.py:110, optimized_layers.py:182).
 * wcat = [bases_weight (F_in x F_g) | comb.weight^T (F_in x W)] row-major; bcat = comb.bias [W] or NULL.
 * bases is written with leading dimension ldb (>= F_g, pad columns zeroed); weightings dense [N,W]. */
int egc_basis_transform_f32(const float* x, const float* wcat, const float* bcat, int64_t n_nodes,
                            int32_t f_in, int32_t f_g, int32_t w_cols, float* bases, int32_t ldb,
                            float* weightings, egc_stream_t stream);

/* Step 1, fast form -- the same GEMM on the 16-bit matrix cores with fp32-level accuracy (split precision).
 * North-star shapes (96 < F_in <= 128, F_g % 32 == 0, 192 padded output columns): every x row and every
 * weight column is scaled by a power of two and split into two fp16 planes (22 significand bits); three
 * cross products are accumulated in fp32 by v_mfma_f32_32x32x16_f16 (egc_gemm_f16x2.hip).  Other shapes:
 * three bf16 planes per operand, six products on v_mfma_f32_32x32x16_bf16 (egc_gemm_bf16x3.hip).  Either way
 * the dropped terms are of the order of one fp32 rounding of |x||w| (parity tests: <= 1e-5 of the output scale).
 * The weight planes are produced once per parameter update by egc_basis_pack into a caller buffer of
 * egc_basis_pack_bytes() bytes (opaque; its layout depends on the shape); egc_basis_transform_packed then
 * has the contract of egc_basis_transform_f32 (same reference call sites) with `packed` in place of `wcat`.
 * On the fp16 path egc_basis_transform_packed needs x 16-byte aligned (EGC_ERR_UNSUPPORTED otherwise). */
size_t egc_basis_pack_bytes(int32_t f_in, int32_t f_g, int32_t w_cols);
int egc_basis_pack(const float* wcat, int32_t f_in, int32_t f_g, int32_t w_cols, void* packed,
                   size_t packed_bytes, egc_stream_t stream);
/* Operand precision of the split GEMM.  flags = 0: as above (fp16x2, 22 significand bits, where the shape has that
 * kernel).  EGC_GEMM_24BIT: three bf16 planes per operand for EVERY shape -- all 24 bits of an fp32 operand survive.
 * Rounds 2-3 ran layers with `std` / `var` on it: var = E[x^2] - E[x]^2 cancels on (nearly) constant neighbourhoods and
 * std = sqrt(relu(var) + 1e-5) (layers.py:203-216) then amplifies what the GEMM dropped from `bases` 158x.  Since round 4
 * the aggregate kernels accumulate the variance about the row's first entry -- E[(x - s)^2] - (E[x] - s)^2, the same number
 * without the cancellation -- and such layers sit within 7e-7 of the float64 value with either operand precision (the
 * float32 formula itself: up to 3.7e-4), so egc_layer_gemm_flags(layer) is 0 for every layer; EGC_GEMM_STDVAR_24BIT=1 in the
 * environment makes it EGC_GEMM_24BIT for std / var layers again.  Pass its value to BOTH calls of a layer (pack and
 * transform must agree); egc_layer_forward_packed applies it by itself, so its `packed` must come from egc_basis_pack_ex
 * with the same flags.  Cost of the 24-bit form at the 128-wide shapes: 63-70 us instead of 40 us per GEMM at ogbn-arxiv
 * size. */
#define EGC_GEMM_24BIT 1
int32_t egc_layer_gemm_flags(const egc_layer* layer);
int egc_basis_pack_ex(const float* wcat, int32_t f_in, int32_t f_g, int32_t w_cols, int32_t flags, void* packed,
                      size_t packed_bytes, egc_stream_t stream);
int egc_basis_transform_packed_ex(const float* x, const void* packed, const float* bcat, int64_t n_nodes, int32_t f_in,
                                  int32_t f_g, int32_t w_cols, int32_t flags, float* bases, int32_t ldb,
                                  float* weightings, egc_stream_t stream);
/* egc_basis_pack of a matrix given TRANSPOSED: wt [f_g + w_cols][ld >= f_in] row-major holds wcat^T (element (k, c) of
 * wcat at wt[c * ld + k]).  The gradient w.r.t. x is [d bases | d weightings] @ wcat^T -- a basis transform whose
 * "wcat" is the transpose of the layer's own operand: this packs it where it lies (no transposed copy per step). */
int egc_basis_pack_transposed(const float* wt, int64_t ld, int32_t f_in, int32_t f_g, int32_t w_cols, void* packed,
                              size_t packed_bytes, egc_stream_t stream);
int egc_basis_transform_packed(const float* x, const void* packed, const float* bcat, int64_t n_nodes,
                               int32_t f_in, int32_t f_g, int32_t w_cols, float* bases, int32_t ldb,
                               float* weightings, egc_stream_t stream);
/* egc_basis_transform_packed_ex with bases = x W + bases_addend (round 6): bases_addend [n_nodes][ldb], 16-byte aligned, joins the
 * bases columns in the kernel's store.  The gradient of a residual block  x + f(conv(x))  w.r.t. x is the layer's d x GEMM plus the
 * upstream gradient itself (autograd's add behind zinc/models.py:66-72): with the addend that sum costs no pass of its own.  Only
 * the long-k kernels (128 < f_in <= 384, the d x GEMM of the reference's 224- and 296-wide nets) carry it: EGC_ERR_UNSUPPORTED
 * elsewhere, and the caller adds; bases_addend == NULL: egc_basis_transform_packed_ex. */
int egc_basis_transform_packed_add(const float* x, const void* packed, const float* bcat, int64_t n_nodes, int32_t f_in,
                                   int32_t f_g, int32_t w_cols, int32_t flags, const float* bases_addend, float* bases, int32_t ldb,
                                   float* weightings, egc_stream_t stream);

/* Scratch bytes egc_aggregate_combine_f32 needs for this layer on this graph.
 * CONTRACT: its first egc_aggregate_workspace_zero_bytes() bytes must be zero before its FIRST use (the long-row
 * arrival counters of the fused kernel: 4 bytes per possible long row); the rest holds chunk records that are written
 * before they are read and may start with any contents -- a per-batch graph costs an allocation and a memset of a
 * few KB, not of the whole capacity-sized buffer.  Every call leaves the workspace ready for the next call on the
 * same stream.  One workspace must not be shared by calls that may run concurrently on different streams, nor
 * between graphs of different sizes (the zero part of one layout overlaps the records of another). */
size_t egc_aggregate_workspace_bytes(const egc_layer* layer, int64_t n_nodes, int64_t n_edges);
size_t egc_aggregate_workspace_zero_bytes(const egc_layer* layer, int64_t n_nodes, int64_t n_edges);
/* The same for one graph: when egc_graph.n_chunks carries the host copy of the plan's chunk count, the records of
 * chunk slots that do not exist are not allocated (ogbn-mag shape: 0.2 GB instead of 1.5 GB); with n_chunks = -1 it
 * equals egc_aggregate_workspace_bytes.  A workspace of this size serves exactly this graph. */
size_t egc_aggregate_workspace_bytes_for(const egc_layer* layer, const egc_graph* graph);

/* Step 2+3 -- fused multi-aggregator neighbourhood reduction + per-node head x basis x aggregator
 * combine (+ weight nonlinearity, + bias).  Replaces, in one pass over the CSR:
 *   gather x_j                      MessagePassing.__collect__ (via propagate, layers.py:191-193)
 *   symnorm message scaling         layers.py:195-199, optimized_layers.py:226-230
 *   scatter / spmm per aggregator   layers.py:201-225, optimized_layers.py:215-278
 *   stack + softmax/sigmoid/hardtanh + weighted sum / bmm + bias
 *                                   layers.py:109-138, optimized_layers.py:183-208
 * out is [n_nodes, out_channels].  arg_max / arg_min must be NULL here: the arg-extremum indices are an
 * output of the training form below (EGC_ERR_UNSUPPORTED otherwise). */
int egc_aggregate_combine_f32(const egc_graph* graph, const egc_layer* layer, const float* bases, int32_t ldb,
                              const float* weightings, const float* bias, float* out,
                              int32_t* arg_max, int32_t* arg_min,
                              void* workspace, size_t workspace_bytes, egc_stream_t stream);

/* egc_aggregate_combine_f32 restricted to the rows [row_begin, row_end): the other rows of `out` are not touched.
 * A vertex-partitioned run numbers its interior rows (all sources owned) first and finishes them while the halo
 * rows of `bases` are still in flight, then the boundary rows (egc_amd/partition.py). */
int egc_aggregate_combine_rows_f32(const egc_graph* graph, const egc_layer* layer, const float* bases, int32_t ldb,
                                   const float* weightings, const float* bias, float* out, int64_t row_begin,
                                   int64_t row_end, void* workspace, size_t workspace_bytes, egc_stream_t stream);

/* egc_aggregate_combine_f32 with the caller's elementwise tail fused into the store: the
 * reference's nets follow every layer with BatchNorm1d -> ReLU -> + identity (zinc/models.py:66-72,
 * mol/pna_style_models.py:71-78, cifar/models.py:67-74); in eval mode the normalisation is a per-channel affine
 * map, so   out = act((z + bias) * scale + shift) + residual   with z the layer's combine result.
 * scale / shift ([out_channels], both or neither), residual ([n_nodes, out_channels]) may be NULL; relu != 0
 * applies max(., 0) before the residual.  residual may be `out` itself: every element is read by the lane that
 * then writes it, so a sum of terms accumulates in place (relational EGC, rmag/models.py:131-144).
 * post == NULL is egc_aggregate_combine_f32. */
typedef struct {
  const float* scale;
  const float* shift;
  const float* residual;
  int32_t relu;
} egc_post;
int egc_aggregate_combine_post_f32(const egc_graph* graph, const egc_layer* layer, const float* bases, int32_t ldb,
                                   const float* weightings, const float* bias, const egc_post* post, float* out,
                                   void* workspace, size_t workspace_bytes, egc_stream_t stream);

/* egc_aggregate_combine_post_f32 whose `weightings` rows are ldw floats apart (ldw >= H*B*A, a multiple of 4,
 * weightings 16-byte aligned): the weightings of several terms computed by ONE GEMM share an array and each term
 * reads its column block.  Relational EGC (rmag/models.py:117-144) computes, per node type, the bases, the root
 * weightings and the weightings of every relation that targets the type from the same x -- one
 * egc_basis_transform_packed with w_cols = all of them, then one call of this per term.  post may be NULL. */
int egc_aggregate_combine_strided_f32(const egc_graph* graph, const egc_layer* layer, const float* bases, int32_t ldb,
                                      const float* weightings, int32_t ldw, const float* bias, const egc_post* post,
                                      float* out, void* workspace, size_t workspace_bytes, egc_stream_t stream);

/* Weight gradient of the layer's two Linear maps in one pass: out[f_in][k_cols] = x^T @ d for x [n_rows][f_in]
 * (row stride ldx) and d [n_rows][k_cols] (row stride ldd; the joint [d_bases | d_weightings] array of
 * egc_aggregate_combine_backward_f32), and, when col_sums != NULL, col_sums[k_cols] = the column sums of d (the
 * gradient of the combination Linear's bias).  Replaces what autograd runs for the reference's
 * `torch.matmul(x, self.bases_weight)` / `self.comb_weights(x)` (experiments/optimized_layers.py:177-178;
 * experiments/layers.py:110-111): exact fp32 products and sums on the fp32 matrix-core instruction, row ranges
 * added in a fixed order (deterministic).  f_in, k_cols, ldx, ldd multiples of 4 and 16-byte aligned pointers,
 * else EGC_ERR_UNSUPPORTED.  workspace: egc_weight_grad_workspace_bytes(n_rows, f_in, k_cols) bytes, contents
 * irrelevant on entry and on exit. */
int64_t egc_weight_grad_workspace_bytes(int64_t n_rows, int32_t f_in, int32_t k_cols);
/* How a call of that shape is laid out (host only, no GPU work; round 6): plan8 = {split-bf16 one-tile kernel (1) or exact-fp32
 * kernel (0), rows of an output tile, columns of an output tile, tiles along f_in, tiles along k_cols, row ranges, rows per range,
 * threads per workgroup}.  The exact-fp32 kernel is compiled for a list of tile shapes (32 .. 224 rows by 64 .. 320 columns) and
 * the host picks shape and row split together by modelled time; tests use this to see that every compiled shape is exercised. */
int egc_weight_grad_plan(int64_t n_rows, int32_t f_in, int32_t k_cols, int32_t* plan8);
/* The same with the column sums of a THIRD array riding along: e [n_rows][e_cols] (row stride lde; e_cols <= 128, a
 * multiple of 4) -> e_sums[e_cols].  A training step wants three reductions over the nodes -- the weight gradient, the
 * column sums of d weightings (bias of the combination Linear) and the column sums of grad_out (the layer's bias,
 * layers.py:137-138 / optimized_layers.py:207-208) -- and this is all three in one pass.  Only where the weight
 * gradient is one 128 x 192 accumulator tile (f_in <= 128, k_cols <= 192) and col_sums is requested; otherwise
 * EGC_ERR_UNSUPPORTED (use egc_column_sums_f32 for e).  e == NULL or e_sums == NULL: egc_weight_grad_f32. */
int64_t egc_weight_grad_ex_workspace_bytes(int64_t n_rows, int32_t f_in, int32_t k_cols, int32_t e_cols);
int egc_weight_grad_ex_f32(const float* x, int64_t ldx, const float* d, int64_t ldd, int64_t n_rows, int32_t f_in,
                           int32_t k_cols, float* out, float* col_sums, const float* e, int64_t lde, int32_t e_cols,
                           float* e_sums, void* workspace, int64_t workspace_bytes, void* stream);
/* egc_weight_grad_ex_f32 for a layer's OWN parameters (round 3): d = [d bases | d weightings] is the gradient of the GEMM
 * operand egc_weights_pack_f32 builds, and the reduction writes x^T d and the column sums of d straight into the gradients of
 * the basis matrices (one [f_in, B L] or B of [f_in, L]), of the combination Linear's weight and of its bias (d_comb_bias:
 * rows permuted with the weight's; or d_bcat: in the operand's order; either may be NULL) through the pack's index map -- no
 * d wcat array and no unpack launch.  e / e_sums as in egc_weight_grad_ex_f32 (the layer's bias gradient = column sums of
 * grad_out).  Workspace: egc_weight_grad_ex_workspace_bytes(n_rows, f_in, B Ls + H B A, e_cols). */
int egc_weight_grad_params_f32(const float* x, int64_t ldx, const float* d, int64_t ldd, int64_t n_rows, int32_t f_in,
                               int32_t num_heads, int32_t num_aggrs, int32_t num_bases, int32_t basis_len, int32_t basis_stride,
                               int32_t permute_hab, float* const* d_bases_parts, int32_t n_parts, float* d_comb_weight,
                               float* d_comb_bias, float* d_bcat, const float* e, int64_t lde, int32_t e_cols, float* e_sums,
                               void* workspace, int64_t workspace_bytes, void* stream);
int egc_weight_grad_f32(const float* x, int64_t ldx, const float* d, int64_t ldd, int64_t n_rows, int32_t f_in,
                        int32_t k_cols, float* out, float* col_sums, void* workspace, int64_t workspace_bytes,
                        void* stream);

/* The caller-side tail of a layer in TRAINING mode: conv -> BatchNorm1d (batch statistics) -> ReLU -> + input, as the
 * reference's nets run it every step (zinc/models.py:66-72, mol/pna_style_models.py:71-78, cifar/models.py:67-74).
 * Two streaming passes forward and two backward instead of one per PyTorch operator; in eval mode the tail needs no
 * pass at all (egc_post).  h, residual, out, dout, dh are dense [n_rows, cols] float32 arrays, cols a multiple of 4
 * (<= 1024 for the moments), every pointer 16-byte aligned (EGC_ERR_UNSUPPORTED otherwise).
 *
 * The ogbn-arxiv net puts a dropout between the ReLU and the residual add (arxiv/norm_models.py:34-40): `keep`
 * ([n_rows, cols] bytes, 0 = dropped, 4-byte aligned; NULL = no dropout) and keep_scale = 1 / (1 - p) carry it through
 * all three passes -- out = act(.) * keep * keep_scale + residual forward, g = dout * keep * keep_scale * [pre > 0]
 * backward.  The mask itself is the caller's (torch's generator on the Python side).
 *
 * n_valid (device int64, may be NULL): the number of leading rows that are real; rows [*n_valid, n_rows) are the
 * padding of a batch brought to the static shape of a hipGraph recording -- the statistics passes skip them and the
 * per-channel launches divide by *n_valid, so one recording serves batches of any size up to (n_rows, E) with
 * nn.BatchNorm1d's numerics on the real rows.  The elementwise passes write zeros to the padding rows (forward: they
 * stay clean for the next layer; backward: the per-channel constant of the BatchNorm gradient must not reach the
 * layer's bias and weight gradients through them).
 *
 *   egc_column_moments_f64      partials[p][0][c] = sum over the p-th block of rows of g[r][c],
 *                               partials[p][1][c] = sum of g[r][c] * b[r][c], accumulated in float64 (the caller adds
 *                               the n_partials blocks).  b == NULL: g = a, second moment of a itself (forward: batch
 *                               mean and variance of h).  b != NULL: g = a * keep * keep_scale * [b * scale + shift > 0]
 *                               (the last factor only with relu != 0; backward: a =
 *                               dout, b = h; the ReLU mask is recomputed from h, not stored), giving the two sums of
 *                               the BatchNorm backward.  partials: n_partials * 2 * cols doubles, 32-byte aligned.
 *                               count_inc (device int64, may be NULL) is incremented by one: nn.BatchNorm1d's
 *                               num_batches_tracked, bumped one launch before egc_bn_forward_finalize reads it.
 *   egc_bn_forward_finalize     everything per channel between the two forward passes, in one launch: the partials
 *                               added in a fixed order, stats = [mean | biased variance | 1/sqrt(var + eps)] (3 * cols
 *                               doubles), affine = [scale | shift] (2 * cols floats) for the elementwise pass, and --
 *                               running_mean != NULL -- the module's running statistics as nn.BatchNorm1d updates them
 *                               (unbiased variance; momentum >= 0, or momentum < 0: the cumulative average 1 /
 *                               *n_tracked with the count already incremented by the caller).  gamma / beta may be NULL.
 *   egc_bn_backward_finalize    the same between the two backward passes: out5 = [d gamma | d beta | coef_g | coef_h |
 *                               coef_1] (5 * cols floats) from the partial sums of the masked gradient and `stats`.
 *   egc_affine_act_residual_f32 out = act(h * scale + shift) + residual   (scale = gamma * rstd, shift = beta - mean *
 *                               scale; relu != 0: act = max(., 0); residual may be NULL)
 *   egc_affine_act_backward_f32 dh = coef_g * g + coef_h * h + coef_1 per channel, g as above: the BatchNorm backward
 *                               with its two sums folded into the three per-channel coefficient vectors. */
int egc_column_moments_f64(const float* a, const float* b, const float* scale, const float* shift, int32_t relu,
                           const uint8_t* keep, float keep_scale, int64_t n_rows, int32_t cols, double* partials,
                           int32_t n_partials, int64_t* count_inc, const int64_t* n_valid, egc_stream_t stream);
/* Statistics pass + per-channel step as ONE call (round 3): egc_column_moments_f64 followed by egc_bn_forward_finalize /
 * egc_bn_backward_finalize with the same arguments.  With `sync` (a device int32 that is zero on entry; it is zero again
 * on exit, so one word serves a module for ever) and at most 64 partial blocks it is also ONE LAUNCH: the block that
 * finishes last adds the partials (in the finalize kernels' order: same bits) and writes the per-channel results -- the
 * batched nets' training step (zinc/models.py:66-72 at 128 graphs per step) is bound by its launches.  sync == NULL or
 * more partial blocks: the two launches.  MEASURED (ZINC-shaped batch of 128, 4 blocks forward + backward, same box): the one
 * launch is 17-18 us per call against ~6 + ~4.5 for the two kernels (a chain of cross-XCD round trips: write-through
 * partials, the arrival atomic, the last block's reads), 426-429 us per replayed step against 372-373 with two launches;
 * eager 0.98-1.18 ms against 1.03-1.23.  The host side (egc_amd.FusedEGCBlock) therefore passes sync only on request
 * (EGC_BN_ONE_LAUNCH=1). */
int egc_bn_forward_stats_f32(const float* h, int64_t n_rows, int32_t cols, double* partials, int32_t n_partials,
                             int64_t* count_inc, const int64_t* n_valid, const float* gamma, const float* beta, double eps,
                             double* stats, float* affine, float* running_mean, float* running_var, double momentum,
                             const int64_t* n_tracked, int32_t* sync, egc_stream_t stream);
int egc_bn_backward_stats_f32(const float* dout, const float* h, const float* scale, const float* shift, int32_t relu,
                              const uint8_t* keep, float keep_scale, int64_t n_rows, int32_t cols, double* partials,
                              int32_t n_partials, const int64_t* n_valid, const double* stats, const float* gamma, float* out5,
                              int32_t* sync, egc_stream_t stream);
/* egc_bn_backward_stats_f32 that also returns the column sums of the dh egc_affine_act_backward_f32 will write, dh_col_sums[cols]
 * (round 6; NULL: the call above).  In the reference's blocks the conv's bias sits in front of the BatchNorm (zinc/models.py:66-72), so
 * its gradient -- autograd's sum of dh over the rows -- is zero but for rounding; here it is formed per channel from the sums this
 * step holds anyway (coef_g sum g + coef_h sum h + n coef_1 with the float32 coefficients as stored), which spares a training step
 * of the wide nets one pass over [n_rows, cols] per layer. */
int egc_bn_backward_stats_sums_f32(const float* dout, const float* h, const float* scale, const float* shift, int32_t relu,
                                   const uint8_t* keep, float keep_scale, int64_t n_rows, int32_t cols, double* partials,
                                   int32_t n_partials, const int64_t* n_valid, const double* stats, const float* gamma, float* out5,
                                   float* dh_col_sums, int32_t* sync, egc_stream_t stream);
int egc_bn_forward_finalize(const double* partials, int32_t n_partials, int32_t cols, int64_t n_rows, const float* gamma,
                            const float* beta, double eps, double* stats, float* affine, float* running_mean,
                            float* running_var, double momentum, const int64_t* n_tracked, const int64_t* n_valid,
                            egc_stream_t stream);
int egc_bn_backward_finalize(const double* partials, int32_t n_partials, int32_t cols, int64_t n_rows, const double* stats,
                             const float* gamma, float* out5, const int64_t* n_valid, egc_stream_t stream);
int egc_affine_act_residual_f32(const float* h, const float* scale, const float* shift, const float* residual,
                                int32_t relu, const uint8_t* keep, float keep_scale, int64_t n_rows, int32_t cols,
                                float* out, const int64_t* n_valid, egc_stream_t stream);
int egc_affine_act_backward_f32(const float* dout, const float* h, const float* scale, const float* shift, int32_t relu,
                                const uint8_t* keep, float keep_scale, const float* coef_g, const float* coef_h,
                                const float* coef_1, int64_t n_rows, int32_t cols, float* dh, const int64_t* n_valid,
                                egc_stream_t stream);

/* The GEMM operand of a layer from its parameters (grad == 0), or the parameters' gradients from the operand's
 * gradient (grad != 0: the parameter arrays are WRITTEN, wcat / bcat read) -- one launch instead of the cat / pad /
 * permute / transpose chain (and its autograd mirror) around the two weight matrices of layers.py:97-111 /
 * optimized_layers.py:177-182:
 *   wcat [f_in][B * basis_stride + H*B*A] = [the B basis matrices side by side, each padded from basis_len to basis_stride
 *                                           columns | comb_weight^T],        bcat [H*B*A] = comb_bias
 * bases_parts: n_parts = 1 -> one [f_in][B * basis_len] matrix (EGConv.bases_weight); n_parts = B -> B matrices
 * [f_in][basis_len] (EfficientGraphConv.bases_weight.{0..B-1}); a HOST array of device pointers.  permute_hab != 0: the
 * Linear's rows are ordered [h][a][b] (EGConv, optimized_layers.py:195-202) and become columns [h][b][a]; 0: they are
 * [h][b][a] already.  bcat / comb_bias may be NULL together.  All arrays dense float32. */
int egc_weights_pack_f32(const float* const* bases_parts, int32_t n_parts, const float* comb_weight, const float* comb_bias,
                         int32_t f_in, int32_t num_heads, int32_t num_aggrs, int32_t num_bases, int32_t basis_len,
                         int32_t basis_stride, int32_t permute_hab, float* wcat, float* bcat, int32_t grad,
                         egc_stream_t stream);

/* Column sums of a row-major array with row stride ld (floats), as n_partials partial rows:
 * partials[p, c] = sum of x[r, c] over the p-th block of rows, c < cols; the caller adds the few partial rows up.
 * The bias gradients of a training step (grad_out summed over the nodes for `bias`, d_weightings for the
 * combination Linear's bias -- what autograd's sum-to-size does behind layers.py:137-138 /
 * optimized_layers.py:182,207-208).  cols and ld multiples of 4, cols <= 1024, x and partials 16-byte aligned
 * (EGC_ERR_UNSUPPORTED otherwise); deterministic (no atomics). */
int egc_column_sums_f32(const float* x, int64_t n_rows, int32_t ld, int32_t cols, float* partials, int32_t n_partials,
                        egc_stream_t stream);
/* out[c] = sum over p of partials[p][c]: adds the partial rows of egc_column_sums_f32 up in a fixed order (cols a
 * multiple of 4, both pointers 16-byte aligned). */
int egc_sum_partials_f32(const float* partials, int32_t n_partials, int32_t cols, float* out, egc_stream_t stream);

/* Mean of the rows of x [n_rows, width] over consecutive segments: out[g] = mean(x[seg_ptr[g] : seg_ptr[g+1]])
 * (0 for an empty segment; x may be NULL when every segment is empty).  global_mean_pool over a PyG batch
 * vector, whose graphs are contiguous
 * (zinc/models.py:73, mol/pna_style_models.py:79, cifar/models.py:75); seg_ptr is int64 [n_segments + 1]. */
int egc_segment_mean_f32(const float* x, const int64_t* seg_ptr, int64_t n_segments, int32_t width, float* out,
                         egc_stream_t stream);

/* Training form of egc_aggregate_combine_f32: same `out`, plus what the backward needs instead of a second
 * gather.  stats (n_nodes * egc_train_stats_floats(layer) floats, opaque to the caller, handed to the backward
 * as it is) receives every row's raw running aggregates after the self-loop term (those of sum / variance -- as the
 * forward forms it, so that the backward's relu mask and std are the forward's -- / max / min / symnorm-weighted sum
 * that the aggregator list uses, [n_nodes][k][ldb]) and, behind them, the arg
 * positions below as one byte each relative to the row's first entry (what the backward gathers per transposed
 * entry: 64 instead of 256 bytes at the north star); cnt [n_nodes] the size of the row's aggregation set.  arg_max / arg_min
 * (each [n_nodes, ldb] int32; NULL allowed when the layer has no max / min aggregator) receive, per basis
 * column, the CSR entry position of the FIRST entry attaining the extremum -- graph.col[pos] is its source,
 * graph.edge_id[pos] its position in the input edge list, i.e. what torch_scatter's scatter_max/min return
 * as `arg` and autograd routes the gradient through -- n_edges for the appended self-loop of a LOOPED set,
 * -1 for an empty row. */
int64_t egc_train_stats_floats(const egc_layer* layer);
int egc_aggregate_combine_train_f32(const egc_graph* graph, const egc_layer* layer, const float* bases, int32_t ldb,
                                    const float* weightings, const float* bias, float* out, float* stats,
                                    int32_t* cnt, int32_t* arg_max, int32_t* arg_min, void* workspace,
                                    size_t workspace_bytes, egc_stream_t stream);
/* The same over rows [row_begin, row_end) only, all arrays full-size (the training counterpart of
 * egc_aggregate_combine_rows_f32: on a vertex partition the interior rows are finished while the halo rows of `bases`
 * travel).  The ranges of one forward are issued in ascending order and the last one ends at n_nodes. */
int egc_aggregate_combine_train_rows_f32(const egc_graph* graph, const egc_layer* layer, const float* bases, int32_t ldb,
                                         const float* weightings, const float* bias, float* out, float* stats,
                                         int32_t* cnt, int32_t* arg_max, int32_t* arg_min, int64_t row_begin,
                                         int64_t row_end, void* workspace, size_t workspace_bytes, egc_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Batches of small graphs (BASELINE configs 3 / 4): the aggregate+combine launch on TILES OF WHOLE GRAPHS, with the
 * graph preparation inside it (egc_aggregate_tile.hip).  The reference's batched nets call the layer with a PyG batch
 * (zinc/models.py:60-74, mol/pna_style_models.py:64-79, cifar/models.py:61-75): graphs numbered one after the other
 * (`batch.ptr` = their node offsets), the edges of one graph contiguous in `edge_index` -- a block-diagonal adjacency.
 *   egc_batch_tile_nodes   lds_nodes: how many nodes' basis rows (ldb floats each) fit the LDS of one workgroup (two per
 *                          CU) next to the per-tile CSR areas for (max_tile_nodes, max_tile_edges); 0 = the layer does not
 *                          qualify (envelope of the register-resident kernels: <= 64 slots per row, A <= 4, B a power
 *                          of two, no softmax).  A tile of at most lds_nodes nodes gathers its basis rows from LDS, a
 *                          larger one (up to max_tile_nodes) from memory like the ordinary kernels.
 *   egc_batch_plan         once per batch, one launch: cuts [0, n_nodes) into n_slots = ceil(n_nodes / slot) slots; the
 *                          graphs whose first node lies in slot k form one tile (n0, n1, e0, e1) -- node range and edge
 *                          range; the non-empty tiles are written to tiles[4 i ..] in any order and counted in *n_tiles
 *                          (device).  A tile holds < slot + (largest graph) nodes.  graph_ptr: int64 [n_graphs + 1];
 *                          edge_ptr: the graphs' edge offsets, int64 [n_graphs + 1] (PyG keeps them for a collated batch),
 *                          or NULL: found by searching the destination row of edge_index.
 *   egc_aggregate_combine_batch_f32
 *                          contract of egc_aggregate_combine_post_f32 (same reference call sites; plus those of
 *                          egc_graph_build: no CSR is built beforehand) for the rows of every tile, straight from the
 *                          COO lists src / dst; n_tiles_bound = n_slots (sizes the persistent launch).  ldw = row stride of
 *                          weightings (0 = dense).  max_index: device scalar, needed only by layers with
 *                          loops_all_nodes = 0.  Inference form.
 * Every edge is checked against its tile: an edge that leaves the tile (list not grouped by graph, id out of range) or a
 * tile beyond max_tile_nodes / max_tile_edges raises *status (bit 0 / bit 1; zero it before the call) and the sticky
 * *host_flag (see egc_coo_to_csr_checked); the rows of such a tile are written as zeros.  A row's entries are summed in the order of
 * the edge list (round 6: the tile's CSR is built in input order -- the order the reference's CPU scatter sums a row in): results are
 * bit-reproducible and bit-identical to egc_aggregate_combine_post_f32 on the CSR of the same batch.
 * ------------------------------------------------------------------------------------------ */
int32_t egc_batch_tile_nodes(const egc_layer* layer, int32_t max_tile_nodes, int32_t max_tile_edges, int32_t with_post);
int egc_batch_plan(const int64_t* graph_ptr, const int64_t* edge_ptr, int64_t n_graphs, const int64_t* dst, int64_t n_edges,
                   int64_t n_nodes, int32_t slot, int32_t* tiles, int32_t n_slots, int32_t* n_tiles, egc_stream_t stream);
int egc_aggregate_combine_batch_f32(const int32_t* tiles, const int32_t* n_tiles, int32_t n_tiles_bound, int32_t lds_nodes,
                                    int32_t max_tile_nodes, int32_t max_tile_edges, const int64_t* src, const int64_t* dst,
                                    int64_t n_nodes, const int32_t* max_index, const egc_layer* layer, const float* bases,
                                    int32_t ldb, const float* weightings, int32_t ldw, const float* bias, const egc_post* post,
                                    float* out, int32_t* status, int32_t* host_flag, egc_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Batches of small graphs, the WHOLE layer in ONE launch (egc_fused_tile.hip): per tile of whole graphs
 *   x rows -> basis transform + weightings Linear on the matrix cores -> LDS -> CSR of the tile's edges in LDS ->
 *   multi-aggregator reduction -> combine (+ bias, egc_post tail) -> out.
 * Neither `bases` nor `weightings` exist in memory and there is no plan launch: every workgroup finds its own tiles from
 * graph_ptr (and edge_ptr, or by searching the destination row of edge_index when it is NULL).
 * Replaces, for such batches, ALL the reference call sites of egc_basis_transform_packed + egc_graph_build +
 * egc_aggregate_combine_post_f32: torch.matmul(x, bases_weight), comb_weight(x) (experiments/layers.py:97-101,110;
 * optimized_layers.py:180-182), gcn_norm / add_remaining_self_loops (optimized_layers.py:127-175; layers.py:172-188),
 * MessagePassing.propagate + scatter per aggregator (optimized_layers.py:186-249; layers.py:191-219), the combine and bias
 * (optimized_layers.py:195-208; layers.py:127-138), and the caller's bn(eval) -> relu -> + identity (zinc/models.py:66-73,
 * cifar/models.py:64-71, mol/pna_style_models.py:70-78).
 *   egc_batch_fused_tile_nodes  rows of a tile whose image (bases + weightings rows, CSR areas for max_tile_edges entries)
 *                               fits the LDS of one CU; 0 = the layer is outside this kernel's envelope.  Two forms share the
 *                               entry points: the register-stationary one (F_in <= 128, ldb + H B A <= 192: the d = 128
 *                               layers) and, since round 5, the WIDE one for the reference's own wider batched nets
 *                               (run_pretrained.sh:7,12,23,24 -- 168 / H8 / B4, 296 / H8 / B4, 224 / H4 / B4): F_in <= 320,
 *                               round32(ldb) + H B A <= 384, weight fragments streamed from L2; both need F_in % 4 == 0 and
 *                               the register-resident aggregate kernels' envelope (ldb <= 256, B a power of two, A <= 4).
 *                               A batch whose largest graph has more nodes than this must take
 *                               egc_aggregate_combine_batch_f32 or the CSR path.
 *   egc_batch_fused_tile_quantum  rows per matrix-core step of the form that serves the layer (16 / 32; 0 = outside):
 *                               tile_nodes must be a multiple of it.
 *   egc_batch_fused_pack_bytes / egc_batch_fused_pack
 *                               wcat [F_in, F_g + W] (+ bcat [W] or NULL) -> the two fp16 weight planes in MFMA fragment
 *                               order + column scales + comb bias; once per parameter update.
 *   egc_layer_forward_batch_fused_f32
 *                               the launch.  tile_nodes: a multiple of egc_batch_fused_tile_quantum, <= egc_batch_fused_tile_nodes(...); status /
 *                               host_flag as egc_aggregate_combine_batch_f32 (bit 0: an edge leaves its tile or the graph
 *                               offsets do not cover [0, n_nodes); bit 1: a tile beyond tile_nodes / max_tile_edges -- the
 *                               rows of such a tile are written as zeros; bit 2: the workgroup-internal hand-over between
 *                               the wavefronts that build a tile's CSR timed out -- a bounded spin, never a hang; the
 *                               launch's output is then undefined).  Inference form, fp16x2-split GEMM (22-bit operands,
 *                               fp32 accumulate: the arithmetic of egc_basis_transform_packed at the north-star shape).
 *                               A row's entries are summed in the order of the edge list (round 6; the reference's CPU
 *                               scatter order): bit-reproducible; max_tile_edges <= 65535 (16-bit cursors of the CSR build).
 * ------------------------------------------------------------------------------------------ */
int32_t egc_batch_fused_tile_nodes(const egc_layer* layer, int32_t max_tile_edges, int32_t with_post);
int32_t egc_batch_fused_tile_quantum(const egc_layer* layer);
int64_t egc_batch_fused_pack_bytes(const egc_layer* layer);
int egc_batch_fused_pack(const egc_layer* layer, const float* wcat, const float* bcat, void* packed, int64_t packed_bytes,
                         egc_stream_t stream);
int egc_layer_forward_batch_fused_f32(const int64_t* graph_ptr, const int64_t* edge_ptr, int64_t n_graphs, const int64_t* src,
                                      const int64_t* dst, int64_t n_edges, int64_t n_nodes, const int32_t* max_index,
                                      const egc_layer* layer, const float* x, const void* packed, const float* bias,
                                      const egc_post* post, float* out, int32_t tile_nodes, int32_t max_tile_edges,
                                      int32_t* status, int32_t* host_flag, egc_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * The BACKWARD of egc_layer_forward_batch_fused_f32, tile-local, in ONE launch (round 5; egc_fused_tile.hip, MODE 1): what
 * PyTorch autograd derives through the layer for a PyG batch in the reference's training loops (zinc/configs.py:53-72:
 * loss.backward() through zinc/models.py:60-74 -> layers.py:89-140 / optimized_layers.py:177-210) -- the backward of the two
 * Linears, of propagate's gathers and of the per-aggregator scatters -- without a CSR, a transposed CSR or any saved
 * intermediate: per tile of whole graphs the forward's `bases` / `weightings` and aggregates are formed again in LDS,
 * d weightings = <grad_out, aggregates>, d aggregates = weightings x grad_out travel to the sources' rows of an LDS image kept as
 * 64-bit fixed point (integer LDS atomics: order-independent sums; the scale comes from the tile's largest |grad_out| and |weightings|,
 * 34 bits under the bound -- float LDS atomics run at one lane per clock per CU on gfx950)
 * (sum / mean / symnorm along every entry, max to the FIRST entry in input order attaining it: torch_scatter's arg rule), and
 *   d_x   [n_nodes, in_channels]  = [d bases | d weightings] [bases_weight | comb_weight^T]^T   (split-precision MFMA GEMM)
 *   d_cat [n_nodes, ld_dcat]      = the gradient of [bases (ldb) | pre-activation weightings (H B A, column (h B + b) A + a)]
 * leave the launch (d_cat may be NULL; the caller's weight gradient is x^T d_cat: egc_weight_grad_*).  Two launches on the same inputs
 * agree to the bit (round 6: the tile's CSR is built in input order; a tile whose grad_out or weightings hold an Inf / NaN leaves its
 * d bases rows -- and with them its rows of d_x -- as NaN instead of a fixed-point image of them).  Envelope (egc_batch_fused_bwd_tile_nodes > 0): B = 4 bases of 16 channels, H = 4 or 8
 * (the d = 64 / 128 layers), F_in <= 128, aggregators of sum / mean / max / symnorm, no weight nonlinearity.
 *   egc_batch_fused_bwd_tile_nodes   rows of a tile (its image also holds d bases, 512 B per row: 80 at the north star), 0 = outside
 *   egc_batch_fused_bwd_pack[_bytes] wcat -> the transposed operand's fp16 planes; once per parameter update
 *   egc_layer_backward_batch_fused_f32   `packed` = egc_batch_fused_pack's buffer (the forward operand, for the recompute);
 *                                    `d_x_add` [n_nodes, in_channels] or NULL: added to d_x in its store -- the gradient that
 *                                    reaches x past the layer (the residual branch of the reference's blocks, zinc/models.py:
 *                                    70-73: x = x + relu(bn(conv(x)))), which autograd otherwise adds in a pass of its own;
 *                                    status / host_flag as the forward launch.
 * ------------------------------------------------------------------------------------------ */
int32_t egc_batch_fused_bwd_tile_nodes(const egc_layer* layer, int32_t max_tile_edges);
int64_t egc_batch_fused_bwd_pack_bytes(const egc_layer* layer);
int egc_batch_fused_bwd_pack(const egc_layer* layer, const float* wcat, void* packed_t, int64_t packed_bytes, egc_stream_t stream);
/* egc_batch_fused_pack + egc_batch_fused_bwd_pack of the same wcat in one launch (a training step packs both once per update) */
int egc_batch_fused_train_pack(const egc_layer* layer, const float* wcat, const float* bcat, void* packed, int64_t packed_bytes,
                               void* packed_t, int64_t packed_t_bytes, egc_stream_t stream);
/* ... straight from the module's parameters through the index map of egc_weights_pack_f32 (bases_parts: ONE [F_in][B L] matrix or B of
 * [F_in][L]; comb_weight [H B A][F_in], its rows [h][a][b] with permute_hab (EGConv) or [h][b][a]; comb_bias in the rows' order, or
 * bcat in the operand's order, or neither): no wcat / bcat arrays and no pack launch in front. */
int egc_batch_fused_train_pack_params(const egc_layer* layer, const float* const* bases_parts, int32_t n_parts, const float* comb_weight,
                                      const float* comb_bias, const float* bcat, int32_t num_heads, int32_t num_aggrs, int32_t num_bases,
                                      int32_t basis_len, int32_t basis_stride, int32_t permute_hab, void* packed, int64_t packed_bytes,
                                      void* packed_t, int64_t packed_t_bytes, egc_stream_t stream);
int egc_layer_backward_batch_fused_f32(const int64_t* graph_ptr, const int64_t* edge_ptr, int64_t n_graphs, const int64_t* src,
                                       const int64_t* dst, int64_t n_edges, int64_t n_nodes, const int32_t* max_index,
                                       const egc_layer* layer, const float* x, const void* packed, const void* packed_t,
                                       const float* grad_out, float* d_x, const float* d_x_add, float* d_cat, int32_t ld_dcat,
                                       int32_t tile_nodes, int32_t max_tile_edges, int32_t* status, int32_t* host_flag,
                                       egc_stream_t stream);

/* Whole layer forward = egc_basis_transform_f32 + egc_aggregate_combine_f32
 * (EfficientGraphConv.forward layers.py:89-140 / EGConv.forward optimized_layers.py:177-210
 * after graph preparation).  bases [N,ldb] and weightings [N,W] are caller-provided intermediates. */
int egc_layer_forward_f32(const egc_graph* graph, const egc_layer* layer, const float* x, const float* wcat,
                          const float* bcat, const float* bias, float* bases, int32_t ldb, float* weightings,
                          float* out, void* workspace, size_t workspace_bytes, egc_stream_t stream);

/* Same as egc_layer_forward_f32 with the GEMM in its packed split-precision form (the default production path). */
int egc_layer_forward_packed(const egc_graph* graph, const egc_layer* layer, const float* x, const void* packed,
                             const float* bcat, const float* bias, float* bases, int32_t ldb, float* weightings,
                             float* out, void* workspace, size_t workspace_bytes, egc_stream_t stream);

/* Backward of egc_aggregate_combine_train_f32 (in the reference PyTorch autograd derives
 * it through layers.py:103-138 / optimized_layers.py:186-208).  Inputs: the forward's `bases`, PRE-activation
 * `weightings` (layout HBA), `stats`, `cnt`, `arg_max` / `arg_min`; grad_out = dL/d out [n_nodes, out_channels];
 * and t_graph = the TRANSPOSED graph (rows = sources, entries = destinations: egc_coo_to_csr with src/dst
 * swapped, then egc_csr_prepare for its long-row plan; its dis_* arrays are not used).  With max / min
 * aggregators the transposed graph must be built from the edge list in DESTINATION-CSR ORDER (edge k = CSR
 * entry k: source graph.col[k], destination = k's row), so that t_graph.edge_id maps a transposed entry to the
 * CSR position that arg_max / arg_min name.
 * Outputs: d_bases [n_src_rows, ldb] -- on a square graph (n_src_rows == n_nodes) every row is written and the
 * array may hold anything on entry; on a rectangular one it MUST be zero-filled by the caller (the partial sums
 * of hub rows arrive by float atomics) -- and d_weightings [n_nodes, H*B*A] (gradient w.r.t. the
 * pre-activation weightings).  ld_d_bases / ld_d_weightings are their row strides in floats (0 = dense: ldb
 * and H*B*A; ld_d_bases a multiple of 4, d_bases 16-byte aligned): a caller whose next step is a GEMM with
 * the concatenated weight matrix [bases_weight | comb_weight^T] passes two column blocks of ONE
 * [n_nodes, ldb + H*B*A] array and saves the concatenation.  The dense gradients (x, bases_weight, comb
 * weight/bias, bias) are plain GEMMs / column sums left to the caller.
 * Workspace: egc_backward_workspace_bytes(layer, n_nodes) holds the per-destination tables; with
 * egc_backward_workspace_bytes_for(layer, graph) bytes (64 more per entry and max / min aggregator) the gradients of
 * max / min travel as one 64-byte record per entry instead of arg bytes + pieces of the X rows (round 3: the source
 * kernel then fetches payload, not sectors).  Either size is accepted; the results are the same.  Round 6: the larger size is
 * returned (and records are built) only where an entry receives at most ten columns on average (ldb * n_nodes / n_edges: 4.6 on
 * the ogbn-arxiv graph at 64 basis columns); on batches of small graphs at wide layers (108 at 224 columns on a molecule batch)
 * every 12-item record overflowed, and both sizes are the tables' size there. */
size_t egc_backward_workspace_bytes(const egc_layer* layer, int64_t n_nodes);
size_t egc_backward_workspace_bytes_for(const egc_layer* layer, const egc_graph* graph);
int egc_aggregate_combine_backward_f32(const egc_graph* graph, const egc_graph* t_graph, const egc_layer* layer,
                                       const float* bases, int32_t ldb, const float* weightings,
                                       const float* grad_out, const float* stats, const int32_t* cnt,
                                       const int32_t* arg_max, const int32_t* arg_min, float* d_bases,
                                       int32_t ld_d_bases, float* d_weightings, int32_t ld_d_weightings,
                                       void* workspace, size_t workspace_bytes, egc_stream_t stream);

/* Human-readable text of the last HIP failure seen on the calling thread ("" if none). */
const char* egc_last_error(void);

/* Library build tag: "egc_hip <version> gfx950". */
const char* egc_version(void);

#ifdef __cplusplus
}
#endif
#endif /* EGC_HIP_H */
