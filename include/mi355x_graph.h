/*
 * mi355x_graph.h -- C ABI of the MI355X (gfx950 / CDNA4) sparse message-passing library.
 *
 * This is the drop-in boundary for the one hot path of dglai/dgl-0.5-benchmark:
 * what DGLGraph.update_all(builtin, builtin) / apply_edges(builtin) / dgl.ops.gspmm /
 * dgl.ops.gsddmm / edge_softmax execute.  The reference itself is pure Python; the seam it
 * crosses is DGL v0.6.x's FFI pair
 *     _CAPI_DGLKernelSpMM (graph, op, reduce_op, U, E, V, ArgU, ArgE)
 *     _CAPI_DGLKernelSDDMM(graph, op, lhs, rhs, out, lhs_target, rhs_target)
 * (UPSTREAM src/array/kernel.cc, called from python/dgl/sparse.py::_gspmm/_gsddmm), reached
 * from the reference at
 *     kernel/dgl-new.py:20,39
 *     end_to_end/full_graph/node_classification/main_dgl_product_sage.py:62
 *     end_to_end/full_graph/node_classification/main_dgl_reddit_gat.py:10,31-55
 *     end_to_end/full_graph/node_classification/main_dgl_proteins_rgcn_for.py:52
 *     end_to_end/full_graph/graph_classification/main_dgl_molhiv_gcn.py:41-52,75
 * Each entry point below names the reference interface it replaces.
 *
 * Conventions (same as the upstream seam):
 *   - every pointer is a DEVICE pointer unless the name ends in `_host`; the CALLER allocates
 *     every output / arg-index / workspace buffer, the library never allocates or frees;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); all work is enqueued
 *     on it and the call returns without synchronising;
 *   - graph index arrays are int32 or int64 (`idx_bits`); element offsets are 64-bit inside;
 *   - features are dense row-major fp32 (`MGX_F32`), `*_len` = elements per row;
 *   - broadcasting ((N,H,F) op (E,H,1)) is expressed by optional device tables
 *     `*_off[out_len]` mapping an output element to the operand element (NULL = identity),
 *     the formulation of DGL's BcastOff;
 *   - every function returns MGX_OK (0) or an error code; mgx_last_error() gives the text
 *     (thread-local).  Nothing aborts the process.
 *
 * No torch types, no C++ types: plain pointers and sizes, bindable from ctypes / cgo / JNI.
 */
#ifndef MI355X_GRAPH_H_
#define MI355X_GRAPH_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  MGX_OK = 0,
  MGX_ERR_INVALID_ARGUMENT = 1,
  MGX_ERR_UNSUPPORTED = 2,
  MGX_ERR_HIP = 3
} mgx_status;

/* binary ops of kernel/utils.py:8-16 (binary_op_dict) */
typedef enum {
  MGX_OP_ADD = 0, MGX_OP_SUB = 1, MGX_OP_MUL = 2, MGX_OP_DIV = 3,
  MGX_OP_COPY_LHS = 4, MGX_OP_COPY_RHS = 5, MGX_OP_DOT = 6
} mgx_op;

/* reducers selectable with kernel/dgl-new.py:51 --spmm-reduce */
typedef enum { MGX_REDUCE_SUM = 0, MGX_REDUCE_MAX = 1, MGX_REDUCE_MIN = 2, MGX_REDUCE_MEAN = 3 } mgx_reduce;

/* flags of mgx_spmm_csr / mgx_spmm_copy_u_strided */
enum {
  MGX_SPMM_ACCUMULATE = 1, /* out += result (SUM / MEAN only) */
  MGX_SPMM_SHORT_ROWS = 2  /* the caller vouches that the work items (plan items, or rows without a plan) are SHORT AND EVEN: for every
                              batch of B = 64 / G consecutive items (G = lanes per feature row = next power of two >= D / 4) the longest is
                              not much above the batch mean.  copy_lhs / copy_rhs with SUM | MEAN then take the kernel that gives every
                              item a lane group of its own (2 - 3.5x on rows of ~3 edges); without the flag, or on skewed items, a wave per
                              item.  Items above 32 edges are taken by the whole wave inside that kernel, one after the other: keep them
                              (rows above 32 edges, the chunks of split rows) out of the way in the plan's `rest` part (mgx_spmm_plan),
                              or do not set the flag.  A hint: results are the same either way (other fp32 summation order). */
};

/* SDDMM operand targets (lhs_target / rhs_target of dgl.ops.gsddmm) */
typedef enum { MGX_TARGET_U = 0, MGX_TARGET_E = 1, MGX_TARGET_V = 2 } mgx_target;

/* A CSR view.  For message passing this is the IN-CSR ("CSC"): one row per destination node,
 * indices[p] = source node, eids[p] = id of the edge in the original COO (NULL = identity). */
typedef struct mgx_csr {
  int64_t num_rows;
  int64_t num_cols;
  int64_t nnz;
  const void* indptr;   /* [num_rows + 1] */
  const void* indices;  /* [nnz] */
  const void* eids;     /* [nnz] or NULL */
  int32_t idx_bits;     /* 32 or 64 */
  int32_t reserved;
} mgx_csr;

/* Optional execution schedule of a CSR for the summing g-SpMM (built once per graph, like the CSR
 * itself; mi355x_graph/schedule.py builds it).  Work items replace the natural row order:
 *   - items are walked in array order, 64 per workgroup, each XCD (own L2) taking a contiguous
 *     range -- a locality-aware order (rows of one community adjacent) turns gathers into L2 hits;
 *   - rows longer than the split threshold appear as several items (edge ranges) whose partial
 *     sums go to `partial_ws` slots and are combined in slot order by a fix-up launch, so a hub
 *     row never serialises on one wavefront and the result stays deterministic.
 * Every row must be covered exactly once (one direct item, or >= 2 slot items + a hub entry). */
typedef struct mgx_spmm_plan {
  int64_t num_items;
  const int32_t* item_row;  /* [num_items] >= 0: row written directly; < 0: partial slot -(v+1) */
  const void* item_beg;     /* [num_items] first edge position (graph index width) */
  const void* item_end;     /* [num_items] one past the last edge position */
  int64_t num_hubs;
  const int32_t* hub_row;       /* [num_hubs] rows that were split */
  const int32_t* hub_slot_ptr;  /* [num_hubs+1] slots of hub h are [ptr[h], ptr[h+1]) */
  int64_t num_slots;            /* partial_ws holds num_slots * out_len floats */
  const int32_t* slot_item;     /* [num_slots] index of the work item that owns each partial slot */
  const int32_t* item_node;     /* [num_items] the row of EVERY item (direct or split); used by mgx_sddmm_csr */
  /* Optional (HOST values): items [xcd_item_start[x], xcd_item_start[x+1]) are walked by XCD x.  Each XCD (own L2) gets a
   * CONTIGUOUS stretch of the schedule; equal item counts leave the XCDs unequal WORK on skewed graphs (R-MAT: 151 % vs 0.1 %
   * of the mean edge count), so the builder cuts the schedule at equal EDGE counts.  All zero: equal item counts.
   * xcd_item_start_dev: the same nine numbers in device memory (kernels read their two from there: a by-value array indexed
   * by blockIdx costs scalar registers and, through them, resident workgroups). */
  int64_t xcd_item_start[9];
  const int64_t* xcd_item_start_dev;
  /* Optional second part (NULL: this plan is the whole schedule).  With `rest`, THIS plan holds only short direct items (item_row >= 0,
   * at most 32 edges, no slots or hubs) and `rest` holds every other item with the hub tables; together they cover each row once.
   * Accepted by mgx_spmm_csr / mgx_spmm_copy_u_strided with MGX_SPMM_SHORT_ROWS for copy_lhs / copy_rhs with SUM | MEAN only: the short
   * items run on the lane-group kernel, `rest` on the wave-per-item kernel (partial_ws: rest->num_slots * out_len floats).  A hub row
   * of 13 k edges is 52 consecutive 256-edge chunks; walked by the lane-group kernel they would be one wave's serial tail. */
  const struct mgx_spmm_plan* rest;
} mgx_spmm_plan;

/* Device-side construction of the plan tables (hub-row splitting over an optional row order; NULL = natural order).
 *   ws      = mgx_spmm_plan_workspace(num_rows) bytes, passed UNCHANGED from _count to _fill (it carries the scans);
 *   _count  writes {num_items, num_hubs, num_slots} to `totals` (device int64[3]); the caller reads them, allocates
 *           item_row/item_node [num_items] (int32), item_beg/item_end [num_items] (graph index width),
 *           hub_row [num_hubs], hub_slot_ptr [num_hubs+1], slot_item [num_slots] (int32) and calls
 *   _fill   with the same csr / split / row_order.
 * The locality-aware row order itself is computed by the host layer (mi355x_graph/schedule.py). */
int64_t mgx_spmm_plan_workspace(int64_t num_rows);
int32_t mgx_spmm_plan_count(const mgx_csr* csr, int64_t split, const void* row_order, int64_t* totals,
                            void* workspace, int64_t workspace_bytes, void* stream);
int32_t mgx_spmm_plan_fill(const mgx_csr* csr, int64_t split, const void* row_order,
                           int32_t* item_row, void* item_beg, void* item_end, int32_t* item_node,
                           int32_t* hub_row, int32_t* hub_slot_ptr, int32_t* slot_item,
                           void* workspace, int64_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------ misc */
const char* mgx_last_error(void);
/* ABI version, bumped on any signature change. */
int32_t mgx_abi_version(void);
/* Name of the kernel family the calling thread's last g-SpMM entry point launched ("rowwave32", "rowgroup32", "rowwave", "fast",
 * "generic", "tile"; "" before the first call): which of the schedules a CSR / width was routed to -- for tests and tuning. */
const char* mgx_last_spmm_kernel(void);
/* Fills *num_cus / *lds_bytes of the current HIP device; MGX_ERR_HIP when no GPU is usable. */
int32_t mgx_device_info(int32_t* num_cus, int32_t* lds_bytes_per_cu, char* arch_name, int32_t arch_name_len);

/* ------------------------------------------------------------------ g-SpMM
 * Replaces _CAPI_DGLKernelSpMM as reached by dgl.ops.gspmm (kernel/dgl-new.py:20) and
 * update_all(fn.copy_src, fn.mean) (main_dgl_product_sage.py:62), update_all(fn.u_mul_e, fn.sum)
 * (GATConv, main_dgl_reddit_gat.py:10; main_dgl_proteins_rgcn_for.py:52), update_all(udf, fn.sum)
 * -> copy_e/sum (main_dgl_molhiv_gcn.py:46).
 *
 *   out[v,k] = dst_scale[v] * REDUCE_{p in row v} op( src_scale[u] * U[u, u_off[k]], E[eid(p), e_off[k]] )
 *
 * op: ADD, MUL, COPY_LHS, COPY_RHS (callers rewrite SUB/DIV as ADD(-E)/MUL(1/E), as DGL does).
 * reduce: SUM, MEAN (sum / max(in_degree,1), the divide DGL performs after the kernel),
 *         MAX / MIN (empty rows give 0; arg_u/arg_e receive the winning source node / edge id,
 *         -1 for empty rows; first extremum in storage order wins).
 * src_scale / dst_scale: optional per-node fp32 factors (NULL = 1); used for the fused backward
 * of `mean` and for symmetric-norm GCN layers.  Only with SUM/MEAN.
 * arg_u/arg_e: same index width as the graph, [num_rows*out_len], may be NULL.
 * NULL offset table: identity when the operand row has out_len elements, otherwise head-wise
 * broadcast k -> k / (out_len / len)  (the (N,H,F) x (E,H,1) pattern of GATConv).
 * Deterministic: no atomics, fixed summation order for a given graph (and plan). */
int32_t mgx_spmm_csr(const mgx_csr* csr, const mgx_spmm_plan* plan /* may be NULL */,
                     int32_t op, int32_t reduce,
                     const float* ufeat, const float* efeat,
                     int64_t u_len, int64_t e_len, int64_t out_len,
                     const int64_t* u_off, const int64_t* e_off,
                     const float* src_scale, const float* dst_scale,
                     float* out, void* arg_u, void* arg_e,
                     float* partial_ws /* [plan->num_slots * out_len] or NULL */,
                     int32_t flags /* MGX_SPMM_ACCUMULATE: out += result (SUM/MEAN only) */, void* stream);

/* Row-sparse gathered matrix (round 2).  The gradient of a loss taken on a subset of the nodes (ogbn-products trains on 8 %,
 * main_dgl_product_sage.py:105-106) is zero in every other row, so the backward aggregation of the last layer gathers rows that
 * are 92 % zeros.  mgx_row_nonzero_bits flags the non-zero rows of x [n, D] (bit r of a ceil(n / 32)-word bitmap, n padded to 64
 * rows); mgx_spmm_copy_u_masked is mgx_spmm_csr(COPY_LHS, SUM | MEAN) that skips gathers of rows whose bit is clear -- the same
 * sum, fewer terms (exact: the skipped terms are zeros).  32-bit indices, D % 4 == 0, operands below 4 GiB, else
 * MGX_ERR_UNSUPPORTED (call mgx_spmm_csr).  plan / partial_ws / flags as for mgx_spmm_csr. */
int32_t mgx_row_nonzero_bits(int64_t n, int64_t D, const float* x, uint32_t* bits /* [2 * ceil(n / 64)] */, void* stream);
int32_t mgx_spmm_copy_u_masked(const mgx_csr* csr, const mgx_spmm_plan* plan /* may be NULL */, int32_t reduce,
                               const float* ufeat, int64_t D, const uint32_t* src_bits, const float* dst_scale /* may be NULL */,
                               float* out, float* partial_ws, int32_t flags, void* stream);

/* copy_u / sum | mean with STRIDED rows (round 2): ufeat rows are u_stride floats apart, out rows out_stride floats apart
 * (both >= D).  Lets the aggregation read and write column blocks of wider matrices in place -- SAGEConv
 * (main_dgl_product_sage.py:61-64) feeds `fc_self(h) + fc_neigh(neigh)`; with h and neigh as the two halves of one [N, 2 D]
 * matrix that is ONE GEMM against the stacked weights instead of two, and the backward aggregation accumulates straight
 * into the h-half of the gradient.  32-bit indices, D and both strides multiples of 4, 16-byte aligned pointers, gathered
 * matrix below 4 GiB; otherwise MGX_ERR_UNSUPPORTED (make the operands dense and call mgx_spmm_csr).  plan / partial_ws / flags
 * as for mgx_spmm_csr. */
int32_t mgx_spmm_copy_u_strided(const mgx_csr* csr, const mgx_spmm_plan* plan /* may be NULL */, int32_t reduce,
                                const float* ufeat, int64_t D, int64_t u_stride, const float* dst_scale /* may be NULL */,
                                float* out, int64_t out_stride, float* partial_ws, int32_t flags, void* stream);

/* copy_u / sum | mean of a CONSTANT matrix of exactly 100 columns (round 5; csrc/spmm_tail.inc) -- the layer-1 aggregation of
 * ogbn-products' input features (main_dgl_product_sage.py:61-62), the largest launch of the epoch.  The row kernel's time is the number of
 * cache-line requests per edge, and a 400-byte row costs a fourth (4.125 on average) for its last 16 bytes.  The features never change, so
 * they are laid out ONCE as
 *     block_a    [num_cols, 96]  columns 0 .. 95, compact: 384-byte rows = three lines each (128-byte aligned)
 *     edge_tail  [nnz, 4]        edge_tail[p] = x[indices[p], 96 .. 99] for every stored position p of the CSR (mgx_edge_tail_fill)
 * and the lane that streams position p's id reads edge_tail[p] in the same coalesced pass: three gathered lines per edge + a 16-byte
 * stream.  out: [num_rows, 100] (row stride out_stride), columns 0 .. 95 bit-identical to mgx_spmm_copy_u_strided of x, 96 .. 99 equal
 * to fp32 rounding (another order of additions).  The tail belongs to ONE CSR (its position order) and ONE x.
 * int32 graphs, one schedule (no two-part plan, no MGX_SPMM_SHORT_ROWS), block_a below 4 GiB; otherwise MGX_ERR_UNSUPPORTED. */
int32_t mgx_edge_tail_fill(const mgx_csr* csr, const float* x /* [num_cols, 100] */, int64_t x_stride, float* edge_tail /* [nnz, 4] */,
                           void* stream);
int32_t mgx_spmm_copy_u_edge_tail(const mgx_csr* csr, const mgx_spmm_plan* plan /* may be NULL */, int32_t reduce, const float* block_a,
                                  const float* edge_tail, const float* dst_scale /* may be NULL */, float* out, int64_t out_stride,
                                  float* partial_ws, int32_t flags, void* stream);

/* copy_u / sum | mean over MOSTLY-ZERO rows of exactly 64 columns (round 5; csrc/spmm_slots.inc) -- the input of a hidden GraphSAGE
 * layer is dropout(relu(.)) (main_dgl_product_sage.py:93-96; 20-25 % non-zero on the benchmark model), and the wave-per-item g-SpMM is
 * bound by the bytes of the gathered rows (L2 -> CU).  mgx_rows_slots_pack turns x [n, 64] into one 128-BYTE SLOT per row:
 *     8 x { meta, v0, v1, v2 } (uint32, float, float, float), meta = c0 | c1 << 8 | c2 << 16 | flag << 24
 *     the row's non-zero values in increasing order of (column % 4) * 16 + column / 4 -- the (component, lane) order of a 16-lane x float4
 *     read --, three per group, c = their columns, 64 = no value (the value word is 0); at most 24 per row; a row with MORE has
 *     flag = 255 in all eight metas (columns 64, values 0): the consumer reads the dense row instead.  *overflow_rows (device, may be
 *     NULL) += the number of such rows.
 *     row_scale (may be NULL): the slots hold row_scale[r] * x[r, :] (the dense matrix is left as it is).
 * mgx_spmm_copy_u_slots is mgx_spmm_copy_u_strided with the slots beside the dense matrix (same rows: slots = pack(ufeat, src_scale)):
 * every work item gathers ONE cache line per edge and adds its values at their columns -- per lane group of 8 lanes a 64-float
 * accumulator row in LDS, updated by plain read - add - write in a fixed order; a wave per item (a whole schedule, the `rest` part of a
 * two-part plan: the eight rows are summed at the end) or a lane group per item (MGX_SPMM_SHORT_ROWS, the head of a two-part plan).
 *     out[v] (+)= dst_scale[v] * SUM | MEAN_{u -> v} src_scale[u] * ufeat[u, :]
 * src_scale (may be NULL) must be the row_scale the slots were packed with: the kernel applies it only to the rows it reads from ufeat
 * (those above 24 non-zeros).  The reversed aggregation of a mean layer is of this form: A^T (D^-1 dy) with dy the gradient behind a
 * relu + dropout.  Exact (values are moved, never rounded; the products by row_scale are the ones the dense kernels form); the order of
 * additions inside a row differs from the dense kernels', so results agree to fp32 rounding, not bit for bit.  products-shaped graph,
 * D = 64, 21 % non-zero: 2.15 -> 1.49 ms per call.
 * int32 graphs, 16-byte aligned operands below 4 GiB (and fewer than 2^25 source rows); otherwise MGX_ERR_UNSUPPORTED. */
int32_t mgx_rows_slots_pack(int64_t n, int64_t D /* 64 */, const float* x, int64_t x_stride, const float* row_scale /* [n] or NULL */,
                            void* slots /* [n, 128 bytes] */, int64_t* overflow_rows /* device, may be NULL */, void* stream);
int32_t mgx_spmm_copy_u_slots(const mgx_csr* csr, const mgx_spmm_plan* plan /* may be NULL */, int32_t reduce, const float* ufeat,
                              int64_t D /* 64 */, int64_t u_stride, const void* slots, const float* src_scale /* may be NULL */,
                              const float* dst_scale /* may be NULL */, float* out, int64_t out_stride, float* partial_ws, int32_t flags,
                              void* stream);

/* LDS-staged copy_u / sum | mean for DENSE neighbourhoods (round 3; csrc/spmm_tile.hip).  On graphs with hundreds of in-edges
 * per node (reddit, proteins: kernel/dgl-new.py:61; main_dgl_reddit_sage.py:73-80) destination rows scheduled next to each other
 * share most of their sources; a TILE of consumers * nacc * 4 work items of the schedule is aggregated by one workgroup of
 * consumers + loaders = 16 waves (one per CU) or 8 waves (two per CU) that gathers each of the tile's re-used sources ONCE from L2 into LDS (chunks of 127 rows x 64 columns, LDS-DMA,
 * 4-deep ring, `loaders` loader waves) and serves every edge into it from LDS; sources used once in the tile are gathered
 * directly.  The tables below replace the graph arrays inside the kernel (built once per CSR by the host layer,
 * mi355x_graph/tileplan.py; all device memory, caller-owned):
 *   tile_chunk_ptr [num_tiles+1]            chunks of tile t are [ptr[t], ptr[t+1])
 *   chunk_ids      [num_chunks*128]         source row of every LDS slot, -1 = all-zero row (slot 127 always)
 *   lds_off        [num_tiles*consumers+1]  first superstep of the stream of (tile, consumer wave): its chunks, then rows j
 *   lds_cnt        [num_chunks*consumers*16] uint16 supersteps (4 steps each) of (chunk, wave, accumulator j < nacc)
 *   lds_stream     [(lds_steps+128)*4]       per superstep and lane group one dword = the LDS slots of its four steps
 *                                           (127 = zero row); 128 supersteps of padding behind the end (prefetch)
 *   dir_off        [num_tiles*consumers+1]  the same for the direct part; dir_cnt int32 [num_tiles*consumers*16];
 *   dir_stream     [(dir_steps+128)*16]      per superstep and lane group four source ids, -1 = padding
 *   tile_item      [num_tiles*R]            item_row of the work item at every (wave, j, lane group), INT32_MIN = none
 *   zero_row       [64] floats of zeros
 *   tile_order     [num_tiles] or NULL: the tile each dispatch slot runs (the builder puts an XCD's longest tiles first)
 * Work items are those of `plan` (hub rows stay split: partial_ws / fix-up as for mgx_spmm_csr; plan may be NULL when the tile
 * plan was built over natural rows).  32-bit indices, D and both strides multiples of 4, 16-byte aligned operands, gathered
 * matrix below 4 GiB; otherwise MGX_ERR_UNSUPPORTED (call mgx_spmm_csr).  Deterministic: no atomics. */
typedef struct mgx_tile_plan {
  int64_t num_tiles, num_chunks, lds_steps, dir_steps; /* steps: supersteps without the padding */
  int32_t consumers, nacc, loaders, lanes_log2; /* lanes per row: 0 | 4 = 16 lanes (64-column passes), 3 = 8 (32), 2 = 4 (16): narrow rows,
                                                    256 slots per chunk, consumers = 7, loaders = 1 */
  const int32_t* tile_chunk_ptr;
  const int32_t* chunk_ids;
  const int32_t* lds_off;
  const uint16_t* lds_cnt;
  const uint32_t* lds_stream;
  const int32_t* dir_off;
  const int32_t* dir_cnt;
  const int32_t* dir_stream;
  const int32_t* tile_item;
  const float* zero_row;
  const int32_t* tile_order; /* optional [num_tiles]: dispatch slot -> tile; slots [x * ceil(T/8), ...) run on XCD x in order */
  const int32_t* tile_node; /* optional [num_tiles*R]: the NODE of the item at every position (hub chunks: their row); mgx_gat_tile_* */
  const uint32_t* lds_stream16; /* optional, mgx_gat_tile_* with drop_p > 0: lds_stream with 16-bit entries, slot | rank << 8, rank = the
                                   edge's rank among the parallel edges of its (row, source) pair by edge id, mod 128; such a plan's
                                   dir_stream carries the rank in bits 24-30 of every source id (sources < 2^24) */
} mgx_tile_plan;
int32_t mgx_spmm_tile_copy_u(const mgx_csr* csr, const mgx_spmm_plan* plan /* may be NULL */, const mgx_tile_plan* tile_plan,
                             int32_t reduce, const float* ufeat, int64_t D, int64_t u_stride, const float* dst_scale /* may be NULL */,
                             float* out, int64_t out_stride, float* partial_ws, int32_t flags, void* stream);

/* ------------------------------------------------------------------ g-SDDMM
 * Replaces _CAPI_DGLKernelSDDMM as reached by dgl.ops.gsddmm (kernel/dgl-new.py:39),
 * apply_edges(fn.u_add_v) inside GATConv (main_dgl_reddit_gat.py:10) and fn.u_dot_v
 * (link_prediction/gcmc_dgl/model.py:342).
 *
 *   out[e,k] = op( L[t_l(e), l_off[k]], R[t_r(e), r_off[k]] ),   e = edge id
 *   DOT: out[e,k] = sum_{j<reduce_size} L[.., l_off[k]*reduce_size+j] * R[.., r_off[k]*reduce_size+j]
 *
 * COO form: src/dst are [nnz] in edge-id order. */
int32_t mgx_sddmm_coo(int64_t num_src, int64_t num_dst, int64_t nnz,
                      const void* src, const void* dst, int32_t idx_bits,
                      int32_t op, const float* lhs, const float* rhs,
                      int32_t lhs_target, int32_t rhs_target,
                      int64_t l_len, int64_t r_len, int64_t out_len, int64_t reduce_size,
                      const int64_t* l_off, const int64_t* r_off,
                      float* out, void* stream);

/* The same walk over an edge list kept in ANOTHER order than the edge ids (round 5): src / dst give the q-th WALKED edge, perm[q] its
 * edge id = the output row.  For graphs that hold an in-CSR beside (or instead of) their COO: walked in the CSR's order -- sorted by
 * destination -- one operand row of consecutive edges is the same row and the other has the g-SpMM's locality, whole output rows are
 * scattered by edge id.  int32 ids, element-wise op (no DOT) on u / v operands [rows, feat_len] below 4 GiB each, else
 * MGX_ERR_UNSUPPORTED (call mgx_sddmm_coo / mgx_sddmm_csr).  Same values as mgx_sddmm_coo bit for bit. */
int32_t mgx_sddmm_coo_perm(int64_t num_src, int64_t num_dst, int64_t nnz, const void* src, const void* dst, const void* perm,
                           int32_t idx_bits, int32_t op, const float* lhs, const float* rhs, int32_t lhs_target, int32_t rhs_target,
                           int64_t feat_len, float* out, void* stream);
/* CSR form (graphs restricted to formats(['csr','csc']), main_dgl_product_sage.py:158): walks
 * the in-CSR, t(e)=V is the row, t(e)=U is indices[p], output still addressed by edge id. */
int32_t mgx_sddmm_csr(const mgx_csr* csr, const mgx_spmm_plan* plan /* may be NULL */,
                      int32_t op, const float* lhs, const float* rhs,
                      int32_t lhs_target, int32_t rhs_target,
                      int64_t l_len, int64_t r_len, int64_t out_len, int64_t reduce_size,
                      const int64_t* l_off, const int64_t* r_off,
                      float* out, void* stream);

/* ------------------------------------------------------------------ edge softmax
 * Replaces dgl.nn.functional.edge_softmax(graph, logits) (norm_by='dst') as called by GATConv
 * (main_dgl_reddit_gat.py:31-55).  DGL 0.6 runs 2 SpMM + 2 SDDMM + exp; here one fused
 * row-segmented kernel.  z, a, da, dz: [nnz, H] addressed by edge id.
 *   With a plan, split (hub) rows are handled as chunks (statistics, combine, normalise) and `ws` must
 *   hold (plan->num_slots + plan->num_hubs) * 2 * H floats; without a plan ws may be NULL.
 *   fwd: a[e,h]  = exp(z[e,h]-max_v) / sum_{e'->v} exp(z[e',h]-max_v)
 *   bwd: dz[e,h] = a*da - a * sum_{e'->v}(a*da)                                            */
int32_t mgx_edge_softmax_fwd(const mgx_csr* csr, const mgx_spmm_plan* plan /* may be NULL */, int64_t H,
                             const float* z, float* a, float* ws, void* stream);
int32_t mgx_edge_softmax_bwd(const mgx_csr* csr, const mgx_spmm_plan* plan /* may be NULL */, int64_t H,
                             const float* a, const float* da, float* dz, float* ws, void* stream);

/* GAT attention with the logits fused in (GATConv, main_dgl_reddit_gat.py:31-55):
 *   fwd: a[e,h]  = softmax_{e->v}( leaky_relu(el[u(e),h] + er[v(e),h], negative_slope) )
 *        = apply_edges(fn.u_add_v) -> leaky_relu -> edge_softmax in ONE launch; the [nnz, H] logits are never written.
 *   bwd: de[e,h] = (a*da - a*sum_{e'->v}(a*da)) * leaky_relu'(el[u]+er[v])   (gradient w.r.t. the pre-activation sum;
 *        the caller reduces it over out-edges / in-edges with copy_e g-SpMMs to get d el / d er).
 * el: [num_cols, H], er: [num_rows, H]; a, da, de: [nnz, H] by edge id.  plan / ws as for mgx_edge_softmax_*. */
int32_t mgx_gat_attention_fwd(const mgx_csr* csr, const mgx_spmm_plan* plan /* may be NULL */, int64_t H,
                              const float* el, const float* er, float negative_slope, float* a, float* ws, void* stream);
int32_t mgx_gat_attention_bwd(const mgx_csr* csr, const mgx_spmm_plan* plan /* may be NULL */, int64_t H,
                              const float* el, const float* er, float negative_slope,
                              const float* a, const float* da, float* de, float* ws, void* stream);

/* One GAT layer's message passing with NO E-sized tensor (GATConv between its projection and its bias,
 * main_dgl_reddit_gat.py:10,31-55: apply_edges(fn.u_add_v) -> leaky_relu -> edge_softmax -> attn_drop ->
 * update_all(fn.u_mul_e, fn.sum)):
 *   fwd: out[v,h,:] = sum_{e:(u->v)} keep(e,h)/(1-p) * a[e,h] * feat[u,h,:],  a = softmax_{e->v}(leaky_relu(el[u,h] + er[v,h]))
 *        nstat[v,h,0..3] = (er, row max, 1 / row sum, -) is what the backward rebuilds a[e,h] from; `a` is never written.
 *   bwd: d_er[v,h], then d_feat[u,h,:] and d_el[u,h] (both or neither); `csc` is the in-CSR the forward ran on, `csr` its
 *        transpose (rows = source nodes) whose eids map to the in-CSR's edge ids (NULL eids = positions) so that both walks
 *        regenerate the same dropout bit for an edge; nstat[.,.,3] receives t = <out, d_out> per head.
 * feat [num_cols, H*F], el [num_cols, H], er [num_rows, H], out / d_out [num_rows, H*F], nstat [num_rows, H, 4], 16-byte
 * aligned; F in {4, 8, ..., 256} with H*F <= 256, or ONE head of any width 4 < F <= 256 (e.g. a 41-class output layer: rows moved
 * in dword-aligned 16-byte windows); 32-bit indices (else MGX_ERR_UNSUPPORTED: callers fall back to mgx_gat_attention_* +
 * mgx_spmm_csr).  keep(e,h) = hash(seed, e*H + h) >= p * 2^32 (counter based; drop_p = 0: no mask).
 * workspace: max over the plans passed of mgx_gat_fused_workspace(plan, H, F) bytes (NULL when no plan splits rows).
 * pack_ws (may be NULL): mgx_gat_fused_pack_workspace(num_src, num_dst, H, F) bytes of scratch; when given and the layer is
 *   narrow enough that a node's feature row and its per-head attention terms fit the row's 128-byte lines (one head of
 *   16 or 41 features: yes; 8 x 16: no, the function returns 0), the gathered operands are first packed into rows
 *   [feat | el] (forward, backward destination walk) and [d_out | er, m, 1/s, t] (backward source walk), so that an edge
 *   costs one L2 request instead of two -- these walks are bound by requests, not bytes.  Same results bit for bit.
 * attn_l (may be NULL; round 3): when el IS (feat * attn_l).sum(-1) -- GATConv's own definition -- pass attn_l [H, F] as well:
 *   layers whose rows are not packed (several heads, H*F >= 64: the 8 x 16 layers) then form el[u,h] from the gathered feature
 *   row inside the kernels (4 multiply-adds and a lane swap or two per edge) instead of gathering it -- one L2 request per edge
 *   less; the `el` array is then only differentiated (d_el), not read.  Pass the same attn_l to the backward.
 * Deterministic: no atomics, hub partial sums combined in slot order. */
int64_t mgx_gat_fused_workspace(const mgx_spmm_plan* plan /* may be NULL */, int64_t H, int64_t F);
int64_t mgx_gat_fused_pack_workspace(int64_t num_src, int64_t num_dst, int64_t H, int64_t F);
int32_t mgx_gat_fused_fwd(const mgx_csr* csr, const mgx_spmm_plan* plan /* may be NULL */, int64_t H, int64_t F,
                          const float* feat, const float* el, const float* attn_l /* [H, F] or NULL */, const float* er,
                          float negative_slope, float drop_p, uint64_t seed, float* out, float* nstat, void* workspace,
                          void* pack_ws, void* stream);
int32_t mgx_gat_fused_bwd(const mgx_csr* csc, const mgx_spmm_plan* csc_plan /* may be NULL */, const mgx_csr* csr,
                          const mgx_spmm_plan* csr_plan /* may be NULL */, int64_t H, int64_t F, const float* feat,
                          const float* el, const float* attn_l /* [H, F] or NULL: as passed to the forward */, float negative_slope,
                          float drop_p, uint64_t seed, const float* out, const float* d_out, float* nstat,
                          float* d_feat /* may be NULL with d_el */, float* d_el, float* d_er, void* workspace, void* pack_ws,
                          void* stream);

/* The same three walks over a TILE plan (csrc/gat_tile.inc): one head of 4, 8, 12 or 16 columns on a graph with dense
 * neighbourhoods (main_dgl_reddit_gat.py on reddit).  `tile_plan`: 4 lanes per row (lanes_log2 = 2), 7 + 1 waves, nacc 3 or 4, with
 * tile_node; `plan`: the tile plan's base plan (its hub tables; workspace = mgx_gat_fused_workspace(plan, 1, F)).  Same arguments,
 * results and statistics as mgx_gat_fused_fwd / _bwd, another fp32 summation order (deterministic for a given plan).  attn_drop:
 * the mask bit of an edge is a function of (seed, destination, source, rank among parallel edges) here -- of (seed, edge id) in
 * the row kernels --: use the tile form for all three walks of a layer or for none when drop_p > 0; it needs lds_stream16 in both
 * plans.  Sources below 2^24 (the direct stream's ids are masked to 24 bits).  Other shapes: MGX_ERR_UNSUPPORTED. */
int32_t mgx_gat_tile_fwd(const mgx_csr* csr, const mgx_spmm_plan* plan, const mgx_tile_plan* tile_plan, int64_t H, int64_t F,
                         const float* feat, const float* el, const float* er, float negative_slope, float drop_p, uint64_t seed,
                         float* out, float* nstat, void* workspace, void* pack_ws, void* stream);
int32_t mgx_gat_tile_bwd(const mgx_csr* csc, const mgx_spmm_plan* csc_plan, const mgx_tile_plan* csc_tile_plan, const mgx_csr* csr,
                         const mgx_spmm_plan* csr_plan, const mgx_tile_plan* csr_tile_plan, int64_t H, int64_t F, const float* feat,
                         const float* el, float negative_slope, float drop_p, uint64_t seed, const float* out, const float* d_out,
                         float* nstat, float* d_feat /* may be NULL with d_el */, float* d_el, float* d_er, void* workspace,
                         void* pack_ws, void* stream);

/* ------------------------------------------------------------------ GAT attention terms
 * el[n,h] = sum_f feat[n,h,f] * attn[h,f] -- GATConv's `(feat * attn_l).sum(-1)` (main_dgl_reddit_gat.py:10, UPSTREAM
 * dgl.nn.pytorch.GATConv.forward), one pass; attn_b/out_b (may be NULL) apply a second attention vector to the same
 * feat in that pass (full-graph GAT: el and er share feat).  Needs H*F <= 256 and F <= 64 or F in {128, 256}
 * (else MGX_ERR_UNSUPPORTED).  Backward: d_feat = d_out_a*attn_a (+ d_out_b*attn_b) written once (d_feat may be NULL),
 * d_attn = column sums of d_out*feat, two-stage in fixed order; workspace of mgx_head_dot_bwd_workspace(H, F) bytes. */
int32_t mgx_head_dot_fwd(int64_t n, int64_t H, int64_t F, const float* feat, const float* attn_a,
                         const float* attn_b, float* out_a, float* out_b, void* stream);
int64_t mgx_head_dot_bwd_workspace(int64_t H, int64_t F);
int32_t mgx_head_dot_bwd(int64_t n, int64_t H, int64_t F, const float* feat, const float* attn_a,
                         const float* attn_b, const float* d_out_a, const float* d_out_b, float* d_feat,
                         float* d_attn_a, float* d_attn_b, void* workspace, void* stream);

/* ------------------------------------------------------------------ segment reduce
 * Replaces dgl.nn.AvgPooling / dgl.ops.segment_reduce (main_dgl_molhiv_gcn.py:75,93).
 * offsets: int64 [num_segments+1] (cumsum of batch_num_nodes).  reduce: SUM, MEAN, MAX, MIN.
 * arg (int64 [num_segments*D], MAX/MIN only) may be NULL. */
int32_t mgx_segment_reduce(int64_t num_segments, const int64_t* offsets, int64_t D, int32_t reduce,
                           const float* x, float* out, int64_t* arg, void* stream);

/* Column sum out[c] = sum_r x[r, c] of a row-major [n, C] matrix, C <= 256: the single-segment case of the readout
 * above (a segment of millions of rows has no row parallelism), used for the bias gradient of the dense layer that
 * follows every aggregation (`nn.Linear` in SAGEConv, main_dgl_product_sage.py:31-33).  Two stages in fixed order
 * (deterministic); workspace of mgx_column_sum_workspace(C) bytes. */
int64_t mgx_column_sum_workspace(int64_t C);
int32_t mgx_column_sum(int64_t n, int64_t C, const float* x, float* out, void* workspace, void* stream);

/* Two column reductions in one pass over a row-major [n, C] matrix (C <= 256): mode 0 -> (sum a, sum a*a), mode 1 ->
 * (sum a, sum a*b) -- the statistics of BatchNorm1d over the node dimension (main_dgl_arxiv_sage.py:70-77) forward and
 * backward; modes 2 / 3 are the same sums taken relative to the first row p of the squared / second operand, mode 2 ->
 * (sum (a-p), sum (a-p)^2) with p = a[0,:], mode 3 -> (sum a, sum a*(b-p)) with p = b[0,:], which keeps the variance
 * accurate when |mean| >> std; workspace of 2 * mgx_column_sum_workspace(C) bytes.  mgx_column_affine: out[r,c] = a[r,c]*A[c] + b[r,c]*B[c] +
 * Cc[c] (b/B may be NULL), C % 4 == 0, 16-byte aligned: the normalisation and its input gradient. */
int32_t mgx_column_pair_sums(int64_t n, int64_t C, int32_t mode, const float* a, const float* b, float* out0, float* out1,
                             void* workspace, void* stream);
int32_t mgx_column_affine(int64_t n, int64_t C, const float* a, const float* b, const float* A, const float* B, const float* Cc,
                          float* out, void* stream);

/* out[M, K] = a^T b for a [n, M] (row stride lda floats), b [n, K] (row stride ldb), out row stride ldc, n in the hundreds of
 * thousands to millions: the weight gradient dW = dY^T X of the dense layer after an aggregation
 * (main_dgl_product_sage.py:31-33,64).  M <= 64 and K <= 128: one tile, operands streamed once.  Up to M <= 256, K <= 1024
 * (round 2): the grid of 64 x 128 tiles in ONE launch, the tiles' waves walking the same rows together so that operand rows
 * are shared through L2 (else MGX_ERR_UNSUPPORTED).  fp32 MFMA, per-wave partial tiles added in fixed order
 * (deterministic); workspace of mgx_xty_workspace(M, K) bytes (-1 if the shape is unsupported). */
int64_t mgx_xty_workspace(int64_t M, int64_t K);
int32_t mgx_xty(int64_t n, int64_t M, int64_t K, const float* a, int64_t lda, const float* b, int64_t ldb, float* out, int64_t ldc,
                void* workspace, void* stream);
/* mgx_xty that also returns the column sums of a (colsum [M]; the bias gradient beside the weight gradient) from the same pass over a,
 * as the product with one more column of ones.  One 64 x 128 tile, b read with 16-byte loads (K % 4 == 0, K >= 64, ldb % 4 == 0,
 * 16-byte aligned), else MGX_ERR_UNSUPPORTED (mgx_xty + mgx_column_sum).  Same workspace. */
int32_t mgx_xty_colsum(int64_t n, int64_t M, int64_t K, const float* a, int64_t lda, const float* b, int64_t ldb, float* out, int64_t ldc,
                       float* colsum, void* workspace, void* stream);

/* C[n, M] = A[n, K] x B (+ bias[M]) for A with millions of rows and small K, M -- the dense projections either side of an aggregation
 * (SAGEConv's fc_self / fc_neigh on [h | mean_agg(h)], main_dgl_product_sage.py:23-24,64, and the input gradient d[h | neigh] = dY x W).
 * B: [K, M] row-major with ldb, or -- b_transposed -- [M, K] (torch.nn.Linear's weight layout).  row_scale (may be NULL): C[r, m] is
 * multiplied by row_scale[r] for the columns m >= scale_from (the 1 / deg of fn.mean's backward on the `neigh` half).  fp32 MFMA, fixed
 * summation order.  K x M padded to 16 must fit a 64 KB stage and be one of the built shapes, else MGX_ERR_UNSUPPORTED (use a GEMM). */
int32_t mgx_rows_gemm(int64_t n, int64_t K, int64_t M, const float* a, int64_t lda, const float* b, int64_t ldb, int32_t b_transposed,
                      const float* bias /* [M] or NULL */, const float* row_scale /* [n] or NULL */, int64_t scale_from, float* c,
                      int64_t ldc, float* c2 /* NULL, or the matrix that receives the columns split_col .. M - 1 (as its columns 0 ..) */,
                      int64_t ldc2, int64_t split_col /* a multiple of 4 */, void* stream);
/* 1 when mgx_rows_gemm has a kernel for these sizes (lda: A's row stride in floats), else 0. */
int32_t mgx_rows_gemm_supported(int64_t K, int64_t M, int64_t lda);
/* The same product followed by relu and inverted dropout (main_dgl_product_sage.py:93-95) in the epilogue: y[r, :] (row stride ldy) and the
 * 4 mask bits per float4 exactly as mgx_relu_dropout_fwd_strided(p, seed, offset) would produce them from the stored product -- which is
 * never written.  M % 4 == 0, y 16-byte aligned, ldy % 4 == 0; mask: [n * M / 4] bytes.
 * slots (optional, M == 64 only): the rows of y as 128-byte slots as well, byte for byte what mgx_rows_slots_pack(y) would write (the
 * epilogue holds a row exactly as that pass reads it), *overflow_rows (device, may be NULL) += rows with more than 24 non-zeros -- the
 * next layer's aggregation (mgx_spmm_copy_u_slots) then needs no pack pass. */
int32_t mgx_rows_gemm_relu_dropout(int64_t n, int64_t K, int64_t M, const float* a, int64_t lda, const float* b, int64_t ldb,
                                   int32_t b_transposed, const float* bias /* [M] or NULL */, float p, uint64_t seed, uint64_t offset,
                                   float* y, int64_t ldy, uint8_t* mask, void* slots /* [n, 128 bytes] or NULL */,
                                   int64_t* overflow_rows /* or NULL */, void* stream);

/* y = dropout_p(relu(x)) in one pass (inverted dropout: kept values scaled by 1/(1-p)); the activation between two
 * aggregations (main_dgl_product_sage.py:93-95).  n elements, n % 4 == 0, 16-byte aligned; mask: n/4 bytes, 4 bits per
 * float4 = (x > 0 AND kept), all that backward needs: dx = mask ? dy/(1-p) : 0.  Random bits are a counter-based function
 * of (seed, offset + element index): the caller advances `offset` by n between calls. */
int32_t mgx_relu_dropout_fwd(int64_t n, const float* x, float p, uint64_t seed, uint64_t offset, float* y, uint8_t* mask,
                             void* stream);
int32_t mgx_relu_dropout_bwd(int64_t n, const float* dy, const uint8_t* mask, float p, float* dx, void* stream);
/* Row-strided forms (round 2): [rows, cols] views whose rows are *_stride floats apart (cols, strides % 4 == 0); the mask
 * stays dense and equals the dense call's for the same (seed, offset). */
int32_t mgx_relu_dropout_fwd_strided(int64_t rows, int64_t cols, const float* x, int64_t x_stride, float p, uint64_t seed,
                                     uint64_t offset, float* y, int64_t y_stride, uint8_t* mask, void* stream);
int32_t mgx_relu_dropout_bwd_strided(int64_t rows, int64_t cols, const float* dy, int64_t dy_stride, const uint8_t* mask, float p,
                                     float* dx, int64_t dx_stride, void* stream);
/* mgx_relu_dropout_bwd_strided for rows of exactly 64 columns that ALSO writes the result's rows, times row_scale[r] when given, as
 * 128-byte slots (mgx_rows_slots_pack's format, byte for byte what that pass would write from row_scale * dx): the gradient behind
 * relu + dropout is as sparse as the activation, and the reversed aggregation that follows gathers the slots (mgx_spmm_copy_u_slots). */
int32_t mgx_relu_dropout_bwd_slots(int64_t rows, const float* dy, int64_t dy_stride, const uint8_t* mask, float p, float* dx,
                                   int64_t dx_stride, const float* row_scale /* [rows] or NULL */, void* slots /* [rows, 128 bytes] */,
                                   int64_t* overflow_rows /* device, may be NULL */, void* stream);
/* Forward whose offset into the random stream is read from device memory when the launch RUNS (offset = *counter *
 * 0x9E3779B97F4A7C15 mod 2^63): captured in a HIP graph next to an increment of the counter, every replay draws a new mask. */
int32_t mgx_relu_dropout_fwd_counter(int64_t rows, int64_t cols, const float* x, int64_t x_stride, float p, uint64_t seed,
                                     const uint64_t* counter, float* y, int64_t y_stride, uint8_t* mask, void* stream);

/* ------------------------------------------------------------------ formats (integer, bit-exact)
 * Replace the lazy COO->CSR/CSC construction behind g.formats(...)/first kernel call
 * (main_dgl_product_sage.py:158, kernel/dgl-new.py:63) and g.in_degrees()
 * (main_dgl_molhiv_gcn.py:41).
 *
 * mgx_coo_to_csr: stable sort of the COO by `row` (edge-id order kept inside a row).
 *   For the in-CSR pass row = dst, col = src.  Outputs indptr[num_rows+1], indices[nnz],
 *   eids[nnz] in the same index width.  `workspace` of mgx_coo_to_csr_workspace() bytes. */
int64_t mgx_coo_to_csr_workspace(int64_t num_rows, int64_t nnz, int32_t idx_bits);
int32_t mgx_coo_to_csr(int64_t num_rows, int64_t nnz, const void* row, const void* col, int32_t idx_bits,
                       void* indptr, void* indices, void* eids,
                       void* workspace, int64_t workspace_bytes, void* stream);
/* Transpose of a CSR (the out-CSR from the in-CSR or back; UPSTREAM aten::CSRTranspose behind g.formats / the reversed
 * graph of every SpMM backward): rows of the result = columns of `csr`, entries of a row in EDGE-ID order -- bit-identical
 * to mgx_coo_to_csr on the graph's COO, so both formats address the same edge tensors and reduce in the same order.
 * csr->eids must be a permutation of [0, nnz) or NULL (= positions).  Outputs indptr_t[num_cols+1], indices_t[nnz],
 * eids_t[nnz]; workspace of mgx_csr_transpose_workspace() bytes. */
int64_t mgx_csr_transpose_workspace(int64_t num_cols, int64_t nnz, int32_t idx_bits);
int32_t mgx_csr_transpose(const mgx_csr* csr, void* indptr_t, void* indices_t, void* eids_t,
                          void* workspace, int64_t workspace_bytes, void* stream);
/* in_degrees / out_degrees from a CSR: deg[v] = indptr[v+1]-indptr[v] (graph index width). */
int32_t mgx_csr_degrees(int64_t num_rows, const void* indptr, int32_t idx_bits, void* deg, void* stream);
/* inv_deg[v] = 1 / max(deg,1) as fp32 (the factor of fn.mean and of its backward). */
int32_t mgx_csr_inv_degrees(int64_t num_rows, const void* indptr, int32_t idx_bits, float* inv_deg, void* stream);

/* Host-side (CPU pointers, no GPU needed) stable COO->CSR used when a graph is prepared on the
 * CPU before .to(device) (main_dgl_product_sage.py:158 does formats() before to()). */
int32_t mgx_coo_to_csr_host(int64_t num_rows, int64_t nnz, const void* row_host, const void* col_host,
                            int32_t idx_bits, void* indptr_host, void* indices_host, void* eids_host);

/* ------------------------------------------------------------------ halo exchange helpers (multi-GPU)
 * New capability (the reference is single-GPU): pack boundary rows for the RCCL all_to_all and
 * add received gradient rows back into their owners.
 *   gather_rows:      out[i,:]      = x[idx[i],:]
 *   scatter_add_rows: x[idx[i],:]  += in[i,:]   (idx sorted & unique per call => no atomics) */
int32_t mgx_gather_rows(int64_t n, const void* idx, int32_t idx_bits, int64_t D,
                        const float* x, float* out, void* stream);
/* The same gather over row-strided operands (round 4): x rows x_stride floats apart, out rows out_stride floats apart (both >= D,
 * multiples of 4, 16-byte aligned pointers; else MGX_ERR_UNSUPPORTED) -- packs the boundary rows straight out of a column block of a
 * wider matrix (the left half of the one-GEMM SAGE layer's [h | neigh] buffer). */
int32_t mgx_gather_rows_strided(int64_t n, const void* idx, int32_t idx_bits, int64_t D, const float* x, int64_t x_stride,
                                float* out, int64_t out_stride, void* stream);
int32_t mgx_scatter_add_rows(int64_t n, const void* idx, int32_t idx_bits, int64_t D,
                             const float* in, float* x, void* stream);

/* Halo rows as bitmaps + packed values (round 5, csrc/rowpack.hip): the exchange of relu + dropout outputs, ~75 % exact zeros
 * (main_dgl_product_sage.py:93-96), moves a 64-bit mask per 64 columns + the non-zero values forward and the values under the SAME
 * mask back.  masks: [n, mgx_rows_mask_words(D)] uint64; bit (c * G + l) of a block's word = column 64 * block + 4 l + c, G = min(16,
 * pow2 >= D / 4); a row's values are stored in increasing bit order, block after block, from offsets[i] on (the caller's exclusive
 * scan of `counts`).  idx: row ids of x (NULL = identity).  D % 4 == 0, D <= 256, strides multiples of 4, 16-byte aligned, else
 * MGX_ERR_UNSUPPORTED.  Values are moved, never rounded: unpack(pack(x)) == x bit for bit. */
int64_t mgx_rows_mask_words(int64_t D);
int32_t mgx_rows_pack_count(int64_t n, const void* idx, int32_t idx_bits, int64_t D, const float* x, int64_t x_stride,
                            uint64_t* masks, int32_t* counts /* [n] non-zeros per row */, void* stream);
int32_t mgx_rows_mask_count(int64_t n, int64_t D, const uint64_t* masks, int32_t* counts, void* stream);
int32_t mgx_rows_pack_values(int64_t n, const void* idx, int32_t idx_bits, int64_t D, const float* x, int64_t x_stride,
                             const uint64_t* masks, const int64_t* offsets /* [n] */, float* values, void* stream);
int32_t mgx_rows_unpack(int64_t n, int64_t D, const uint64_t* masks, const int64_t* offsets, const float* values,
                        float* out /* [n, D] rows out_stride apart; zeros where the mask is clear */, int64_t out_stride, void* stream);
/* out[v, :] += sum over the entries p = positions[q], q in [indptr[v], indptr[v+1]), of the packed row p (int32 CSR over the n output rows;
 * entries added in CSR order: deterministic): the returned halo-row gradients added into their owners without a dense intermediate.
 * `values` holds fewer than 2^32 floats (offsets are used as 32-bit numbers on the D <= 64 path). */
int32_t mgx_rows_unpack_add_csr(int64_t n, const int32_t* indptr, const int32_t* positions, int64_t D, const uint64_t* masks,
                                const int64_t* offsets, const float* values, float* out, int64_t out_stride, void* stream);

/* ------------------------------------------------------------------ neighbor sampling
 * Replaces the CPU-side sampling of dgl.dataloading.MultiLayerNeighborSampler / dgl.sampling.sample_neighbors
 * (end_to_end/sampling/node-classification/reddit/ns-sage-dgl.py:132-141): for every seed v (a row of the in-CSR)
 * keep all in-edges when in_degree(v) <= fanout, else `fanout` distinct in-edges drawn uniformly (Floyd's
 * algorithm, counter-based generator: the same rng_seed gives the same sample).  The caller computes
 * out_offsets = exclusive cumsum of min(in_degree, fanout) ([num_seeds+1], int64) and allocates out_src / out_eid
 * (graph index width, out_offsets[num_seeds] entries).  Picks of one seed are written in CSR order.
 * fanout in [1, 64]. */
int32_t mgx_sample_neighbors(const mgx_csr* csr, int64_t num_seeds, const void* seeds, int32_t fanout,
                             uint64_t rng_seed, const int64_t* out_offsets, void* out_src, void* out_eid,
                             void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MI355X_GRAPH_H_ */
