"""Settings of the host layer.

USER switches are environment variables, read where they are used; the complete list is the table in README.md ("Switches"):
    MGX_LIB_PATH  MGX_DATA_ROOT  MGX_CACHE_DIR  MGX_SCHEDULE  MGX_TILE  MGX_GAT_TILE  MGX_GAT_FUSED  MGX_TORCH_OPS  MGX_SDDMM_WALK
    MGX_SPARSE_HALO  MGX_ACCELERATE_LINEAR  MGX_SAGE_SPARSE_LAST  MGX_SAGE_L1_PROJECT_FIRST  MGX_PLAIN_MODEL  MGX_HOST_THREADS
(+ the bench / test hooks MGX_DIST_BACKEND, MGX_BENCH_SHARE_GPU, MGX_BENCH_TUNABLEOP, MGX_BENCH_STEP_TIMES, MGX_DATASET_SCALE).

Everything below is NOT a user switch: module attributes with fixed defaults that choose between a fused form and the composition it
stands for (both kept: the composition is the fallback for shapes the fused kernel does not take, and the tests compare the two), or
hold a measured constant.  Tests flip them with monkeypatch.setattr(config, ...).  Variants that were measured and lost (per-slot LDS
flags, shrinking tail tiles, the tile kernel's direct part as its own launch, longest-tile-first dispatch, 8 waves per SIMD, the own-matrix
backward, ...) are not switches any more: their code is gone, their numbers are in docs/LOG_r0*.md.
"""

# ---- SAGEConv forms (ops.py, dist.py, full_graph.py)
SAGE_FUSED_LAYER = True    # a mean-aggregator layer as ONE autograd node (ops.SageMeanLayerFn / SageMeanCatFn)
SAGE_CAT = True            # ... over an [h | neigh] buffer: one GEMM per layer (ops.CatBuffer)
SAGE_STATIC_CAT = True     # an unmodified input tensor stays resident in that buffer (identity + version counter)
SAGE_FUSED_ADD = True      # fc_self(h) + fc_neigh(neigh) as two GEMMs, the second accumulating (ops.linear_sum)
SAGE_FUSED_ACT = True      # relu + dropout in the layer GEMM's epilogue (mgx_rows_gemm_relu_dropout)
SAGE_PROJECT_FIRST = True  # project before aggregating when that moves fewer columns (reddit: 602 -> 16)
ROWS_GEMM = True           # mgx_rows_gemm for the projections of a layer over >= 65 536 rows (else the library GEMM)
LINEAR_XTY = True          # tall-skinny weight gradients by mgx_xty (else the library GEMM)
XTY_COLSUM = True          # ... with the bias gradient from the same pass (mgx_xty_colsum)

# ---- GAT (nn.py, sparse.py)
GAT_AGG_FIRST = True       # aggregate before projecting when in_feats < heads * out_feats
GAT_PACK = True            # one packed gather operand per node for the fused walks (mgx_gat_fused_pack_workspace); False: separate arrays

# ---- schedules (schedule.py, tileplan.py)
PLAN_BUILDER = "device"    # "device" (mgx_spmm_plan_count / _fill) | "torch" | "host": the same tables three ways (tests compare them)
HUB_SPLIT = 256            # rows longer than this become several work items (256 measured best on MI355X; 1024: +5..10 %)
LP_ROUNDS = 8              # label-propagation rounds of the locality order
CLUSTER_MIN_NNZ = 4_000_000   # smaller graphs keep their natural row order
TILE_HUB_SPLIT = 2048      # hub threshold of the tile kernel's work items
TILE_CONFIG = {4: (7, 6, 1, 3), 3: (7, 3, 1, 3), 2: (7, 3, 1, 2)}   # lanes_log2 -> (consumers, nacc, loaders, tau)
GAT_TILE_CONFIG = (7, 3, 1, 2)
TILE_MIN_WIDTH = 4         # rows of 4 .. 20 columns take the 16-column tile pass, 24 .. 44 the 32-column one, wider the 64-column one
TILE_VALIDATE = True       # check every tile plan's tables on the device before a kernel may walk them (a dozen reductions per plan)
SHORT_ROWS_IMBALANCE = 6.0    # two-part plans: sum over batches of max(len) * B may exceed the short edges by at most this factor

# ---- row-sparse gradients, wide rows (sparse.py)
SPARSE_GRAD = False        # skip all-zero rows of a gradient in the reversed aggregation (measured: no gain on dense gradients)
SPARSE_GRAD_MIN_NNZ = 2_000_000
WIDE_PAD = True            # rows wider than 256 columns on dense graphs: line-padded passes
WIDE_PAD_MIN_NNZ = 1 << 20

# ---- mostly-zero rows as 128-byte slots (ops._packed_rows, csrc/spmm_slots.inc)
PACKED_GATHER = True       # the forward aggregation of a [N, 64] relu + dropout output gathers one 128-byte slot per edge instead of the 256-byte row
PACKED_GATHER_MIN_NNZ = 20_000_000   # products shape x 0.1 / 0.2 / 0.3 / 0.5 / 1 (12 .. 124 M edges): dense 0.21 / 0.39 / 0.59 / 1.02 / 2.13 ms, slots +
                                     # pack pass 0.21 / 0.36 / 0.53 / 0.82 / 1.62 ms: small operands sit in L2 / MALL, where whole rows are cheap
PACKED_GATHER_PROBE = True          # dgl.ops.gspmm / update_all(copy_u, sum | mean) of an UNTAGGED [N, 64] operand on such a graph: pack it and let the overflow count decide
PACKED_GATHER_MAX_OVERFLOW = 0.10    # share of rows with more than 24 non-zeros (read from the dense matrix) above which the dense kernels are used

# ---- a constant 100-column input as [N, 96] + its last four columns along the edge list (ops._edge_tail_operands, csrc/spmm_tail.inc)
EDGE_TAIL = True           # products layer 1: 4.74 -> 4.09 ms per aggregation, for 16 bytes per edge (2 GB) + the [N, 96] copy, laid out once
EDGE_TAIL_MIN_NNZ = 20_000_000
