/*
 * arnoldi_hip.h -- C ABI of libarnoldi_hip.so (MI355X / gfx950).
 *
 * The library is the device half of a drop-in for
 *     arnoldi.krylov_schur.partial_schur          (src/arnoldi/krylov_schur.py:10-114)
 * of cournape/arnoldi-py.  Everything O(n) on that path runs in the HIP kernels
 * behind these entry points; the O(m^3) Schur / reorder step stays on the host
 * (LAPACK through SciPy) exactly as in the reference.
 *
 * Conventions
 *   - Plain C types only.  Every pointer named  d_*  is DEVICE memory owned by
 *     the caller (the Python host allocates it with torch; any hipMalloc'ed
 *     buffer works).  `stream` is a hipStream_t passed as void* (NULL = the
 *     default stream).  All launches are asynchronous on that stream; no entry
 *     point synchronises or allocates, so a sequence of calls can be captured
 *     into a hipGraph.
 *   - complex128 is two consecutive doubles (re, im) -- numpy's layout.
 *   - The Krylov basis V is column-major n x (m+1) with leading dimension ldv
 *     (elements), i.e. numpy order="F" as in krylov_schur.py:42.  H is the
 *     reference's row-major (m+1) x m array (krylov_schur.py:43), ld = ldh.
 *   - Return value: AKS_OK (0) or a negative AKS_ERR_*; aks_last_error() gives
 *     the text (thread-local).  No C++ exception crosses the boundary.
 *   - The library keeps NO mutable global state (the only static datum is the thread-local
 *     error string): independent call sequences may run concurrently from several host
 *     threads, on one device or on several; a workspace / probe / communicator handle belongs
 *     to one sequence at a time.  aks_device_init() must have run once on every device whose
 *     kernels need more than the default 64 KiB of dynamic LDS (the binned SpMV, and
 *     aks_truncate / aks_combine with a large Qp).
 *   - Breakdown (reference: decomposition.py:61-63) is detected ON THE DEVICE:
 *     the control block's `broken` word is set, `n_iter` records j+1, and every
 *     later launch that is handed the same workspace becomes a no-op.  The host
 *     reads the control block once per expansion.
 */
#ifndef ARNOLDI_HIP_H
#define ARNOLDI_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AKS_ABI_VERSION 6

#define AKS_OK 0
#define AKS_ERR_ARG (-1)         /* bad argument (null pointer, size, alignment) */
#define AKS_ERR_HIP (-2)         /* a HIP runtime call / launch failed          */
#define AKS_ERR_UNSUPPORTED (-3) /* size outside what the kernels are built for  */

/* limits of this build */
#define AKS_MAX_DIM 128          /* max_dim (m) supported by the kernels         */
#define AKS_MAX_TRUNC 96         /* restart size p supported by aks_truncate     */
#define AKS_SPMV_TILE_NNZ 256    /* non-zeros per wave tile (aks_csr_plan_tiles) */

typedef struct aks_c128 { double re, im; } aks_c128;

/* Device-resident control block at offset 0 of the workspace (64 bytes).
 * Written only by kernels; the host copies it back after an expansion. */
typedef struct aks_ctrl {
    int32_t broken;        /* 1 once a step ended with beta < tol (breakdown)            */
    int32_t n_iter;        /* j+1 of the step that broke down (valid iff broken)         */
    int32_t steps_done;    /* Arnoldi steps completed since aks_workspace_init           */
    int32_t second_passes; /* how many of them ran the DGKS second pass (ortho.py:101)   */
    double beta_in;        /* ||w|| before orthogonalisation, last step (ortho.py:92)    */
    double beta;           /* ||w|| after orthogonalisation, last step (ortho.py:98/105) */
    int32_t real_mode;     /* 1: real-packed panel (aks_workspace_set_real); read by the reductions */
    int32_t deferred;      /* 1: the last step left its new column raw (AKS_EXPAND_DEFER_SCALE was honoured)  */
    uint32_t ticket[4];    /* arrival counters of the panel kernels whose last workgroup sums the per-workgroup partial
                              rows itself (ABI 4); zero between launches: the last arriver resets its word           */
    double reserved;
} aks_ctrl;

/* Byte offsets of the workspace regions (all 256-byte aligned). */
typedef struct aks_ws_layout {
    int64_t total_bytes;
    int64_t ctrl_off;      /* aks_ctrl                                                   */
    int64_t red1_off;      /* (m+2) c128: [V^H w ; ||w||^2]       first projection       */
    int64_t red2_off;      /* (m+2) c128: [V^H w'; ||w'||^2]      re-projection          */
    int64_t red3_off;      /* 2 c128:     [||w''||^2]             norm after 2nd update  */
    int64_t partial_off;   /* n_blocks x ld_partial c128 per-block partial sums          */
    int32_t n_blocks;      /* row blocks used by the reduction kernels                   */
    int32_t ld_partial;    /* m + 2                                                      */
    int32_t red_len;       /* m + 2 (c128 elements in red1 / red2)                       */
    int32_t pad_;
    int64_t colscale_off;  /* AKS_MAX_DIM + 2 doubles: scale of basis column c, 0 = the column is normalised
                              (deferred normalisation, AKS_EXPAND_DEFER_SCALE); zeroed by aks_workspace_init  */
} aks_ws_layout;

const char *aks_last_error(void);
int32_t aks_abi_version(void);
/* Raises the dynamic-LDS limit (hipFuncAttributeMaxDynamicSharedMemorySize, a per-device, per-kernel
 * attribute) of every kernel that may use up to 160 KiB, on the CURRENT device.  Call it once per device
 * and process before the first aks_pb_spmv* / aks_truncate / aks_combine there; calling it again is
 * harmless, and it may be called from any thread.  Without it those launches fail with AKS_ERR_HIP. */
int aks_device_init(void);

/* ---- workspace ---------------------------------------------------------- */
/* Pure host computation of the layout for a local row count and max_dim. */
int aks_workspace_layout(int64_t n_rows, int32_t max_dim, aks_ws_layout *out);
/* Zeroes the control block and the reduction slots (async on stream). */
int aks_workspace_init(void *d_ws, int64_t ws_bytes, int64_t n_rows, int32_t max_dim, void *stream);

/* ---- operator apply: replaces  w[:] = A @ V[:, j]  (decomposition.py:58) ---- */
/* Host helper.  Cuts the rows of a CSR matrix into wave tiles of at most
 * tile_nnz non-zeros (a longer row gets a tile of its own).  Writes the tile
 * start rows, terminated by n_rows, to tiles_out (host memory, capacity `cap`
 * entries) and returns the number of tiles, or a negative error. */
int64_t aks_csr_plan_tiles(const int32_t *indptr_host, int64_t n_rows, int32_t tile_nnz,
                           int32_t *tiles_out, int64_t cap);

/* y = A x   (accumulate == 0)   or   y += A x   (accumulate != 0).
 * CSR with int32 indptr/indices; values are float64 (values_complex == 0) or
 * complex128; x, y are complex128.  d_tiles / n_tiles come from
 * aks_csr_plan_tiles.  lanes_per_row is 1, 2, 4, ... 64 (0 = choose from the
 * mean row length).  d_ws may be NULL; if given, the launch is a no-op once the
 * control block says `broken`. */
int aks_csr_spmv(int64_t n_rows, const int32_t *d_indptr, const int32_t *d_indices,
                 const void *d_values, int32_t values_complex, const int32_t *d_tiles,
                 int64_t n_tiles, int32_t lanes_per_row, const aks_c128 *d_x, aks_c128 *d_y,
                 int32_t accumulate, const void *d_ws, void *stream);

/* ---- tile-binned two-phase SpMV (same operation, for matrices without column locality) ----
 * When the columns of a row are scattered over all of x (random graphs), every 16-B gather of
 * the CSR kernel misses the 4 MiB L2 of its XCD and the kernel runs at the fabric's request
 * rate (one 128-B request per non-zero).  The binned form trades that for two streaming passes
 * over an intermediate array of products; every global access of both is a coalesced stream:
 *   phase 1  one workgroup per SUB-SLAB of 2^13 columns: the sub-slab's 8192 x entries (128 KiB)
 *            are staged in LDS, the sub-slab's non-zeros -- ordered by (row block, row) -- are
 *            streamed (value, 13-bit local column) and val * x[col] is written SEQUENTIALLY;
 *   phase 2  one workgroup (8 waves) per ROW BLOCK of 2^13 rows with the block's 8192 complex
 *            accumulators in LDS.  The block's products -- the (sub-slab, row block) tiles, each
 *            contiguous in phase-1 order, taken sub-slab by sub-slab -- are cut into WAVE-LOADS of
 *            up to 64 entries: one lane each, gathered from up to THREE contiguous pieces (the tail
 *            of one tile, a whole tile, the head of the next), so that the LDS adds and the product
 *            loads run with (nearly) full waves whatever the tile size.  32 consecutive wave-loads
 *            (4 per wave) form a ROUND.  Two entries of one round that hit the same row from
 *            different waves get different LEVELS (level = number of lower-index waves adding to that
 *            row); the round's LDS adds are issued level by level with a workgroup barrier after each,
 *            a wave's own adds complete in program order, and rounds follow each other in order, so
 *            every row receives its addends in one fixed order: results are bitwise reproducible.  The
 *            (level, row) words of a round are a fixed block of 2048, laid out [wave][lane][k]:
 *            a lane fetches its four words with one 8-byte load.
 * aks_pb_plan_* are pure host functions that build the arrays from a canonical CSR matrix
 * (int32 indices); the caller uploads them and fills aks_pb_matrix with device pointers.     */
#ifndef AKS_PB_SLAB_BITS         /* (overridable at build time for tuning experiments)        */
#define AKS_PB_SLAB_BITS 13      /* columns per sub-slab = 8192 (x slice in LDS: 128 KiB)     */
#endif
#ifndef AKS_PB_ROWBLOCK_BITS
#define AKS_PB_ROWBLOCK_BITS 13  /* rows per phase-2 workgroup = 8192 (accumulators: 128 KiB) */
#endif
#ifndef AKS_PB_WAVES
#define AKS_PB_WAVES 8           /* waves of a phase-2 workgroup                              */
#endif
#ifndef AKS_PB_RUNS_PER_WAVE
#define AKS_PB_RUNS_PER_WAVE 4   /* wave-loads a wave takes per round (round = 32 of them; 8 builds too, for
                                    experiments through the C ABI only: measured slower) */
#endif
#define AKS_PB_RUN_MAX 64        /* entries per wave-load (one lane each)                     */
#define AKS_PB_CHUNKS 256        /* phase-2 workgroups (CUs of an MI355X).  With n_chunks = min(n_rowblocks, 256),
                                    chunk c owns the row blocks c, c + n_chunks, c + 2 n_chunks, ...; d_runs /
                                    d_lrow / d_rb_run_ptr hold the row blocks in THAT order: position p of
                                    d_rb_run_ptr is the k-th row block of chunk c, where the first
                                    n_rowblocks % n_chunks chunks own one row block more than the others     */
#define AKS_PB_PIECES 3          /* contiguous pieces a wave-load gathers from                */
#define AKS_PB_ROUND_WORDS (AKS_PB_WAVES * AKS_PB_RUNS_PER_WAVE * AKS_PB_RUN_MAX)   /* 2048  */

typedef struct aks_pb_run {      /* one wave-load, 16 bytes; n_runs of them, a multiple of 32 per row block.
                                    Lane l < total reads entry (l < l0 ? start0 : l < l01 ? start1 : start2) + l
                                    of the phase-1 order (start1 / start2 are stored minus the lanes before
                                    their piece, modulo 2^32).                                                 */
    uint32_t start0, start1, start2;
    uint32_t info;               /* bits 0-6 l0, 7-13 l01 = l0 + l1, 14-20 total (0 = padding slot),
                                    21-24 levels of the wave-load's round (>= 1), 25 set on the slots of a
                                    row block's LAST round (every row block owns at least one round)           */
} aks_pb_run;

typedef struct aks_pb_matrix {
    int64_t n_rows, n_cols, nnz;
    int64_t nnz_pad;                /* phase-1 slots: every sub-slab starts on a multiple of 8 */
    int64_t n_runs, n_lrow;         /* lengths of d_runs / d_lrow (n_lrow = 64 n_runs)         */
    int32_t n_slabs, n_rowblocks, values_complex, pad_;
    const void *d_val;              /* nnz_pad values (f64 or c128), phase-1 order, pads = 0   */
    const uint16_t *d_lcol;         /* nnz_pad: column - sub-slab * 8192                       */
    const int32_t *d_slab_begin;    /* n_slabs: first phase-1 slot of each sub-slab            */
    const int32_t *d_slab_end;      /* n_slabs: one past its last entry                        */
    const aks_pb_run *d_runs;       /* n_runs wave-load descriptors, row block by row block    */
    const int32_t *d_rb_run_ptr;    /* n_rowblocks + 1: first wave-load per row-block POSITION  */
    const uint16_t *d_lrow;         /* n_lrow = n_runs * 64: level << 13 | row - rowblock * 8192;
                                       round r (= slots 32 r ..) owns words 2048 r + wave * 256 + lane * 4 + k */
    aks_c128 *d_prod;               /* nnz_pad complex128 scratch (the products)               */
} aks_pb_matrix;

typedef struct aks_pb_sizes {       /* array lengths a plan needs (see aks_pb_matrix)          */
    int64_t nnz_pad, n_runs, n_lrow;
    int32_t n_slabs, n_rowblocks;
} aks_pb_sizes;

/* The constants above as compiled into the library: sub-slab bits, row-block bits, runs per round. */
int aks_pb_params(int32_t *slab_bits, int32_t *rowblock_bits, int32_t *runs_per_round);

/* Host planner.  aks_pb_plan_create reads a canonical CSR matrix (host pointers) and returns an opaque
 * plan (NULL on error: aks_last_error) whose array lengths are written to *sizes; aks_pb_plan_export
 * copies the planned arrays into caller-provided host buffers of those lengths; aks_pb_plan_destroy
 * frees the plan.  AKS_ERR_UNSUPPORTED (via aks_last_error / a NULL plan) if the matrix is too large
 * for 32-bit entry positions or has more than 2^26 (sub-slab, row block) tiles. */
void *aks_pb_plan_create(const int32_t *indptr, const int32_t *indices, const void *values,
                         int32_t values_complex, int64_t n_rows, int64_t n_cols, aks_pb_sizes *sizes);
int aks_pb_plan_export(const void *plan, void *val_out, uint16_t *lcol_out, int32_t *slab_begin_out,
                       int32_t *slab_end_out, aks_pb_run *runs_out, int32_t *rb_run_ptr_out,
                       uint16_t *lrow_out);
/* Zero-copy alternative to aks_pb_plan_export: the plan's own host arrays (lengths as in aks_pb_sizes; `val` holds
 * nnz_pad doubles, 2 nnz_pad if the values are complex).  Valid until aks_pb_plan_destroy; read-only for the caller.
 * A caller that only uploads the arrays saves a second 0.6 GB host copy at n = 10M (added in round 4, ABI 4). */
typedef struct aks_pb_plan_arrays {
    const void *val;
    const uint16_t *lcol;
    const int32_t *slab_begin, *slab_end;
    const aks_pb_run *runs;
    const int32_t *rb_run_ptr;
    const uint16_t *lrow;
} aks_pb_plan_arrays;
int aks_pb_plan_view(const void *plan, aks_pb_plan_arrays *out);
void aks_pb_plan_destroy(void *plan);
/* y = A x or y += A x with the binned form (two launches on `stream`). */
int aks_pb_spmv(const aks_pb_matrix *A, const aks_c128 *d_x, aks_c128 *d_y, int32_t accumulate,
                const void *d_ws, void *stream);

/* ---- sliced SpMV (same operation, for matrices with short rows of similar length: stencils, bands) ----
 * One lane per row.  A SLICE is 64 consecutive rows stored entry-major -- entry k of the 64 rows, then
 * entry k + 1, ... -- padded to the slice's longest row with column -1 (value 0): the value and column
 * loads of a wave are contiguous, a row is summed in column order in registers, and the workgroups of
 * one XCD take a contiguous eighth of the slices so that x entries shared by rows a grid line or plane
 * apart are fetched into one L2.  Worth it while the padding is small (the host layer uses it up to
 * 1.25 x nnz, by measurement against the CSR-stream kernel).
 *   slice_ptr[s] = first slot of slice s (a multiple of 64), slice_ptr[n_slices] = nnz_pad;
 *   slot slice_ptr[s] + k * 64 + (row - 64 s) holds entry k of `row`.                                  */
typedef struct aks_sell_matrix {
    int64_t n_rows, n_cols, nnz, nnz_pad, n_slices;     /* n_slices = ceil(n_rows / 64)                  */
    int32_t values_complex, pad_;
    const int64_t *d_slice_ptr;                         /* n_slices + 1                                  */
    const int32_t *d_col;                               /* nnz_pad column indices, -1 = padding          */
    const void *d_val;                                  /* nnz_pad values (f64 or c128)                  */
} aks_sell_matrix;
/* Host helpers (pure functions of a canonical CSR matrix, host pointers).  aks_sell_plan_size returns
 * nnz_pad (or a negative error); aks_sell_plan_fill writes the three arrays (lengths n_slices + 1,
 * nnz_pad, nnz_pad). */
int64_t aks_sell_plan_size(const int32_t *indptr, int64_t n_rows);
int aks_sell_plan_fill(const int32_t *indptr, const int32_t *indices, const void *values, int32_t values_complex,
                       int64_t n_rows, int64_t *slice_ptr_out, int32_t *col_out, void *val_out);
/* y = A x or y += A x with the sliced form (one launch on `stream`); _real: float64 vectors. */
int aks_sell_spmv(const aks_sell_matrix *A, const aks_c128 *d_x, aks_c128 *d_y, int32_t accumulate,
                  const void *d_ws, void *stream);

/* ---- orthogonalisation: replaces dgks_gs (ortho.py:56-107) ---------------
 * Stage entry points, in call order.  Between stages a multi-GPU host
 * all-reduces the named slot over the row shards (RCCL); with one GPU the
 * stages are simply chained (aks_dgks_gs below does that).
 *
 *   aks_gs_project        red1[0:J] = V[:, :J]^H w ; red1[J] = ||w||^2            (ortho.py:92-94)
 *   aks_gs_update_project w -= V red1[0:J] ; red2[0:J] = V^H w ; red2[J] = ||w||^2 (ortho.py:96-98, 102)
 *   aks_gs_update_norm    if sqrt(red2[J]) < eta*sqrt(red1[J]):                  (ortho.py:101,104-105)
 *                              w -= V red2[0:J] ; red3[0] = ||w||^2
 *   aks_gs_finish         h = red1 (+ red2 if second pass) -> H[0:J, j];         (ortho.py:95,103,107)
 *                         beta -> ctrl; breakdown = beta < tol;
 *                         if normalize != 0 and no breakdown, the caller's next two
 *                         lines too: H[J, j] = beta and w /= beta              (decomposition.py:61-66)
 * Chained inside the library (aks_dgks_gs, aks_arnoldi_expand) the last two stages are ONE launch whenever no n-sized
 * normalisation follows (normalize 0, or 2 = deferred): the second-pass kernel books the step itself (ABI 4).
 */
int aks_gs_project(int64_t n_rows, int32_t J, const aks_c128 *d_V, int64_t ldv,
                   const aks_c128 *d_w, void *d_ws, int64_t ws_bytes, int32_t max_dim, void *stream);
int aks_gs_update_project(int64_t n_rows, int32_t J, const aks_c128 *d_V, int64_t ldv,
                          aks_c128 *d_w, void *d_ws, int64_t ws_bytes, int32_t max_dim, void *stream);
int aks_gs_update_norm(int64_t n_rows, int32_t J, const aks_c128 *d_V, int64_t ldv,
                       aks_c128 *d_w, double eta, void *d_ws, int64_t ws_bytes, int32_t max_dim,
                       void *stream);
int aks_gs_finish(int64_t n_rows, int32_t J, aks_c128 *d_w, aks_c128 *d_Hcol, int64_t ldh,
                  double tol, double eta, int32_t normalize, void *d_ws, int64_t ws_bytes,
                  int32_t max_dim, void *stream);

/* All four stages chained (single GPU).  d_Hcol points at H[0, j]; the kernel
 * writes H[i*ldh] for i <= J. */
int aks_dgks_gs(int64_t n_rows, int32_t J, const aks_c128 *d_V, int64_t ldv, aks_c128 *d_w,
                aks_c128 *d_Hcol, int64_t ldh, double tol, double eta, int32_t normalize,
                void *d_ws, int64_t ws_bytes, int32_t max_dim, void *stream);

/* ---- the operator of one rank: its rows of A, and how the x entries of other ranks reach it -------
 * One GPU: `diag` is the whole matrix, `comm` is NULL and the remaining fields are unused.
 * Row-sharded over several GPUs (one process per GPU): `diag` holds the columns this rank
 * owns (local column ids), `off` the others with column ids = positions in the ghost buffer
 * (off.n_rows == 0 if there are none).  Before each product the entries other ranks need are packed
 * (d_send_idx -> d_sendbuf), exchanged (grouped ncclSend / ncclRecv on a side stream of the
 * communicator) WHILE the diagonal block is applied, then the off-diagonal block accumulates
 * y += A_off ghost.  send_counts / recv_counts (host arrays, one entry per rank, in vector entries)
 * say how d_sendbuf / d_ghostbuf are cut; both buffers are ordered by peer rank. */
typedef struct aks_csr_block {
    int64_t n_rows, n_cols;             /* n_rows == 0: empty block                                */
    const int32_t *d_indptr, *d_indices;
    const void *d_values;               /* float64 or complex128                                   */
    const int32_t *d_tiles;             /* aks_csr_plan_tiles                                      */
    int64_t n_tiles;
    int32_t values_complex, lanes_per_row;
    const aks_pb_matrix *pb;            /* not NULL: apply the block with the tile-binned form     */
    const aks_sell_matrix *sell;        /* not NULL: apply the block with the sliced form          */
} aks_csr_block;

typedef struct aks_shard {
    aks_csr_block diag, off;
    void *comm;                         /* aks_comm_create handle; NULL = single GPU               */
    const int32_t *d_send_idx;          /* n_send local row ids to pack                            */
    void *d_sendbuf, *d_ghostbuf;       /* n_send / n_ghost vector entries (complex128, or float64
                                           in real-packed mode)                                    */
    int64_t n_send, n_ghost;
    const int64_t *send_counts, *recv_counts;
    int32_t any_exchange, pad_;         /* 0: no rank needs anything from any other                */
} aks_shard;

/* ---- communicator (RCCL over xGMI; one process per GPU) -------------------------------------------
 * Rank 0 draws an id with aks_comm_unique_id (AKS_COMM_ID_BYTES bytes), the host layer hands it to every
 * rank (any out-of-band channel: torch.distributed, MPI, a file), every rank calls aks_comm_create.
 * The handle owns an ncclComm_t, a side stream for the ghost exchange and two events. */
#define AKS_COMM_ID_BYTES 128
int aks_comm_unique_id(void *id_out);
int aks_comm_create(const void *id, int32_t rank, int32_t size, void **comm_out);
int aks_comm_destroy(void *comm);
/* In-place sum over the ranks of `count` doubles on `stream` (the Gram-Schmidt reductions). */
int aks_comm_allreduce_sum(void *comm, double *d_buf, int64_t count, void *stream);
/* Which all-reduce the communicator runs: 0 = ncclAllReduce (the default), 1 = the one-shot mailbox exchange
 * (AKS_ALLREDUCE=oneshot in the environment of EVERY rank at aks_comm_create: ONE single-workgroup kernel per reduction
 * writes this rank's [h ; ||w||^2] into a row of every peer's fine-grained mailbox, raises the peers' arrival counters,
 * polls its own counter -- with a deadline: every wave has an exit, nothing blocks a hardware queue -- and sums the
 * rows in rank order: identical bits on all ranks; SURVEY 5 / 8(e).  The call counter lives on the device, so the
 * launch can be captured into a hipGraph and replayed).
 * aks_comm_create proves the exchange (posts that never arrive are a failed proof after AKS_ONESHOT_SELFTEST_MS, default
 * 2000, not a hang) and lets the ranks vote; if any rank cannot, ALL stay with ncclAllReduce and `why_not` (optional,
 * NUL-terminated, at most why_bytes) says why.  Ranks that share a process are voted down (their streams can share a
 * hardware queue: a polling reduction in front of the post it polls for times out) unless AKS_ONESHOT_SAME_PROCESS=1.
 * Negative: error.  (AKS_ONESHOT_FAULT_RANK=<rank> loses that rank's posts: fault injection for tests.) */
int aks_comm_allreduce_path(void *comm, char *why_not, int64_t why_bytes);
/* 0: fine.  1: a one-shot reduction of this rank waited AKS_ONESHOT_TIMEOUT_MS (default 30000) for its peers' posts in
 * vain; its result and every later one is NaN, `why` (optional) names the call.  Reads one word of pinned host memory:
 * no device call, callable at any time (arnoldi_amd.engine checks it with every read-back of H).  ABI 6. */
int aks_comm_status(void *comm, char *why, int64_t why_bytes);
/* hipGraphs that captured operations of this communicator (a replayed expansion: the ghost exchange's grouped send /
 * recv, the reductions) must be destroyed BEFORE it: ncclCommDestroy does not return while a graph holds a captured send
 * / recv (profiles/r05_capture_crash.txt).  The host layer counts them here -- retain after a capture, release after
 * hipGraphExecDestroy; both return the new count -- and aks_comm_destroy REFUSES (AKS_ERR_ARG, communicator untouched)
 * while the count is not zero: a forgotten graph is an error message, not a hang.  ABI 6. */
int aks_comm_graph_retain(void *comm);
int aks_comm_graph_release(void *comm);
/* Personalised exchange of BYTES between all ranks on `stream`, one group of sends / receives: rank r receives
 * send_bytes[r] bytes starting at d_send + send_offsets[r] of every peer into d_recv + recv_offsets[peer]
 * (recv_bytes[peer] must equal what that peer sends here; this rank's own slice is a device copy).  The arrays are
 * host arrays of `size` entries.  This is what carries the host layer's SET-UP exchanges -- which ghost entries each
 * rank needs from which owner, the row blocks of the final Schur vectors -- over the library's own communicator, so
 * that a multi-rank solve needs no other transport than the few bytes of the communicator id (the reference's only
 * distributed comparator gathers through MPI: scripts/utils.py:212-235).  An all-gather is the call with every
 * send_offsets[r] = 0 and send_bytes[r] = this rank's block. */
int aks_comm_alltoallv(void *comm, const void *d_send, const int64_t *send_offsets, const int64_t *send_bytes,
                       void *d_recv, const int64_t *recv_offsets, const int64_t *recv_bytes, void *stream);

/* y = A x for this rank's rows: pack, exchange overlapped with the diagonal block, off-diagonal block.
 * flags: AKS_EXPAND_REAL_PACKED for float64 vectors.  A no-op on the device once the control block of
 * d_ws (may be NULL) says `broken`. */
int aks_shard_apply(const aks_shard *A, const void *d_x, void *d_y, const void *d_ws, void *stream,
                    int32_t flags);

/* ---- Arnoldi expansion: replaces arnoldi_decomposition (decomposition.py:13-68)
 * for j in [start_dim, end_dim):  V[:, j+1] = A V[:, j]; dgks_gs; normalise.
 * ONE host call per expansion on every rank, no host synchronisation inside; results (H columns,
 * control block) are read back by the caller afterwards.  With A->comm set, the reductions
 * [V^H w; ||w||^2] of the Gram-Schmidt stages are all-reduced over the ranks on `stream` between the
 * stage kernels: two per step, plus a third (the norm after the second DGKS pass) unless
 * AKS_EXPAND_LAZY_THIRD is given.  With that flag a step whose DGKS test fires leaves a rank-local
 * norm behind: the caller must look at the control block's second_passes afterwards and repeat the
 * expansion without the flag if it moved (arnoldi_amd.engine does; matrices whose steps never need a
 * second pass -- random graphs, Markov chains -- then run with two collectives per step).
 * flags: AKS_EXPAND_FROM_W  V[:, start_dim+1] already holds A V[:, start_dim] (applied ahead of time,
 *                           e.g. while the host did the restart's Schur step): the first step starts
 *                           at the orthogonalisation; identical results;
 *        AKS_EXPAND_REAL_PACKED  see "real-packed mode" below.
 *        AKS_EXPAND_DEFER_SCALE  deferred normalisation: the reference divides every new basis vector by its norm
 *                           at once (decomposition.py:66), a pass of 32 n bytes per step that only scales.  With
 *                           this flag the new columns V[:, start_dim+1 .. end_dim] stay RAW -- column c holds
 *                           beta_c v_c and the workspace's colscale[c] = beta_c -- and every reader inside the library
 *                           divides a raw column's entries as it loads them (the same IEEE division, so every
 *                           number that enters the arithmetic is bit for bit the one the normalised column would
 *                           have held).  Precondition: columns 0 .. start_dim - 1 are normalised (column start_dim
 *                           may be the raw column a restart compression carried over).  Honoured only while
 *                           this rank's diagonal block is in the binned form (an x entry is divided once) or in the
 *                           sliced form with a mean padded row length of at most 8 (divided once per non-zero: cheap
 *                           next to the 32 n byte pass only for short rows); otherwise ignored.  The caller must follow the expansion
 *                           with aks_truncate_ws (scaled coefficients, see there) before anything but
 *                           aks_arnoldi_expand / aks_shard_apply_col / aks_truncate_ws reads V.  H and the control
 *                           block of an expansion are bit for bit the same with and without the flag. */
#define AKS_EXPAND_FROM_W 1
#define AKS_EXPAND_REAL_PACKED 2
#define AKS_EXPAND_LAZY_THIRD 4
#define AKS_EXPAND_DEFER_SCALE 8
int aks_arnoldi_expand(const aks_shard *A, aks_c128 *d_V, int64_t ldv, aks_c128 *d_H, int64_t ldh,
                       int32_t start_dim, int32_t end_dim, double tol, double eta, void *d_ws,
                       int64_t ws_bytes, int32_t max_dim, void *probe, void *stream, int32_t flags);

/* ---- real-packed mode (the reference's "real arithmetic" TODO, README.md:112-119) ----------------
 * For a real matrix and a real start vector the whole Krylov basis is real.  A real column of n_rows
 * float64 IS a complex128 column of ceil(n_rows/2) slots (rows 2i, 2i+1 in slot i; an odd tail keeps
 * Im = 0), and on such panels
 *     Re(V^H w) = V^T w,    w -= V h (h real),    ||w||,    V <- V Q (Q real)
 * are exactly the real operations -- so every panel kernel above is reused on HALF the rows (half the
 * HBM traffic); only the operator needs real-vector forms.  aks_workspace_set_real(ws, 1) makes the
 * reductions drop the imaginary parts of the projections (they are not part of the real dot product).
 * Callers pass n_panel = ceil(n_rows / 2) as n_rows to the aks_gs_*, aks_truncate, aks_combine and
 * aks_scale entry points and real coefficients (Im = 0) in Qp / S. */
int aks_workspace_set_real(void *d_ws, int32_t real_packed, void *stream);
int aks_csr_spmv_real(int64_t n_rows, const int32_t *d_indptr, const int32_t *d_indices,
                      const double *d_values, const int32_t *d_tiles, int64_t n_tiles,
                      int32_t lanes_per_row, const double *d_x, double *d_y, int32_t accumulate,
                      const void *d_ws, void *stream);
/* Binned form; A->d_prod (nnz complex128) is used as nnz float64. */
int aks_pb_spmv_real(const aks_pb_matrix *A, const double *d_x, double *d_y, int32_t accumulate,
                     const void *d_ws, void *stream);
/* Sliced form (real matrix values). */
int aks_sell_spmv_real(const aks_sell_matrix *A, const double *d_x, double *d_y, int32_t accumulate,
                       const void *d_ws, void *stream);
int aks_gather_f64(int64_t count, const int32_t *d_idx, const double *d_src, double *d_dst, void *stream);

/* Real-packed expansion: aks_arnoldi_expand(..., flags | AKS_EXPAND_REAL_PACKED): diag.n_rows is the matrix
 * dimension of this rank's rows; V columns are real-packed, ldv counts complex slots >= ceil(n_rows/2); the
 * workspace is laid out for ceil(n_rows/2) rows and set to real mode. */

/* y = A V[:, col] for this rank's rows, where column `col` of the basis may be raw (its scale is taken from the
 * workspace): the look-ahead product A V[:, end_dim] behind an expansion that deferred its normalisations. */
int aks_shard_apply_col(const aks_shard *A, const aks_c128 *d_V, int64_t ldv, int32_t col, void *d_y, void *d_ws,
                        int64_t ws_bytes, int32_t max_dim, void *stream, int32_t flags);

/* ---- restart compression: replaces krylov_schur.py:78 and :81 --------------
 * V[:, :p] = V[:, :m] @ Qp   (in place: a wave reads all m columns of its 64 rows before it
 * overwrites any; f64 MFMA)   and   V[:, p] = V[:, m].   d_Qp is m x p complex128, row-major
 * (ld = p).  Limits: p <= AKS_MAX_TRUNC and ceil(m/4)*4 * ceil(p/8)*8 * 16 B <= 160 KiB (Qp in LDS). */
int aks_truncate(int64_t n_rows, int32_t m, int32_t p, aks_c128 *d_V, int64_t ldv,
                 const aks_c128 *d_Qp, void *stream);

/* The same for a basis whose columns c >= some c0 are raw (an expansion with AKS_EXPAND_DEFER_SCALE).  The CALLER folds
 * the scales into the coefficients: row c of d_Qp must hold Qp[c, :] / beta_c for a raw column c (beta_c = H[c, c-1],
 * known to the host) -- the kernel then multiplies raw entries by scaled coefficients, no division on the device
 * (this is the one place where deferring changes the arithmetic: (beta v) (q / beta) instead of v q, a rounding-level
 * difference once per restart).  V[:, p] = V[:, m] is a bit copy: column p inherits the scale of column m, i.e. the
 * next expansion starts from a raw column.  d_V points at the FIRST column of the (sub-)basis, which is column col0 of
 * the whole basis (0, or l with l locked columns in front); the scales of columns col0 .. col0 + m are updated
 * accordingly (stream-ordered). */
int aks_truncate_ws(int64_t n_rows, int32_t m, int32_t p, aks_c128 *d_V, int64_t ldv, const aks_c128 *d_Qp,
                    int32_t col0, void *d_ws, int64_t ws_bytes, int32_t max_dim, void *stream);

/* ---- Ritz vectors: replaces  ritz_vectors = V_m @ S  (decomposition.py:125) and
 * eivecs = V[:, :nev] @ Y  (explicit_restarts.py:167) --------------------------------
 * out[:, :q] = V[:, :m] @ S, out of place (out must not overlap V; the in-place restart form is
 * aks_truncate).  d_S is m x q complex128, row-major (ld = q); q <= AKS_MAX_TRUNC and the LDS limit of
 * aks_truncate apply.  Same f64-MFMA kernel as aks_truncate. */
int aks_combine(int64_t n_rows, int32_t m, int32_t q, const aks_c128 *d_V, int64_t ldv,
                const aks_c128 *d_S, aks_c128 *d_out, int64_t ldo, void *stream);

/* w *= alpha: the normalisation that ends the explicit-restart solvers' mgs (explicit_restarts.py:74-77;
 * the projections before it are aks_gs_project / aks_gs_update_project with J = 1, one column at a time). */
int aks_scale(int64_t n_rows, aks_c128 *d_w, double alpha_re, double alpha_im, void *stream);

/* ---- device-time probes (measurement only; bench.py's roofline figures) --------
 * A probe owns `capacity` hipEvent pairs.  When one is handed to
 * aks_arnoldi_expand, every SpMV gets a pair with tag AKS_PROBE_SPMV and every
 * orthogonalisation (project .. finish) a pair with tag AKS_PROBE_ORTHO.  The
 * events ride on the kernel launches themselves (hipExtLaunchKernelGGL: start =
 * begin of the first kernel of the group, stop = end of its last kernel), so
 * no marker packets sit between the kernels of the timed region; only a
 * sharded SpMV with a ghost exchange is bracketed by recorded events.  aks_probe_read waits for
 * the last recorded event and returns the number of pairs with `tag` and the
 * sum of their elapsed times in milliseconds. */
#define AKS_PROBE_SPMV 0
#define AKS_PROBE_ORTHO 1
/* inside a sharded SpMV (communicator with an exchange): packing the entries other ranks need; the grouped
 * send / recv (recorded on the communicator's side stream); the diagonal block; waiting for the ghost entries
 * + the off-diagonal block.  PACK + max(EXCHANGE, DIAG) + the off-diagonal kernel ~ the SPMV pair. */
#define AKS_PROBE_PACK 2
#define AKS_PROBE_EXCHANGE 3
#define AKS_PROBE_DIAG 4
#define AKS_PROBE_OFFDIAG 5
/* every reduction over the ranks between the Gram-Schmidt stages (aks_comm_allreduce_sum: ncclAllReduce or the
 * one-shot exchange), bracketed by recorded events on the compute stream: 2 - 3 pairs per Arnoldi step */
#define AKS_PROBE_ALLREDUCE 6
int aks_probe_create(int32_t capacity, void **probe_out);
int aks_probe_destroy(void *probe);
int aks_probe_reset(void *probe);
int aks_probe_read(void *probe, int32_t tag, int32_t *count_out, double *total_ms_out);

/* ---- measurement aids (bench.py) ---------------------------------------------
 * d_dst[0:bytes] = d_src[0:bytes] as one non-temporal read + write stream of 16-byte items (bytes a multiple of 16, both
 * pointers 16-byte aligned): the streaming rate of THIS box, next to which a restart rate is comparable across boxes. */
int aks_stream_copy(void *d_dst, const void *d_src, int64_t bytes, void *stream);
/* Versions of what the process actually runs on: hipRuntimeGetVersion, hipDriverGetVersion, ncclGetVersion of the
 * librccl.so the library loaded (-1: none loaded yet, 0: the test stand-in).  Any pointer may be NULL. */
int aks_runtime_versions(int32_t *hip_runtime, int32_t *hip_driver, int32_t *rccl);

/* ---- small utilities used by the host driver --------------------------------
 * dst[i] = src[idx[i]]  -- packs the x entries another row shard needs
 * before the SpMV exchange (multi-GPU). */
int aks_gather_c128(int64_t count, const int32_t *d_idx, const aks_c128 *d_src, aks_c128 *d_dst,
                    void *stream);

#ifdef __cplusplus
}
#endif
#endif /* ARNOLDI_HIP_H */
