/*
 * dualip_hip.h -- C ABI of libdualip_hip.so: the MI355X (gfx950) dual-ascent hot path of the matching LP.
 *
 * This is the drop-in boundary for the reference's (linkedin/DuaLip v5.0.1, pure PyTorch) hot path.  The
 * reference has no FFI of its own -- every "kernel" is a sequence of ATen ops -- so each entry point below names
 * the reference Python function (path:line under /root/reference) whose work it replaces.  INTEGRATION.md shows
 * the ctypes stub a DuaLip maintainer would add to call them.
 *
 * Conventions
 *   - every function returns int: 0 = ok, <0 = -(hipError_t), >0 = DL_E_* argument/contract error.
 *     dl_last_error_string() describes the last non-zero status of the calling thread.
 *   - all array arguments are DEVICE pointers unless the name ends in _host.
 *   - all work is enqueued on the given hipStream_t (pass torch.cuda.current_stream().cuda_stream); nothing
 *     synchronises unless documented.  One handle per (process, device); handles are not thread-safe.
 *   - the caller owns every input/output buffer; a handle owns only derived metadata (re-encoded row indices,
 *     the wave-tile table, per-workgroup partial slabs, the step-size ring).
 *   - no C++ exception crosses the boundary; no torch types appear in any signature.
 */
#ifndef DUALIP_HIP_H
#define DUALIP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* dl_stream_t; /* hipStream_t */

enum { DL_F32 = 0, DL_F64 = 1 };  /* value dtype of A, c, lambda, b, x */
enum { DL_I32 = 0, DL_I64 = 1, DL_U16 = 2 };  /* dtype of the caller's ccol_indices / row_indices (torch CSC gives int64); DL_U16: row indices only
                                                  (dl_matching_create2: indices narrowed on the way to the device, dl_stage_to_device) */

/* projection kinds: reference registry names (src/dualip/projections/{box,cone,simplex}.py) */
enum {
    DL_PROJ_NONE = 0,        /* identity: "cone" with neither bound (cone.py:26-28) / column in no ProjectionEntry */
    DL_PROJ_BOX = 1,         /* "box": clamp(x, p0=lower, p1=upper)               box.py:15-16 */
    DL_PROJ_CONE_LOWER = 2,  /* "cone" lower=p0: clamp(min=p0)                    cone.py:22-23 */
    DL_PROJ_CONE_UPPER = 3,  /* "cone" upper=p0: clamp(max=p0)                    cone.py:24-25 */
    DL_PROJ_SIMPLEX = 4,     /* "simplex": {x>=0, sum x <= p0=z}, 1e-6 slack      simplex.py:126-236,239-255 */
    DL_PROJ_SIMPLEX_EQ = 5   /* "simplex_eq": {x>=0, sum x == z} over the column's own non-zeros (simplex.py:258-274;
                                the reference's dependence on the padded bucket height is NOT reproduced) */
};

enum {
    DL_OK = 0,
    DL_E_ARG = 1,        /* null pointer / negative size / bad enum                      (reference: ValueError) */
    DL_E_PROJ = 2,       /* unknown projection kind or z <= 0                            (ValueError / assert z > 0) */
    DL_E_LAYOUT = 3,     /* ccol_indices not monotone, row index out of range, nnz mismatch (non-CSC input) */
    DL_E_STATE = 4,      /* call sequence violates the handle's contract */
    DL_E_NOMEM = 5
};

/* One ProjectionEntry's operator (src/dualip/projections/base.py:8-12) without its index list. */
typedef struct {
    int32_t kind;   /* DL_PROJ_* */
    int32_t flags;  /* DL_PROJ_FLAG_*; 0 for the matching handle */
    double p0;      /* box: lower | cone: the bound | simplex: z */
    double p1;      /* box: upper */
} dl_proj_desc;

/* dl_project_dense only: the reference's method="bisection_search" for the simplex kinds (simplex.py:6-123: nu bisected to
 * 1e-6 after its feasibility and top-2 shortcuts) instead of the exact threshold.  A matching handle rejects it: inside the
 * objective such entries take the dense-block route of user-defined operators (objectives/matching.py:_CustomBlocks). */
enum { DL_PROJ_FLAG_BISECTION = 1 };
/* dl_matching_create only: keep this entry's columns in window tiles (no column-per-lane slices).  The fairness objective sets
 * it: the sliced kernel with a fourth streamed array spills registers (10M entities, all-simplex: 0.36 ms per iteration sliced
 * against 0.32 ms in window tiles), and its folded fallback rewrites c every iteration. */
enum { DL_PROJ_FLAG_NO_SLICES = 2 };

typedef struct dl_matching dl_matching; /* opaque: one matching objective (A, c) on one device */
typedef struct dl_agd dl_agd;           /* opaque: device-resident state of one maximize() run */

const char* dl_last_error_string(void);
int dl_version(void);

/* ---------------------------------------------------------------------------------------------------------
 * Matching objective
 * ------------------------------------------------------------------------------------------------------- */

/* Replaces MatchingSolverDualObjectiveFunction.__init__/_compute_buckets (src/dualip/objectives/matching.py:43-114).
 *   A and c are CSC (m x n) with the SAME pattern: colptr[n+1], rowidx[nnz] (idx_dtype), a[nnz], c[nnz] (val_dtype);
 *   one primal variable per stored non-zero.  a and c are referenced, not copied (the reference keeps references
 *   too) and must not change afterwards: their maxima and the row histogram, taken here, bound the fixed-point
 *   gradient accumulators -- apply Jacobi pre-conditioning BEFORE creating the handle.  rowidx is re-encoded once to
 *   uint16 (m <= 65536) or uint32; colptr is consumed once to build the wave-tile table (16-byte aligned windows of
 *   <= 256 non-zeros of whole columns, four per lane; <= 64 non-zeros, one per lane, when a / c are not 16-byte aligned
 *   or nnz < 1024; tiles never mix projection entries) that replaces the reference's power-of-two buckets.
 *   col_proj: int32[n], index into projs_host per column, -1 = no entry; NULL = every column uses projs_host[0].
 * Synchronises `stream` once (one-off host-side tile packing). */
int dl_matching_create(dl_matching** out, int64_t m, int64_t n, int64_t nnz, const void* colptr, const void* rowidx,
                       int idx_dtype, const void* a, const void* c, int val_dtype, const dl_proj_desc* projs_host,
                       int32_t n_proj, const int32_t* col_proj, dl_stream_t stream);
/* The same with the row indices in a type of their own: DL_I32 / DL_I64, or DL_U16 (m <= 65536) -- what dl_stage_to_device produces when it
 * narrows a host-resident int64 index array on its way across the link; colptr keeps idx_dtype (DL_I32 / DL_I64).  dl_matching_create is
 * this call with row_dtype = idx_dtype. */
int dl_matching_create2(dl_matching** out, int64_t m, int64_t n, int64_t nnz, const void* colptr, int idx_dtype, const void* rowidx,
                        int row_dtype, const void* a, const void* c, int val_dtype, const dl_proj_desc* projs_host, int32_t n_proj,
                        const int32_t* col_proj, dl_stream_t stream);
int dl_matching_destroy(dl_matching* h);

/* HOST -> DEVICE staging for callers that keep the problem in host memory (the reference's drivers do: run_solver.py:17-32
 * `transfer_tensors_to_device`, benchmark/run_matching_benchmark_dist.py:95-110 hand every rank CPU shards).  Copies `count` elements of
 * `src_bytes` each from `src_host` -- ordinary pageable memory -- to `dst_dev`: `threads` host threads (0: a default) claim 16 MB chunks,
 * copy them into pinned buffers of their own and queue each chunk's DMA as soon as it is filled, so the link runs back to back from several
 * queues.  src_bytes == dst_bytes: a plain copy.  8 -> 4, 8 -> 2 or 4 -> 2: integers NARROWED on the host on their way into the pinned
 * buffer (dst_unsigned != 0: to uint32 / uint16) -- torch's int64 CSC row indices cross the link at a quarter of their size and never
 * exist in device memory in 64-bit form; *bad_out_host counts the values that did not fit (the caller treats > 0 as an error).
 * SYNCHRONOUS (returns when the data are in device memory); works on the calling thread's current device; one call at a time per process.
 * *seconds_out_host: wall clock of the call.  Not a replacement for anything the reference has native code for -- its copy is torch's. */
int dl_stage_to_device(void* dst_dev, const void* src_host, int64_t count, int src_bytes, int dst_bytes, int dst_unsigned, int threads,
                       int64_t* bad_out_host, double* seconds_out_host);

/* Footprint: make the handle SELF-CONTAINED.  A handle borrows the caller's value arrays (window tiles, single-column tiles and in-place
 * slices read them every launch) although the columns it keeps in column-per-lane slices are never read from them again, and it owns
 * re-encoded row indices for all nnz.  After this call it owns a copy of the prefix [0, K) of a / c / rows that tiles read in place
 * (K = one window past the last non-zero of a column that is not sliced: half the arrays for the benchmark's box-then-simplex map, a
 * few hundred elements for an all-simplex one) and never touches the caller's arrays again -- the caller may free A and c (the
 * reference keeps them alive for the objective's lifetime, matching.py:79-85: 16 + 8 bytes per non-zero with int64 indices).  Results
 * are bit-identical.  dl_matching_update_costs / _values are refused afterwards.  dl_matching_info(h, 2001) = 1 once owned,
 * (h, 2002) = K.  Only handles of the 256-wide layout without the fairness stream. */
int dl_matching_own_inputs(dl_matching* h, dl_stream_t stream);

/* The caller rewrote the values of c in place (same pattern): refresh what the handle derived from them -- the transposed copy
 * the column-per-lane slices read, and max |c| when a projection in use does not bound x itself.  A stays read-only (the
 * fixed-point scale of the gradient is taken from it).  Use: re-solving with new costs on the same graph; the folded form of
 * the fairness objective (dualip_amd/objectives/matching_fairness.py) calls it every iteration. */
int dl_matching_update_costs(dl_matching* h, dl_stream_t stream);
/* The same after the caller rewrote the values of A (and possibly c) in place, pattern unchanged: max |a|, max |c| and the
 * handle's transposed copies of both are refreshed.  CONTRACT: a handle borrows the caller's value arrays (window tiles read them
 * in every launch) AND owns re-laid copies of the sliced columns' values -- after any in-place change of A or c one of these two
 * calls must run before the next launch, or the two kinds of tiles see different data. */
int dl_matching_update_values(dl_matching* h, dl_stream_t stream);

/* Size/introspection: what = 0 number of wave tiles, 1 workgroups used, 2 LDS bytes per workgroup,
 * 3 lambda staged in LDS (0/1), 4 gradient privatised in LDS (0/1), 5 bytes of owned device memory,
 * 6 number of single-column ("long") tiles, 7 row-index width in bytes, 8 tile layout (non-zeros per lane: 4 or 1),
 * 9 rows whose GRADIENT accumulator the hot-rows plan keeps in LDS (0 = plan not in use: all rows or none are), 10 share of the
 * non-zeros in those rows x 1e6,
 * 11 single-column tiles long enough (> 1024 non-zeros; > 2048 for handles of the second binary) to be walked by a whole workgroup,
 * 12 column-per-lane slices (64 short columns of a simplex entry each, sorted by length; the handle owns a transposed copy of
 * their values and row indices), 13 columns in slices, 14 slice elements including padding, 15 non-zeros in slices,
 * 16 dwords per window descriptor (12; 2 when every window is point-wise: compact table),
 * 17 columns of slices that hold more than one column length (only their length bytes are read per launch),
 * 18 + w (w < 1024): rounds of workgroup w in the window tiles' cyclic deal (-1: no table; synchronous device read),
 * 2000 columns held in slices with K = 2 .. 32 lanes per column (counted in 13 too): those of 25 .. 512 non-zeros, and a handle's
 *      FEW short columns, which join the two-lane class,
 * 2001 the handle owns its inputs (dl_matching_own_inputs), 2002 elements of the caller-ordered arrays read in place (owned: kept),
 * 2003 rows whose DUAL entry the hot-rows plan stages in LDS (>= what 9 reports; = m when the whole dual vector fits: no tile gathers
 *      from L2), 2004 launches take the fused kernel's second binary (K-lane / in-place slices, dynamic deal inside a workgroup),
 * 2005 share (ppm) of the one-lane slices that only the early-finishing half of the workgroups walks (two-phase deal of handles whose
 *      window tiles do not adapt, e.g. all-simplex maps; -1: even deal; synchronous device read), 2006 updates of that share so far,
 * 2007 bytes per element of the per-workgroup gradient slabs (8; 4 for fp32 handles with the whole gradient in LDS whose projections all bound
 *      x: a workgroup flushes the low words of its 64-bit accumulators and, only when a share does not fit 32 bits, the high words too --
 *      the slab flush and the slab sums normally move half the bytes; the same exact integer sums), 2008 workgroups that sent high words
 *      in the last fused launch (synchronous device read),
 * 2009 hot-rows plan: 1 = the cold rows add into one accumulator array PER XCD with L2-local (workgroup-scope) atomics -- taken only on a device
 *      that passed the creation-time self-check of that assumption (exact counts under 2048 contending workgroups) --, 0 = one shared array,
 * 2100 mask of the PLAN switches (environment variables that force one of the handle's kernel plans -- every plan computes the same
 *      function) that were set when the handle was created: bit i = dl_switch_name(i),
 * 2101 1 for a developer build of the library (-DDL_DEVTOOLS: ablation switches that skip work are live), 0 for the shipped one. */
int64_t dl_matching_info(const dl_matching* h, int what);
/* Name of plan switch `bit` (0 .. count-1), NULL beyond the last: the only environment variables the kernels' side of the shipped
 * library reads, besides DUALIP_COMM_TIMEOUT_MS. */
const char* dl_switch_name(int bit);

/* The local part of calculate() -- K1..K5 of the reference in ONE pass over the CSC arrays
 * (matching.py:116-161 with b_vec=None; sparse_utils.py:54-85 left_multiply_sparse, :26-51 elementwise_csc,
 *  :133-220 apply_F_to_columns, :223-243 row_sums_csc):
 *     v_k = a_k * (-(1/gamma) * lambda[row_k]) + (-(1/gamma) * c_k);   x = Proj_column(v)
 *     packed_out[0..m) = A x (double);  packed_out[m] = c . x;  packed_out[m+1] = sum x^2
 *   lambda: val_dtype[m].  packed_out: double[m+2] (sum-all-reducible across column shards).
 *   x_out: val_dtype[nnz] or NULL (primal_var; only written when non-NULL, matching.py:185-187). */
int dl_matching_calculate(dl_matching* h, const void* lambda, double gamma, double* packed_out, void* x_out,
                          dl_stream_t stream);

/* Measurement hook: while enabled, every fused-pass launch of this handle is bracketed by HIP events recorded on
 * the launch stream (bench.py's roofline leg).  dl_matching_profile_read waits for the last recorded launch and
 * returns the number of bracketed launches and their summed duration in milliseconds since the hook was (re-)enabled.
 * enable = N > 1 brackets every N-th launch only (a pair of event records costs about as much as a small kernel launch). */
int dl_matching_profile(dl_matching* h, int enable);
int dl_matching_profile_read(dl_matching* h, double* total_ms_host, int64_t* launches_host);
/* Fairness pair -- the extension the reference documents in docs/demo/matching_complex.rst:8-168 (two extra constraint
 * rows whose coefficients +f_k / -f_k sit on EVERY stored non-zero k; A_fairness there is a scaled copy of A's values).
 * Create the handle with m = K + 2 rows (row indices stay < K); after this call rows K and K+1 are that pair:
 *     v_k = a_k s[r_k] + f_k (s[K] - s[K+1]) + c_k (-1/gamma),   (A x)_K = sum_k f_k x_k,   (A x)_{K+1} = -(A x)_K
 * so lambda / packed_out / b keep their m = K + 2 layout and the optimiser entry points need no change.
 *   f_values: val_dtype[nnz] in the order of a / c, 16-byte aligned, caller-owned, read by every launch (NULL: switch off).
 * One more streamed value per non-zero (16 instead of 12 bytes in fp32).  Needs the 256-wide tile layout with the dual vector
 * and the gradient (or their hot rows) in LDS; otherwise DL_E_STATE. */
int dl_matching_set_fairness(dl_matching* h, const void* f_values, dl_stream_t stream);

/* simplex_eq reference-compatibility mode.  By default "simplex_eq" is the exact projection onto
 * {x >= 0, sum x = z} over the column's own entries.  The reference projects inside a zero-padded [L x K] block per
 * nnz-bucket (sparse_utils.py:185-209; matching.py:87-114), so when a clamped column sums to less than z its deficit is
 * spread over L >= len entries.  heights_host[q*32 + j] = L of projection entry q for bucket j = bucketize(len, [0,2,4,8,..])
 * (j = 1 for len <= 2, else ceil(log2 len)); n_rows = number of projection entries.  NULL restores the exact projection. */
int dl_matching_set_eq_padding(dl_matching* h, const int32_t* heights_host, int32_t n_rows, dl_stream_t stream);
/* Developer aid (handles created with DUALIP_HIP_TIMELINE=1 in the environment): per-workgroup 100 MHz wall-clock
 * stamps of the LAST fused launch, out_host[4*wg + {0,1,2,3}] = start, prologue done, tile loop done, end.
 * Synchronises the device. */
int dl_matching_timeline_read(dl_matching* h, uint64_t* out_host, int64_t capacity);

/* The rest of calculate() for the non-distributed objective and for rank 0 of the distributed one
 * (calc_grad matching.py:25-34, slacks :164-178, distributed :280-299):
 *     grad_out = val(packed[0..m)) - b;  reg = gamma/2 * sum x^2;  dual_obj = c.x + reg + lambda . grad_out
 *   scal_out (double[6], device): dual_objective, reg_penalty, primal_objective (c.x), dual_val_times_grad,
 *   max_pos_slack, sum_pos_slack. */
int dl_dual_epilogue(int64_t m, int val_dtype, const double* packed, const void* b, const void* lambda, double gamma,
                     void* grad_out, double* scal_out, dl_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * Accelerated gradient ascent (AcceleratedGradientDescent.maximize, src/dualip/optimizers/agd.py:121-229)
 * ------------------------------------------------------------------------------------------------------- */

/* Device-resident optimiser state: x (the point the gradient is taken at), y, the previous gradient/dual pair
 * and the ring of 14 Lipschitz estimates of calculate_step_size (optimizers/agd_utils.py:65-89), the current
 * max_step_size (coupled to gamma decay, agd.py:102-109), and per-iteration logs.
 *   beta_seq_host: float32[max_iter] exactly as agd.py:93-100 computes it (the host side owns that recipe).
 *   eq_mask: uint8[m] (non-zero = equality row, left unprojected, agd.py:13-21) or NULL.
 *   lambda0: val_dtype[m] initial dual (x = y = lambda0, agd.py:144-145). */
int dl_agd_create(dl_agd** out, int64_t m, int val_dtype, int64_t max_iter, const float* beta_seq_host,
                  double initial_step_size, double max_step_size, const uint8_t* eq_mask, const void* lambda0,
                  dl_stream_t stream);
int dl_agd_destroy(dl_agd* s);

/* Pointer to the device vector the next gradient must be evaluated at (x, val_dtype[m]). */
const void* dl_agd_x(const dl_agd* s);
/* Pointer to y (val_dtype[m]) -- SolverResult.dual_val after the last step. */
const void* dl_agd_y(const dl_agd* s);
/* Pointer to the gradient (A x - b, val_dtype[m]) of the most recent step. */
const void* dl_agd_grad(const dl_agd* s);

/* Copy one of the state vectors into caller memory (device to device, asynchronous): which = 0 x, 1 y, 2 grad. */
int dl_agd_get(const dl_agd* s, int which, void* dst, dl_stream_t stream);

/* One iteration of the maximiser after the objective's local pass (and, when sharded, after the sum-all-reduce
 * of `packed`): epilogue (as dl_dual_epilogue, on x) + calculate_step_size + projected ascent step + momentum
 * (agd.py:163-187).  `iter` is 1-based.  gamma is the value used by the objective for THIS iteration.
 * decay_now != 0 applies max_step_size = step * decay_factor after the step (agd.py:106-107; the caller
 * multiplies its own gamma).  Nothing is copied to the host. */
int dl_agd_step(dl_agd* s, const double* packed, const void* b, double gamma, int64_t iter, int decay_now,
                double decay_factor, dl_stream_t stream);

/* Whole single-device loop without returning to the host between iterations: for iter = first_iter ..
 * first_iter + n_iters - 1: dl_matching_calculate(x) ; dl_agd_step.  gamma_decay_steps = 0 disables decay
 * (gamma_decay_type None); otherwise gamma *= decay_factor after every iteration with iter % steps == 0.
 * *gamma_io_host: gamma in / gamma after the last iteration out.  x_out (val_dtype[nnz] or NULL) receives the
 * primal of the LAST iteration only (agd.py:155-158 save_primal). */
int dl_agd_run_matching(dl_agd* s, dl_matching* f, const void* b, int64_t first_iter, int64_t n_iters,
                        double* gamma_io_host, int64_t gamma_decay_steps, double decay_factor, void* x_out,
                        dl_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * The exchange of the column-sharded objective -- MatchingSolverDualObjectiveFunctionDistributed.calculate's
 * three dist.reduce calls (src/dualip/objectives/matching.py:272-277) and maximize()'s barrier + two broadcasts
 * (src/dualip/optimizers/agd.py:204-206; driver benchmark/run_matching_benchmark_dist.py:136-149) become ONE
 * sum-all-reduce of double[m + 2] = [A x | c.x | sum x^2] per iteration; every rank then applies the identical update.
 * One process per GPU.  Two back-ends:
 *   DL_COMM_RCCL  ncclAllReduce over xGMI (librccl.so.1 is opened at run time, the copy PyTorch-ROCm loaded if present);
 *   DL_COMM_P2P   one-shot all-to-all over hipIpc-mapped fine-grained mailboxes, fused into the slab-reduction and step
 *                 kernels (no collective launch; every rank adds the W partial vectors in rank order, so all ranks hold
 *                 bit-identical sums).  Up to 16 ranks of one node; ranks may share a device.
 * ------------------------------------------------------------------------------------------------------- */
typedef struct dl_comm dl_comm;
enum { DL_COMM_RCCL = 1, DL_COMM_P2P = 2 };
#define DL_RCCL_ID_BYTES 128   /* ncclUniqueId */
#define DL_IPC_HANDLE_BYTES 64 /* hipIpcMemHandle_t */

/* RCCL: rank 0 calls dl_comm_rccl_unique_id and hands the 128 bytes to every rank (any side channel: the reference's
 * torch.distributed store, MPI, a file); then every rank calls dl_comm_create_rccl (collective: ncclCommInitRank) with the
 * device it computes on current.  max_count: largest all-reduce in doubles (m + 2). */
int dl_comm_rccl_unique_id(void* id_out_host);
int dl_comm_create_rccl(dl_comm** out, int32_t world, int32_t rank, const void* unique_id_host, int64_t max_count);
/* RCCL: wrap a communicator the caller already owns (ncclComm_t); it is not destroyed with the handle. */
int dl_comm_adopt_rccl(dl_comm** out, void* nccl_comm, int64_t max_count);
/* P2P, two steps around one all-gather on the caller's side channel: _begin allocates this rank's mailbox and returns its
 * hipIpcMemHandle (DL_IPC_HANDLE_BYTES); _connect takes all ranks' handles in rank order (world x DL_IPC_HANDLE_BYTES) and maps
 * them.  The caller must make sure every rank has connected before the first exchange (a barrier on the side channel). */
int dl_comm_p2p_begin(dl_comm** out, int32_t world, int32_t rank, int64_t max_count, void* ipc_handle_out_host);
int dl_comm_p2p_connect(dl_comm* c, const void* all_handles_host);
int dl_comm_destroy(dl_comm* c);
/* what = 0 back-end (DL_COMM_*), 1 world size, 2 rank, 3 exchanges issued so far, 4 capacity in doubles, 5 fenced variant of the
 * P2P ordering (0 / 1), 6 the HIP device ordinal the communicator lives on, 7 / 8 rank count / this rank AS THE RCCL COMMUNICATOR
 * ITSELF REPORTS THEM (ncclCommCount / ncclCommUserRank; -1 for P2P) -- so a run can show that the collective really spans the
 * ranks the launcher believes it started. */
int64_t dl_comm_info(const dl_comm* c, int what);
/* In-place sum over the ranks of buf[0..count) (double, device), enqueued on `stream`; count <= max_count.  Every rank must
 * issue the same sequence of exchanges on a communicator. */
int dl_allreduce_sum(dl_comm* c, double* buf, int64_t count, dl_stream_t stream);
/* Synchronises `stream` and reports whether a P2P exchange ever FAILED (DL_E_STATE): a wait for a rank timed out (waits are
 * bounded -- 20 s, DUALIP_COMM_TIMEOUT_MS -- so a lost rank cannot hang the device), or a slot's payload did not match the
 * checksum its sender announced with the flag (every exchange is verified: data behind its flag, a torn or corrupted slot).
 * The state is sticky; the results of a run that saw it are invalid on THIS rank -- ranks should meet (dl_comm_status over the
 * caller's side channel) so that all of them stop together. */
int dl_comm_check(dl_comm* c, dl_stream_t stream);
/* The same without turning the state into an error: *dead_out_host = 0 healthy, 1 a wait timed out, 2 payload checksum mismatch. */
int dl_comm_status(dl_comm* c, int32_t* dead_out_host, dl_stream_t stream);
/* Test hook (P2P): damage this rank's NEXT contribution whose sequence number is `at_exchange` (dl_comm_info(c, 3) + 1 = the next),
 * once, as it is stored into rank `target_rank`'s mailbox -- kind 1: one element with a flipped bit, kind 2: the data stores
 * dropped (the slot keeps what it held two exchanges ago) while the flag is raised as usual; kind 0 disarms.  The reader on
 * `target_rank` must notice (dl_comm_status = 2). */
int dl_comm_inject_fault(dl_comm* c, int32_t kind, int32_t target_rank, uint64_t at_exchange);
/* Bound of the in-kernel waits of the P2P exchange from now on, in milliseconds (> 0).  The Python communicator runs its
 * creation-time self-test under a short bound (2 s) and restores the default afterwards. */
int dl_comm_set_timeout_ms(dl_comm* c, int64_t ms);
/* The P2P exchange has two variants of its memory ordering (dualip_amd/csrc/comm.h): the default one relies on write-through
 * system-scope stores completing with the wavefront's vmcnt (the hardware's behaviour; no cache write-back / invalidate), the
 * FENCED one issues the system-scope release / acquire fences the HIP memory model asks for (slower, correct by the book).
 * Every rank must choose the same.  dl_comm_info(c, 5) reports the choice. */
int dl_comm_set_fenced(dl_comm* c, int fenced);
/* Soak test of the exchange, a collective call: `rounds` sum-all-reduces of max_count known values with a pseudo-random delay
 * of up to max_delay_us per rank and round (a third of the rounds back to back), every element compared with the sum each
 * rank can form by itself.  *mismatches_host = wrong elements seen by THIS rank (0 = passed); a wait that timed out shows in
 * dl_comm_check.  Works for both back-ends. */
int dl_comm_selftest(dl_comm* c, int32_t rounds, uint64_t seed, int64_t max_delay_us, int64_t* mismatches_host, dl_stream_t stream);
/* Clear the sticky timed-out state (after a failed self-test, once EVERY rank has left it) so that the communicator can be
 * tried again with another setting. */
int dl_comm_reset(dl_comm* c, dl_stream_t stream);
/* Developer aid: multiply every exchanged sum by `scale` (one rank standing in for W equal shards). */
int dl_comm_set_emulation(dl_comm* c, double scale);
/* Measurement hook: HIP events from the end of the last fused pass of an iteration to the end of the step's first kernel
 * (slab reduction + exchange + gradient statistics) inside dl_agd_run_matching_sharded. */
int dl_comm_profile(dl_comm* c, int enable);  /* enable = N > 1: every N-th exchange */
int dl_comm_profile_read(dl_comm* c, double* total_ms_host, int64_t* exchanges_host);

/* dl_agd_run_matching for a column shard: this rank's columns as n_blocks (1..4) matching handles created with the same m
 * and dtype, the exchange inside the loop, nothing returns to the host between iterations.  All ranks must call it with the
 * same iteration range.  With n_blocks > 1 and RCCL the collectives run on a communicator-owned side stream, those of all
 * but the last block overlapping the next block's fused pass.  save_primal is not offered (the reference raises
 * NotImplementedError for it in distributed mode, matching.py:255-256). */
int dl_agd_run_matching_sharded(dl_agd* s, dl_matching* const* blocks, int32_t n_blocks, dl_comm* comm, const void* b,
                                int64_t first_iter, int64_t n_iters, double* gamma_io_host, int64_t gamma_decay_steps,
                                double decay_factor, dl_stream_t stream);

/* Copy logs to the host (synchronises the stream).  rows [first, first+count) of the per-iteration log; each row
 * is 8 doubles: dual_objective, step_size, reg_penalty, dual_val_times_grad, max_pos_slack, sum_pos_slack,
 * ||grad||_2, primal_objective (c.x). */
int dl_agd_read_log(dl_agd* s, int64_t first, int64_t count, double* rows_host, dl_stream_t stream);
/* Current max_step_size (after decay coupling); synchronises. */
int dl_agd_read_max_step(dl_agd* s, double* out_host, dl_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * Stand-alone operators of the reference API
 * ------------------------------------------------------------------------------------------------------- */

/* ProjectionOperator.__call__ on a dense row-major [L x K] block, one vector per column
 * (src/dualip/projections/base.py:15-36; box.py, cone.py, simplex.py).  out may alias in. */
int dl_project_dense(int64_t L, int64_t K, int val_dtype, const void* in, void* out, const dl_proj_desc* proj_host,
                     dl_stream_t stream);

/* The stand-alone CSC primitives of the reference (src/dualip/utils/sparse_utils.py).  Inside a solve their work is fused into
 * dl_matching_calculate; these serve callers that use them on their own.  Value arrays follow the CSC order of the pattern
 * arrays given; out may alias in.
 *   dl_csc_scale_rows      left_multiply_sparse  (:54-85):   out[k] = in[k] * v[row_k]
 *   dl_csc_scale_cols      right_multiply_sparse (:88-130):  out[k] = in[k] * v[col_k]
 *   dl_csc_elementwise     elementwise_csc       (:26-51):   out[k] = a[k] op b[k], op = DL_OP_*
 *   dl_csc_row_sums        row_sums_csc          (:223-243): out[i] = sum of the values stored in row i (double accumulation;
 *                          synchronises the stream)
 *   dl_csc_project_columns apply_F_to_columns    (:133-220) for an operator with a kernel form: every selected column
 *                          (cols: int64[n_sel] device, NULL = columns 0 .. n_sel-1) is replaced by its projection; other columns'
 *                          entries of vals_out are not touched. */
enum { DL_OP_ADD = 0, DL_OP_SUB = 1, DL_OP_MUL = 2, DL_OP_DIV = 3 };
int dl_csc_scale_rows(int64_t nnz, const void* rowidx, int idx_dtype, const void* vals_in, const void* v, void* vals_out, int val_dtype,
                      dl_stream_t stream);
int dl_csc_scale_cols(int64_t n, int64_t nnz, const void* colptr, int idx_dtype, const void* vals_in, const void* v, void* vals_out,
                      int val_dtype, dl_stream_t stream);
int dl_csc_elementwise(int64_t nnz, const void* a, const void* b, void* out, int op, int val_dtype, dl_stream_t stream);
int dl_csc_row_sums(int64_t m, int64_t nnz, const void* rowidx, int idx_dtype, const void* vals, void* out, int val_dtype,
                    dl_stream_t stream);
int dl_csc_project_columns(int64_t n_sel, const int64_t* cols, const void* colptr, int idx_dtype, const void* vals_in, void* vals_out,
                           const dl_proj_desc* proj_host, int val_dtype, dl_stream_t stream);

/* Measurement hook (bench.py: aux.read_ceiling_GBps): best of `reps` streaming passes over buf[0..bytes) (16-byte aligned,
 * >= 1 MiB) with non-temporal loads, 256 workgroups x 1024 threads, in two shapes -- one stream of 16-byte loads; three streams
 * side by side (16 + 16 + 8 bytes per lane, what a window of the fused pass reads) -- the better of which is returned: the read
 * bandwidth this device reaches, in GB/s.  Synchronises the stream. */
int dl_measure_read_bandwidth(const void* buf, int64_t bytes, int32_t reps, double* gbps_out_host, dl_stream_t stream);

/* jacobi_precondition (src/dualip/preprocessing/precondition.py:8-28): row_norms_out[m] = ||A_i||_2,
 * then a[k] *= 1/row_norms[row_k] and b *= 1/row_norms in place.  rowidx in idx_dtype. */
int dl_jacobi_precondition(int64_t m, int64_t nnz, const void* rowidx, int idx_dtype, void* a, void* b,
                           void* row_norms_out, int val_dtype, dl_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * Generic LP ("miplib2017") dual objective -- MIPLIB2017ObjectiveFunction (src/dualip/objectives/miplib.py:28-109).
 * x has one entry per variable:  z = (-1/gamma)(A^T lambda' + c),  x = clamp(z, lower, upper),  lambda' = lambda * inv_row_norm.
 * The handle BORROWS every array (device pointers, caller keeps them alive): A in both CSC (for A^T lambda) and CSR
 * (for A x) with int64 pointer arrays and int32 index arrays; c, lower, upper val[n] (-inf / +inf where a bound is
 * absent); inv_row_norm val[m] = 1/||A_i|| for the Jacobi-preconditioned variant (miplib.py:48-56,74-75,94-95) or NULL.
 * --------------------------------------------------------------------------------------------------------- */
typedef struct dl_lp dl_lp;
int dl_lp_create(dl_lp** out, int64_t m, int64_t n, int64_t nnz, const int64_t* colptr, const int32_t* rowidx, const void* vals_csc,
                 const int64_t* rowptr, const int32_t* colidx, const void* vals_csr, const void* c, const void* lower, const void* upper,
                 const void* inv_row_norm, int val_dtype);
int dl_lp_destroy(dl_lp* h);
/* miplib.py:77-99 in one call: packed_out[0..m) = (A x)_i [* inv_row_norm_i], packed_out[m] = c.x, packed_out[m+1] = sum x^2
 * (the layout dl_dual_epilogue / dl_agd_step consume; pass b [* inv_row_norm] there).  x_out val[n] or NULL. */
int dl_lp_calculate(dl_lp* h, const void* lambda, double gamma, double* packed_out, void* x_out, dl_stream_t stream);
/* The two halves, for projection maps that are not point-wise bounds (miplib.py:80-92 applies arbitrary registered
 * operators to index sets): dl_lp_primal with apply_bounds = 0 returns z, the caller projects, dl_lp_gradient finishes. */
int dl_lp_primal(dl_lp* h, const void* lambda, double gamma, int apply_bounds, void* x_out, dl_stream_t stream);
int dl_lp_gradient(dl_lp* h, const void* x, double* packed_out, dl_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DUALIP_HIP_H */
