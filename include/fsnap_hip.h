/* fsnap_hip.h — C ABI of libfsnap_hip.so: the MI355X (gfx950) linear-fit hot path for
 * FitSNAP, callable from the reference's Python over ctypes.
 *
 * Boundary rules: extern "C"; plain pointers and sizes only (no torch / numpy types);
 * every function returns an int status — 0 = ok, negative = argument / runtime error
 * (text via fsnap_last_error), positive = numerical failure (e.g. non-SPD normal
 * matrix); no C++ exception ever crosses the boundary.  The caller owns every buffer it
 * passes.  Matrices are row-major fp64; sizes are int64_t; the training mask is uint8_t
 * with 1 = training row (the negation of the reference's fitsnap_dict['Testing']).
 * A context is bound to ONE GPU (one process per GPU; a multi-GPU job gives every rank's
 * context a native RCCL communicator, fsnap_comm_*, and all-reduces the packed statistics).
 * Calls on one context must not be made concurrently from several threads; the ctypes
 * shim releases the GIL for the duration of each call.
 *
 * Each entry point names the reference code it replaces (file:line in FitSNAP/FitSNAP
 * at the surveyed revision).
 */
#ifndef FSNAP_HIP_H
#define FSNAP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct fsnap_ctx fsnap_ctx;

/* status codes */
#define FSNAP_OK 0
#define FSNAP_E_ARG (-1)      /* bad argument (null pointer, size, unsupported K ...) */
#define FSNAP_E_HIP (-2)      /* HIP runtime error; text in fsnap_last_error          */
#define FSNAP_E_STATE (-3)    /* call order (no rows uploaded, no weights ...)        */
#define FSNAP_E_NOMEM (-4)    /* device / host allocation failed -> MemoryError       */
#define FSNAP_NUM_NOT_SPD 1   /* normal matrix not positive definite -> LinAlgError   */
#define FSNAP_NUM_SINGULAR 2  /* exactly singular system -> LinAlgError               */
#define FSNAP_NUM_NONFINITE 3 /* NaN/Inf in the statistics -> ValueError              */

/* solve kinds for fsnap_solve */
#define FSNAP_SOLVE_CHOL 0       /* G beta = c, SPD required                                        */
#define FSNAP_SOLVE_LSTSQ 1      /* min-norm least squares, param = rcond  (svd.py:54 lstsq(aw,bw,1e-13)) */
#define FSNAP_SOLVE_RIDGE 2      /* (G + alpha I) beta = c, param = alpha (ridge.py:47-57, sklearn Ridge) */
#define FSNAP_SOLVE_RIDGE_INV 3  /* beta = inv(G + alpha I) c             (regressor.py:10-16 Local_Ridge) */
/* LSTSQ / RIDGE / RIDGE_INV without the fallback behind the Cholesky factorisations: when no Cholesky factorisation resolves the system the call returns
 * FSNAP_OK at once with *rank = -1 and beta = 0 (and *rcond_est as described at fsnap_solve) instead of running the
 * cyclic-Jacobi eigendecomposition -- O(K^3) per sweep on one core: 58 s at K = 1595.  For callers that have something
 * better to fall back on: the rows (fsnap_lstsq_rows) or a LAPACK eigensolver. */
#define FSNAP_SOLVE_LSTSQ_PROBE 4
#define FSNAP_SOLVE_RIDGE_PROBE 5
#define FSNAP_SOLVE_RIDGE_INV_PROBE 6 /* ... instead of the scalar LU with partial pivoting (np.linalg.inv semantics) */

/* packed statistics buffer: [ G (K*K row-major) | c (K) | bTb, sum(w*b), n_train ] */
#define FSNAP_PACKED_LEN(K) ((int64_t)(K) * (K) + (K) + 3)

/* ---- library / context ------------------------------------------------------------ */

/* ABI version (major*100 + minor). */
int fsnap_version(void);

/* Number of visible HIP devices. */
int fsnap_device_count(int* count);

/* Create a context on HIP device `device` (own non-blocking stream, timing events).
 * Replaces nothing in the reference: the reference's solvers are rank-0 numpy. */
int fsnap_ctx_create(int device, fsnap_ctx** ctx);
int fsnap_ctx_destroy(fsnap_ctx* ctx);

/* Run all subsequent work of this context on the caller's hipStream_t (e.g.
 * torch.cuda.current_stream().cuda_stream) so that RCCL collectives issued by the host
 * layer order after the kernels without a host sync.  NULL = the HIP default stream. */
int fsnap_ctx_set_stream(fsnap_ctx* ctx, void* hip_stream);

/* Go back to the context's own non-blocking stream. */
int fsnap_ctx_use_own_stream(fsnap_ctx* ctx);

/* Options (all optional; defaults in brackets).  Every key steers a path a default configuration can take -- the A/B forms
 * of rounds 1-5 that lost their measurements (kernels 1 / 1L / 1T2, Cholesky forms 0-4, reduce-to-root fits, ...) are gone,
 * their records are in profiles/ and HISTORY.md:
 * "nblocks" [0 = auto] workgroups of the SYRK kernels 1A / 1P / 1Q (clusters of 1QC, row chunks of 1S);
 * "nsplit" [0 = a scheduling model] row splits of the tiled kernel 1T;
 * "tiled" [0] 1 = the general-K tiled kernel 1T at every width (what short systems of 145 ... 512 columns, K > 512 and the
 *   row-space passes at K > 144 run on by default);
 * "short" [-1 = systems of 81 ... 144 columns with at most three staging phases of rows per pair of CUs (a phase: 128 rows, 112 at
 *   129 ... 144 columns -- 43 008 ... 49 152 rows on 256 CUs) take kernel 1S, which deals the tile triangle over 16 waves per row chunk
 *   instead of giving every wave the whole triangle (13 035 x 142, examples/Ta_PACE_RIDGE)] 0 = never (kernel 1A), 1 = at
 *   every row count (chunks longer than a phase are staged in several);
 * "quad_min_rows" [-1 = 8 192 rows for 145 ... 288 columns (kernel 1Q), 300 000 for 289 ... 512 (kernel 1QC)] fewest rows for
 *   the accumulator-resident kernels at those widths; shorter systems take the tiled kernel;
 * "fused_pack" [1] kernels 1A / 1S / 1P / 1Q / 1QC form the per-row pairs (mask * w, mask * w * b) of their rows in LDS inside the
 *   SYRK launch whenever a workgroup's rows fit; 0 = always the separate packing kernel (the form of larger shards and of the
 *   row-space passes);
 * "repack" [0] 1 = recompute the per-row pairs and the b-only scalars on EVERY fit even when b, w and the mask are
 *   context-owned and unchanged (what a step of bench.py does);
 * "timing_every" [1] HIP events bracket every N-th SYRK launch only -- an event record between two dependent kernels idles
 *   the stream for ~5.6 us; 0 = no events; the first launch after the option is set is a sampled one (fsnap_timing /
 *   fsnap_timing_history see the sampled ones);
 * "device_solve" [0 = K >= 232 is factorised on the GPU by the blocked kernels] 1 = from 129 columns on, 2 = never (host);
 * "chol_reuse" [1] fsnap_solve_device_rhs with a right-hand side of its own -- the refinement steps of a fit, the sweeps of the
 *   condition estimate -- runs a forward and a backward sweep with the factor the last solve of the context's own statistics
 *   left on the device (0.2 instead of 0.54 ms at K = 1595); 0 = factorise again;
 * "fused_residual" [1] fsnap_residual_rhs for K <= 288 in one pass over the rows; 0 = the two-kernel form of wider systems;
 * "rowspace_reuse_stats" [0; one-shot, cleared by the next fsnap_lstsq_rows] 1 = the caller states that the fit from the
 *   statistics which just ran -- fsnap_fit_resident on this context -- saw the rows, weights and mask as they are now; a
 *   single-rank fsnap_lstsq_rows on a system the host factorises then starts its first pass from that fit's statistics, still
 *   in the page-locked mirror, instead of computing them again: 0.35 ms of a 10^6 x 128 call;
 * "reduce_triangle" [-1 = systems of >= 256 columns all-reduce [upper triangle | c | scalars], K (K + 1) / 2 + K + 3 doubles,
 *   between a pack and an unpack kernel] 0 = always the full K^2 + K + 3, 1 = always the triangle -- every rank of a job must
 *   use the same setting;
 * "comm_timeout" [0 = the FSNAP_COMM_TIMEOUT environment default] seconds: bound of every wait behind a collective of this
 *   context and of fsnap_comm_init;
 * "staged_upload" [0 = the runtime's pageable copy: it pins the caller's pages and reads them in place, 27 ms for 1.03 GB
 *   where pinning is cheap] fsnap_upload_rows: 2 = a page-locked double buffer filled by FSNAP_UPLOAD_THREADS (default 4)
 *   host threads while the DMA drains the other slot -- for hosts with free cores and slow pinning; 1 = double buffer, handing
 *   the rest to the pageable copy when the host fills its first two 16 MiB slots at less than 20 GB/s.
 * Unknown key -> FSNAP_E_ARG. */
int fsnap_set_option(fsnap_ctx* ctx, const char* key, int64_t value);

/* Last error text of this context (or of the library when ctx is NULL).  Never NULL. */
const char* fsnap_last_error(const fsnap_ctx* ctx);

/* ---- rows: the A matrix and truth vector ----------------------------------------- */

/* Copy the m x K row-major fp64 matrix A (leading dimension lda doubles) and b[m] from
 * host memory into HBM; they stay resident for any number of re-weightings / fits.
 * Device-side counterpart of pt.shared_arrays['a'|'b'].array
 * (fitsnap3lib/parallel_tools.py:352-389, 944-1077; calculator.py:287-288). */
int fsnap_upload_rows(fsnap_ctx* ctx, const double* A, int64_t m, int64_t K, int64_t lda, const double* b);

/* Same, but A and b already live in device memory (not copied, not owned; the A
 * allocation must be readable for 16 bytes past its last element). */
int fsnap_bind_rows(fsnap_ctx* ctx, const double* dA, int64_t m, int64_t K, int64_t lda, const double* db);

/* This context holds no rows any more (context-owned copies are freed, bound ones forgotten): what a rank of a
 * multi-GPU job calls when the configurations dealt to it for the NEXT fit are none (config i -> rank i % nranks,
 * fitsnap3lib/parallel_tools.py:612-651, with fewer configurations than ranks) -- fsnap_fit_dist / fsnap_lstsq_rows
 * then contribute zeros instead of the rows of an earlier fit. */
int fsnap_drop_rows(fsnap_ctx* ctx);

/* Allocate resident, zero-filled A (m x K), b[m], w[m] in HBM without a host copy: the
 * device-side counterpart of Calculator.create_a -> pt.create_shared_array('a'|'b'|'w')
 * (fitsnap3lib/calculators/calculator.py:261-289), to be filled by fsnap_assemble. */
int fsnap_rows_alloc(fsnap_ctx* ctx, int64_t m, int64_t K);

/* Post-LAMMPS assembly of a batch of configurations straight into the resident rows
 * [row0, row0 + nrows): the `_collect_lammps` transform
 * (fitsnap3lib/calculators/lammps_snap.py:391-556, lammps_pace.py:369-509; single-config
 * form lammps_snap.py:224-389) without the host copy of A.
 *   raw[raw_rows * raw_ld]  the LAMMPS `compute snap|pace` global arrays of the batch, stacked
 *                           (what _extract_compute_np views, lammps_base.py:280-307);
 *                           raw_ld >= ncoeff*ntypes + 1, column ncoeff*ntypes = reference potential
 *   per output row r: src_row[r] (row of raw), kind[r] (0 energy: A = x/d, b = (truth-ref)/d;
 *     1 force: A = x, b = truth-ref; 2 virial: A = (1.6021765e6 x)/d, b = truth-ref;
 *     3 extra per-atom energy row: A = x/d, b = 0, w = 0), d[r] (atoms or cell volume),
 *     truth[r], weight[r], frac[r] (row of `fractions`, or -1)
 *   fractions[nfrac * ntypes]  per-configuration atom-type fractions (offset column of energy rows)
 *   blank2J[K]; offcol = 1 when bzeroflag = 0 (K = ntypes*(ncoeff+1)), else 0.
 * All pointers are host memory; synchronous. */
int fsnap_assemble(fsnap_ctx* ctx, const double* raw, int64_t raw_rows, int64_t raw_ld, int64_t nrows, int64_t row0,
                   const int64_t* src_row, const int32_t* kind, const int32_t* frac, const double* d,
                   const double* truth, const double* weight, const double* fractions, int64_t nfrac,
                   const double* blank2J, int32_t ntypes, int32_t ncoeff, int32_t offcol);

/* Copy the resident rows back to host arrays (any pointer may be NULL): fills the host
 * view of pt.shared_arrays['a'|'b'|'w'] after device-side assembly, and serves
 * Calculator.extras' Descriptors.npy / Truth-Ref.npy / Weights.npy dumps
 * (calculator.py:329-348). */
int fsnap_download_rows(fsnap_ctx* ctx, double* A, int64_t lda, double* b, double* w);

/* Row weights w[m] and training mask[m] (1 = train; NULL = all rows train) from host
 * memory.  pt.shared_arrays['w'].array and `training = [not elem for elem in
 * fitsnap_dict['Testing']]` (svd.py:35-44, ridge.py:28-37, ard.py:18-19). */
int fsnap_set_weights(fsnap_ctx* ctx, const double* w, const uint8_t* mask);
/* (fsnap_set_weights* copy the host arrays into page-locked staging before returning -- the caller may reuse them at
 * once -- and hand the DMA to the context's stream without waiting for it.) */

/* Re-weighting form of the reference's explicit-array call (svd.py:46, ridge.py:39: `w` multiplies `a[training]`
 * without being masked, i.e. the caller hands ONE WEIGHT PER TRAINING ROW): w_train[ntrain] in row order.  mask[m]
 * (1 = train) and rank[m] (rank[i] = number of training rows before row i) are host arrays, or both NULL to keep
 * the mask of the previous call (the genetic-algorithm loop of examples/library/genetic_algorithm/libmod_optimize.py:
 * 461-488 re-weights the same training set hundreds of times).  The per-row weights are expanded on the GPU. */
int fsnap_set_weights_train(fsnap_ctx* ctx, const double* w_train, int64_t ntrain, const uint8_t* mask, const int32_t* rank);

/* Same with device pointers (not copied, not owned). */
int fsnap_bind_weights(fsnap_ctx* ctx, const double* dw, const uint8_t* dmask);

/* ---- hot path --------------------------------------------------------------------- */

/* Fused mask x weight x normal equations on the resident rows:
 *   aw = w[:,None]*A[training]; bw = w*b[training]          (svd.py:44-46, ridge.py:37-39)
 *   G = aw.T @ aw; c = aw.T @ bw                            (svd.py:50-51, ridge.py:42-43,
 *                                                            regressor.py:11-12,
 *                                                            transpose_trick/example.py:234-240)
 * without materialising aw.  Outputs (host, any may be NULL): G[K*K], c[K],
 * scalars[3] = { bw.bw, sum(bw), n_train }.  Synchronous. */
int fsnap_normal_eq(fsnap_ctx* ctx, double* G, double* c, double* scalars);

/* Asynchronous form: launches on the context's stream and leaves the packed statistics
 * (FSNAP_PACKED_LEN(K) doubles) in DEVICE memory at d_packed — the buffer the host
 * layer all-reduces with RCCL (the reference's comm.Allreduce of c and d,
 * examples/library/transpose_trick/example.py:245-246). */
int fsnap_normal_eq_async(fsnap_ctx* ctx, double* d_packed);

/* Streaming form: d_packed (device, FSNAP_PACKED_LEN(K) doubles, zeroed by the caller before the first batch)
 * += statistics of the resident rows.  The reference's per-configuration accumulation `c += cm; d += dm`
 * (examples/library/transpose_trick/example.py:230-237): batches of rows are uploaded (or assembled), accumulated
 * and dropped, so A never has to be resident as a whole.  Asynchronous. */
int fsnap_normal_eq_accumulate(fsnap_ctx* ctx, double* d_packed);

/* The same loop with the assembly fused in -- `a, b, w = process_single(configuration)`, `c += aw.T @ aw`,
 * `d += aw.T @ bw` of examples/library/transpose_trick/example.py:230-237 in ONE pass: the rows of the batch are formed
 * in registers from the raw LAMMPS arrays (the transform of fsnap_assemble: lammps_snap.py:391-556, lammps_pace.py:
 * 369-509), weighted and fed to the matrix pipe; A is never written to (or read from) device memory, the context needs
 * no resident rows and the ones it holds are left alone.  Arguments as fsnap_assemble (no row0: nothing is stored);
 * every row of the batch is a training row with weight[r] (rows of kind 3 carry weight 0, as there).
 * d_packed (device, FSNAP_PACKED_LEN(K) doubles, K = ntypes * (ncoeff + offcol), zeroed by the caller before the
 * first batch) += [G | c | b.W^2.b, sum(w b), rows].  Bit-identical to fsnap_assemble into resident rows followed by
 * fsnap_normal_eq_accumulate with the tiled kernel (option "tiled" = 1; the default for K > 128).  Synchronous (the
 * host arrays may be reused on return). */
int fsnap_assemble_accumulate(fsnap_ctx* ctx, const double* raw, int64_t raw_rows, int64_t raw_ld, int64_t nrows,
                              const int64_t* src_row, const int32_t* kind, const int32_t* frac, const double* d,
                              const double* truth, const double* weight, const double* fractions, int64_t nfrac,
                              const double* blank2J, int32_t ntypes, int32_t ncoeff, int32_t offcol, double* d_packed);

/* Same as fsnap_normal_eq_async into a context-owned device buffer whose address is
 * returned in *d_packed (valid until the next call on this context); asynchronous.
 * For K <= 128 the reduction kernel also writes the statistics into a page-locked host mirror
 * which fsnap_solve_device uses instead of a D2H copy when it is
 * handed this same pointer: read-only for the caller -- to modify the buffer (e.g. all-reduce it)
 * use fsnap_normal_eq_async with a buffer of your own. */
int fsnap_normal_eq_resident(fsnap_ctx* ctx, double** d_packed);

/* Mirror packed statistics that live in device memory (e.g. the buffer a RCCL all-reduce just summed over the ranks,
 * the reference's comm.Allreduce(c), comm.Allreduce(d) in examples/library/transpose_trick/example.py:245-246) into
 * the context's page-locked host mirror; asynchronous (a small copy kernel + an event on the context's stream).  A
 * following fsnap_solve_device on the same pointer then needs no D2H copy.  No-op for K >= 232 (those are factorised
 * on the GPU).  The caller must not modify the buffer between this call and the solve. */
int fsnap_mirror_packed(fsnap_ctx* ctx, const double* d_packed, int64_t K);

/* Copy packed statistics from device memory to host arrays (any may be NULL); synchronous. */
int fsnap_download_packed(fsnap_ctx* ctx, const double* d_packed, int64_t K, double* G, double* c, double* scalars);

/* Stand-alone wavefront row weighting: aw = w[:,None]*A, bw = w*b for ALL m rows
 * (masked rows are written as zeros); host outputs, leading dimension ldaw.
 * svd.py:46 / ridge.py:39 / solver.py:75 for callers that need aw, bw themselves. */
int fsnap_weight_rows(fsnap_ctx* ctx, double* aw, int64_t ldaw, double* bw);

/* Same with device outputs, asynchronous on the context's stream. */
int fsnap_weight_rows_device(fsnap_ctx* ctx, double* d_aw, int64_t ldaw, double* d_bw);

/* preds = A @ beta (solver.py:377) for all m rows; host in/out.  preds may be NULL.
 * If sse is not NULL it receives sum over training rows of (w*(b - A beta))^2 using
 * the current weights (sklearn ARDRegression's per-iteration `rmse_`). */
int fsnap_predict(fsnap_ctx* ctx, const double* beta, double* preds, double* sse);

/* Right-hand side of one step of iterative refinement of the least-squares solution:
 *   s[K] = (wA)^T (wb - wA beta)   over the training rows, with the current weights / mask
 * (two streaming passes over the resident A).  Solving G delta = s and setting beta += delta
 * ("corrected semi-normal equations") brings the normal-equation error (~kappa^2 eps) back
 * to ~kappa eps, i.e. to what the reference's lstsq on A_w delivers (svd.py:54).
 * beta, s: host; *sse (optional) receives sum (w (b - A beta))^2. */
int fsnap_residual_rhs(fsnap_ctx* ctx, const double* beta, double* s, double* sse);

/* ---- K x K solve (host side, no context needed) ----------------------------------- */

/* Solve the K x K system given the statistics.  `kind` is one of FSNAP_SOLVE_*;
 * `param` is rcond (LSTSQ) or alpha (RIDGE, RIDGE_INV), ignored for CHOL.
 * beta[K] out; *rank (may be NULL) receives the numerical rank used; *rcond_est (may be NULL) what the factorisation
 * knows about the conditioning of the Jacobi-scaled matrix S (unit diagonal, so 1 <= lambda_max <= K): for the LSTSQ
 * kinds -- which stand in for an SVD of the rows, svd.py:54 -- min(smallest pivot, lambda_min(S) estimated from the factor
 * by 2 ... 8 Lanczos steps on S^-1, i.e. pairs of triangular sweeps: dpocon's idea, csrc/fsnap_condest.h), an estimate
 * from ABOVE that is within a factor ~1.3 of lambda_min wherever the statistics still resolve it (lambda_min > ~K eps)
 * -- for a factor on the DEVICE (fsnap_solve_device*, K >= 232): the Rayleigh-Ritz value of S^-1 on 31 probe vectors that the
 * factorisation carries in its right-hand-side strip, scaled by 120 / K, within ~[lambda_min / 5, 10 lambda_min], no sweep --;
 * a factor whose estimate falls below 64 K eps counts as unresolved exactly like a failed pivot.  For the other kinds
 * (sklearn's Cholesky, np.linalg.inv: neither looks at the conditioning) the smallest pivot alone, an upper bound of
 * lambda_min that can be off by a factor exponential in K.
 * Replaces scipy.linalg.lstsq (svd.py:54), sklearn Ridge's Cholesky solve
 * (ridge.py:47-57) and np.linalg.inv (regressor.py:15). */
int fsnap_solve(int kind, double param, int64_t K, const double* G, const double* c, double* beta, int* rank,
                double* rcond_est);

/* What the LAST K x K solve of the calling thread (fsnap_solve, fsnap_solve_device*, fsnap_fit_resident, fsnap_fit_dist) learned
 * about the conditioning: info[0] = smallest scaled pivot, info[1] = lambda_min estimate from the factor (0 = none taken or
 * numerically singular), info[2] = applications of S^-1 it took (0 = none: not an LSTSQ kind, or a reused factor),
 * info[3] = 0 host factor / 1 device factor.  Diagnostic: *rcond_est already carries min(info[0], info[1]). */
int fsnap_cond_info(double info[4]);

/* LASSO by cyclic coordinate descent on the statistics: minimises (1/2) w^T Q w - q^T w + l1_reg |w|_1 with Q = A_w^T A_w,
 * q = A_w^T b_w, i.e. scikit-learn's Lasso(alpha, fit_intercept=False, max_iter).fit(aw, bw) of the reference
 * (fitsnap3lib/solvers/lasso.py:23-28) with l1_reg = alpha * n_samples.  Stopping rule as scikit-learn's: duality gap
 * < tol * y_norm2 (y_norm2 = b_w^T b_w), looked at once a sweep's largest update drops below tol x the largest
 * coefficient, or after max_iter sweeps.  w[K]: start vector in (zeros in the reference), coefficients out;
 * *n_iter / *gap (may be NULL): sweeps run and the last duality gap.  Host side, no context needed. */
int fsnap_lasso_gram(int64_t K, const double* Q, const double* q, double y_norm2, double l1_reg, int64_t max_iter, double tol,
                     double* w, int64_t* n_iter, double* gap);

/* Same solve, taking the packed statistics [G | c | ...] from DEVICE memory (the buffer
 * fsnap_normal_eq_async / the all-reduce left in HBM).  Small systems are copied to the host
 * (page-locked staging) and solved there (faster than any GPU factorisation of a 128-step recurrence); for
 * K >= 232 (option "device_solve") a blocked Cholesky runs on the GPU and only beta crosses PCIe, provided the
 * system is well conditioned after Jacobi scaling -- otherwise the general host path decides.  Same status codes and semantics as fsnap_solve. */
int fsnap_solve_device(fsnap_ctx* ctx, int kind, double param, int64_t K, const double* d_packed, double* beta,
                       int* rank, double* rcond_est);

/* One call per fit on resident rows: fsnap_normal_eq_resident followed by fsnap_solve_device on its buffer
 * (the whole of SVD / RIDGE.perform_fit, svd.py:44-54 / ridge.py:37-59, for rows and weights already on the
 * device).  *d_packed (may be NULL) receives the address of the statistics, as fsnap_normal_eq_resident does. */
int fsnap_fit_resident(fsnap_ctx* ctx, int kind, double param, double* beta, int* rank, double* rcond_est,
                       double** d_packed);

/* Same as fsnap_solve_device with the right-hand side replaced by rhs (HOST, K doubles; NULL = the c part of the
 * packed buffer): G delta = s of an iterative-refinement step (solver.py has no counterpart: the reference's
 * lstsq works on the rows, see fsnap_residual_rhs) without bringing G to the host.
 * Factor reuse (K >= 232, option chol_reuse): when d_packed is the context's OWN statistics buffer (the address
 * fsnap_fit_resident / fsnap_normal_eq_resident / fsnap_fit_dist handed out) and the last solve of it left its factor on the
 * device, a call with rhs runs two sweeps instead of a factorisation; every library call that rewrites that buffer
 * forgets the factor.  A caller-owned device buffer is factorised on every call -- the library cannot see writes to it. */
int fsnap_solve_device_rhs(fsnap_ctx* ctx, int kind, double param, int64_t K, const double* d_packed, const double* rhs,
                           double* beta, int* rank, double* rcond_est);

/* ---- row-space least squares (ill-conditioned / rank-deficient systems) --------------------------------------- */

/* Optional dense kernel of the host language for the K x K end of fsnap_lstsq_rows when a TRUNCATION is needed and the
 * factor is large (K > 256): x = pinv_rcond(T) y for the n x n upper triangular T (row-major), singular values below
 * rcond * sigma_max dropped, *rank = number kept.  `token` changes with every new T (a caller may cache its
 * decomposition per token; two applications per solve).  Return 0 on success; anything else makes the library fall back
 * on its own one-sided Jacobi SVD -- exact, but O(n^3) per sweep on one core (~20 s at n = 1595, where LAPACK's gesdd
 * takes ~1 s).  fn = NULL removes the hook.  The Python host layer installs scipy.linalg.svd. */
typedef int (*fsnap_dense_pinv_fn)(void* user, int64_t token, int64_t n, const double* T, double rcond, const double* y,
                                   double* x, int* rank);
int fsnap_set_dense_pinv(fsnap_ctx* ctx, fsnap_dense_pinv_fn fn, void* user);

/* fit = scipy.linalg.lstsq(aw, bw, rcond) of the resident rows, computed on the ROWS like the reference's dgelsd
 * (fitsnap3lib/solvers/svd.py:44-54) instead of from the normal equations -- for systems whose K x K statistics are
 * numerically singular (kappa(A_w) beyond ~1e7) or rank deficient: shifted CholeskyQR passes on the GPU
 * (A_w = Q R_hat, Q m x K with orthonormal columns, kept in HBM next to A), then the K x K end of dgelsd on R_hat
 * (singular values below rcond * sigma_max dropped, minimum-norm solution) and one refinement step with the residual
 * of the original rows.  Needs m * K * 8 more bytes of HBM.  Collective when the context has a communicator (a rank
 * without rows passes K and contributes nothing).  *rank = numerical rank used.  info (may be NULL, 8 doubles):
 * passes, last max|Q^T Q - I|, converged (0/1), how the K x K end was solved (0 = back substitution: nothing to drop;
 * 3 = back substitution between two projections: 1...24 dropped directions found by subspace iteration, every other
 * singular value certified above the cut; 1 = the library's one-sided Jacobi SVD; 2 = the host language's dense kernel),
 * sigma_max, sigma_min estimate (bounds unless the SVD ran), relative size of the refinement step, last shift.
 * FSNAP_ROWSPACE_DEFLATE=0 in the environment takes form 3 out (A/B). */
int fsnap_lstsq_rows(fsnap_ctx* ctx, double rcond, int64_t K, double* beta, int* rank, double* info);

/* The two host steps of that solve for callers that run the passes themselves (rows streamed through
 * fsnap_normal_eq_accumulate, or the CPU tests): no context, no GPU.
 * fsnap_rowspace_factor: G = Gram matrix Q^T Q of the current Q (K x K).  first = 1 initialises R_hat (K x K,
 *   A_w = Q R_hat) to the identity.  Unless first, a deviation max|G - I| <= tol means Q is done: info[1] = 1 and nothing
 *   else changes.  Otherwise Rp (K x K, upper triangular) receives the factor to divide out (Q <- Q Rp^-1 by
 *   substitution) and R_hat <- Rp R_hat.  info (may be NULL, 3 doubles): deviation, converged, shift.
 * fsnap_rowspace_solve: beta = pinv_rcond(R_hat) z with z = Q^T (w b), dgelsd semantics; info (may be NULL, 4 doubles):
 *   how it was solved (0 / 3 / 1 as in fsnap_lstsq_rows), sigma_max, sigma_min kept, Jacobi sweeps. */
int fsnap_rowspace_factor(int64_t K, const double* G, int first, double tol, double* Rhat, double* Rp, double* info);
int fsnap_rowspace_solve(int64_t K, const double* Rhat, const double* z, double rcond, double* beta, int* rank, double* info);

/* The same K x K end for K > 256, where fsnap_lstsq_rows keeps the factors of the passes apart: R = nfac upper triangular
 * K x K factors, R[0] the first pass (R_hat = R[nfac-1] ... R[0] is NOT formed unless a truncation is needed: the product costs
 * K^3 / 3 flops per pass on one host core).  beta = pinv_rcond(R_hat) z: by nfac back substitutions when an upper estimate
 * of cond(R_hat) (sqrt(||R||_1 ||R||_inf) x Hager / Higham 1-norm estimates of the inverses, per factor) shows that no singular value can fall
 * below rcond sigma_max, through the multiplied-out factor and fsnap_rowspace_solve's path otherwise.  active (may be NULL =
 * all): columns that take part (zero columns of A_w get beta = 0).  The chain is taken only when the estimate of cond(R_hat)
 * leaves two orders of margin (estimate x rcond < 1e-2: the estimators are lower bounds x safety factors, not bounds);
 * otherwise the multiplied-out factor is judged with provable Frobenius bounds and, if those cannot exclude a truncation,
 * by the SVD.  info[4] = {1 if solved through the chain, estimate of ||R_hat||, of ||R_hat^-1||, their product}.  Host side,
 * no context needed. */
int fsnap_rowspace_chain(int64_t K, int64_t nfac, const double* R, const unsigned char* active, const double* z, double rcond,
                         double* beta, int* rank, double* info);

/* Grouped error statistics of Solver.error_analysis (solver.py:108-133: the function applied to every
 * (Groups, Testing, Row_Type) group of the DataFrame, solver.py:391-405) for the resident rows and weights:
 * cat[m] (host) = category id of each row in [0, ncat) (negative = skip; NULL = the categories of the previous call
 * are still valid: re-weighting loops re-use them), beta = coefficients.
 * stats[ncat][10] (host) = n, count_nonzero(w), sum t, sum w t, sum|r|, sum r^2, sum (t - mean t)^2,
 * sum|w r|, sum (w r)^2, sum (w t - sum(w t)/n_w)^2 with r = t - a.beta; mae = sum|r| / n,
 * rmse = sqrt(sum r^2 / n), rsq = 1 - sum r^2 / sum (t - mean)^2 (weighted: w_mae = sum|w r| / n,
 * w_rmse = sqrt(sum (w r)^2 / n_w), ...).  Predictions (GEMV) and both reduction passes run on the GPU. */
int fsnap_error_stats(fsnap_ctx* ctx, const double* beta, const int32_t* cat, int ncat, double* stats);

/* ---- multi-GPU: one process per GPU; RCCL or one-shot peer-to-peer over xGMI -------- */

/* The reference's data-parallel form of this path (examples/library/transpose_trick/example.py:230-254): every MPI
 * rank accumulates c += aw.T aw, d += aw.T bw over ITS configurations, then comm.Allreduce(c), comm.Allreduce(d)
 * (:245-246) and one solve.  Here a rank is a process that owns one context (= one GPU) holding the rows of its
 * configurations (config i -> rank i % nranks, fitsnap3lib/parallel_tools.py:612-651); the two Allreduce calls are
 * ONE in-place ncclAllReduce(ncclDouble, ncclSum) of the packed statistics on the context's stream.  librccl is
 * loaded on the first call of this group (dlopen), never at library load. */
#define FSNAP_COMM_ID_BYTES 128

/* Every wait behind a collective (fsnap_comm_init itself, the host-buffer collectives, the solve that follows the
 * all-reduce of a fit) is bounded by the environment variable FSNAP_COMM_TIMEOUT (seconds, default 300): when a peer died
 * or never arrived the call returns FSNAP_E_HIP with one line in fsnap_last_error naming rank, world size and the wait
 * that ran out, instead of hanging; the context then aborts its communicator (ncclCommAbort) when it is destroyed.
 * A failure that only ONE rank sees before a collective of fsnap_fit_dist / fsnap_lstsq_rows (wrong K, no weights, an
 * allocation that failed) does not keep that rank out of the collective: it contributes NaN statistics, so that every
 * rank returns FSNAP_NUM_NONFINITE from the same call and the failing rank returns its own error. */

/* Rank 0: create the communicator id (ncclGetUniqueId).  The caller distributes the 128 bytes to every rank by
 * whatever it has -- mpi4py comm.bcast on the reference side, a file or a socket in fitsnap_amd/rendezvous.py.
 * With FSNAP_DIST_TRANSPORT=p2p in the environment the id is one of the peer-to-peer transport (next entry). */
int fsnap_comm_id(char* id);

/* Rank 0: the id of a PEER-TO-PEER communicator (csrc/fsnap_p2p.cpp) -- the second transport, for the GPUs of ONE node.
 * The transport travels with the id: fsnap_comm_init recognises it, so every rank takes the same one.  Every rank
 * exports a double-buffered window of device memory through a hipIpc handle (exchanged, like the joins, through a POSIX
 * shared-memory segment named after the id); the all-reduce of a fit is then ONE launch per rank: own statistics ->
 * own window, a flag pushed into every peer's window, a bounded wait for the peers' flags, the sum of the N windows in
 * RANK ORDER (bit-identical on every rank).  The host-buffer collectives below go through mailboxes in the same
 * segment (no launch, no staging).  hipIpc handles open between processes that share a device, so this transport also
 * runs N ranks on ONE GPU -- what RCCL refuses -- which is how the N > 1 paths are tested on one-GPU boxes.
 * FSNAP_P2P_SLOT_MB (default 24; larger payloads travel in pieces) / FSNAP_P2P_MAILBOX_MB (default 4) size the window
 * slots and the mailboxes; needs HSA_ENABLE_IPC_MODE_LEGACY=0 where the host driver only supports dmabuf IPC. */
int fsnap_comm_id_p2p(char* id);

/* Collective: join the communicator of `nranks` ranks as `rank` (ncclCommInitRank on the context's device, or the
 * peer-to-peer rendezvous when `id` comes from fsnap_comm_id_p2p). */
int fsnap_comm_init(fsnap_ctx* ctx, int nranks, int rank, const char* id);
int fsnap_comm_destroy(fsnap_ctx* ctx);

/* *transport = 0 (no communicator), 1 (RCCL), 2 (peer-to-peer). */
int fsnap_comm_transport(fsnap_ctx* ctx, int* transport);

/* *nranks / *rank of the context's communicator (1 / 0 without one); either pointer may be NULL. */
int fsnap_comm_info(fsnap_ctx* ctx, int* nranks, int* rank);

/* In-place sum over the ranks of n doubles in DEVICE memory (e.g. the packed statistics of fsnap_normal_eq_async),
 * asynchronous on the context's stream: comm.Allreduce(c), comm.Allreduce(d) of transpose_trick/example.py:245-246. */
int fsnap_allreduce_device(fsnap_ctx* ctx, double* d_buf, int64_t n);

/* Same for n doubles in HOST memory (staged through HBM; synchronous).  op: 0 = sum, 1 = max, 2 = min.  The scalar
 * reductions around a fit: row / configuration counts (parallel_tools.py:562-577), the refinement right-hand side,
 * the max-over-ranks wall time of the benchmark. */
int fsnap_allreduce_host(fsnap_ctx* ctx, double* buf, int64_t n, int op);

/* Broadcast nbytes of host memory from rank `root` (comm.bcast, parallel_tools.py:579-592); synchronous. */
int fsnap_bcast_host(fsnap_ctx* ctx, void* buf, int64_t nbytes, int root);

/* recv[rank * nbytes ...] = send of every rank (equal sizes; comm.allgather of the fixed-size error tables and of the
 * pickled row-label lists, parallel_tools.py:426-441); host memory, synchronous. */
int fsnap_allgather_host(fsnap_ctx* ctx, const void* send, int64_t nbytes, void* recv);

/* All ranks reach this point and the context's stream is idle (comm.Barrier, parallel_tools.py:245-249). */
int fsnap_barrier(fsnap_ctx* ctx);

/* One call per fit of a multi-GPU job (the whole of transpose_trick/example.py:230-254 for resident rows): this
 * rank's fused statistics, in-place all-reduce on the same stream, then the K x K solve of fsnap_solve_device -- on
 * EVERY rank (the solve is deterministic and the ranks hold bit-identical sums, so no broadcast of beta is needed;
 * the host layer keeps the reference's "fit on rank 0" contract).  K must be given because a rank may own no rows
 * (it then contributes zeros).  Without a communicator: FSNAP_E_STATE (a single-GPU fit is fsnap_fit_resident; there is
 * no silent fallback that would fit one rank's shard).  *d_packed (may be NULL) receives the address of the reduced
 * statistics (context-owned device memory, valid until the next fit). */
int fsnap_fit_dist(fsnap_ctx* ctx, int kind, double param, int64_t K, double* beta, int* rank, double* rcond_est,
                   double** d_packed);

/* ---- raw device memory for callers without a HIP binding of their own --------------- */

/* hipMalloc / hipFree on the context's device; fsnap_dev_sync waits for the context's stream. */
int fsnap_dev_alloc(fsnap_ctx* ctx, int64_t nbytes, void** d_ptr);
int fsnap_dev_free(fsnap_ctx* ctx, void* d_ptr);
int fsnap_dev_sync(fsnap_ctx* ctx);
/* Synchronous copies between host memory and memory of fsnap_dev_alloc (or any device pointer of this GPU). */
int fsnap_dev_upload(fsnap_ctx* ctx, void* d_dst, const void* h_src, int64_t nbytes);
int fsnap_dev_download(fsnap_ctx* ctx, void* h_dst, const void* d_src, int64_t nbytes);

/* ---- measurement ------------------------------------------------------------------ */

/* HIP-event timings of the last fsnap_normal_eq* call, milliseconds:
 * ms[0] = SYRK kernel, ms[1] = partial reduction kernel, ms[2] = last H2D upload,
 * ms[3] = last stand-alone weighting kernel, ms[4] = last predict kernel; about the last large fsnap_upload_rows:
 * ms[5] = GB/s at which the host filled the first page-locked slots (0: not probed), ms[6] = 1 if the whole matrix went
 * through the page-locked double buffer.
 * Synchronises the context's stream.  n = number of entries of ms to fill (<= 8). */
int fsnap_timing(fsnap_ctx* ctx, double* ms, int n);

/* Kernel times (ms) of the last n event-bracketed fits (option "timing_every") of this context, oldest first (n <= 256): syrk_ms[i] = SYRK kernel,
 * reduce_ms[i] (may be NULL) = partial reduction.  HIP events on the kernels' stream, read after the fact, so a
 * timed loop does not have to synchronise for its measurements.  Synchronises the context's stream. */
int fsnap_timing_history(fsnap_ctx* ctx, double* syrk_ms, double* reduce_ms, int n);

/* For the same n event-bracketed fits: allreduce_ms[i] = time between the end of this rank's partial reduction and the end
 * of the collective of fsnap_fit_dist on this rank's stream (ncclAllReduce, or the peer-to-peer kernel) -- it
 * includes waiting for the slowest peer -- or -1 for a fit without a collective.  Synchronises the context's stream. */
int fsnap_timing_history_comm(fsnap_ctx* ctx, double* allreduce_ms, int n);

/* How many SYRK launches this context has made (*launches) and how many of them were bracketed by events (*sampled:
 * what fsnap_timing_history can return); either pointer may be NULL.  Measurement plumbing of bench.py, no reference
 * counterpart. */
int fsnap_timing_count(fsnap_ctx* ctx, int64_t* sampled, int64_t* launches);

/* Launch geometry of the SYRK kernel for the current rows: info[0] = workgroups,
 * info[1] = threads per workgroup, info[2] = 4-row chunks per row-wave (kernel 1) / per
 * workgroup (kernel 1L) / per wave (tiled), info[3] = NB (16-column blocks), info[4] =
 * split (kernel 1) or waves per workgroup (kernel 1L), info[5] = compute units of the
 * device, info[6] = kernel id (1 wave-triangle, 2 LDS-shared, 3 one-wave triangle 1A, 4 packed
 * wave-triangle 1P, 5 workgroup triangle 1Q -- info[2] is then chunks per WORKGROUP; 7 short-system
 * kernel 1S -- info[0] = row chunks (two workgroups each), info[2] = ROWS per chunk; tiled:
 * superblock pairs), info[7] = row splits (tiled kernel) / 1 when the kernel packs the
 * per-row pairs itself. */
int fsnap_launch_info(fsnap_ctx* ctx, int64_t* info, int n);

#ifdef __cplusplus
}
#endif
#endif /* FSNAP_HIP_H */
