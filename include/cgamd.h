/* cgamd.h -- extended C ABI of the MI355X-native CG solver (gfx950, HIP).
 *
 * Everything here is plain C: pointers, sizes, opaque handles.  No torch types.
 * Device pointers are ordinary `void*` values in the HIP address space (e.g.
 * torch.Tensor.data_ptr()); `stream` arguments are hipStream_t passed as void*
 * (0/NULL = the context's own stream).
 *
 * The API mirrors the reference's operator surface for the CG hot path:
 *   reference cl.py:16-31   initialize_cl_environment*, get_gpu_devices -> cgamd_ctx_*
 *   reference cl.py:33-42   kernels {'axpy','aypx','spmv','sub','vdot'}  -> cgamd_{axpy,aypx,spmv,sub,vdot}
 *   reference cl.py:44-200 / clcg.c:111-466   CG()/cg()                  -> cgamd_solver_* and cg()
 * plus what SURVEY §8(f) asks for: a persistent handle (matrix stays resident
 * across solves), the residual history the reference computes but drops
 * (clcg.c:274-292,384-387), status codes, and row-partitioned multi-GPU CG.
 *
 * Value types: the reference is fp32/complex64 only (clcg.h:3-5); f64/c128 are
 * added for the headline metric.  Complex values are interleaved (re,im).
 * Multiple right-hand sides are RHS-major: element i of RHS r at [i + r*size].
 *
 * All functions return CGAMD_OK (0) or a negative/positive cgamd_status; the
 * message of the last failure on the calling thread is cgamd_last_error().
 * There is no CPU fallback anywhere: without a HIP device every compute entry
 * fails with CGAMD_ERR_NO_DEVICE.
 */
#ifndef CGAMD_H
#define CGAMD_H
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum { CGAMD_F32 = 0, CGAMD_F64 = 1, CGAMD_C64 = 2, CGAMD_C128 = 3 } cgamd_dtype;

typedef enum {
    CGAMD_OK = 0,
    CGAMD_ERR_INVALID = 1,    /* bad argument (NULL, negative size, misaligned pointer, bad dtype) */
    CGAMD_ERR_NO_DEVICE = 2,  /* no HIP device / runtime unavailable */
    CGAMD_ERR_HIP = 3,        /* a HIP runtime call failed */
    CGAMD_ERR_ALLOC = 4,
    CGAMD_ERR_IO = 5,         /* Matrix-Market file problems */
    CGAMD_ERR_COMM = 6,       /* RCCL failure / not initialised */
    CGAMD_ERR_STATE = 7       /* call order (e.g. iterate before set_rhs) */
} cgamd_status;

typedef struct cgamd_ctx cgamd_ctx;
typedef struct cgamd_solver cgamd_solver;

const char *cgamd_last_error(void);
int cgamd_version(void);
size_t cgamd_dtype_size(int dtype);

/* run-time configuration: 15 keys (INTEGRATION.md section 6 has the table).  Loop selection: "resident" (1; 0 = launched loops
 * only, 2 = cross-XCD form), "resident_min" (8), "resident_wide" (1), "resident_wide_min" (16), "resident_claim_ms" (200),
 * "two_launch" (1), "spmm_rowmajor" (1; 2 = every supported width, 0 = never).  Matrix stream: "index_codes" (1),
 * "index_codes16" (1), "index_codes_min_mb" (32), "pad_rows" (1).  Placement / cache policy: "spmv_nt", "vec_nt" (-1 = by
 * working-set size), "spmv_cycle" (64 row blocks per XCD turn), "vec_grid" (0 = auto).  A handle keeps the configuration it
 * was created under (snapshot at create); the call is thread-safe.  Unknown key: CGAMD_ERR_INVALID.  Keys that start with
 * "dev." are test / rehearsal hooks of this repository's own suite and scripts, not part of the interface. */
int cgamd_tune(const char *key, int value);

/* ---- devices / context (reference cl.py:16-31) -------------------------- */
int cgamd_device_count(void);                                /* <0 on error */
int cgamd_device_name(int device, char *buf, size_t buflen);
int cgamd_ctx_create(int device, cgamd_ctx **out);           /* owns one HIP stream + workspace */
int cgamd_ctx_destroy(cgamd_ctx *ctx);
int cgamd_ctx_set_stream(cgamd_ctx *ctx, void *stream);      /* borrow caller's stream (torch) */
void *cgamd_ctx_stream(cgamd_ctx *ctx);
int cgamd_ctx_device(cgamd_ctx *ctx);
int cgamd_ctx_synchronize(cgamd_ctx *ctx);

/* device memory helpers for hosts without torch (ctypes + numpy only) */
int cgamd_malloc(cgamd_ctx *ctx, size_t bytes, void **dptr);
int cgamd_free(cgamd_ctx *ctx, void *dptr);
int cgamd_memcpy_h2d(cgamd_ctx *ctx, void *dst, const void *src, size_t bytes);   /* synchronous */
int cgamd_memcpy_d2h(cgamd_ctx *ctx, void *dst, const void *src, size_t bytes);   /* synchronous */
int cgamd_memcpy_d2d(cgamd_ctx *ctx, void *dst, const void *src, size_t bytes);   /* async on ctx stream */
int cgamd_memset(cgamd_ctx *ctx, void *dst, int value, size_t bytes);             /* async on ctx stream */

/* ---- the five kernels of the hot path, device pointers, async on ctx stream
 * spmv : y[row + r*size] = sum_j aValues[j] * x[aCols[j] + r*size]
 *        (reference kernel/real/spmv.cl:5-50, kernel/complex/spmv.cl:7-53)
 * vdot : result[r] = sum_i a[i + r*size] * b[i + r*size]   (UNCONJUGATED,
 *        reference kernel/complex/vdot.cl:15; includes the final reduction the
 *        reference leaves to the host, clcg.c:274-279); result is a DEVICE array
 * axpy : y += a[r]*x (aSign != 0)  /  y -= a[r]*x (aSign == 0); a is a DEVICE array
 *        (reference kernel/real/axpy.cl:2-17)
 * aypx : y = a[r]*y + x            (reference kernel/real/aypx.cl:2-10)
 * sub  : result = a - b            (reference kernel/real/sub.cl:2-12)
 */
int cgamd_spmv(cgamd_ctx *ctx, int dtype, int size, long long nnz, const void *aValues,
               const int *aPointers, const int *aCols, const void *x, void *y, int nRHS);
int cgamd_vdot(cgamd_ctx *ctx, int dtype, int size, const void *a, const void *b, void *result, int nRHS);
int cgamd_axpy(cgamd_ctx *ctx, int dtype, int size, const void *x, void *y, const void *a, int aSign, int nRHS);
int cgamd_aypx(cgamd_ctx *ctx, int dtype, int size, const void *x, void *y, const void *a, int nRHS);
int cgamd_sub(cgamd_ctx *ctx, int dtype, int size, const void *a, const void *b, void *result, int nRHS);

/* ---- persistent solver handle ------------------------
 * flags for cgamd_solver_create */
#define CGAMD_MATRIX_ON_DEVICE 1   /* aValues/aPointers/aCols are device pointers, borrowed (not copied) */
#define CGAMD_NO_GRAPH 2           /* plain stream launches instead of hipGraph replay */
#define CGAMD_UNFUSED 4            /* reference op structure: spmv, vdot, axpy, axpy, vdot, aypx (6 kernels) */
#define CGAMD_DIST_NO_OVERLAP 32    /* cgamd_dist_create: exchange first, then one SpMV (no interior/boundary overlap) */
#define CGAMD_DIST_P2P 64            /* cgamd_dist_create: no RCCL; peers write each other's IPC mailboxes (attach_p2p) */
#define CGAMD_DIST_GRAPH 8         /* cgamd_dist_create: replay each iteration (incl. RCCL ops) from a hipGraph */
#define CGAMD_DIST_P2P_STAGED 128   /* with CGAMD_DIST_P2P: separate push / wait+unpack launches and all-reduce launches (7 per
                                     * iteration) instead of the default four-launch iteration (push and wait inside the SpMV
                                     * launch, halo read in place from the mailbox, beta all-reduce inside the aypx launch) */

#define CGAMD_DIST_SINGLE_REDUCTION 256   /* cgamd_dist_create: the single-reduction form of the recurrence (Chronopoulos-Gear: w = A r,
                                     * ONE global exchange of {r.r, w.r} per iteration instead of two, two launches per iteration with
                                     * CGAMD_DIST_P2P): the same iterates in exact arithmetic, different rounding -- opt-in, held to a
                                     * stated tolerance against the reference's iterates, not bit for bit (csrc/cg1.hip) */

#define CGAMD_DIST_RESIDENT 512      /* cgamd_dist_create: iterate() calls of at least `resident_wide_min` iterations run in ONE launch (csrc/slab.hip:
                                     * vectors in registers, matrix streamed, the two reductions as in-launch all-gathers) where the rank's
                                     * slab fits (up to ~3M rows of at most 8 entries on average, not complex128); the reference's
                                     * recurrence, results held to the oracle like the chip-wide resident loop's.  cgamd_dist_loop_launches()
                                     * reports 0 when it applies; otherwise the flag changes nothing */

int cgamd_solver_create(cgamd_ctx *ctx, int dtype, int size, long long nnz, const void *aValues,
                        const int *aPointers, const int *aCols, int nRHS, int flags, cgamd_solver **out);
int cgamd_solver_destroy(cgamd_solver *s);
/* new values / pattern of the SAME size (size, nnz, nRHS, dtype) into a handle that owns its matrix (created from host
 * arrays): keeps allocations, stream and -- when the row pointers are unchanged -- the plan and the captured graphs.
 * The next call must be cgamd_solver_set_rhs. */
int cgamd_solver_reload_matrix(cgamd_solver *s, const void *aValues, const int *aPointers, const int *aCols);
/* b, x0: nRHS*size values, host (on_device=0) or device (on_device=1) memory; x0 may be NULL (zeros).
 * Computes r = b - A x0, d = r, delta0 = r.r  (reference clcg.c:255-292) and resets the iteration count. */
int cgamd_solver_set_rhs(cgamd_solver *s, const void *b, const void *x0, int on_device);
/* Run exactly nIterations iterations (reference clcg.c:297-419).  Handles on a LAUNCHED loop (cgamd_solver_loop_launches() >= 2)
 * only enqueue: asynchronous, no host sync.  Handles on a RESIDENT loop (loop_launches() 0 or 1, calls of at least
 * "resident_min" / "resident_wide_min" iterations) synchronise: the call takes the GPU's resident-launch lock (one resident
 * grid per GPU at a time, across processes; at most "resident_claim_ms" of waiting), launches, and waits for the kernel before
 * it releases the lock -- so it returns with the iterations done.  When the lock or the CUs cannot be had in that time the call
 * runs the same iterations on the handle's launched loop instead (asynchronous again); results are those of that loop. */
int cgamd_solver_iterate(cgamd_solver *s, int nIterations);
/* Tolerance stop on the device (one right-hand side; handles whose loop is resident, cgamd_solver_loop_launches() < 2): runs until
 * sqrt|r.r| < tol (or NaN), at most maxIterations; *iterations_run = iterations of this call; x is the iterate of exactly that
 * many (reference: the `tol` loop of p_h-PY_C-CL.py:1338-1369).  CGAMD_ERR_STATE if the handle runs a launched loop. */
int cgamd_solver_iterate_tol(cgamd_solver *s, int maxIterations, double tol, int *iterations_run);
/* nIterations iterations with plain launches and a HIP event pair around every SpMV launch on the solver's stream;
 * returns the average in-loop SpMV duration (and optionally the average iteration time), in ms.  Synchronises. */
int cgamd_solver_iterate_timed(cgamd_solver *s, int nIterations, float *spmv_ms_avg, float *iter_ms_avg);
/* copy the current iterate; synchronises the stream when on_device == 0 */
int cgamd_solver_get_x(cgamd_solver *s, void *x, int on_device);
/* residual history: entry k (k = 0..iterations done) holds delta_k[r] for r < nRHS, value type = dtype.
 * Synchronises.  Returns the number of entries written (<= max_entries) or a negative status. */
int cgamd_solver_history(cgamd_solver *s, void *history, int max_entries);
int cgamd_solver_iterations_done(cgamd_solver *s);
/* device pointers of the solver's resident state (for zero-copy inspection): which = 0:x 1:r 2:d 3:q */
void *cgamd_solver_vector(cgamd_solver *s, int which);
/* leading dimension (values between consecutive right-hand sides) of the handle's own vectors: `size` rounded up to a whole number
 * of 16-byte packs (2 values in fp64 / complex64, 4 in fp32) -- the handle carries such systems with 1-3 empty rows appended so that
 * every right-hand side stays 16-byte aligned; b / x0 / x in the caller's arrays keep stride `size` (tuning key "pad_rows" 0: as passed) */
int cgamd_solver_ld(cgamd_solver *s);
/* Diagonal (Jacobi) preconditioning -- the reference's PCG(A, b, M) with a diagonal CSR M, z = M.dot(r)
 * (helmFE_var.py:546-586).  m: `size` values of the solver's type (1/diag(A) for Jacobi), host or
 * device; NULL removes it.  Takes effect at the next cgamd_solver_set_rhs; history keeps holding r.r (the stopping
 * test of the reference, helmFE_var.py:580-584), the recurrence uses rho = r.z.  Handles the chip-wide resident loop can take over
 * (cgamd_solver_loop_launches() == 1 once the preconditioner is set) run the recurrence inside that loop, with the bits of the
 * four-launch PCG loop of the same handle; cgamd_solver_iterate_tol then stops it on the device. */
int cgamd_solver_set_preconditioner(cgamd_solver *s, const void *m, int on_device);
/* convenience: set_rhs + iterate + get_x (+ history if non-NULL, (nIterations+1)*nRHS values), host arrays */
int cgamd_solver_solve(cgamd_solver *s, const void *b, void *x, int nIterations, void *history);
/* the solver's SpMV (optionally fused with the d.q partial reduction) on caller vectors -- bench/profiling */
int cgamd_solver_spmv(cgamd_solver *s, const void *x, void *y, int fused_dot);
/* SpMM on the matrix cores (BASELINE config 4, "MFMA tall-B tile path"): Y[size][nRHS] = A * X[size][nRHS] with
 * the right-hand-side block in ROW-MAJOR layout (element i of RHS r at [i*nRHS + r]); f64 with nRHS = 16 or 32, f32 with 16, 32
 * or 64, complex64 with 16 or 32; any CSR matrix.  Solvers created with such a width keep their vectors in this layout
 * internally and run this kernel in the CG loop (cgamd_solver_layout() == 1); set_rhs / get_x / solve still take and
 * return the reference's RHS-major blocks.  cgamd_transpose converts between the two: out[c*rows + r] = in[r*cols + c]. */
int cgamd_solver_spmm_rowmajor(cgamd_solver *s, const void *x, void *y, int nRHS);
/* 0: the handle's vectors (cgamd_solver_vector) are RHS-major [nRHS][size]; 1: row-major [size][nRHS] (decided by the
 * last cgamd_solver_set_rhs) */
int cgamd_solver_layout(cgamd_solver *s);
/* launches per iteration of the loop cgamd_solver_iterate runs for this handle: 0 = the resident loop (small systems: every
 * iteration of a call of at least `resident_min` iterations inside ONE launch, csrc/resident.hip), 1 = its chip-wide form
 * (one launch per call of at least `resident_wide_min` iterations as well; shorter calls take the launched loops of the same handle,
 * which return the same bits), 2 / 3 / 4 / 5 = the
 * loops of DESIGN.md section 4, 8 = the reference's op structure (CGAMD_UNFUSED); negative: error */
int cgamd_solver_loop_launches(cgamd_solver *s);
/* > 0: this handle's single-RHS SpMV reads one-byte column codes instead of aCols (4 -> 1 byte of index traffic per non-zero),
 * the value is the number of distinct (column - row) offsets of the matrix (at most 256; stencil / structured-grid FE matrices
 * have 5 to 27).  Built at create / reload for matrices above 32 MB (tuning key "index_codes_min_mb"; smaller systems run
 * the resident or two-launch loops; "index_codes" 0 disables).  Exact: the kernel rebuilds the same column, results do not change by a bit.
 * 65536: the matrix has more offsets than that, and the SpMV reads 16-bit columns relative to the first column of every 256-row block
 * (2 index bytes per non-zero; any matrix whose row blocks span fewer than 65 536 columns each: banded random patterns, meshes in a
 * bandwidth-reducing order, what Matrix-Market files hold; tuning key "index_codes16" 0 disables).  Exact like the one-byte form.
 * 0: the kernel reads aCols as the reference's does (kernel/real/spmv.cl:21-27). */
int cgamd_solver_index_codes(cgamd_solver *s);
/* distinct matrix entries behind the one-byte VALUE codes of the handle's single-RHS SpMV (matrices of at most 256 distinct entries
 * that also run on one-byte column codes: 2 bytes per non-zero from memory, same bits); 0 = the SpMV reads aValues.
 * (cgamd_tune("dev.value_codes", 0) turns the form off for A/B runs.) */
int cgamd_solver_value_codes(cgamd_solver *s);
/* > 0: the SpMV reads ONE byte per non-zero that names the (column offset, value) pair -- matrices with at most 256 distinct pairs whose
 * longest row fits one batch of the row walk (constant-coefficient stencils: as many pairs as offsets); the value is the number of
 * pairs.  0: it reads the column codes and the value codes (2 bytes), or aCols / aValues. */
int cgamd_solver_joint_codes(cgamd_solver *s);
int cgamd_transpose(cgamd_ctx *ctx, int dtype, int rows, int cols, const void *in, void *out);
/* algorithmic HBM bytes of one SpMV / one CG iteration of this solver (SURVEY §8d formulae: 14 vector passes for the
 * reference's op structure, 11 for its "fused minimum"; the default loop here moves 10) */
long long cgamd_solver_spmv_bytes(cgamd_solver *s);
long long cgamd_solver_iter_bytes(cgamd_solver *s, int fused);
/* the bytes this handle's own kernels move per SpMV / per iteration of its launched loop: index bytes per non-zero as the SpMV
 * reads them (1 with one-byte column codes, 2 with 16-bit block-relative columns, 4 with aCols) and the loop's own vector passes
 * (10 by default).  This is the figure a roofline FRACTION is priced on; the SURVEY 8(d) figures above are the
 * reference's CSR byte model (an "effective" rate). */
long long cgamd_solver_spmv_moved_bytes(cgamd_solver *s);
long long cgamd_solver_iter_moved_bytes(cgamd_solver *s);

/* one-call typed solve on host arrays: cg() generalised to all four dtypes, with history and status */
int cgamd_cg(int dtype, int size, long long nnz, const void *aValues, const void *b, const int *aPointers,
             const int *aCols, void *x, int nRHS, int nIterations, void *history, int device);
/* Wall-clock split of the calling thread's last cgamd_cg() / cg() call, in milliseconds:
 * [0] device state (context, allocations, plan; or the cache check), [1] matrix upload + validation, [2] right-hand side
 * upload + setup kernels, [3] iterations (enqueue + wait), [4] solution download, [5] 1.0 when the call reused the
 * thread's cached device state.  cgamd_cg keeps one context + handle per calling thread and reuses them when dtype,
 * size, nonZeros, nRHS and device repeat (the reference rebuilds everything per call, clcg.c:142-214; the call itself
 * stays stateless: the matrix is re-uploaded every time).  cgamd_cg_release_cache() frees the calling thread's cache;
 * environment CGAMD_CG_NO_CACHE=1 disables it. */
int cgamd_cg_last_timing(double *ms6);
int cgamd_cg_release_cache(void);

/* ---- synthetic matrix generators, written straight into device memory -----
 * 7-point 3-D Laplacian (x fastest), Dirichlet, diag 6 / off-diag -1 (SURVEY §8d "M", "C5").
 * Generates global rows [row_begin,row_end) with GLOBAL column indices; canonical CSR.
 * Pass aPointers==NULL to query nnz of the slab via *nnz_out. */
int cgamd_gen_laplace3d(cgamd_ctx *ctx, int dtype, int nx, int ny, int nz, long long row_begin,
                        long long row_end, void *aValues, int *aPointers, int *aCols, long long *nnz_out);
/* 5-point 2-D Laplacian N x N, diag 4 / off-diag -1 (reference Poisson(), p_h-PY_C-CL.py:1642-1682) */
int cgamd_gen_poisson2d(cgamd_ctx *ctx, int dtype, int N, void *aValues, int *aPointers, int *aCols,
                        long long *nnz_out);

/* P1 finite-element Helmholtz matrices of the reference's drivers, Nhoriz x Nvert nodes, complex symmetric, 7 entries per interior
 * row (canonical CSR; dtype complex64 or complex128; values evaluated in complex double in the reference's operation order):
 *   cgamd_gen_helm_fe_var: helmFE_var(N, omega, C, rho, Nhoriz, Nvert) of helmFE_var.py:9-331 -- BASELINE config 3 is N = Nhoriz =
 *     Nvert = 500, omega = 12, C = 1, rho = 0.15.  C: (Nvert - 1) x (Nhoriz - 1) wave speeds, row-major, HOST doubles (NULL = all 1);
 *   cgamd_gen_local_rect:  local_rect(N, k, eps, eta, L, Nhoriz, Nvert) of p_h-PY_C-CL.py:1439-1639 -- the sub-domain matrices
 *     as_prec hands to cg().
 * Pass aPointers == NULL to query the number of non-zeros via *nnz_out. */
int cgamd_gen_helm_fe_var(cgamd_ctx *ctx, int dtype, int N, double omega, const double *C, double rho, int Nhoriz, int Nvert,
                          void *aValues, int *aPointers, int *aCols, long long *nnz_out);
int cgamd_gen_local_rect(cgamd_ctx *ctx, int dtype, int N, double k, double eps, double eta, double L, int Nhoriz, int Nvert,
                         void *aValues, int *aPointers, int *aCols, long long *nnz_out);

/* Right-hand sides of the reference's Helmholtz drivers on an N x N node grid (helmFE_var.py:333-389), written to device memory
 * as N * N values, entry (row, col) at row * N + col: kind 0 = rhs(N, k) (plane-wave boundary data; complex types only), 1 = rhsL(N, k)
 * (k^2 on the left boundary without its corners), 2 = rhsA(N, k) (k^2 on the four boundary lines: BASELINE config 3 uses rhsA(500, 12)). */
int cgamd_gen_rhs(cgamd_ctx *ctx, int dtype, int kind, int N, double k, void *b);

/* ---- Matrix-Market ingest (reference main.c:20-33 via BeBOP) ---------------
 * Reads a coordinate file (real/complex/integer/pattern x general/symmetric/hermitian/skew-symmetric),
 * expands symmetric storage, sums duplicates, converts 1-based -> 0-based CSR with sorted columns.
 * Values are returned as double (is_complex=0) or interleaved double pairs (is_complex=1).
 * Free the three arrays with cgamd_mm_free. */
int cgamd_mm_read(const char *path, int *size, long long *nnz, int *is_complex, double **values,
                  int **pointers, int **cols);
void cgamd_mm_free(void *p);

/* ---- row-partitioned multi-GPU CG (one process per GPU, RCCL over xGMI) ----
 * The host (torch.distributed) builds the partition plan; this library runs the loop.
 * See DESIGN.md "Multi-GPU" for the plan layout. */
typedef struct cgamd_dist cgamd_dist;
int cgamd_comm_unique_id(void *id128);      /* rank 0: 128 bytes to broadcast */
/* A peer may be the rank itself (periodic coupling inside one partition; also how a single GPU exercises the
 * exchange): its send list is then gathered into its own halo slots through ncclSend/ncclRecv to self. */
int cgamd_dist_create(cgamd_ctx *ctx, const void *id128, int rank, int nranks, int dtype,
                      int n_local, int n_halo, long long nnz_local, const void *aValues,
                      const int *aPointers, const int *aCols, /* device, cols in [0,n_local+n_halo) */
                      int n_peers, const int *peer_rank, const int *send_count, const int *recv_count,
                      const int *send_index /* device: concatenated local row ids to send */,
                      int flags, cgamd_dist **out);
int cgamd_dist_destroy(cgamd_dist *d);
int cgamd_dist_set_rhs(cgamd_dist *d, const void *b_local, const void *x0_local);   /* device pointers */
int cgamd_dist_iterate(cgamd_dist *d, int nIterations);
int cgamd_dist_get_x(cgamd_dist *d, void *x_local);                                  /* device pointer */
int cgamd_dist_history(cgamd_dist *d, void *history, int max_entries);
int cgamd_dist_synchronize(cgamd_dist *d);
/* Peer-to-peer backend (CGAMD_DIST_P2P): instead of RCCL, every rank owns an uncached IPC-shared mailbox
 * (16 KiB header + n_halo values) that its peers write over xGMI; all-reduces are sums in rank order of values
 * deposited in per-rank slots (bitwise identical on all ranks).  Sequence: mailbox_alloc on every rank -> gather
 * the 64-byte handles of all ranks (torch.distributed) -> dist_create(..., id128 = NULL, flags | CGAMD_DIST_P2P)
 * -> attach_p2p(handles[nranks*64], dst_offset[n_peers] = where my entries land in each peer's halo area). */
int cgamd_p2p_mailbox_alloc(cgamd_ctx *ctx, long long halo_values, int dtype, void **mailbox, void *handle64);
int cgamd_p2p_mailbox_free(cgamd_ctx *ctx, void *mailbox);
int cgamd_dist_attach_p2p(cgamd_dist *d, void *my_mailbox, const void *handles, const int *dst_offset);
int cgamd_dist_p2p_error(cgamd_dist *d);
/* CGAMD_DIST_RESIDENT on a peer-to-peer handle that has peers: the slab loop publishes d in two buffers of n_local + n_halo values
 * inside every rank's mailbox allocation, behind the halo area (mailbox = 16 KiB header | n_halo values | ds0 | ds1, each part
 * rounded up to 256 bytes), so the peers write their boundary entries into the tail directly.  Allocate the mailbox with
 * halo_values >= n_halo + 2 (n_local + n_halo) + 96 (cgamd_p2p_mailbox_alloc), attach, then call this with that halo_values and every
 * rank's n_local / n_halo (all-gathered by the host).  OK whether or not the loop applies; cgamd_dist_loop_launches() == 0 tells. */
int cgamd_dist_enable_resident(cgamd_dist *d, long long mailbox_values, const int *rank_n_local, const int *rank_n_halo);
/* as cgamd_solver_index_codes, for this rank's local matrix (the halo columns of a slab partition sit at constant offsets) */
int cgamd_dist_index_codes(cgamd_dist *d);
/* stream operations per iteration of the loop this handle runs (kernel launches, plus RCCL calls with that backend) */
int cgamd_dist_loop_launches(cgamd_dist *d);
/* number of ranks of the RCCL communicator behind this handle as RCCL itself reports it (ncclCommCount);
 * 0 when the handle has no communicator (peer-to-peer backend, or one rank without peers) */
int cgamd_dist_comm_ranks(cgamd_dist *d);

#ifdef __cplusplus
}
#endif
#endif /* CGAMD_H */
