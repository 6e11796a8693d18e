// Host-side internal interfaces between the kernel launchers (spmv.hip, vector.hip, p2p.hip, cg1.hip, index_codes.hip, rowmajor.hip, resident.hip, slab.hip), the solver
// (solver.cpp), the C ABI (api.cpp) and the multi-GPU loop (dist.cpp).
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/cgamd.h"

namespace cgamd {

// ---- error plumbing ---------------------------------------------------------
void set_error(const std::string &msg);
int fail(int status, const std::string &msg);
#define CG_HIP(expr)                                                                         \
    do {                                                                                     \
        hipError_t e__ = (expr);                                                             \
        if (e__ != hipSuccess)                                                               \
            return ::cgamd::fail(e__ == hipErrorNoDevice || e__ == hipErrorInvalidDevice     \
                                     ? CGAMD_ERR_NO_DEVICE : CGAMD_ERR_HIP,                  \
                                 std::string(#expr) + ": " + hipGetErrorString(e__));        \
    } while (0)

inline size_t dtype_size(int dt) { return dt == 0 ? 4 : dt == 1 ? 8 : dt == 2 ? 8 : dt == 3 ? 16 : 0; }
inline size_t acc_size(int dt) { return (dt == 0 || dt == 1) ? 8 : 16; }
inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Launch geometry of the persistent streaming kernels.
constexpr int kBlock = 256;          // threads per work-group (4 waves)
constexpr int kMaxGrid = 2048;       // 256 CUs x 8 resident work-groups
constexpr int kQuadsPerThread = 2;   // SpMV: 2 x 4 non-zeros per thread per chunk -> 2048 nnz / chunk

struct SpmvPlan {
    int grid = 0;        // work-groups launched (multiple of 8 when >= 8)
    int row_blocks = 0;  // ceil(n / kBlock)
    int max_span = 0;    // largest 4-aligned nnz span of a kBlock-row slice (0 = unknown -> generic kernel)
    int kind = 0;        // kernel chosen by finalize_spmv_plan (0 generic, 5 row-block, 6 its SpMM form, 7 chunked row-block)
    int chunk_span[5] = {0, 0, 0, 0, 0};   // largest 4-aligned span of a 128 / 64 / 32 / 16 / 8-row slice
    bool wide = false;   // kind 6 only: small system, one work-group per (row block, RHS) runs the single-RHS kernel
    int max_row = 0;     // longest row (0 = unknown): the row-block kernel's batch length follows it
    int max_quad = 0;    // most non-zeros in 4 consecutive rows starting at a multiple of 4 (row-major SpMM: K-steps per quad)
    int lpr = 1;         // kind 7: lanes per row (2, 4, 8, 16, 32) = chunks per 256-row block
    int n_partials = 0;  // fused-dot partials per RHS written by that kernel
    int fold_max = 0;    // most d.q partials the folded alpha sums (0 = default; handles the chip-wide resident loop takes over: 4096)
    int nt = 1;          // matrix stream loaded non-temporally (finalize_spmv_plan: off when the matrix fits the Infinity Cache)
    int vec_nt = 3;      // axpy2_dot streaming hints (see Tuning::vec_nt), resolved by finalize_spmv_plan
    // one-byte column codes of the single-RHS row-block kernel (build_index_codes; owned by the caller, not by the plan)
    const unsigned char *codes = nullptr;   // [nnz + pad]: aCols[j] = row(j) + dict[codes[j]]
    const int *dict = nullptr;              // [256] distinct (column - row) offsets of the matrix
    const int *codes_for = nullptr;         // the aCols array the codes were made from
    bool codes16 = false;                   // the codes are 16-bit columns relative to the row block's first column (build_index_codes16)
    // one-byte VALUE codes (build_value_codes; with the one-byte column codes only): aValues[j] == vdict[vcodes[j]]
    const unsigned char *vcodes = nullptr;  // [nnz + pad]
    const void *vdict = nullptr;            // [256] values
    const void *vcodes_for = nullptr;       // the aValues array the codes were made from
    // one-byte JOINT codes (build_joint_codes): (aCols[j] - row, aValues[j]) == (jdict_off[jcodes[j]], jdict_val[jcodes[j]])
    const unsigned char *jcodes = nullptr;  // [nnz + pad]
    const int *jdict_off = nullptr;         // [256] offsets
    const void *jdict_val = nullptr;        // [256] values
};
SpmvPlan make_spmv_plan(int n);
// fills plan->max_span / chunk_span from the matrix structure; synchronises `st`; scratch_dev: >= 32 bytes
int compute_spmv_plan(const int *ptr_dev, const int *cols_dev, int n, int *scratch_dev, hipStream_t st, SpmvPlan *plan);
// device-resident CSR (and an optional device index list with entries in [0, index_bound)): CGAMD_ERR_INVALID on a bad
// pointer array or an out-of-range column / index; synchronises `st`; scratch_dev: >= 4 bytes
int validate_csr_device(int n, long long nnz, int ncols, const int *ptr_dev, const int *cols_dev, const int *index_dev, int n_index,
                        int index_bound, int *scratch_dev, hipStream_t st);
// Column indices as one-byte codes into a dictionary of the matrix's distinct (column - row) offsets: 4 -> 1 byte per
// non-zero of index traffic for every matrix with at most 256 distinct offsets (stencils and FE matrices on structured grids:
// 5 to 27).  Exact: the kernel rebuilds the very same column, so results do not change by a bit.  On success *codes_out
// (nnz + 64 bytes) and *dict_out (256 ints) are device allocations the caller frees; both null when the matrix has more
// offsets than that.  Synchronises `st`.
int build_index_codes(int n, long long nnz, const int *ptr_dev, const int *cols_dev, hipStream_t st, unsigned char **codes_out,
                      int **dict_out, int *distinct_out);
// 16-bit columns relative to the first column of every 256-row block, for matrices with more offsets than the dictionary holds
// whose row blocks span fewer than 65 536 columns: *codes_out 2 nnz + 64 bytes, *base_out one int per row block; null when not codable
int build_joint_codes(int dtype, long long nnz, const unsigned char *codes, const unsigned char *vcodes, const int *dict, const void *vdict,
                      hipStream_t st, unsigned char **jcodes_out, int **joff_out, void **jval_out, int *n_pairs);
int build_value_codes(int dtype, long long nnz, const void *vals_dev, hipStream_t st, unsigned char **vcodes_out, void **vdict_out, int *n_values);
int build_index_codes16(int n, long long nnz, const int *ptr_dev, const int *cols_dev, hipStream_t st, unsigned char **codes_out, int **base_out);
void finalize_spmv_plan(SpmvPlan *plan, int dtype, int nrhs, int n, long long nnz, const void *vals, const int *cols);
constexpr int kChunkBytes = 32 * 1024;      // kind 7: preferred LDS chunk slice (4-5 work-groups per CU)
constexpr int kMaxChunkBytes = 48 * 1024;   //         largest accepted, with 8 lanes per row (3 work-groups per CU)
constexpr int kMaxSpmmSliceBytes = 64 * 1024;   // SpMM forms (nRHS > 1, row-major matrix-core op): largest 256-row LDS slice
constexpr int kMaxSliceBytes = 40 * 1024;   // variant 5: largest 256-row LDS slice (4 work-groups per CU); denser rows -> chunked kernel.
                                            // 27-point f32 (55 KB): 114 us one lane per row, 91 us chunked; 7-point c128 (36 KB): 314 vs 357

// run-time tuning (cgamd_tune); defaults are the shipped configuration.  PUBLIC keys (include/cgamd.h): the 15
// of the first block.  The second block are development hooks (key prefix "dev."): what the tests need to force a loop family or a
// failure, and rehearsal switches; experiments that were decided have no knob any more.
struct Tuning {
    // ---- public
    int index_codes = 1;    // single-RHS row-block SpMV on one-byte column codes (0 = always aCols)
    int index_codes16 = 1;  // ... and, where the matrix has more than 256 offsets, on 16-bit block-relative columns (0 = aCols then)
    int index_codes_min_mb = 32;    // ... for matrices above this size (smaller systems run the resident / two-launch loops, which read aCols)
    int resident = 1;       // systems of at most 32768 rows whose matrix slices fit LDS: all iterations of an iterate() call in ONE launch
                            // (resident.hip); 0 = never, 2 = always in the cross-XCD form (up to 65536 rows; for the any-placement tests)
    int resident_min = 8;   // ... for iterate() calls of at least this many iterations
    int resident_wide = 1;  // systems the one-XCD loop cannot hold (rows of <= 10 entries, up to ~1M rows): chip-wide resident groups; 0 = launched
    int resident_wide_min = 16; // ... for iterate() calls of at least this many iterations (also the slab loop's threshold, slab.hip)
    int resident_claim_ms = 200;   // resident loops: how long a call waits for the GPU's resident-launch lock, and work-groups for their group to
                               // fill (CUs held by other kernels), before the launch gives up untouched and the handle takes the launched loops
    int two_launch = 1;     // small systems: beta / d = beta d + r inside the next SpMV launch, two launches per iteration (0 = three / four)
    int spmm_rowmajor = 1;  // 1: solvers keep the block row-major where that loop is the faster one (f64 x 32); 2: for every supported type
                            // (f32 16/32/64, f64 16/32, complex64 16/32); 0: never
    int pad_rows = 1;       // sizes that are not whole 16-byte packs are carried with 1-3 empty rows appended (0 = as passed)
    int spmv_nt = -1;       // non-temporal matrix loads: 1 on, 0 off, -1 auto = on unless the matrix fits the 256 MB Infinity Cache
    int vec_nt = -1;        // -1 auto (by working-set size); bit0 = x loaded/stored non-temporally, bit1 = q loaded non-temporally
    int spmv_cycle = 64;    // row-block schedule: block-cyclic over the XCDs, cycle length in row blocks (1 = contiguous eighths)
    int vec_grid = 0;       // vector kernels: work-groups, 0 = auto
    // ---- development hooks ("dev." keys): tests, rehearsals, profiling
    int dev_no_fold_alpha = 0;      // 1: small systems keep the separate cg_alpha launch (the four-launch family at sizes that would fold it)
    int value_codes = 1;            // one-byte value codes on top of the one-byte column codes where the matrix has at most 256 distinct entries (0 = off: A/B, tests)
    int dev_joint_codes = 1;        // value-coded SpMV: one byte per non-zero naming the (offset, value) pair where at most 256 pairs occur (0 = two bytes: A/B)
    int dev_vc_pipe = 1;            // value-coded SpMV: gathers pipelined across the row blocks of a work-group (0 = one block at a time: A/B)
    int dev_generic_spmv = 0;       // 1: the generic chunked CSR stream for every matrix (the row-block kernels' fallback, tested against them)
    int resident_lock = 1;          // 0: no per-GPU serialisation of resident launches (ranks of ONE job sharing a GPU in a rehearsal)
    int slab_trim = -1;             // slab loop: 1024-row granules a member that pushes to a peer owns fewer than the others (-1 = 2, 0 = equal members)
    int slab_cus = 0;               // slab loop: CUs (= members at most) a handle may use; 0 = all (ranks that share a GPU must fit side by side)
    int resident_test_short_grid = 0; // launch one work-group too few, so that no group can fill (the untouched-fallback tests)
    int resident_wide_rpt = 0;      // rows per thread of the chip-wide loop: 0 = the smallest that fits (4, then 8), or 4 / 8
    int resident_window = 1;        // one-XCD resident loop: stage the column range of a member's rows in LDS (0 = per-non-zero gathers)
    int vec_ppt = 0;                // vector kernels: 16-byte packs per thread the grid is sized for (0 = by size: 1, 2 or 4)
    int spmv_unroll = 0;            // row walk: gathers in flight per lane (0 = fitted to the longest row)
    int spmv_grid = 0;              // generic kernel: work-groups, 0 = auto
    int spmv_slice_kb = 0;          // largest 256-row LDS slice the one-lane-per-row kernel accepts, in KB (0 = kMaxSliceBytes)
    int spmv_chunk_kb = 0;          // chunked row-block kernel: preferred LDS chunk in KB (0 = kChunkBytes)
    int spmv_chunked = 1;           // rows too dense for the row-block kernel: chunked row-block kernel (0 = generic kernel)
    int spmm_ynt = -1;              // row-major SpMM y stores: 0 plain, 1 non-temporal, 2 write-through sc1 (-1 = default: 2)
    int spmm_group = 0;             // SpMM: right-hand sides per register group (0 = equal-width groups of at most 8, 4 for complex128)
    int spmm_rb = 0;                // SpMM: right-hand sides per launch (0 = all in one launch)
    int spmm_wgs = 0;               // row-major SpMM sweep: work-groups per XCD (0 = 64)
    int spmm_lead = 0;              // row-major SpMM sweep: steps a wave may gather ahead of its XCD's slowest (0 = default, -1 = unpaced)
    int spmm_wide_max = -1;         // multi-RHS, RHS-major: largest row_blocks x nRHS for the one-work-group-per-RHS form (-1 = 4096, 0 = never)
};
// g_tune is the process-wide configuration cgamd_tune edits (under a mutex).  Nothing on a compute path reads it
// directly: every solver / distributed handle copies it at creation (tune_snapshot()), and each C-ABI entry installs the
// handle's copy for the calling thread (TuneScope) -- launches of a handle always run with the configuration it was
// created under, whatever other threads set meanwhile (reference threading model: one thread per device, each with its own
// context and solver, p_h-PY_C-CL-multi-GPU.py:2149-2179).  tune() = the installed copy, or a per-thread snapshot of the
// global one for handle-less entries.
extern Tuning g_tune;
Tuning tune_snapshot();
const Tuning &tune();
} // namespace cgamd
#include <functional>
namespace cgamd {
void tune_set(const std::function<void(Tuning &)> &edit);
// the calling thread's next row-block SpMV launch records pair[0] / pair[1] on the dispatch itself (kernel duration)
void set_kernel_event_pair(hipEvent_t *pair);
void thread_hip_setup();      // once per thread: stream-capture interaction mode "relaxed" (see tune.cpp)
struct TuneScope {
    const Tuning *prev;
    explicit TuneScope(const Tuning *t);
    ~TuneScope();
};
int vec_grid(long long n_elems_per_rhs, int dtype, int nrhs = 1);

// ---- kernel launchers (all asynchronous on `st`) ------------------------------
// y = A x for nrhs vectors; x has leading dimension ldx (>= number of columns), y has ldy.
// If partials != nullptr also writes per-work-group partial sums of dvec.y (unconjugated),
// laid out partials[r * plan.grid + wg] in accumulator precision.
int launch_spmv(int dtype, const SpmvPlan &plan, int n, long long nnz, const void *vals, const int *ptr,
                const int *cols, const void *x, long long ldx, void *y, long long ldy, int nrhs,
                const void *dvec, void *partials, hipStream_t st, const int *rb_list = nullptr, int rb_count = 0);
// flag[rb] = 1 when row block rb references a column >= n_local (needs the halo); kBlock rows per block
int launch_halo_flags(int n, const int *ptr, const int *cols, int n_local, int row_blocks, int *flag, hipStream_t st);
// partial sums of a.b -> partials[r*grid + wg]
int launch_dot_partials(int dtype, int n, const void *a, const void *b, long long ld, int nrhs, void *partials,
                        int grid, hipStream_t st);
// result[r] = sum partials (value type)
int launch_reduce_to_value(int dtype, const void *partials, int grid, int nrhs, void *result, hipStream_t st);
int launch_axpy(int dtype, int n, const void *x, void *y, long long ld, const void *a, int sign, int nrhs, hipStream_t st);
int launch_aypx(int dtype, int n, const void *x, void *y, long long ld, const void *a, int nrhs, hipStream_t st);
int launch_sub(int dtype, int n, const void *a, const void *b, void *res, long long ld, int nrhs, hipStream_t st);
// fused x += alpha d ; r -= alpha q ; partials(r.r)
int launch_axpy2_dot(int dtype, int n, const void *d, void *x, const void *q, void *r, long long ld,
                     const void *alpha, int nrhs, void *partials, int grid, hipStream_t st, int vec_nt = 3);

// device-resident scalar state of one CG run
struct CgScalars {
    void *alpha = nullptr;     // T[nrhs]
    void *beta = nullptr;      // T[nrhs]
    void *delta = nullptr;     // T[nrhs]   current delta_new
    void *history = nullptr;   // T[cap][nrhs]
    int *iter = nullptr;       // iterations completed
    int history_cap = 0;
    void *stage = nullptr;     // optional, two-level cg_alpha: acc[nrhs][32] part sums
    unsigned *ticket = nullptr;   //          and one zero-initialised ticket counter per RHS
    // order of the prologue sums over the d.q / r.r partials: 0 = thread-strided; K > 0 = member-blocked (reduce_device.h
    // thread_partials): handles the chip-wide resident loop can take over, K = 256-row (256-pack) blocks of one resident member
    int kdq = 0, krr = 0;
    // chip-wide resident loop only: the diagonal preconditioner and the rho parity buffer of a handle that runs the PCG recurrence
    const void *pcg_m = nullptr;
    void *pcg_rho2 = nullptr;
};
// ten-vector-pass iteration (x update deferred into the aypx launch): see vector.hip
int launch_axpy_dot(int dtype, int n, const void *q, void *r, long long ld, const void *alpha, int nrhs, void *partials, int grid,
                    hipStream_t st, int vec_nt = 3);
int launch_axpy_dot_alpha(int dtype, int n, const void *q, void *r, long long ld, const void *part_dq, int P, const CgScalars &sc,
                          int nrhs, void *partials, int grid, hipStream_t st);
int launch_aypx_beta_x(int dtype, int n, const void *x, void *y, void *xs, long long ld, const void *partials, int P, int nrhs,
                       const CgScalars &sc, hipStream_t st, int vec_nt = 3);
// small systems: alpha = delta / sum(part_dq) in the prologue (three-launch iteration); fold_alpha_ok says when
bool fold_alpha_ok(int n_partials, int fold_max = 0);      // fold_max 0 = the default limit (2048 partials)
// two-launch iteration (spmv.hip "Two-launch iteration"): the SpMV launch computes beta and d_new = beta d_old + r on the
// fly; d_old / d_new are different buffers.  fused2_ok: the plan's row-block kernels apply and the system is small enough
bool fused2_ok(const SpmvPlan &plan, int dtype, int nrhs, const void *vals, const int *cols);
int launch_spmv_fused(int dtype, const SpmvPlan &plan, int n, long long nnz, const void *vals, const int *ptr, const int *cols,
                      const void *d_old, void *d_new, const void *r, void *q, int nrhs, void *part_dq, const void *part_rr, int P,
                      const CgScalars &sc, hipStream_t st);
// delta / beta / history[iter] of the last iteration of an iterate() call of that loop
int launch_cg_tail(int dtype, const void *part_rr, int P, int nrhs, const CgScalars &sc, hipStream_t st);
int launch_axpy2_dot_alpha(int dtype, int n, const void *d, void *x, const void *q, void *r, long long ld, const void *part_dq,
                           int P, const CgScalars &sc, int nrhs, void *partials, int grid, hipStream_t st);
// delta[r] = sum partials ; history[0][r] = delta[r] ; *iter = 0
int launch_cg_delta0(int dtype, const void *partials, int grid, int nrhs, const CgScalars &s, hipStream_t st);
// alpha[r] = delta[r] / sum partials_dq
int launch_cg_alpha(int dtype, const void *partials, int grid, int nrhs, const CgScalars &s, hipStream_t st);
// dn = sum partials_rr ; beta = dn/delta ; delta = dn ; history[++iter] = dn
int launch_cg_beta(int dtype, const void *partials, int grid, int nrhs, const CgScalars &s, hipStream_t st);

// d = beta d + r with beta = (sum of the P r.r partials) / history[iter-1] computed in every work-group's prologue;
// work-group 0 records delta, beta and history[iter] (the cg_beta launch folded into aypx)
int launch_aypx_beta(int dtype, int n, const void *x, void *y, long long ld, const void *partials, int P, int nrhs,
                     const CgScalars &sc, hipStream_t st);

// partials -> accumulator-precision scalar per RHS (all-reduce input); halo pack out[k] = v[index[k]]
int launch_reduce_to_acc(int dtype, const void *partials, int grid, int nrhs, void *out, hipStream_t st);
int launch_pack(int dtype, int count, const int *index, const void *v, void *out, hipStream_t st);

// ---- row-major multi-RHS path (rowmajor.hip): the RHS block is X[n][nrhs], element (i, r) at i*nrhs + r ----------------
// Y = A X on the matrix cores (+ per-RHS d.q partials [nrhs][spmm_rm_grid(n)] in accumulator precision when partials != 0,
// the dot being x.y); f64 with 16 / 32 right-hand sides, f32 with 16 / 32 / 64, complex64 with 16 / 32; any CSR matrix
bool spmm_rm_supported(int dtype, int nrhs, int n);
// work-groups (= fused-dot partials per RHS) the launch will use for this problem
int spmm_rm_grid(int dtype, int nrhs, int n, int max_quad, bool dot);
// max_quad: most non-zeros in 4 consecutive rows (SpmvPlan::max_quad; 0 = unknown), picks the fp64 kernel's K-steps per quad
// pace: kSpmmPaceInts zeroed ints owned by the caller and used by ONE stream at a time, always for the same problem (the sweep's
// per-wave progress words, cumulative over launches; one set for the launches with partials, one for those without), or null =
// unpaced sweep
int launch_spmm_rm(int dtype, int n, long long nnz, const void *vals, const int *ptr, const int *cols, const void *x, void *y,
                   int nrhs, void *partials, int max_quad, int *pace, hipStream_t st);
constexpr int kSpmmPaceInts = 2 * 8 * 256 / 4;
// vector kernels with per-column scalars; partials[r * grid + wg]
int rm_vec_grid(long long total_elems, int dtype);
int launch_rm_dot(int dtype, int n, int nrhs, const void *a, const void *b, void *partials, int grid, hipStream_t st);
int launch_rm_axpy_dot(int dtype, int n, int nrhs, const void *q, void *r, const void *alpha, void *partials, int grid, hipStream_t st);
int launch_rm_aypx_x(int dtype, int n, int nrhs, const void *r, void *d, void *x, const void *alpha, const void *beta, int grid,
                     hipStream_t st);
// ---- resident loop (resident.hip): every iteration of an iterate() call of a small system inside ONE launch ---------------
struct ResidentPlan {
    int G = 0;          // work-groups (of 1024 rows) per right-hand side
    int cap = 0;        // LDS entries of the largest 1024-row matrix slice
    int wcap = 0;       // LDS entries for the per-iteration window of beta d + r (0 = every non-zero gathers from L2)
    int unroll = 8;     // gathers in flight per row walk round
    int local = 1;      // a group lives on one XCD (plain stores, sc1 loads); 0: write-through stores (groups wider than an XCD)
    int lg = 0;         // groups per ticket counter (per XCD when local)
    int slots = 0;      // group slots the sync words cover
    size_t lds_bytes = 0, sync_bytes = 0;
};
// false = the loop does not apply (size, alignment, LDS, partial-sum structure of the two-launch loop); ptr_host: n + 1 row pointers
// max_window: resident_max_window()'s answer (0 = unknown: no LDS window)
bool resident_plan(int dtype, int n, int vgrid, int row_blocks, int n_cus, const int *ptr_host, int max_window, ResidentPlan *out);
int resident_max_window(int dtype, int n, const int *ptr_dev, const int *cols_dev, int *scratch_dev, hipStream_t st, int *out);
// K iterations (number it0 + 1 ... it0 + K) of every right-hand side; state in and out is the two-launch loop's (x, r, d of
// iteration k in (k & 1 ? d1 : d0), r.r partials, delta / beta / alpha / history / iter), bit for bit.  Synchronises `st`;
// sync: rp.sync_bytes of device memory.
int run_cg_resident(int dtype, const ResidentPlan &rp, int n, int nrhs, const void *vals, const int *ptr, const int *cols, void *x, void *r,
                    void *d0, void *d1, void *part_rr, int P_rr, int row_blocks, const CgScalars &sc, int it0, int K, void *sync,
                    int n_cus, hipStream_t st, bool *untouched = nullptr, double tol = 0., int *stopped_at = nullptr);

// serialises resident launches per GPU (resident.hip); held() == false: not obtained within wait_ms -- fall back, touch nothing
struct ResidentLock {
    ResidentLock(int device, int wait_ms);
    ~ResidentLock();
    ResidentLock(const ResidentLock &) = delete;
    ResidentLock &operator=(const ResidentLock &) = delete;
    bool held() const { return held_; }
private:
    void *mutex_ = nullptr;
    int fd_ = -1;
    bool held_ = false;
};

// wide resident loop (resident.hip): chip-wide groups (one right-hand side each at a time), matrix rows in registers
constexpr int kResWideBlocksPerRpt = 2;     // 256-row blocks of a chip-wide resident member per row of a thread (512 threads: resident_device.h)
struct ResidentWidePlan {
    bool ok = false;
    int rpt = 0, unroll = 0, G = 0, NG = 1, wcap = 0;      // NG concurrent groups of G work-groups
    size_t lds_bytes = 0, sync_bytes = 0;
};
int resident_wide_plan(int dtype, int n, long long nnz, int nrhs, int n_cus, const int *ptr_dev, const int *cols_dev, int *scratch_dev,
                       hipStream_t st, ResidentWidePlan *out, bool small_ok = false);
// state in and out: x, r, d of iteration k in (k & 1 ? d1 : d0), delta / beta / alpha / history / iter; d_ready: on entry d is
// already beta d + r (three / four-launch loops); on exit d is always the direction of the last iteration (two-launch
// convention) and the launched loops' r.r partials are NOT maintained -- the caller converts / rebuilds
int run_cg_resident_wide(int dtype, const ResidentWidePlan &wp, int n, int nrhs, const void *vals, const int *ptr, const int *cols, void *x,
                         void *r, void *d0, void *d1, bool d_ready, const CgScalars &sc, int it0, int K, void *sync, int n_cus, hipStream_t st,
                         bool *untouched = nullptr, double tol = 0., int *stopped_at = nullptr);

// slab loop (slab.hip): every iteration of a call in one launch, vectors in registers, matrix streamed; systems of up to ~3M rows
struct SlabPlan {
    bool ok = false;
    int rows_m = 0, G = 0, nsteps = 0, cap = 0, ccap = 0, unroll = 8;      // rows_m / nsteps: of the largest member
    size_t lds_bytes = 0, sync_bytes = 0;
    const int *mstart = nullptr, *gmember = nullptr;      // device arrays of a non-uniform partition (slab_partition), owned by the caller
    const unsigned char *vcodes = nullptr;                // one-byte value codes of the matrix and their dictionary (build_value_codes), owned by
    const void *vdict = nullptr;                          // the caller; null = the members stream the values
};
bool slab_plan(int dtype, int n, int n_cus, const SpmvPlan &plan, bool coded, SlabPlan *out);
int slab_partition(const SlabPlan &sp, int n, const std::vector<char> &boundary, int trim, int max_members, std::vector<int> *mstart,
                   std::vector<int> *gmember);
size_t slab_sync_bytes(int G);
// row-partitioned run of the slab loop: the peer-to-peer mailboxes of the handle (device arrays as in P2pExchange) and, per peer and
// buffer parity, where this rank's boundary entries land in the PEER's published-d buffer
struct SlabComm {
    int n_halo = 0, nranks = 1, rank = 0, n_peers = 0;
    char *const *mailbox = nullptr;
    const int *peer_rank = nullptr, *send_off = nullptr, *send_count = nullptr, *recv_count = nullptr, *send_index = nullptr;
    void *const *push_dst = nullptr;                // device [2][n_peers]
    unsigned long long *halo_epoch = nullptr, *red_seq = nullptr;
};
// state in and out: x, r, din = d (already beta d + r), delta / beta / alpha / history / iter of the three / four-launch loops; ds0 / ds1:
// n + n_halo values each, where the iterations publish d; codes / dict: the one-byte column codes of the matrix (required)
int run_cg_slab(int dtype, const SlabPlan &sp, int n, long long nnz, const void *vals, const int *ptr, const unsigned char *codes,
                const int *dict, void *x, void *r, void *din, void *ds0, void *ds1, const SlabComm *cm, const CgScalars &sc, int it0, int K,
                void *sync, hipStream_t st, bool *untouched = nullptr);

// [rows][cols] -> [cols][rows]: RHS-major (the reference ABI) <-> row-major
int launch_transpose(int dtype, int rows, int cols, const void *in, void *out, hipStream_t st);

// ---- peer-to-peer backend (p2p.hip "Peer-to-peer communication over xGMI") ----------------------------
constexpr size_t kMailboxHeader = 16384;  // slots + flags + error word + single-reduction slots; halo entries follow
struct P2pExchange {
    char *const *mailbox = nullptr;   // device array [nranks] of mapped mailbox bases
    int rank = 0, n_peers = 0, n_local = 0;
    const int *peer_rank = nullptr, *send_off = nullptr, *send_count = nullptr, *dst_off = nullptr, *recv_off = nullptr,
              *recv_count = nullptr;  // device arrays [n_peers]
    const int *send_index = nullptr;
    unsigned long long *epoch = nullptr;
    unsigned *counters = nullptr;     // device, zero-initialised: [0] unpack, [1 + p] push of peer p
    int max_count = 0;                // largest send/recv count over the peers
    const void *my_halo = nullptr;    // halo area of MY mailbox (entry h = column n_local + h)
};
int p2p_push_chunks(const P2pExchange &e);
// diagonally preconditioned CG (helmFE_var.py:546-586 with a diagonal M): see vector.hip
int launch_pcg_axpy2_dot2(int dtype, bool init, int n, const void *d, void *x, const void *q, void *r, const void *m,
                          long long ld, const void *alpha, int nrhs, void *part_rz, void *part_rr, int grid, hipStream_t st);
int launch_pcg_aypx_beta(int dtype, int n, const void *r, void *p, const void *m, long long ld, const void *part_rz,
                         const void *part_rr, int P, int nrhs, const CgScalars &sc, void *rho2, void *xs, hipStream_t st);
int launch_pcg_p_update(int dtype, int n, const void *r, void *p, const void *m, long long ld, const void *beta, int nrhs, hipStream_t st);
int launch_pcg_delta0(int dtype, const void *part_rz, const void *part_rr, int P, int nrhs, const CgScalars &sc, void *rho2, hipStream_t st);
// four-launch peer-to-peer iteration (see p2p.hip): SpMV with the push and the wait inside, aypx with the beta all-reduce.
// halo_flag: device int per row block (1 = references a halo column); rotate: first row block of the visiting order
int launch_spmv_p2p(int dtype, const SpmvPlan &plan, int n, long long nnz, const void *vals, const int *ptr, const int *cols,
                    const void *d_ext, void *q, void *partials, const int *halo_flag, int rotate, const P2pExchange &e,
                    hipStream_t st);
int spmv_p2p_grid(const SpmvPlan &plan);
int launch_aypx_beta_p2p(int dtype, int n, const void *x, void *y, void *xs, const void *partials, int P, char *const *mailbox,
                         int rank, int nranks, int which, const unsigned long long *epoch, const CgScalars &sc, hipStream_t st,
                         int vec_nt = 0);
int launch_p2p_exchange(int dtype, const P2pExchange &e, void *v_ext, hipStream_t st);
// global sum (rank order) of the local sum of `partials`, followed in the same launch by the scalar step that consumes
// it: mode 1 = cg_delta0, 2 = cg_alpha, 3 = cg_beta; which in {0,1} selects the slot set
int launch_p2p_allreduce(int dtype, int mode, const void *partials, int grid, char *const *mailbox, int rank, int nranks,
                         int which, unsigned long long *epoch, const CgScalars &sc, hipStream_t st,
                         unsigned long long *bump0 = nullptr, unsigned long long *bump1 = nullptr);

// ---- single-reduction (Chronopoulos-Gear) loop of the row-partitioned solver (cg1.hip): w = A r with r.w and r.r partials
// ([2][row_blocks]); one global exchange per iteration in the prologue of the update launch
struct Cg1Update {
    int n = 0;
    void *r = nullptr, *p = nullptr, *s = nullptr, *x = nullptr;   // r: own part of the extended residual
    const void *w = nullptr;
    const void *red = nullptr;          // mailbox == null: accumulators {w.r, r.r} already summed over all ranks
    const void *partials = nullptr;     // mailbox != null: [2][P] local partials of the SpMV launch
    int P = 0;
    char *const *mailbox = nullptr;
    int rank = 0, nranks = 1;
    const unsigned long long *slot_epoch = nullptr;
    unsigned long long *halo_epoch = nullptr;
    void *state = nullptr;              // T[2][2] {gamma, alpha} by iteration parity
    CgScalars sc;
};
bool cg1_supported(const SpmvPlan &plan, const void *vals, const int *cols);
// e == nullptr: the halo is already in r_ext (RCCL backend, single rank); else push / wait inside the launch.  Bumps *iter and
// *slot_epoch (when non-null)
int launch_spmv_cg1(int dtype, const SpmvPlan &plan, int n, long long nnz, const void *vals, const int *ptr, const int *cols,
                    const void *r_ext, void *w, void *partials, const int *halo_flag, int rotate, const P2pExchange *e, int *iter,
                    unsigned long long *slot_epoch, hipStream_t st);
int launch_cg1_rowblock_rr(int dtype, int n, const void *r, void *partials, hipStream_t st);
int launch_cg1_update(int dtype, const Cg1Update &c, hipStream_t st, int vec_nt = 0);
// history[*iter] = r.r (no state of the recurrence changes); slot_epoch_rw: the slot epoch word (peer-to-peer form, advanced here)
int launch_cg1_tail(int dtype, const Cg1Update &c, unsigned long long *slot_epoch_rw, hipStream_t st);

// synthetic generators (device)
int launch_gen_laplace3d(int dtype, int nx, int ny, int nz, long long row_begin, long long row_end, void *vals,
                         int *ptr, int *cols, hipStream_t st);
long long laplace3d_ptr(long long i, int nx, int ny, int nz);
int launch_gen_poisson2d(int dtype, int N, void *vals, int *ptr, int *cols, hipStream_t st);
// generators.hip: variable = 1 helmFE_var (p0 = omega, p1 = rho; C_dev = wave speed per square or nullptr... never nullptr), 0 local_rect (p0 = k, p1 = eps, p2 = eta)
long long helm_fe_nnz(int Nh, int Nv);
int launch_gen_helm_fe(int dtype, int variable, int N, double p0, double p1, double p2, double L, const double *C_dev, int Nh, int Nv, void *vals,
                       int *ptr, int *cols, hipStream_t st);
long long poisson2d_ptr(long long i, int N);
// kind 0 rhs, 1 rhsL, 2 rhsA of helmFE_var.py:333-389 on an N x N node grid; b: N * N values of `dtype` (device)
int launch_gen_rhs(int dtype, int kind, int N, double k, void *b, hipStream_t st);

}  // namespace cgamd

// ---- context (shared by api.cpp / solver.cpp / dist.cpp) ---------------------------
struct cgamd_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    void *partials = nullptr;   // workspace for the stand-alone vdot op
    size_t partials_bytes = 0;
};
