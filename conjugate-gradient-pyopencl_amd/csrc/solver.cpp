// Persistent single-GPU CG solver: the recurrence of reference clcg.c:250-430 (== cl.py:96-200 ==
// helmFE_var.py:507-544) with every scalar kept on the device, no host synchronisation inside the
// loop, and the per-iteration kernel sequence replayed from a hipGraph.
//
// Per iteration (fused, default) -- 4 launches instead of the reference's 6 kernels + 4 blocking copies:
//   spmv+dot   q = A d, partials of d.q                     (clcg.c:299-315)
//   cg_alpha   alpha = delta / (d.q)                          (clcg.c:317-334, done on the host there)
//   axpy2_dot  x += alpha d ; r -= alpha q ; partials of r.r (clcg.c:338-374)
//   aypx_beta  beta = delta_new/delta_old ; history ; d = beta d + r   (clcg.c:376-415; beta in the prologue)
// CGAMD_UNFUSED replays the reference's own op structure (spmv, vdot, axpy, axpy, vdot, aypx).
// 16 / 32 right-hand sides (f64, complex64; f32 also 64): the block is kept ROW-MAJOR inside the handle and the
// product runs on the matrix cores (rowmajor.hip); the reference's RHS-major layout is converted in set_rhs / get_x.
#include <algorithm>
#include <cstring>
#include <vector>

#include "cgamd_internal.h"

using namespace cgamd;

struct cgamd_solver {
    Tuning tune;            // configuration this handle was created under (installed per call: TuneScope)
    cgamd_ctx *ctx = nullptr;
    int dtype = 0, n = 0, nrhs = 1, flags = 0;
    // n is the size every kernel of the handle works on; n_user the caller's.  They differ when `size` is not a whole number of
    // 16-byte packs (odd sizes in fp64 / complex64, not a multiple of 4 in fp32): the system is then carried with 1-3 EMPTY rows
    // appended (row pointers repeated, b = x0 = 0 there, so r, d, q and x stay exactly 0 in them and every sum gains exact zeros),
    // which keeps every right-hand side's vectors 16-byte aligned: the vectorised multi-RHS kernels and the resident loops apply
    // to any size.  The caller's arrays keep their own stride (strided copies in set_rhs / get_x).
    int n_user = 0;
    bool own_ptr = false;   // a borrowed device matrix whose row pointers were copied to append the padding rows
    long long nnz = 0;
    void *vals = nullptr;
    int *ptr = nullptr, *cols = nullptr;
    bool own_matrix = false;
    std::vector<int> ptr_host;   // own_matrix: the row pointers as uploaded (cgamd_solver_reload_matrix compares against them)
    SpmvPlan plan;
    int vgrid = 1;
    void *x = nullptr, *r = nullptr, *d = nullptr, *q = nullptr, *b = nullptr;
    void *slab = nullptr;   // backing store of x, r, d, q, b
    void *part_dq = nullptr, *part_rr = nullptr;
    size_t part_dq_cap = 0;      // entries per RHS
    // diagonal preconditioner (cgamd_solver_set_preconditioner): z = mdiag .* r; r.z partials; rho parity buffer
    void *mdiag = nullptr, *part_rz = nullptr, *rho2 = nullptr;
    CgScalars sc;
    bool rhs_set = false;
    int iters = 0;  // iterations enqueued since set_rhs
    // captured iteration sequences, per parity of the iteration count they start at (the two-launch loop ping-pongs d)
    hipGraphExec_t g1[2] = {nullptr, nullptr}, gU[2] = {nullptr, nullptr};
    hipGraph_t g1g[2] = {nullptr, nullptr}, gUg[2] = {nullptr, nullptr};
    int U = 8;
    bool graph_failed = false;
    // row-major multi-RHS path (rowmajor.hip): x, r, d, q, b hold [n][nrhs]; rm_ok is decided at creation, `rm` per set_rhs
    bool rm_ok = false, rm = false;
    int rm_nwg = 0, rm_vgrid = 0;
    size_t part_rr_cap = 0;
    int *rm_pace = nullptr;     // progress counters of the paced SpMM sweep (kSpmmPaceInts, zero between launches)
    // event hooks around the SpMV launch of enqueue_iteration (cgamd_solver_iterate_timed)
    hipEvent_t *ev_pair = nullptr;
    // two-launch loop (small systems): d of iteration k lives in dbuf[k & 1] (dbuf[0] = d, the initial r); decided at creation
    bool fused2 = false;
    void *d2 = nullptr;
    // resident loop (resident.hip): iterate() calls of a small system in ONE launch; decided with fused2
    bool res_ok = false;
    ResidentPlan res;
    void *res_sync = nullptr;
    int n_cus = 0;
    // wide resident loop: one chip-wide group for a single right-hand side (resident.hip)
    ResidentWidePlan resw;
    void *resw_sync = nullptr;
    unsigned char *codes = nullptr;   // one-byte column codes of the single-RHS SpMV (build_index_codes), with their dictionary
    int *dict = nullptr;
    int n_offsets = 0;                // distinct (column - row) offsets behind the codes; 0 = the SpMV reads aCols
    unsigned char *vcodes = nullptr;  // one-byte value codes on top of the one-byte column codes (build_value_codes), with their dictionary
    void *vdict = nullptr;
    int n_values = 0;                 // distinct matrix entries behind the value codes; 0 = the SpMV reads aValues
    unsigned char *jcodes = nullptr;  // one-byte joint (offset, value) codes where at most 256 pairs occur (build_joint_codes)
    int *jdict_off = nullptr;
    void *jdict_val = nullptr;
    int n_pairs = 0;
    // cgamd_solver_iterate_tol: tolerance of the device-side stop for the call in progress (0 = none), and what it reported
    double tol_req = 0.;
    bool tol_served = false, tol_stopped = false;
};

static void destroy_graphs(cgamd_solver *s) {
    for (int p = 0; p < 2; ++p) {
        if (s->g1[p]) (void)hipGraphExecDestroy(s->g1[p]);
        if (s->gU[p]) (void)hipGraphExecDestroy(s->gU[p]);
        if (s->g1g[p]) (void)hipGraphDestroy(s->g1g[p]);
        if (s->gUg[p]) (void)hipGraphDestroy(s->gUg[p]);
        s->g1[p] = s->gU[p] = nullptr;
        s->g1g[p] = s->gUg[p] = nullptr;
    }
}

static int dmalloc(void **p, size_t bytes, const char *what) {
    hipError_t e = hipMalloc(p, bytes ? bytes : 16);
    if (e != hipSuccess)
        return fail(CGAMD_ERR_ALLOC, std::string("hipMalloc(") + what + ", " + std::to_string(bytes) + " B): " + hipGetErrorString(e));
    return CGAMD_OK;
}

static int validate_csr_host(int n, long long nnz, const int *ptr, const int *cols) {
    if (ptr[0] != 0) return fail(CGAMD_ERR_INVALID, "CSR: aPointers[0] != 0");
    for (int i = 0; i < n; ++i)
        if (ptr[i + 1] < ptr[i]) return fail(CGAMD_ERR_INVALID, "CSR: aPointers not monotone at row " + std::to_string(i));
    if (ptr[n] != nnz) return fail(CGAMD_ERR_INVALID, "CSR: aPointers[size] != nonZeros");
    for (long long j = 0; j < nnz; ++j)
        if (cols[j] < 0 || cols[j] >= n) return fail(CGAMD_ERR_INVALID, "CSR: column index out of range at entry " + std::to_string(j));
    return CGAMD_OK;
}

// the SpMV (SpMM) launch of the iteration, bracketed by the caller's event pair when one is installed
static void *dbuf(cgamd_solver *s, int k) { return (s->fused2 && (k & 1)) ? s->d2 : s->d; }
static bool fused2_now(const cgamd_solver *s) { return s->fused2 && !s->rm && !s->mdiag && !(s->flags & CGAMD_UNFUSED); }

// k = iterations already enqueued since set_rhs (the iteration being enqueued is number k + 1)
static int enqueue_spmv(cgamd_solver *s, int k, hipStream_t st) {
    const int dt = s->dtype, n = s->n, nr = s->nrhs;
    // timed pass: the row-block kernel of a single right-hand side takes the event pair on its dispatch (kernel duration);
    // every other SpMV form is bracketed by hipEventRecord (duration + launch gaps)
    const bool fused2 = fused2_now(s);
    const bool ext = s->ev_pair && !fused2 && !s->rm && nr == 1 && s->plan.kind == 5;
    if (s->ev_pair && !ext) CG_HIP(hipEventRecord(s->ev_pair[0], st));
    if (ext) set_kernel_event_pair(s->ev_pair);
    int rc;
    if (fused2)
        rc = launch_spmv_fused(dt, s->plan, n, s->nnz, s->vals, s->ptr, s->cols, dbuf(s, k), dbuf(s, k + 1), s->r, s->q, nr, s->part_dq,
                               s->part_rr, s->vgrid, s->sc, st);
    else if (s->rm) rc = launch_spmm_rm(dt, n, s->nnz, s->vals, s->ptr, s->cols, s->d, s->q, nr, s->part_dq, s->plan.max_quad, s->rm_pace, st);
    else if (s->flags & CGAMD_UNFUSED) rc = launch_spmv(dt, s->plan, n, s->nnz, s->vals, s->ptr, s->cols, s->d, n, s->q, n, nr, nullptr, nullptr, st);
    else rc = launch_spmv(dt, s->plan, n, s->nnz, s->vals, s->ptr, s->cols, s->d, n, s->q, n, nr, s->d, s->part_dq, st);
    set_kernel_event_pair(nullptr);
    if (rc) return rc;
    if (s->ev_pair && !ext) CG_HIP(hipEventRecord(s->ev_pair[1], st));
    return CGAMD_OK;
}

static int enqueue_iteration(cgamd_solver *s, int k, hipStream_t st) {
    const int dt = s->dtype, n = s->n, nr = s->nrhs;
    int rc;
    if (fused2_now(s)) {   // two launches: [beta, d = beta d + r, q = A d, d.q] and [alpha, x += alpha d, r -= alpha q, r.r]
        if ((rc = enqueue_spmv(s, k, st))) return rc;
        return launch_axpy2_dot_alpha(dt, n, dbuf(s, k + 1), s->x, s->q, s->r, n, s->part_dq, s->plan.n_partials, s->sc, nr, s->part_rr, s->vgrid, st);
    }
    if (s->rm) {      // row-major block: SpMM on the matrix cores (+ d.q partials), alpha, r update (+ r.r), beta, x and d updates
        if ((rc = enqueue_spmv(s, k, st))) return rc;
        if ((rc = launch_cg_alpha(dt, s->part_dq, s->rm_nwg, nr, s->sc, st))) return rc;
        if ((rc = launch_rm_axpy_dot(dt, n, nr, s->q, s->r, s->sc.alpha, s->part_rr, s->rm_vgrid, st))) return rc;
        if ((rc = launch_cg_beta(dt, s->part_rr, s->rm_vgrid, nr, s->sc, st))) return rc;
        return launch_rm_aypx_x(dt, n, nr, s->r, s->d, s->x, s->sc.alpha, s->sc.beta, s->rm_vgrid, st);
    }
    if (s->mdiag) {   // preconditioned recurrence (helmFE_var.py:560-585); delta holds rho = r.z
        if ((rc = enqueue_spmv(s, k, st))) return rc;
        if ((rc = launch_cg_alpha(dt, s->part_dq, s->plan.n_partials, nr, s->sc, st))) return rc;
        if ((rc = launch_pcg_axpy2_dot2(dt, false, n, s->d, s->x, s->q, s->r, s->mdiag, n, s->sc.alpha, nr, s->part_rz, s->part_rr,
                                        s->vgrid, st))) return rc;
        return launch_pcg_aypx_beta(dt, n, s->r, s->d, s->mdiag, n, s->part_rz, s->part_rr, s->vgrid, nr, s->sc, s->rho2, s->x, st);
    }
    if (!(s->flags & CGAMD_UNFUSED)) {
        if ((rc = enqueue_spmv(s, k, st))) return rc;
        const bool fold = fold_alpha_ok(s->plan.n_partials, s->plan.fold_max);      // small system: alpha in the next launch's prologue
        if (!fold && (rc = launch_cg_alpha(dt, s->part_dq, s->plan.n_partials, nr, s->sc, st))) return rc;
        // r -= alpha q (+ r.r) ; then beta, x += alpha d, d = beta d + r : 3 + 5 vector passes (x is read by nothing inside the loop, so
        // its update rides in the aypx launch, which reads d anyway)
        if (fold) rc = launch_axpy_dot_alpha(dt, n, s->q, s->r, n, s->part_dq, s->plan.n_partials, s->sc, nr, s->part_rr, s->vgrid, st);
        else rc = launch_axpy_dot(dt, n, s->q, s->r, n, s->sc.alpha, nr, s->part_rr, s->vgrid, st, s->plan.vec_nt);
        if (rc) return rc;
        return launch_aypx_beta_x(dt, n, s->r, s->d, s->x, n, s->part_rr, s->vgrid, nr, s->sc, st, s->plan.vec_nt);
    }
    if ((rc = enqueue_spmv(s, k, st))) return rc;
    if ((rc = launch_dot_partials(dt, n, s->d, s->q, n, nr, s->part_rr, s->vgrid, st))) return rc;
    if ((rc = launch_cg_alpha(dt, s->part_rr, s->vgrid, nr, s->sc, st))) return rc;
    if ((rc = launch_axpy(dt, n, s->d, s->x, n, s->sc.alpha, 1, nr, st))) return rc;
    if ((rc = launch_axpy(dt, n, s->q, s->r, n, s->sc.alpha, 0, nr, st))) return rc;
    if ((rc = launch_dot_partials(dt, n, s->r, s->r, n, nr, s->part_rr, s->vgrid, st))) return rc;
    if ((rc = launch_cg_beta(dt, s->part_rr, s->vgrid, nr, s->sc, st))) return rc;
    return launch_aypx(dt, n, s->r, s->d, n, s->sc.beta, nr, st);
}

// the resident loop applies where the two-launch loop does and the matrix slices fit LDS (needs the row pointers on the host)
// The launched loops of a handle the chip-wide resident loop can take over produce ITS bits (and the other way round): one
// 16-byte pack per thread in the vector launches (the r.r partial of a work-group = 256 consecutive packs), alpha folded whatever
// the size, and every prologue sum in the member-blocked order (reduce_device.h thread_partials; K = the blocks of one member).
// So iterate(15) twice and iterate(30) return the same bits although the first takes the launched loop and the second the
// resident one.  Called whenever the wide plan may have changed; captured graphs hold grids and orders, so they go when it did.
static void apply_wide_order(cgamd_solver *s) {
    const int E = (int)(16 / dtype_size(s->dtype));
    // (where the one-XCD resident loop applies it runs, with the strided order -- unless a preconditioner is set: that recurrence only
    // has the chip-wide form)
    const bool wide = s->resw.ok && (!s->res_ok || s->mdiag != nullptr);
    const int kdq = wide ? kResWideBlocksPerRpt * s->resw.rpt : 0, krr = wide ? kdq / E : 0;
    const int vgrid = wide ? (s->n / E + kBlock - 1) / kBlock : vec_grid(s->n, s->dtype, s->nrhs);
    const int fold_max = 0;      // (alpha folded beyond 2048 partials was tried for these handles: every work-group summing 3907 partials, 1M rows 30 -> 52 us)
    if (kdq == s->sc.kdq && krr == s->sc.krr && vgrid == s->vgrid && fold_max == s->plan.fold_max) return;
    destroy_graphs(s);
    s->sc.kdq = kdq; s->sc.krr = krr; s->vgrid = vgrid; s->plan.fold_max = fold_max;
    s->fused2 = fused2_ok(s->plan, s->dtype, s->nrhs, s->vals, s->cols);
}

// the resident loop applies where the two-launch loop does and the matrix slices fit LDS (needs the row pointers on the host)
static int setup_resident_wide_plan(cgamd_solver *s) {
    s->resw.ok = false;
    if (tune().resident_wide == 0 || tune().resident == 0 || (s->flags & CGAMD_UNFUSED)) return CGAMD_OK;
    if (s->rm_ok && tune().spmm_rowmajor >= 2) return CGAMD_OK;      // the row-major loop was asked for
    if (!aligned16(s->x) || !aligned16(s->r) || !aligned16(s->d) || !aligned16(s->d2)) return CGAMD_OK;
    if (s->plan.kind != 5 && s->plan.kind != 6) return CGAMD_OK;    // the launched loops' 256-row d.q partials are what the members reproduce
    if (s->n % (int)(16 / dtype_size(s->dtype)) != 0) return CGAMD_OK;
    if (!s->n_cus) CG_HIP(hipDeviceGetAttribute(&s->n_cus, hipDeviceAttributeMultiprocessorCount, s->ctx->device));
    ResidentWidePlan wp;
    // (systems of up to 32768 rows only where the two-launch loop is the handle's launched loop and the one-XCD loop cannot hold them:
    // complex128 with 7-entry rows, 1024 rows x 20 bytes per entry do not fit LDS)
    if (int rc = resident_wide_plan(s->dtype, s->n, s->nnz, s->nrhs, s->n_cus, s->ptr, s->cols, s->sc.iter, s->ctx->stream, &wp, s->fused2)) return rc;
    if (!wp.ok) return CGAMD_OK;
    if ((size_t)((s->n / (int)(16 / dtype_size(s->dtype)) + kBlock - 1) / kBlock) > s->part_rr_cap) return CGAMD_OK;     // (cannot happen: sized at creation)
    if (s->resw_sync && wp.sync_bytes > s->resw.sync_bytes) { (void)hipFree(s->resw_sync); s->resw_sync = nullptr; }
    if (!s->resw_sync)
        if (int rc = dmalloc(&s->resw_sync, wp.sync_bytes, "wide resident sync words")) return rc;
    s->resw = wp;
    s->rm_ok = false;       // the chip-wide resident groups keep the caller's RHS-major layout (and beat the row-major loop: 1M x 32 fp64)
    return CGAMD_OK;
}

// (re)build the one-byte column codes for the matrix now in s->cols; dropped when they do not apply
static int setup_index_codes(cgamd_solver *s) {
    if (s->codes) { (void)hipFree(s->codes); s->codes = nullptr; }
    if (s->dict) { (void)hipFree(s->dict); s->dict = nullptr; }
    s->plan.codes = nullptr; s->plan.dict = nullptr; s->plan.codes_for = nullptr; s->plan.codes16 = false;
    s->n_offsets = 0;
    if (s->vcodes) { (void)hipFree(s->vcodes); s->vcodes = nullptr; }
    if (s->vdict) { (void)hipFree(s->vdict); s->vdict = nullptr; }
    s->plan.vcodes = nullptr; s->plan.vdict = nullptr; s->plan.vcodes_for = nullptr;
    s->n_values = 0;
    if (s->jcodes) { (void)hipFree(s->jcodes); s->jcodes = nullptr; }
    if (s->jdict_off) { (void)hipFree(s->jdict_off); s->jdict_off = nullptr; }
    if (s->jdict_val) { (void)hipFree(s->jdict_val); s->jdict_val = nullptr; }
    s->plan.jcodes = nullptr; s->plan.jdict_off = nullptr; s->plan.jdict_val = nullptr;
    s->n_pairs = 0;
    const size_t matrix_bytes = (size_t)s->nnz * (dtype_size(s->dtype) + 4);
    // a handle whose iterations run in the chip-wide resident loop (matrix in registers) would pay the two coding passes at every
    // create / reload (the stateless cg() reloads per call) for the few launched SpMVs around it
    if (s->resw.ok && !(s->flags & CGAMD_NO_GRAPH) && s->tune.index_codes < 2) return CGAMD_OK;
    if (!s->tune.index_codes || s->nrhs != 1 || (s->plan.kind != 5 && s->plan.kind != 7) || s->tune.index_codes_min_mb < 0 ||
        matrix_bytes <= ((size_t)s->tune.index_codes_min_mb << 20))
        return CGAMD_OK;
    if (int rc = build_index_codes(s->n, s->nnz, s->ptr, s->cols, s->ctx->stream, &s->codes, &s->dict, &s->n_offsets)) return rc;
    if (s->codes) {
        s->plan.codes = s->codes; s->plan.dict = s->dict; s->plan.codes_for = s->cols; s->plan.codes16 = false;
        // matrices of at most 256 distinct entries (constant-coefficient stencils): one-byte value codes as well, 2 bytes per non-zero
        if (s->tune.value_codes && s->plan.kind == 5) {
            if (int rc = build_value_codes(s->dtype, s->nnz, s->vals, s->ctx->stream, &s->vcodes, &s->vdict, &s->n_values)) return rc;
            if (s->vcodes) { s->plan.vcodes = s->vcodes; s->plan.vdict = s->vdict; s->plan.vcodes_for = s->vals; }
            if (s->vcodes && s->tune.dev_joint_codes) {
                if (int rc = build_joint_codes(s->dtype, s->nnz, s->codes, s->vcodes, s->dict, s->vdict, s->ctx->stream, &s->jcodes, &s->jdict_off,
                                               &s->jdict_val, &s->n_pairs)) return rc;
                if (s->jcodes) { s->plan.jcodes = s->jcodes; s->plan.jdict_off = s->jdict_off; s->plan.jdict_val = s->jdict_val; }
            }
        }
        return CGAMD_OK;
    }
    // more than 256 distinct offsets (unstructured patterns, Matrix-Market inputs): 16-bit columns relative to the row block's first
    if (s->tune.index_codes16 == 0) return CGAMD_OK;
    if (int rc = build_index_codes16(s->n, s->nnz, s->ptr, s->cols, s->ctx->stream, &s->codes, &s->dict)) return rc;
    if (s->codes) { s->plan.codes = s->codes; s->plan.dict = s->dict; s->plan.codes_for = s->cols; s->plan.codes16 = true; s->n_offsets = 65536; }
    return CGAMD_OK;
}

static int setup_resident_local(cgamd_solver *s);
static int setup_resident(cgamd_solver *s) {
    const int rc = setup_resident_local(s);
    apply_wide_order(s);
    return rc;
}
// the one-XCD loop where it applies (bit-identical to the two-launch loop as it is); else the chip-wide groups
static int setup_resident_one_xcd(cgamd_solver *s);
static int setup_resident_local(cgamd_solver *s) {
    s->res_ok = false;
    s->resw.ok = false;
    if (int rc = setup_resident_one_xcd(s)) return rc;
    // where the one-XCD loop applies the chip-wide plan is only needed once a preconditioner is set (that recurrence has no one-XCD
    // form): cgamd_solver_set_preconditioner asks for it then -- the stateless cg() entry reloads per call and would pay the scan
    if (s->res_ok && !s->mdiag) return CGAMD_OK;
    return setup_resident_wide_plan(s);
}
static int setup_resident_one_xcd(cgamd_solver *s) {
    if (!s->fused2 || tune().resident == 0 || s->n > 65536) return CGAMD_OK;
    std::vector<int> tmp;
    const int *ph = s->ptr_host.size() == (size_t)s->n + 1 ? s->ptr_host.data() : nullptr;
    if (!ph) {          // borrowed device matrix
        tmp.resize((size_t)s->n + 1);
        CG_HIP(hipMemcpyAsync(tmp.data(), s->ptr, tmp.size() * 4, hipMemcpyDeviceToHost, s->ctx->stream));
        CG_HIP(hipStreamSynchronize(s->ctx->stream));
        ph = tmp.data();
    }
    if (!s->n_cus) CG_HIP(hipDeviceGetAttribute(&s->n_cus, hipDeviceAttributeMultiprocessorCount, s->ctx->device));
    ResidentPlan rp;
    if (!aligned16(s->x) || !aligned16(s->r) || !aligned16(s->d) || !aligned16(s->d2)) return CGAMD_OK;
    int max_window = 0;
    if (s->n % (int)(16 / dtype_size(s->dtype)) == 0)
        if (int rc = resident_max_window(s->dtype, s->n, s->ptr, s->cols, s->sc.iter, s->ctx->stream, &max_window)) return rc;
    // (checked against the plain launched configuration: the vector grid of a handle no chip-wide loop takes over)
    if (!resident_plan(s->dtype, s->n, vec_grid(s->n, s->dtype, s->nrhs), s->plan.n_partials, s->n_cus, ph, max_window, &rp)) return CGAMD_OK;
    if (s->res_sync && rp.sync_bytes > s->res.sync_bytes) { (void)hipFree(s->res_sync); s->res_sync = nullptr; }
    if (!s->res_sync)
        if (int rc = dmalloc(&s->res_sync, rp.sync_bytes, "resident sync words")) return rc;
    s->res = rp;
    s->res_ok = true;
    return CGAMD_OK;
}

static int capture(cgamd_solver *s, int k0, int iters, hipGraph_t *g, hipGraphExec_t *ge) {
    hipStream_t st = s->ctx->stream;
    hipError_t e = hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed);
    if (e != hipSuccess) return fail(CGAMD_ERR_HIP, std::string("hipStreamBeginCapture: ") + hipGetErrorString(e));
    int rc = CGAMD_OK;
    for (int i = 0; i < iters && rc == CGAMD_OK; ++i) rc = enqueue_iteration(s, k0 + i, st);
    e = hipStreamEndCapture(st, g);
    if (rc != CGAMD_OK) return rc;
    if (e != hipSuccess) return fail(CGAMD_ERR_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
    e = hipGraphInstantiate(ge, *g, nullptr, nullptr, 0);
    if (e != hipSuccess) return fail(CGAMD_ERR_HIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(e));
    return CGAMD_OK;
}

static int ensure_history(cgamd_solver *s, int entries) {
    if (entries <= s->sc.history_cap) return CGAMD_OK;
    int cap = std::max(entries, std::max(1024, s->sc.history_cap * 2));
    const size_t vs = dtype_size(s->dtype);
    void *nh = nullptr;
    if (int rc = dmalloc(&nh, (size_t)cap * s->nrhs * vs, "history")) return rc;
    if (s->sc.history) {
        CG_HIP(hipStreamSynchronize(s->ctx->stream));
        // on the context's stream, never the legacy NULL stream: work on the NULL stream while ANOTHER thread's stream is
        // being captured is an error in that thread too ("implicit dependency on the legacy stream")
        CG_HIP(hipMemcpyAsync(nh, s->sc.history, (size_t)s->sc.history_cap * s->nrhs * vs, hipMemcpyDeviceToDevice, s->ctx->stream));
        CG_HIP(hipStreamSynchronize(s->ctx->stream));
        CG_HIP(hipFree(s->sc.history));
    }
    s->sc.history = nh;
    s->sc.history_cap = cap;
    destroy_graphs(s);  // history pointer / capacity are baked into captured kernel arguments
    return CGAMD_OK;
}

extern "C" {

int cgamd_solver_create(cgamd_ctx *ctx, int dtype, int size, long long nnz, const void *aValues, const int *aPointers,
                        const int *aCols, int nRHS, int flags, cgamd_solver **out) {
    if (!out) return fail(CGAMD_ERR_INVALID, "solver_create: out is NULL");
    *out = nullptr;
    if (!ctx) return fail(CGAMD_ERR_INVALID, "solver_create: ctx is NULL");
    if (dtype < 0 || dtype > 3) return fail(CGAMD_ERR_INVALID, "solver_create: bad dtype");
    if (size < 1 || nnz < 0 || nRHS < 1) return fail(CGAMD_ERR_INVALID, "solver_create: bad size/nnz/nRHS");
    if (nnz > 2147483647LL - 8192) return fail(CGAMD_ERR_INVALID, "solver_create: nnz exceeds int32 row pointers");
    if (!aPointers || (nnz > 0 && (!aValues || !aCols))) return fail(CGAMD_ERR_INVALID, "solver_create: null matrix pointer");
    CG_HIP(hipSetDevice(ctx->device));
    const size_t vs = dtype_size(dtype);
    cgamd_solver *s = new cgamd_solver();
    s->tune = tune_snapshot();
    TuneScope ts(&s->tune);
    s->ctx = ctx; s->dtype = dtype; s->n = size; s->n_user = size; s->nnz = nnz; s->nrhs = nRHS; s->flags = flags;
    // Row-major block + matrix-core SpMM inside the loop: by default only where the whole iteration is faster than the RHS-major
    // one (measured in one process at N = 1M, profiles/r2_experiments/spmm_ab12.log: f64 x 32 +8 %; f64 x 16, f32 x 32 equal within
    // 1 %, complex64 x 16 slower).  spmm_rowmajor = 2 takes it for every supported type, 0 never.
    const int rm_knob = tune().spmm_rowmajor;
    // (up to 32768 rows the resident loop is several times faster than any launched loop; between that and ~1M rows the chip-wide
    // resident groups take over when they apply, setup_resident_wide_plan)
    const bool rm_wins = dtype == CGAMD_F64 && nRHS == 32 && size > 32768;
    s->rm_ok = nRHS > 1 && (rm_knob >= 2 || (rm_knob == 1 && rm_wins)) && !(flags & CGAMD_UNFUSED) && spmm_rm_supported(dtype, nRHS, size);
    if (s->rm_ok) s->rm_vgrid = rm_vec_grid((long long)size * nRHS, dtype);
    {
        const int E = (int)(16 / vs);      // values per 16-byte pack
        if (size % E != 0 && !s->rm_ok && tune().pad_rows != 0 && size < 2147483647 - 8) s->n = (size + E - 1) / E * E;
    }
    const int n_int = s->n;
    s->plan = make_spmv_plan(n_int);
    s->vgrid = vec_grid(n_int, dtype, nRHS);
    int rc = CGAMD_OK;
    std::vector<int> pad_ptr((size_t)(n_int - size), (int)nnz);      // row pointers of the appended empty rows
    if (flags & CGAMD_MATRIX_ON_DEVICE) {
        s->vals = const_cast<void *>(aValues);
        s->ptr = const_cast<int *>(aPointers);
        s->cols = const_cast<int *>(aCols);
        if (n_int != size) {
            s->ptr = nullptr;
            rc = dmalloc((void **)&s->ptr, (size_t)(n_int + 1) * 4, "aPointers (padded copy)");
            if (!rc) {
                s->own_ptr = true;
                hipError_t e = hipMemcpyAsync(s->ptr, aPointers, (size_t)(size + 1) * 4, hipMemcpyDeviceToDevice, ctx->stream);
                if (e == hipSuccess) e = hipMemcpyAsync(s->ptr + size + 1, pad_ptr.data(), pad_ptr.size() * 4, hipMemcpyHostToDevice, ctx->stream);
                if (e != hipSuccess) rc = fail(CGAMD_ERR_HIP, std::string("copy aPointers: ") + hipGetErrorString(e));
            }
        }
    } else {
        rc = validate_csr_host(size, nnz, aPointers, aCols);
        s->own_matrix = true;
        if (!rc) rc = dmalloc(&s->vals, (size_t)nnz * vs + 64, "aValues");
        if (!rc) rc = dmalloc((void **)&s->ptr, (size_t)(n_int + 1) * 4, "aPointers");
        if (!rc) rc = dmalloc((void **)&s->cols, (size_t)nnz * 4 + 64, "aCols");
        if (!rc && nnz) {
            hipError_t e = hipMemcpyAsync(s->vals, aValues, (size_t)nnz * vs, hipMemcpyHostToDevice, ctx->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(s->cols, aCols, (size_t)nnz * 4, hipMemcpyHostToDevice, ctx->stream);
            if (e != hipSuccess) rc = fail(CGAMD_ERR_HIP, std::string("upload matrix: ") + hipGetErrorString(e));
        }
        if (!rc) {
            hipError_t e = hipMemcpyAsync(s->ptr, aPointers, (size_t)(size + 1) * 4, hipMemcpyHostToDevice, ctx->stream);
            if (e == hipSuccess && n_int != size)
                e = hipMemcpyAsync(s->ptr + size + 1, pad_ptr.data(), pad_ptr.size() * 4, hipMemcpyHostToDevice, ctx->stream);
            if (e != hipSuccess) rc = fail(CGAMD_ERR_HIP, std::string("upload aPointers: ") + hipGetErrorString(e));
            else {
                s->ptr_host.assign(aPointers, aPointers + size + 1);
                s->ptr_host.insert(s->ptr_host.end(), pad_ptr.begin(), pad_ptr.end());
            }
        }
    }
    const size_t vbytes = (size_t)n_int * nRHS * vs;
    {
        // the five vectors live in one slab, each at a 4 KiB-aligned offset plus a per-vector skew: the update kernels
        // stream up to four of them in lock-step, and equal strides between them alias onto the same HBM channels
        const size_t pitch = (vbytes + 4095) & ~(size_t)4095;
        if (!rc) rc = dmalloc(&s->slab, pitch * 6 + 4096, "vectors");
        if (!rc && n_int != size && hipMemsetAsync(s->slab, 0, pitch * 6 + 4096, ctx->stream) != hipSuccess)     // the padding rows of b
            rc = fail(CGAMD_ERR_HIP, "hipMemsetAsync(vectors)");
        if (!rc) {
            char *base = static_cast<char *>(s->slab);
            s->x = base; s->r = base + pitch; s->d = base + 2 * pitch; s->q = base + 3 * pitch; s->b = base + 4 * pitch;
            s->d2 = base + 5 * pitch;
        }
    }
    // the row-major SpMM writes one d.q partial per work-group of its sweep: at most 8 XCDs x 32 CUs x 8 work-groups
    s->part_dq_cap = (size_t)std::max(std::max(s->plan.grid, s->plan.row_blocks), s->rm_ok ? 2048 : 0);
    if (!rc) rc = dmalloc(&s->part_dq, acc_size(dtype) * s->part_dq_cap * nRHS, "partials_dq");
    // (handles the chip-wide resident loop may take over size their vector launches one pack per thread: apply_wide_order)
    s->part_rr_cap = (size_t)std::max(std::max(s->vgrid, s->rm_vgrid), n_int <= (1 << 20) + 4096 ? (int)((n_int / (16 / vs) + kBlock - 1) / kBlock) : 0);
    if (!rc) rc = dmalloc(&s->part_rr, acc_size(dtype) * s->part_rr_cap * nRHS, "partials_rr");
    if (!rc) rc = dmalloc(&s->sc.alpha, vs * nRHS, "alpha");
    if (!rc) rc = dmalloc(&s->sc.beta, vs * nRHS, "beta");
    if (!rc) rc = dmalloc(&s->sc.delta, vs * nRHS, "delta");
    if (!rc) rc = dmalloc((void **)&s->sc.iter, 64, "iter");
    if (!rc) rc = dmalloc(&s->sc.stage, acc_size(dtype) * 32 * (size_t)nRHS, "alpha stage");
    if (!rc) rc = dmalloc((void **)&s->sc.ticket, sizeof(unsigned) * (size_t)nRHS, "alpha tickets");
    if (!rc && hipMemsetAsync(s->sc.ticket, 0, sizeof(unsigned) * (size_t)nRHS, ctx->stream) != hipSuccess) rc = fail(CGAMD_ERR_HIP, "hipMemsetAsync(alpha tickets)");
    if (!rc) rc = ensure_history(s, 1024);
    if (!rc && (flags & CGAMD_MATRIX_ON_DEVICE))      // host matrices were checked before the upload
        rc = validate_csr_device(size, nnz, size, aPointers, s->cols, nullptr, 0, 0, s->sc.iter, ctx->stream);     // the caller's arrays, as passed
    if (!rc) rc = compute_spmv_plan(s->ptr, s->cols, n_int, s->sc.iter, ctx->stream, &s->plan);
    if (!rc) finalize_spmv_plan(&s->plan, dtype, nRHS, n_int, nnz, s->vals, s->cols);
    if (!rc && s->rm_ok) s->rm_nwg = spmm_rm_grid(dtype, nRHS, size, s->plan.max_quad, true);
    if (!rc && s->rm_ok) rc = dmalloc((void **)&s->rm_pace, sizeof(int) * kSpmmPaceInts, "spmm pace counters");
    if (!rc && s->rm_ok && hipMemsetAsync(s->rm_pace, 0, sizeof(int) * kSpmmPaceInts, ctx->stream) != hipSuccess) rc = fail(CGAMD_ERR_HIP, "hipMemsetAsync(spmm pace counters)");
    if (!rc) s->fused2 = fused2_ok(s->plan, dtype, nRHS, s->vals, s->cols);
    if (!rc) rc = setup_resident(s);
    if (!rc) rc = setup_index_codes(s);
    if (!rc) {
        hipError_t e = hipStreamSynchronize(ctx->stream);  // host matrix arrays may go away after return
        if (e != hipSuccess) rc = fail(CGAMD_ERR_HIP, std::string("solver_create sync: ") + hipGetErrorString(e));
    }
    if (rc) {
        std::string keep = cgamd_last_error();
        cgamd_solver_destroy(s);
        set_error(keep);
        return rc;
    }
    *out = s;
    return CGAMD_OK;
}

// New matrix VALUES / PATTERN of the same size into an existing handle (host arrays; the handle must own its matrix):
// what the stateless cg() needs to reuse its cached device state -- allocations, stream, captured graphs -- from one call
// to the next.  The pattern-dependent plan is recomputed only when the row pointers differ from the ones uploaded before.
int cgamd_solver_reload_matrix(cgamd_solver *s, const void *aValues, const int *aPointers, const int *aCols) {
    if (!s || !aPointers || (s->nnz > 0 && (!aValues || !aCols))) return fail(CGAMD_ERR_INVALID, "reload_matrix: null argument");
    if (!s->own_matrix) return fail(CGAMD_ERR_STATE, "reload_matrix: the handle borrows a device matrix");
    TuneScope ts(&s->tune);
    CG_HIP(hipSetDevice(s->ctx->device));
    if (int rc = validate_csr_host(s->n_user, s->nnz, aPointers, aCols)) return rc;
    hipStream_t st = s->ctx->stream;
    const size_t vs = dtype_size(s->dtype);
    s->rhs_set = false;
    if (s->nnz) {
        CG_HIP(hipMemcpyAsync(s->vals, aValues, (size_t)s->nnz * vs, hipMemcpyHostToDevice, st));
        CG_HIP(hipMemcpyAsync(s->cols, aCols, (size_t)s->nnz * 4, hipMemcpyHostToDevice, st));
    }
    const bool same_ptr = s->ptr_host.size() == (size_t)s->n + 1 && memcmp(s->ptr_host.data(), aPointers, ((size_t)s->n_user + 1) * 4) == 0;
    if (!same_ptr) {
        // (the row pointers of the appended empty rows all equal nnz, which does not change: they stay as uploaded at creation)
        CG_HIP(hipMemcpyAsync(s->ptr, aPointers, ((size_t)s->n_user + 1) * 4, hipMemcpyHostToDevice, st));
        std::copy(aPointers, aPointers + s->n_user + 1, s->ptr_host.begin());
        destroy_graphs(s);          // kernel choice, LDS size and partial counts are baked into the captured launches
        s->plan = make_spmv_plan(s->n);
        if (int rc = compute_spmv_plan(s->ptr, s->cols, s->n, s->sc.iter, st, &s->plan)) return rc;
        finalize_spmv_plan(&s->plan, s->dtype, s->nrhs, s->n, s->nnz, s->vals, s->cols);
        if ((size_t)std::max(s->plan.grid, s->plan.row_blocks) > s->part_dq_cap) return fail(CGAMD_ERR_STATE, "reload_matrix: partial buffer too small");
        if (s->rm_ok) s->rm_nwg = spmm_rm_grid(s->dtype, s->nrhs, s->n, s->plan.max_quad, true);
        s->fused2 = fused2_ok(s->plan, s->dtype, s->nrhs, s->vals, s->cols);
    }
    if (int rc = setup_resident(s)) return rc;      // also with unchanged row pointers: the column range of a row slice may have moved
    {                                               // the columns were replaced: their codes go with them
        const bool had = s->codes != nullptr;
        if (had) destroy_graphs(s);                 // captured launches hold the old code array
        if (int rc = setup_index_codes(s)) return rc;
        if (!had && s->codes) destroy_graphs(s);
    }
    CG_HIP(hipStreamSynchronize(st));   // the host arrays may go away after return
    return CGAMD_OK;
}

int cgamd_solver_destroy(cgamd_solver *s) {
    if (!s) return CGAMD_OK;
    thread_hip_setup();
    (void)hipSetDevice(s->ctx->device);
    (void)hipStreamSynchronize(s->ctx->stream);
    destroy_graphs(s);
    if (s->own_ptr && s->ptr) (void)hipFree(s->ptr);
    if (s->own_matrix) {
        if (s->vals) (void)hipFree(s->vals);
        if (s->ptr) (void)hipFree(s->ptr);
        if (s->cols) (void)hipFree(s->cols);
    }
    void *bufs[] = {s->slab, s->part_dq, s->part_rr, s->sc.alpha, s->sc.beta, s->sc.delta,
                    s->sc.history, s->sc.iter, s->mdiag, s->part_rz, s->rho2, s->sc.stage, s->sc.ticket, s->res_sync, s->resw_sync, s->codes, s->dict, s->rm_pace, s->vcodes, s->vdict, s->jcodes, s->jdict_off, s->jdict_val};
    for (void *p : bufs)
        if (p) (void)hipFree(p);
    delete s;
    return CGAMD_OK;
}

int cgamd_solver_set_rhs(cgamd_solver *s, const void *b, const void *x0, int on_device) {
    if (!s || !b) return fail(CGAMD_ERR_INVALID, "set_rhs: null argument");
    TuneScope ts(&s->tune);
    CG_HIP(hipSetDevice(s->ctx->device));
    hipStream_t st = s->ctx->stream;
    const size_t vbytes = (size_t)s->n * s->nrhs * dtype_size(s->dtype);
    const hipMemcpyKind kind = on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    const bool rm = s->rm_ok && !s->mdiag;      // the preconditioned recurrence keeps the RHS-major kernels
    if (rm != s->rm) destroy_graphs(s);         // captured launch sequences belong to one layout
    s->rm = rm;
    int rc;
    if (rm) {
        // caller's blocks are RHS-major [nrhs][n] (reference spmv.cl:25,48); the handle keeps [n][nrhs]: q is the staging area
        CG_HIP(hipMemcpyAsync(s->q, b, vbytes, kind, st));
        if ((rc = launch_transpose(s->dtype, s->nrhs, s->n, s->q, s->b, st))) return rc;
        if (x0) {
            CG_HIP(hipMemcpyAsync(s->q, x0, vbytes, kind, st));
            if ((rc = launch_transpose(s->dtype, s->nrhs, s->n, s->q, s->x, st))) return rc;
        } else {
            CG_HIP(hipMemsetAsync(s->x, 0, vbytes, st));
        }
        // r = b - A x0 ; d = r ; delta0 = r.r   (clcg.c:255-292)
        if ((rc = launch_spmm_rm(s->dtype, s->n, s->nnz, s->vals, s->ptr, s->cols, s->x, s->q, s->nrhs, nullptr, s->plan.max_quad, s->rm_pace, st))) return rc;
        if ((rc = launch_sub(s->dtype, s->n * s->nrhs, s->b, s->q, s->r, (long long)s->n * s->nrhs, 1, st))) return rc;
        CG_HIP(hipMemcpyAsync(s->d, s->r, vbytes, hipMemcpyDeviceToDevice, st));
        if ((rc = launch_rm_dot(s->dtype, s->n, s->nrhs, s->r, s->r, s->part_rr, s->rm_vgrid, st))) return rc;
        if ((rc = launch_cg_delta0(s->dtype, s->part_rr, s->rm_vgrid, s->nrhs, s->sc, st))) return rc;
        if (!on_device) CG_HIP(hipStreamSynchronize(st));
        s->rhs_set = true;
        s->iters = 0;
        return CGAMD_OK;
    }
    if (s->n != s->n_user) {     // caller's blocks are [nrhs][size]; the handle's carry the padding rows (b and x stay 0 there)
        const size_t vs = dtype_size(s->dtype);
        CG_HIP(hipMemcpy2DAsync(s->b, (size_t)s->n * vs, b, (size_t)s->n_user * vs, (size_t)s->n_user * vs, (size_t)s->nrhs, kind, st));
        CG_HIP(hipMemsetAsync(s->x, 0, vbytes, st));
        if (x0) CG_HIP(hipMemcpy2DAsync(s->x, (size_t)s->n * vs, x0, (size_t)s->n_user * vs, (size_t)s->n_user * vs, (size_t)s->nrhs, kind, st));
    } else {
        CG_HIP(hipMemcpyAsync(s->b, b, vbytes, kind, st));
        if (x0) CG_HIP(hipMemcpyAsync(s->x, x0, vbytes, kind, st));
        else CG_HIP(hipMemsetAsync(s->x, 0, vbytes, st));
    }
    // r = b - A x0 ; d = r ; delta0 = r.r   (clcg.c:255-292)
    if ((rc = launch_spmv(s->dtype, s->plan, s->n, s->nnz, s->vals, s->ptr, s->cols, s->x, s->n, s->q, s->n, s->nrhs,
                          nullptr, nullptr, st))) return rc;
    if ((rc = launch_sub(s->dtype, s->n, s->b, s->q, s->r, s->n, s->nrhs, st))) return rc;
    if (s->mdiag) {   // z0 = M r0, p0 = z0, rho0 = r0.z0 (helmFE_var.py:562-573)
        if ((rc = launch_pcg_axpy2_dot2(s->dtype, true, s->n, s->d, s->x, s->q, s->r, s->mdiag, s->n, nullptr, s->nrhs, s->part_rz,
                                        s->part_rr, s->vgrid, st))) return rc;
        if ((rc = launch_pcg_delta0(s->dtype, s->part_rz, s->part_rr, s->vgrid, s->nrhs, s->sc, s->rho2, st))) return rc;
    } else {
        CG_HIP(hipMemcpyAsync(s->d, s->r, vbytes, hipMemcpyDeviceToDevice, st));
        if ((rc = launch_dot_partials(s->dtype, s->n, s->r, s->r, s->n, s->nrhs, s->part_rr, s->vgrid, st))) return rc;
        if ((rc = launch_cg_delta0(s->dtype, s->part_rr, s->vgrid, s->nrhs, s->sc, st))) return rc;
    }
    if (!on_device) CG_HIP(hipStreamSynchronize(st));  // host buffers may be released by the caller
    s->rhs_set = true;
    s->iters = 0;
    return CGAMD_OK;
}

// z = m .* r between residual and search direction: the reference's PCG with a diagonal CSR M (helmFE_var.py:546-586;
// pass 1/diag(A) for Jacobi).  m: `size` values of the solver's type, shared by all right-hand sides; NULL removes the
// preconditioner.  The next cgamd_solver_set_rhs starts the preconditioned recurrence; history then holds r.r as before.
int cgamd_solver_set_preconditioner(cgamd_solver *s, const void *m, int on_device) {
    if (!s) return fail(CGAMD_ERR_INVALID, "set_preconditioner: solver is NULL");
    TuneScope ts(&s->tune);
    CG_HIP(hipSetDevice(s->ctx->device));
    CG_HIP(hipStreamSynchronize(s->ctx->stream));
    destroy_graphs(s);
    s->rhs_set = false;
    if (!m) {
        if (s->mdiag) { (void)hipFree(s->mdiag); s->mdiag = nullptr; }
        apply_wide_order(s);
        return CGAMD_OK;
    }
    const size_t vs = dtype_size(s->dtype);
    int rc = CGAMD_OK;
    if (!s->mdiag) rc = dmalloc(&s->mdiag, (size_t)s->n * vs, "preconditioner");
    if (!rc && !s->part_rz) rc = dmalloc(&s->part_rz, acc_size(s->dtype) * s->part_rr_cap * s->nrhs, "partials_rz");
    if (!rc && !s->rho2) rc = dmalloc(&s->rho2, 2 * vs * (size_t)s->nrhs, "rho");
    if (rc) return rc;
    if (s->n != s->n_user) CG_HIP(hipMemsetAsync(s->mdiag, 0, (size_t)s->n * vs, s->ctx->stream));
    CG_HIP(hipMemcpyAsync(s->mdiag, m, (size_t)s->n_user * vs, on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, s->ctx->stream));
    CG_HIP(hipStreamSynchronize(s->ctx->stream));
    if (!s->resw.ok)
        if (int rc2 = setup_resident_wide_plan(s)) return rc2;      // (skipped at creation where the one-XCD loop runs the plain recurrence)
    apply_wide_order(s);
    return CGAMD_OK;
}

int cgamd_solver_iterate(cgamd_solver *s, int nIterations) {
    if (!s) return fail(CGAMD_ERR_INVALID, "iterate: solver is NULL");
    TuneScope ts(&s->tune);
    if (!s->rhs_set) return fail(CGAMD_ERR_STATE, "iterate: call set_rhs first");
    if (nIterations < 0) return fail(CGAMD_ERR_INVALID, "iterate: negative iteration count");
    CG_HIP(hipSetDevice(s->ctx->device));
    if (int rc = ensure_history(s, s->iters + nIterations + 1)) return rc;
    hipStream_t st = s->ctx->stream;
    int left = nIterations, k = s->iters;
    const bool use_graph = !(s->flags & CGAMD_NO_GRAPH) && !s->graph_failed;
    const bool two = fused2_now(s);
    if (s->resw.ok && !(two && s->res_ok) && !s->rm && !(s->flags & (CGAMD_NO_GRAPH | CGAMD_UNFUSED)) &&
        (nIterations >= std::max(1, tune().resident_wide_min) || s->tol_req > 0.)) {
        // one chip-wide resident group (single right-hand side, matrix rows in registers).  d ping-pongs inside the launch; handles
        // of the launched loops that keep d in one buffer get it back there, and the launched loops' r.r partials are rebuilt.
        const bool keeps_new_d = !two;       // three / four-launch loops: between iterations d already is beta d + r
        // a launch stays around a second at most (groups solve their right-hand sides in turn, ~10 us per iteration)
        const int rounds_w = (s->nrhs + s->resw.NG - 1) / s->resw.NG, kmax_w = std::min(1 << 15, std::max(64, 100000 / rounds_w));
        for (int left = nIterations; left > 0;) {
            const int K = std::min(left, kmax_w);
            void *cur = dbuf(s, s->iters), *other = cur == s->d ? s->d2 : s->d;
            void *d0 = (s->iters & 1) ? other : cur, *d1 = (s->iters & 1) ? cur : other;
            bool untouched = false;
            int stop = -1;
            s->tol_served = s->tol_req > 0.;
            CgScalars scw = s->sc;           // the Jacobi-preconditioned recurrence runs in the same loop (rho for delta, z = m r)
            if (s->mdiag) { scw.pcg_m = s->mdiag; scw.pcg_rho2 = s->rho2; }
            if (int rc = run_cg_resident_wide(s->dtype, s->resw, s->n, s->nrhs, s->vals, s->ptr, s->cols, s->x, s->r, d0, d1,
                                              keeps_new_d && s->iters > 0, scw, s->iters, K, s->resw_sync, s->n_cus, st, &untouched, s->tol_req,
                                              &stop)) {
                if (!untouched) {            // x / r / d / delta may be partly advanced: the handle demands a fresh set_rhs
                    s->rhs_set = false;
                    s->resw.ok = false;
                    return rc;
                }
                s->resw.ok = false;          // the chip is shared with something that does not yield: this handle keeps the launched loops
                if (int rc2 = launch_dot_partials(s->dtype, s->n, s->r, s->r, s->n, s->nrhs, s->part_rr, s->vgrid, st)) return rc2;
                if (s->tol_req > 0.) { s->tol_served = false; return rc; }      // (the launched loops have no device-side stop: the caller checks from the host)
                return cgamd_solver_iterate(s, left);
            }
            const int done = stop >= 0 ? stop - s->iters : K;      // the tolerance may end the solve before K iterations
            void *fin = ((s->iters + done) & 1) ? d1 : d0;
            s->iters += done;
            left = stop >= 0 ? 0 : left - K;
            s->tol_stopped = stop >= 0;
            if (done > 0) {                  // (a stop before the first iteration leaves d as the caller had it)
                if (fin != dbuf(s, s->iters))
                    CG_HIP(hipMemcpyAsync(dbuf(s, s->iters), fin, (size_t)s->n * s->nrhs * dtype_size(s->dtype), hipMemcpyDeviceToDevice, st));
                if (s->mdiag) {              // p = beta p + m r (helmFE_var.py:583-585)
                    if (int rc = launch_pcg_p_update(s->dtype, s->n, s->r, dbuf(s, s->iters), s->mdiag, s->n, s->sc.beta, s->nrhs, st)) return rc;
                } else if (keeps_new_d)      // d = beta d + r with the beta the launch recorded last (clcg.c:415)
                    if (int rc = launch_aypx(s->dtype, s->n, s->r, dbuf(s, s->iters), s->n, s->sc.beta, s->nrhs, st)) return rc;
            }
        }
        return launch_dot_partials(s->dtype, s->n, s->r, s->r, s->n, s->nrhs, s->part_rr, s->vgrid, st);
    }
    if (two && s->res_ok && !(s->flags & CGAMD_NO_GRAPH) && (nIterations >= std::max(1, tune().resident_min) || s->tol_req > 0.)) {
        // small system: the whole call in one launch per 2^15 iterations (resident.hip; a launch stays well below the bound of its
        // waits); same state, same bits as the loop below
        const int groups_r = std::max(1, 8 * s->res.lg), rounds_r = (s->nrhs + groups_r - 1) / groups_r;
        const int kmax_r = std::min(1 << 15, std::max(64, 200000 / rounds_r));      // a launch stays around a second at most
        for (int left = nIterations; left > 0;) {
            const int K = std::min(left, kmax_r);
            bool untouched = false;
            int stop = -1;
            s->tol_served = s->tol_req > 0.;
            if (int rc = run_cg_resident(s->dtype, s->res, s->n, s->nrhs, s->vals, s->ptr, s->cols, s->x, s->r, s->d, s->d2, s->part_rr,
                                         s->vgrid, s->plan.n_partials, s->sc, s->iters, K, s->res_sync, s->n_cus, st, &untouched, s->tol_req, &stop)) {
                if (!untouched) {            // (see above)
                    s->rhs_set = false;
                    s->res_ok = false;
                    return rc;
                }
                s->res_ok = false;           // no group could form (CUs held by other work): this handle keeps the launched loops
                if (s->tol_req > 0.) { s->tol_served = false; return rc; }
                return cgamd_solver_iterate(s, left);
            }
            if (stop >= 0) {             // the tolerance ended the solve: the launched loops' r.r partials of that state are rebuilt
                s->iters = stop;
                s->tol_stopped = true;
                return launch_dot_partials(s->dtype, s->n, s->r, s->r, s->n, s->nrhs, s->part_rr, s->vgrid, st);
            }
            s->iters += K;
            left -= K;
        }
        return CGAMD_OK;
    }
    // graphs start at a fixed parity of the iteration count (d ping-pongs in the two-launch loop; U is even)
    while (use_graph && !s->graph_failed && left > 0) {
        const int par = two ? (k & 1) : 0;
        const bool big = left >= s->U;
        hipGraphExec_t &ge = big ? s->gU[par] : s->g1[par];
        if (!ge && capture(s, par, big ? s->U : 1, big ? &s->gUg[par] : &s->g1g[par], &ge) != CGAMD_OK) {
            s->graph_failed = true;
            destroy_graphs(s);
            break;
        }
        CG_HIP(hipGraphLaunch(ge, st));
        left -= big ? s->U : 1;
        k += big ? s->U : 1;
    }
    for (; left > 0; --left, ++k)
        if (int rc = enqueue_iteration(s, k, st)) {
            s->iters = k + 1;      // the device counter may have advanced for the broken iteration too
            return rc;
        }
    s->iters = k;
    if (two && nIterations > 0) return launch_cg_tail(s->dtype, s->part_rr, s->vgrid, s->nrhs, s->sc, st);
    return CGAMD_OK;
}

// Tolerance-stopping run ON THE DEVICE (one right-hand side, handles that take a resident loop): iterations until
// sqrt|r.r| < tol (or NaN), at most maxIterations -- the reference's NumPy sub-solver with `tol` (p_h-PY_C-CL.py:1338-1369) and
// PCG's stopping rule without preconditioner (helmFE_var.py:575-577).  Every member of the resident group sees the same delta and
// leaves the loop in the same iteration, so x is the iterate of exactly *iterations_run iterations; no read-back per check, no
// re-run.  CGAMD_ERR_STATE when the handle has no resident loop (the caller then checks the history from the host).
int cgamd_solver_iterate_tol(cgamd_solver *s, int maxIterations, double tol, int *iterations_run) {
    if (!s || !iterations_run) return fail(CGAMD_ERR_INVALID, "iterate_tol: null argument");
    if (!(tol > 0.) || maxIterations < 0) return fail(CGAMD_ERR_INVALID, "iterate_tol: tol must be positive, maxIterations >= 0");
    if (!s->rhs_set) return fail(CGAMD_ERR_STATE, "iterate_tol: call set_rhs first");
    if (s->nrhs != 1) return fail(CGAMD_ERR_STATE, "iterate_tol: one right-hand side");
    {
        TuneScope ts(&s->tune);
        const bool local = fused2_now(s) && s->res_ok, wide = s->resw.ok && !s->rm && !(s->res_ok && !s->mdiag);
        if ((!local && !wide) || (s->flags & (CGAMD_NO_GRAPH | CGAMD_UNFUSED)))
            return fail(CGAMD_ERR_STATE, "iterate_tol: this handle runs a launched loop (check the history from the host)");
    }
    const int start = s->iters;
    s->tol_req = tol;
    s->tol_served = s->tol_stopped = false;
    int rc = maxIterations > 0 ? cgamd_solver_iterate(s, maxIterations) : CGAMD_OK;
    const bool served = s->tol_served || maxIterations == 0;
    s->tol_req = 0.;
    if (rc) return rc;
    if (!served) return fail(CGAMD_ERR_STATE, "iterate_tol: the resident loop was not available for this call");
    *iterations_run = s->iters - start;
    return CGAMD_OK;
}

// nIterations plain-launch iterations of the SAME launch sequence cgamd_solver_iterate replays (enqueue_iteration), with a
// HIP event pair around every SpMV launch on the solver's stream: the in-loop duration of the dominant kernel
// (bench.py's roofline).  Synchronises.
int cgamd_solver_iterate_timed(cgamd_solver *s, int nIterations, float *spmv_ms_avg, float *iter_ms_avg) {
    if (!s || !spmv_ms_avg) return fail(CGAMD_ERR_INVALID, "iterate_timed: null argument");
    TuneScope ts(&s->tune);
    if (!s->rhs_set) return fail(CGAMD_ERR_STATE, "iterate_timed: call set_rhs first");
    if (nIterations < 1) return fail(CGAMD_ERR_INVALID, "iterate_timed: needs >= 1 iteration");
    CG_HIP(hipSetDevice(s->ctx->device));
    if (int rc = ensure_history(s, s->iters + nIterations + 1)) return rc;
    hipStream_t st = s->ctx->stream;
    std::vector<hipEvent_t> ev((size_t)2 * nIterations + 2, nullptr);
    int rc = CGAMD_OK, done = 0;
    hipError_t e = hipSuccess;
    for (auto &x : ev)
        if (e == hipSuccess) e = hipEventCreate(&x);
    if (e == hipSuccess) e = hipEventRecord(ev[(size_t)2 * nIterations], st);
    for (int i = 0; i < nIterations && e == hipSuccess && !rc; ++i) {
        s->ev_pair = &ev[(size_t)2 * i];
        rc = enqueue_iteration(s, s->iters + i, st);
        ++done;
    }
    s->ev_pair = nullptr;
    s->iters += done;       // whatever was enqueued counts, also on the error paths below
    if (e == hipSuccess && !rc) e = hipEventRecord(ev[(size_t)2 * nIterations + 1], st);
    if (e == hipSuccess && !rc && fused2_now(s)) rc = launch_cg_tail(s->dtype, s->part_rr, s->vgrid, s->nrhs, s->sc, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    double sum = 0.0;
    float total = 0.f;
    for (int i = 0; i < nIterations && e == hipSuccess && !rc; ++i) {
        float ms = 0.f;
        e = hipEventElapsedTime(&ms, ev[(size_t)2 * i], ev[(size_t)2 * i + 1]);
        sum += ms;
    }
    if (e == hipSuccess && !rc) e = hipEventElapsedTime(&total, ev[(size_t)2 * nIterations], ev[(size_t)2 * nIterations + 1]);
    for (auto &x : ev)
        if (x) (void)hipEventDestroy(x);
    if (rc) return rc;
    if (e != hipSuccess) return fail(CGAMD_ERR_HIP, std::string("iterate_timed: ") + hipGetErrorString(e));
    *spmv_ms_avg = (float)(sum / nIterations);
    if (iter_ms_avg) *iter_ms_avg = total / nIterations;
    return CGAMD_OK;
}

int cgamd_solver_get_x(cgamd_solver *s, void *x, int on_device) {
    if (!s || !x) return fail(CGAMD_ERR_INVALID, "get_x: null argument");
    TuneScope ts(&s->tune);
    CG_HIP(hipSetDevice(s->ctx->device));
    const size_t vbytes = (size_t)s->n * s->nrhs * dtype_size(s->dtype);
    const void *src = s->x;
    if (s->rm) {    // back to the caller's RHS-major layout; q is dead between iterations (recomputed first thing in the next)
        if (int rc = launch_transpose(s->dtype, s->n, s->nrhs, s->x, s->q, s->ctx->stream)) return rc;
        src = s->q;
    }
    const hipMemcpyKind kind = on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
    if (s->n != s->n_user) {
        const size_t vs = dtype_size(s->dtype);
        CG_HIP(hipMemcpy2DAsync(x, (size_t)s->n_user * vs, src, (size_t)s->n * vs, (size_t)s->n_user * vs, (size_t)s->nrhs, kind, s->ctx->stream));
    } else {
        CG_HIP(hipMemcpyAsync(x, src, vbytes, kind, s->ctx->stream));
    }
    if (!on_device) CG_HIP(hipStreamSynchronize(s->ctx->stream));
    return CGAMD_OK;
}

int cgamd_solver_iterations_done(cgamd_solver *s) { return s ? s->iters : -CGAMD_ERR_INVALID; }

int cgamd_solver_history(cgamd_solver *s, void *history, int max_entries) {
    if (!s || !history) { fail(CGAMD_ERR_INVALID, "history: null argument"); return -CGAMD_ERR_INVALID; }
    if (!s->rhs_set) { fail(CGAMD_ERR_STATE, "history: no right-hand side set"); return -CGAMD_ERR_STATE; }
    thread_hip_setup();
    if (hipSetDevice(s->ctx->device) != hipSuccess) return -CGAMD_ERR_NO_DEVICE;
    const int entries = std::min(std::min(s->iters + 1, s->sc.history_cap), max_entries);
    hipError_t e = hipMemcpyAsync(history, s->sc.history, (size_t)entries * s->nrhs * dtype_size(s->dtype),
                                  hipMemcpyDeviceToHost, s->ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(s->ctx->stream);
    if (e != hipSuccess) { fail(CGAMD_ERR_HIP, std::string("history: ") + hipGetErrorString(e)); return -CGAMD_ERR_HIP; }
    return entries;
}

int cgamd_solver_ld(cgamd_solver *s) { return s ? s->n : -CGAMD_ERR_INVALID; }

void *cgamd_solver_vector(cgamd_solver *s, int which) {
    if (!s) return nullptr;
    switch (which) { case 0: return s->x; case 1: return s->r; case 2: return dbuf(s, s->iters); case 3: return s->q; default: return nullptr; }
}

int cgamd_solver_solve(cgamd_solver *s, const void *b, void *x, int nIterations, void *history) {
    if (!s || !b || !x) return fail(CGAMD_ERR_INVALID, "solve: null argument");
    int rc;
    if ((rc = cgamd_solver_set_rhs(s, b, x, 0))) return rc;   // x is in/out: initial guess (clcg.c:210)
    if ((rc = cgamd_solver_iterate(s, nIterations))) return rc;
    if ((rc = cgamd_solver_get_x(s, x, 0))) return rc;
    if (history) {
        const int got = cgamd_solver_history(s, history, nIterations + 1);
        if (got < 0) return -got;
    }
    return CGAMD_OK;
}

int cgamd_solver_spmv(cgamd_solver *s, const void *x, void *y, int fused_dot) {
    if (!s || !x || !y) return fail(CGAMD_ERR_INVALID, "solver_spmv: null argument");
    TuneScope ts(&s->tune);
    CG_HIP(hipSetDevice(s->ctx->device));
    // the caller's vectors have the caller's stride; the plan fits both sizes (same row blocks, the appended rows are empty)
    return launch_spmv(s->dtype, s->plan, s->n_user, s->nnz, s->vals, s->ptr, s->cols, x, s->n_user, y, s->n_user, s->nrhs,
                       fused_dot ? x : nullptr, fused_dot ? s->part_dq : nullptr, s->ctx->stream);
}

int cgamd_solver_spmm_rowmajor(cgamd_solver *s, const void *x, void *y, int nRHS) {
    if (!s || !x || !y) return fail(CGAMD_ERR_INVALID, "spmm_rowmajor: null argument");
    TuneScope ts(&s->tune);
    CG_HIP(hipSetDevice(s->ctx->device));
    return launch_spmm_rm(s->dtype, s->n_user, s->nnz, s->vals, s->ptr, s->cols, x, y, nRHS, nullptr, s->plan.max_quad, s->rm_pace, s->ctx->stream);
}

int cgamd_solver_layout(cgamd_solver *s) { return s ? (s->rm ? 1 : 0) : -CGAMD_ERR_INVALID; }
int cgamd_solver_loop_launches(cgamd_solver *s) {
    if (!s) return -CGAMD_ERR_INVALID;
    TuneScope ts(&s->tune);
    if (s->flags & CGAMD_UNFUSED) return 8;
    if (s->rm) return 5;
    if (s->mdiag) return (s->resw.ok && !s->rm && !(s->flags & (CGAMD_NO_GRAPH | CGAMD_UNFUSED))) ? 1 : 4;
    if (fused2_now(s) && s->res_ok && !(s->flags & CGAMD_NO_GRAPH)) return 0;
    if (s->resw.ok && !(s->flags & CGAMD_NO_GRAPH)) return 1;       // chip-wide resident group (same bits as the launched loops of this handle)
    if (fused2_now(s)) return 2;
    return fold_alpha_ok(s->plan.n_partials, s->plan.fold_max) ? 3 : 4;
}

int cgamd_solver_index_codes(cgamd_solver *s) { return s ? s->n_offsets : -CGAMD_ERR_INVALID; }

long long cgamd_solver_spmv_bytes(cgamd_solver *s) {
    if (!s) return 0;
    const long long V = (long long)dtype_size(s->dtype);
    return s->nnz * (V + 4) + ((long long)s->n_user + 1) * 4 + 2LL * s->n_user * V * s->nrhs;
}
long long cgamd_solver_iter_bytes(cgamd_solver *s, int fused) {
    if (!s) return 0;
    const long long V = (long long)dtype_size(s->dtype);
    return s->nnz * (V + 4) + ((long long)s->n_user + 1) * 4 + (fused ? 11LL : 14LL) * s->n_user * V * s->nrhs;
}

// What the handle's own kernels MOVE (the physical byte model the roofline fraction is priced on): index bytes per non-zero as
// the SpMV really reads them (1 with the one-byte column codes, 2 with 16-bit block-relative columns, else 4) and the vector
// passes of the launched loop the handle runs (10 with the deferred x update, 11 without, 12 preconditioned, 14 for the
// reference's op structure).  Handles whose iterate() runs a resident loop report the launched loop they fall back to.
static long long index_bytes_per_nnz(const cgamd_solver *s) { return s->plan.codes ? (s->plan.codes16 ? 2 : 1) : 4; }
// the single-RHS SpMV of this handle runs on joint codes (launch condition of spmv_impl: rows that fit one batch of the walk)
static bool joint_form(const cgamd_solver *s) {
    const int fit = s->plan.max_row <= 0 ? 8 : s->plan.max_row <= 4 ? 4 : s->plan.max_row == 5 ? 5 : s->plan.max_row <= 7 ? 7 : 8;
    return s->plan.jcodes && s->nrhs == 1 && s->plan.max_row > 0 && s->plan.max_row <= fit && dtype_size(s->dtype) <= 8;
}
long long cgamd_solver_spmv_moved_bytes(cgamd_solver *s) {
    if (!s) return 0;
    const long long V = (long long)dtype_size(s->dtype);
    const long long value_bytes = joint_form(s) ? 0 : s->plan.vcodes ? 1 : V;      // value codes: one byte per entry instead of the value; joint codes: one byte for both
    return s->nnz * (value_bytes + index_bytes_per_nnz(s)) + ((long long)s->n_user + 1) * 4 + 2LL * s->n_user * V * s->nrhs;
}
int cgamd_solver_value_codes(cgamd_solver *s) { return s ? s->n_values : -CGAMD_ERR_INVALID; }
int cgamd_solver_joint_codes(cgamd_solver *s) { return s ? (joint_form(s) ? s->n_pairs : 0) : -CGAMD_ERR_INVALID; }
long long cgamd_solver_iter_moved_bytes(cgamd_solver *s) {
    if (!s) return 0;
    const long long V = (long long)dtype_size(s->dtype);
    const long long passes = (s->flags & CGAMD_UNFUSED) ? 14 : s->mdiag ? 12 : 10;
    const long long value_bytes = joint_form(s) ? 0 : s->plan.vcodes ? 1 : V;
    return s->nnz * (value_bytes + index_bytes_per_nnz(s)) + ((long long)s->n_user + 1) * 4 + passes * s->n_user * V * s->nrhs;
}

}  // extern "C"
