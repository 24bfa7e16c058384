// Context, error strings and the host-side index maps of libfcdiff_hip.so.
#include <new>
#include <stdlib.h>

#include "fcd_common.h"
#include "fcd_fastmath.h"

#ifdef FCD_ABLATE
__device__ int fcd_abl_level[4];
__device__ unsigned long long *fcd_trace_buf;
void fcd_abl_refresh(hipStream_t s) {
    unsigned long long *tp = nullptr;
    if (const char *e = getenv("FCD_TRACE_PTR")) tp = (unsigned long long *)strtoull(e, nullptr, 0);
    (void)hipMemcpyToSymbolAsync(HIP_SYMBOL(fcd_trace_buf), &tp, sizeof(tp), 0, hipMemcpyHostToDevice, s);
    int v[4] = {0, 0, 0, 0};
    if (const char *e = getenv("FCD_ABL_F")) v[0] = atoi(e);
    if (const char *e = getenv("FCD_ABL_PANEL")) v[1] = atoi(e);
    if (const char *e = getenv("FCD_ABL_DIAG")) v[2] = atoi(e);
    if (const char *e = getenv("FCD_TRACE_ROW")) v[3] = atoi(e);       // stamps inside one row of the pipelined scan
    (void)hipMemcpyToSymbolAsync(HIP_SYMBOL(fcd_abl_level), v, sizeof(v), 0, hipMemcpyHostToDevice, s);
}
#endif

int fcd_ws_reserve(fcd_ctx *ctx, size_t bytes) {
    if (bytes <= ctx->ws_bytes) return FCD_OK;
    // grow: the old block may still be in use by kernels already queued, so drain first
    FCD_HIP_TRY(hipDeviceSynchronize());
    if (ctx->ws) FCD_HIP_TRY(hipFree(ctx->ws));
    ctx->ws = nullptr;
    ctx->ws_bytes = 0;
    size_t want = bytes < (1u << 20) ? (1u << 20) : bytes;
    FCD_HIP_TRY(hipMalloc(&ctx->ws, want));
    ctx->ws_bytes = want;
    ctx->n_alloc += 1;
    return FCD_OK;
}

int fcd_fsq_reserve(fcd_ctx *ctx, size_t bytes) {
    if (bytes <= ctx->fsq_bytes) return FCD_OK;
    FCD_HIP_TRY(hipDeviceSynchronize());
    if (ctx->fsq) FCD_HIP_TRY(hipFree(ctx->fsq));
    ctx->fsq = nullptr;
    ctx->fsq_bytes = 0;
    FCD_HIP_TRY(hipMalloc(&ctx->fsq, bytes));
    ctx->fsq_bytes = bytes;
    ctx->n_alloc += 1;
    return FCD_OK;
}

void fcd_sweep_ws_bytes(const fcd_ctx *ctx, int64_t Nreg, int64_t U, int64_t GW, size_t *ws_bytes, size_t *fsq_bytes) {
    const size_t f = fcd_f_pass_ws_bytes(Nreg, U, GW), r = fcd_r_pass_ws_bytes(Nreg, U, GW, ctx->knobs.r_path);
    *ws_bytes = f > r ? f : r;
    *fsq_bytes = fcd_fsq_need_bytes(Nreg, U, GW);
}

// "0", "" and unset mean "default"; anything else is the number
static double knob_env(const char *name) {
    const char *e = getenv(name);
    return (e && *e) ? atof(e) : 0.0;
}

void fcd_prof_begin(fcd_ctx *ctx, int slot, hipStream_t s) {
    if (!ctx->prof_on) return;
    if (ctx->prof_n[slot] >= ctx->prof_cap[slot]) {
        const int cap = ctx->prof_cap[slot] ? ctx->prof_cap[slot] * 2 : 256;
        hipEvent_t *ev = new (std::nothrow) hipEvent_t[2 * cap];
        if (!ev) return;
        for (int i = 0; i < 2 * ctx->prof_cap[slot]; ++i) ev[i] = ctx->prof_ev[slot][i];
        for (int i = 2 * ctx->prof_cap[slot]; i < 2 * cap; ++i)
            if (hipEventCreate(&ev[i]) != hipSuccess) { delete[] ev; return; }
        delete[] ctx->prof_ev[slot];
        ctx->prof_ev[slot] = ev;
        ctx->prof_cap[slot] = cap;
    }
    (void)hipEventRecord(ctx->prof_ev[slot][2 * ctx->prof_n[slot]], s);
}

void fcd_prof_end(fcd_ctx *ctx, int slot, hipStream_t s) {
    if (!ctx->prof_on || ctx->prof_n[slot] >= ctx->prof_cap[slot]) return;
    (void)hipEventRecord(ctx->prof_ev[slot][2 * ctx->prof_n[slot] + 1], s);
    ctx->prof_n[slot] += 1;
}

extern "C" {

int fcd_prof_enable(fcd_ctx *ctx, int on) {
    if (!ctx) return FCD_ERR_ARG;
    ctx->prof_on = on ? 1 : 0;
    for (int i = 0; i < FCD_PROF_SLOTS; ++i) ctx->prof_n[i] = 0;
    return FCD_OK;
}

int fcd_prof_collect(fcd_ctx *ctx, int slot, double *total_ms, int64_t *count) {
    if (!ctx || slot < 0 || slot >= FCD_PROF_SLOTS || !total_ms || !count) return FCD_ERR_ARG;
    double tot = 0.0;
    for (int i = 0; i < ctx->prof_n[slot]; ++i) {
        FCD_HIP_TRY(hipEventSynchronize(ctx->prof_ev[slot][2 * i + 1]));
        float ms = 0.f;
        FCD_HIP_TRY(hipEventElapsedTime(&ms, ctx->prof_ev[slot][2 * i], ctx->prof_ev[slot][2 * i + 1]));
        tot += ms;
    }
    *total_ms = tot;
    *count = ctx->prof_n[slot];
    ctx->prof_n[slot] = 0;
    return FCD_OK;
}

int fcd_abi_version(void) { return FCD_ABI_VERSION; }

const char *fcd_strerror(int code) {
    switch (code) {
        case FCD_OK: return "ok";
        case FCD_ERR_ARG: return "invalid argument (null pointer or non-positive size)";
        case FCD_ERR_SHAPE: return "invalid shape (number of connections must be a triangular number, Nreg >= 2)";
        case FCD_ERR_UNSUPPORTED: return "shape outside what the gfx950 kernels are built for";
        case FCD_ERR_INDEX: return "reference edge id out of range";
        case FCD_ERR_DEVICE: return "a kernel abandoned a device-side wait; the chain state is unusable";
        case FCD_ERR_COMM: return "RCCL: library not found or a call failed";
        default: break;
    }
    if (code > 0) return hipGetErrorString((hipError_t)code);
    return "unknown fcdiff_hip error";
}

const char *fcd_last_message(const fcd_ctx *ctx) { return ctx ? ctx->msg : ""; }

int fcd_ctx_create(fcd_ctx **out) {
    if (!out) return FCD_ERR_ARG;
    *out = nullptr;
    int dev = 0;
    FCD_HIP_TRY(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    FCD_HIP_TRY(hipGetDeviceProperties(&prop, dev));
    fcd_ctx *ctx = new (std::nothrow) fcd_ctx();
    if (!ctx) return (int)hipErrorOutOfMemory;
    ctx->device = dev;
    ctx->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    ctx->ws = nullptr;
    ctx->ws_bytes = 0;
    ctx->log_tab = nullptr;
    ctx->dev_err = nullptr;
    ctx->fsq = nullptr;
    ctx->fsq_bytes = 0;
    ctx->acc = nullptr;
    ctx->dbg = nullptr;
    ctx->comm = nullptr;
    ctx->comm_world = ctx->comm_rank = 0;
    ctx->pool_counts = nullptr;
    ctx->corr_tickets = nullptr;
    ctx->corr_tickets_n = 0;
    ctx->prof_on = 0;
    for (int i = 0; i < FCD_PROF_SLOTS; ++i) {
        ctx->prof_ev[i] = nullptr;
        ctx->prof_n[i] = ctx->prof_cap[i] = 0;
    }
    ctx->msg[0] = 0;
    ctx->n_alloc = 0;
    for (int i = 0; i < FCD_KA_N; ++i) ctx->lds_attr[i] = 0;
    for (int i = 0; i < 3; ++i) ctx->pipe_occ[i] = -1;
    // the only place the environment is read: defaults of the knobs (fcd_ctx_set_knob changes them later)
    ctx->knobs.r_path = (int)knob_env("FCD_R_PATH");
    ctx->knobs.r_ub = (int)knob_env("FCD_R_UB");
    ctx->knobs.r_nopad = (int)knob_env("FCD_R_NOPAD");
    ctx->knobs.r_dsplit = (int)knob_env("FCD_R_DSPLIT");
    ctx->knobs.r_coop = (int)knob_env("FCD_R_COOP");
    ctx->knobs.qr_form = (int)knob_env("FCD_QR_FORM");
    ctx->knobs.r_refill = (int)knob_env("FCD_R_REFILL");
    ctx->knobs.r_tol = knob_env("FCD_R_TOL");
    ctx->knobs.f_tol = knob_env("FCD_F_TOL");
    ctx->knobs.f_form = (int)knob_env("FCD_F_FORM");
    ctx->knobs.corr_form = (int)knob_env("FCD_CORR_FORM");
    ctx->knobs.r_poll_limit = 0;        // test hooks: through fcd_ctx_set_knob only, never from the environment
    ctx->knobs.r_withhold = 0;
    ctx->r_form_last = 0;
    int rc = fcd_ws_reserve(ctx, 1u << 20);
    if (rc == FCD_OK) {
        void *pin = nullptr;
        rc = (int)hipHostMalloc(&pin, sizeof(unsigned), hipHostMallocDefault);
        if (rc == FCD_OK) {
            ctx->dev_err = (volatile unsigned *)pin;
            *ctx->dev_err = 0u;
        }
    }
    if (rc) {
        if (ctx->ws) (void)hipFree(ctx->ws);
        delete ctx;
        return rc;
    }
    {
        // tables of K_lik's exp and log (fcd_fastmath.h): 2^(-j/64) and {1/m_i, log m_i}, made with the host's libm
        struct { double exp_tab[FCD_EXP_CELLS]; fcd_log_cell log_tab[FCD_LOG_CELLS]; } tabs;
        fcd_fm_make_tables(tabs.exp_tab, tabs.log_tab, exp2, log);
        hipError_t e = hipMalloc(&ctx->log_tab, sizeof(tabs));
        ctx->n_alloc += 1;
        if (e == hipSuccess) e = hipMemcpy(ctx->log_tab, &tabs, sizeof(tabs), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMalloc(&ctx->acc, 8 * sizeof(unsigned long long));
        ctx->n_alloc += 1;
        if (e == hipSuccess) e = hipMemset(ctx->acc, 0, 8 * sizeof(unsigned long long));
        if (e == hipSuccess) e = hipMalloc(&ctx->dbg, 8 * sizeof(unsigned long long));
        ctx->n_alloc += 1;
        if (e == hipSuccess) e = hipMemset(ctx->dbg, 0, 8 * sizeof(unsigned long long));
        if (e != hipSuccess) {
            fcd_ctx_destroy(ctx);
            return (int)e;
        }
    }
    *out = ctx;
    return FCD_OK;
}

int fcd_ctx_destroy(fcd_ctx *ctx) {
    if (!ctx) return FCD_OK;
    (void)fcd_comm_destroy(ctx);
    if (ctx->pool_counts) (void)hipFree(ctx->pool_counts);
    hipError_t e = hipSuccess;
    if (ctx->ws) e = hipFree(ctx->ws);
    if (ctx->log_tab) (void)hipFree(ctx->log_tab);
    if (ctx->dev_err) (void)hipHostFree((void *)ctx->dev_err);
    if (ctx->fsq) (void)hipFree(ctx->fsq);
    if (ctx->acc) (void)hipFree(ctx->acc);
    if (ctx->dbg) (void)hipFree(ctx->dbg);
    if (ctx->corr_tickets) (void)hipFree(ctx->corr_tickets);
    for (int i = 0; i < FCD_PROF_SLOTS; ++i) {
        for (int j = 0; j < 2 * ctx->prof_cap[i]; ++j) (void)hipEventDestroy(ctx->prof_ev[i][j]);
        delete[] ctx->prof_ev[i];
    }
    delete ctx;
    return (int)e;
}

int fcd_ctx_reserve(fcd_ctx *ctx, int64_t Nreg, int64_t U, int64_t G) {
    if (!ctx || Nreg < 2 || U < 1 || G < 1) return fcd_fail(ctx, FCD_ERR_ARG, "fcd_ctx_reserve: bad argument");
    const int64_t GW = (G + 63) / 64;
    size_t need = (size_t)64 * GW * 64 * sizeof(double);        // log-joint partials
    const size_t en = (size_t)ctx->num_cu * 8 * 8 * sizeof(double);  // energy partials
    if (en > need) need = en;
    size_t sweep = 0, fsq = 0;
    fcd_sweep_ws_bytes(ctx, Nreg, U, GW, &sweep, &fsq);             // f / r pass scratch and the square f copy
    if (sweep > need) need = sweep;
    int rc = fcd_ws_reserve(ctx, need);
    if (rc) return rc;
    return fsq ? fcd_fsq_reserve(ctx, fsq) : FCD_OK;
}

int fcd_ctx_set_knob(fcd_ctx *ctx, const char *name, double value) {
    if (!ctx || !name) return FCD_ERR_ARG;
    fcd_knobs &k = ctx->knobs;
    if (!strcmp(name, "r_path")) k.r_path = (int)value;
    else if (!strcmp(name, "r_ub")) k.r_ub = (int)value;
    else if (!strcmp(name, "r_nopad")) k.r_nopad = (int)value;
    else if (!strcmp(name, "r_dsplit")) k.r_dsplit = (int)value;
    else if (!strcmp(name, "r_coop")) k.r_coop = (int)value;
    else if (!strcmp(name, "qr_form")) k.qr_form = (int)value;
    else if (!strcmp(name, "r_refill")) k.r_refill = (int)value;
    else if (!strcmp(name, "r_tol")) k.r_tol = value;
    else if (!strcmp(name, "f_tol")) k.f_tol = value;
    else if (!strcmp(name, "f_form")) k.f_form = (int)value;
    else if (!strcmp(name, "corr_form")) k.corr_form = (int)value;
    else if (!strcmp(name, "r_poll_limit")) k.r_poll_limit = (int)value;
    else if (!strcmp(name, "r_withhold")) k.r_withhold = (int)value;
    else return fcd_fail(ctx, FCD_ERR_ARG, "fcd_ctx_set_knob: unknown knob");
    return FCD_OK;
}

int fcd_ctx_check(fcd_ctx *ctx) {
    if (!ctx) return FCD_ERR_ARG;
    if (ctx->dev_err && *ctx->dev_err)
        return fcd_fail(ctx, FCD_ERR_DEVICE, "a device-side wait of the pipelined r pass was abandoned: the chain state is unusable");
    return FCD_OK;
}

int fcd_ctx_clear_error(fcd_ctx *ctx) {
    if (!ctx) return FCD_ERR_ARG;
    if (ctx->dev_err) *ctx->dev_err = 0u;
    // whatever an abandoned launch left in the context-owned accumulators / tickets (zero between launches by contract)
    if (ctx->acc) FCD_HIP_TRY(hipMemset(ctx->acc, 0, 8 * sizeof(unsigned long long)));
    if (ctx->corr_tickets && ctx->corr_tickets_n > 0)
        FCD_HIP_TRY(hipMemset(ctx->corr_tickets, 0, (size_t)ctx->corr_tickets_n * sizeof(unsigned)));
    return FCD_OK;
}

int fcd_ctx_stat(const fcd_ctx *ctx, const char *name, int64_t *out) {
    if (!ctx || !name || !out) return FCD_ERR_ARG;
    if (!strcmp(name, "n_alloc")) *out = ctx->n_alloc;
    else if (!strcmp(name, "ws_bytes")) *out = (int64_t)ctx->ws_bytes;
    else if (!strcmp(name, "fsq_bytes")) *out = (int64_t)ctx->fsq_bytes;
    else if (!strcmp(name, "r_form_last")) *out = ctx->r_form_last;
    else if (!strcmp(name, "comm_world")) *out = ctx->comm ? ctx->comm_world : 0;
    else if (!strcmp(name, "dev_err")) *out = ctx->dev_err ? (int64_t)*ctx->dev_err : 0;
    else if (!strcmp(name, "f_repeats") || !strcmp(name, "r_exact_rows")) {
        // event counters kept on the device (a synchronising read: diagnostics only)
        unsigned long long v[8];
        FCD_HIP_TRY(hipMemcpy(v, ctx->dbg, sizeof(v), hipMemcpyDeviceToHost));
        *out = (int64_t)v[name[0] == 'f' ? 0 : 1];
    }
    else return FCD_ERR_ARG;
    return FCD_OK;
}

int64_t fcd_N_to_C(int64_t Nreg) { return fcd_tri(Nreg); }

int64_t fcd_C_to_N(int64_t C) {
    if (C < 1) return FCD_ERR_SHAPE;
    int64_t n = (int64_t)((sqrt(8.0 * (double)C + 1.0) - 1.0) * 0.5) + 1;
    while (fcd_tri(n) > C) --n;
    while (fcd_tri(n + 1) <= C) ++n;
    return fcd_tri(n) == C ? n : (int64_t)FCD_ERR_SHAPE;
}

int64_t fcd_nm_to_c(int64_t n, int64_t m) { return fcd_tri(n) + m; }

int fcd_c_to_nm(int64_t c, int64_t *n, int64_t *m) {
    if (c < 0 || !n || !m) return FCD_ERR_ARG;
    int nn, mm;
    fcd_edge_to_pair(c, nn, mm);
    *n = nn;
    *m = mm;
    return FCD_OK;
}

}  // extern "C"
