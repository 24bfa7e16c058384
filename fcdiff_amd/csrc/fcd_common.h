// Shared host/device helpers of libfcdiff_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/fcdiff_hip.h"

#define FCD_PROF_SLOTS 4
#define FCD_PROF_LIK 0
#define FCD_PROF_F 1
#define FCD_PROF_RSTEP 2
#define FCD_PROF_PACK 3

// Tuning / test knobs.  Read ONCE from the environment by fcd_ctx_create (FCD_R_PATH, FCD_R_UB, FCD_R_NOPAD, FCD_R_TOL,
// FCD_F_TOL, FCD_F_FORM, FCD_R_POLL_LIMIT, FCD_R_WITHHOLD: "0" / unset = default), changed afterwards only through
// fcd_ctx_set_knob: no entry point reads the environment.
struct fcd_knobs {
    int r_path;        // 0: blocked r pass -- pipelined one-launch form where its grid fits the device at once, else one launch
                       // per block step; 3: one launch per block step always
    int r_ub;          // patients per panel workgroup of the blocked r pass: 0 = automatic, else 1 / 2 / 4
    int r_nopad;       // 1: no empty workgroups beside the in-order workgroups
    int r_refill;      // 1: the packing launch writes the panel-value sentinels in every sweep (default: only in the first sweep of a fcd_gibbs_run call)
    int qr_form;       // variational q_R update: 0 = by size; 1 = gathers from the edge-major table inside the region loop (rounds 1-3);
                       // 2 = region-major weights made first, whatever their size
    int r_coop;        // pipelined r pass: 1 = cooperative launch (the runtime checks that the grid is co-resident; +16 us per pass), else plain
    int r_dsplit;      // 1: ONE in-order workgroup per patient in the pipelined r pass (default: two, 8 chain words each, where there are more than 8)
    double r_tol;      // > default: widen the margin inside which an r draw is re-decided with the exact logit
    double f_tol;      // > default: the same for the f draws
    int r_poll_limit;  // TEST HOOK: > 0 bounds every device-side poll of the pipelined r pass by this many polls (default 2^20, ~1 s)
    int r_withhold;    // TEST HOOK: 1 = the in-order role of the pipelined r pass never sets its marks (a panel wave then gives up)
    int corr_form;     // 1: K_corr in 64 x 64 blocks with a moments pass also where the one-workgroup-per-subject kernel would run
    int f_form;        // 0: automatic; 2: the any-U pair kernel also where the U <= 64 one would run; 3: scalar-mask form
};

// kernels whose dynamic-LDS limit is raised with hipFuncSetAttribute: done once per (kernel, size) and remembered here
enum { FCD_KA_F_GENERIC = 0, FCD_KA_F_COND, FCD_KA_F_DIFF, FCD_KA_F_PAIR, FCD_KA_F_PAIR_BIG = FCD_KA_F_PAIR + 4,
       FCD_KA_R_STEP = FCD_KA_F_PAIR_BIG + 4, FCD_KA_R_PIPE = FCD_KA_R_STEP + 4, FCD_KA_CORR = FCD_KA_R_PIPE + 4, FCD_KA_N = FCD_KA_CORR + 1 };

struct fcd_ctx {
    int device;
    int num_cu;
    void *ws;          // reduction / partial-sum workspace
    size_t ws_bytes;
    fcd_knobs knobs;
    long long n_alloc;             // device allocations made by this context so far (fcd_ctx_stat "n_alloc")
    size_t lds_attr[FCD_KA_N];     // largest dynamic-LDS size already set per kernel
    int pipe_occ[3];               // pipelined r pass: workgroups per CU of the three kernel variants (-1: not asked yet) ...
    size_t pipe_occ_shmem[3];      // ... for this much dynamic LDS
    int pipe_occ_threads[3];       // ... and this many threads
    int r_form_last;               // form of the last blocked r pass: 1 step-per-launch, 2 pipelined, 3 one-launch with counters (fcd_ctx_stat)
    void *log_tab;     // K_lik tables (fcd_fastmath.h): 64 x 2^(-j/64), 512 x {1/m_i, log m_i} (device, 8.5 KiB)
    volatile unsigned *dev_err;   // pinned host word: error word of the one-launch r pass, copied back after each pass
    void *acc;         // 8 x uint64, zero between launches: the tally's pooled sums [0..3] and its ticket [4]
    void *comm;        // ncclComm_t of the context (fcd_comm_init), or nullptr: fcd_gibbs_run pools its M-step counts over it
    int comm_world, comm_rank;
    void *pool_counts; // 8 x int64 (device): the counts vector the all-reduce works on in place
    void *dbg;         // 8 x uint64 event counters (fcd_ctx_stat): [0] waves of the f pass that repeated an edge's sums in fp64,
                       // [1] rows of the r pass's in-order role decided on the exact path; never reset by the library
    void *corr_tickets;            // K_corr: one ticket per subject, zero between launches (the last taker resets it)
    size_t corr_tickets_n;
    void *fsq;         // square copy of the f state [w][n][m][lane] kept by fcd_gibbs_sweeps between its f and r pass
    size_t fsq_bytes;
    // optional per-kernel timing with HIP events on the launch stream (fcd_prof_enable / fcd_prof_collect)
    int prof_on;
    hipEvent_t *prof_ev[FCD_PROF_SLOTS];   // pairs (begin, end)
    int prof_n[FCD_PROF_SLOTS];            // pairs recorded
    int prof_cap[FCD_PROF_SLOTS];
    char msg[256];
};

#define FCD_HIP_TRY(expr)                       \
    do {                                        \
        hipError_t e__ = (expr);                \
        if (e__ != hipSuccess) return (int)e__; \
    } while (0)

#define FCD_LAUNCH_CHECK()                      \
    do {                                        \
        hipError_t e__ = hipGetLastError();     \
        if (e__ != hipSuccess) return (int)e__; \
    } while (0)

static inline int fcd_fail(fcd_ctx *ctx, int code, const char *fmt, long long a = 0, long long b = 0) {
    if (ctx) snprintf(ctx->msg, sizeof(ctx->msg), fmt, a, b);
    return code;
}

// A kernel whose table addresses assume that its dynamic LDS array starts at address 0 must declare no static LDS:
// asked of the runtime once per kernel (*done)
static inline int fcd_static_lds_check(fcd_ctx *ctx, const void *fn, int *done) {
    if (*done) return FCD_OK;
    hipFuncAttributes fa;
    hipError_t e = hipFuncGetAttributes(&fa, fn);
    if (e != hipSuccess) return (int)e;
    if (fa.sharedSizeBytes != 0) {
        return fcd_fail(ctx, FCD_ERR_UNSUPPORTED, "kernel carries %lld bytes of static LDS: its table addresses assume none",
                        (long long)fa.sharedSizeBytes);
    }
    *done = 1;
    return FCD_OK;
}

int fcd_comm_allreduce_counts(fcd_ctx *ctx, long long *counts, hipStream_t stream);     // fcd_comm.hip: no-op without a communicator
int fcd_ws_reserve(fcd_ctx *ctx, size_t bytes);
// square copy of the f state (see fcd_gibbs_sweeps): grown like the workspace
int fcd_fsq_reserve(fcd_ctx *ctx, size_t bytes);
// bytes the sweep kernels need at this shape: f pass scratch, r pass scratch (both in ctx->ws, one after the other in
// time: the larger counts), square f copy.  ONE formula shared by fcd_ctx_reserve and the step functions.
void fcd_sweep_ws_bytes(const fcd_ctx *ctx, int64_t Nreg, int64_t U, int64_t GW, size_t *ws_bytes, size_t *fsq_bytes);
size_t fcd_f_pass_ws_bytes(int64_t Nreg, int64_t U, int64_t GW);     // fcd_gibbs.hip
size_t fcd_r_pass_ws_bytes(int64_t Nreg, int64_t U, int64_t GW, int r_path);   // fcd_gibbs_r.hip
size_t fcd_fsq_need_bytes(int64_t Nreg, int64_t U, int64_t GW);      // fcd_gibbs.hip: 0 when the fused driver keeps no square copy
// raise a kernel's dynamic-LDS limit if this size was not set before (no HIP call otherwise)
static inline int fcd_lds_attr(fcd_ctx *ctx, int slot, const void *fn, size_t shmem) {
    if (shmem <= 64 * 1024 || shmem <= ctx->lds_attr[slot]) return FCD_OK;
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    if (e != hipSuccess) return (int)e;
    ctx->lds_attr[slot] = shmem;
    return FCD_OK;
}
// f / r pass with the square copy of the f state (fcd_gibbs_sweeps: the f pass also writes f of edge (n, m) at [n][m] and
// [m][n], the r pass then packs its f words from contiguous rows instead of gathering 64-byte pieces).  fsq == nullptr:
// the plain entry points.  Symmetric edge ids only.
int fcd_gibbs_f_step_sq(fcd_ctx *ctx, const double *S_B, const double *lM, const double *lMf, const double *hyper,
                        uint8_t *f_state, const uint64_t *r_bits, int64_t Nreg, int64_t U, int64_t G, int64_t chain0,
                        uint64_t seed, int64_t sweep, hipStream_t stream, uint8_t *fsq, bool ru_ready, size_t ru_off = 0);
// The f half of the tally (pooled counts of f, marginal counters of the edges): it needs nothing of the r pass, so the r
// pass's packing launch can carry it in extra workgroups of its own instead of the tally launch after the pass.
struct fcd_tally_f {
    const uint8_t *f_state;
    int64_t C, G;
    int GW;
    unsigned long long *acc;           // nullable: context-owned sums, [1..3] = number of f == 0, 1, 2
    uint32_t *cnt_f;                   // nullable
};
// tally_f != nullptr: asked to carry the f half; *tally_f_done says whether it did (the blocked path with a packing launch).
// sentinels_in_place: the two panel-value buffers at the head of the workspace still hold the sentinels a COMPLETED pipelined
// pass of the same shape left there (every slot gets its sentinel back when its value is consumed): the packing launch
// need not write them again.  ru_off (f pass): where in the workspace the slot words live.
int fcd_gibbs_r_step_sq(fcd_ctx *ctx, const double *lM, const double *lMd, const double *hyper,
                        const uint8_t *f_state, uint64_t *r_bits, int64_t Nreg, int64_t U, int64_t G, int64_t chain0,
                        uint64_t seed, int64_t sweep, int edge_mode, hipStream_t stream, const uint8_t *fsq,
                        const fcd_tally_f *tally_f = nullptr, bool *tally_f_done = nullptr, bool sentinels_in_place = false);
// bracket ONE kernel launch with events when profiling is on (no-ops otherwise)
void fcd_prof_begin(fcd_ctx *ctx, int slot, hipStream_t s);
void fcd_prof_end(fcd_ctx *ctx, int slot, hipStream_t s);

// hyper block offsets
#define FCD_H_LNGAMMA 0
#define FCD_H_LNPI0 3
#define FCD_H_LNPI1 4

// RNG kinds (4th counter word)
#define FCD_KIND_INIT_F 0u
#define FCD_KIND_INIT_R 1u
#define FCD_KIND_F 2u
#define FCD_KIND_R 3u

// ---------------------------------------------------------------------------------------------
// edge <-> region-pair maps, fcdiff/util.py:40-84.  c = n(n-1)/2 + m, n > m.
// ---------------------------------------------------------------------------------------------
__host__ __device__ static inline int64_t fcd_tri(int64_t n) { return n * (n - 1) / 2; }

// Inverse with an integer correction step (the reference's float sqrt formula, util.py:82, is only
// trusted near perfect squares up to the sizes it was written for).
__host__ __device__ static inline void fcd_edge_to_pair(int64_t c, int &n, int &m) {
    int64_t nn = (int64_t)((sqrt(8.0 * (double)c + 1.0) - 1.0) * 0.5) + 1;
    while (fcd_tri(nn) > c) --nn;
    while (fcd_tri(nn + 1) <= c) ++nn;
    n = (int)nn;
    m = (int)(c - fcd_tri(nn));
}

// Edge id the region update uses for the ORDERED pair (n, m), m != n.
__host__ __device__ static inline int64_t fcd_pair_to_edge(int n, int m, int mode) {
    if (mode == FCD_EDGE_REFERENCE || n > m) return fcd_tri(n) + m;  // fit.py:186 / util.py:60
    return fcd_tri(m) + n;
}

// ---------------------------------------------------------------------------------------------
// Philox4x32-10 (Random123 constants).  counter = (idx, chain, sweep, kind), key = seed.
// ---------------------------------------------------------------------------------------------
struct fcd_u4 {
    uint32_t x, y, z, w;
};

__host__ __device__ static inline fcd_u4 fcd_philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                    uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return fcd_u4{c0, c1, c2, c3};
}

// 53 high bits of (hi:lo) as a double in [0, 1).
__host__ __device__ static inline double fcd_u53(uint32_t hi, uint32_t lo) {
    uint64_t w = ((uint64_t)hi << 32) | lo;
    return (double)(w >> 11) * (1.0 / 9007199254740992.0);
}

// One 32-bit word as a double in [0, 1): what the 3-way f draws of the sweeps use (four edges per counter block).
__host__ __device__ static inline double fcd_u32(uint32_t w) { return (double)w * (1.0 / 4294967296.0); }
__host__ __device__ static inline uint32_t fcd_word(const fcd_u4 &x, int i) { return i == 0 ? x.x : i == 1 ? x.y : i == 2 ? x.z : x.w; }

__host__ __device__ static inline double fcd_site_uniform(uint64_t seed, uint32_t idx, uint32_t chain,
                                                          uint32_t sweep, uint32_t kind, int half) {
    fcd_u4 x = fcd_philox(idx, chain, sweep, kind, (uint32_t)seed, (uint32_t)(seed >> 32));
    return half ? fcd_u53(x.z, x.w) : fcd_u53(x.x, x.y);
}

// ---------------------------------------------------------------------------------------------
// draws: identical formulas in oracle/fcdiff_oracle.py (draw_f, draw_r) and oracle/fcdiff_oracle.c
// ---------------------------------------------------------------------------------------------
__device__ static inline int fcd_draw_f(double a0, double a1, double a2, double x) {
    double mx = fmax(a0, fmax(a1, a2));
    double e0 = exp(a0 - mx), e1 = exp(a1 - mx), e2 = exp(a2 - mx);
    double t = x * ((e0 + e1) + e2);
    return (t < e0) ? 0 : ((t < e0 + e1) ? 1 : 2);
}

// The same draw from three hardware fp32 exponentials (~20 instructions instead of ~150), for log-odds b1, b2 against type 0
// that are themselves only known to within +-delta (fp32 accumulation of the f pass; 0 for fp64 sums).  *amb is set when the
// outcome could be another one in exact arithmetic: the caller then repeats the draw with fcd_draw_f on fp64 sums, so the
// outcome is always fcd_draw_f's.  After the shift by the largest exponent ONE weight is exactly 1; the other two carry a
// relative uncertainty eta = e^(2 delta) - 1 (both b's off in opposite directions) + 2e-5 (fp32 exp: argument rounding
// times |argument| <= 87, then 1 ulp), so x * sum and either boundary move by at most eta * (sum - 1) each -- an
// uncertainty RELATIVE TO THE WEIGHT OFF THE MODE, not to the sum: a conditional that is all but one-hot (the usual case)
// never repeats on account of it.  `floor_rel` * sum covers the fp32 roundings of x, the sum and the product (default
// FCD_DRAW_F_MARGIN; the test hook f_tol raises it: 1e30 sends every draw down the exact path).  NaN-safe: a NaN or
// infinite sum or bound is ambiguous.
#define FCD_DRAW_F_MARGIN 1e-6f
__host__ __device__ static inline float fcd_draw_f_eta(float delta) { return expm1f(2.0f * delta) * 1.001f + 2e-5f; }
__device__ static inline int fcd_draw_f_fast(double b1, double b2, double x, float eta, float floor_rel, bool *amb) {
    const double mx = fmax(0.0, fmax(b1, b2));
    const float e0 = __expf((float)(0.0 - mx)), e1 = __expf((float)(b1 - mx)), e2 = __expf((float)(b2 - mx));
    const float s = (e0 + e1) + e2;
    const float t = (float)x * s;
    const float m = 2.0f * eta * (s - 1.0f) + floor_rel * s;
    *amb = !(fabsf(t - e0) >= m && fabsf(t - (e0 + e1)) >= m);
    return (t < e0) ? 0 : ((t < e0 + e1) ? 1 : 2);
}
// The draw entirely in fp32 (the pair forms of the f pass: their sums are fp32 already): bf1, bf2 = log-odds against type 0
// known to within +-delta, eta = fcd_draw_f_eta(delta), xf = the uniform rounded to fp32 (covered by floor_rel).  Same rule,
// same outcomes.
__device__ static inline int fcd_draw_f_fast32(float bf1, float bf2, float xf, float eta, float floor_rel, bool *amb) {
    const float mx = fmaxf(0.f, fmaxf(bf1, bf2));
    const float e0 = __expf(0.f - mx), e1 = __expf(bf1 - mx), e2 = __expf(bf2 - mx);
    const float s = (e0 + e1) + e2;
    const float t = xf * s;
    const float m = 2.0f * eta * (s - 1.0f) + floor_rel * s;
    *amb = !(fabsf(t - e0) >= m && fabsf(t - (e0 + e1)) >= m);
    return (t < e0) ? 0 : ((t < e0 + e1) ? 1 : 2);
}
// The draw without an exponential, where it is certain: if the largest of (0, b1, b2) leads the second by more than 15 + eta
// (eta >= twice the sums' error bound), the two other weights together are below 2 e^-15 = 6.1e-7 of the largest, and any x in
// [1e-6, 1 - 1e-6] falls into the largest one's interval of the inverse CDF whatever the order of the three: k = argmax,
// exactly what fcd_draw_f returns.  Returns false where that cannot be said (the caller then takes fcd_draw_f_fast32).
// On well separated data nearly every draw is of this kind; on weak data the test costs a dozen instructions.
__device__ static inline bool fcd_draw_f_sure(float bf1, float bf2, float xf, float eta, int *k) {
    const float mx = fmaxf(0.f, fmaxf(bf1, bf2)), mn = fminf(0.f, fminf(bf1, bf2));
    const float mid = ((bf1 + bf2) - mx) - mn;                        // (the middle one of 0, b1, b2, to within rounding: covered by the 1e-2)
    *k = (bf2 >= mx) ? 2 : ((bf1 >= mx) ? 1 : 0);
    return (mx - mid) - eta > 15.01f && xf > 1e-6f && xf < 0.999999f;
}
// absolute error bound of an fp32 sum of n terms, each first rounded to fp32, given B >= sum of |terms|: the n conversions
// cost at most 2^-24 B together, each of the n - 1 additions at most 2^-24 times a partial sum of magnitude <= B
__host__ __device__ static inline float fcd_f32_sum_err(int n_terms, float B) { return (float)(n_terms + 1) * 5.97e-8f * 1.01f * B; }
// ... and of  (float)c + that sum  for an offset |c| <= cm: the offset's conversion and one more addition
__host__ __device__ static inline float fcd_f32_offset_err(float cm, float B) { return 2.0f * 5.97e-8f * 1.01f * (cm + B); }

// r = 1 with probability sigmoid(s1 - s0):  x < 1/(1+exp(s0-s1))  <=>  logit(x) < s1 - s0.
// The threshold depends on the random number only, so it is computed off the region-to-region chain.
__device__ static inline double fcd_logit(double x) { return log(x / (1.0 - x)); }
__device__ static inline int fcd_draw_r(double s0, double s1, double x) { return fcd_logit(x) < (s1 - s0) ? 1 : 0; }
// logit(x) to within FCD_LOGIT_FAST_ERR (absolute) from two hardware fp32 logarithms (8 instructions instead of a
// double-precision division and logarithm).  A draw decided with it is re-decided with fcd_logit whenever the
// compared quantity lies within the tolerance, so the outcome is always that of fcd_logit.
#define FCD_LOGIT_FAST_ERR 2e-5
__device__ static inline double fcd_logit_fast(double x) {
    const float a = __log2f((float)x), b = __log2f((float)(1.0 - x));   // x = 0 -> -inf, like fcd_logit
    return (double)((a - b) * 0.69314718f);
}

// ---------------------------------------------------------------------------------------------
// wave64 helpers
// ---------------------------------------------------------------------------------------------
__device__ static inline double fcd_wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// per-lane select driven by a wave-uniform 64-bit mask held in an SGPR pair:
// lane i gets (mask bit i) ? b : a.  One v_cndmask_b32, no per-lane shift.
__device__ static inline uint32_t fcd_sel_mask(uint32_t a, uint32_t b, uint64_t mask) {
    uint32_t out;
    asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(out) : "v"(a), "v"(b), "s"(mask));
    return out;
}

// mask of the chains of word w that exist (the last word of G chains may be partial)
__host__ __device__ static inline uint64_t fcd_active_mask(int w, int64_t G) {
    const int64_t rem = G - (int64_t)w * 64;
    return rem >= 64 ? ~0ull : ((1ull << rem) - 1ull);
}

#ifdef __HIPCC__
// The f half of the tally for workgroup lin of nblk (WAVES waves each): f_state is read once, 16 bytes per lane -- a wave
// covers the 16 chain words x 64 chains of an edge with one load instruction; four edges per round, their loads issued
// together (the pass is a few memory round trips long: what counts is the number of bytes in flight).  red: LDS,
// [WAVES][3].  Integer sums: any order gives the same totals.
template <int WAVES>
__device__ __forceinline__ void fcd_tally_f_block(const fcd_tally_f &a, int lin, int nblk, unsigned long long (*red)[3]) {
    const uint8_t *__restrict__ f_state = a.f_state;
    uint32_t *__restrict__ cnt_f = a.cnt_f;
    const int64_t C = a.C, G = a.G;
    const int GW = a.GW;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane & 3, wrow = lane >> 2;          // 16-byte piece of the 64-byte row, chain word within a group of 16
    unsigned long long tot1 = 0, tot2 = 0, n_edges = 0;
    constexpr int TE = 4;
    for (int64_t c0 = ((int64_t)lin * WAVES + wave) * TE; c0 < C; c0 += (int64_t)nblk * WAVES * TE) {
        for (int wg = 0; wg < GW; wg += 16) {
            const int w = wg + wrow;
            uint4 vv[TE];
#pragma unroll
            for (int t = 0; t < TE; ++t) {
                const int64_t c = (c0 + t < C) ? c0 + t : C - 1;
                vv[t] = *reinterpret_cast<const uint4 *>(f_state + ((int64_t)(w < GW ? w : 0) * C + c) * 64 + sub * 16);
            }
            const uint32_t act = (w < GW) ? (uint32_t)(fcd_active_mask(w, G) >> (sub * 16)) & 0xFFFFu : 0u;
#pragma unroll
            for (int t = 0; t < TE; ++t) {
                uint4 v = vv[t];
                if (act != 0xFFFFu) {   // partial (or absent) chain word: drop the bytes of chains that do not exist
                    v.x &= (((act & 15u) * 0x00204081u) & 0x01010101u) * 0xFFu;
                    v.y &= ((((act >> 4) & 15u) * 0x00204081u) & 0x01010101u) * 0xFFu;
                    v.z &= ((((act >> 8) & 15u) * 0x00204081u) & 0x01010101u) * 0xFFu;
                    v.w &= ((((act >> 12) & 15u) * 0x00204081u) & 0x01010101u) * 0xFFu;
                }
                uint32_t ones = __popc(v.x & 0x01010101u) + __popc(v.y & 0x01010101u) + __popc(v.z & 0x01010101u) +
                                __popc(v.w & 0x01010101u);
                uint32_t twos = __popc((v.x >> 1) & 0x01010101u) + __popc((v.y >> 1) & 0x01010101u) +
                                __popc((v.z >> 1) & 0x01010101u) + __popc((v.w >> 1) & 0x01010101u);
                for (int o = 32; o > 0; o >>= 1) {
                    ones += __shfl_xor(ones, o, 64);
                    twos += __shfl_xor(twos, o, 64);
                }
                if (lane == 0 && c0 + t < C) {
                    const int64_t c = c0 + t;
                    if (cnt_f) {
                        // (atomics because they do not wait for the old value to come back)
                        atomicAdd(&cnt_f[c * 3 + 1], ones);
                        atomicAdd(&cnt_f[c * 3 + 2], twos);
                        if (wg == 0) atomicAdd(&cnt_f[c * 3 + 0], (uint32_t)G);
                        atomicAdd(&cnt_f[c * 3 + 0], 0u - ones - twos);
                    }
                    tot1 += ones;
                    tot2 += twos;
                    if (wg == 0) n_edges += 1;
                }
            }
        }
    }
    if (!a.acc) return;
    // one set of atomics per workgroup (same-address atomics serialise): wave sums -> LDS -> threads 0..2
    if (lane == 0) {
        red[wave][0] = n_edges * (unsigned long long)G - tot1 - tot2;
        red[wave][1] = tot1;
        red[wave][2] = tot2;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        unsigned long long t = 0;
        for (int q = 0; q < WAVES; ++q) t += red[q][threadIdx.x];
        if (t) {
            // with the old value asked for, the add has been performed at the memory side once it returns (every access
            // to acc[] is a device-scope atomic): whoever takes a ticket after this workgroup's barrier finds it there
            const unsigned long long old = atomicAdd(&a.acc[1 + threadIdx.x], t);
            asm volatile("" ::"v"(old));
        }
    }
    __syncthreads();
}
#endif

// ---------------------------------------------------------------------------------------------
// host: shape checks shared by the sampler entry points
// ---------------------------------------------------------------------------------------------
struct fcd_geo {
    int64_t C;
    int GW;
};

static inline int fcd_geo_check(fcd_ctx *ctx, int64_t Nreg, int64_t U, int64_t G, int64_t chain0, fcd_geo &g) {
    if (!ctx) return FCD_ERR_ARG;
    if (Nreg < 2 || U < 1 || G < 1)
        return fcd_fail(ctx, FCD_ERR_SHAPE, "need Nreg >= 2, U >= 1, G >= 1 (Nreg=%lld, U=%lld)", Nreg, U);
    if (Nreg > 46340 || U > (1 << 20) || G > (1ll << 31) || chain0 < 0 || chain0 + G > (1ll << 32))
        return fcd_fail(ctx, FCD_ERR_UNSUPPORTED, "shape out of range (Nreg=%lld, G=%lld)", Nreg, G);
    g.C = fcd_tri(Nreg);
    g.GW = (int)((G + 63) / 64);
    if (g.GW > 65535) return fcd_fail(ctx, FCD_ERR_UNSUPPORTED, "G=%lld exceeds 65535 chain words per launch", G);
    if ((Nreg + 1) / 2 * U > (1ll << 32) || (g.C + 1) / 2 > (1ll << 32))
        return fcd_fail(ctx, FCD_ERR_UNSUPPORTED, "site index exceeds the 32-bit counter word");
    return FCD_OK;
}

// ---------------------------------------------------------------------------------------------
// Ablation build (make ABLATE=1 -> libfcdiff_hip_abl.so, NEVER the product library): kernels read a level
// from device memory and skip parts of their work so that phases can be timed A/B in one process.
// Results are wrong whenever the level is non-zero.  In the product build FCD_ABL(x) folds to false.
// ---------------------------------------------------------------------------------------------
#ifdef FCD_ABLATE
extern __device__ int fcd_abl_level[4];   // [0] f kernel, [1] panel, [2] diag, [3] row stamps of the pipelined scan
#ifdef FCD_TRACE_ONLY     // (make ABLATE=1 TRACE_ONLY=1: the time stamps without the ablation switches -- the product's code otherwise)
#define FCD_ABL(slot, lvl) false
#else
#define FCD_ABL(slot, lvl) (fcd_abl_level[slot] >= (lvl))
#endif
void fcd_abl_refresh(hipStream_t s);
// Timeline of the r step kernel: record (launch, workgroup) x 8 words of the 100 MHz clock, written by thread 0
// when FCD_TRACE_PTR names a device buffer (profiles/trace_r.py).
extern __device__ unsigned long long *fcd_trace_buf;
#define FCD_TRACE(rec, slot)                                                                      \
    do {                                                                                          \
        if (fcd_trace_buf && threadIdx.x == 0) fcd_trace_buf[(size_t)(rec) * 8 + (slot)] = wall_clock64(); \
    } while (0)
#define FCD_TRACE_VAL(rec, slot, v)                                                               \
    do {                                                                                          \
        if (fcd_trace_buf && threadIdx.x == 0) fcd_trace_buf[(size_t)(rec) * 8 + (slot)] = (unsigned long long)(v); \
    } while (0)
// the same from thread t0 of the workgroup
#define FCD_TRACE_AT(t0, rec, slot)                                                               \
    do {                                                                                          \
        if (fcd_trace_buf && threadIdx.x == (t0)) fcd_trace_buf[(size_t)(rec) * 8 + (slot)] = wall_clock64(); \
    } while (0)
#define FCD_TRACE_VAL_AT(t0, rec, slot, v)                                                        \
    do {                                                                                          \
        if (fcd_trace_buf && threadIdx.x == (t0)) fcd_trace_buf[(size_t)(rec) * 8 + (slot)] = (unsigned long long)(v); \
    } while (0)
#else
#define FCD_ABL(slot, lvl) false
#define FCD_TRACE(rec, slot) do { } while (0)
#define FCD_TRACE_VAL(rec, slot, v) do { } while (0)
#define FCD_TRACE_AT(t0, rec, slot) do { } while (0)
#define FCD_TRACE_VAL_AT(t0, rec, slot, v) do { } while (0)
static inline void fcd_abl_refresh(hipStream_t) {}
#endif
