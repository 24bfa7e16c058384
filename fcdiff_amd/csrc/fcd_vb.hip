// Variational updates of UnsharedRegionFit (fcdiff/fit.py): _update_lq_F (157-174), _update_lq_R
// (176-198), _eval_energy and its six terms (142-155, 447-539), _update_pi/_update_gamma (208-220).
//
// All four read lM (C,U,3,3) once per call: HBM/L2-bound, algorithmic bytes 72*C*U + small.
#include "fcd_common.h"

namespace {

// scipy.special.logsumexp over 3 / 2 values: log(sum(exp(a - amax))) + amax (amax -> 0 if not finite)
__device__ inline double lse3(double a0, double a1, double a2) {
    double mx = fmax(a0, fmax(a1, a2));
    if (!isfinite(mx)) mx = 0.0;
    return log((exp(a0 - mx) + exp(a1 - mx)) + exp(a2 - mx)) + mx;
}
__device__ inline double lse2(double a0, double a1) {
    double mx = fmax(a0, a1);
    if (!isfinite(mx)) mx = 0.0;
    return log(exp(a0 - mx) + exp(a1 - mx)) + mx;
}

// sum_u sum_l w_l(u) * lM[c,u,k,l] for k = 0..2, one wave per edge, lanes over patients.
// w = (q0n q0m, q1n q1m, q0n q1m + q1n q0m): _eval_q_R_w, fit.py:382-406.
__device__ inline void edge_weighted_sums(const double *__restrict__ lq_R, const double *__restrict__ lM,
                                          int64_t c, int U, int lane, double out[3]) {
    int n, m;
    fcd_edge_to_pair(c, n, m);
    double t0 = 0.0, t1 = 0.0, t2 = 0.0;
    for (int u = lane; u < U; u += 64) {
        const double q0n = exp(lq_R[((int64_t)n * U + u) * 2 + 0]);
        const double q1n = exp(lq_R[((int64_t)n * U + u) * 2 + 1]);
        const double q0m = exp(lq_R[((int64_t)m * U + u) * 2 + 0]);
        const double q1m = exp(lq_R[((int64_t)m * U + u) * 2 + 1]);
        const double w0 = q0n * q0m;
        const double w1 = q1n * q1m;
        double w2 = q0n * q1m;
        w2 += q1n * q0m;
        const double *p = lM + (c * U + u) * 9;
        t0 += (w0 * p[0] + w1 * p[1]) + w2 * p[2];
        t1 += (w0 * p[3] + w1 * p[4]) + w2 * p[5];
        t2 += (w0 * p[6] + w1 * p[7]) + w2 * p[8];
    }
    out[0] = fcd_wave_sum(t0);
    out[1] = fcd_wave_sum(t1);
    out[2] = fcd_wave_sum(t2);
}

__global__ __launch_bounds__(256) void vb_qF_kernel(const double *__restrict__ lq_R, const double *__restrict__ S_B,
                                                    const double *__restrict__ lM, const double *__restrict__ hyper,
                                                    int64_t C, int U, double *__restrict__ lq_F) {
    const int lane = threadIdx.x & 63;
    const int64_t c = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= C) return;
    double t[3];
    edge_weighted_sums(lq_R, lM, c, U, lane, t);
    if (lane == 0) {
        // lq_F[c,:,k] = ln gamma_k + (sum_h lpB + sum lM)        fit.py:165, 171-173
        const double a0 = hyper[FCD_H_LNGAMMA + 0] + (S_B[c * 3 + 0] + t[0]);
        const double a1 = hyper[FCD_H_LNGAMMA + 1] + (S_B[c * 3 + 1] + t[1]);
        const double a2 = hyper[FCD_H_LNGAMMA + 2] + (S_B[c * 3 + 2] + t[2]);
        const double z = lse3(a0, a1, a2);                        // fit.py:174
        lq_F[c * 3 + 0] = a0 - z;
        lq_F[c * 3 + 1] = a1 - z;
        lq_F[c * 3 + 2] = a2 - z;
    }
}

// One workgroup per patient u (patients are independent given q_F); regions strictly in order with
// q_R[n] refreshed before region n+1 (Gauss-Seidel, fit.py:184-197).  Threads split the m-sum.
constexpr int QR_BLOCK = 256;
// q_F = exp(lq_F), once per call (fit.py:180 takes the exponential once, too)
__global__ __launch_bounds__(256) void vb_expF_kernel(const double *__restrict__ lq_F, int64_t n, double *__restrict__ qF) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) qF[i] = exp(lq_F[i]);
}

// One block per patient (patients are independent), regions strictly in order (Gauss-Seidel, fit.py:184-197), the
// threads split the sum over m.  The operands of region n+1 (its 9 table values and 3 q_F per thread and m) do not
// depend on what region n decides: they are requested before region n is reduced, so the scattered 72-byte table
// reads (one memory round trip, ~2 us) hide behind the reduction instead of adding up 200 times.
template <int MPT>   // regions m per thread: Nreg <= MPT * QR_BLOCK
__global__ __launch_bounds__(QR_BLOCK) void vb_qR_kernel(const double *__restrict__ qF, const double *__restrict__ lM,
                                                         const double *__restrict__ hyper, int Nreg, int U, int mode,
                                                         double *__restrict__ lq_R) {
    __shared__ double red[2][2 * (QR_BLOCK / 64)];      // wave partials, double-buffered: one barrier per region
    const int u = blockIdx.x;
    const int tid = threadIdx.x;
    // q of the thread's own regions m = tid + j * QR_BLOCK lives in registers; every thread finishes every region
    // itself (same arithmetic in all lanes), so nothing but the wave partials crosses threads
    double q0m[MPT], q1m[MPT];
#pragma unroll
    for (int j = 0; j < MPT; ++j) {
        const int m = tid + j * QR_BLOCK;
        q0m[j] = m < Nreg ? exp(lq_R[((int64_t)m * U + u) * 2 + 0]) : 0.0;
        q1m[j] = m < Nreg ? exp(lq_R[((int64_t)m * U + u) * 2 + 1]) : 0.0;
    }
    struct Op {
        double p[9], f[3];
        bool on;
    };
    auto load = [&](int n, Op (&o)[MPT]) {
#pragma unroll
        for (int j = 0; j < MPT; ++j) {
            const int m = tid + j * QR_BLOCK;
            o[j].on = m < Nreg && m != n;
            const int64_t c = o[j].on ? fcd_pair_to_edge(n, m, mode) : 0;
            const double *p = lM + (c * U + u) * 9;
#pragma unroll
            for (int x = 0; x < 9; ++x) o[j].p[x] = p[x];
#pragma unroll
            for (int k = 0; k < 3; ++k) o[j].f[k] = qF[c * 3 + k];
        }
    };
    // three operand buffers: region n is computed from one while region n+2 is on its way into another
    Op bufA[MPT], bufB[MPT], bufC[MPT];
    load(0, bufA);
    if (Nreg > 1) load(1, bufB);
    const double lnpi0 = hyper[FCD_H_LNPI0], lnpi1 = hyper[FCD_H_LNPI1];
    auto step = [&](int n, Op (&cur)[MPT], Op (&fill)[MPT]) {
        if (n + 2 < Nreg) load(n + 2, fill);
        double t0 = 0.0, t1 = 0.0;
#pragma unroll
        for (int j = 0; j < MPT; ++j) {
            if (!cur[j].on) continue;
            const double qm0 = q0m[j], qm1 = q1m[j];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const double qFk = cur[j].f[k];
                const double lM_00 = qm0 * cur[j].p[k * 3 + 0];
                const double lM_1neq = qm1 * cur[j].p[k * 3 + 2];
                t0 += qFk * (lM_00 + lM_1neq);         // fit.py:188-190
                const double lM_11 = qm1 * cur[j].p[k * 3 + 1];
                const double lM_0neq = qm0 * cur[j].p[k * 3 + 2];
                t1 += qFk * (lM_11 + lM_0neq);         // fit.py:192-194
            }
        }
        t0 = fcd_wave_sum(t0);
        t1 = fcd_wave_sum(t1);
        double *rb = red[n & 1];
        if ((tid & 63) == 0) {
            rb[(tid >> 6) * 2 + 0] = t0;
            rb[(tid >> 6) * 2 + 1] = t1;
        }
        __syncthreads();
        double s0 = lnpi0, s1 = lnpi1;
        for (int w = 0; w < QR_BLOCK / 64; ++w) {
            s0 += rb[w * 2 + 0];
            s1 += rb[w * 2 + 1];
        }
        const double z = lse2(s0, s1);                 // fit.py:196
        s0 -= z;
        s1 -= z;
        if (tid < 2) lq_R[((int64_t)n * U + u) * 2 + tid] = tid == 0 ? s0 : s1;
        const double e0 = exp(s0), e1 = exp(s1);       // fit.py:197
#pragma unroll
        for (int j = 0; j < MPT; ++j) {
            if (tid + j * QR_BLOCK == n) {
                q0m[j] = e0;
                q1m[j] = e1;
            }
        }
    };
    for (int n = 0; n < Nreg; n += 3) {
        step(n, bufA, bufC);
        if (n + 1 < Nreg) step(n + 1, bufB, bufA);
        if (n + 2 < Nreg) step(n + 2, bufC, bufB);
    }
}

// ---- energy: per-block partials (fixed order) then one block folds them --------------------
// partial layout: ws[block*8 + t], t = 0..5
constexpr int EN_BLOCK = 256;
__device__ inline double xlogy0(double q, double lq) { return q == 0.0 ? 0.0 : q * lq; }

__global__ __launch_bounds__(EN_BLOCK) void vb_energy_edges(const double *__restrict__ lq_F, const double *__restrict__ lq_R,
                                                            const double *__restrict__ S_B, const double *__restrict__ lM,
                                                            const double *__restrict__ hyper, int64_t C, int U,
                                                            double *__restrict__ ws) {
    __shared__ double acc[4][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double e_F = 0, e_B = 0, e_M = 0, e_qF = 0;
    for (int64_t c = (int64_t)blockIdx.x * 4 + wave; c < C; c += (int64_t)gridDim.x * 4) {
        double t[3];
        edge_weighted_sums(lq_R, lM, c, U, lane, t);
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const double lq = lq_F[c * 3 + k];
                const double q = exp(lq);
                e_F += q * hyper[FCD_H_LNGAMMA + k];   // fit.py:458
                e_B += q * S_B[c * 3 + k];             // fit.py:472, H-sum taken first
                e_M += q * t[k];                       // fit.py:509-510
                e_qF += xlogy0(q, lq);                 // fit.py:525
            }
        }
    }
    if (lane == 0) {
        acc[wave][0] = e_F; acc[wave][1] = e_B; acc[wave][2] = e_M; acc[wave][3] = e_qF;
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        const int j = threadIdx.x;
        ws[(int64_t)blockIdx.x * 8 + j] = ((acc[0][j] + acc[1][j]) + acc[2][j]) + acc[3][j];
    }
}

__global__ __launch_bounds__(256) void vb_energy_fold(const double *__restrict__ ws, int n_blocks,
                                                      const double *__restrict__ lq_R, const double *__restrict__ hyper,
                                                      int64_t NU, double *__restrict__ terms6) {
    __shared__ double red[4][6];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double v[6] = {0, 0, 0, 0, 0, 0};
    for (int bI = tid; bI < n_blocks; bI += 256) {
        v[0] += ws[(int64_t)bI * 8 + 0];
        v[1] += ws[(int64_t)bI * 8 + 1];
        v[3] += ws[(int64_t)bI * 8 + 2];
        v[4] += ws[(int64_t)bI * 8 + 3];
    }
    const double lnpi0 = hyper[FCD_H_LNPI0], lnpi1 = hyper[FCD_H_LNPI1];
    for (int64_t i = tid; i < NU; i += 256) {
        const double l0 = lq_R[i * 2 + 0], l1 = lq_R[i * 2 + 1];
        const double q0 = exp(l0), q1 = exp(l1);
        v[2] += q0 * lnpi0 + q1 * lnpi1;               // fit.py:486
        v[5] += xlogy0(q0, l0) + xlogy0(q1, l1);       // fit.py:539
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const double s = fcd_wave_sum(v[j]);
        if (lane == 0) red[wave][j] = s;
    }
    __syncthreads();
    if (tid < 6) terms6[tid] = ((red[0][tid] + red[1][tid]) + red[2][tid]) + red[3][tid];
}

// ---- theta step: pi* = mean q_R[:,:,1], gamma* = mean_c q_F (fit.py:208-220), one block ----------
__global__ __launch_bounds__(1024) void vb_theta_kernel(const double *__restrict__ lq_F, const double *__restrict__ lq_R,
                                                        int64_t C, int64_t NU, double *__restrict__ out4,
                                                        double *__restrict__ hyper) {
    __shared__ double red[16][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double v[4] = {0, 0, 0, 0};
    for (int64_t i = tid; i < NU; i += 1024) v[0] += exp(lq_R[i * 2 + 1]);
    for (int64_t c = tid; c < C; c += 1024) {
        v[1] += exp(lq_F[c * 3 + 0]);
        v[2] += exp(lq_F[c * 3 + 1]);
        v[3] += exp(lq_F[c * 3 + 2]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const double s = fcd_wave_sum(v[j]);
        if (lane == 0) red[wave][j] = s;
    }
    __syncthreads();
    if (tid < 4) {
        double s = 0.0;
        for (int w = 0; w < 16; ++w) s += red[w][tid];
        s /= (tid == 0) ? (double)NU : (double)C;
        out4[tid] = s;
        if (hyper) {
            if (tid == 0) {
                hyper[FCD_H_LNPI0] = log(1 - s);
                hyper[FCD_H_LNPI1] = log(s);
            } else {
                hyper[FCD_H_LNGAMMA + tid - 1] = log(s);
            }
        }
    }
}

__global__ void hyper_set_kernel(double *hyper, double g0, double g1, double g2, double p0, double p1) {
    hyper[0] = log(g0);
    hyper[1] = log(g1);
    hyper[2] = log(g2);
    hyper[3] = log(p0);
    hyper[4] = log(p1);
    hyper[5] = 0.0; hyper[6] = 0.0; hyper[7] = 0.0;
}

int check_shape(fcd_ctx *ctx, const char *fn, int64_t Nreg, int64_t U) {
    if (Nreg < 2 || U < 1) return fcd_fail(ctx, FCD_ERR_SHAPE, "need Nreg >= 2 and U >= 1 (Nreg=%lld, U=%lld)", Nreg, U);
    if (Nreg > 46340 || U > (1 << 24)) return fcd_fail(ctx, FCD_ERR_UNSUPPORTED, "Nreg=%lld / U=%lld too large", Nreg, U);
    (void)fn;
    return FCD_OK;
}

}  // namespace

extern "C" int fcd_hyper_set(fcd_ctx *ctx, double *hyper, const double *gamma3, const double *pi2, fcd_stream stream) {
    if (!ctx || !hyper || !gamma3 || !pi2) return fcd_fail(ctx, FCD_ERR_ARG, "fcd_hyper_set: null pointer");
    hipLaunchKernelGGL(hyper_set_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, hyper, gamma3[0], gamma3[1],
                       gamma3[2], pi2[0], pi2[1]);
    FCD_LAUNCH_CHECK();
    return FCD_OK;
}

extern "C" int fcd_vb_update_qF(fcd_ctx *ctx, const double *lq_R, const double *S_B, const double *lM,
                                const double *hyper, int64_t Nreg, int64_t U, double *lq_F, fcd_stream stream) {
    if (!ctx || !lq_R || !S_B || !lM || !hyper || !lq_F) return fcd_fail(ctx, FCD_ERR_ARG, "fcd_vb_update_qF: null pointer");
    int rc = check_shape(ctx, "fcd_vb_update_qF", Nreg, U);
    if (rc) return rc;
    const int64_t C = fcd_tri(Nreg);
    hipLaunchKernelGGL(vb_qF_kernel, dim3((unsigned)((C + 3) / 4)), dim3(256), 0, (hipStream_t)stream, lq_R, S_B, lM,
                       hyper, C, (int)U, lq_F);
    FCD_LAUNCH_CHECK();
    return FCD_OK;
}

extern "C" int fcd_vb_update_qR(fcd_ctx *ctx, const double *lq_F, const double *lM, const double *hyper,
                                int64_t Nreg, int64_t U, int edge_mode, double *lq_R, fcd_stream stream) {
    if (!ctx || !lq_F || !lM || !hyper || !lq_R) return fcd_fail(ctx, FCD_ERR_ARG, "fcd_vb_update_qR: null pointer");
    int rc = check_shape(ctx, "fcd_vb_update_qR", Nreg, U);
    if (rc) return rc;
    if (edge_mode != FCD_EDGE_REFERENCE && edge_mode != FCD_EDGE_SYMMETRIC)
        return fcd_fail(ctx, FCD_ERR_ARG, "fcd_vb_update_qR: edge_mode %lld", edge_mode);
    if (edge_mode == FCD_EDGE_REFERENCE && Nreg == 2)
        return fcd_fail(ctx, FCD_ERR_INDEX, "reference edge ids: index 1 is out of bounds for axis 0 with size 1 (Nreg=2)");
    const size_t shmem = 0;
    if (Nreg > 4 * QR_BLOCK)
        return fcd_fail(ctx, FCD_ERR_UNSUPPORTED, "fcd_vb_update_qR: Nreg=%lld exceeds 4 regions per thread", Nreg);
    const int64_t nF = fcd_tri(Nreg) * 3;
    rc = fcd_ws_reserve(ctx, (size_t)nF * sizeof(double));
    if (rc) return rc;
    double *qF = (double *)ctx->ws;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(vb_expF_kernel, dim3((unsigned)((nF + 255) / 256 < 1024 ? (nF + 255) / 256 : 1024)), dim3(256), 0, s, lq_F, nF, qF);
    FCD_LAUNCH_CHECK();
    const int mpt = (int)((Nreg + QR_BLOCK - 1) / QR_BLOCK);
    if (mpt <= 1)
        hipLaunchKernelGGL(vb_qR_kernel<1>, dim3((unsigned)U), dim3(QR_BLOCK), shmem, s, qF, lM, hyper, (int)Nreg, (int)U, edge_mode, lq_R);
    else if (mpt == 2)
        hipLaunchKernelGGL(vb_qR_kernel<2>, dim3((unsigned)U), dim3(QR_BLOCK), shmem, s, qF, lM, hyper, (int)Nreg, (int)U, edge_mode, lq_R);
    else
        hipLaunchKernelGGL(vb_qR_kernel<4>, dim3((unsigned)U), dim3(QR_BLOCK), shmem, s, qF, lM, hyper, (int)Nreg, (int)U, edge_mode, lq_R);
    FCD_LAUNCH_CHECK();
    return FCD_OK;
}

extern "C" int fcd_vb_energy(fcd_ctx *ctx, const double *lq_F, const double *lq_R, const double *S_B,
                             const double *lM, const double *hyper, int64_t Nreg, int64_t U, double *terms6,
                             fcd_stream stream) {
    if (!ctx || !lq_F || !lq_R || !S_B || !lM || !hyper || !terms6) return fcd_fail(ctx, FCD_ERR_ARG, "fcd_vb_energy: null pointer");
    int rc = check_shape(ctx, "fcd_vb_energy", Nreg, U);
    if (rc) return rc;
    const int64_t C = fcd_tri(Nreg);
    int64_t n_blocks = (C + 3) / 4;
    const int64_t cap = (int64_t)ctx->num_cu * 8;
    if (n_blocks > cap) n_blocks = cap;
    rc = fcd_ws_reserve(ctx, (size_t)n_blocks * 8 * sizeof(double));
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(vb_energy_edges, dim3((unsigned)n_blocks), dim3(EN_BLOCK), 0, s, lq_F, lq_R, S_B, lM, hyper, C,
                       (int)U, (double *)ctx->ws);
    FCD_LAUNCH_CHECK();
    hipLaunchKernelGGL(vb_energy_fold, dim3(1), dim3(256), 0, s, (const double *)ctx->ws, (int)n_blocks, lq_R, hyper,
                       Nreg * U, terms6);
    FCD_LAUNCH_CHECK();
    return FCD_OK;
}

extern "C" int fcd_vb_theta_step(fcd_ctx *ctx, const double *lq_F, const double *lq_R, int64_t Nreg, int64_t U,
                                 double *out4, double *hyper, fcd_stream stream) {
    if (!ctx || !lq_F || !lq_R || !out4) return fcd_fail(ctx, FCD_ERR_ARG, "fcd_vb_theta_step: null pointer");
    int rc = check_shape(ctx, "fcd_vb_theta_step", Nreg, U);
    if (rc) return rc;
    hipLaunchKernelGGL(vb_theta_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, lq_F, lq_R, fcd_tri(Nreg),
                       Nreg * U, out4, hyper);
    FCD_LAUNCH_CHECK();
    return FCD_OK;
}
