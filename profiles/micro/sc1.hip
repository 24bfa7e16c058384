// Agent-scope (sc1) loads against plain loads on the box: per-wave time of a 512-byte row load (one 8-byte element per
// lane), `depth` rows in flight, rows spread over a buffer larger than the L2s; one workgroup of 16 waves per CU on
// `ncu` CUs while the other CUs stream (or idle).  What the in-order scan of the pipelined r pass sees when it reads its
// panel values.
//   hipcc --offload-arch=gfx950 -O3 sc1.hip -o sc1 && ./sc1
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %d at %d\n", e, __LINE__); return 1; } } while (0)

// 16-byte elements (1 KB rows), plain loads only: is the cost per instruction or per byte?
template <int DEPTH>
__global__ __launch_bounds__(1024) void k16(const double2 *buf, size_t rows, int iters, double *out, long long *cyc) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    size_t r = ((size_t)blockIdx.x * 16 + wave) * 977u;
    double acc = 0.0;
    double2 v[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) v[d] = buf[((r + d * 131u) % rows) * 64 + lane];
    const long long t0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            acc += v[d].x + v[d].y;
            v[d] = buf[((r + (size_t)(it * DEPTH + d + DEPTH) * 131u) % rows) * 64 + lane];
        }
    }
    const long long t1 = wall_clock64();
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) acc += v[d].x;
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int DEPTH, bool SC1>
__global__ __launch_bounds__(1024) void k(const double *buf, size_t rows, int iters, double *out, long long *cyc) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    size_t r = ((size_t)blockIdx.x * 16 + wave) * 977u;
    double acc = 0.0;
    double v[DEPTH];
    auto ld = [&](size_t row) -> double {
        const double *p = buf + (row % rows) * 64 + lane;
        if (SC1) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return *p;
    };
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) v[d] = ld(r + d * 131u);
    const long long t0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            acc += v[d];                                     // use the oldest, ask for a new one
            v[d] = ld(r + (size_t)(it * DEPTH + d + DEPTH) * 131u);
        }
    }
    const long long t1 = wall_clock64();
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) acc += v[d];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

__global__ void stream_k(const double4 *a, double4 *b, size_t n, int reps) {
    for (int r = 0; r < reps; ++r)
        for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
            double4 x = a[i];
            x.x += 1.0;
            b[i] = x;
        }
}

template <int DEPTH, bool SC1>
int run(const double *buf, size_t rows, double *out, long long *cyc, int ncu, bool busy, const double4 *sa, double4 *sb, size_t sn) {
    const int iters = 2000 / DEPTH;
    hipStream_t s2;
    CHECK(hipStreamCreate(&s2));
    if (busy) hipLaunchKernelGGL(stream_k, dim3(2048), dim3(256), 0, s2, sa, sb, sn, 40);
    hipLaunchKernelGGL((k<DEPTH, SC1>), dim3(ncu), dim3(1024), 0, 0, buf, rows, iters, out, cyc);
    CHECK(hipDeviceSynchronize());
    long long h[256];
    CHECK(hipMemcpy(h, cyc, sizeof(long long) * ncu, hipMemcpyDeviceToHost));
    double m = 0;
    for (int i = 0; i < ncu; ++i) m += (double)h[i];
    m /= ncu;
    printf("%-5s depth %d  %3d CUs  %-22s  %.3f us per row and wave\n", SC1 ? "sc1" : "plain", DEPTH, ncu,
           busy ? "(other CUs streaming)" : "(idle device)", m / 100.0 / (iters * DEPTH));
    CHECK(hipStreamDestroy(s2));
    return 0;
}

template <int DEPTH>
int run16(const double *buf, size_t rows, double *out, long long *cyc, int ncu) {
    const int iters = 2000 / DEPTH;
    hipLaunchKernelGGL((k16<DEPTH>), dim3(ncu), dim3(1024), 0, 0, reinterpret_cast<const double2 *>(buf), rows / 2, iters, out, cyc);
    CHECK(hipDeviceSynchronize());
    long long h[256];
    CHECK(hipMemcpy(h, cyc, sizeof(long long) * ncu, hipMemcpyDeviceToHost));
    double m = 0;
    for (int i = 0; i < ncu; ++i) m += (double)h[i];
    m /= ncu;
    printf("plain 16-byte elements (1 KB rows) depth %d  %3d CUs  (idle device)  %.3f us per row and wave\n", DEPTH, ncu, m / 100.0 / (iters * DEPTH));
    return 0;
}

int main() {
    const size_t rows = (size_t)1 << 19;                     // 512 B each: 256 MB
    double *buf, *out;
    long long *cyc;
    double4 *sa, *sb;
    const size_t sn = (size_t)1 << 24;                       // 2 x 512 MB
    CHECK(hipMalloc(&buf, rows * 512));
    CHECK(hipMemset(buf, 0, rows * 512));
    CHECK(hipMalloc(&out, 256 * 1024 * 8));
    CHECK(hipMalloc(&cyc, 256 * 8));
    CHECK(hipMalloc(&sa, sn * 32));
    CHECK(hipMalloc(&sb, sn * 32));
    CHECK(hipMemset(sa, 0, sn * 32));
    for (int ncu : {1, 200}) {
        if (run16<1>(buf, rows, out, cyc, ncu)) return 1;
        if (run16<2>(buf, rows, out, cyc, ncu)) return 1;
        if (run16<4>(buf, rows, out, cyc, ncu)) return 1;
        if (run16<8>(buf, rows, out, cyc, ncu)) return 1;
        if (run<8, false>(buf, rows, out, cyc, ncu, false, sa, sb, sn)) return 1;
        if (run<4, false>(buf, rows, out, cyc, ncu, false, sa, sb, sn)) return 1;
    }
    for (int busy = 0; busy < 2; ++busy) {
        for (int ncu : {1, 50}) {
            if (run<1, false>(buf, rows, out, cyc, ncu, busy, sa, sb, sn)) return 1;
            if (run<1, true>(buf, rows, out, cyc, ncu, busy, sa, sb, sn)) return 1;
            if (run<2, false>(buf, rows, out, cyc, ncu, busy, sa, sb, sn)) return 1;
            if (run<2, true>(buf, rows, out, cyc, ncu, busy, sa, sb, sn)) return 1;
            if (run<4, false>(buf, rows, out, cyc, ncu, busy, sa, sb, sn)) return 1;
            if (run<4, true>(buf, rows, out, cyc, ncu, busy, sa, sb, sn)) return 1;
            if (run<8, false>(buf, rows, out, cyc, ncu, busy, sa, sb, sn)) return 1;
            if (run<8, true>(buf, rows, out, cyc, ncu, busy, sa, sb, sn)) return 1;
        }
    }
    return 0;
}
