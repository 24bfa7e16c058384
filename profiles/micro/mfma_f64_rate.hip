// What does v_mfma_f64_16x16x4_f64 sustain on this chip?  NACC independent accumulators per wave, W waves per SIMD,
// every CU; nothing but MFMAs in the loop.   hipcc --offload-arch=gfx950 -O3 mfma_f64_rate.hip -o mfma_f64_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %d at %d\n", e, __LINE__); return 1; } } while (0)
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(1024) void k(double *out, int iters) {
    d4 acc[NACC];
    for (int j = 0; j < NACC; ++j) acc[j] = d4{0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[j], 0, 0, 0);
    }
    double s = 0;
    for (int j = 0; j < NACC; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    if (s == 12345.678) out[0] = s;
}

template <int NACC>
int run(int cus, int threads, const char *what) {
    double *out;
    CHECK(hipMalloc(&out, 8));
    const int iters = 2000;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<NACC>, dim3(cus), dim3(threads), 0, 0, out, iters);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
    }
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double flops = (double)cus * (threads / 64) * iters * 4 * NACC * 2048.0;
    printf("%-44s %7.2f TFLOP/s  (%.3f ms)\n", what, flops / (ms * 1e-3) / 1e12, ms);
    return 0;
}

// the same with a workgroup barrier every 24 MFMAs and dynamic LDS (what the Gram kernel's k-step looks like)
template <int NACC>
__global__ __launch_bounds__(1024) void kb(double *out, int iters) {
    extern __shared__ double sm[];
    d4 acc[NACC];
    for (int j = 0; j < NACC; ++j) acc[j] = d4{0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[j], 0, 0, 0);
        sm[threadIdx.x + 1024 * (it & 1)] = a;
        __syncthreads();
        a += sm[(threadIdx.x ^ 1) + 1024 * (it & 1)] * 1e-9;
    }
    double s = 0;
    for (int j = 0; j < NACC; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    if (s == 12345.678) out[0] = s;
}
// random operands, different for every MFMA (from LDS, as in a real product): what does the chip sustain then?
template <int NACC>
__global__ __launch_bounds__(1024) void kr(double *out, int iters, const double *rnd) {
    extern __shared__ double sm[];
    for (int i = threadIdx.x; i < 8192; i += 1024) sm[i] = rnd[i];
    __syncthreads();
    d4 acc[NACC];
    for (int j = 0; j < NACC; ++j) acc[j] = d4{0, 0, 0, 0};
    const int l = threadIdx.x & 63;
    for (int it = 0; it < iters; ++it) {
        const double *P = sm + ((it * 64) & 4095) + l;
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(P[(r * NACC + j) * 64], P[(r * NACC + j) * 64 + 2048], acc[j], 0, 0, 0);
    }
    double s = 0;
    for (int j = 0; j < NACC; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    if (s == 12345.678) out[0] = s;
}
int runr(int wgs, const char *what) {
    double *out, *rnd;
    CHECK(hipMalloc(&out, 8));
    CHECK(hipMalloc(&rnd, 8192 * 8));
    double *h = (double *)malloc(8192 * 8);
    unsigned long long x = 88172645463325252ull;
    for (int i = 0; i < 8192; ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; h[i] = (double)(x >> 11) / 9007199254740992.0 * 2.0 - 1.0; }
    CHECK(hipMemcpy(rnd, h, 8192 * 8, hipMemcpyHostToDevice));
    const int iters = 2000;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&kr<6>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(kr<6>, dim3(wgs), dim3(1024), 65536, 0, out, iters, rnd);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
    }
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double flops = (double)wgs * 16 * iters * 4 * 6 * 2048.0;
    printf("%-44s %7.2f TFLOP/s  (%.3f ms)\n", what, flops / (ms * 1e-3) / 1e12, ms);
    return 0;
}
int runb(int wgs, size_t lds, const char *what) {
    double *out;
    CHECK(hipMalloc(&out, 8));
    const int iters = 2000;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&kb<6>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(kb<6>, dim3(wgs), dim3(1024), lds, 0, out, iters);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
    }
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double flops = (double)wgs * 16 * iters * 4 * 6 * 2048.0;
    printf("%-44s %7.2f TFLOP/s  (%.3f ms)\n", what, flops / (ms * 1e-3) / 1e12, ms);
    return 0;
}

int main() {
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    run<1>(cus, 256, "1 wave/SIMD, 1 accumulator (dependent)");
    run<2>(cus, 256, "1 wave/SIMD, 2 accumulators");
    run<4>(cus, 256, "1 wave/SIMD, 4 accumulators");
    run<6>(cus, 256, "1 wave/SIMD, 6 accumulators");
    run<6>(cus, 512, "2 waves/SIMD, 6 accumulators");
    run<6>(cus, 1024, "4 waves/SIMD, 6 accumulators");
    run<1>(cus, 1024, "4 waves/SIMD, 1 accumulator");
    runb(cus, 66560, "256 WGs, barrier per 24 MFMAs, 65 KB LDS");
    runb(200, 66560, "200 WGs, barrier per 24 MFMAs, 65 KB LDS");
    runb(200, 16384, "200 WGs, barrier per 24 MFMAs, 16 KB LDS");
    runr(cus, "256 WGs, random operands from LDS");
    runr(200, "200 WGs, random operands from LDS");
    return 0;
}
