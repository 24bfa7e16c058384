// fp64 MFMA peak on the box ("do not trust remembered datasheet values"): back-to-back
// v_mfma_f64_16x16x4_f64 on independent accumulators, 1 / 2 / 4 waves per SIMD, every CU.
//   hipcc --offload-arch=gfx950 -O3 mfma64.hip -o mfma64 && ./mfma64
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double double4_t __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %d at %d\n", e, __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(1024) void k(double *out, int iters, double a0, double b0) {
    double4_t acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = double4_t{0.0, 0.0, 0.0, 0.0};
    const double a = a0 + threadIdx.x * 1e-9, b = b0 - threadIdx.x * 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    double *out;
    CHECK(hipMalloc(&out, (size_t)cus * 1024 * 8));
    for (int waves_per_simd : {1, 2, 4}) {
        const int threads = 64 * 4 * waves_per_simd, iters = 20000;
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k, dim3(cus), dim3(threads), 0, 0, out, 100, 1.0000001, 0.9999999);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k, dim3(cus), dim3(threads), 0, 0, out, iters, 1.0000001, 0.9999999);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double mfma = (double)cus * 4 * waves_per_simd * iters * 16;
        const double tflops = mfma * 2048.0 / (ms * 1e-3) / 1e12;
        printf("v_mfma_f64_16x16x4_f64: %d CUs, %d waves/SIMD: %.3f ms, %.1f TFLOP/s, %.1f cycles per MFMA per SIMD at 2.4 GHz\n", cus,
               waves_per_simd, ms, tflops, ms * 1e-3 * 2.4e9 / ((double)waves_per_simd * iters * 16));
    }
    return 0;
}
