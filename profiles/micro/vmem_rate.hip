// How many cycles of a CU does ONE wave-wide vector-memory instruction cost, by width?  (profiles/r04_ubench_vmem_rate.txt)
// Every CU runs 2 workgroups of 16 waves; each wave issues K independent, fully coalesced loads (or stores) of W bytes per lane
// from a buffer that stays in its L1/L2 (64 KB per workgroup), 8 in flight.  cycles per instruction and CU = time * clock / (K * 32).
//   hipcc -O3 --offload-arch=gfx950 profiles/micro/vmem_rate.hip -o profiles/micro/vmem_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

template <typename T>
__global__ __launch_bounds__(1024) void k_load(const T *__restrict__ buf, int K, int words_per_wg, uint32_t *out) {
    const T *p = buf + (size_t)blockIdx.x * words_per_wg;
    const int lane_off = threadIdx.x;             // wave w reads elements [w*64 .. w*64+63] + i*1024: coalesced
    uint32_t acc = 0;
    for (int i = 0; i < K; i += 8) {
        T v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = p[(((i + j) * 1024) + lane_off) % words_per_wg];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t *q = reinterpret_cast<const uint32_t *>(&v[j]);
            for (unsigned t = 0; t < sizeof(T) / 4 || t < 1; ++t) acc += sizeof(T) >= 4 ? q[t] : (uint32_t)(*reinterpret_cast<const uint8_t *>(&v[j]));
        }
    }
    if (acc == 0xdeadbeef) out[0] = acc;
}
template <typename T>
__global__ __launch_bounds__(1024) void k_store(T *__restrict__ buf, int K, int words_per_wg) {
    T *p = buf + (size_t)blockIdx.x * words_per_wg;
    T v;
    memset(&v, threadIdx.x & 0xff, sizeof(T));
    for (int i = 0; i < K; ++i) p[((i * 1024) + threadIdx.x) % words_per_wg] = v;
}

template <typename T>
void run(const char *name, bool store) {
    const int K = 4096, wgs = 512, bytes_per_wg = 64 * 1024;
    const int words = bytes_per_wg / sizeof(T);
    void *buf;
    uint32_t *out;
    hipMalloc(&buf, (size_t)wgs * bytes_per_wg);
    hipMemset(buf, 1, (size_t)wgs * bytes_per_wg);
    hipMalloc(&out, 64);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(a);
        if (store) hipLaunchKernelGGL(k_store<T>, dim3(wgs), dim3(1024), 0, 0, (T *)buf, K, words);
        else hipLaunchKernelGGL(k_load<T>, dim3(wgs), dim3(1024), 0, 0, (const T *)buf, K, words, out);
        hipEventRecord(b);
        hipEventSynchronize(b);
    }
    float ms;
    hipEventElapsedTime(&ms, a, b);
    hipDeviceProp_t pr;
    hipGetDeviceProperties(&pr, 0);
    const double clk = pr.clockRate * 1e3;        // Hz
    const double instr_per_cu = (double)K * 16 * wgs / pr.multiProcessorCount;
    printf("%-28s %7.3f ms  %6.1f cycles per wave-instruction and CU (at %.2f GHz), %7.1f GB/s\n", name, ms, ms * 1e-3 * clk / instr_per_cu,
           clk * 1e-9, (double)K * wgs * 1024 * sizeof(T) / (ms * 1e-3) * 1e-9);
    hipFree(buf);
    hipFree(out);
}

int main() {
    run<uint8_t>("load   1 B/lane", false);
    run<uint32_t>("load   4 B/lane", false);
    run<uint2>("load   8 B/lane", false);
    run<uint4>("load  16 B/lane", false);
    run<uint8_t>("store  1 B/lane", true);
    run<uint32_t>("store  4 B/lane", true);
    run<uint2>("store  8 B/lane", true);
    run<uint4>("store 16 B/lane", true);
    return 0;
}
