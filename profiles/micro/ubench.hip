// Micro-benchmarks behind the kernel design choices (LDS gather patterns, fp64 add rate, clock).
// hipcc --offload-arch=gfx950 -O3 ubench.hip -o ubench && ./ubench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %d at %d\n", e, __LINE__); return 1; } } while (0)

// MODE 0: ds_read_b64, each lane one of 6 addresses inside a 48-byte record (our gather pattern)
// MODE 1: ds_read_b64, all 64 lanes distinct consecutive addresses
// MODE 2: ds_read_b64, all lanes the same address
// MODE 3: ds_read_b128, 3 distinct 16-byte addresses (f-step pattern)
// MODE 4: no LDS: v_add_f64 only
// MODE 5: ds_read_b64 pattern 0 without the add (read + xor into an int accumulator)
// MODE 6: ds_read_b32 gather (6 addresses) + v_add_u32 (round 4: is a 4-byte gather cheaper than an 8-byte one?)
// MODE 7 / 8: ds_read_b128 / ds_read_b64 gather of which ONE dword is used -- the compiler shrinks both to ds_read_b32
// MODE 9: ds_read_b64 gather + 64-bit INTEGER add (fixed-point sums: v_lshl_add_u64)
// MODE 10: ds_read_b64 gather + v_pk_add_f32 (two fp32 sums side by side)
template <int MODE>
__global__ __launch_bounds__(1024) void k(const int *sel, double *out, long long *cyc, int iters) {
    __shared__ __attribute__((aligned(16))) double lds[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = 1e-3 * i;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int s = sel[threadIdx.x & 63];     // 0..5
    uint32_t a;
    if (MODE == 0 || MODE == 5 || MODE == 8 || MODE == 9 || MODE == 10) a = s * 8;
    else if (MODE == 6) a = s * 4;
    else if (MODE == 1) a = lane * 8;
    else if (MODE == 2) a = 0;
    else a = (s % 3) * 16;
    double d0 = 0, d1 = 0, d2 = 0, d3 = 0;
    uint32_t i0 = 0, i1 = 0, i2 = 0, i3 = 0;
    unsigned long long q0 = 0, q1 = 0, q2 = 0, q3 = 0;
    typedef float f2v __attribute__((ext_vector_type(2)));
    f2v p0 = {0, 0}, p1 = {0, 0}, p2 = {0, 0}, p3 = {0, 0};
    const char *b = (const char *)lds;
    const long long rt0 = (long long)wall_clock64();
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const uint32_t o = (uint32_t)((it & 7) * 6144 + j * 192);
            if (MODE == 4) {
                d0 += 1.0000001; d1 += 1.0000002; d2 += 1.0000003; d3 += 1.0000004;
            } else if (MODE == 6) {
                i0 += *(const uint32_t *)(b + a + o); i1 += *(const uint32_t *)(b + a + o + 48);
                i2 += *(const uint32_t *)(b + a + o + 96); i3 += *(const uint32_t *)(b + a + o + 144);
            } else if (MODE == 7) {
                const uint4 v0 = *(const uint4 *)(b + (a & ~15u) + o), v1 = *(const uint4 *)(b + (a & ~15u) + o + 48);
                const uint4 v2 = *(const uint4 *)(b + (a & ~15u) + o + 96), v3 = *(const uint4 *)(b + (a & ~15u) + o + 144);
                i0 ^= v0.x; i1 ^= v1.y; i2 ^= v2.z; i3 ^= v3.w;
            } else if (MODE == 8) {
                const uint2 v0 = *(const uint2 *)(b + a + o), v1 = *(const uint2 *)(b + a + o + 48);
                const uint2 v2 = *(const uint2 *)(b + a + o + 96), v3 = *(const uint2 *)(b + a + o + 144);
                i0 ^= v0.x; i1 ^= v1.y; i2 ^= v2.x; i3 ^= v3.y;
            } else if (MODE == 9) {
                q0 += *(const unsigned long long *)(b + a + o); q1 += *(const unsigned long long *)(b + a + o + 48);
                q2 += *(const unsigned long long *)(b + a + o + 96); q3 += *(const unsigned long long *)(b + a + o + 144);
            } else if (MODE == 10) {
                typedef float f2 __attribute__((ext_vector_type(2)));
                p0 += *(const f2 *)(b + a + o); p1 += *(const f2 *)(b + a + o + 48);
                p2 += *(const f2 *)(b + a + o + 96); p3 += *(const f2 *)(b + a + o + 144);
            } else if (MODE == 3) {
                const double2 v0 = *(const double2 *)(b + a + o), v1 = *(const double2 *)(b + a + o + 48);
                d0 += v0.x; d1 += v0.y; d2 += v1.x; d3 += v1.y;
            } else {
                d0 += *(const double *)(b + a + o);
                d1 += *(const double *)(b + a + o + 48);
                d2 += *(const double *)(b + a + o + 96);
                d3 += *(const double *)(b + a + o + 144);
            }
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = d0 + d1 + d2 + d3 + (double)(i0 + i1 + i2 + i3) + (double)(q0 + q1 + q2 + q3) + (double)(p0.x + p1.y + p2.x + p3.y);
    if (threadIdx.x == 0) {
        cyc[blockIdx.x] = t1 - t0;
        cyc[256 + blockIdx.x] = rt0;                       // 100 MHz wall clock at the start / end of the loop
        cyc[512 + blockIdx.x] = (long long)wall_clock64();
    }
}

template <int MODE>
int run(const char *name, int waves_per_block, int *sel, double *out, long long *cyc) {
    const int iters = 2000, blocks = 256;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64 * waves_per_block), 0, 0, sel, out, cyc, 10);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64 * waves_per_block), 0, 0, sel, out, cyc, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    long long c;
    CHECK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
    {
        // were all blocks in flight together, and what clock did they run at?  (s_memtime = shader cycles, wall_clock64 = 100 MHz)
        std::vector<long long> h(768);
        CHECK(hipMemcpy(h.data(), cyc, 768 * 8, hipMemcpyDeviceToHost));
        long long s0 = h[256], s1 = h[256], e0 = h[512], e1 = h[512];
        double clk = 0;
        for (int b = 0; b < blocks; ++b) {
            if (h[256 + b] < s0) s0 = h[256 + b];
            if (h[256 + b] > s1) s1 = h[256 + b];
            if (h[512 + b] < e0) e0 = h[512 + b];
            if (h[512 + b] > e1) e1 = h[512 + b];
            clk += (double)h[b] / ((double)(h[512 + b] - h[256 + b]) * 10.0);       // cycles per ns
        }
        printf("    blocks start within %.1f us, end within %.1f us of each other; loop %.1f us; mean in-kernel clock %.2f GHz\n",
               (s1 - s0) / 100.0, (e1 - e0) / 100.0, (e1 - s0) / 100.0, clk / blocks);
    }
    const double wave_instr = (double)iters * 16 * 4;      // read(+add) groups per wave
    const double ns_per = ms * 1e6 / wave_instr;           // per wave-instruction-group, one wave's timeline
    printf("%-44s waves/CU=%2d  %8.3f ms  %6.2f ns per (read+add) per wave  -> %6.2f ns per SIMD-slot; memtime ticks %lld (%.1f ticks/us)\n",
           name, waves_per_block, ms, ns_per, ns_per / (waves_per_block / 4.0), c, c / (ms * 1e3));
    return 0;
}

int main() {
    int h[64];
    for (int i = 0; i < 64; ++i) h[i] = (i * 7 + i / 5) % 6;
    int *sel; double *out; long long *cyc;
    CHECK(hipMalloc(&sel, 256)); CHECK(hipMalloc(&out, 256 * 1024 * 8)); CHECK(hipMalloc(&cyc, 768 * 8));
    CHECK(hipMemcpy(sel, h, 256, hipMemcpyHostToDevice));
    for (int wpb : {4, 8, 16}) {
        run<0>("b64 gather 6 addrs/48B (panel pattern)", wpb, sel, out, cyc);
        run<1>("b64 64 distinct consecutive", wpb, sel, out, cyc);
        run<2>("b64 all lanes same address", wpb, sel, out, cyc);
        run<3>("b128 3 addrs (f pattern; 2 reads+4 adds/grp)", wpb, sel, out, cyc);
        run<4>("v_add_f64 only (4 adds/grp)", wpb, sel, out, cyc);
        run<6>("b32 gather 6 addrs + v_add_u32", wpb, sel, out, cyc);
        run<8>("b64 gather 6 addrs, one dword used (= b32)", wpb, sel, out, cyc);
        run<7>("b128 gather 3 addrs, one dword used (= b32)", wpb, sel, out, cyc);
        run<9>("b64 gather 6 addrs + 64-bit integer add", wpb, sel, out, cyc);
        run<10>("b64 gather 6 addrs + v_pk_add_f32", wpb, sel, out, cyc);
    }
    return 0;
}
