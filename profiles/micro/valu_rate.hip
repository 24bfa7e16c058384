// How many cycles of its SIMD does a wave64 vector instruction take?  (The sweep kernels are bound by vector-instruction
// issue: this decides what an address or a sum may cost.)  8 waves per SIMD, every CU; cycles from s_memtime in the kernel,
// so the clock the chip holds does not matter.
//   hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate && ./valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %d at %d\n", e, __LINE__); return 1; } } while (0)

#define REP8(x) x x x x x x x x
#define BODY(NAME, ASM)                                                                                   \
    __global__ __launch_bounds__(512) void NAME(uint64_t *out, int iters) {                               \
        uint32_t a = threadIdx.x, b = threadIdx.x * 3 + 1, c = 7, d = 9;                                   \
        double x = threadIdx.x, y = 1.5;                                                                  \
        const uint64_t t0 = __builtin_amdgcn_s_memtime();                                                 \
        for (int it = 0; it < iters; ++it) { REP8(REP8(asm volatile(ASM : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(x), "+v"(y));)) } \
        const uint64_t t1 = __builtin_amdgcn_s_memtime();                                                 \
        if (threadIdx.x % 64 == 0) out[blockIdx.x * 8 + threadIdx.x / 64] = t1 - t0;                     \
        if (a + b + c + d + (uint32_t)x + (uint32_t)y == 0x12345) out[0] = 0;                             \
    }
// each ASM body = 2 independent instructions
BODY(k_and, "v_and_b32 %0, %1, %0\n v_and_b32 %2, %3, %2")
BODY(k_lshr, "v_lshrrev_b32 %0, 3, %0\n v_lshrrev_b32 %2, 5, %2")
BODY(k_andor, "v_and_or_b32 %0, %1, 24, %0\n v_and_or_b32 %2, %3, 24, %2")
BODY(k_andlit, "v_and_b32 %0, 0x1e0, %0\n v_and_b32 %2, 0x1f8, %2")
BODY(k_addu, "v_add_u32 %0, %1, %0\n v_add_u32 %2, %3, %2")
BODY(k_lshladd, "v_lshl_add_u32 %0, %1, 3, %0\n v_lshl_add_u32 %2, %3, 3, %2")
BODY(k_bfe, "v_bfe_u32 %0, %1, 8, 6\n v_bfe_u32 %2, %3, 16, 6")
BODY(k_madu24, "v_mad_u32_u24 %0, %1, 8, %0\n v_mad_u32_u24 %2, %3, 8, %2")
BODY(k_perm, "v_perm_b32 %0, %1, %0, %3\n v_perm_b32 %2, %3, %2, %1")
BODY(k_fma32, "v_fma_f32 %0, %1, %0, %1\n v_fma_f32 %2, %3, %2, %3")
BODY(k_add64, "v_add_f64 %4, %4, %5\n v_add_f64 %5, %5, %4")
BODY(k_mulhi, "v_mul_hi_u32 %0, %1, %0\n v_mul_hi_u32 %2, %3, %2")
BODY(k_mullo, "v_mul_lo_u32 %0, %1, %0\n v_mul_lo_u32 %2, %3, %2")
BODY(k_cndmask, "v_cndmask_b32 %0, %1, %0, vcc\n v_cndmask_b32 %2, %3, %2, vcc")
BODY(k_sdwa, "v_or_b32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_or_b32_sdwa %2, %3, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2")
BODY(k_pkadd, "v_pk_add_f32 %4, %4, %5\n v_pk_add_f32 %5, %5, %4")
BODY(k_fmac64, "v_fmac_f64_e32 %4, 1.0, %5\n v_fmac_f64_e32 %5, 1.0, %4")
BODY(k_fma64, "v_fma_f64 %4, %4, 1.0, %5\n v_fma_f64 %5, %5, 1.0, %4")
BODY(k_cndmask_s, "v_cndmask_b32_e64 %0, %1, %0, s[20:21]\n v_cndmask_b32_e64 %2, %3, %2, s[20:21]")
BODY(k_cndmask_c, "v_cndmask_b32_e64 %0, 0, 1, s[20:21]\n v_cndmask_b32_e64 %2, 0, 2, s[20:21]")
BODY(k_addc, "v_addc_co_u32 %0, vcc, %1, %0, vcc\n v_addc_co_u32 %2, vcc, %3, %2, vcc")
BODY(k_cmp, "v_cmp_lt_f32 vcc, %0, %1\n v_cmp_lt_f32 vcc, %2, %3")
BODY(k_cmp64, "v_cmp_lt_f64 vcc, %4, %5\n v_cmp_lt_f64 vcc, %5, %4")
BODY(k_exp, "v_exp_f32 %0, %0\n v_exp_f32 %2, %2")
BODY(k_cvt, "v_cvt_f32_f64 %0, %4\n v_cvt_f32_f64 %2, %5")
BODY(k_fmac32, "v_fmac_f32_e32 %0, 1.0, %1\n v_fmac_f32_e32 %2, 1.0, %3")
// 64-bit product of two 32-bit words in ONE instruction (what the compiler makes of Philox's mulhilo): the pair %4 (x) / %5 (y) as 64-bit destinations
BODY(k_mad64, "v_mad_u64_u32 %4, vcc, %0, %1, 0\n v_mad_u64_u32 %5, vcc, %2, %3, 0")
BODY(k_exp_log, "v_log_f32 %0, %0\n v_log_f32 %2, %2")
BODY(k_max3, "v_max3_f32 %0, %1, %0, %3\n v_max3_f32 %2, %3, %2, %1")
BODY(k_cvtu, "v_cvt_f32_u32 %0, %0\n v_cvt_f32_u32 %2, %2")

int main() {
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    uint64_t *out;
    CHECK(hipMalloc(&out, (size_t)cus * 8 * 4 * 8));
    struct { const char *name; void (*fn)(uint64_t *, int); } ks[] = {
        {"v_and_b32", k_and}, {"v_lshrrev_b32", k_lshr}, {"v_and_or_b32 (inline const)", k_andor}, {"v_and_b32 (literal)", k_andlit},
        {"v_add_u32", k_addu}, {"v_lshl_add_u32", k_lshladd}, {"v_bfe_u32", k_bfe}, {"v_mad_u32_u24", k_madu24}, {"v_perm_b32", k_perm},
        {"v_fma_f32", k_fma32}, {"v_add_f64", k_add64}, {"v_mul_hi_u32", k_mulhi}, {"v_mul_lo_u32", k_mullo}, {"v_cndmask_b32", k_cndmask},
        {"v_or_b32_sdwa (byte select)", k_sdwa}, {"v_pk_add_f32", k_pkadd}, {"v_fmac_f64_e32 (VOP2, x 1.0)", k_fmac64}, {"v_fma_f64 (VOP3)", k_fma64}, {"v_fmac_f32_e32", k_fmac32}, {"v_cndmask_b32_e64 (SGPR-pair mask)", k_cndmask_s}, {"v_cndmask_b32_e64 (constants, SGPR mask)", k_cndmask_c}, {"v_addc_co_u32 (vcc in/out)", k_addc}, {"v_cmp_lt_f32 -> vcc", k_cmp}, {"v_cmp_lt_f64 -> vcc", k_cmp64}, {"v_exp_f32", k_exp}, {"v_cvt_f32_f64", k_cvt},
        {"v_mad_u64_u32 (64-bit product)", k_mad64}, {"v_log_f32", k_exp_log}, {"v_max3_f32", k_max3}, {"v_cvt_f32_u32", k_cvtu}};
    const int iters = 2000;
    uint64_t *h = (uint64_t *)malloc((size_t)cus * 4 * 8 * 8);
    for (auto &k : ks) {
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(k.fn, dim3(cus * 4), dim3(512), 0, 0, out, iters);       // 4 blocks x 8 waves = 32 waves per CU = 8 per SIMD
            CHECK(hipDeviceSynchronize());
        }
        CHECK(hipMemcpy(h, out, (size_t)cus * 4 * 8 * 8, hipMemcpyDeviceToHost));
        double s = 0;
        for (int i = 0; i < cus * 4 * 8; ++i) s += (double)h[i];
        s /= cus * 4 * 8;
        // each wave issued iters * 64 * 2 instructions; 8 waves share a SIMD
        printf("%-32s %6.2f cycles of the SIMD per wave-instruction (8 waves per SIMD)\n", k.name, s / ((double)iters * 128 * 8));
    }
    return 0;
}
