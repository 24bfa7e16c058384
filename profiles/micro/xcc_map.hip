// Which XCD does workgroup i of a 1-D grid land on?  (512 workgroups of 1024 threads, 80 KB of LDS each: two per CU, the
// shape of the pipelined r pass.)   hipcc --offload-arch=gfx950 -O3 xcc_map.hip -o xcc_map
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %d at %d\n", e, __LINE__); return 1; } } while (0)
__global__ __launch_bounds__(1024) void k(unsigned *out, int spin) {
    extern __shared__ double sm[];
    if (threadIdx.x == 0) {
        const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xf;            // HW_REG_XCC_ID
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);                     // HW_REG_HW_ID
        out[blockIdx.x] = (xcc << 16) | (hw & 0xffff);
    }
    sm[threadIdx.x] = threadIdx.x;
    for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(64);                           // stay resident for a while
}
int main() {
    unsigned *out, h[1024];
    CHECK(hipMalloc(&out, sizeof(h)));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k), hipFuncAttributeMaxDynamicSharedMemorySize, 81920));
    for (int rep = 0; rep < 3; ++rep) {
        const int n = rep == 2 ? 500 : 512;
        hipLaunchKernelGGL(k, dim3(n), dim3(1024), 81920, 0, out, 2000);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
        int bad = 0, cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = 0; i < n; ++i) {
            const int x = (h[i] >> 16) & 0xf;
            cnt[x & 7]++;
            if (x != i % 8) ++bad;
        }
        printf("launch %d (%d workgroups): workgroups whose XCC id != blockIdx %% 8: %d; per XCC:", rep, n, bad);
        for (int x = 0; x < 8; ++x) printf(" %d", cnt[x]);
        printf("\n   first 24 ids:");
        for (int i = 0; i < 24; ++i) printf(" %u", (h[i] >> 16) & 0xf);
        printf("\n");
    }
    return 0;
}
