// Where do the waves of co-resident workgroups land? 1024 workgroups x 4 waves, 39.9 KiB of LDS each (4 per CU, as
// k_conv_mfma): prints, per workgroup, the HW_ID fields of its four waves. hipcc --offload-arch=gfx950 -O2 uhwid.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256) void k(unsigned* out, int spin)
{
    extern __shared__ float smem[];
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);      // HW_ID, all 32 bits
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);     // XCC_ID
    smem[threadIdx.x] = (float)hw;
    for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(10);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { out[2 * (blockIdx.x * 4 + threadIdx.x / 64)] = hw; out[2 * (blockIdx.x * 4 + threadIdx.x / 64) + 1] = xcc; }
}
int main()
{
    const int B = 1024;
    unsigned* d; hipMalloc(&d, B * 4 * 2 * sizeof(unsigned));
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 40900);
    hipLaunchKernelGGL(k, dim3(B), dim3(256), 40900, 0, d, 2000);
    hipDeviceSynchronize();
    std::vector<unsigned> h(B * 4 * 2);
    hipMemcpy(h.data(), d, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost);
    // HW_ID (gfx9): wave_id [3:0], simd_id [5:4], pipe [7:6], cu_id [11:8], sh [12], se [15:13]
    for (int b = 0; b < B; ++b) {
        if (!(b < 40 || (b % 97) == 0)) continue;
        printf("wg %4d:", b);
        for (int w = 0; w < 4; ++w) {
            const unsigned hw = h[2 * (b * 4 + w)], x = h[2 * (b * 4 + w) + 1];
            printf("  [xcc %u se %u cu %2u simd %u slot %u]", x & 15, (hw >> 13) & 7, (hw >> 8) & 15, (hw >> 4) & 3, hw & 15);
        }
        printf("\n");
    }
    // how many workgroups share (xcc, se, cu), and how their wave-0 SIMDs and slots distribute
    int same_simd = 0, total = 0;
    for (int b = 0; b < B; ++b)
        for (int c = b + 1; c < B; ++c) {
            const unsigned hb = h[2 * b * 4], hc = h[2 * c * 4];
            if ((h[2 * b * 4 + 1] & 15) == (h[2 * c * 4 + 1] & 15) && ((hb >> 8) & 0xff) == ((hc >> 8) & 0xff)) {
                ++total;
                if (((hb >> 4) & 3) == ((hc >> 4) & 3)) ++same_simd;
            }
        }
    printf("pairs of workgroups on one CU: %d, of which wave 0 on the same SIMD: %d\n", total, same_simd);
    return 0;
}
