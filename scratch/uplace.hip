// uplace.hip — where do the waves of a workgroup land? Launches G workgroups of W waves that all stay resident
// (each spins for a while) and records, per wave, HW_ID (wave slot, SIMD, CU, SH, SE) and XCC_ID. Prints how the
// waves with index w of the workgroups that share a CU are spread over the CU's four SIMDs — the question behind
// the role assignment of the wave-specialised kernels (k_lstm_pipe: which SIMD runs the recurrent wave?).
//   hipcc --offload-arch=gfx950 -O2 scratch/uplace.hip -o scratch/uplace && scratch/uplace [waves] [groups] [lds_bytes]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

#define GETREG(id, off, size) __builtin_amdgcn_s_getreg(((size - 1) << 11) | ((off) << 6) | (id))

__global__ void k(unsigned* rec, int spin)
{
    extern __shared__ float smem[];
    const int wave = threadIdx.x >> 6;
    const unsigned hw = GETREG(4, 0, 32);       // HW_REG_HW_ID
    const unsigned xcc = GETREG(20, 0, 32);     // HW_REG_XCC_ID
    float a = threadIdx.x * 1e-9f;
    for (int i = 0; i < spin; ++i) a = __builtin_fmaf(a, 1.000001f, 1e-7f);
    if ((threadIdx.x & 63) == 0) {
        unsigned* r = rec + ((size_t)blockIdx.x * (blockDim.x >> 6) + wave) * 2;
        r[0] = hw; r[1] = xcc;
    }
    if (a == 123.f) smem[threadIdx.x] = a;
}

int main(int argc, char** argv)
{
    const int W = argc > 1 ? atoi(argv[1]) : 3, G = argc > 2 ? atoi(argv[2]) : 1024, lds = argc > 3 ? atoi(argv[3]) : 8192;
    unsigned* d; hipMalloc(&d, (size_t)G * W * 8);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    k<<<G, W * 64, lds>>>(d, 20000);
    hipDeviceSynchronize();
    std::vector<unsigned> h((size_t)G * W * 2);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    // key: (xcc, se, sh, cu) -> per wave index: histogram over simd
    std::map<unsigned, std::vector<std::vector<int>>> cus;       // [wave][simd] counts
    std::map<unsigned, std::vector<int>> wgs;                     // workgroups per CU
    for (int g = 0; g < G; ++g)
        for (int w = 0; w < W; ++w) {
            const unsigned hw = h[((size_t)g * W + w) * 2], xcc = h[((size_t)g * W + w) * 2 + 1] & 0xf;
            const unsigned simd = (hw >> 4) & 3, cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
            const unsigned key = (xcc << 12) | (se << 8) | (sh << 4) | cu;
            auto& c = cus[key];
            if (c.empty()) c.assign(W, std::vector<int>(4, 0));
            c[w][simd]++;
            if (w == 0) wgs[key].push_back(g);
        }
    printf("%d workgroups x %d waves, %d B LDS: %zu CUs used\n", G, W, lds, cus.size());
    int shown = 0;
    for (auto& kv : cus) {
        if (shown++ >= 6) break;
        printf("CU %04x  WGs:", kv.first);
        for (int g : wgs[kv.first]) printf(" %d", g);
        printf("\n");
        for (int w = 0; w < W; ++w)
            printf("   wave %d on SIMD0..3: %d %d %d %d\n", w, kv.second[w][0], kv.second[w][1], kv.second[w][2], kv.second[w][3]);
    }
    // summary: for each wave index, how many CUs have all of their wave-w instances on ONE simd
    for (int w = 0; w < W; ++w) {
        int same = 0, spread = 0;
        std::map<int, int> maxper;
        for (auto& kv : cus) {
            int mx = 0, tot = 0;
            for (int s = 0; s < 4; ++s) { mx = kv.second[w][s] > mx ? kv.second[w][s] : mx; tot += kv.second[w][s]; }
            if (tot > 1 && mx == tot) ++same; else ++spread;
            maxper[mx]++;
        }
        printf("wave %d: CUs with all instances on one SIMD: %d, otherwise: %d; max-per-SIMD histogram:", w, same, spread);
        for (auto& m : maxper) printf(" %dx%d", m.first, m.second);
        printf("\n");
    }
    return 0;
}
