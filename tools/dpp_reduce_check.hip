// Check of the DPP wave maximum used by search.hip (wave_max_u32) against a shuffle reduction, on random and adversarial lanes.
//   hipcc -O3 --offload-arch=gfx950 tools/dpp_reduce_check.hip -o tools/bin/dpp_reduce_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdint>
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#define LBH_DPP_MAX(ctrl)                                                                              \
    {                                                                                                  \
        const uint32_t t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, 0xF, 0xF, false);    \
        v = t > v ? t : v;                                                                             \
    }
    LBH_DPP_MAX(0x111) LBH_DPP_MAX(0x112) LBH_DPP_MAX(0x114) LBH_DPP_MAX(0x118) LBH_DPP_MAX(0x142) LBH_DPP_MAX(0x143)
#undef LBH_DPP_MAX
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__global__ void k(const uint32_t* in, uint32_t* out_dpp, uint32_t* out_ref) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    uint32_t v = in[i];
    uint32_t r = v;
    for (int o = 32; o > 0; o >>= 1) { const uint32_t t = (uint32_t)__shfl_xor((int)r, o); r = t > r ? t : r; }
    const uint32_t d = wave_max_u32(v);
    if (threadIdx.x == 0) { out_dpp[blockIdx.x] = d; out_ref[blockIdx.x] = r; }
}
int main() {
    const int W = 4096;
    std::vector<uint32_t> h(W * 64);
    uint32_t x = 12345;
    for (int w = 0; w < W; ++w)
        for (int l = 0; l < 64; ++l) {
            x = x * 1664525u + 1013904223u;
            h[w * 64 + l] = (w < 64) ? (l == w ? 0xFFFFFFF0u + (w & 7) : (x >> 8)) : (w < 128 ? (l == (w - 64) ? 5u : 0u) : x);
        }
    uint32_t *in, *a, *b;
    hipMalloc(&in, h.size() * 4); hipMalloc(&a, W * 4); hipMalloc(&b, W * 4);
    hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    k<<<W, 64>>>(in, a, b);
    std::vector<uint32_t> ha(W), hb(W);
    hipMemcpy(ha.data(), a, W * 4, hipMemcpyDeviceToHost); hipMemcpy(hb.data(), b, W * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int w = 0; w < W; ++w) {
        uint32_t m = 0;
        for (int l = 0; l < 64; ++l) m = h[w * 64 + l] > m ? h[w * 64 + l] : m;
        if (ha[w] != m || hb[w] != m) { if (bad < 5) printf("wave %d: dpp %u shfl %u host %u\n", w, ha[w], hb[w], m); ++bad; }
    }
    printf("%d waves checked, %d mismatches\n", W, bad);
    return bad != 0;
}
