// Micro-benchmark: cycles per v_mfma_f32_32x32x16_bf16 on one SIMD for (a) one dependent chain, (b) two
// alternating chains, with 1 or 2 waves per SIMD, and for the fp32 32x32x2 form.  Build:
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_rate.hip -o tools/bin/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ void k(float* out, long long* cyc, int iters, float seedv) {
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(seedv * (threadIdx.x + e)); b[e] = (__bf16)(seedv * (e + 1)); }
    f32x16 c0 = {0}, c1 = {0};
    float af = seedv * threadIdx.x, bf = seedv;
    bf16x8 ra[4], rb[4];  // MODE 3: operands with random mantissas (different per lane and per k-step)
    if (MODE == 3) {
        unsigned x = 0x9E3779B9u * (threadIdx.x + 1) + 12345u * blockIdx.x;
        for (int s = 0; s < 4; ++s)
            for (int e = 0; e < 8; ++e) {
                x = x * 1664525u + 1013904223u; ra[s][e] = (__bf16)(((int)(x >> 8) & 0xFFFF) * (1.0f / 65536.f) - 0.5f);
                x = x * 1664525u + 1013904223u; rb[s][e] = (__bf16)(((int)(x >> 8) & 0xFFFF) * (1.0f / 65536.f) - 0.5f);
            }
    }
    __syncthreads();
    const long long t0 = wall_clock64();
    const long long s0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {
#pragma unroll
            for (int u = 0; u < 24; ++u) c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
        } else if (MODE == 1) {
#pragma unroll
            for (int u = 0; u < 12; ++u) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
            }
        } else if (MODE == 3) {
#pragma unroll
            for (int u = 0; u < 12; ++u) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ra[u & 3], rb[(u + 1) & 3], c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ra[(u + 2) & 3], rb[u & 3], c1, 0, 0, 0);
            }
            if ((i & 15) == 15) {  // keep the accumulators bounded: fresh start every 16 rounds
#pragma unroll
                for (int r = 0; r < 16; ++r) { c0[r] *= 1e-3f; c1[r] *= 1e-3f; }
            }
        } else {
#pragma unroll
            for (int u = 0; u < 24; ++u) c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf, c0, 0, 0, 0);
        }
    }
    const long long s1 = __builtin_readcyclecounter();
    const long long t1 = wall_clock64();
    float acc = 0;
    for (int r = 0; r < 16; ++r) acc += c0[r] + c1[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0) { cyc[2 * blockIdx.x] = s1 - s0; cyc[2 * blockIdx.x + 1] = t1 - t0; }
}

template <int MODE>
void run(const char* name, int threads, int blocks) {
    float* out; long long* cyc;
    hipMalloc(&out, sizeof(float) * threads * blocks);
    hipMalloc(&cyc, sizeof(long long) * 2 * blocks);
    const int iters = 2000;
    k<MODE><<<blocks, threads>>>(out, cyc, 10, 0.001f);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k<MODE><<<blocks, threads>>>(out, cyc, iters, 0.001f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(2 * blocks);
    hipMemcpy(h.data(), cyc, sizeof(long long) * 2 * blocks, hipMemcpyDeviceToHost);
    const double waves_per_simd = threads / 64 / 4.0;
    const double mfma_per_simd = 24.0 * iters * (waves_per_simd < 1 ? 1 : waves_per_simd);
    printf("%-34s threads/WG %4d blocks %4d: %.3f ms, shader cycles/MFMA/SIMD %.1f, ns/MFMA/SIMD %.2f, wallclk ticks %lld\n", name, threads,
           blocks, ms, (double)h[0] / mfma_per_simd, ms * 1e6 / mfma_per_simd, h[1]);
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int blocks : {1, 256}) {
        run<0>("bf16 32x32x16 one chain", 256, blocks);
        run<1>("bf16 32x32x16 two chains", 256, blocks);
        run<0>("bf16 32x32x16 one chain, 2 waves", 512, blocks);
        run<1>("bf16 32x32x16 two chains, 2 waves", 512, blocks);
        run<3>("bf16 random operands", 256, blocks);
        run<3>("bf16 random operands, 2 waves", 512, blocks);
        run<2>("f32 32x32x2 one chain", 256, blocks);
        run<2>("f32 32x32x2 one chain, 2 waves", 512, blocks);
    }
    return 0;
}
