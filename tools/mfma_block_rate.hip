// Micro-benchmark of the K3 block pattern: a chain of four v_mfma_f32_32x32x16_f16 from a zero accumulator, the 16-way lane
// maximum, one compare + wave vote, with 1..4 waves per SIMD and operands (a) in registers, (b) B fragments re-read from LDS.
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_block_rate.hip -o tools/bin/mfma_block_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>  // 0: operands in registers, 1: B from LDS per block, 2: as 1 without the maximum (uses acc[0] only)
__global__ __launch_bounds__(1024) void k(float* out, int iters, float thr) {
    __shared__ f32x4 ldsb[8 * 4 * 64];
    const int lane = threadIdx.x & 63;
    unsigned x = 0x9E3779B9u * (threadIdx.x + 1) + 12345u * blockIdx.x;
    f16x8 a[4], b[4];
    for (int s = 0; s < 4; ++s)
        for (int e = 0; e < 8; ++e) {
            x = x * 1664525u + 1013904223u; a[s][e] = (_Float16)(((int)(x >> 8) & 0xFFFF) * (1.0f / 65536.f) - 0.5f);
            x = x * 1664525u + 1013904223u; b[s][e] = (_Float16)(((int)(x >> 8) & 0xFFFF) * (1.0f / 65536.f) - 0.5f);
        }
    for (int i = threadIdx.x; i < 8 * 4 * 64; i += blockDim.x) {
        x = x * 1664525u + 1013904223u;
        f16x8 v;
        for (int e = 0; e < 8; ++e) v[e] = (_Float16)(((int)((x >> (e * 3)) & 0xFF)) * (1.0f / 256.f) - 0.5f);
        ldsb[i] = __builtin_bit_cast(f32x4, v);
    }
    __syncthreads();
    int hits = 0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll 1
        for (int g = 0; g < 8; ++g) {
            if (MODE >= 1) {
#pragma unroll
                for (int s = 0; s < 4; ++s) b[s] = __builtin_bit_cast(f16x8, ldsb[(g * 4 + s) * 64 + lane]);
            }
            f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[s], b[s], acc, 0, 0, 0);
            float mx;
            if (MODE == 2) mx = acc[0];
            else {
                const float m0 = __builtin_fmaxf(__builtin_fmaxf(acc[0], acc[1]), acc[2]);
                const float m1 = __builtin_fmaxf(__builtin_fmaxf(acc[3], acc[4]), acc[5]);
                const float m2 = __builtin_fmaxf(__builtin_fmaxf(acc[6], acc[7]), acc[8]);
                const float m3 = __builtin_fmaxf(__builtin_fmaxf(acc[9], acc[10]), acc[11]);
                const float m4 = __builtin_fmaxf(__builtin_fmaxf(acc[12], acc[13]), acc[14]);
                mx = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(m0, m1), m2), __builtin_fmaxf(__builtin_fmaxf(m3, m4), acc[15]));
            }
            if (__any(mx >= thr)) ++hits;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)hits;
}

template <int MODE>
void run(const char* name, int threads, int blocks) {
    float* out;
    hipMalloc(&out, sizeof(float) * threads * blocks);
    const int iters = 400;
    k<MODE><<<blocks, threads>>>(out, 4, 1e30f);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k<MODE><<<blocks, threads>>>(out, iters, 1e30f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double waves_per_simd = threads / 64 / 4.0;
    const double mfma_per_simd = 32.0 * iters * waves_per_simd;
    printf("%-28s waves/SIMD %.0f blocks %4d: %.3f ms, ns/MFMA/SIMD %.2f\n", name, waves_per_simd, blocks, ms, ms * 1e6 / mfma_per_simd);
    hipFree(out);
}

int main() {
    for (int blocks : {1, 256}) {
        for (int threads : {256, 512, 1024}) {
            run<0>("registers + max", threads, blocks);
            run<1>("B from LDS + max", threads, blocks);
            run<2>("B from LDS, no max", threads, blocks);
        }
    }
    return 0;
}
