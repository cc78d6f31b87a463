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

// The same 32 x 32 x 64 block from eight v_mfma_f32_16x16x32_f16 (four 16 x 16 accumulators of 4 registers, two k-steps each):
// equal FLOPs and cycles per block; what differs is the clock the chip holds (MI355X_MICROARCH.md, DVFS give-back (7)).
// DUAL: two blocks' chains are issued back to back and their epilogues follow (8 independent accumulators).
typedef float f32x4v __attribute__((ext_vector_type(4)));
template <bool DUAL>
__global__ __launch_bounds__(1024) void k16(float* out, int iters, float thr) {
    unsigned x = 0x9E3779B9u * (threadIdx.x + 1) + 12345u * blockIdx.x;
    f16x8 a[2][2], b[2][2][2];  // a[row half][k-step], b[group parity][col half][k-step]
    for (int u = 0; u < 2; ++u)
        for (int s = 0; s < 2; ++s)
            for (int e = 0; e < 8; ++e) {
                x = x * 1664525u + 1013904223u; a[u][s][e] = (_Float16)(((int)(x >> 8) & 0xFFFF) * (1.0f / 65536.f) - 0.5f);
                for (int p = 0; p < 2; ++p) { x = x * 1664525u + 1013904223u; b[p][u][s][e] = (_Float16)(((int)(x >> 8) & 0xFFFF) * (1.0f / 65536.f) - 0.5f); }
            }
    int hits = 0;
    auto chain = [&](f32x4v (&acc)[4], int p) {
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int t = 0; t < 4; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[t >> 1][s], b[p][t & 1][s], acc[t], 0, 0, 0);
    };
    auto epi = [&](const f32x4v (&acc)[4]) {
        float m[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) m[t] = __builtin_fmaxf(__builtin_fmaxf(acc[t][0], acc[t][1]), __builtin_fmaxf(acc[t][2], acc[t][3]));
        const float mx = __builtin_fmaxf(__builtin_fmaxf(m[0], m[1]), __builtin_fmaxf(m[2], m[3]));
        if (__any(mx >= thr)) ++hits;
    };
    for (int i = 0; i < iters; ++i) {
#pragma unroll 1
        for (int g = 0; g < 8; g += 2) {
            f32x4v acc0[4], acc1[4];
            if (DUAL) {
                chain(acc0, 0);
                chain(acc1, 1);
                epi(acc0);
                epi(acc1);
            } else {
                chain(acc0, 0);
                epi(acc0);
                chain(acc1, 1);
                epi(acc1);
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)hits;
}

// 32x32x16 with two accumulators: block g + 1's chain is issued before block g's epilogue
__global__ __launch_bounds__(1024) void kdual(float* out, int iters, float thr) {
    unsigned x = 0x9E3779B9u * (threadIdx.x + 1) + 12345u * blockIdx.x;
    f16x8 a[4], b[2][4];
    for (int s = 0; s < 4; ++s)
        for (int e = 0; e < 8; ++e) {
            x = x * 1664525u + 1013904223u; a[s][e] = (_Float16)(((int)(x >> 8) & 0xFFFF) * (1.0f / 65536.f) - 0.5f);
            for (int p = 0; p < 2; ++p) { x = x * 1664525u + 1013904223u; b[p][s][e] = (_Float16)(((int)(x >> 8) & 0xFFFF) * (1.0f / 65536.f) - 0.5f); }
        }
    int hits = 0;
    auto chain = [&](int p) {
        f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[s], b[p][s], acc, 0, 0, 0);
        return acc;
    };
    auto epi = [&](const f32x16& acc) {
        const float m0 = __builtin_fmaxf(__builtin_fmaxf(acc[0], acc[1]), acc[2]);
        const float m1 = __builtin_fmaxf(__builtin_fmaxf(acc[3], acc[4]), acc[5]);
        const float m2 = __builtin_fmaxf(__builtin_fmaxf(acc[6], acc[7]), acc[8]);
        const float m3 = __builtin_fmaxf(__builtin_fmaxf(acc[9], acc[10]), acc[11]);
        const float m4 = __builtin_fmaxf(__builtin_fmaxf(acc[12], acc[13]), acc[14]);
        const float mx = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(m0, m1), m2), __builtin_fmaxf(__builtin_fmaxf(m3, m4), acc[15]));
        if (__any(mx >= thr)) ++hits;
    };
    for (int i = 0; i < iters; ++i) {
#pragma unroll 1
        for (int g = 0; g < 8; g += 2) {
            const f32x16 acc0 = chain(0);
            const f32x16 acc1 = chain(1);
            epi(acc0);
            epi(acc1);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)hits;
}

// The same 32 x 32 x 64 block from TWO v_mfma_i32_32x32x32_i8 (int8 operands, 16 per lane and k-step; i32 accumulators) and an
// integer 16-way maximum: half the matrix cycles of the fp16 block.
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(1024) void ki8(float* out, int iters, float thr) {
    unsigned x = 0x9E3779B9u * (threadIdx.x + 1) + 12345u * blockIdx.x;
    i32x4 a[2], b[2];
    for (int s = 0; s < 2; ++s)
        for (int e = 0; e < 4; ++e) {
            x = x * 1664525u + 1013904223u; a[s][e] = (int)x;
            x = x * 1664525u + 1013904223u; b[s][e] = (int)x;
        }
    const int ithr = (int)thr;
    int hits = 0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll 1
        for (int g = 0; g < 8; ++g) {
            i32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int s = 0; s < 2; ++s) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[s], b[s], acc, 0, 0, 0);
            int m = acc[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) m = acc[r] > m ? acc[r] : m;
            if (__any(m >= ithr)) ++hits;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)hits;
}

template <class K>
void run_k(const char* name, K kern, int threads, int blocks) {
    float* out;
    hipMalloc(&out, sizeof(float) * threads * blocks);
    const int iters = 400;
    kern<<<blocks, threads>>>(out, 4, 1e30f);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    kern<<<blocks, threads>>>(out, iters, 1e30f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double waves_per_simd = threads / 64 / 4.0;
    const double mfma_per_simd = 32.0 * iters * waves_per_simd;  // in units of one 32x32x16 MFMA (= two 16x16x32)
    printf("%-28s waves/SIMD %.0f blocks %4d: %.3f ms, ns per 32x32x16-equivalent/SIMD %.2f\n", name, waves_per_simd, blocks, ms, ms * 1e6 / mfma_per_simd);
    hipFree(out);
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
            if (threads <= 512) {
                run_k("32x32x16, 2 accumulators", kdual, threads, blocks);
                run_k("16x16x32 + max", k16<false>, threads, blocks);
                run_k("16x16x32, 2 blocks ahead", k16<true>, threads, blocks);
            }
            run_k("int8 2 x 32x32x32 + max", ki8, threads, blocks);  // (per fp16-MFMA equivalent: one block = 4 units)
        }
    }
    return 0;
}
