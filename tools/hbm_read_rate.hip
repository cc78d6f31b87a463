// What a pure READ stream reaches on this GPU: every wave reads contiguous 16 KiB tiles (64 lanes x 16 B x U loads in flight),
// as the single-query scan does, and folds them into one float.  Sweeps the loads in flight per wave (U), the blocks per CU,
// non-temporal vs plain loads, the buffer size, and a double-buffered variant (next tile's loads issued before the fold).
//   hipcc -O3 --offload-arch=gfx950 tools/hbm_read_rate.hip -o lbm_amd/variants/hbm_read_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int U, bool NT, bool DB>
__global__ __launch_bounds__(256) void rd(const f32x4* __restrict__ p, long long n_vec, float* out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long tile_vecs = 64LL * U;
    const long long n_tiles = n_vec / tile_vecs;
    const long long per_block = (n_tiles + gridDim.x - 1) / gridDim.x;
    long long t = (long long)blockIdx.x * per_block + wave;
    const long long t_end = min(n_tiles, (long long)(blockIdx.x + 1) * per_block);
    float acc = 0.f;
    if (!DB) {
        for (; t < t_end; t += 4) {
            f32x4 x[U];
#pragma unroll
            for (int j = 0; j < U; ++j) x[j] = NT ? __builtin_nontemporal_load(&p[t * tile_vecs + j * 64 + lane]) : p[t * tile_vecs + j * 64 + lane];
#pragma unroll
            for (int j = 0; j < U; ++j) acc += x[j].x + x[j].y + x[j].z + x[j].w;
        }
    } else {
        f32x4 x[U], y[U];
        if (t < t_end) {
#pragma unroll
            for (int j = 0; j < U; ++j) x[j] = NT ? __builtin_nontemporal_load(&p[t * tile_vecs + j * 64 + lane]) : p[t * tile_vecs + j * 64 + lane];
        }
        for (; t < t_end; t += 4) {
            const long long tn = t + 4 < t_end ? t + 4 : t;
#pragma unroll
            for (int j = 0; j < U; ++j) y[j] = NT ? __builtin_nontemporal_load(&p[tn * tile_vecs + j * 64 + lane]) : p[tn * tile_vecs + j * 64 + lane];
#pragma unroll
            for (int j = 0; j < U; ++j) acc += x[j].x + x[j].y + x[j].z + x[j].w;
#pragma unroll
            for (int j = 0; j < U; ++j) x[j] = y[j];
        }
    }
    if (acc == 12345.678f) out[0] = acc;
}

template <int U, bool NT, bool DB>
void run(const f32x4* p, long long n_vec, int blocks, float* out, const char* tag) {
    rd<U, NT, DB><<<blocks, 256>>>(p, n_vec, out);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 10;
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) rd<U, NT, DB><<<blocks, 256>>>(p, n_vec, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps;
    printf("%-22s U %2d blocks %4d: %8.1f us  %6.2f TB/s\n", tag, U, blocks, us, (double)n_vec * 16 / us / 1e6);
}

int main(int argc, char** argv) {
    const double gb = argc > 1 ? atof(argv[1]) : 2.56;
    const long long n_vec = (long long)(gb * 1e9 / 16) / 1024 * 1024;
    f32x4* p; float* out;
    if (hipMalloc(&p, n_vec * 16) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMalloc(&out, 4);
    hipMemset(p, 0, n_vec * 16);
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    const int cu = pr.multiProcessorCount;
    printf("buffer %.2f GB, %d CUs\n", n_vec * 16 / 1e9, cu);
    for (int b : {1, 2, 3, 4, 8}) {
        run<16, true, false>(p, n_vec, cu * b, out, "nt");
        run<16, false, false>(p, n_vec, cu * b, out, "plain");
        run<8, true, false>(p, n_vec, cu * b, out, "nt");
        run<32, true, false>(p, n_vec, cu * b, out, "nt");
        run<16, true, true>(p, n_vec, cu * b, out, "nt double-buffered");
        run<8, true, true>(p, n_vec, cu * b, out, "nt double-buffered");
    }
    return 0;
}
