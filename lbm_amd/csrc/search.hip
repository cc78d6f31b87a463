// search.hip -- scoring + filtering + top-k for ContextGraph.compile (lb/context_graph.py:104-121).
//
// Fast path (one query, k <= 64):
//   K2  scan_f32_kernel   streams the fp32 copy of the claim matrix once (256 B/claim), fuses the
//                         since_ms / tag masks (+8 B / +8 B per claim), keeps a per-wave top-k
//                         candidate buffer in LDS and emits, per block, every row whose fp32 score is
//                         within LBH_MARGIN of the block's k-th best;
//   K4  select_fast_kernel  (one block) finds the global k-th best fp32 score, keeps every row within
//                         LBH_MARGIN of it, re-scores those rows from the fp64 master matrix in the
//                         reference's exact operation order (lb/latent.py:63: products added left to
//                         right, no FMA) and sorts them by (-score, hash key) like
//                         lb/context_graph.py:120.  The exact top-k is provably inside that set.
// Exact path (fallback when the fp32 candidate set cannot be bounded -- mass near-ties, e.g. an
// empty query makes every score 0.0 -- and for result pages beyond the first 64):
//   scan_exact_kernel / select_exact_kernel do the same with fp64 scores from the master matrix and
//   the full (score, key) total order, optionally starting strictly after a cursor.
//
// This file is compiled with -ffp-contract=off: fp64 arithmetic must round like CPython floats.
// FMAs in the fp32 scan are written explicitly (fmaf).
#include <cmath>
#include <cstdlib>
#include <mutex>

#include "lbh_internal.h"

namespace {

constexpr uint32_t PAD_ROW = 0xFFFFFFFFu;
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct WaveSync {
    __device__ __forceinline__ void operator()() const { __builtin_amdgcn_wave_barrier(); }
};
struct BlockSync {
    __device__ __forceinline__ void operator()() const { __syncthreads(); }
};

// Bitonic sort, descending, of n (power of two) u64 values in LDS by `nthreads` cooperating threads.
template <class Sync>
__device__ __forceinline__ void sort_desc_u64(uint64_t* buf, int n, int tid, int nthreads, Sync sync) {
    for (int k = 2; k <= n; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < (n >> 1); t += nthreads) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const int l = i | j;
                const bool desc = (i & k) == 0;
                const uint64_t a = buf[i], b = buf[l];
                if (desc ? (a < b) : (a > b)) { buf[i] = b; buf[l] = a; }
            }
            sync();
        }
    }
}

// The same sort for ONE wave, in registers: entry i = r * 64 + lane sits in register r of its lane, the compare-exchange
// partner of a step with distance j is another register of the same lane (j >= 64) or the same register of lane ^ j (two
// 32-bit cross-lane moves).  The LDS version above spends ~500 cycles per step on dependent LDS round trips and wave
// barriers (5 us for 64-128 entries, measured with the K2 block timeline); this one is ~20 instructions per step.
// buf[0 .. 64 R) must be readable; entries at and beyond `cnt` are taken as 0 (they sort last).
template <int R>
__device__ __forceinline__ void wave_sort_desc_u64(uint64_t* buf, int cnt, int lane) {
    uint64_t v[R];
#pragma unroll
    for (int r = 0; r < R; ++r) v[r] = (r * 64 + lane) < cnt ? buf[r * 64 + lane] : 0ull;
#pragma unroll
    for (int k = 2; k <= 64 * R; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j >= 64) {
                constexpr int dummy = 0; (void)dummy;
                const int jr = j >> 6;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    if ((r & jr) == 0) {  // r and r | jr are partners; the lower index keeps the larger value when descending
                        const int i = r * 64 + lane;
                        const bool desc = (i & k) == 0;
                        const uint64_t a = v[r], b = v[r | jr];
                        const bool sw = desc ? (a < b) : (a > b);
                        v[r] = sw ? b : a;
                        v[r | jr] = sw ? a : b;
                    }
                }
            } else {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int i = r * 64 + lane;
                    const bool desc = (i & k) == 0;
                    const bool lower = (lane & j) == 0;  // this lane holds the lower index of the pair
                    const uint64_t mine = v[r];
                    const uint32_t olo = (uint32_t)__shfl_xor((int)(uint32_t)mine, j);
                    const uint32_t ohi = (uint32_t)__shfl_xor((int)(uint32_t)(mine >> 32), j);
                    const uint64_t other = ((uint64_t)ohi << 32) | olo;
                    // the lower index keeps max when descending, min when ascending; the upper index the opposite
                    const bool want_max = desc == lower;
                    v[r] = want_max ? (mine > other ? mine : other) : (mine < other ? mine : other);
                }
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r = 0; r < R; ++r) buf[r * 64 + lane] = v[r];
    __builtin_amdgcn_wave_barrier();
}

// n2 (a power of two, 64 .. 256) entries of buf, of which the first cnt are real
__device__ __forceinline__ void wave_sort_desc_dispatch(uint64_t* buf, int cnt, int n2, int lane) {
    if (n2 <= 64) wave_sort_desc_u64<1>(buf, cnt, lane);
    else if (n2 <= 128) wave_sort_desc_u64<2>(buf, cnt, lane);
    else wave_sort_desc_u64<4>(buf, cnt, lane);
}

// hash-key comparison: true when key(ra) < key(rb) (big-endian integer order of the 32 key bytes)
__device__ __forceinline__ bool key_less(const uint64_t* __restrict__ keys, uint32_t ra, uint32_t rb) {
    const uint64_t* a = keys + (size_t)ra * 4;
    const uint64_t* b = keys + (size_t)rb * 4;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const uint64_t x = a[w], y = b[w];
        if (x != y) return x < y;
    }
    return false;
}

// reference order (lb/context_graph.py:120): score descending, then hash ascending
__device__ __forceinline__ bool before_exact(uint64_t sa, uint32_t ra, uint64_t sb, uint32_t rb,
                                             const uint64_t* __restrict__ keys) {
    if (sa != sb) return sa > sb;
    if (ra == rb || ra == PAD_ROW) return false;
    if (rb == PAD_ROW) return true;
    return key_less(keys, ra, rb);
}

template <class Sync>
__device__ __forceinline__ void sort_exact(uint64_t* so, uint32_t* ro, int n, int tid, int nthreads,
                                           const uint64_t* __restrict__ keys, Sync sync) {
    for (int k = 2; k <= n; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < (n >> 1); t += nthreads) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const int l = i | j;
                const bool fwd = (i & k) == 0;
                const uint64_t sa = so[i], sb = so[l];
                const uint32_t ra = ro[i], rb = ro[l];
                const bool swap = fwd ? before_exact(sb, rb, sa, ra, keys) : before_exact(sa, ra, sb, rb, keys);
                if (swap) { so[i] = sb; so[l] = sa; ro[i] = rb; ro[l] = ra; }
            }
            sync();
        }
    }
}

__device__ __forceinline__ int pow2_at_least(int n, int lo) {
    int p = lo;
    while (p < n) p <<= 1;
    return p;
}

// exact reference score: sum(x*y) left to right starting from 0.0 (lb/latent.py:63)
__device__ __forceinline__ double exact_dot(const double* __restrict__ q, const double* __restrict__ row) {
    const double2* r2 = reinterpret_cast<const double2*>(row);
    double acc = 0.0;
#pragma unroll 8
    for (int d = 0; d < 32; ++d) {
        const double2 v = r2[d];
        acc = acc + q[2 * d] * v.x;
        acc = acc + q[2 * d + 1] * v.y;
    }
    return acc;
}

// Same arithmetic for the re-rank of a handful of rows: one load per cache line of the 512-byte row goes out first, so
// the row costs one HBM round trip instead of four in a row (the 8-deep loop above keeps the register count low and
// would otherwise meet a new line, and a new miss, in every group; 32 loads at once spilled under the 1024-thread
// launch bound).
__device__ __forceinline__ double exact_dot_latency(const double* __restrict__ q, const double* __restrict__ row) {
    const double t0 = row[0], t1 = row[16], t2 = row[32], t3 = row[48];
    asm volatile("" ::"v"(t0), "v"(t1), "v"(t2), "v"(t3));
    return exact_dot(q, row);
}

// ------------------------------------------------------------------------------------------------
// K2: fp32 streaming scan.  A wave reads one 64-row tile (16 KiB) per step as 16 fully coalesced
// 1-KiB dwordx4 loads: lane l, load j covers floats [4c, 4c+4) of tile row 4j + g (c = l & 15,
// g = l >> 4).  Four xor-butterfly steps inside each 16-lane row both reduce over c and transpose,
// leaving lane (g, c) with the complete score of tile row 4c + g.  Every row goes through the same
// reduction tree, so bit-identical rows get bit-identical scores wherever they sit.
// ------------------------------------------------------------------------------------------------
//
// HALF = true is the opt-in fp16 flavour (`scan_mode` 1): it streams the 128 B/claim fp16 copy of the matrix instead.
// The copy is stored in MFMA-fragment order (lbh_emb16_off): a 64-row step is 8 KiB = two 32-row tiles x four 1-KiB
// chunks; lane (j = l & 31, h = l >> 5) gets, from chunk s of tile u, the 8 halves dims 16 s + 8 h .. + 7 of row
// 32 u + j.  Each lane sums its 32 products per tile in fp32 (v_dot2_f32_f16), ONE cross-half exchange completes the
// two tiles' scores at once (lane (j, h) ends up with row 32 h + j), and the candidate cut uses the fp16 margin
// (LBH_MARGIN_F16, see scan_batched_tiled_kernel for the bound).  The exact re-rank makes the result identical; only
// the bytes per claim change.
typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
typedef _Float16 h8_t __attribute__((ext_vector_type(8)));

#ifdef LBH_K2_TIMING  // instrumented build (tools/scripts/k2_phases.py): 100 MHz wall-clock stamps of every block's first wave
__device__ long long lbh_k2_dbg[LBH_MAX_BLOCKS][6];
#define K2_STAMP(i) do { if (threadIdx.x == 0) lbh_k2_dbg[blockIdx.x][i] = wall_clock64(); } while (0)
__device__ long long lbh_k4_dbg[16];  // the same for the select kernel's phases (thread 0)
#define K4_STAMP(i) do { if (threadIdx.x == 0) lbh_k4_dbg[i] = wall_clock64(); } while (0)
#else
#define K2_STAMP(i)
#define K4_STAMP(i)
#endif
// first tile of block b under a weighted split (b = grid gives n_tiles): the weight of the blocks below b over the weight
// of all, times the tile count.  Blocks [0, n_first) are the first on their CU, the rest arrive beside an older block.
__host__ __device__ inline uint64_t split_cum(const lbh_split& sp, int b) {
    const int f = b < sp.n_first ? b : sp.n_first;
    const uint64_t ev1 = (uint64_t)((f + 1) / 2), od1 = (uint64_t)(f / 2);                                       // even / odd indices in [0, f)
    const uint64_t ev2 = (uint64_t)((b + 1) / 2 - (f + 1) / 2), od2 = (uint64_t)(b / 2 - f / 2);                // ... in [f, b)
    return ev1 * sp.w[0] + od1 * sp.w[1] + ev2 * sp.w[2] + od2 * sp.w[3];
}
__host__ __device__ inline int64_t split_start(const lbh_split& sp, int b, int grid, int64_t n_tiles) {
    if (b >= grid) return n_tiles;
    return (int64_t)(((uint64_t)n_tiles * split_cum(sp, b)) / split_cum(sp, grid));  // < 2^26 tiles x < 2^27 weight: fits
}

// Maximum over the 64 lanes of a wave, returned in every lane (wave-uniform).  Six DPP steps on the VALU (row shifts, then
// the row broadcasts of GFX9) and one v_readlane: ~60 cycles, where six ds_bpermute exchanges cost ~700
// (tools/dpp_reduce_check.hip checks it against the shuffle reduction on the device).
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#define LBH_DPP_MAX(ctrl)                                                                              \
    {                                                                                                  \
        const uint32_t t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, 0xF, 0xF, false);    \
        v = t > v ? t : v;                                                                             \
    }
    LBH_DPP_MAX(0x111) LBH_DPP_MAX(0x112) LBH_DPP_MAX(0x114) LBH_DPP_MAX(0x118)  // row_shr 1, 2, 4, 8: lane 15 of a row holds the row's maximum
    LBH_DPP_MAX(0x142) LBH_DPP_MAX(0x143)                                        // row_bcast 15, 31: lane 63 holds the wave's
#undef LBH_DPP_MAX
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// Wave-uniform lower bound of the k-th largest of the 64 x XV values the lanes of a wave hold (0 = an empty slot): a t,
// at most 2^PREC ordered-float steps below the k-th largest, such that at least k values are >= t; 0 when fewer than k
// are set.  Any such t is good for a cut, so the search gallops down from the maximum (the k-th largest of a set of
// maxima sits a few 2^-10 of it away) and bisects what is left: a dozen probes of XV compares + wave votes, no sort.
// `floor`: values at or below it are of no use to the caller (its current cut) -- one probe there ends the search early.
template <int XV, int PREC = 8>
__device__ __forceinline__ uint32_t wave_kth_bound(const uint32_t (&v)[XV], int k, uint32_t floor = 0) {
    auto count = [&](uint32_t t) {
        int c = 0;
#pragma unroll
        for (int i = 0; i < XV; ++i) c += __popcll(__ballot(v[i] >= t));
        return c;
    };
    uint32_t hi = 0;
#pragma unroll
    for (int i = 0; i < XV; ++i) hi = v[i] > hi ? v[i] : hi;
    hi = wave_max_u32(hi);
    if (hi <= floor) return 0;
    if (floor && count(floor + 1) < k) return 0;
    if (count(hi) >= k) return hi;
    uint32_t lo = floor;  // count(lo) >= k holds (by the probe above, or by convention for 0: "no bound"); count(hi) < k
    for (uint32_t step = 1u << 14; step; step <<= 2) {
        const uint32_t probe = hi - lo > step ? hi - step : lo;
        if (probe == lo) break;
        if (count(probe) >= k) { lo = probe; break; }
        hi = probe;
    }
    while (hi - lo > (1u << PREC)) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if (count(mid) >= k) lo = mid; else hi = mid;
    }
    return lo;
}

// the select step (K4 body, defined below) as run by the scan's last block: LBH_SCAN_THREADS threads wide
__device__ void fused_select(const lbh_dev_index& ix, const double* qs, int k, const uint64_t* cand, int E, lbh_result* res,
                             float margin, const lbh_xchg& xc);
__device__ __forceinline__ void publish_result(lbh_result* res, int seq);
constexpr int LBH_FLAG_TICKET = 6;  // flags[6..7]: 64-bit ticket word of a fused scan -- low half = blocks done, bit 63 = candidate set unbounded
constexpr unsigned long long LBH_TICKET_OVERFLOW = 1ull << 63;

template <bool SINCE, bool TAGS, bool NT, bool HALF, bool FUSE>
__global__ __launch_bounds__(LBH_SCAN_THREADS) void scan_f32_kernel(lbh_dev_index ix, const uint16_t* __restrict__ emb16,
                                                                     const double* __restrict__ qptr, lbh_qarg qa,
                                                                     int k, float margin, int64_t since_ms, uint64_t tag_mask,
                                                                     uint64_t* __restrict__ cand,
                                                                     int32_t* __restrict__ flags, lbh_split sp,
                                                                     lbh_result* __restrict__ res, int seq, lbh_xchg xc,
                                                                     uint32_t* __restrict__ bcnt, int xb_k) {
    constexpr bool fuse = FUSE;  // the scan's last block also runs the select step (see the tail of this kernel)
    __shared__ __attribute__((aligned(16))) uint64_t lds_all[2 * LBH_SCAN_WAVES * LBH_WCAP];  // the waves' buffers, then the block's merge buffer; the fused select's stage at the end
    uint64_t (*wbuf)[LBH_WCAP] = reinterpret_cast<uint64_t (*)[LBH_WCAP]>(lds_all);
    uint64_t* mbuf = lds_all + LBH_SCAN_WAVES * LBH_WCAP;
    __shared__ int wcnt[LBH_SCAN_WAVES];

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int c = lane & 15, g = lane >> 4;     // fp32 matrix: 16-byte chunk of a row / row within a 4-row load
    const int hj = lane & 31, hh = lane >> 5;   // fp16 copy: row within a 32-row tile / which 8 dims of every k-step
    const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    // query: device pointer (search_text: embedded on the GPU) or the kernel-argument copy (no H2D copy)
    const double* __restrict__ qd = qptr ? qptr : qa.q;
    const f32x4 qf = {(float)qd[4 * (c & 15)], (float)qd[4 * (c & 15) + 1], (float)qd[4 * (c & 15) + 2], (float)qd[4 * (c & 15) + 3]};
    h2_t qh[4][4];  // HALF: this lane's 32 query dims (16 s + 8 hh + 2 e, + 1) as packed fp16, one rounding to nearest even
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            qh[ks][e].x = (_Float16)qd[16 * ks + 8 * hh + 2 * e];
            qh[ks][e].y = (_Float16)qd[16 * ks + 8 * hh + 2 * e + 1];
        }
    }
    const f32x4* __restrict__ E = reinterpret_cast<const f32x4*>(ix.emb32);
    const f32x4* __restrict__ E16 = reinterpret_cast<const f32x4*>(emb16);
    const int64_t n = ix.n;
    const int64_t n_tiles = (n + 63) >> 6;
    const int64_t n_vec = n * 16;  // 16-byte chunks in the fp32 matrix (the fp16 copy is padded to whole 64-row steps)
    uint64_t* buf = wbuf[wave];

    K2_STAMP(0);
    uint32_t thr = 0;  // ordered image of (k-th best so far - margin); 0 admits everything
    // first compaction as soon as a threshold can exist (keeps the warm-up short on small corpora)
    int trigger = 4 * k < 64 ? 64 : (4 * k > LBH_WCAP - 64 ? LBH_WCAP - 64 : 4 * k);
    int cnt = 0;
    bool dead = false;

    // Cross-block bound for large k.  A wave sees only n / (4 x grid) rows, so its own k-th best is a weak threshold when
    // k is large (1 M claims, k = 64: the 64th of 977 rows admits 6.5 % of everything that follows, the buffer is sorted
    // again every ~64 admitted rows, and the blocks hand 16 000+ candidates to the select step).  But every block's BEST
    // score so far is known early, and the k-th largest of the blocks' bests belongs to k different rows: a lower bound
    // of the corpus's k-th best that is already sharp after two or three tiles.  At a compaction -- a slow path anyway --
    // a wave raises its block's slot (bcnt[LBH_MAX_BLOCKS ..]: 64-bit words, launch number << 32 | ordered score, so a
    // slot left by an earlier launch reads as empty and nothing has to be cleared) and takes the k-th largest of the
    // slots as its threshold.  Stale or missing slots only loosen the bound.  Nothing is stored or loaded while streaming.
    __shared__ uint32_t sh_bmax;
    const bool xb = bcnt != nullptr && xb_k > 0 && k >= xb_k && (int)gridDim.x >= k;
    unsigned long long* xslot = reinterpret_cast<unsigned long long*>(bcnt + LBH_MAX_BLOCKS);
    const int xg = (int)gridDim.x < LBH_XB_SLOTS ? (int)gridDim.x : LBH_XB_SLOTS;  // a subset of the blocks is as valid
    uint32_t omax = 0, wpub = 0;  // this lane's best score so far; the wave's best at its last publication
    if (xb) {
        if (threadIdx.x == 0) sh_bmax = 0;
        __syncthreads();
        if (2 * k < trigger) trigger = 2 * k < 64 ? 64 : 2 * k;  // the shared bound does not need 4 k rows of this wave
    }
    auto xb_publish = [&]() {
        const uint32_t wm = wave_max_u32(omax);
        if (lane == 0 && wm > wpub) {
            const uint32_t old = atomicMax(&sh_bmax, wm);
            if (wm > old && (int)blockIdx.x < LBH_XB_SLOTS)
                __hip_atomic_store(&xslot[blockIdx.x], ((unsigned long long)(uint32_t)seq << 32) | wm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        wpub = wm;
    };
    constexpr int XV = LBH_XB_SLOTS / 64;
    auto xb_bound = [&]() -> uint32_t {  // a lower bound of the k-th largest slot (wave_kth_bound); 0 = fewer than k slots set
        uint32_t v[XV];
#pragma unroll
        for (int i = 0; i < XV; ++i) {
            const int b = i * 64 + lane;
            unsigned long long w = 0;
            if (b < xg) w = __hip_atomic_load(&xslot[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            v[i] = (uint32_t)(w >> 32) == (uint32_t)seq ? (uint32_t)w : 0u;
        }
        return wave_kth_bound<XV>(v, k, thr);  // a bound at or below the current threshold is of no use
    };

    // sort the wave buffer, raise the threshold, drop what fell below it
    auto compact = [&](bool in_loop) {
        const int n2 = pow2_at_least(cnt, 64);
        __builtin_amdgcn_wave_barrier();
        const bool ex = xb && in_loop;
        if (ex) xb_publish();
        wave_sort_desc_dispatch(buf, cnt, n2, lane);  // in registers; slots [cnt, n2) come back as 0
        if (ex) {
            const uint32_t bound = xb_bound();
            if (bound) {
                const uint32_t nt = lbh_ord32(lbh_unord32(bound) - margin);
                if (nt > thr) thr = nt;
            }
        }
        if (cnt >= k || ex) {
            if (cnt >= k) {
                const uint32_t kth = (uint32_t)(buf[k - 1] >> 32);
                const uint32_t nt = lbh_ord32(lbh_unord32(kth) - margin);
                if (nt > thr) thr = nt;
            }
            int kept = 0;
            for (int i = lane; i < n2; i += 64)
                kept += __popcll(__ballot(i < cnt && (uint32_t)(buf[i] >> 32) >= thr));
            cnt = kept;
        }
        if (cnt > LBH_WCAP - 64) {  // more near-ties than the buffer holds: let the exact path decide
            if (lane == 0) {
                if (fuse) atomicOr(reinterpret_cast<unsigned long long*>(&flags[LBH_FLAG_TICKET]), LBH_TICKET_OVERFLOW);
                else atomicOr(&flags[0], 1);
            }
            dead = true;
            cnt = 0;
        }
    };

    // this block's tiles: a weighted contiguous range (lbh_split), or the grid-stride loop when the weights are all zero
    int64_t tile = (int64_t)blockIdx.x * LBH_SCAN_WAVES + wave, tile_end = n_tiles, tile_step = (int64_t)gridDim.x * LBH_SCAN_WAVES;
    if (sp.w[0]) {
        const int64_t t0 = split_start(sp, (int)blockIdx.x, (int)gridDim.x, n_tiles);
        tile_end = split_start(sp, (int)blockIdx.x + 1, (int)gridDim.x, n_tiles);
        tile = t0 + wave;
        tile_step = LBH_SCAN_WAVES;
    }
    for (; tile < tile_end && !dead; tile += tile_step) {
        float s;
        int64_t myrow;
        if (!HALF) {
            const int64_t v0 = tile * 1024 + lane;
            f32x4 x[16];
            if (v0 + 1024 - lane <= n_vec) {
#pragma unroll
                for (int j = 0; j < 16; ++j) x[j] = NT ? __builtin_nontemporal_load(&E[v0 + j * 64]) : E[v0 + j * 64];
            } else {
#pragma unroll
                for (int j = 0; j < 16; ++j) x[j] = (v0 + j * 64 < n_vec) ? E[v0 + j * 64] : f32x4{0.f, 0.f, 0.f, 0.f};
            }
            myrow = tile * 64 + 4 * c + g;
            float p[16];
#pragma unroll
            for (int j = 0; j < 16; ++j)
                p[j] = fmaf(x[j].w, qf.w, fmaf(x[j].z, qf.z, fmaf(x[j].y, qf.y, x[j].x * qf.x)));
            // butterfly: after the step with mask m a lane keeps the rows whose bit matches its own
            float r8[8], r4[4], r2[2];
            const bool b8 = c & 8, b4 = c & 4, b2 = c & 2, b1 = c & 1;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float keep = b8 ? p[j + 8] : p[j], send = b8 ? p[j] : p[j + 8];
                r8[j] = keep + __shfl_xor(send, 8);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float keep = b4 ? r8[j + 4] : r8[j], send = b4 ? r8[j] : r8[j + 4];
                r4[j] = keep + __shfl_xor(send, 4);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float keep = b2 ? r4[j + 2] : r4[j], send = b2 ? r4[j] : r4[j + 2];
                r2[j] = keep + __shfl_xor(send, 2);
            }
            const float keep1 = b1 ? r2[1] : r2[0], send1 = b1 ? r2[0] : r2[1];
            s = keep1 + __shfl_xor(send1, 1);
        } else {
            const int64_t v0 = tile * 512 + lane;
            f32x4 x[8];  // 16 raw bytes = 8 halves per load (typed as floats only to reuse the 128-bit load path)
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = NT ? __builtin_nontemporal_load(&E16[v0 + j * 64]) : E16[v0 + j * 64];
            myrow = tile * 64 + 32 * hh + hj;
            float p[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                float a = 0.f;  // fp16 x fp16 products are exact in fp32; fp32 accumulation
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    // whole-vector bit cast, then element pairs: a bit cast of the single floats .x/.y/.z/.w to half pairs
                    // made hipcc 7.2 use word 0 of the chunk four times (seen in the ISA; same family of bug as the
                    // loop-indexed integer vector noted in DESIGN.md)
                    const h8_t xv = __builtin_bit_cast(h8_t, x[u * 4 + ks]);
                    a = __builtin_amdgcn_fdot2(h2_t{xv[0], xv[1]}, qh[ks][0], a, false);
                    a = __builtin_amdgcn_fdot2(h2_t{xv[2], xv[3]}, qh[ks][1], a, false);
                    a = __builtin_amdgcn_fdot2(h2_t{xv[4], xv[5]}, qh[ks][2], a, false);
                    a = __builtin_amdgcn_fdot2(h2_t{xv[6], xv[7]}, qh[ks][3], a, false);
                }
                p[u] = a;
            }
            // lane (j, 0) holds the hh = 0 half of both tiles' row j, lane (j, 1) the other half: each keeps the tile it
            // will report (u = hh) and sends the other one across; the sum is commutative, so a row gets the same score
            // in either half of the wave
            const float keep = hh ? p[1] : p[0], send = hh ? p[0] : p[1];
            s = keep + __shfl_xor(send, 32);
        }
        bool valid = myrow < n;
        if (SINCE) valid = valid && (ix.created[valid ? myrow : 0] >= since_ms);
        if (TAGS) valid = valid && ((ix.tags[myrow < n ? myrow : 0] & tag_mask) != 0);

        const uint32_t o = lbh_ord32(s);
        omax = (valid && o > omax) ? o : omax;
        const bool pass = valid && o >= thr;
        const uint64_t m = __ballot(pass);
        if (m) {
            if (pass) buf[cnt + __popcll(m & lt_mask)] = ((uint64_t)o << 32) | (uint32_t)myrow;
            cnt += __popcll(m);
            __builtin_amdgcn_wave_barrier();
            if (cnt >= trigger) {
                compact(true);
                trigger = cnt + 64 < LBH_WCAP - 64 + 1 ? cnt + 64 : LBH_WCAP - 64 + 1;
            }
        }
    }
    K2_STAMP(1);
    // the other blocks' cuts (see the block merge below): a device-scope load takes 1-2 us to come back, so it goes out now
    uint32_t gseen = 0;
    if (wave == 0) gseen = __hip_atomic_load(reinterpret_cast<unsigned int*>(&flags[64]) + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // Final cut of the wave's buffer: no sort (the block's merged list is sorted below anyway) -- a lower bound of the k-th
    // best of the <= 256 entries from a dozen wave votes (wave_kth_bound; the register sort's cross-lane exchanges took
    // 1.1-1.6 us at the end of every block), survivors stay in registers until the block knows where they go.
    uint64_t fe[LBH_WCAP / 64];
    bool fk[LBH_WCAP / 64];
#pragma unroll
    for (int r = 0; r < LBH_WCAP / 64; ++r) { fe[r] = 0; fk[r] = false; }
    if (!dead && cnt > 0) {
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < LBH_WCAP / 64; ++r) fe[r] = (r * 64 + lane) < cnt ? buf[r * 64 + lane] : 0ull;
        if (cnt >= k) {
            uint32_t b;
            if (cnt <= 128) {
                // the usual case: the exact k-th best by rank counting from broadcast LDS reads (entries are unique) -- the
                // lane that finds k - 1 entries above one of its own holds it
                int r0 = 0, r1 = 0;
                int j = 0;
                for (; j + 4 <= cnt; j += 4) {
                    const uint64_t a = buf[j], bb = buf[j + 1], c2 = buf[j + 2], d = buf[j + 3];
                    r0 += (int)(a > fe[0]) + (int)(bb > fe[0]) + (int)(c2 > fe[0]) + (int)(d > fe[0]);
                    r1 += (int)(a > fe[1]) + (int)(bb > fe[1]) + (int)(c2 > fe[1]) + (int)(d > fe[1]);
                }
                for (; j < cnt; ++j) {
                    const uint64_t a = buf[j];
                    r0 += (int)(a > fe[0]);
                    r1 += (int)(a > fe[1]);
                }
                const bool h0 = lane < cnt && r0 == k - 1, h1 = lane + 64 < cnt && r1 == k - 1;
                const uint32_t mine = h0 ? (uint32_t)(fe[0] >> 32) : (h1 ? (uint32_t)(fe[1] >> 32) : 0u);
                b = wave_max_u32(mine);  // exactly one lane holds a non-zero value
            } else {
                uint32_t v4[LBH_WCAP / 64];
#pragma unroll
                for (int r = 0; r < LBH_WCAP / 64; ++r) v4[r] = (uint32_t)(fe[r] >> 32);
                b = wave_kth_bound<LBH_WCAP / 64>(v4, k);
            }
            if (b) {
                const uint32_t nt = lbh_ord32(lbh_unord32(b) - margin);
                if (nt > thr) thr = nt;
            }
        }
        int kept = 0;
#pragma unroll
        for (int r = 0; r < LBH_WCAP / 64; ++r) {
            fk[r] = (r * 64 + lane) < cnt && (uint32_t)(fe[r] >> 32) >= thr;
            kept += __popcll(__ballot(fk[r]));
        }
        cnt = kept;
        if (cnt > LBH_WCAP - 64) {  // more near-ties than the buffer holds: let the exact path decide
            if (lane == 0) {
                if (fuse) atomicOr(reinterpret_cast<unsigned long long*>(&flags[LBH_FLAG_TICKET]), LBH_TICKET_OVERFLOW);
                else atomicOr(&flags[0], 1);
            }
            cnt = 0;
#pragma unroll
            for (int r = 0; r < LBH_WCAP / 64; ++r) fk[r] = false;
        }
    } else {
        cnt = 0;
    }
    K2_STAMP(2);
    if (lane == 0) wcnt[wave] = cnt;
    __syncthreads();
    K2_STAMP(3);

    // block merge: concatenate the wave lists, sort, keep everything within the margin of the block's k-th
    int off = 0, total = 0;
#pragma unroll
    for (int w = 0; w < LBH_SCAN_WAVES; ++w) {
        if (w < wave) off += wcnt[w];
        total += wcnt[w];
    }
    {
        int pos = off;
#pragma unroll
        for (int r = 0; r < LBH_WCAP / 64; ++r) {
            const uint64_t m = __ballot(fk[r]);
            if (fk[r]) mbuf[pos + __popcll(m & lt_mask)] = fe[r];
            pos += __popcll(m);
        }
    }
    const int n2 = pow2_at_least(total, 64);
    for (int i = total + threadIdx.x; i < n2; i += LBH_SCAN_THREADS) mbuf[i] = 0;
    __syncthreads();
    int my_ns = 0;  // entries this block emits (valid in thread 0)
    if (total > 0) {
    // A block's k-th best score is a lower bound of the corpus's, so (k-th - margin) of ANY block may cut every block's
    // output.  flags[64..127] are 64 slots of such cuts (ordered images; zeroed by the select step): a block stores
    // its own cut into slot (block % 64) with a plain device-scope store and applies the largest of the 64 values it then
    // reads.  Nothing here needs to be exact -- a lost or stale slot is merely a looser, still valid cut -- so there is
    // no atomic (one shared word raised by atomicMax made every finishing block queue behind all the others: the 1 M-claim
    // scan went from 56 to 66 us).  Late blocks emit almost nothing, and the select step sorts a few hundred entries
    // instead of radix-selecting among #blocks x k.
    unsigned int* gslot = reinterpret_cast<unsigned int*>(&flags[64]);
    // The slots are read FIRST (the load is in flight while the merged list is sorted) and the block's own cut is stored
    // LAST, with nothing waiting for it: a device-scope 4-byte store takes microseconds to complete, and behind it in the
    // same wave's memory queue the load of the slots cost every block 5-8 us of its tail.
    // (issued by wave 0 right after its loop, ahead of the final cut and the block's barriers: see gseen above)
    auto shared_cut = [&](uint32_t own) -> uint32_t {  // wave 0 only
        const uint32_t g = wave_max_u32(gseen);
        return g > own ? g : own;
    };
    auto publish_cut = [&](uint32_t own) {
        if (lane == 0) __hip_atomic_store(&gslot[blockIdx.x & 63], own, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    // Where the survivors go.  Two-kernel flavour: appended to ONE flat array (one atomicAdd per block for the position).
    // Fused flavour: into the block's OWN region of the array, LBH_LIST_CAP entries apart -- no atomic, no round trip
    // before the stores can go out -- written through to memory (device-coherent stores), because the block that reads
    // them runs on another XCD, whose L2 is not coherent with this one's.
    // (entry-major: entry j of block b sits at cand[j * LBH_MAX_BLOCKS + b], so that the last block's loads -- one lane
    // per block -- are coalesced; a block's own 8-12 stores go to one line each either way)
    uint64_t* region = cand + blockIdx.x;
#ifndef LBH_SCAN_WAVE_EPI
#define LBH_SCAN_WAVE_EPI 1
#endif
    if (LBH_SCAN_WAVE_EPI && total <= 128) {
        // the usual case (each wave kept ~k rows): one wave sorts the merged list, no more block barriers
        if (wave == 0) {
            // sort by rank counting: a lane holds entries lane and lane + 64 and counts the entries above each from
            // broadcast LDS reads (independent loads, pipelined; entries are unique, so the ranks are a permutation), then
            // writes its entries at their ranks into the other half of the merge buffer.  ~40 entries for top-8: 0.3 us,
            // where the bitonic network's cross-lane exchanges took 1.1-1.8 (a v_readlane loop: 1.6 -- every step waits
            // for the scalar result of the one before).
            {
                const uint64_t e0 = lane < total ? mbuf[lane] : 0ull, e1 = lane + 64 < total ? mbuf[lane + 64] : 0ull;
                int r0 = 0, r1 = 0;
                int j = 0;
                for (; j + 4 <= total; j += 4) {
                    const uint64_t a = mbuf[j], b = mbuf[j + 1], c2 = mbuf[j + 2], d = mbuf[j + 3];
                    r0 += (int)(a > e0) + (int)(b > e0) + (int)(c2 > e0) + (int)(d > e0);
                    r1 += (int)(a > e1) + (int)(b > e1) + (int)(c2 > e1) + (int)(d > e1);
                }
                for (; j < total; ++j) {
                    const uint64_t a = mbuf[j];
                    r0 += (int)(a > e0);
                    r1 += (int)(a > e1);
                }
                uint64_t* srt = mbuf + 2 * LBH_WCAP;  // the merge buffer holds 4 x LBH_WCAP entries; this path uses <= 128
                if (lane < total) srt[r0] = e0;
                if (lane + 64 < total) srt[r1] = e1;
                __builtin_amdgcn_wave_barrier();
                mbuf = srt;
            }
            uint32_t own = 0;
            if (total >= k) own = lbh_ord32(lbh_unord32((uint32_t)(mbuf[k - 1] >> 32)) - margin);
            const uint32_t bthr = shared_cut(own);
            int ns = 0;  // survivors are a prefix of the sorted list
            for (int i = lane; i < n2; i += 64) ns += __popcll(__ballot(i < total && (uint32_t)(mbuf[i] >> 32) >= bthr));
            K2_STAMP(4);
            if (fuse) {
                for (int i = lane; i < ns; i += 64) __hip_atomic_store(&region[(size_t)i * LBH_MAX_BLOCKS], mbuf[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                int base = 0;
                if (lane == 0 && ns > 0) base = atomicAdd(&flags[1], ns);
                base = __shfl(base, 0);
                for (int i = lane; i < ns; i += 64) cand[base + i] = mbuf[i];
            }
            my_ns = ns;
            if (total >= k) publish_cut(own);
            if (!fuse && bcnt && lane == 0 && (int)blockIdx.x < LBH_BBEST_SLOTS)  // the block's best score, for the select kernel's cut
                reinterpret_cast<unsigned long long*>(bcnt)[blockIdx.x] = ((unsigned long long)(uint32_t)seq << 32) | (uint32_t)(mbuf[0] >> 32);
            K2_STAMP(5);
        }
    } else {
        sort_desc_u64(mbuf, n2, threadIdx.x, LBH_SCAN_THREADS, BlockSync());
        uint32_t bthr = 0;
        if (total >= k) bthr = lbh_ord32(lbh_unord32((uint32_t)(mbuf[k - 1] >> 32)) - margin);
        const uint32_t own_cut = bthr;
        // survivors are a prefix of the sorted array
        __shared__ int nsurv, gbase;
        __shared__ uint32_t sh_bthr;
        if (wave == 0) {
            const uint32_t cut = shared_cut(bthr);
            if (lane == 0) { nsurv = 0; sh_bthr = cut; }
        }
        __syncthreads();
        bthr = sh_bthr;  // one value for the whole block
        for (int i = threadIdx.x; i < total; i += LBH_SCAN_THREADS) {
            const bool in = (uint32_t)(mbuf[i] >> 32) >= bthr;
            const bool next_in = (i + 1 < total) && (uint32_t)(mbuf[i + 1] >> 32) >= bthr;
            if (in && !next_in) nsurv = i + 1;
        }
        __syncthreads();
        int ns = nsurv;
        if (ns > LBH_LIST_CAP) {
            if (threadIdx.x == 0) {
                if (fuse) atomicOr(reinterpret_cast<unsigned long long*>(&flags[LBH_FLAG_TICKET]), LBH_TICKET_OVERFLOW);
                else atomicOr(&flags[0], 1);
            }
            ns = LBH_LIST_CAP;
        }
        if (fuse) {
            for (int i = threadIdx.x; i < ns; i += LBH_SCAN_THREADS) __hip_atomic_store(&region[(size_t)i * LBH_MAX_BLOCKS], mbuf[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            if (threadIdx.x == 0) gbase = atomicAdd(&flags[1], ns);
            __syncthreads();
            const int base = gbase;
            for (int i = threadIdx.x; i < ns; i += LBH_SCAN_THREADS) cand[base + i] = mbuf[i];
        }
        my_ns = ns;
        if (wave == 0 && total >= k) publish_cut(own_cut);
        if (!fuse && bcnt && threadIdx.x == 0 && (int)blockIdx.x < LBH_BBEST_SLOTS)
            reinterpret_cast<unsigned long long*>(bcnt)[blockIdx.x] = ((unsigned long long)(uint32_t)seq << 32) | (uint32_t)(mbuf[0] >> 32);
    }
    }
    if (!fuse) return;
    // ---- fused select: the block that finishes LAST cuts the candidate lists and re-ranks, in this very launch.  No
    // select kernel follows a fused scan: its launch hand-off (4-6 us on the latency path of every query) is gone.
    // Every block writes its entry count next to its entries, waits until its stores are acknowledged, then takes a
    // ticket; the holder of the last ticket goes on.  No fence anywhere: an agent-scope release would write back the whole
    // L2 of this XCD (measured: +25 us at 1 M claims, +75 us at 10 M); the entries were stored write-through instead and
    // are read with device-coherent loads.  Each of those is a ~2 us round trip to the memory side, so the tail is built
    // to need as few DEPENDENT ones as possible: acknowledgements -> ticket (its return value carries the overflow flag)
    // -> ONE round of loads (every block's count and the head of its list, the cut slots).
    __shared__ int sh_last, sh_stage, sh_over;
    unsigned long long* tw = reinterpret_cast<unsigned long long*>(&flags[LBH_FLAG_TICKET]);
    if (threadIdx.x == 0) __hip_atomic_store(&bcnt[blockIdx.x], (uint32_t)my_ns, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __atomic_signal_fence(__ATOMIC_SEQ_CST);
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): expcnt / lgkmcnt fields left at their maxima
    __atomic_signal_fence(__ATOMIC_SEQ_CST);
    __syncthreads();
    const int G = (int)gridDim.x;
    if (threadIdx.x == 0) {
        const unsigned long long old = atomicAdd(tw, 1ull);
        sh_last = (int)(old & 0xFFFFFFFFull) == G - 1;
        sh_over = (old & LBH_TICKET_OVERFLOW) != 0;
        sh_stage = 0;
    }
    __syncthreads();
    if (!sh_last) return;
    K4_STAMP(0);
    const bool overflowed = sh_over;
    // One round of loads: thread t takes block t -- its count and the first H entries of its list (sorted descending, so
    // entry 0 is the block's best score); grids beyond one block per thread are handled in further rounds below.
    constexpr int H = 8;
    uint32_t c0 = 0;
    uint64_t h[H];
#pragma unroll
    for (int j = 0; j < H; ++j) h[j] = 0;
    if ((int)threadIdx.x < G) {
        c0 = __hip_atomic_load(&bcnt[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int j = 0; j < H; ++j)
            h[j] = __hip_atomic_load(&cand[(size_t)j * LBH_MAX_BLOCKS + threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // the shared cut slots: every wave reads all 64 (one per lane) and reduces them itself -- no LDS hand-over
    uint32_t gcut = __hip_atomic_load(reinterpret_cast<unsigned int*>(&flags[64]) + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // every block's best score (0: it emitted nothing); the first LBH_XB_SLOTS blocks serve for the bound (a subset of
    // the groups gives a valid, slightly lower bound)
    __shared__ uint32_t bm[LBH_XB_SLOTS];
    const int GB = G < LBH_XB_SLOTS ? G : LBH_XB_SLOTS;
    for (int b = threadIdx.x + LBH_SCAN_THREADS; b < GB; b += LBH_SCAN_THREADS) {
        const uint32_t cb = __hip_atomic_load(&bcnt[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint64_t e0 = __hip_atomic_load(&cand[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bm[b] = cb ? (uint32_t)(e0 >> 32) : 0u;
    }
#ifdef LBH_K2_TIMING
    __builtin_amdgcn_s_waitcnt(0x0F70);
    K4_STAMP(11);  // the round of loads is back
#endif
    if ((int)threadIdx.x < GB) bm[threadIdx.x] = c0 ? (uint32_t)(h[0] >> 32) : 0u;
    // Two cuts, both lower bounds of (corpus k-th best - margin), applied while the lists are staged in LDS:
    //  (1) the largest block cut in the slots (every block has published by now);
    //  (2) the k-th largest of the blocks' BEST scores: those are k different rows, so the corpus's k-th best is at least
    //      that -- a bound only a few ranks below it, where a block's own k-th best sits hundreds of ranks down
    //      (1 M claims, top-8: ~1 300 entries in the lists, ~700 reach (1), ~10 reach (2); top-64: 16 000 -> ~80).
    // The true top-k rows pass every such cut, so whatever else is dropped cannot change the result.  Every wave works
    // both out for itself from the same values (wave_kth_bound: a dozen probes), so one barrier -- bm complete -- is all
    // the block needs here.
    gcut = wave_max_u32(gcut);
    __syncthreads();
    // (2) only matters above (1): the search starts with one probe at the score (1) stands for, and stops at 2^-12 of the
    // range (a cut 2^-13 looser than it might be lets no extra row through in practice)
    const uint32_t cut_floor = gcut ? lbh_ord32(lbh_unord32(gcut) + margin) : 0u;
    uint32_t kth_of_maxima;
    if (GB <= 256) {  // one block per CU: four values per lane, half the work per probe
        uint32_t bv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) bv[i] = (i * 64 + lane) < GB ? bm[i * 64 + lane] : 0u;
        kth_of_maxima = wave_kth_bound<4, 12>(bv, k, cut_floor);
    } else {
        uint32_t bv[LBH_XB_SLOTS / 64];
#pragma unroll
        for (int i = 0; i < LBH_XB_SLOTS / 64; ++i) bv[i] = (i * 64 + lane) < GB ? bm[i * 64 + lane] : 0u;
        kth_of_maxima = wave_kth_bound<LBH_XB_SLOTS / 64, 12>(bv, k, cut_floor);
    }
    const uint32_t precut2 = kth_of_maxima ? lbh_ord32(lbh_unord32(kth_of_maxima) - margin) : 0u;
    K4_STAMP(12);  // cuts known
#ifdef LBH_K2_TIMING
    // the same computation twice more through ONE copy of the code: the first pass fetches it, the second finds it cached
#pragma unroll 1
    for (int rep = 0; rep < 2; ++rep) {
        uint32_t bv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) bv[i] = (i * 64 + lane) < GB ? bm[i * 64 + lane] : 0u;
        asm volatile("" : "+v"(bv[0]), "+v"(bv[1]), "+v"(bv[2]), "+v"(bv[3]));
        const uint32_t again = wave_kth_bound<4, 12>(bv, k + (rep >> 4), cut_floor);
        if (again == 0xFFFFFFFFu) sh_stage = 1;  // never: keeps the computation alive
        if (threadIdx.x == 0) lbh_k4_dbg[13 + rep] = wall_clock64();
    }
#endif
    constexpr int STAGE_CAP = 2 * LBH_SCAN_WAVES * LBH_WCAP;
    if (!overflowed) {
        const uint32_t precut = gcut > precut2 ? gcut : precut2;
        auto push = [&](uint64_t v) {
            const int pos = atomicAdd(&sh_stage, 1);
            if (pos < STAGE_CAP) lds_all[pos] = v;
        };
#pragma unroll
        for (int j = 0; j < H; ++j)
            if (j < (int)c0 && (uint32_t)(h[j] >> 32) >= precut) push(h[j]);
        // the rest of a list whose head passed whole, and the lists of the blocks beyond the first round (rare both)
        for (int b = threadIdx.x; b < G; b += LBH_SCAN_THREADS) {
            int j0 = 0;
            uint32_t cb = c0;
            if (b == (int)threadIdx.x) {
                if ((int)c0 <= H || (uint32_t)(h[H - 1] >> 32) < precut) continue;
                j0 = H;
            } else {
                cb = __hip_atomic_load(&bcnt[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            for (int j = j0; j < (int)cb; ++j) {
                const uint64_t v = __hip_atomic_load(&cand[(size_t)j * LBH_MAX_BLOCKS + b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((uint32_t)(v >> 32) < precut) break;  // sorted descending: nothing further down passes
                push(v);
            }
        }
    }
    __syncthreads();
    K4_STAMP(1);
#ifdef LBH_K2_TIMING
    if (threadIdx.x == 0) { lbh_k4_dbg[8] = G; lbh_k4_dbg[10] = sh_stage; }
#endif
    // the ticket word and the shared cuts start the next launch at zero
    if (threadIdx.x == 0) __hip_atomic_store(tw, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (threadIdx.x >= 64 && threadIdx.x < 128) flags[threadIdx.x] = 0;
    if (overflowed || sh_stage > STAGE_CAP) {
        // mass near-ties: more rows within the margin of the k-th best than the buffers hold -- the exact path decides
        if (threadIdx.x == 0) { res->count = 0; res->overflow = 1; res->candidates = 0; }
        publish_result(res, seq);
        return;
    }
    __shared__ double fq[64];
    if (threadIdx.x < 64) fq[threadIdx.x] = qd[threadIdx.x];
    __syncthreads();
    fused_select(ix, fq, k, lds_all, sh_stage, res, margin, xc);
    K4_STAMP(6);
    publish_result(res, seq);
    K4_STAMP(7);
}

// ------------------------------------------------------------------------------------------------
// radix select over a flat array of E entries: returns the `krem`-th largest value among the
// entries for which val(e, &v) is true.  All threads of the block call it; result is block-uniform.
// ------------------------------------------------------------------------------------------------
constexpr int SEL_THREADS = 1024;
constexpr int LBH_SORT_ALL = 1024;  // select_fast_body: candidate lists up to this size are sorted whole
constexpr int LBH_RANK_ALL = 768;   // ... and up to this size the k-th best is found by rank counting, without a sort
constexpr int LBH_SEL_REGS = 24;    // ... and up to 24 x 1024 entries the scores are selected from registers (28 spills under the 1024-thread bound)

// Core: `each(body)` calls body(v) for every value of the calling thread that is in play (the values may sit in global
// memory, in LDS or in registers).
template <int NT = SEL_THREADS, class Each>
__device__ uint64_t block_radix_select_each(int nbytes, int64_t krem, Each each, int* hist, int64_t* sh_krem, int* sh_bin) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // value range first: digits are taken from (v - lo), so clustered values (scores share their high
    // bits) spread over the bins instead of serialising every LDS atomic on one bin
    __shared__ uint64_t sh_lo[NT / 64], sh_hi[NT / 64];
    uint64_t lo = ~0ull, hi = 0;
    each([&](uint64_t v) { lo = v < lo ? v : lo; hi = v > hi ? v : hi; });
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const uint64_t l2 = __shfl_xor(lo, off), h2 = __shfl_xor(hi, off);
        lo = l2 < lo ? l2 : lo;
        hi = h2 > hi ? h2 : hi;
    }
    if (lane == 0) { sh_lo[wave] = lo; sh_hi[wave] = hi; }
    __syncthreads();
    for (int w = 0; w < NT / 64; ++w) {
        lo = sh_lo[w] < lo ? sh_lo[w] : lo;
        hi = sh_hi[w] > hi ? sh_hi[w] : hi;
    }
    if (hi < lo) return 0;  // nothing in play
    const uint64_t range = hi - lo;
    nbytes = range ? (64 - __clzll(range) + 7) / 8 : 1;
    uint64_t prefix = 0, pmask = 0;
    for (int b = nbytes - 1; b >= 0; --b) {
        for (int t = tid; t < 256; t += NT) hist[t] = 0;
        __syncthreads();
        each([&](uint64_t v) {
            if (((v - lo) & pmask) == prefix) atomicAdd(&hist[(int)(((v - lo) >> (8 * b)) & 255)], 1);
        });
        __syncthreads();
        if (wave == 0) {  // wave-parallel search for the bin holding the krem-th largest value
            const int h0 = hist[4 * lane], h1 = hist[4 * lane + 1], h2 = hist[4 * lane + 2], h3 = hist[4 * lane + 3];
            const int s4 = h0 + h1 + h2 + h3;
            int suf = s4;  // inclusive suffix sum over lanes (higher lanes = higher bins)
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int t = __shfl_down(suf, off);
                if (lane + off < 64) suf += t;
            }
            const int64_t above = suf - s4;
            if (above < krem && (int64_t)suf >= krem) {
                int64_t rem = krem - above;
                int bin;
                if (h3 >= rem) bin = 4 * lane + 3;
                else if (h3 + h2 >= rem) { bin = 4 * lane + 2; rem -= h3; }
                else if (h3 + h2 + h1 >= rem) { bin = 4 * lane + 1; rem -= h3 + h2; }
                else { bin = 4 * lane; rem -= h3 + h2 + h1; }
                *sh_krem = rem;
                *sh_bin = bin;
            }
        }
        __syncthreads();
        krem = *sh_krem;
        prefix |= (uint64_t)(*sh_bin) << (8 * b);
        pmask |= 0xFFull << (8 * b);
        __syncthreads();
    }
    return prefix + lo;
}

// The same over E indexed entries: val(e, &v) says whether entry e is in play and yields its value.
template <int NT = SEL_THREADS, class F>
__device__ uint64_t block_radix_select(int E, int nbytes, int64_t krem, F val, int* hist, int64_t* sh_krem,
                                       int* sh_bin) {
    auto each = [&](auto body) {
        for (int e = threadIdx.x; e < E; e += NT) {
            uint64_t v;
            if (val(e, &v)) body(v);
        }
    };
    return block_radix_select_each<NT>(nbytes, krem, each, hist, sh_krem, sh_bin);
}

// shared tail of both final kernels: rows xr[0..M) (+ ord64 scores xo when have_scores) -> exact
// scores -> exact sort -> result block
template <int NT = SEL_THREADS>
__device__ void rerank_and_emit(uint64_t* xo, uint32_t* xr, int M, int k, const lbh_dev_index& ix,
                                const double* qs, lbh_result* __restrict__ res, bool have_scores, const lbh_xchg& xc,
                                const uint32_t* scan_ord = nullptr,   // scan_ord[i]: fp32 score image of row tid + i * NT
                                int out_off = 0) {                    // result page: ranks [out_off, out_off + LBH_MAX_PAGE) of the top-k are written
    const int tid = threadIdx.x;
    const int n2 = pow2_at_least(M, 2);
    __shared__ uint32_t sh_err;  // largest |scan score - exact score| among the re-scored rows (float bits, >= 0)
    if (tid == 0) sh_err = 0;
    __syncthreads();
    if (!have_scores) {
#pragma unroll
        for (int i = 0; i < (LBH_MCAP + NT - 1) / NT; ++i) {
            const int j = tid + i * NT;
            if (j < M) {
                const double exact = exact_dot_latency(qs, ix.emb64 + (size_t)xr[j] * 64);
                xo[j] = lbh_ord64(exact);
                const float err = (float)__builtin_fabs((double)lbh_unord32(scan_ord[i]) - exact);
                atomicMax(&sh_err, __builtin_bit_cast(uint32_t, err));
            }
        }
    }
    for (int j = M + tid; j < n2; j += NT) { xo[j] = 0; xr[j] = PAD_ROW; }
    __syncthreads();
    K4_STAMP(3);  // exact scores
    if (n2 <= 128) {  // the usual case (M ~ k): one wave sorts without block barriers
        if (tid < 64) sort_exact(xo, xr, n2, tid, 64, ix.keys, WaveSync());
        __syncthreads();
    } else {
        sort_exact(xo, xr, n2, tid, NT, ix.keys, BlockSync());
    }
    K4_STAMP(4);  // sorted
    const int cnt_all = M < k ? M : k;
    int cnt = cnt_all - out_off;
    cnt = cnt < 0 ? 0 : (cnt > LBH_MAX_PAGE ? LBH_MAX_PAGE : cnt);
    xo += out_off;
    xr += out_off;
    if (tid < cnt) {
        const uint32_t r = xr[tid];
        res->rows[tid] = r;
        res->scores[tid] = lbh_unord64(xo[tid]);
#pragma unroll
        for (int w = 0; w < 4; ++w) res->keys[tid][w] = ix.keys[(size_t)r * 4 + w];
        if (xc.dst) {  // the same result as the packed list the ranks exchange, left in device memory
            int64_t* e = xc.dst + 6 * (tid + 1);
            e[0] = (int64_t)__builtin_bit_cast(uint64_t, lbh_unord64(xo[tid]));
            e[1] = (int64_t)((ix.ids ? ix.ids[r] : (uint64_t)r) * xc.gid_mul + xc.gid_add);
#pragma unroll
            for (int w = 0; w < 4; ++w) e[2 + w] = (int64_t)ix.keys[(size_t)r * 4 + w];
        }
    }
    if (tid == 0) {
        res->count = cnt; res->overflow = 0; res->candidates = M; res->max_err = __builtin_bit_cast(float, sh_err);
        if (xc.dst) xc.dst[0] = cnt;
    }
}

// three-way hash-key comparison: < 0 when key(ra) < key(rb)
__device__ __forceinline__ int key_cmp3(const uint64_t* __restrict__ keys, uint32_t ra, uint32_t rb) {
    const uint64_t* a = keys + (size_t)ra * 4;
    const uint64_t* b = keys + (size_t)rb * 4;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const uint64_t x = a[w], y = b[w];
        if (x != y) return x < y ? -1 : 1;
    }
    return 0;
}

// Select step for a SHORT candidate list (E <= 128; `cand` in LDS or global memory): ONE wave, no LDS, no barrier.  Every
// candidate is re-scored exactly -- a lane takes entries lane and lane + 64, the loads of both rows go out together --
// so no threshold is needed; a lane then counts the entries that come before its own in the reference's order
// (score descending, key ascending; row index last, so that even duplicate keys get distinct ranks) from wave-uniform
// broadcasts, and writes its entry at that rank if the rank is below k.  The fused scan's last block stages 8-80 entries
// for top-8 .. top-64, so this is its usual path: 3.5 us of barriers, LDS sort and re-packing became ~2.
// Threads other than the first wave fall through; the caller's publish_result has the barrier.
__device__ void select_small(const lbh_dev_index& ix, const double* qs, int k, const uint64_t* cand, int E,
                             lbh_result* __restrict__ res, const lbh_xchg& xc) {
    if (threadIdx.x >= 64) return;
    const int lane = threadIdx.x;
    uint64_t sc[2] = {0, 0};
    uint32_t row[2] = {PAD_ROW, PAD_ROW}, so[2] = {0, 0};
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int e = lane + 64 * r;
        if (e < E) {
            const uint64_t v = cand[e];
            row[r] = (uint32_t)v;
            so[r] = (uint32_t)(v >> 32);
        }
    }
    uint64_t kw[2][4];  // the rows' hash keys: needed last (result block, ties), requested first
    {   // one load per cache line of both rows and their keys go out together: everything costs ONE HBM round trip
        const uint32_t s0 = row[0] == PAD_ROW ? 0u : row[0], s1 = row[1] == PAD_ROW ? 0u : row[1];
        const double* r0 = ix.emb64 + (size_t)s0 * 64;
        const double* r1 = ix.emb64 + (size_t)s1 * 64;
        const double a0 = r0[0], a1 = r0[16], a2 = r0[32], a3 = r0[48], b0 = r1[0], b1 = r1[16], b2 = r1[32], b3 = r1[48];
#pragma unroll
        for (int w = 0; w < 4; ++w) { kw[0][w] = ix.keys[(size_t)s0 * 4 + w]; kw[1][w] = ix.keys[(size_t)s1 * 4 + w]; }
        asm volatile("" ::"v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(b0), "v"(b1), "v"(b2), "v"(b3));
    }
    float err = 0.f;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        if (row[r] != PAD_ROW) {
            const double exact = exact_dot(qs, ix.emb64 + (size_t)row[r] * 64);
            sc[r] = lbh_ord64(exact);
            const float d = (float)__builtin_fabs((double)lbh_unord32(so[r]) - exact);
            err = d > err ? d : err;
        }
    }
    int rank[2] = {0, 0};
    for (int j = 0; j < E; ++j) {  // wave-uniform: entry j is broadcast from the lane and register that hold it
        const int jl = j & 63;
        const uint64_t src_s = j < 64 ? sc[0] : sc[1];
        const uint32_t src_r = j < 64 ? row[0] : row[1];
        // (v_readlane with a uniform lane number: a few cycles each; ds_bpermute would make this loop 3 LDS round trips long)
        const uint64_t sj = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(src_s >> 32), jl) << 32) |
                            (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)src_s, jl);
        const uint32_t rj = (uint32_t)__builtin_amdgcn_readlane((int)src_r, jl);
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            if (row[r] == PAD_ROW || rj == row[r]) continue;
            if (sj > sc[r]) ++rank[r];
            else if (sj == sc[r]) {
                const int c = key_cmp3(ix.keys, rj, row[r]);
                if (c < 0 || (c == 0 && rj < row[r])) ++rank[r];
            }
        }
    }
    const int cnt = E < k ? E : k;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        if (row[r] != PAD_ROW && rank[r] < cnt) {
            const int t = rank[r];
            res->rows[t] = row[r];
            res->scores[t] = lbh_unord64(sc[r]);
#pragma unroll
            for (int w = 0; w < 4; ++w) res->keys[t][w] = kw[r][w];
            if (xc.dst) {
                int64_t* e = xc.dst + 6 * (t + 1);
                e[0] = (int64_t)__builtin_bit_cast(uint64_t, lbh_unord64(sc[r]));
                e[1] = (int64_t)((ix.ids ? ix.ids[row[r]] : (uint64_t)row[r]) * xc.gid_mul + xc.gid_add);
#pragma unroll
                for (int w = 0; w < 4; ++w) e[2 + w] = (int64_t)kw[r][w];
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float t = __shfl_xor(err, o);
        err = t > err ? t : err;
    }
    if (lane == 0) {
        res->count = cnt; res->overflow = 0; res->candidates = E; res->max_err = err;
        if (xc.dst) xc.dst[0] = cnt;
    }
}

// K4 body: global fp32 threshold over E flat candidates, candidate cut, exact fp64 re-rank.
// `qs` (64 doubles in LDS) must be filled and a barrier passed before the call.
template <int NT = SEL_THREADS>
__device__ void select_fast_body(const lbh_dev_index& ix, const double* qs, int k, const uint64_t* cand,
                                 int E, lbh_result* __restrict__ res, float margin, const lbh_xchg& xc = lbh_xchg{nullptr, 0, 0},
                                 int out_off = 0) {
    __shared__ uint64_t sbuf[2 * LBH_MCAP];      // 32 KB: entries, later (ord64 | rows)
    __shared__ int hist[256];
    __shared__ int64_t sh_krem;
    __shared__ int sh_bin, sh_cnt;
    const int tid = threadIdx.x;
    if (E <= 64 && k <= LBH_MAX_PAGE && out_off == 0) {  // block-uniform (one entry per lane; with two, 65-128 entries took longer than the block-wide path below)
        select_small(ix, qs, k, cand, E, res, xc);
        return;
    }
    if (tid == 0) sh_cnt = 0;
    __syncthreads();
    int M = 0;
    uint32_t thr = 0;
    constexpr int RANK_ALL = NT >= 1024 ? LBH_RANK_ALL : 2 * NT;  // a narrower block counts ranks for two entries per thread
    constexpr int RPT = (RANK_ALL + NT - 1) / NT;
    if (E <= RANK_ALL) {
        // A short list (the scan blocks share their cuts, so this is the usual case): no sort at all.  Thread t counts the
        // entries above entry t (entries are unique: the row sits in the low word; every thread reads the same LDS word at
        // the same time, a broadcast) -- the one with k - 1 above it is the k-th best.  Four barriers instead of the
        // bitonic sort's 36-45.
        __shared__ uint32_t sh_thr;
        uint64_t v[RPT];
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            const int e = tid + r * NT;
            v[r] = e < E ? cand[e] : 0ull;
            if (e < E) sbuf[e] = v[r];
        }
        if (tid == 0) sh_thr = 0;
        __syncthreads();
        if (E >= k) {
            int above[RPT];
#pragma unroll
            for (int r = 0; r < RPT; ++r) above[r] = 0;
            for (int i = 0; i < E; ++i) {
                const uint64_t o = sbuf[i];
#pragma unroll
                for (int r = 0; r < RPT; ++r) above[r] += o > v[r];
            }
#pragma unroll
            for (int r = 0; r < RPT; ++r)
                if (tid + r * NT < E && above[r] == k - 1) sh_thr = lbh_ord32(lbh_unord32((uint32_t)(v[r] >> 32)) - margin);
        }
        __syncthreads();
        thr = sh_thr;
        int pos[RPT];
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            const bool keep = tid + r * NT < E && (uint32_t)(v[r] >> 32) >= thr;
            pos[r] = keep ? atomicAdd(&sh_cnt, 1) : -1;
        }
        __syncthreads();  // every entry is in a register before the survivors are packed into sbuf
#pragma unroll
        for (int r = 0; r < RPT; ++r)
            if (pos[r] >= 0) sbuf[pos[r]] = v[r];
        __syncthreads();
        M = sh_cnt;
    } else if (NT >= SEL_THREADS && E <= LBH_SORT_ALL) {  // few entries: sort them all; more: radix-select the k-th score first (a 4096-entry sort took 20+ us)
        for (int e = tid; e < E; e += NT) sbuf[e] = cand[e];
        const int n2 = pow2_at_least(E, 2);
        for (int i = E + tid; i < n2; i += NT) sbuf[i] = 0;
        __syncthreads();
        sort_desc_u64(sbuf, n2, tid, NT, BlockSync());
        if (E >= k) thr = lbh_ord32(lbh_unord32((uint32_t)(sbuf[k - 1] >> 32)) - margin);
        for (int i = tid; i < E; i += NT) {
            const bool in = (uint32_t)(sbuf[i] >> 32) >= thr;
            const bool next_in = (i + 1 < E) && (uint32_t)(sbuf[i + 1] >> 32) >= thr;
            if (in && !next_in) sh_cnt = i + 1;
        }
        __syncthreads();
        M = sh_cnt;
    } else if (E <= 2 * LBH_MCAP) {
        // (Tried instead for k <= 16: every wave sorts a 1/16 slice in registers, wave 0 merges the slices' k best.  Two
        // barriers instead of the radix passes, but 16 waves of cross-lane sorts at once took 13.3 us against this
        // path's 6.9 for 1 200-2 000 entries -- dropped.)
        // the usual single-query case (512 blocks x k candidates): ONE pass over the global list into LDS; the
        // min/max pass, the digit passes and the cut then read LDS (each global pass cost 1-2 us of L2 latency)
        for (int e = tid; e < E; e += NT) sbuf[e] = cand[e];
        __syncthreads();
        if (E >= k) {
            auto val = [&](int e, uint64_t* v) {
                *v = sbuf[e] >> 32;
                return true;
            };
            const uint32_t kth = (uint32_t)block_radix_select<NT>(E, 4, k, val, hist, &sh_krem, &sh_bin);
            thr = lbh_ord32(lbh_unord32(kth) - margin);
        }
        uint64_t mine[2 * LBH_MCAP / NT];  // read everything before the survivors are packed into sbuf
#pragma unroll
        for (int i = 0; i < 2 * LBH_MCAP / NT; ++i) {
            const int e = tid + i * NT;
            mine[i] = e < E ? sbuf[e] : 0ull;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2 * LBH_MCAP / NT; ++i) {
            if (tid + i * NT < E && (uint32_t)(mine[i] >> 32) >= thr) sbuf[atomicAdd(&sh_cnt, 1)] = mine[i];
        }
        __syncthreads();
        M = sh_cnt;
    } else if (NT == SEL_THREADS && E <= LBH_SEL_REGS * SEL_THREADS) {
        // A long list (top_k 64: ~16 000-23 000 entries from 256-512 blocks; beyond 24 576 the path below): every thread takes the scores of its
        // entries into REGISTERS with one batch of independent loads; the min/max pass, the digit passes and the cut then
        // run without touching global memory (five passes over the global list cost 25 of this kernel's 41 us).
        uint32_t sv[LBH_SEL_REGS];
#pragma unroll
        for (int i = 0; i < LBH_SEL_REGS; ++i) {
            const int e = tid + i * SEL_THREADS;
            sv[i] = e < E ? (uint32_t)(cand[e] >> 32) : 0u;
        }
        if (E >= k) {
            auto each = [&](auto body) {
#pragma unroll
                for (int i = 0; i < LBH_SEL_REGS; ++i)
                    if (tid + i * SEL_THREADS < E) body((uint64_t)sv[i]);
            };
            const uint32_t kth = (uint32_t)block_radix_select_each(4, k, each, hist, &sh_krem, &sh_bin);
            thr = lbh_ord32(lbh_unord32(kth) - margin);
        }
#pragma unroll
        for (int i = 0; i < LBH_SEL_REGS; ++i) {
            const int e = tid + i * SEL_THREADS;
            if (e < E && sv[i] >= thr) {  // the row comes from the list again, for the few survivors only
                const int pos = atomicAdd(&sh_cnt, 1);
                if (pos < 2 * LBH_MCAP) sbuf[pos] = cand[e];
            }
        }
        __syncthreads();
        M = sh_cnt;
    } else {
        if (E >= k) {
            auto val = [&](int e, uint64_t* v) {
                *v = cand[e] >> 32;
                return true;
            };
            const uint32_t kth = (uint32_t)block_radix_select<NT>(E, 4, k, val, hist, &sh_krem, &sh_bin);
            thr = lbh_ord32(lbh_unord32(kth) - margin);
        }
        for (int e = tid; e < E; e += NT) {
            const uint64_t v = cand[e];
            if ((uint32_t)(v >> 32) >= thr) {
                const int pos = atomicAdd(&sh_cnt, 1);
                if (pos < 2 * LBH_MCAP) sbuf[pos] = v;
            }
        }
        __syncthreads();
        M = sh_cnt;
    }
    K4_STAMP(2);  // candidates loaded and cut
#ifdef LBH_K2_TIMING
    if (tid == 0) lbh_k4_dbg[9] = M;
#endif
    if (M > LBH_MCAP) {
        if (tid == 0) { res->count = 0; res->overflow = 1; res->candidates = M; }
        return;
    }
    // rows out of sbuf into registers, then re-use sbuf as (ord64[MCAP], rows[MCAP])
    constexpr int MPT = (LBH_MCAP + NT - 1) / NT;
    uint32_t myrows[MPT], myord[MPT];
#pragma unroll
    for (int i = 0; i < MPT; ++i) {
        const int j = tid + i * NT;
        myrows[i] = j < M ? (uint32_t)sbuf[j] : PAD_ROW;
        myord[i] = j < M ? (uint32_t)(sbuf[j] >> 32) : 0u;
    }
    __syncthreads();
    uint64_t* xo = sbuf;
    uint32_t* xr = reinterpret_cast<uint32_t*>(sbuf + LBH_MCAP);
#pragma unroll
    for (int i = 0; i < MPT; ++i)
        if (tid + i * NT < M) xr[tid + i * NT] = myrows[i];
    __syncthreads();
    rerank_and_emit<NT>(xo, xr, M, k, ix, qs, res, false, xc, myord, out_off);
}

__device__ void fused_select(const lbh_dev_index& ix, const double* qs, int k, const uint64_t* cand, int E, lbh_result* res,
                             float margin, const lbh_xchg& xc) {
    select_fast_body<LBH_SCAN_THREADS>(ix, qs, k, cand, E, res, margin, xc);  // `cand`: the staged list in LDS
}

// Completion ticket: the result block lives in device-mapped pinned host memory, so once every writer
// has fenced to system scope thread 0 publishes the caller's sequence number and the host may read the
// block as soon as it sees that number (no stream synchronisation on the latency path).
__device__ __forceinline__ void publish_result(lbh_result* res, int seq) {
    if (threadIdx.x < 64) __threadfence_system();  // the result block is written by threads 0..63 only (<= LBH_MAX_PAGE rows + the header)
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(&res->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// K4 (fast path, one query): one block.
__global__ __launch_bounds__(SEL_THREADS) void select_fast_kernel(lbh_dev_index ix, const double* __restrict__ qptr,
                                                                   lbh_qarg qa, int k,
                                                                   const uint64_t* __restrict__ cand,
                                                                   int32_t* __restrict__ flags,
                                                                   lbh_result* __restrict__ res, int seq, float margin, lbh_xchg xc,
                                                                   const unsigned long long* __restrict__ bbest, int n_blocks) {
    __shared__ double qs[64];
    const int tid = threadIdx.x;
    K4_STAMP(0);
    const int overflowed = flags[0];
    int E = flags[1];
    if (tid < 64) qs[tid] = qptr ? qptr[tid] : qa.q[tid];
    __syncthreads();
    K4_STAMP(1);  // counters read
#ifdef LBH_K2_TIMING
    if (tid == 0) lbh_k4_dbg[8] = E;
#endif
    if (tid < 4) flags[tid] = 0;  // leave the counters zeroed for the next launch (no memset node needed)
    if (tid >= 64 && tid < 128) flags[tid] = 0;  // and the scan blocks' shared cuts
    if (overflowed) {
        if (tid == 0) { res->count = 0; res->overflow = 1; res->candidates = 0; }
    } else {
        // A sharper cut before anything is sorted or selected (round 3): every scan block left its BEST score (launch-tagged,
        // bbest[block]); the k-th largest of those belongs to k different rows, so the corpus's k-th best is at least that,
        // a bound only a few ranks below it -- where the blocks' own cuts let 1 500-2 000 entries into the flat list, ~10
        // pass this one, and those go to the short-list select (one wave, no sort) instead of the radix select.
        __shared__ uint32_t bm[LBH_BBEST_SLOTS];
        __shared__ uint64_t pre_stage[2048];
        __shared__ uint32_t sh_pre;
        __shared__ int sh_pn;
        const uint64_t* list = cand;
        if (bbest && E > 512 && n_blocks >= k) {  // (a list that short is ranked whole faster than the bound is found)
            const int GB = n_blocks < LBH_BBEST_SLOTS ? n_blocks : LBH_BBEST_SLOTS;
            for (int b = tid; b < LBH_BBEST_SLOTS; b += SEL_THREADS) {
                const unsigned long long w = b < GB ? bbest[b] : 0ull;
                bm[b] = (uint32_t)(w >> 32) == (uint32_t)seq ? (uint32_t)w : 0u;
            }
            if (tid == 0) { sh_pre = 0; sh_pn = 0; }
            __syncthreads();
            if (tid < 64) {
                uint32_t v[LBH_BBEST_SLOTS / 64];
#pragma unroll
                for (int i = 0; i < LBH_BBEST_SLOTS / 64; ++i) v[i] = bm[i * 64 + tid];
                const uint32_t kth = wave_kth_bound<LBH_BBEST_SLOTS / 64, 12>(v, k);
                if (tid == 0 && kth) sh_pre = lbh_ord32(lbh_unord32(kth) - margin);
            }
            __syncthreads();
            const uint32_t precut = sh_pre;
            if (precut) {
                for (int e = tid; e < E; e += SEL_THREADS) {
                    const uint64_t v = cand[e];
                    if ((uint32_t)(v >> 32) >= precut) {
                        const int pos = atomicAdd(&sh_pn, 1);
                        if (pos < 2048) pre_stage[pos] = v;
                    }
                }
                __syncthreads();
                if (sh_pn <= 2048) {  // (else: mass near-ties -- the full list goes through the usual path)
                    list = pre_stage;
                    E = sh_pn;
                }
            }
        }
        select_fast_body(ix, qs, k, list, E, res, margin, xc);
    }
    K4_STAMP(6);  // result rows written
    publish_result(res, seq);
    K4_STAMP(7);  // fenced to the host, ticket stored
}

// ------------------------------------------------------------------------------------------------
// K3: batched queries on the matrix cores.  C[N x Q] = E[N x 64] * Q^T is never materialised: a
// workgroup of 8 waves walks 64-claim super-tiles of E (converted and staged once in LDS, double-
// buffered); wave w owns queries 32w..32w+31 and keeps their 64 dims as MFMA B fragments in registers
// for the whole launch.  The D fragment puts one QUERY on each lane (col = lane & 31) and 16 claims in
// its registers, so the running threshold of a query is a per-lane register and the since/tag mask is
// one 32-bit word per tile.  Candidates go to per-query LDS buffers (BQ_CAP entries); a buffer above
// BQ_TRIG entries is cut at (k-th best - margin) by rank counting (one query) or by per-lane top-k lists
// (all queries of the wave at once).  Two scan precisions share this machinery: scan_batched_kernel
// <.., BF16 = false> (exact fp32 products) and <.., BF16 = true> (split bf16, three products); the default
// precision (one fp16 product) has its own, barrier-free kernel further down (scan_batched_tiled_kernel).  The
// margin is twice the error bound of each.
// ------------------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));
#ifndef LBH_BQ_ABLATE
#define LBH_BQ_ABLATE 0  // 1 / 2: timing-only builds for tools (no MFMA / no epilogue)
#endif
constexpr int BQ_THREADS = 512, BQ_MAXQ = 256, BQ_CAP = 56, BQ_TRIG = 24;
// LDS image of a 64-claim super-tile (double-buffered): row stride 68 floats, each row stored as [even dims | odd dims]
// so that a lane's 32 A values are contiguous (8 x ds_read_b128, conflict-free at this stride)
constexpr int BQ_TS = 68, BQ_TILE_F = 64 * BQ_TS, BQ_NBUF = 2;
constexpr int BQ_OVER = 0x40000000;  // bit of a query's counter: its candidate set could not be bounded here
constexpr int BQ_PERQ = 8192;  // candidate slots per query in HBM (256 workgroups x 32)
// Stride (in ints) between the per-query list counters: one 128-byte line each.  Packed into 1 KiB, the ~72 k appends of
// a 256-query pass were atomics on EIGHT cache lines and serialised there (about 11 ns each: 75 us per pass, whatever
// the kernel around them did); a line per query leaves ~280 per line.
constexpr int BQ_QT = LBH_BQ_QT_STRIDE;
constexpr size_t BQ_LDS = (size_t)BQ_MAXQ * BQ_CAP * 8 + BQ_NBUF * BQ_TILE_F * 4 + BQ_MAXQ * 4 * 2;  // 151,552 B

struct bq_ctx {          // per-wave view of the K3 candidate state (LDS) for the out-of-line slow path
    uint64_t* qbuf;
    int* qcnt;
    uint32_t* qthr;
    int32_t* qtotal;
    int k, lane, wave, myq, h;
    bool qactive;
    float margin;  // 2 x (error bound of the scan's scores): LBH_MARGIN for fp32 MFMA, LBH_MARGIN_BF16X3 for split bf16
};

// one query's buffer (<= 56 entries, one per lane): find the k-th largest by rank counting (entries are
// unique: the row sits in the low word), raise the threshold, keep what is still within the margin.
// No sort: the survivors are re-packed in lane order, which is all the later stages need.
__device__ __forceinline__ void bq_compact_impl(const bq_ctx& c, int q) {
    const int lane = c.lane;
    const int cnt = c.qcnt[q] < BQ_CAP ? c.qcnt[q] : BQ_CAP;  // the counter may have run past a full buffer
    const uint64_t* buf = c.qbuf + (size_t)q * BQ_CAP;
    const uint64_t v = lane < cnt ? buf[lane] : 0ull;
    uint32_t t2 = c.qthr[q];
    int newcnt = cnt;
    if (cnt >= c.k) {
        int rank = 0;  // number of entries strictly greater than mine
        for (int i = 0; i < cnt; ++i) rank += buf[i] > v;  // wave-uniform address: an LDS broadcast read
        const uint64_t sel = __ballot(lane < cnt && rank == c.k - 1);
        const int src = __ffsll((long long)sel) - 1;
        const uint32_t kth = (uint32_t)(__shfl(v, src) >> 32);
        const uint32_t nt = lbh_ord32(lbh_unord32(kth) - c.margin);
        if (nt > t2) t2 = nt;
        const bool keep = lane < cnt && (uint32_t)(v >> 32) >= t2;
        const uint64_t km = __ballot(keep);
        newcnt = __popcll(km);
        const uint64_t lt = lane ? (~0ull >> (64 - lane)) : 0ull;
        __builtin_amdgcn_wave_barrier();  // every lane holds its entry in a register before the re-pack
        if (keep) c.qbuf[(size_t)q * BQ_CAP + __popcll(km & lt)] = v;
    }
    if (lane == 0) {
        c.qcnt[q] = newcnt;
        c.qthr[q] = t2;
        if (newcnt > BQ_TRIG) atomicOr(&c.qtotal[q * BQ_QT], BQ_OVER);  // too many near-ties for this path: that query only
    }
    __builtin_amdgcn_wave_barrier();
}

__device__ __noinline__ void bq_compact(const bq_ctx& c, int q) { bq_compact_impl(c, q); }

// All of a wave's queries at once: lane L serves query wave*32 + (L & 31); the two lanes of a query
// (L, L + 32) each scan every other entry of its buffer, keeping a sorted top-KB list of scores in
// registers, then merge their lists -> k-th best -> new threshold; lane L < 32 re-packs the buffer in
// place.  One call costs about what bq_compact_impl costs for TWO queries, whatever the number of
// queries in `need` -- with cold thresholds (sample pass, first tiles) every query needs it at once.
template <int KB>
__device__ __forceinline__ void bq_compact_all_impl(const bq_ctx& c, uint64_t need) {
    const int lane = c.lane, qi = lane & 31, half = lane >> 5;
    const int q = c.wave * 32 + qi;
    const bool mine = (need >> qi) & 1ull;
    uint64_t* buf = c.qbuf + (size_t)q * BQ_CAP;
    int cnt = 0;
    if (mine) cnt = c.qcnt[q] < BQ_CAP ? c.qcnt[q] : BQ_CAP;
    uint32_t top[KB];
#pragma unroll
    for (int m = 0; m < KB; ++m) top[m] = 0u;
    for (int i = half; i < cnt; i += 2) {
        uint32_t v = (uint32_t)(buf[i] >> 32);
#pragma unroll
        for (int m = 0; m < KB; ++m) {
            const uint32_t hi = top[m] > v ? top[m] : v;
            v = top[m] > v ? v : top[m];
            top[m] = hi;
        }
    }
    // merge with the partner lane's list (same query, other half of the entries)
    uint32_t other[KB];
#pragma unroll
    for (int m = 0; m < KB; ++m) other[m] = __shfl_xor(top[m], 32);
#pragma unroll
    for (int e = 0; e < KB; ++e) {
        uint32_t v = other[e];
#pragma unroll
        for (int m = 0; m < KB; ++m) {
            const uint32_t hi = top[m] > v ? top[m] : v;
            v = top[m] > v ? v : top[m];
            top[m] = hi;
        }
    }
    if (mine && half == 0) {
        uint32_t t2 = c.qthr[q];
        int newcnt = cnt;
        if (cnt >= c.k) {
            uint32_t kth = 0;
#pragma unroll
            for (int m = 0; m < KB; ++m) kth = (m == c.k - 1) ? top[m] : kth;
            const uint32_t nt = lbh_ord32(lbh_unord32(kth) - c.margin);
            if (nt > t2) t2 = nt;
            newcnt = 0;
            for (int i = 0; i < cnt; i += 4) {  // in-place: the write position never passes the read position
                uint64_t v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = i + u < cnt ? buf[i + u] : 0ull;
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (i + u < cnt && (uint32_t)(v[u] >> 32) >= t2) buf[newcnt++] = v[u];
            }
        }
        c.qcnt[q] = newcnt;
        c.qthr[q] = t2;
        if (newcnt > BQ_TRIG) atomicOr(&c.qtotal[q * BQ_QT], BQ_OVER);  // too many near-ties for this path: that query only
    }
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ void bq_compact_all(const bq_ctx& c, uint64_t need) {
    if (c.k <= 8) bq_compact_all_impl<8>(c, need);
    else bq_compact_all_impl<LBH_BQ_KMAX>(c, need);
}

// slow path of the K3 epilogue: exact ord compare, mask, append, compaction.  Returns the lane's
// (possibly raised) threshold.  Inlined into the scan loop: an out-of-line call spilled the caller's live
// registers around every candidate tile, which dominated the cold sample pass.
template <bool INL>
__device__ __forceinline__ uint32_t bq_harvest_impl(const bq_ctx& c, const f32x16& acc, int64_t row0, uint32_t rowmask,
                                                    uint32_t thr) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int rowbit = (r & 3) + 8 * (r >> 2) + 4 * c.h;
        const uint32_t o = lbh_ord32(acc[r]);
        const bool pass = c.qactive && ((rowmask >> rowbit) & 1u) && o >= thr;
        if (__any(pass)) {  // wave-uniform skip: in steady state almost every register has no candidate
            if (pass) {
                const int pos = atomicAdd(&c.qcnt[c.myq], 1);
                if (pos < BQ_CAP) c.qbuf[(size_t)c.myq * BQ_CAP + pos] = ((uint64_t)o << 32) | (uint32_t)(row0 + rowbit);
                else atomicOr(&c.qtotal[c.myq * BQ_QT], BQ_OVER);
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    const uint64_t need = __ballot(c.lane < 32 && c.qactive && c.qcnt[c.myq] > BQ_TRIG);
    if (need) {
        if (__popcll(need) > 2) {
            bq_compact_all(c, need);
        } else {
            uint64_t rest = need;
            while (rest) {
                const int qi = __ffsll((long long)rest) - 1;
                rest &= rest - 1;
                if (INL) bq_compact_impl(c, c.wave * 32 + qi);
                else bq_compact(c, c.wave * 32 + qi);
            }
        }
        if (c.qactive) thr = c.qthr[c.myq];
    }
    return thr;
}
// The same slow path for a caller that already knows WHICH lanes hold a candidate (the fp16 kernel's max test):
// only those lanes walk their 16 scores, everyone else skips the loop, and the buffer fill is taken from the
// positions the appends returned instead of a second LDS read.  A candidate tile costs the wave about a third of
// bq_harvest_impl's time -- and the other seven waves of the workgroup wait for it at the tile barrier.
__device__ __forceinline__ uint32_t bq_harvest_lanes(const bq_ctx& c, const f32x16& acc, int64_t row0, uint32_t rowmask,
                                                     uint32_t thr, bool lanehit) {
    int last = -1;  // highest buffer position this lane took
    if (lanehit && c.qactive) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rowbit = (r & 3) + 8 * (r >> 2) + 4 * c.h;
            const uint32_t o = lbh_ord32(acc[r]);
            if (((rowmask >> rowbit) & 1u) && o >= thr) {
                const int pos = atomicAdd(&c.qcnt[c.myq], 1);
                if (pos < BQ_CAP) c.qbuf[(size_t)c.myq * BQ_CAP + pos] = ((uint64_t)o << 32) | (uint32_t)(row0 + rowbit);
                else atomicOr(&c.qtotal[c.myq * BQ_QT], BQ_OVER);
                last = pos;
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    uint64_t need = __ballot(last >= BQ_TRIG);  // the query's count passed the trigger with this tile
    need = (need | (need >> 32)) & 0xFFFFFFFFull;  // lanes L and L + 32 serve the same query
    if (need) {
        if (__popcll(need) > 2) {
            bq_compact_all(c, need);
        } else {
            uint64_t rest = need;
            while (rest) {
                const int qi = __ffsll((long long)rest) - 1;
                rest &= rest - 1;
                bq_compact_impl(c, c.wave * 32 + qi);
            }
        }
        if (c.qactive) thr = c.qthr[c.myq];
    }
    return thr;
}
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <bool SINCE, bool TAGS, bool BF16>
__global__ __launch_bounds__(BQ_THREADS) void scan_batched_kernel(lbh_dev_index ix, const double* __restrict__ qmat,
                                                                   int nq, int k, float margin, int64_t since_ms, uint64_t tag_mask,
                                                                   int64_t row_begin, int64_t row_end,
                                                                   const uint32_t* __restrict__ seed,
                                                                   uint64_t* __restrict__ cand3,
                                                                   int32_t* __restrict__ qtotal,
                                                                   int32_t* __restrict__ flags) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint64_t* qbuf = reinterpret_cast<uint64_t*>(smem);                                  // [256][64]
    float* tile = reinterpret_cast<float*>(smem + (size_t)BQ_MAXQ * BQ_CAP * 8);           // [2][64][68]
    int* qcnt = reinterpret_cast<int*>(smem + (size_t)BQ_MAXQ * BQ_CAP * 8 + BQ_NBUF * BQ_TILE_F * 4);
    uint32_t* qthr = reinterpret_cast<uint32_t*>(qcnt + BQ_MAXQ);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int myq = wave * 32 + j;
    const bool qactive = myq < nq;
    const bool wave_active = wave * 32 < nq;
    const int64_t n = row_end < ix.n ? row_end : ix.n;  // this launch scans rows [row_begin, n)
    const f32x4* __restrict__ E = reinterpret_cast<const f32x4*>(ix.emb32);

    // seed: per-query lower bounds on the final threshold (k-th best of a sample pass - margin)
    if (tid < BQ_MAXQ) { qcnt[tid] = 0; qthr[tid] = (seed && tid < nq) ? seed[tid] : 0u; }
    // B operand = this lane's query.  fp32 MFMA (32x32x2): dims 2s + h.  Split-bf16 MFMA (32x32x16): k-step s
    // takes dims 16s + 8h + 0..7, each value as hi = bf16(v) and lo = bf16(v - hi).
    float b[32];
    bf16x8 bh[4], bl[4];
    if (!BF16) {
#pragma unroll
        for (int s = 0; s < 32; ++s) {
            // unconditional load from a clamped row, then select: a load under a lane-dependent branch would
            // be waited for on the spot (32 serial round trips)
            const float v = (float)qmat[(size_t)(qactive ? myq : 0) * 64 + 2 * s + h];
            b[s] = qactive ? v : 0.f;
        }
    } else {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float v0 = (float)qmat[(size_t)(qactive ? myq : 0) * 64 + 16 * s + 8 * h + e];
                const float v = qactive ? v0 : 0.f;
                const __bf16 hi = (__bf16)v;
                bh[s][e] = hi;
                bl[s][e] = (__bf16)(v - (float)hi);
            }
        }
    }
    uint32_t thr = (seed && qactive) ? seed[myq] : 0u;

    // staging role: a super-tile is 64 claims (two MFMA tiles); thread -> (row, 16-byte chunk) x 2
    const int srow = tid >> 4, sc4 = tid & 15;
    auto load_half = [&](int64_t st, int half) -> f32x4 {
        // always load (row clamped into the matrix): a guarded load is waited for inside its branch, which
        // would expose the HBM latency every super-tile; rows >= n are dropped by row_mask() instead
        int64_t r = row_begin + st * 64 + half * 32 + srow;
        r = r < n ? r : n - 1;
#if LBH_BQ_ABLATE == 4  // timing-only build: loads stay inside a 1 MiB, L2-resident window
        r &= 4095;
#endif
        return __builtin_nontemporal_load(&E[r * 16 + sc4]);
    };
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    auto store_half = [&](int buf, int half, f32x4 v) {
        if (!BF16) {
            float* d = tile + (size_t)buf * BQ_TILE_F + (half * 32 + srow) * BQ_TS + 2 * sc4;
            *reinterpret_cast<f32x2*>(d) = f32x2{v.x, v.z};        // even dims 4c, 4c+2  -> slots 2c, 2c+1
            *reinterpret_cast<f32x2*>(d + 32) = f32x2{v.y, v.w};   // odd dims            -> 32 + 2c, 32 + 2c+1
        } else {
            // same 272-byte rows: [64 x bf16 hi | 64 x bf16 lo | pad]; the split is done once here, by the
            // staging thread, not by the eight waves that consume the tile
            char* d = reinterpret_cast<char*>(tile + (size_t)buf * BQ_TILE_F + (half * 32 + srow) * BQ_TS) + 8 * sc4;
            const bf16x4 hi = __builtin_convertvector(v, bf16x4);
            const bf16x4 lo = __builtin_convertvector(v - __builtin_convertvector(hi, f32x4), bf16x4);
            *reinterpret_cast<bf16x4*>(d) = hi;
            *reinterpret_cast<bf16x4*>(d + 128) = lo;
        }
    };
    bq_ctx bc;
    bc.qbuf = qbuf; bc.qcnt = qcnt; bc.qthr = qthr; bc.qtotal = qtotal;
    bc.k = k; bc.lane = lane; bc.wave = wave; bc.myq = myq; bc.h = h; bc.qactive = qactive;
    bc.margin = margin;

    // rows of one 32-claim tile that pass the since / tag filters, as one word (lanes 0..31 vote)
    auto row_mask = [&](int64_t row0) -> uint32_t {
        bool rv = false;
        if (lane < 32) {
            const int64_t r = row0 + lane;
            rv = r < n;
            if (SINCE) rv = rv && ix.created[rv ? r : 0] >= since_ms;
            if (TAGS) rv = rv && (ix.tags[r < n ? r : 0] & tag_mask) != 0;
        }
        return (uint32_t)__ballot(rv);
    };
    // float image of thr: a cheap superset test in the common case (idle lanes never hit)
    float thr_f = qactive ? (thr ? lbh_unord32(thr) : -__builtin_inff()) : __builtin_inff();
    auto harvest = [&](const f32x16& acc, int64_t row0, uint32_t rowmask) {
        thr = bq_harvest_impl<true>(bc, acc, row0, rowmask, thr);  // inlined: a call per candidate tile spilled more than it saved
        if (qactive) thr_f = thr ? lbh_unord32(thr) : -__builtin_inff();
    };
    auto any_hit = [&](const f32x16& acc) -> bool {
        bool hit = false;
#pragma unroll
        for (int r = 0; r < 16; ++r) hit |= acc[r] >= thr_f;
        return __any(hit);
    };

    // A fragments of the current super-tile live in registers; the LDS image is double-buffered so one
    // barrier per super-tile is enough (the image being filled was last read two barriers ago).
    const int64_t n_super = n > row_begin ? (n - row_begin + 63) >> 6 : 0;
    const int64_t G = gridDim.x;
    int64_t st = blockIdx.x;
    float a0[32], a1[32];
    bf16x8 a0h[4], a0l[4], a1h[4], a1l[4];
    auto read_frags = [&](int buf) {
        if (!BF16) {
            const f32x4* p0 = reinterpret_cast<const f32x4*>(tile + (size_t)buf * BQ_TILE_F + j * BQ_TS + h * 32);
            const f32x4* p1 = reinterpret_cast<const f32x4*>(tile + (size_t)buf * BQ_TILE_F + (32 + j) * BQ_TS + h * 32);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const f32x4 u = p0[i], w = p1[i];
                a0[4 * i] = u.x; a0[4 * i + 1] = u.y; a0[4 * i + 2] = u.z; a0[4 * i + 3] = u.w;
                a1[4 * i] = w.x; a1[4 * i + 1] = w.y; a1[4 * i + 2] = w.z; a1[4 * i + 3] = w.w;
            }
        } else {
            // lane (j, h), k-step s: 8 consecutive bf16 = dims 16s + 8h .. +7 of row j -> one ds_read_b128 at
            // byte 32s + 16h of the row (hi) and 128 + 32s + 16h (lo); conflict-free at the 272-byte stride
            const char* p0 = reinterpret_cast<const char*>(tile + (size_t)buf * BQ_TILE_F + j * BQ_TS) + 16 * h;
            const char* p1 = reinterpret_cast<const char*>(tile + (size_t)buf * BQ_TILE_F + (32 + j) * BQ_TS) + 16 * h;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                a0h[s] = *reinterpret_cast<const bf16x8*>(p0 + 32 * s);
                a0l[s] = *reinterpret_cast<const bf16x8*>(p0 + 128 + 32 * s);
                a1h[s] = *reinterpret_cast<const bf16x8*>(p1 + 32 * s);
                a1l[s] = *reinterpret_cast<const bf16x8*>(p1 + 128 + 32 * s);
            }
        }
    };
    store_half(0, 0, load_half(st, 0));
    store_half(0, 1, load_half(st, 1));
    // global loads run TWO super-tiles ahead (one iteration is shorter than the loaded HBM latency once the
    // MFMAs are the short bf16 ones): nx = tile it+1 (stored this iteration), ny = tile it+2
    f32x4 nx0 = load_half(st + G, 0), nx1 = load_half(st + G, 1);
    f32x4 ny0 = load_half(st + 2 * G, 0), ny1 = load_half(st + 2 * G, 1);
    __syncthreads();
    if (wave_active) read_frags(0);
    for (int it = 0; st < n_super; st += G, ++it) {
#if LBH_BQ_ABLATE == 6  // timing-only build: loaded data is consumed but not staged
        asm volatile("" ::"v"(nx0), "v"(nx1));
#else
        store_half((it + 1) & 1, 0, nx0);   // super-tile it+1 (prefetched during the previous iteration)
        store_half((it + 1) & 1, 1, nx1);
#endif
        nx0 = ny0;
        nx1 = ny1;
        ny0 = load_half(st + 3 * G, 0);
        ny1 = load_half(st + 3 * G, 1);
        if (wave_active) {
            const int64_t row0 = row_begin + st * 64;
            const uint32_t m0 = row_mask(row0), m1 = row_mask(row0 + 32);
            f32x16 acc0 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            f32x16 acc1 = acc0;
#if LBH_BQ_ABLATE == 1  // timing-only build: no matrix work
#pragma unroll
            for (int s = 0; s < 32; ++s) { acc0[s & 15] += a0[s] * b[s]; acc1[s & 15] += a1[s] * b[s]; }
            if (BF16) asm volatile("" ::"v"(a0h[0]), "v"(a0l[0]), "v"(a1h[0]), "v"(a1l[0]), "v"(bh[0]), "v"(bl[0]));
#else
            if (!BF16) {
#pragma unroll
                for (int s = 0; s < 32; ++s) acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[s], b[s], acc0, 0, 0, 0);
#pragma unroll
                for (int s = 0; s < 32; ++s) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s], b[s], acc1, 0, 0, 0);
            } else {
                // x*q ~ hi*hi + hi*lo + lo*hi (the dropped lo*lo and the split residues are < 3 * 2^-18 |x q|):
                // 12 MFMAs of 32 cycles per tile instead of 32 of 64; cross terms first, they are the small ones
#if LBH_BQ_ABLATE == 8  // timing-only build: every fragment is read and waited for, no matrix work
#pragma unroll
                for (int s = 0; s < 4; ++s) asm volatile("" ::"v"(a0l[s]), "v"(a0h[s]), "v"(a1l[s]), "v"(a1h[s]), "v"(bh[s]), "v"(bl[s]));
#else
                // the two tiles' chains alternate, so consecutive MFMAs never depend on each other
#if LBH_BQ_ABLATE != 7  // 7: timing-only build with a third of the matrix work (hi*hi only)
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0l[s], bh[s], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1l[s], bh[s], acc1, 0, 0, 0);
                }
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0h[s], bl[s], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1h[s], bl[s], acc1, 0, 0, 0);
                }
#endif
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0h[s], bh[s], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1h[s], bh[s], acc1, 0, 0, 0);
                }
#endif
            }
#endif
#if LBH_BQ_ABLATE >= 2  // timing-only builds: no epilogue
            asm volatile("" ::"v"(acc0), "v"(acc1));
#else
            if (any_hit(acc0)) harvest(acc0, row0, m0);
            if (any_hit(acc1)) harvest(acc1, row0 + 32, m1);
#endif
        }
        __syncthreads();
#if LBH_BQ_ABLATE != 5  // 5: timing-only build without the per-tile fragment reads
        if (wave_active) read_frags((it + 1) & 1);
#endif
    }
    // flush: final cut per query, then append to the query's flat candidate list in HBM
    if (wave_active) {
        bq_compact_all(bc, __ballot(lane < 32 && qactive && qcnt[myq] > 0));
        // one global atomic round trip for the wave's 32 queries (lane = query), then the copies
        int mycnt = 0, mybase = 0;
        if (lane < 32 && qactive) {
            mycnt = qcnt[myq] < BQ_CAP ? qcnt[myq] : BQ_CAP;
            if (mycnt) {
                mybase = atomicAdd(&qtotal[myq * BQ_QT], mycnt) & (BQ_OVER - 1);
                if (mybase + mycnt > BQ_PERQ) {
                    atomicOr(&qtotal[myq * BQ_QT], BQ_OVER);
                    mycnt = 0;
                }
            }
        }
        for (int qi = 0; qi < 32 && wave * 32 + qi < nq; ++qi) {
            const int q = wave * 32 + qi;
            const int cnt = __shfl(mycnt, qi), base = __shfl(mybase, qi);
            if (lane < cnt) cand3[(size_t)q * BQ_PERQ + base + lane] = qbuf[(size_t)q * BQ_CAP + lane];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K3, fp16 flavour (the default): ONE v_mfma_f32_32x32x16_f16 product per k-step (4 MFMAs per 32 x 32 tile), claim
// tiles PRIVATE to a wave.
//
// Why fp16: the split-bf16 kernel above is power-bound, not issue-bound (tools/mfma_rate.hip: 13.4 ns per 32x32x16 MFMA
// on an idle chip, 16.7 ns with every CU issuing, 20 ns with random operand bits); less matrix work is the only lever
// and the exact re-rank makes it safe: the scan only has to bound its error.  fp16 keeps 11 significant bits, so
// |fp16(x) fp16(q) - x q| <= (2^-11 + 2^-11 + 2^-22) |x q|; summed over a pair of unit vectors (sum |x_i q_i| <= 1)
// that is < 9.8e-4, plus 64 fp32 accumulations (< 1e-5) and fp16 denormal flushes of components below 6e-5
// (< 64 * 6e-5 * 6e-5).  eps < 1e-3, margin LBH_MARGIN_F16 = 2.2e-3 > 2 eps.
//
// Why wave-private tiles (round 2): the round-1 kernel staged every 64-claim super-tile through LDS for eight waves
// that each owned 32 queries, with one workgroup barrier per super-tile; a candidate in any of the 256 queries held up
// all eight waves at the next barrier, the 256 per-query LDS candidate buffers left room for one workgroup per CU, and
// the kernel ran at 0.74 us per super-tile against 0.21 us of matrix time (profiles/r01_k3_wave_timeline.txt).  Here
//   * the fp16 copy of the matrix is stored in MFMA-fragment order (lbh_emb16_off): a wave fetches the A fragments of a
//     32-claim tile with four coalesced 1-KiB global_load_dwordx4 straight into registers -- no LDS image, no
//     conversion, no barrier anywhere in the kernel; three tiles are in flight per wave;
//   * the wave multiplies its tile by ALL query groups: the B fragments of up to 256 queries (8 groups x 4 k-steps x
//     16 B per lane = 128 VGPRs) stay in registers for the whole launch, loaded once from a pre-converted image;
//   * a lane's threshold for each of its 8 queries is a register; the common-case epilogue of a 32 x 32 block is the
//     lane maximum (8 v_max3_f32), one compare, one wave vote.  Thresholds are FIXED for a launch (see the passes
//     below), so there is no candidate buffer to compact: the rare hits go to a 256-entry per-wave staging list in LDS
//     (ballot-prefix append) that is flushed to the queries' flat lists in HBM with one atomic per entry -- in the
//     steady state once, at the end of the kernel.  A hit stalls only the wave that found it.
// Passes (launch_batched_tiled): (1) SAMPLE: scan_batched_max_kernel runs the same loop over 1/14 of the tiles, spread
// evenly over the index, and keeps only group maxima; seed_batched_kernel turns the k-th largest of a query's group maxima
// into its threshold (a lower bound of the corpus's k-th best score, minus the margin, never exceeds the final cut).
// (2) the full pass from those thresholds: about 14 k rows per query reach them, whatever the score distribution.
// A corpus of <= 8192 rows skips (1): every row of every query is listed.  (Round 2 started with a DENSE pass that listed
// every score of 8192 rows, a sort, and an intermediate pass over 286 k rows: 103 us of kernels at 10 M claims where
// the sample pass + seed take 38 us, and three launches fewer.)
// ------------------------------------------------------------------------------------------------
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#ifndef K3T_BREG
#define K3T_BREG 1                  // 1: all eight query groups' B fragments in registers (2 waves per SIMD); 0: re-read from LDS (4 waves)
#endif
#ifndef K3T_THREADS
#define K3T_THREADS (K3T_BREG ? 256 : 512)  // workgroup of the full-pass kernel
#endif
#ifndef K3T_BLOCKS_PER_CU
#define K3T_BLOCKS_PER_CU 2
#endif
constexpr int K3T_WAVES = K3T_THREADS / 64;
constexpr int K3T_STAGE = 256;      // per-wave staging list (entries)
#ifndef K3T_RING
#define K3T_RING 2                  // claim tiles in flight per wave (register sets of 16 VGPRs)
#endif
// the sample pass (scan_batched_max_kernel) keeps the LDS variant's shape: ~104 registers, 4 waves per SIMD
#define K3M_THREADS 512
#define K3M_BLOCKS_PER_CU 2
constexpr int K3M_WAVES = K3M_THREADS / 64;
#define K3M_WPS (K3M_BLOCKS_PER_CU * K3M_THREADS / 256)
#define K3T_WPS (K3T_BLOCKS_PER_CU * K3T_THREADS / 256)  // waves per SIMD the register budget is sized for


// B operand image: [group g][k-step s][lane] x 16 B = dims 16 s + 8 h .. + 7 of query 32 g + j as fp16 (one rounding
// from the fp64 query); queries >= nq are zero
//
// I8 (bq_mode 3): the queries as int8 B fragments of v_mfma_i32_32x32x32_i8 -- [group g][k-step s][lane] x 16 B = dims
// 32 s + 16 h .. + 15 of query 32 g + j as Q_i = rint(q_i / t_q), t_q = (largest |q_i|) / 127 rounded up (1 for a zero
// query) -- and qscale[q] = emb_scale * t_q, the score one integer unit of the product stands for.
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
template <bool I8>
__global__ void prep_bfrag_kernel(const double* __restrict__ qmat, int nq, f16x8* __restrict__ bfrag, int32_t* __restrict__ qtotal,
                                  int32_t* __restrict__ flags, double* __restrict__ qcopy, float* __restrict__ qscale, float emb_scale) {
    if (I8) {
        __shared__ uint32_t qmaxb[32];
        const int t = threadIdx.x, jq = t & 31, part = t >> 5, q = 32 * blockIdx.x + jq;
        if (t < 32) qmaxb[t] = 0u;
        __syncthreads();
        float m = 0.f;
        if (q < nq) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const double a = __builtin_fabs(qmat[(size_t)q * 64 + 8 * part + e]);
                float f = (float)a;
                if ((double)f < a) f = lbh_next_up(f);
                m = f > m ? f : m;
            }
        }
        if (m > 0.f) atomicMax(&qmaxb[jq], __builtin_bit_cast(uint32_t, m));
        __syncthreads();
        const float qm = __builtin_bit_cast(float, qmaxb[jq]);
        const float tq = qm > 0.f ? lbh_next_up(qm / 127.0f) : 1.0f;
        if (part == 0 && q < BQ_MAXQ) qscale[q] = emb_scale * tq;
        if (blockIdx.x == 0) {
            qtotal[threadIdx.x * BQ_QT] = 0;
            if (threadIdx.x < 4) flags[threadIdx.x] = 0;
        }
        if (qcopy) {
            const int tt = blockIdx.x * 256 + threadIdx.x;
            if (tt * 8 < nq * 64) {
#pragma unroll
                for (int e = 0; e < 8; ++e) qcopy[tt * 8 + e] = qmat[tt * 8 + e];
            }
        }
        if (t < 128) {  // (s, lane): 16 dims each
            const int lane = t & 63, ks = t >> 6, j = lane & 31, h = lane >> 5, qq = 32 * blockIdx.x + j;
            const float qm2 = __builtin_bit_cast(float, qmaxb[j]);
            const double inv = qm2 > 0.f ? 1.0 / (double)lbh_next_up(qm2 / 127.0f) : 1.0;
            union { int8_t b[16]; i32x4 v; } u;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                int qi = qq < nq ? __double2int_rn(qmat[(size_t)qq * 64 + 32 * ks + 16 * h + e] * inv) : 0;
                qi = qi > 127 ? 127 : (qi < -127 ? -127 : qi);
                u.b[e] = (int8_t)qi;
            }
            reinterpret_cast<i32x4*>(bfrag)[(blockIdx.x * 2 + ks) * 64 + lane] = u.v;
        }
        return;
    }
    if (blockIdx.x == 0) {  // first kernel of a batched search: also clears the per-query counters (no memset launches)
        qtotal[threadIdx.x * BQ_QT] = 0;
        if (threadIdx.x < 4) flags[threadIdx.x] = 0;
    }
    if (qcopy) {  // qmat is mapped host memory (no copy-engine hop): leave a device copy for the select kernel's fp64 re-rank
        const int t = blockIdx.x * 256 + threadIdx.x;  // 2048 threads x 8 doubles = 256 queries x 64
        if (t * 8 < nq * 64) {
#pragma unroll
            for (int e = 0; e < 8; ++e) qcopy[t * 8 + e] = qmat[t * 8 + e];
        }
    }
    const int lane = threadIdx.x & 63, s = (threadIdx.x >> 6) & 3, g = blockIdx.x;
    const int j = lane & 31, h = lane >> 5, q = 32 * g + j;
    f16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = q < nq ? (_Float16)qmat[(size_t)q * 64 + 16 * s + 8 * h + e] : (_Float16)0.f;
    bfrag[(g * 4 + s) * 64 + lane] = v;
}

// staging list -> the queries' flat lists in HBM: one atomic per entry (a handful per wave and launch in the steady state)
__device__ __forceinline__ void k3t_flush(const uint64_t* se, const uint16_t* sq, int cnt, int lane, uint64_t* __restrict__ cand3,
                                          int32_t* __restrict__ qtotal) {
    for (int i = lane; i < cnt; i += 64) {
        const uint64_t e = se[i];
        const int q = sq[i];
        const int pos = atomicAdd(&qtotal[q * BQ_QT], 1) & (BQ_OVER - 1);
        if (pos < BQ_PERQ) cand3[(size_t)q * BQ_PERQ + pos] = e;
        else atomicOr(&qtotal[q * BQ_QT], BQ_OVER);  // more near-candidates than the list holds: that query is re-run alone
    }
    __builtin_amdgcn_wave_barrier();
}

// Slow path of the tiled kernel, deliberately OUT OF LINE: the scan loop holds 24 copies of the 32 x 32 epilogue (3 ring
// slots x 8 query groups); with this code inlined into each of them the loop body was 290 KB of instructions, the fast
// path hopped from one instruction-cache miss to the next and the full pass ran 3x slower than the round-1 kernel.  One
// shared copy stays resident.  Returns the new fill of the wave's staging list.
#ifndef K3T_INLINE_HARVEST
#define K3T_INLINE_HARVEST 0
#endif
#if K3T_INLINE_HARVEST
__device__ __forceinline__
#else
__device__ __noinline__
#endif
int k3t_harvest(f32x16 acc, float thr, float scale, bool lanehit, int q, int64_t row0, int64_t n,
                                        const int64_t* __restrict__ created, const uint64_t* __restrict__ tags,
                                        int has_since, int64_t since_ms, uint64_t tag_mask, uint64_t* se, uint16_t* sq, int cnt,
                                        uint64_t* __restrict__ cand3, int32_t* __restrict__ qtotal) {
    const int lane = threadIdx.x & 63, h = lane >> 5;
    const uint64_t lt_mask = lane ? (~0ull >> (64 - lane)) : 0ull;
    uint32_t rmask = 0xFFFFFFFFu;  // rows of the tile that exist and pass the since / tag filters
    if (has_since || tag_mask || row0 + 32 > n) {
        bool rv = false;
        if (lane < 32) {
            const int64_t r = row0 + lane;
            rv = r < n;
            if (has_since) rv = rv && created[rv ? r : 0] >= since_ms;
            if (tag_mask) rv = rv && (tags[r < n ? r : 0] & tag_mask) != 0;
        }
        rmask = (uint32_t)__ballot(rv);
    }
    // which of the lane's 16 rows pass (bit r), then one ballot per register that has any
    uint32_t bits = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int rowbit = (r & 3) + 8 * (r >> 2) + 4 * h;
        bits |= (lanehit && acc[r] >= thr && ((rmask >> rowbit) & 1u)) ? (1u << r) : 0u;
    }
    uint32_t any_bits = bits;  // OR over the wave
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) any_bits |= (uint32_t)__shfl_xor((int)any_bits, off);
    any_bits = __builtin_amdgcn_readfirstlane(any_bits);
    if (any_bits && cnt + __popcll(__ballot(bits != 0)) * 16 > K3T_STAGE) {  // room for the worst case of this call
        k3t_flush(se, sq, cnt, lane, cand3, qtotal);
        cnt = 0;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        if (!((any_bits >> r) & 1u)) continue;  // wave-uniform
        const int rowbit = (r & 3) + 8 * (r >> 2) + 4 * h;
        const bool pass = (bits >> r) & 1u;
        const uint64_t pm = __ballot(pass);
        if (cnt + 64 > K3T_STAGE) {  // a flood (cold thresholds): flush between registers
            k3t_flush(se, sq, cnt, lane, cand3, qtotal);
            cnt = 0;
        }
        if (pass) {
            const int pos = cnt + __popcll(pm & lt_mask);
            se[pos] = ((uint64_t)lbh_ord32(acc[r] * scale) << 32) | (uint32_t)(row0 + rowbit);  // scale: 1 (fp16 scores), or the score of one integer unit (int8)
            sq[pos] = (uint16_t)q;
        }
        cnt += __popcll(pm);
    }
    __builtin_amdgcn_wave_barrier();
    return cnt;
}

// The scan kernel.  Two builds of the B operand (K3T_BREG):
//  1 (default): every wave keeps the B fragments of all eight query groups in registers for the whole launch (128 of its
//    ~250 VGPRs, two waves per SIMD, 256-thread workgroups); no LDS traffic in the loop, the group loop is fully unrolled.
//    5-6 % faster than (0) on the same box (10 M x 256: scan kernels 0.392 vs 0.416 ms) -- the microbenchmark of the block
//    pattern (tools/mfma_block_rate.hip) gives 23.3 ns per MFMA at two waves with B in registers against 27.8 ns at four
//    waves with B from LDS.  A third tile in flight (ring of 3) spills 12 registers and gives the gain back.
//    Touching the tile after next with one dword per lane, to pull it into L2 a tile-time early with a single extra
//    register, made the pass 4 % slower (0.403 -> 0.42 ms of scan kernels): the extra requests cost more than they hide.
//  0: the waves of a 512-thread workgroup share one read-only LDS image of the queries (B fragments of all groups,
//    32 KiB) filled once in the prologue, four waves per SIMD.  The notes below are from this variant's development.
// Both keep the 256 thresholds in LDS; the prologue's barrier is the only one of the kernel.  Keeping
// the 128 VGPRs of B fragments in registers instead left room for two waves per SIMD at best, with spills next to the
// tile ring; a wave alone on its SIMD cannot issue the epilogue's vector instructions fast enough (4 cycles each instead
// of 2) and, with the accumulators in AGPRs, needs 16 extra moves per block (measured: 456-490 us for the full pass
// against 379 us at two waves per SIMD).  From LDS a group's B fragments cost four conflict-free ds_read_b128 per 32 x 32
// block (a third of the LDS bandwidth when the matrix pipes are saturated), are prefetched one group ahead, and a wave
// needs ~110 registers, so K3T_WPS waves per SIMD overlap one wave's MFMAs with the others' epilogues and loads.
#if K3T_BREG
#define K3T_LDSB bfrag
#else
#define K3T_LDSB ldsb
#endif
// I8 (bq_mode 3): the same kernel over the int8 copy -- a tile is 2 KiB = two k-steps of v_mfma_i32_32x32x32_i8 (half the
// bytes, half the matrix cycles, 64 VGPRs of B fragments instead of 128, so three tiles are in flight); products and sums
// are exact integers, compared in integer units against thresholds converted once per launch (one unit low, to be safe),
// and a hit is listed with its score in real units (units x qscale[q]).
#ifndef K3I_RING
#define K3I_RING 3
#endif
template <bool I8>
__global__ __launch_bounds__(K3T_THREADS, K3T_WPS) void scan_batched_tiled_kernel(
    lbh_dev_index ix, const f32x4* __restrict__ emb16t, const f32x4* __restrict__ bfrag, int nq, int has_since, int64_t since_ms,
    uint64_t tag_mask, int64_t n_tiles, int64_t n_sel, uint64_t step_q32, const uint32_t* __restrict__ seed,
    uint64_t* __restrict__ cand3, int32_t* __restrict__ qtotal, const float* __restrict__ qscale) {
    constexpr int KS = I8 ? 2 : 4;            // k-steps (16-byte loads per lane) of a tile
    constexpr int TILE_V = KS * 64;           // 16-byte chunks of a tile
    constexpr int RING = I8 ? K3I_RING : K3T_RING;
#if !K3T_BREG
    __shared__ f32x4 ldsb[8 * 4 * 64];  // [group][k-step][lane] x 16 B
#endif
    __shared__ float ldsthr[BQ_MAXQ];
    __shared__ float ldsscale[I8 ? BQ_MAXQ : 1];
    __shared__ uint64_t stage_e[K3T_WAVES][K3T_STAGE];
    __shared__ uint16_t stage_q[K3T_WAVES][K3T_STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31;
    const int ngroups = (nq + 31) >> 5;
    const int64_t n = ix.n;
    uint64_t* se = stage_e[wave];
    uint16_t* sq = stage_q[wave];
    int cnt = 0;

#if !K3T_BREG
    for (int i = tid; i < ngroups * 256; i += K3T_THREADS) ldsb[i] = bfrag[i];
#endif
    for (int q = tid; q < BQ_MAXQ; q += K3T_THREADS) {
        const uint32_t o = (seed && q < nq) ? seed[q] : 0u;
        float t = q < nq ? (o ? lbh_unord32(o) : -__builtin_inff()) : __builtin_inff();
        if (I8) {
            const float sc = qscale[q];
            ldsscale[q] = sc;
            if (q < nq && o) t = t / sc - 1.0f;  // integer units, one unit low: rounding here must never raise a threshold
        }
        ldsthr[q] = t;
#ifdef K3T_NOHIT  // timing-only build: no row ever reaches a threshold (what the scan costs without the harvest path)
        ldsthr[q] = __builtin_inff();
#endif
    }
    __syncthreads();

    // the i-th selected tile: tiles spread evenly over the index ((i * n_tiles) / n_sel in 32.32 fixed point; distinct
    // for distinct i because the step is >= 1)
    auto tile_of = [&](int64_t i) -> int64_t {
        const int64_t t = (int64_t)(((unsigned __int128)(uint64_t)i * step_q32) >> 32);
        return t < n_tiles ? t : n_tiles - 1;
    };
    auto load_tile = [&](f32x4 (&a)[KS], int64_t i) {  // always issued (index clamped): no load under a lane-dependent guard
        int64_t t = tile_of(i < n_sel ? i : n_sel - 1);
#ifdef K3T_L2WINDOW  // timing-only build: every tile comes from a 1 MiB, cache-resident window (what the scan costs without HBM)
        t &= 255;
#endif
        const f32x4* p = emb16t + t * TILE_V + lane;
#pragma unroll
        for (int s = 0; s < KS; ++s) a[s] = __builtin_nontemporal_load(p + s * 64);
    };
    auto load_b_lds = [&](f32x4 (&b)[KS], int g) {  // K3T_BREG: once per launch, straight from the global image
#pragma unroll
        for (int s = 0; s < KS; ++s) b[s] = K3T_BREG ? bfrag[(g * KS + s) * 64 + lane] : K3T_LDSB[(g * KS + s) * 64 + lane];
    };
    [[maybe_unused]] auto load_b = [&](f32x4 (&b)[KS], int g) {
#ifndef K3T_NOBREAD  // timing-only build: the B fragments are read once (what the scan costs without the LDS traffic)
        load_b_lds(b, g);
#endif
    };
    // one 32 x 32 block: 4 MFMAs, lane maximum as a 3-level tree of v_max3_f32, one compare, one wave vote
    auto block = [&](const f32x4 (&araw)[KS], const f32x4 (&braw)[KS], int g, int64_t row0) {
        if constexpr (I8) {
            i32x16 ai = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int s = 0; s < KS; ++s)
                ai = __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4, araw[s]), __builtin_bit_cast(i32x4, braw[s]), ai, 0, 0, 0);
            const float thr = ldsthr[g * 32 + j];
            auto mx3 = [](int a, int b, int c) { const int t = a > b ? a : b; return t > c ? t : c; };
            const int m0 = mx3(ai[0], ai[1], ai[2]), m1 = mx3(ai[3], ai[4], ai[5]), m2 = mx3(ai[6], ai[7], ai[8]);
            const int m3 = mx3(ai[9], ai[10], ai[11]), m4 = mx3(ai[12], ai[13], ai[14]);
            const int m5 = mx3(m0, m1, m2), m6 = mx3(m3, m4, ai[15]);
            const float mx = (float)(m5 > m6 ? m5 : m6);  // |sum| <= 64 x 127 x 127 < 2^24: exact
            const bool lanehit = mx >= thr;
            if (__any(lanehit)) {
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = (float)ai[r];
                cnt = k3t_harvest(acc, thr, ldsscale[g * 32 + j], lanehit, 32 * g + j, row0, n, ix.created, ix.tags, has_since, since_ms, tag_mask,
                                  se, sq, cnt, cand3, qtotal);
            }
            return;
        }
        f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if constexpr (!I8) {
#pragma unroll
        for (int s = 0; s < KS; ++s)
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, araw[s]), __builtin_bit_cast(f16x8, braw[s]), acc, 0, 0, 0);
        }
        const float thr = ldsthr[g * 32 + j];
        // (v_max3_f32 through inline asm would save the two quieting v_max(x, x) the compiler adds, but the hazard
        // recogniser does not see an asm operand as a reader of MFMA results: it dropped the wait states and rows were missed)
        const float m0 = __builtin_fmaxf(__builtin_fmaxf(acc[0], acc[1]), acc[2]);
        const float m1 = __builtin_fmaxf(__builtin_fmaxf(acc[3], acc[4]), acc[5]);
        const float m2 = __builtin_fmaxf(__builtin_fmaxf(acc[6], acc[7]), acc[8]);
        const float m3 = __builtin_fmaxf(__builtin_fmaxf(acc[9], acc[10]), acc[11]);
        const float m4 = __builtin_fmaxf(__builtin_fmaxf(acc[12], acc[13]), acc[14]);
        const float m5 = __builtin_fmaxf(__builtin_fmaxf(m0, m1), m2);
        const float m6 = __builtin_fmaxf(__builtin_fmaxf(m3, m4), acc[15]);
#ifdef K3T_NOEPI  // timing-only build: what the scan costs without the 16-way maximum
        const float mx = acc[0];
        (void)m5; (void)m6;
#else
        const float mx = __builtin_fmaxf(m5, m6);
#endif
        const bool lanehit = mx >= thr;
        if (__any(lanehit))  // rare: some query of this group has a row at or above its threshold
            cnt = k3t_harvest(acc, thr, 1.0f, lanehit, 32 * g + j, row0, n, ix.created, ix.tags, has_since, since_ms, tag_mask, se, sq,
                              cnt, cand3, qtotal);
    };

    const int64_t W = (int64_t)gridDim.x * K3T_WAVES;
    const int64_t i0 = (int64_t)blockIdx.x * K3T_WAVES + wave;

    // one tile against every query group; the next group's B fragments are read from LDS while this group's MFMAs run
    // (two register sets, ping-pong; after the last group the prefetch wraps to group 0 for the next tile)
#if K3T_BREG  // experiment: all eight groups' B fragments in registers (128 VGPRs), two waves per SIMD, no LDS reads in the loop
    f32x4 breg[8][KS];
#pragma unroll
    for (int g = 0; g < 8; ++g) load_b_lds(breg[g], g < ngroups ? g : 0);
    auto process = [&](const f32x4 (&araw)[KS], int64_t i) {
        const int64_t row0 = tile_of(i) * 32;
#pragma unroll
        for (int g = 0; g < 8; ++g)
            if (g < ngroups) block(araw, breg[g], g, row0);  // wave-uniform
    };
#else
    f32x4 b0[KS], b1[KS];
    load_b_lds(b0, 0);
#ifdef K3T_NOBREAD
    load_b_lds(b1, 0);
#endif
    auto process = [&](const f32x4 (&araw)[KS], int64_t i) {
        const int64_t row0 = tile_of(i) * 32;
        for (int g = 0; g < ngroups; g += 2) {
            load_b(b1, g + 1 < ngroups ? g + 1 : 0);
            block(araw, b0, g, row0);
            if (g + 1 < ngroups) {
                load_b(b0, g + 2 < ngroups ? g + 2 : 0);
                block(araw, b1, g + 1, row0);
            } else {
#pragma unroll
                for (int s = 0; s < KS; ++s) b0[s] = b1[s];  // odd group count: b1 already holds group 0
            }
        }
    };
#endif

    // K3T_RING claim tiles in flight per wave.  The ring loop is fully unrolled, so every register set has a static name
    // (rotating the sets with register moves would make each move wait for the youngest load).
    f32x4 ring[RING][KS];
#pragma unroll
    for (int r = 0; r < RING - 1; ++r) load_tile(ring[r], i0 + r * W);
    for (int64_t i = i0; i < n_sel; i += RING * W) {
#pragma unroll
        for (int r = 0; r < RING; ++r) {
            if (i + r * W < n_sel) {  // wave-uniform
                load_tile(ring[(r + RING - 1) % RING], i + (r + RING - 1) * W);
                process(ring[r], i + r * W);
            }
        }
    }
    if (cnt > 0) k3t_flush(se, sq, cnt, lane, cand3, qtotal);
}

// SAMPLE pass: group maxima instead of candidates.  The k-th largest of the maxima of G disjoint groups of rows is a
// lower bound of the k-th best score of the corpus (those k maxima belong to k distinct rows), so it seeds the thresholds
// of the full pass without listing, sorting or compacting anything: a workgroup runs the scan loop of the kernel above
// over its share of an evenly spread sample of tiles, every lane keeps one running maximum per query group in a register,
// and the workgroup writes ONE value per query -- slot blockIdx.x of the query's list, in the list's entry format, 0 when
// no row of the group passes the filters -- for seed_batched_kernel to pick the k-th largest of.  Scores come out of the
// same instruction sequence on the same operands as in the full pass, hence bit-identical.
template <bool I8>
__global__ __launch_bounds__(K3M_THREADS, K3M_WPS) void scan_batched_max_kernel(
    lbh_dev_index ix, const f32x4* __restrict__ emb16t, const f32x4* __restrict__ bfrag, int nq, int has_since, int64_t since_ms,
    uint64_t tag_mask, int64_t n_tiles, int64_t n_sel, uint64_t step_q32, uint64_t* __restrict__ cand3, const float* __restrict__ qscale) {
    constexpr int KS = I8 ? 2 : 4;
    constexpr int TILE_V = KS * 64;
    __shared__ f32x4 ldsb[8 * KS * 64];
    __shared__ uint32_t blkmax[BQ_MAXQ];
    __shared__ float wmax[K3M_WAVES][8 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int ngroups = (nq + 31) >> 5;
    const int64_t n = ix.n;
    const bool filtered = has_since || tag_mask;
    for (int i = tid; i < ngroups * TILE_V; i += K3M_THREADS) ldsb[i] = bfrag[i];
    for (int q = tid; q < BQ_MAXQ; q += K3M_THREADS) blkmax[q] = 0u;
    __syncthreads();
    auto tile_of = [&](int64_t i) -> int64_t {
        const int64_t t = (int64_t)(((unsigned __int128)(uint64_t)i * step_q32) >> 32);
        return t < n_tiles ? t : n_tiles - 1;
    };
    auto load_tile = [&](f32x4 (&a)[KS], int64_t i) {
        const int64_t t = tile_of(i < n_sel ? i : n_sel - 1);
        const f32x4* p = emb16t + t * TILE_V + lane;
#pragma unroll
        for (int s = 0; s < KS; ++s) a[s] = __builtin_nontemporal_load(p + s * 64);
    };
    // running maxima: one LDS word per (query group, lane), private to the wave (registers would need a fully unrolled
    // group loop, and the compiler then hoists all 8 groups' B fragments: 266 spilled registers)
    float* gmax = wmax[wave];
    for (int g = 0; g < 8; ++g) gmax[g * 64 + lane] = -__builtin_inff();
    auto load_b = [&](f32x4 (&b)[KS], int g) {
#pragma unroll
        for (int s = 0; s < KS; ++s) b[s] = ldsb[(g * KS + s) * 64 + lane];
    };
    uint32_t rmask = 0xFFFFFFFFu;
    bool masked = false;
    auto block = [&](const f32x4 (&araw)[KS], const f32x4 (&braw)[KS], int g) {
        f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if constexpr (I8) {  // exact integer sums, kept as floats in integer units (< 2^24: exact); real scores at the end
            i32x16 ai = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int s = 0; s < KS; ++s)
                ai = __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4, araw[s]), __builtin_bit_cast(i32x4, braw[s]), ai, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = (float)ai[r];
        } else {
#pragma unroll
            for (int s = 0; s < KS; ++s)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, araw[s]), __builtin_bit_cast(f16x8, braw[s]), acc, 0, 0, 0);
        }
        if (masked) {  // wave-uniform: filters, or the index's last tile
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rowbit = (r & 3) + 8 * (r >> 2) + 4 * h;
                if (!((rmask >> rowbit) & 1u)) acc[r] = -__builtin_inff();
            }
        }
        const float m0 = __builtin_fmaxf(__builtin_fmaxf(acc[0], acc[1]), acc[2]);
        const float m1 = __builtin_fmaxf(__builtin_fmaxf(acc[3], acc[4]), acc[5]);
        const float m2 = __builtin_fmaxf(__builtin_fmaxf(acc[6], acc[7]), acc[8]);
        const float m3 = __builtin_fmaxf(__builtin_fmaxf(acc[9], acc[10]), acc[11]);
        const float m4 = __builtin_fmaxf(__builtin_fmaxf(acc[12], acc[13]), acc[14]);
        const float m5 = __builtin_fmaxf(__builtin_fmaxf(m0, m1), m2);
        const float m6 = __builtin_fmaxf(__builtin_fmaxf(m3, m4), acc[15]);
        float* slot = &gmax[g * 64 + lane];
        *slot = __builtin_fmaxf(*slot, __builtin_fmaxf(m5, m6));
    };
    f32x4 b0[KS], b1[KS];
    load_b(b0, 0);
    auto process = [&](const f32x4 (&araw)[KS], int64_t i) {
        const int64_t row0 = tile_of(i) * 32;
        masked = filtered || row0 + 32 > n;
        if (masked) {
            bool rv = false;
            if (lane < 32) {
                const int64_t r = row0 + lane;
                rv = r < n;
                if (has_since) rv = rv && ix.created[rv ? r : 0] >= since_ms;
                if (tag_mask) rv = rv && (ix.tags[r < n ? r : 0] & tag_mask) != 0;
            }
            rmask = (uint32_t)__ballot(rv);
        }
        for (int g = 0; g < ngroups; g += 2) {
            load_b(b1, g + 1 < ngroups ? g + 1 : 0);
            block(araw, b0, g);
            if (g + 1 < ngroups) {
                load_b(b0, g + 2 < ngroups ? g + 2 : 0);
                block(araw, b1, g + 1);
            } else {
#pragma unroll
                for (int s = 0; s < KS; ++s) b0[s] = b1[s];
            }
        }
    };
    const int64_t W = (int64_t)gridDim.x * K3M_WAVES;
    const int64_t i0 = (int64_t)blockIdx.x * K3M_WAVES + wave;
    f32x4 ring[2][KS];
    load_tile(ring[0], i0);
    for (int64_t i = i0; i < n_sel; i += 2 * W) {
        load_tile(ring[1], i + W);
        process(ring[0], i);
        if (i + W < n_sel) {
            load_tile(ring[0], i + 2 * W);
            process(ring[1], i + W);
        }
    }
    __builtin_amdgcn_wave_barrier();
    for (int g = 0; g < ngroups; ++g) {
        float m = __builtin_fmaxf(gmax[g * 64 + j], gmax[g * 64 + 32 + j]);
        if (I8 && m > -__builtin_inff()) m = m * qscale[32 * g + j];  // integer units -> score, the multiplication the full pass's harvest does
        if (lane < 32 && m > -__builtin_inff()) atomicMax(&blkmax[32 * g + j], lbh_ord32(m));
    }
    __syncthreads();
    for (int q = tid; q < nq; q += K3M_THREADS) cand3[(size_t)q * BQ_PERQ + blockIdx.x] = (uint64_t)blkmax[q] << 32;
}

// Per-query seed threshold from the sample pass: (k-th best fp32 score of the sample) - margin.  The
// sample is a subset of the corpus, so this never exceeds the final cut (global k-th - margin).
// dense > 0: the list holds exactly `dense` slots written by the SAMPLE pass (one group maximum each; 0, which sorts below
// every score, for a group without a row that passes the filters).
__global__ __launch_bounds__(SEL_THREADS) void seed_batched_kernel(int k, float margin, const uint64_t* __restrict__ cand3,
                                                                    int32_t* __restrict__ qtotal,
                                                                    uint32_t* __restrict__ seed, int dense) {
    __shared__ int hist[256];
    __shared__ int64_t sh_krem;
    __shared__ int sh_bin;
    __shared__ uint32_t sord[BQ_PERQ];  // the list's scores (ordered images), staged once: the select's five passes read LDS
    const int q = blockIdx.x;
    const int raw = qtotal[q * BQ_QT];
    int E = dense > 0 ? dense : (raw & (BQ_OVER - 1));
    if (E > BQ_PERQ) E = BQ_PERQ;
    const uint64_t* cand = cand3 + (size_t)q * BQ_PERQ;
    uint32_t thr = 0;
    if (E >= k) {
        for (int e = threadIdx.x; e < E; e += SEL_THREADS) sord[e] = (uint32_t)(cand[e] >> 32);
        __syncthreads();
        auto val = [&](int e, uint64_t* v) {
            *v = sord[e];
            return true;
        };
        const uint32_t kth = (uint32_t)block_radix_select(E, 4, k, val, hist, &sh_krem, &sh_bin);
        thr = kth ? lbh_ord32(lbh_unord32(kth) - margin) : 0u;  // kth == 0: fewer than k real scores in the sample
    }
    __syncthreads();
    if (threadIdx.x == 0) { seed[q] = thr; qtotal[q * BQ_QT] = raw & BQ_OVER; }
}

// K4 for a batch: one block per query.
__global__ __launch_bounds__(SEL_THREADS) void select_batched_kernel(lbh_dev_index ix, const double* __restrict__ qmat,
                                                                      int k, float margin, const uint64_t* __restrict__ cand3,
                                                                      const int32_t* __restrict__ qtotal,
                                                                      const int32_t* __restrict__ flags,
                                                                      lbh_result* __restrict__ res_arr, int seq, int out_off) {
    __shared__ double qs[64];
    const int tid = threadIdx.x, q = blockIdx.x;
    lbh_result* res = res_arr + q;
    if (tid < 64) qs[tid] = qmat[(size_t)q * 64 + tid];
    __syncthreads();
    const int raw = qtotal[q * BQ_QT];
    if (flags[0] || (raw & BQ_OVER)) {
        if (tid == 0) { res->count = 0; res->overflow = 1; res->candidates = 0; }
        publish_result(res, seq);
        return;
    }
    int E = raw;
    if (E > BQ_PERQ) E = BQ_PERQ;
    select_fast_body(ix, qs, k, cand3 + (size_t)q * BQ_PERQ, E, res, margin, lbh_xchg{nullptr, 0, 0}, out_off);
    publish_result(res, seq);  // every query's block carries its own completion ticket (mapped host memory)
}

// ------------------------------------------------------------------------------------------------
// Exact path.  One lane per row: 64 sequential fp64 multiply-adds in the reference's order.
// Per-wave top-k under the full (score desc, key asc) order, so ties never overflow a buffer.
// ------------------------------------------------------------------------------------------------
template <bool SINCE, bool TAGS>
__global__ __launch_bounds__(LBH_SCAN_THREADS) void scan_exact_kernel(lbh_dev_index ix, const double* __restrict__ qptr, lbh_qarg qa,
                                                                       int k, int64_t since_ms, uint64_t tag_mask,
                                                                       lbh_cursor cur,
                                                                       uint64_t* __restrict__ xc_ord,
                                                                       uint32_t* __restrict__ xc_row,
                                                                       int32_t* __restrict__ flags,
                                                                       const int32_t* __restrict__ gate) {
    // gate (allocation-free entry point only): run only if the fast path reported an unbounded candidate set
    if (gate && *gate != 1) return;
    __shared__ uint64_t wo[LBH_SCAN_WAVES][LBH_WCAP];
    __shared__ uint32_t wr[LBH_SCAN_WAVES][LBH_WCAP];
    __shared__ double qs[64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    if (threadIdx.x < 64) qs[threadIdx.x] = qptr ? qptr[threadIdx.x] : qa.q[threadIdx.x];
    __syncthreads();
    uint64_t* bo = wo[wave];
    uint32_t* br = wr[wave];
    const int64_t n = ix.n;
    const int64_t n_tiles = (n + 63) >> 6;
    int cnt = 0;
    bool have_thr = false;
    uint64_t thr_ord = 0;
    uint32_t thr_row = PAD_ROW;

    auto compact = [&]() {
        const int n2 = pow2_at_least(cnt, 64);
        for (int i = cnt + lane; i < n2; i += 64) { bo[i] = 0; br[i] = PAD_ROW; }
        __builtin_amdgcn_wave_barrier();
        sort_exact(bo, br, n2, lane, 64, ix.keys, WaveSync());
        if (cnt >= k) {
            cnt = k;
            have_thr = true;
            thr_ord = bo[k - 1];
            thr_row = br[k - 1];
        }
    };

    for (int64_t tile = (int64_t)blockIdx.x * LBH_SCAN_WAVES + wave; tile < n_tiles;
         tile += (int64_t)gridDim.x * LBH_SCAN_WAVES) {
        const int64_t myrow = tile * 64 + lane;
        bool valid = myrow < n;
        const int64_t r = valid ? myrow : 0;
        if (SINCE) valid = valid && (ix.created[r] >= since_ms);
        if (TAGS) valid = valid && ((ix.tags[r] & tag_mask) != 0);
        uint64_t o = 0;
        if (valid) o = lbh_ord64(exact_dot(qs, ix.emb64 + (size_t)r * 64));
        if (valid && cur.active) {  // strictly after the cursor in (score desc, key asc) order
            if (o > cur.ord) valid = false;
            else if (o == cur.ord) {
                const uint64_t* kk = ix.keys + (size_t)r * 4;
                bool greater = false, decided = false;
#pragma unroll
                for (int w = 0; w < 4; ++w)
                    if (!decided && kk[w] != cur.key[w]) { greater = kk[w] > cur.key[w]; decided = true; }
                valid = greater;
            }
        }
        bool pass = valid;
        if (pass && have_thr) pass = before_exact(o, (uint32_t)r, thr_ord, thr_row, ix.keys);
        const uint64_t m = __ballot(pass);
        if (m) {
            if (pass) {
                const int pos = cnt + __popcll(m & lt_mask);
                bo[pos] = o;
                br[pos] = (uint32_t)r;
            }
            cnt += __popcll(m);
            __builtin_amdgcn_wave_barrier();
            if (cnt > LBH_WCAP - 64) compact();
        }
    }
    if (cnt > 0) compact();
    int base = 0;
    if (lane == 0 && cnt > 0) base = atomicAdd(&flags[2], cnt);
    base = __shfl(base, 0);
    for (int i = lane; i < cnt; i += 64) {
        xc_ord[base + i] = bo[i];
        xc_row[base + i] = br[i];
    }
}

__global__ __launch_bounds__(SEL_THREADS) void select_exact_kernel(lbh_dev_index ix, int k,
                                                                    const uint64_t* __restrict__ xc_ord,
                                                                    const uint32_t* __restrict__ xc_row,
                                                                    int32_t* __restrict__ flags,
                                                                    lbh_result* __restrict__ res, int seq,
                                                                    const int32_t* __restrict__ gate, lbh_xchg xc) {
    __shared__ uint64_t sbuf[2 * LBH_MCAP];
    __shared__ int hist[256];
    __shared__ int64_t sh_krem;
    __shared__ int sh_bin, sh_cnt, sh_gt, sh_eq;
    const int tid = threadIdx.x;
    if (gate && *gate != 1) return;  // block-uniform; gate points into *res, which is first written after the barrier below
    if (tid == 0) { sh_cnt = 0; sh_gt = 0; sh_eq = 0; }
    const int E = flags[2];
    __syncthreads();
    if (tid < 4) flags[tid] = 0;
    if (tid >= 64 && tid < 128) flags[tid] = 0;
    uint64_t* xo = sbuf;
    uint32_t* xr = reinterpret_cast<uint32_t*>(sbuf + LBH_MCAP);

    uint64_t kth = 0;       // admit everything unless there are more entries than the final sort holds
    uint64_t cut[4] = {~0ull, ~0ull, ~0ull, ~0ull};  // mass tie on the k-th score: the tie group is cut by the hash key,
    int ncut = 0;                                     // one 64-bit word at a time (ascending), until what is left fits
    if (E > LBH_MCAP) {
        auto val = [&](int e, uint64_t* v) { *v = xc_ord[e]; return true; };
        kth = block_radix_select(E, 8, k, val, hist, &sh_krem, &sh_bin);
        int gt = 0, eq = 0;
        for (int e = tid; e < E; e += SEL_THREADS) {
            const uint64_t o = xc_ord[e];
            gt += o > kth;
            eq += o == kth;
        }
        if (gt) atomicAdd(&sh_gt, gt);
        if (eq) atomicAdd(&sh_eq, eq);
        __syncthreads();
        int64_t kept = sh_gt;            // entries that are certainly in
        int64_t need = k - sh_gt;        // how many of the current tie group are wanted (>= 1)
        bool fits = sh_gt + sh_eq <= LBH_MCAP;
        for (int w = 0; w < 4 && !fits; ++w) {
            // among the entries that tie on the score and on key words 0..w-1: the need-th smallest word w
            auto kval = [&](int e, uint64_t* v) {
                if (xc_ord[e] != kth) return false;
                const uint64_t* kp = ix.keys + (size_t)xc_row[e] * 4;
                for (int j2 = 0; j2 < w; ++j2)
                    if (kp[j2] != cut[j2]) return false;
                *v = ~kp[w];
                return true;
            };
            cut[w] = ~block_radix_select(E, 8, need, kval, hist, &sh_krem, &sh_bin);
            ncut = w + 1;
            __syncthreads();
            if (tid == 0) { sh_gt = 0; sh_eq = 0; }   // re-used: group members below / at the cut word
            __syncthreads();
            int lt = 0, eqw = 0;
            for (int e = tid; e < E; e += SEL_THREADS) {
                uint64_t v;
                if (kval(e, &v)) {
                    lt += ~v < cut[w];
                    eqw += ~v == cut[w];
                }
            }
            if (lt) atomicAdd(&sh_gt, lt);
            if (eqw) atomicAdd(&sh_eq, eqw);
            __syncthreads();
            fits = kept + sh_gt + sh_eq <= LBH_MCAP;  // keeping everything up to and including the cut word fits
            kept += sh_gt;
            need -= sh_gt;
            __syncthreads();
        }
    }
    for (int e = tid; e < E; e += SEL_THREADS) {
        const uint64_t o = xc_ord[e];
        const uint32_t r = xc_row[e];
        bool in = o >= kth;
        if (in && ncut && o == kth) {  // lexicographic: key words 0..ncut-1 <= cut
            const uint64_t* kp = ix.keys + (size_t)r * 4;
            for (int w = 0; w < ncut; ++w) {
                if (kp[w] != cut[w]) { in = kp[w] < cut[w]; break; }
            }
        }
        if (in) {
            const int pos = atomicAdd(&sh_cnt, 1);
            if (pos < LBH_MCAP) { xo[pos] = o; xr[pos] = r; }
        }
    }
    __syncthreads();
    const int M = sh_cnt;
    if (M > LBH_MCAP) {
        if (tid == 0) {
            res->count = 0; res->overflow = 2; res->candidates = M;
            if (xc.dst) xc.dst[0] = -1;  // duplicate keys: every rank reports the error after the exchange
        }
    } else {
        rerank_and_emit(xo, xr, M, k, ix, nullptr, res, true, xc);
    }
    publish_result(res, seq);
}

// ------------------------------------------------------------------------------------------------
// N3: recency query (get_recent_claims, lb/node.py:736-762): live rows with created_ms >= since_ms,
// newest first.  The reference sorts with list.sort(key=created_ms, reverse=True), which is stable:
// claims that share a timestamp keep the claims dict's insertion order, i.e. ascending `order`.
// With a = created_ms ^ signbit and b = ~order the answer is the top-`page` of UNIQUE 128-bit keys
// (a, b) in descending order -- integer work, no tolerance anywhere.
//
// recent_scan_kernel streams created[] once (8 B/row, 16-byte loads, 1024 rows per wave step) from the
// END of the index backwards: rows are appended in publish order, so the newest claims come first, the
// per-wave threshold (its page-th best key so far) is tight after the first step and order[] is only
// read for rows that reach that threshold.  Each block emits its best <= page entries to a flat array;
// recent_select_kernel (one block) cuts that array to the global top-page and sorts it.
// ------------------------------------------------------------------------------------------------
constexpr int RC_WCAP = 256;    // per-wave candidate buffer: <= 128 kept + one 128-row group
constexpr int RC_GROUPS = 8;    // 128-row groups per wave step: 8 x 16-byte loads in flight per lane
constexpr int RC_STEP = RC_GROUPS * 128;

__device__ __forceinline__ bool rc_before(uint64_t a1, uint64_t b1, uint32_t r1, uint64_t a2, uint64_t b2, uint32_t r2) {
    if (a1 != a2) return a1 > a2;
    if (b1 != b2) return b1 > b2;
    return r1 < r2;  // only padding (row = PAD_ROW) can tie on (a, b); it sorts last
}

template <class Sync>
__device__ __forceinline__ void sort_recent(uint64_t* sa, uint64_t* sb, uint32_t* sr, int n, int tid, int nthreads, Sync sync) {
    for (int k = 2; k <= n; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < (n >> 1); t += nthreads) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const int l = i | j;
                const bool fwd = (i & k) == 0;
                const uint64_t a1 = sa[i], a2 = sa[l], b1 = sb[i], b2 = sb[l];
                const uint32_t r1 = sr[i], r2 = sr[l];
                const bool swap = fwd ? rc_before(a2, b2, r2, a1, b1, r1) : rc_before(a1, b1, r1, a2, b2, r2);
                if (swap) { sa[i] = a2; sa[l] = a1; sb[i] = b2; sb[l] = b1; sr[i] = r2; sr[l] = r1; }
            }
            sync();
        }
    }
}

typedef long long i64x2 __attribute__((ext_vector_type(2)));

// Threshold seed: the page-th best key inside a 256-row window is a valid lower bound of the global
// page-th best (a subset's k-th best never exceeds the whole set's).  The window sits where the next
// results are expected when claims were appended in time order: at the end of the index for the first
// page, just below the previous page's rows afterwards.  With a seed the scan admits only the handful of
// rows that can matter, so no wave spends time sorting candidates it will throw away; on corpora in any
// other order the seed is merely loose.  seed[0] = 1 when set, seed[1..2] = (a, b) admitted INCLUSIVELY.
__global__ __launch_bounds__(256) void recent_seed_kernel(const int64_t* __restrict__ created,
                                                           const uint64_t* __restrict__ order, int64_t n, int has_since,
                                                           int64_t since_ms, lbh_rcursor cur, int page, int64_t w0,
                                                           uint64_t* __restrict__ seed) {
    __shared__ uint64_t sa[256], sb[256];
    __shared__ int sv[256];
    constexpr uint64_t SIGN = 0x8000000000000000ull;
    const int tid = threadIdx.x;
    const int64_t row = w0 + tid;
    bool valid = row >= 0 && row < n;
    uint64_t a = 0, b = 0;
    if (valid) {
        const int64_t c = created[row];
        a = (uint64_t)c ^ SIGN;
        b = ~order[row];
        if (has_since && c < since_ms) valid = false;
        if (cur.active && !(a < cur.a || (a == cur.a && b < cur.b))) valid = false;
    }
    sa[tid] = a;
    sb[tid] = b;
    sv[tid] = valid;
    if (tid == 0) seed[0] = 0;
    __syncthreads();
    if (!valid) return;
    int better = 0;
    for (int i = 0; i < 256; ++i) better += (sv[i] && (sa[i] > a || (sa[i] == a && sb[i] > b))) ? 1 : 0;
    if (better == page - 1) {  // keys are unique: exactly one thread, and only if the window holds >= page rows
        seed[1] = a;
        seed[2] = b;
        __threadfence();
        seed[0] = 1;
    }
}

template <bool SINCE>
__global__ __launch_bounds__(LBH_SCAN_THREADS) void recent_scan_kernel(const int64_t* __restrict__ created,
                                                                        const uint64_t* __restrict__ order, int64_t n,
                                                                        int64_t since_ms, lbh_rcursor cur, int page,
                                                                        uint64_t* __restrict__ ca, uint64_t* __restrict__ cb,
                                                                        uint32_t* __restrict__ crow,
                                                                        unsigned long long* __restrict__ counters,
                                                                        const uint64_t* __restrict__ seed) {
    __shared__ uint64_t wa[LBH_SCAN_WAVES][RC_WCAP], wb[LBH_SCAN_WAVES][RC_WCAP];
    __shared__ uint32_t wr[LBH_SCAN_WAVES][RC_WCAP];
    __shared__ uint64_t ma[LBH_SCAN_WAVES * LBH_MAX_PAGE], mb[LBH_SCAN_WAVES * LBH_MAX_PAGE];
    __shared__ uint32_t mr[LBH_SCAN_WAVES * LBH_MAX_PAGE];
    __shared__ int wcnt[LBH_SCAN_WAVES];
    __shared__ unsigned long long sh_match;
    __shared__ int gbase;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    constexpr uint64_t SIGN = 0x8000000000000000ull;
    uint64_t* A = wa[wave];
    uint64_t* B = wb[wave];
    uint32_t* R = wr[wave];
    if (threadIdx.x == 0) sh_match = 0;
    __syncthreads();

    bool have_thr = false;
    uint64_t ta = 0, tb = 0;  // the wave's page-th best key so far: only strictly better keys are kept
    if (seed[0]) {            // start from the seed; it is inclusive, so step one key below it
        const uint64_t s_a = seed[1], s_b = seed[2];
        if (s_b != 0) { have_thr = true; ta = s_a; tb = s_b - 1; }
        else if (s_a != 0) { have_thr = true; ta = s_a - 1; tb = ~0ull; }
    }
    int cnt = 0;
    unsigned long long nmatch = 0;
    const int64_t n_tiles = (n + RC_STEP - 1) / RC_STEP;

    auto compact = [&]() {
        const int n2 = pow2_at_least(cnt, 64);
        for (int i = cnt + lane; i < n2; i += 64) { A[i] = 0; B[i] = 0; R[i] = PAD_ROW; }
        __builtin_amdgcn_wave_barrier();
        sort_recent(A, B, R, n2, lane, 64, WaveSync());
        if (cnt >= page) {  // never below the seed: a wave's own page-th best is at least as good
            cnt = page;
            ta = A[page - 1];
            tb = B[page - 1];
            have_thr = true;
        }
        __builtin_amdgcn_wave_barrier();
    };

    for (int64_t t = (int64_t)blockIdx.x * LBH_SCAN_WAVES + wave; t < n_tiles; t += (int64_t)gridDim.x * LBH_SCAN_WAVES) {
        const int64_t base = (n_tiles - 1 - t) * RC_STEP;
        int64_t cr[RC_GROUPS][2];
        if (base + RC_STEP <= n) {
#pragma unroll
            for (int j = 0; j < RC_GROUPS; ++j) {
                const i64x2 v = *reinterpret_cast<const i64x2*>(created + base + j * 128 + 2 * lane);
                cr[j][0] = v.x;
                cr[j][1] = v.y;
            }
        } else {
#pragma unroll
            for (int j = 0; j < RC_GROUPS; ++j) {
                const int64_t r = base + j * 128 + 2 * lane;
                cr[j][0] = r < n ? created[r] : 0;
                cr[j][1] = r + 1 < n ? created[r + 1] : 0;
            }
        }
#pragma unroll
        for (int j = 0; j < RC_GROUPS; ++j) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int64_t row = base + j * 128 + 2 * lane + h;
                const int64_t c = cr[j][h];
                bool valid = row < n;
                if (SINCE) valid = valid && c >= since_ms;
                nmatch += valid ? 1u : 0u;
                const uint64_t a = (uint64_t)c ^ SIGN;
                const bool maybe = valid && (!have_thr || a >= ta) && (!cur.active || a <= cur.a);
                uint64_t b = 0;
                if (maybe) b = ~order[row];
                const bool pass = maybe && (!have_thr || a > ta || b > tb) && (!cur.active || a < cur.a || b < cur.b);
                const uint64_t m = __ballot(pass);
                if (m) {
                    if (pass) {
                        const int pos = cnt + __popcll(m & lt_mask);
                        A[pos] = a;
                        B[pos] = b;
                        R[pos] = (uint32_t)row;
                    }
                    cnt += __popcll(m);
                }
            }
            __builtin_amdgcn_wave_barrier();
            if (cnt > RC_WCAP - 128) compact();
        }
    }
    if (cnt > 0) compact();
    if (nmatch) atomicAdd(&sh_match, nmatch);
    if (lane == 0) wcnt[wave] = cnt;
    __syncthreads();

    int off = 0, total = 0;
#pragma unroll
    for (int w = 0; w < LBH_SCAN_WAVES; ++w) {
        if (w < wave) off += wcnt[w];
        total += wcnt[w];
    }
    if (threadIdx.x == 0 && sh_match) atomicAdd(&counters[1], sh_match);
    if (total == 0) return;
    for (int i = lane; i < cnt; i += 64) { ma[off + i] = A[i]; mb[off + i] = B[i]; mr[off + i] = R[i]; }
    const int n2 = pow2_at_least(total, 2);
    for (int i = total + threadIdx.x; i < n2; i += LBH_SCAN_THREADS) { ma[i] = 0; mb[i] = 0; mr[i] = PAD_ROW; }
    __syncthreads();
    sort_recent(ma, mb, mr, n2, threadIdx.x, LBH_SCAN_THREADS, BlockSync());
    const int ns = total < page ? total : page;
    if (threadIdx.x == 0) gbase = (int)atomicAdd(&counters[0], (unsigned long long)ns);
    __syncthreads();
    const int gb = gbase;
    for (int i = threadIdx.x; i < ns; i += LBH_SCAN_THREADS) { ca[gb + i] = ma[i]; cb[gb + i] = mb[i]; crow[gb + i] = mr[i]; }
}

__global__ __launch_bounds__(SEL_THREADS) void recent_select_kernel(int page, const uint64_t* __restrict__ ca,
                                                                     const uint64_t* __restrict__ cb,
                                                                     const uint32_t* __restrict__ crow,
                                                                     unsigned long long* __restrict__ counters,
                                                                     lbh_recent_result* __restrict__ res, int seq) {
    __shared__ uint64_t sa[LBH_MCAP], sb[LBH_MCAP];
    __shared__ uint32_t sr[LBH_MCAP];
    __shared__ int hist[256];
    __shared__ int64_t sh_krem;
    __shared__ int sh_bin, sh_cnt, sh_gt, sh_eq;
    const int tid = threadIdx.x;
    constexpr uint64_t SIGN = 0x8000000000000000ull;
    const int E = (int)counters[0];
    const unsigned long long matched = counters[1];
    if (tid == 0) { sh_cnt = 0; sh_gt = 0; sh_eq = 0; }
    __syncthreads();
    if (tid < 2) counters[tid] = 0;  // ready for the next launch

    uint64_t T = 0, U = 0;
    bool cut_b = false;
    if (E > LBH_MCAP) {
        auto val = [&](int e, uint64_t* v) { *v = ca[e]; return true; };
        T = block_radix_select(E, 8, page, val, hist, &sh_krem, &sh_bin);
        int gt = 0, eq = 0;
        for (int e = tid; e < E; e += SEL_THREADS) {
            const uint64_t a = ca[e];
            gt += a > T;
            eq += a == T;
        }
        if (gt) atomicAdd(&sh_gt, gt);
        if (eq) atomicAdd(&sh_eq, eq);
        __syncthreads();
        if (sh_gt + sh_eq > LBH_MCAP) {  // mass tie on the page-th timestamp: cut the tie group by insertion order
            const int64_t need = page - sh_gt;
            auto bval = [&](int e, uint64_t* v) {
                if (ca[e] != T) return false;
                *v = cb[e];
                return true;
            };
            U = block_radix_select(E, 8, need, bval, hist, &sh_krem, &sh_bin);
            cut_b = true;
        }
    }
    for (int e = tid; e < E; e += SEL_THREADS) {
        const uint64_t a = ca[e], b = cb[e];
        bool in = a >= T;
        if (in && cut_b && a == T) in = b >= U;
        if (in) {
            const int pos = atomicAdd(&sh_cnt, 1);
            if (pos < LBH_MCAP) { sa[pos] = a; sb[pos] = b; sr[pos] = crow[e]; }
        }
    }
    __syncthreads();
    const int M = sh_cnt < LBH_MCAP ? sh_cnt : LBH_MCAP;  // (a, b) is unique, so sh_cnt <= LBH_MCAP by construction
    const int n2 = pow2_at_least(M, 2);
    for (int j = M + tid; j < n2; j += SEL_THREADS) { sa[j] = 0; sb[j] = 0; sr[j] = PAD_ROW; }
    __syncthreads();
    sort_recent(sa, sb, sr, n2, tid, SEL_THREADS, BlockSync());
    const int cnt = M < page ? M : page;
    if (tid < cnt) {
        res->rows[tid] = sr[tid];
        res->created[tid] = (int64_t)(sa[tid] ^ SIGN);
        res->order[tid] = ~sb[tid];
    }
    if (tid == 0) { res->count = cnt; res->matched = (int64_t)matched; }
    __threadfence_system();
    __syncthreads();
    if (tid == 0) __hip_atomic_store(&res->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

}  // namespace

hipError_t lbh_launch_recent(const int64_t* created, const uint64_t* order, int64_t n, const lbh_workspace& ws,
                             int page, int has_since, int64_t since_ms, const lbh_rcursor& cur, int64_t anchor_row,
                             int n_cu, int seq, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1) {
    const int64_t steps = (n + RC_STEP - 1) / RC_STEP;
    int64_t blocks = (steps + LBH_SCAN_WAVES - 1) / LBH_SCAN_WAVES;
    const int64_t cap = 2 * (int64_t)n_cu < LBH_RC_MAX_BLOCKS ? 2 * (int64_t)n_cu : LBH_RC_MAX_BLOCKS;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    uint64_t* seed = reinterpret_cast<uint64_t*>(ws.rc_cnt) + 4;
    int64_t w0 = (anchor_row < n ? anchor_row : n) - 192;  // 192 rows below the anchor, 64 above
    if (w0 + 256 > n) w0 = n - 256;
    recent_seed_kernel<<<1, 256, 0, s>>>(created, order, n, has_since, since_ms, cur, page, w0, seed);
    if (ev0) (void)hipEventRecord(ev0, s);
    const dim3 g((unsigned)blocks), b(LBH_SCAN_THREADS);
    if (has_since) recent_scan_kernel<true><<<g, b, 0, s>>>(created, order, n, since_ms, cur, page, ws.rc_a, ws.rc_b, ws.rc_row, ws.rc_cnt, seed);
    else recent_scan_kernel<false><<<g, b, 0, s>>>(created, order, n, since_ms, cur, page, ws.rc_a, ws.rc_b, ws.rc_row, ws.rc_cnt, seed);
    if (ev1) (void)hipEventRecord(ev1, s);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    recent_select_kernel<<<1, SEL_THREADS, 0, s>>>(page, ws.rc_a, ws.rc_b, ws.rc_row, ws.rc_cnt, ws.rc_result_dev, seq);
    return hipGetLastError();
}

int lbh_scan_grid(int64_t n_rows, int n_cu, int blocks_per_cu, bool half) {
    // blocks_per_cu <= 0: by size.  (Re-measured with the contiguous tile ranges: 2.5 M rows 101 us with one block per CU
    // against 108 with two, 3.5 M 144 / 154 (whole query 175 / 180), 4 M 157.6 / 156.9 (whole query 189 / 182) -- the switch
    // sits at 3.5 M.)  Up to ~3.5 M rows ONE 4-wave block per CU is faster (1 M rows: scan 56 -> 50 us, 300 k:
    // 24.8 -> 23.2, 100 k: 16.2 -> 14.4): half as many block epilogues and candidate lists, and the streaming loop has too
    // few tiles per wave for a second block to hide anything.  From ~3 M rows on two blocks per CU win by 1-3 %
    // (tools/scripts/scan_block_sweep.sh; 512-thread blocks were measured too: within noise of these two).
    // `half`: the UNFILTERED fp16 flavour (half as many bytes per tile) is faster with one block per CU at every size
    // measured (10 M rows: 189.4 against 195.2 us, 100 M: 1749 against 1820; tools/scripts/scan_split_ab_fp16.sh).  With a
    // since / tag mask its extra per-row loads want the second block's waves to hide behind (100 M rows, since filter:
    // 2148 us with one block per CU, 1921 with two), so the callers ask for this only without filters.
    if (blocks_per_cu <= 0) blocks_per_cu = (half || n_rows <= 3500000) ? 1 : 2;
    const int64_t tiles = (n_rows + 63) / 64;
    int64_t blocks = (tiles + LBH_SCAN_WAVES - 1) / LBH_SCAN_WAVES;
    const int64_t cap = (int64_t)n_cu * blocks_per_cu < LBH_MAX_BLOCKS ? (int64_t)n_cu * blocks_per_cu : LBH_MAX_BLOCKS;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

namespace {
uint32_t g_split_w[4] = {0, 0, 0, 0};
bool g_split_set = false;
}  // namespace

void lbh_set_scan_split(const uint32_t w[4]) {
    for (int i = 0; i < 4; ++i) g_split_w[i] = w[i];
    g_split_set = true;
}

lbh_split lbh_scan_split(int64_t n_rows, int grid, int n_cu) {
    lbh_split sp;
    sp.n_first = grid < n_cu ? grid : n_cu;
    // Defaults (tools/scripts/scan_split_ab.sh, same-box A/B against the grid-stride loop).  One block per CU: equal
    // contiguous ranges -- the loop-end spread of the 1 M-claim scan narrows (last block 44 -> 40 us) and the scan goes
    // from 50.3-51.0 to 47.8-48.1 us; shares by XCD parity change nothing on top.  Two blocks per CU: shares proportional
    // to the rate at which each class streamed under equal shares (the second block on a CU is ~15 % slower than the
    // first, odd XCDs ~10 % slower than even ones): 10 M claims 376-386 -> 374-376 us.  The end time barely moves with
    // the shares -- every variant, perfectly levelled ones included, ends at 368-373 us, i.e. at ~6.9 TB/s: that is the
    // rate this memory system sustains, not a balance problem.
    // (constants from one box's block timelines; checked on three other GPUs in round 3: 1.3-2.7 % ahead of equal shares and
    // of the grid-stride loop on each, profiles/r03_scan_split_three_boxes.txt)
    static const uint32_t one[4] = {1000, 1000, 1000, 1000}, two[4] = {1127, 1016, 977, 880};
    const uint32_t* w = g_split_set ? g_split_w : (grid > n_cu ? two : one);
    for (int i = 0; i < 4; ++i) sp.w[i] = w[i];
    const int64_t tiles = (n_rows + 63) / 64;
    if (sp.w[0] == 0 || sp.w[1] == 0 || sp.w[2] == 0 || sp.w[3] == 0 || tiles < 2 * (int64_t)grid * LBH_SCAN_WAVES)
        sp.w[0] = sp.w[1] = sp.w[2] = sp.w[3] = 0;  // off, or fewer than two tiles per wave: the grid-stride loop
    return sp;
}

hipError_t lbh_launch_search_fast(const lbh_dev_index& ix, const uint16_t* emb16, const lbh_workspace& ws, const double* qptr,
                                  const lbh_qarg& qa, int k, int has_since,
                                  int64_t since_ms, uint64_t tag_mask, int grid, int n_cu, int nt, int seq, float margin,
                                  hipStream_t s, hipEvent_t ev0, hipEvent_t ev1, const lbh_xchg* xc, int fuse, int xb_k) {
    hipError_t e;
    // fuse: 0 never, 1 when the grid is one block per CU AND the index is small or medium (up to 4 M rows: there the
    // hand-off between two launches is a visible share of the query; beyond, e.g. the fp16 flavour's one-block grid
    // at 10 M rows, it buys < 1 % and the scan kernel keeps a duration of its own), 2 always
    // (a claim-sharded search fuses too: the select step writes the rank's packed list either way, and the gated exact
    // pass and the all-gather behind it are stream-ordered after this launch)
    fuse = (fuse >= 2 || (fuse == 1 && grid <= n_cu && ix.n <= 4000000u)) && ws.bmax;
    // the fp16 flavour keeps the grid-stride loop: contiguous ranges cost it 2-3 % (10 M rows: 195 -> 198-201 us)
    const lbh_split sp = emb16 ? lbh_split{{0, 0, 0, 0}, 0} : lbh_scan_split(ix.n, grid, n_cu);
    if (ev0) (void)hipEventRecord(ev0, s);
    const dim3 g(grid), b(LBH_SCAN_THREADS);
    const lbh_xchg xcv = xc ? *xc : lbh_xchg{nullptr, 0, 0};
#define LBH_SCAN_ARGS ix, emb16, qptr, qa, k, margin, since_ms, tag_mask, ws.cand, ws.flags, sp, ws.result_dev, seq, xcv, ws.bmax, xb_k
#define LBH_SCAN_DISPATCH(NTV, HV, FV)                                                                           \
    do {                                                                                                         \
        if (has_since && tag_mask) scan_f32_kernel<true, true, NTV, HV, FV><<<g, b, 0, s>>>(LBH_SCAN_ARGS);      \
        else if (has_since) scan_f32_kernel<true, false, NTV, HV, FV><<<g, b, 0, s>>>(LBH_SCAN_ARGS);            \
        else if (tag_mask) scan_f32_kernel<false, true, NTV, HV, FV><<<g, b, 0, s>>>(LBH_SCAN_ARGS);             \
        else scan_f32_kernel<false, false, NTV, HV, FV><<<g, b, 0, s>>>(LBH_SCAN_ARGS);                          \
    } while (0)
#define LBH_SCAN_DISPATCH2(NTV, HV)          \
    do {                                     \
        if (fuse) LBH_SCAN_DISPATCH(NTV, HV, true); \
        else LBH_SCAN_DISPATCH(NTV, HV, false);     \
    } while (0)
    if (emb16) {
        if (nt) LBH_SCAN_DISPATCH2(true, true);
        else LBH_SCAN_DISPATCH2(false, true);
    } else {
        if (nt) LBH_SCAN_DISPATCH2(true, false);
        else LBH_SCAN_DISPATCH2(false, false);
    }
#undef LBH_SCAN_DISPATCH2
#undef LBH_SCAN_DISPATCH
#undef LBH_SCAN_ARGS
    if (ev1) (void)hipEventRecord(ev1, s);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (fuse) return hipSuccess;  // the scan's last block publishes the result
    select_fast_kernel<<<1, SEL_THREADS, 0, s>>>(ix, qptr, qa, k, ws.cand, ws.flags, ws.result_dev, seq, margin, xcv,
                                                 reinterpret_cast<const unsigned long long*>(ws.bmax), grid);
    return hipGetLastError();
}

hipError_t lbh_launch_search_exact(const lbh_dev_index& ix, const lbh_workspace& ws, const double* qptr,
                                   const lbh_qarg& qa, int k, int has_since,
                                   int64_t since_ms, uint64_t tag_mask, const lbh_cursor& cur, int grid, int seq,
                                   hipStream_t s, hipEvent_t ev0, hipEvent_t ev1, const lbh_xchg* xc, const int32_t* gate) {
    hipError_t e;
    if (ev0) (void)hipEventRecord(ev0, s);
    const dim3 g(grid), b(LBH_SCAN_THREADS);
#define LBH_XSCAN_ARGS ix, qptr, qa, k, since_ms, tag_mask, cur, ws.xcand_ord, ws.xcand_row, ws.flags, gate
    if (has_since && tag_mask) scan_exact_kernel<true, true><<<g, b, 0, s>>>(LBH_XSCAN_ARGS);
    else if (has_since) scan_exact_kernel<true, false><<<g, b, 0, s>>>(LBH_XSCAN_ARGS);
    else if (tag_mask) scan_exact_kernel<false, true><<<g, b, 0, s>>>(LBH_XSCAN_ARGS);
    else scan_exact_kernel<false, false><<<g, b, 0, s>>>(LBH_XSCAN_ARGS);
#undef LBH_XSCAN_ARGS
    if (ev1) (void)hipEventRecord(ev1, s);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    select_exact_kernel<<<1, SEL_THREADS, 0, s>>>(ix, k, ws.xcand_ord, ws.xcand_row, ws.flags, ws.result_dev, seq, gate,
                                                  xc ? *xc : lbh_xchg{nullptr, 0, 0});
    return hipGetLastError();
}

namespace {
// gathered lists: device memory -> device-mapped host memory, then the completion ticket (system-scope release)
__global__ void xchg_publish_kernel(const int64_t* __restrict__ recv, int64_t* __restrict__ host, int n_words, int32_t* ticket, int seq) {
    for (int i = threadIdx.x; i < n_words; i += blockDim.x) host[i] = recv[i];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(ticket, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
}  // namespace

hipError_t lbh_launch_xchg_publish(const int64_t* recv_dev, int64_t* recv_host_dev, int n_words, int32_t* ticket_host_dev, int seq,
                                   hipStream_t s) {
    xchg_publish_kernel<<<1, 256, 0, s>>>(recv_dev, recv_host_dev, n_words, ticket_host_dev, seq);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Allocation-free search (lbh_search_dev_noalloc): the same scan / select kernels, with every scratch
// buffer they need -- the flat candidate list, the exact path's candidate arrays, the counters and the
// result block -- carved out of ONE file-scope device array instead of the index's hipMalloc'd workspace.
// Each GPU the library runs on holds its own copy of the array until the process ends; the host only ever
// uses the per-device address that hipGetSymbolAddress returned (lbh_noalloc_init), never the symbol's name.
// ------------------------------------------------------------------------------------------------
namespace {
constexpr int NA_FAST_BLOCKS = 512;    // scan grid cap of the allocation-free path
constexpr int NA_EXACT_BLOCKS = 128;
constexpr size_t NA_OFF_CAND = 0;
constexpr size_t NA_OFF_XORD = NA_OFF_CAND + (size_t)NA_FAST_BLOCKS * LBH_LIST_CAP * 8;                        // 1 MiB
constexpr size_t NA_OFF_XROW = NA_OFF_XORD + (size_t)NA_EXACT_BLOCKS * LBH_SCAN_WAVES * LBH_MAX_PAGE * 8;      // + 256 KiB
constexpr size_t NA_OFF_FLAGS = NA_OFF_XROW + (size_t)NA_EXACT_BLOCKS * LBH_SCAN_WAVES * LBH_MAX_PAGE * 4;     // + 128 KiB
constexpr size_t NA_OFF_RESULT = NA_OFF_FLAGS + 512;
constexpr size_t NA_BYTES = NA_OFF_RESULT + ((sizeof(lbh_result) + 255) / 256) * 256;
#ifndef LBH_EXPERIMENT
static_assert(NA_BYTES == LBH_NOALLOC_SCRATCH_BYTES, "include/lbhip.h states the size of the scratch array");
#endif

static __device__ __align__(256) unsigned char scratch[NA_BYTES];

// device result block -> the caller's device buffers (rows as int64, fp64 scores, count; -1 = duplicate keys)
__global__ void export_result_kernel(const lbh_result* __restrict__ res, int64_t* __restrict__ out_rows,
                                     double* __restrict__ out_scores, int64_t* __restrict__ out_count) {
    const int t = threadIdx.x;
    const int cnt = res->overflow ? 0 : res->count;
    if (t < cnt) {
        out_rows[t] = (int64_t)res->rows[t];
        out_scores[t] = res->scores[t];
    }
    if (t == 0) *out_count = res->overflow ? -1 : cnt;
}
}  // namespace

hipError_t lbh_noalloc_init(void** out_ptr) {
    // the symbol's name is a host-side stand-in; the array's address on the CURRENT device comes from the runtime
    return hipGetSymbolAddress(out_ptr, HIP_SYMBOL(scratch));
}

hipError_t lbh_launch_search_noalloc(void* scratch_ptr, const lbh_dev_index& ix, const lbh_qarg& qa, int k, int has_since,
                                     int64_t since_ms, uint64_t tag_mask, int n_cu, int nt, float margin,
                                     int64_t* out_rows_dev, double* out_scores_dev, int64_t* out_count_dev, hipStream_t s) {
    unsigned char* base = static_cast<unsigned char*>(scratch_ptr);
    uint64_t* cand = reinterpret_cast<uint64_t*>(base + NA_OFF_CAND);
    uint64_t* xord = reinterpret_cast<uint64_t*>(base + NA_OFF_XORD);
    uint32_t* xrow = reinterpret_cast<uint32_t*>(base + NA_OFF_XROW);
    int32_t* flags = reinterpret_cast<int32_t*>(base + NA_OFF_FLAGS);
    lbh_result* res = reinterpret_cast<lbh_result*>(base + NA_OFF_RESULT);
    // the array keeps the last call's contents: counters and the result header must start at zero
    hipError_t e = hipMemsetAsync(flags, 0, 512 + 64, s);
    if (e != hipSuccess) return e;
    int grid = lbh_scan_grid(ix.n, n_cu, 0);
    if (grid > NA_FAST_BLOCKS) grid = NA_FAST_BLOCKS;
    const dim3 g(grid), b(LBH_SCAN_THREADS);
    const lbh_split sp = lbh_scan_split(ix.n, grid, n_cu);
#define LBH_NA_ARGS ix, nullptr, nullptr, qa, k, margin, since_ms, tag_mask, cand, flags, sp, res, 1, lbh_xchg{nullptr, 0, 0}, nullptr, 0
    // not fused: the sequence is stream-ordered with no host step in between, there is nothing to gain
    if (nt) {
        if (has_since && tag_mask) scan_f32_kernel<true, true, true, false, false><<<g, b, 0, s>>>(LBH_NA_ARGS);
        else if (has_since) scan_f32_kernel<true, false, true, false, false><<<g, b, 0, s>>>(LBH_NA_ARGS);
        else if (tag_mask) scan_f32_kernel<false, true, true, false, false><<<g, b, 0, s>>>(LBH_NA_ARGS);
        else scan_f32_kernel<false, false, true, false, false><<<g, b, 0, s>>>(LBH_NA_ARGS);
    } else {
        if (has_since && tag_mask) scan_f32_kernel<true, true, false, false, false><<<g, b, 0, s>>>(LBH_NA_ARGS);
        else if (has_since) scan_f32_kernel<true, false, false, false, false><<<g, b, 0, s>>>(LBH_NA_ARGS);
        else if (tag_mask) scan_f32_kernel<false, true, false, false, false><<<g, b, 0, s>>>(LBH_NA_ARGS);
        else scan_f32_kernel<false, false, false, false, false><<<g, b, 0, s>>>(LBH_NA_ARGS);
    }
#undef LBH_NA_ARGS
    select_fast_kernel<<<1, SEL_THREADS, 0, s>>>(ix, nullptr, qa, k, cand, flags, res, 1, margin, lbh_xchg{nullptr, 0, 0}, nullptr, 0);
    // mass near-ties (res->overflow == 1): the exact fp64 path decides; both kernels return at once otherwise.  No host
    // round trip in between, so the whole sequence is stream-ordered.
    lbh_cursor cur;
    cur.active = 0; cur.ord = 0; cur.key[0] = cur.key[1] = cur.key[2] = cur.key[3] = 0;
    const dim3 gx(grid < NA_EXACT_BLOCKS ? grid : NA_EXACT_BLOCKS);
#define LBH_NAX_ARGS ix, nullptr, qa, k, since_ms, tag_mask, cur, xord, xrow, flags, &res->overflow
    if (has_since && tag_mask) scan_exact_kernel<true, true><<<gx, b, 0, s>>>(LBH_NAX_ARGS);
    else if (has_since) scan_exact_kernel<true, false><<<gx, b, 0, s>>>(LBH_NAX_ARGS);
    else if (tag_mask) scan_exact_kernel<false, true><<<gx, b, 0, s>>>(LBH_NAX_ARGS);
    else scan_exact_kernel<false, false><<<gx, b, 0, s>>>(LBH_NAX_ARGS);
#undef LBH_NAX_ARGS
    select_exact_kernel<<<1, SEL_THREADS, 0, s>>>(ix, k, xord, xrow, flags, res, 2, &res->overflow, lbh_xchg{nullptr, 0, 0});
    export_result_kernel<<<1, LBH_MAX_PAGE, 0, s>>>(res, out_rows_dev, out_scores_dev, out_count_dev);
    return hipGetLastError();
}

// K3, fp16: the pass schedule described above scan_batched_tiled_kernel.
template <bool I8>
static hipError_t launch_batched_tiled(const lbh_dev_index& ix, const void* emb_tiled, const lbh_workspace& ws, const double* q_src, int nq,
                                       int k, int has_since, int64_t since_ms, uint64_t tag_mask, int n_cu, float margin, int seq,
                                       hipStream_t s, hipEvent_t ev0, hipEvent_t ev1, float emb_scale) {
    hipError_t e;
    const f32x4* mat = reinterpret_cast<const f32x4*>(emb_tiled);
    const f32x4* bfrag = reinterpret_cast<const f32x4*>(ws.bq_bfrag);
    // all 8 groups (unused ones as zeros); also clears the per-query counters and the flag words
    prep_bfrag_kernel<I8><<<8, 256, 0, s>>>(q_src, nq, reinterpret_cast<f16x8*>(ws.bq_bfrag), ws.bq_total, ws.flags,
                                            q_src != ws.bq_qmat ? ws.bq_qmat : nullptr, ws.bq_qscale, emb_scale);
    const int64_t T = (ix.n + 31) / 32;
    auto step_of = [&](int64_t n_sel) -> uint64_t {
        // selected tile i = (i * T) / n_sel in 32.32 fixed point, rounded down so that it never passes T - 1
        return n_sel >= T ? (1ull << 32) : (uint64_t)(((unsigned __int128)(uint64_t)T << 32) / (uint64_t)n_sel);
    };
    auto launch = [&](int64_t n_sel, const uint32_t* seed) {
        int64_t blocks = (n_sel + K3T_WAVES - 1) / K3T_WAVES;
        if (blocks > K3T_BLOCKS_PER_CU * (int64_t)n_cu) blocks = K3T_BLOCKS_PER_CU * (int64_t)n_cu;
        scan_batched_tiled_kernel<I8><<<dim3((unsigned)blocks), K3T_THREADS, 0, s>>>(ix, mat, bfrag, nq, has_since, since_ms, tag_mask, T, n_sel,
                                                                                    step_of(n_sel), seed, ws.bq_cand, ws.bq_total, ws.bq_qscale);
    };
    if (ev0) (void)hipEventRecord(ev0, s);
    const uint32_t* seed = nullptr;
    if (ix.n > BQ_PERQ) {
        // SAMPLE pass over 1/div of the tiles (at least 4096 tiles, or all of them), then thresholds from the group maxima.
        // About k * div rows per query then reach their threshold in the full pass; the harvest path costs ~0.2 us of a
        // wave's time per hit, the sample pass its share of a full pass, which puts the optimum near div = 14 at top-8 (below).
        static const int div = [] {
            const char* ev = getenv("LBH_BQ_SAMPLE_DIV");
            const int v = ev ? atoi(ev) : 0;
            return v >= 1 && v <= 4096 ? v : 0;
        }();
        // fp16: 14.  int8: the margin is ~15x wider, so a sample's k-th best lets more rows through; a denser sample pays
        static const int div_i8 = [] {
            const char* ev = getenv("LBH_BQ_SAMPLE_DIV_I8");
            const int v = ev ? atoi(ev) : 0;
            return v >= 1 && v <= 4096 ? v : 6;
        }();
        // the harvest path's cost grows with k x div, the sample pass's with 1 / div: the optimum moves as 1 / sqrt(k)
        // (10 M x 256: k = 64, div 14 -> 5: 0.64 -> 0.60 ms; k = 200, 14 -> 3: 1.36 -> 1.05 ms)
        int div_k = (int)(40.0 / __builtin_sqrt((double)(k < 1 ? 1 : k)) + 0.5);
        div_k = div_k < 3 ? 3 : (div_k > 14 ? 14 : div_k);
        const int div_eff = div ? div : (I8 ? div_i8 : div_k);
        int64_t n_sel = T / div_eff;
        if (n_sel < 4096) n_sel = T < 4096 ? T : 4096;
        int64_t blocks = (n_sel + K3M_WAVES - 1) / K3M_WAVES;
        if (blocks > K3M_BLOCKS_PER_CU * (int64_t)n_cu) blocks = K3M_BLOCKS_PER_CU * (int64_t)n_cu;
        if (blocks > BQ_PERQ) blocks = BQ_PERQ;  // one list slot per workgroup
        scan_batched_max_kernel<I8><<<dim3((unsigned)blocks), K3M_THREADS, 0, s>>>(ix, mat, bfrag, nq, has_since, since_ms, tag_mask, T, n_sel,
                                                                                  step_of(n_sel), ws.bq_cand, ws.bq_qscale);
        seed_batched_kernel<<<nq, SEL_THREADS, 0, s>>>(k, margin, ws.bq_cand, ws.bq_total, ws.bq_seed, (int)blocks);
        seed = ws.bq_seed;
    }
    launch(T, seed);  // seed == nullptr (a corpus that fits the lists): every live row of every query is listed
    if (ev1) (void)hipEventRecord(ev1, s);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    select_batched_kernel<<<nq, SEL_THREADS, 0, s>>>(ix, ws.bq_qmat, k, margin, ws.bq_cand, ws.bq_total, ws.flags, ws.bq_result_dev, seq, 0);
    return hipGetLastError();
}

// a further result page of the batched search just run on `ws` (top_k > LBH_MAX_PAGE): the select step again, emitting
// ranks [out_off, out_off + LBH_MAX_PAGE) of every query into its result block
hipError_t lbh_launch_select_batched_page(const lbh_dev_index& ix, const lbh_workspace& ws, int nq, int k, float margin, int seq,
                                          int out_off, hipStream_t s) {
    select_batched_kernel<<<nq, SEL_THREADS, 0, s>>>(ix, ws.bq_qmat, k, margin, ws.bq_cand, ws.bq_total, ws.flags, ws.bq_result_dev, seq, out_off);
    return hipGetLastError();
}

hipError_t lbh_launch_search_batched(const lbh_dev_index& ix, const uint16_t* emb16, const lbh_workspace& ws, const double* q_src, int nq,
                                     int k, int has_since, int64_t since_ms, uint64_t tag_mask, int n_cu, int bf16, float norm_scale,
                                     int seq, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1, const int8_t* emb8, float emb8_scale,
                                     float margin_i8) {
    // one-time set-up (thread-safe: searches on different indexes may arrive concurrently)
    static std::once_flag once;
    static hipError_t once_err = hipSuccess;
    static int shifts[4] = {8, 4, 0, 0};
    static int n_shifts = 2;
    std::call_once(once, [] {
        const void* fns[] = {(const void*)scan_batched_kernel<true, true, false>, (const void*)scan_batched_kernel<true, false, false>,
                             (const void*)scan_batched_kernel<false, true, false>, (const void*)scan_batched_kernel<false, false, false>,
                             (const void*)scan_batched_kernel<true, true, true>, (const void*)scan_batched_kernel<true, false, true>,
                             (const void*)scan_batched_kernel<false, true, true>, (const void*)scan_batched_kernel<false, false, true>};
        for (const void* f : fns) {
            const hipError_t e1 = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)BQ_LDS);
            if (e1 != hipSuccess) once_err = e1;
        }
        if (const char* ev = getenv("LBH_BQ_SAMPLE_SHIFTS")) {  // e.g. "8,4" (default), "6", "" = none
            n_shifts = 0;
            for (const char* p = ev; *p && n_shifts < 4;) {
                char* end;
                const long v = strtol(p, &end, 10);
                if (end == p) break;
                if (v > 0 && v < 32) shifts[n_shifts++] = (int)v;
                p = (*end == ',') ? end + 1 : end;
            }
        }
    });
    if (once_err != hipSuccess) return once_err;
    hipError_t e;
    // the error bounds are relative to |x||q| (1 for the reference's unit vectors); fp16 additionally needs the
    // components inside its range, so anything but unit-scale data takes the split-bf16 kernel instead (the caller
    // passes the fp16 copy only for unit-scale data)
    if (bf16 == 3 && emb8)
        return launch_batched_tiled<true>(ix, emb8, ws, q_src, nq, k, has_since, since_ms, tag_mask, n_cu, margin_i8, seq, s, ev0, ev1, emb8_scale);
    if (bf16 == 3) bf16 = 2;
    if (bf16 == 2 && (!emb16 || norm_scale > 1.001f || norm_scale < 0.25f)) bf16 = 1;
    if (bf16 == 2) return launch_batched_tiled<false>(ix, emb16, ws, q_src, nq, k, has_since, since_ms, tag_mask, n_cu, LBH_MARGIN_F16 * norm_scale, seq, s, ev0, ev1, 0.f);
    const float margin = (bf16 ? LBH_MARGIN_BF16X3 : LBH_MARGIN) * norm_scale;
    e = hipMemsetAsync(ws.flags, 0, 16, s);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(ws.bq_total, 0, BQ_MAXQ * BQ_QT * 4, s);
    if (e != hipSuccess) return e;
    const dim3 b(BQ_THREADS);
    auto launch = [&](int64_t r0, int64_t r1, const uint32_t* seed) {
        const int64_t tiles = (r1 - r0 + 63) / 64;
        const dim3 g((unsigned)(tiles < n_cu ? tiles : n_cu));
#define LBH_BSCAN_ARGS ws.bq_qmat, nq, k, margin, since_ms, tag_mask, r0, r1, seed, ws.bq_cand, ws.bq_total, ws.flags
        if (bf16) {  // split-bf16 products on v_mfma_f32_32x32x16_bf16
            if (has_since && tag_mask) scan_batched_kernel<true, true, true><<<g, b, BQ_LDS, s>>>(ix, LBH_BSCAN_ARGS);
            else if (has_since) scan_batched_kernel<true, false, true><<<g, b, BQ_LDS, s>>>(ix, LBH_BSCAN_ARGS);
            else if (tag_mask) scan_batched_kernel<false, true, true><<<g, b, BQ_LDS, s>>>(ix, LBH_BSCAN_ARGS);
            else scan_batched_kernel<false, false, true><<<g, b, BQ_LDS, s>>>(ix, LBH_BSCAN_ARGS);
        } else {     // exact fp32 products on v_mfma_f32_32x32x2_f32
            if (has_since && tag_mask) scan_batched_kernel<true, true, false><<<g, b, BQ_LDS, s>>>(ix, LBH_BSCAN_ARGS);
            else if (has_since) scan_batched_kernel<true, false, false><<<g, b, BQ_LDS, s>>>(ix, LBH_BSCAN_ARGS);
            else if (tag_mask) scan_batched_kernel<false, true, false><<<g, b, BQ_LDS, s>>>(ix, LBH_BSCAN_ARGS);
            else scan_batched_kernel<false, false, false><<<g, b, BQ_LDS, s>>>(ix, LBH_BSCAN_ARGS);
        }
#undef LBH_BSCAN_ARGS
    };
    if (ev0) (void)hipEventRecord(ev0, s);
    // Sample passes seed every query's threshold before the full pass: with cold thresholds nearly every
    // tile holds candidates and the (VALU + LDS) harvest path, not the matrix pipe, sets the pace.  Each
    // pass scans the first n >> shift rows starting from the previous pass's seeds (a subset's k-th best
    // never exceeds the corpus's, so seeds are always valid lower bounds).  Default: n/256, then n/16
    // (measured at 10 M x 256 with the fp16 scan: "8,4" 0.82 ms, "9,5" 0.86, "6" 0.98, "8,4,2" 0.95).
    const uint32_t* seed = nullptr;
    for (int i = 0; i < n_shifts; ++i) {
        const int64_t ns = ((ix.n >> shifts[i]) / 64) * 64;
        if (ns < 16384) continue;
        launch(0, ns, seed);
        seed_batched_kernel<<<nq, SEL_THREADS, 0, s>>>(k, margin, ws.bq_cand, ws.bq_total, ws.bq_seed, 0);
        seed = ws.bq_seed;
    }
    launch(0, ix.n, seed);
    if (ev1) (void)hipEventRecord(ev1, s);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    select_batched_kernel<<<nq, SEL_THREADS, 0, s>>>(ix, ws.bq_qmat, k, margin, ws.bq_cand, ws.bq_total, ws.flags, ws.bq_result_dev, seq, 0);
    return hipGetLastError();
}

#ifdef LBH_K2_TIMING
extern "C" int lbh_debug_k2_times(long long* out) {  // instrumented build only: LBH_MAX_BLOCKS x 6 stamps
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(lbh_k2_dbg), sizeof(long long) * LBH_MAX_BLOCKS * 6);
}
extern "C" int lbh_debug_k4_times(long long* out) {  // 16 stamps of the select kernel
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(lbh_k4_dbg), sizeof(long long) * 16);
}
#endif
#if LBH_BQ_ABLATE == 10
extern "C" int lbh_debug_k3_times(long long* out) {  // instrumented build only: 3 x 256 x 5 ticks (tools/scripts/k3_phases.py)
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(lbh_k3_dbg), sizeof(long long) * 3 * 256 * 5);
}
#endif
