// lbh_internal.h -- shared declarations of liblbhip.so (not installed; C ABI is include/lbhip.h).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/lbhip.h"

#define LBH_WAVE 64
#ifndef LBH_SCAN_THREADS
#define LBH_SCAN_THREADS 256                    // 4 waves per scan block
#endif
#define LBH_SCAN_WAVES (LBH_SCAN_THREADS / 64)
#define LBH_WCAP 256                            // per-wave candidate buffer (entries) in LDS
#define LBH_LIST_CAP 256                        // per-block candidate list written to HBM
#define LBH_MCAP 2048                           // max rows the single-block final stage sorts exactly
#define LBH_BQ_MAXQ 256                         // queries per batched (MFMA) launch
#define LBH_BQ_PERQ 8192                        // candidate slots per query of a batched launch
#define LBH_BQ_QT_STRIDE 32                     // ints between two queries' list counters (one cache line each)
#define LBH_BQ_KMAX 16                          // largest top_k the batched path serves
#define LBH_MAX_BLOCKS 2048                     // max scan grid (lists in the workspace)
#define LBH_BBEST_SLOTS 1024                    // blocks whose best score the two-kernel scan leaves for the select kernel's cut (bmax[0 .. ) as 64-bit words)
#define LBH_XB_SLOTS 512                        // blocks that publish their best score for the cross-block bound (K2, large k)

// fp32 scan vs reference fp64 score: rigorous bound on |s32 - s64| is < 4e-6 for unit 64-vectors
// (input rounding 2*2^-24 + 64 fp32 roundings, DESIGN.md "Error bound"); candidates are every row
// within 2*eps of the k-th best fp32 score, so the exact top-k is always among them.
#define LBH_MARGIN 1.0e-5f
// Batched scan with split-bf16 products (x = hi + lo, x*q ~ hi*hi + hi*lo + lo*hi): the dropped lo*lo term and
// the split residues are < 3 * 2^-18 |x q| (1.2e-5 summed over a unit vector pair); 192 fp32 accumulations of
// partial sums <= ~1 add at most 192 * 2^-23 = 2.3e-5 even if the matrix pipe truncated instead of rounding.
// eps < 3.5e-5, margin = 1e-4 > 2 eps.  lbh_stats.max_score_err reports the largest error actually seen.
#define LBH_MARGIN_BF16X3 1.0e-4f
// Batched scan with one fp16 product per dimension: eps < 2^-10 + 1e-5 (see scan_batched_f16_kernel).
#define LBH_MARGIN_F16 2.2e-3f

// ---- order-preserving integer images of floats ------------------------------------------------
__host__ __device__ inline uint32_t lbh_ord32(float f) {
    uint32_t b = __builtin_bit_cast(uint32_t, f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__host__ __device__ inline float lbh_unord32(uint32_t o) {
    uint32_t b = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o;
    return __builtin_bit_cast(float, b);
}
__host__ __device__ inline uint64_t lbh_ord64(double d) {
    uint64_t b = __builtin_bit_cast(uint64_t, d);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__host__ __device__ inline double lbh_unord64(uint64_t o) {
    uint64_t b = (o >> 63) ? (o & 0x7FFFFFFFFFFFFFFFull) : ~o;
    return __builtin_bit_cast(double, b);
}

// ---- fp16 copy of the claim matrix (emb16): stored in MFMA-fragment order -------------------------------------
// A tile = 32 consecutive rows = 4 KiB = four 1-KiB chunks, one per k-step s of v_mfma_f32_32x32x16_f16.  Chunk s holds,
// for lane L = 32 h + j, the 8 halves dims 16 s + 8 h .. + 7 of row 32 t + j at byte 16 L: a wave reads a whole A
// fragment with ONE fully coalesced global_load_dwordx4, no LDS staging and no conversion, and the single-query fp16
// scan streams the same chunks (a lane then holds 8 dims of one row per load).  Capacity is padded to whole tiles.
__host__ __device__ inline int64_t lbh_emb16_off(int64_t row, int d) {  // offset in halves of element (row, d)
    return (row >> 5) * 2048 + (int64_t)(d >> 4) * 512 + (int64_t)((((d >> 3) & 1) << 5) + (row & 31)) * 8 + (d & 7);
}

// ---- int8 copy of the claim matrix (emb8): the operand of the int8 batched scan (bq_mode 3) ----------------------------
// x_i ~ scale * X_i with ONE scale for the whole index (|x_i| <= 127 scale for every element of every row) and X_i =
// rint(x_i / scale) in [-127, 127].  Stored in the fragment order of v_mfma_i32_32x32x32_i8: a tile = 32 rows = 2 KiB =
// two 1-KiB chunks, one per k-step s; chunk s holds, for lane L = 32 h + j, the 16 bytes dims 32 s + 16 h .. + 15 of row
// 32 t + j at byte 16 L.  (The queries' B fragments use the same (lane, byte) -> dim map, so whatever order the matrix
// core takes the 16 bytes of a lane in, both operands agree on it.)
__host__ __device__ inline int64_t lbh_emb8_off(int64_t row, int d) {  // offset in bytes of element (row, d)
    return (((row >> 5) * 2 + (int64_t)(d >> 5)) * 64 + (int64_t)((((d >> 4) & 1) << 5) + (row & 31))) * 16 + (d & 15);
}
// Error bound of one int8 score: x_i = s X_i + e_i, |e_i| <= s/2; q_i = t Q_i + f_i, |f_i| <= t/2, so
// |x.q - s t sum X_i Q_i| <= (s/2)(|q|_1 + 32 t) + (t/2)(|x|_1 + 32 s) + 16 s t, with |x|_1 <= 8 |x|_2.
__host__ __device__ inline float lbh_next_up(float f) {  // the next float above a non-negative finite f
    return __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, f) + 1u);
}
#define LBH_I8_MARGIN_MAX 0.09f                 // a looser bound than this (odd data: one huge component) takes the fp16 kernel

// ---- device arrays of one index (all row-major, row r at the natural offset) ------------------
struct lbh_dev_index {
    const float* emb32;       // n x 64 fp32   (256 B rows)  -- the scan copy
    const double* emb64;      // n x 64 fp64   (512 B rows)  -- the master, reference bits
    const int64_t* created;   // n
    const uint64_t* tags;     // n
    const uint64_t* keys;     // n x 4, word 0 most significant (big-endian image of the 32 key bytes)
    const uint64_t* ids;      // n: stable claim id of each row (NULL for caller-owned arrays: results then carry rows only)
    int64_t n;
};

// Optional second destination of a search's final kernel (claim-sharded search): the result as the packed list the ranks
// exchange -- (LBH_MAX_PAGE + 1) x 6 int64 in DEVICE memory, row 0 = {count, ...}, then {fp64 score bits, global claim id
// = id * gid_mul + gid_add, 4 key words} per result -- so that the all-gather reads it where it was produced.
struct lbh_xchg {
    int64_t* dst;             // NULL = not wanted
    uint64_t gid_mul, gid_add;
};

// cursor for paging the exact path: only rows strictly AFTER (score, key) in the reference order pass
struct lbh_cursor {
    int32_t active;
    uint64_t ord;      // lbh_ord64(score)
    uint64_t key[4];
};

// the query vector as a kernel argument (512 B of the 4 KB kernarg segment): no H2D copy per search
struct lbh_qarg {
    double q[64];
};

// How the single-query scan divides the 64-row tiles among its blocks: block b streams the contiguous range
// [start(b), start(b + 1)), start(b) = n_tiles x (weight of blocks below b) / (weight of all blocks).  Four weight classes:
// even / odd blockIdx (the XCD a block lands on alternates with its index, and the odd XCDs are served ~5-10 % slower),
// and first / later residency (block b >= n_first shares its CU with an older block whose waves win arbitration).
// All-zero weights = the plain grid-stride loop (every block the same number of tiles).
struct lbh_split {
    uint32_t w[4];     // even-first, odd-first, even-later, odd-later
    int32_t n_first;   // blocks resident first: min(grid, CUs)
};
lbh_split lbh_scan_split(int64_t n_rows, int grid, int n_cu);
void lbh_set_scan_split(const uint32_t w[4]);  // tuning override (env LBH_SCAN_SPLIT, "a,b,c,d"; all 0 = grid-stride)

// result block written by the final kernels (device) and copied to pinned host memory
struct lbh_result {
    int32_t count;
    int32_t overflow;         // 1: fp32 path could not bound its candidate set -> caller re-runs exactly; 2: exact path, duplicate keys
    int32_t candidates;       // rows that were re-scored in fp64
    int32_t seq;              // completion ticket: written last, after a system-scope fence (see publish_result)
    float max_err;            // largest |scan score - exact fp64 score| among the re-scored rows (0 on the exact path)
    int32_t pad;
    uint32_t rows[LBH_MAX_PAGE];
    double scores[LBH_MAX_PAGE];
    uint64_t keys[LBH_MAX_PAGE][4];   // hash keys of the results (shard merge, paging cursor)
};
#define LBH_XCHG_WORDS ((LBH_MAX_PAGE + 1) * 6)  // int64 words of one rank's packed list

// cursor for paging the recency query: only rows strictly AFTER (a, b) in (created desc, order asc) pass
struct lbh_rcursor {
    int32_t active;
    uint64_t a;        // created_ms with the sign bit flipped (unsigned order = signed order)
    uint64_t b;        // ~order (larger = earlier in insertion order)
};

// result block of the recency query (device-mapped pinned host memory, like lbh_result)
struct lbh_recent_result {
    int32_t count;
    int32_t seq;              // completion ticket, written last
    int64_t matched;          // live rows with created_ms >= since_ms (whole index, cursor ignored)
    uint32_t rows[LBH_MAX_PAGE];
    int64_t created[LBH_MAX_PAGE];
    uint64_t order[LBH_MAX_PAGE];
};

#define LBH_RC_MAX_BLOCKS 512                   // recency scan grid cap (entries per block <= LBH_MAX_PAGE)

struct lbh_workspace {
    double* q64;              // 64: query embedded on the GPU (search_text); vector queries travel as kernargs
    uint64_t* cand;           // flat, LBH_MAX_BLOCKS x LBH_LIST_CAP entries (ord32<<32 | row)   fast path
    uint64_t* xcand_ord;      // flat, LBH_MAX_BLOCKS*4 x LBH_MAX_PAGE entries: ord64(score)     exact path
    uint32_t* xcand_row;      // same shape: row
    int32_t* flags;           // 128 ints: [0] overflow, [1] entries in cand, [2] entries in xcand, [6..7] 64-bit ticket word of a fused scan (blocks finished; bit 63: unbounded candidate set), [64..127] the scan blocks' shared cuts (zeroed by the select step)
    uint32_t* bmax;           // LBH_MAX_BLOCKS: entries each scan block left in its region of `cand` (fused select: the last block reads them);
                              // then LBH_XB_SLOTS 64-bit words: launch << 32 | best ordered score of a block (cross-block bound)
    // batched (K3) workspace, allocated on first use
    double* bq_qmat;          // 256 x 64 fp64 query matrix
    uint64_t* bq_cand;        // 256 x LBH_BQ_PERQ flat candidate lists, one per query
    int32_t* bq_total;        // 256 counters, LBH_BQ_QT_STRIDE ints apart
    uint32_t* bq_seed;        // 256 per-query seed thresholds (ord32) from the sample pass
    void* bq_bfrag;           // 32 KiB: the queries as fp16 (or int8: 16 KiB) MFMA B fragments (scan_batched_tiled_kernel)
    float* bq_qscale;         // 256: int8 mode, score of one integer unit of query q (emb8 scale x the query's own scale)
    lbh_result* bq_result_host;  // 256 results, pinned + device-mapped
    double* bq_qhost;            // 256 x 64 fp64 queries, pinned + device-mapped (read by the fp16 mode's first kernel)
    double* bq_qhost_dev;
    lbh_result* bq_result_dev;
    // recency query (get_recent_claims) workspace, allocated on first use
    uint64_t* rc_a;           // LBH_RC_MAX_BLOCKS x LBH_MAX_PAGE candidate keys: created (biased)
    uint64_t* rc_b;           //                                                  ~order
    uint32_t* rc_row;
    unsigned long long* rc_cnt;  // [0] entries in rc_*, [1] matched rows, [4..6] threshold seed (set, a, b)
    lbh_recent_result* rc_result_host;
    lbh_recent_result* rc_result_dev;
    lbh_result* result_host;  // pinned, device-mapped host memory: the final kernel writes it directly
    lbh_result* result_dev;   // device-side address of result_host
};

// ---- kernel launchers (defined in the .hip files) ---------------------------------------------
// embed.hip
// dim: LBH_DIM for the index; 1..63 = embed(text, dim) of lb/latent.py:41 (components >= dim are written as 0)
// K1's token memo (embed.hip): a table of 128-byte entries in HBM that maps a lower-cased token to its 64 digest bytes.
// `epoch` numbers the launches that use the table (strictly increasing, < 2^24): a launch trusts only entries whose tag
// carries a smaller epoch.  Launches that use it must not overlap (they run on the context's stream, one after another).
struct lbh_memo {
    uint64_t* table;   // (mask + 1) entries x 16 words; NULL = no memo
    uint32_t mask;     // entries - 1 (a power of two, >= 4)
    uint32_t epoch;
};
hipError_t lbh_launch_embed(const uint8_t* utf8_dev, const int64_t* offsets_dev, int64_t n, double* out64_dev,
                            float* out32_dev, hipStream_t s, int dim = LBH_DIM, const lbh_memo* memo = nullptr);
hipError_t lbh_launch_cosine(const double* a_dev, const double* b_dev, int64_t n, double* out_dev, hipStream_t s);
hipError_t lbh_launch_f64_to_f32(const double* in_dev, float* out_dev, int64_t n_elems, hipStream_t s);
hipError_t lbh_launch_move_row(float* emb32, double* emb64, uint16_t* emb16, int8_t* emb8, int64_t* created, uint64_t* tags, uint64_t* keys,
                               uint64_t* order, uint64_t* ids, int64_t dst, int64_t src, hipStream_t s);
hipError_t lbh_launch_move_rows(float* emb32, double* emb64, uint16_t* emb16, int8_t* emb8, int64_t* created, uint64_t* tags, uint64_t* keys,
                                uint64_t* order, uint64_t* ids, const int64_t* pairs_dev, int64_t n_pairs, hipStream_t s);
hipError_t lbh_launch_small_meta(const int64_t* created_src, const uint64_t* tags_src, const uint64_t* keys_src, int64_t* created,
                                 uint64_t* tags, uint64_t* keys, uint64_t* order, uint64_t* ids, int64_t r0, uint64_t id0, int n,
                                 hipStream_t s);
// int8 copy: largest |element| of rows [row0, row1) as float bits, raised into *out_bits_dev by atomicMax; then the conversion
hipError_t lbh_launch_absmax_f64(const double* emb64_dev, int64_t row0, int64_t row1, uint32_t* out_bits_dev, hipStream_t s);
hipError_t lbh_launch_f64_to_i8_tiled(const double* emb64_dev, int8_t* emb8_dev, int64_t row0, int64_t row1, double inv_scale, hipStream_t s);
hipError_t lbh_launch_gather_rows_f64(const double* emb64, const int64_t* rows_dev, int64_t n, double* out_dev, hipStream_t s);
hipError_t lbh_launch_f64_to_f16_tiled(const double* emb64_dev, uint16_t* emb16_dev, int64_t row0, int64_t row1, hipStream_t s);
hipError_t lbh_launch_iota_u64(uint64_t* dst, uint64_t first, int64_t n, hipStream_t s);
hipError_t lbh_launch_bswap_u64(uint64_t* p, int64_t n, hipStream_t s);
hipError_t lbh_launch_scatter_u64(uint64_t* dst, const int64_t* rows_dev, const uint64_t* vals_dev, int64_t n, hipStream_t s);
// search.hip
int lbh_scan_grid(int64_t n_rows, int n_cu, int blocks_per_cu, bool half = false);  // half: the unfiltered fp16 scan
// emb16: NULL = scan the fp32 matrix; else the fp16 copy (opt-in scan_mode 1; `margin` must then be the fp16 one)
hipError_t lbh_launch_search_fast(const lbh_dev_index& ix, const uint16_t* emb16, const lbh_workspace& ws, const double* qptr,
                                  const lbh_qarg& qa, int k, int has_since,
                                  int64_t since_ms, uint64_t tag_mask, int grid, int n_cu, int nt, int seq, float margin,
                                  hipStream_t s, hipEvent_t ev0, hipEvent_t ev1, const lbh_xchg* xc = nullptr, int fuse = 1, int xb_k = 0);
hipError_t lbh_launch_search_exact(const lbh_dev_index& ix, const lbh_workspace& ws, const double* qptr,
                                   const lbh_qarg& qa, int k, int has_since,
                                   int64_t since_ms, uint64_t tag_mask, const lbh_cursor& cur, int grid, int seq,
                                   hipStream_t s, hipEvent_t ev0, hipEvent_t ev1, const lbh_xchg* xc = nullptr,
                                   const int32_t* gate = nullptr);
// copies the gathered lists (n_words int64) from device memory into device-mapped host memory and publishes `seq`
hipError_t lbh_launch_xchg_publish(const int64_t* recv_dev, int64_t* recv_host_dev, int n_words, int32_t* ticket_host_dev, int seq,
                                   hipStream_t s);
hipError_t lbh_launch_recent(const int64_t* created, const uint64_t* order, int64_t n, const lbh_workspace& ws,
                             int page, int has_since, int64_t since_ms, const lbh_rcursor& cur, int64_t anchor_row,
                             int n_cu, int seq, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1);
// emb16: NULL, or the fp16 copy of the matrix as operand source of the fp16 mode (bf16 == 2)
// q_src: where the kernels read the fp64 queries from -- ws.bq_qmat (already copied), or device-mapped host memory (fp16 mode
// only: its first kernel then leaves the device copy in ws.bq_qmat); seq: completion ticket written into every query's result
// bf16 == 3: the int8 mode -- emb8 / emb8_scale: the int8 copy and its scale, margin_i8: the candidate margin the caller
// derived from the error bound for THESE queries (absolute, not multiplied by norm_scale)
hipError_t lbh_launch_search_batched(const lbh_dev_index& ix, const uint16_t* emb16, const lbh_workspace& ws, const double* q_src, int nq,
                                     int k, int has_since, int64_t since_ms, uint64_t tag_mask, int n_cu, int bf16, float norm_scale,
                                     int seq, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1, const int8_t* emb8 = nullptr,
                                     float emb8_scale = 0.f, float margin_i8 = 0.f);

hipError_t lbh_launch_select_batched_page(const lbh_dev_index& ix, const lbh_workspace& ws, int nq, int k, float margin, int seq,
                                          int out_off, hipStream_t s);
#define LBH_BQ_PAGED_KMAX 256                   // largest top_k the tiled batched kernels serve (in result pages of LBH_MAX_PAGE)

// allocation-free search: per-device address of the file-scope scratch array, and the launch sequence on it
hipError_t lbh_noalloc_init(void** out_ptr);
hipError_t lbh_launch_search_noalloc(void* scratch_ptr, const lbh_dev_index& ix, const lbh_qarg& qa, int k, int has_since,
                                     int64_t since_ms, uint64_t tag_mask, int n_cu, int nt, float margin,
                                     int64_t* out_rows_dev, double* out_scores_dev, int64_t* out_count_dev, hipStream_t s);
