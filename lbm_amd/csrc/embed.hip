// embed.hip -- K1: batched claim vectorisation (tokenise + 2x SHA-256 per token + in-order fp64
// accumulation + L2 normalisation), bit-identical to lb/latent.py:34-57 of the reference.
//
// One wave64 per group of EMB_GROUP consecutive claims (see embed_kernel).  Lanes play three roles in turn:
//   1. byte classifier: 64 text bytes per step, __ballot -> token start / end bit masks
//      ([A-Za-z0-9_]+, lb/latent.py:34; every byte >= 0x80 is a separator);
//   2. hasher: lane l hashes token (l >> 1), variant (l & 1): sha256(t) or sha256(t + "\0")
//      (lb/latent.py:47-48), 32 tokens = 64 digests per batch, lower-casing on the fly (:38);
//   3. dimension owner: lane d accumulates v[d] += LUT[digest byte d] strictly in token order
//      (:50-52) into the accumulator of the token's claim, so the fp64 rounding sequence is the reference's.
// The norm is a sequential sum over d = 0..63 (mul, then add, no FMA: this file is built with
// -ffp-contract=off), IEEE sqrt and IEEE divide (:54-57).  Rows are written as one coalesced
// 512-B (fp64 master) and one 256-B (fp32 scan copy) store per claim.
#include "lbh_internal.h"

namespace {

__constant__ uint32_t K256[64] = {
    0x428a2f98u, 0x71374491u, 0xb5c0fbcfu, 0xe9b5dba5u, 0x3956c25bu, 0x59f111f1u, 0x923f82a4u, 0xab1c5ed5u,
    0xd807aa98u, 0x12835b01u, 0x243185beu, 0x550c7dc3u, 0x72be5d74u, 0x80deb1feu, 0x9bdc06a7u, 0xc19bf174u,
    0xe49b69c1u, 0xefbe4786u, 0x0fc19dc6u, 0x240ca1ccu, 0x2de92c6fu, 0x4a7484aau, 0x5cb0a9dcu, 0x76f988dau,
    0x983e5152u, 0xa831c66du, 0xb00327c8u, 0xbf597fc7u, 0xc6e00bf3u, 0xd5a79147u, 0x06ca6351u, 0x14292967u,
    0x27b70a85u, 0x2e1b2138u, 0x4d2c6dfcu, 0x53380d13u, 0x650a7354u, 0x766a0abbu, 0x81c2c92eu, 0x92722c85u,
    0xa2bfe8a1u, 0xa81a664bu, 0xc24b8b70u, 0xc76c51a3u, 0xd192e819u, 0xd6990624u, 0xf40e3585u, 0x106aa070u,
    0x19a4c116u, 0x1e376c08u, 0x2748774cu, 0x34b0bcb5u, 0x391c0cb3u, 0x4ed8aa4au, 0x5b9cca4fu, 0x682e6ff3u,
    0x748f82eeu, 0x78a5636fu, 0x84c87814u, 0x8cc70208u, 0x90befffau, 0xa4506cebu, 0xbef9a3f7u, 0xc67178f2u};

constexpr int EMB_WAVES = 4;       // waves (claims in flight) per block
constexpr int TOK_RING = 128;      // token (start, end) ring per wave
constexpr int BATCH = 32;          // tokens hashed per batch (2 digests each = 64 lanes)

__device__ __forceinline__ uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }

__device__ __forceinline__ bool is_word(uint32_t c) {
    return (c - 'a' < 26u) || (c - 'A' < 26u) || (c - '0' < 10u) || c == '_';
}

// byte `p` of the padded message of a token: lower(tok[p]) for p < len, 0x00 for the optional NUL
// at p == len, 0x80 at p == total, 0 elsewhere (the 64-bit length is patched in by the caller).
__device__ __forceinline__ uint32_t msg_byte(const uint8_t* tok, int len, int total, int p) {
    if (p < len) {
        uint32_t c = tok[p];
        if (c - 'A' < 26u) c += 32;
        return c;
    }
    return p == total ? 0x80u : 0u;
}

__device__ __forceinline__ void sha256_compress(uint32_t st[8], uint32_t w[16]) {
    uint32_t a = st[0], b = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6], h = st[7];
#pragma unroll
    for (int i = 0; i < 64; ++i) {
        uint32_t wi;
        if (i < 16) {
            wi = w[i];
        } else {
            uint32_t w15 = w[(i - 15) & 15], w2 = w[(i - 2) & 15];
            uint32_t s0 = rotr(w15, 7) ^ rotr(w15, 18) ^ (w15 >> 3);
            uint32_t s1 = rotr(w2, 17) ^ rotr(w2, 19) ^ (w2 >> 10);
            wi = w[i & 15] + s0 + w[(i - 7) & 15] + s1;
            w[i & 15] = wi;
        }
        uint32_t S1 = rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25);
        uint32_t ch = (e & f) ^ (~e & g);
        uint32_t t1 = h + S1 + ch + K256[i] + wi;
        uint32_t S0 = rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22);
        uint32_t mj = (a & b) ^ (a & c) ^ (b & c);
        uint32_t t2 = S0 + mj;
        h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    st[0] += a; st[1] += b; st[2] += c; st[3] += d; st[4] += e; st[5] += f; st[6] += g; st[7] += h;
}

__device__ __forceinline__ double readlane_f64(double v, int l) {
    uint64_t b = __builtin_bit_cast(uint64_t, v);
    uint32_t lo = __builtin_amdgcn_readlane((int)(uint32_t)b, l);
    uint32_t hi = __builtin_amdgcn_readlane((int)(uint32_t)(b >> 32), l);
    return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
}

#ifdef LBH_K1_TIMING  // instrumented build (make ablate-style, tools/scripts/k1_phases.py): 10 ns ticks per phase, summed over all waves
__device__ unsigned long long lbh_k1_dbg[8];
#define K1_T(var) const long long var = wall_clock64()
#define K1_ADD(slot, a, b) k1_t[slot] += (unsigned long long)((b) - (a))  // per wave; one atomic per slot at the end
#define k1_count(g) k1_t[4] += (unsigned long long)(g)
#else
#define K1_T(var)
#define K1_ADD(slot, a, b)
#define k1_count(g)
#endif
// One wave per GROUP of consecutive claims.  The hashing stage works on 32 tokens at a time (lane = token x variant)
// and is where the time goes (8 of a claim's 11 us per wave with one claim per wave: tools/scripts/k1_phases.py); a
// synthetic claim has 16 tokens on average, so half of those lanes idled.  The wave therefore streams the tokens of
// EMB_GROUP consecutive claims through ONE token ring and hashes them in full batches, whatever claim they belong to
// (0.62 batches per claim instead of 1); each token's digest is then added to ITS claim's accumulator, in token order
// (lb/latent.py:50-52 is order-sensitive per claim; claims do not interact).  Token positions are relative to their
// own claim's text.
constexpr int EMB_GROUP = 4;
#ifndef LBH_EMB_HEAD
#define LBH_EMB_HEAD 256
#endif
constexpr int EMB_HEAD = LBH_EMB_HEAD;  // bytes of every claim fetched up front (one round of independent loads per group); tokens inside it
                                        // take the word-wise message assembly and the memo (128 covered 44 % of the bench corpus's hash batches)

__global__ __launch_bounds__(EMB_WAVES * 64) void embed_kernel(const uint8_t* __restrict__ utf8,
                                                                const int64_t* __restrict__ offsets, int64_t n,
                                                                double* __restrict__ out64,
                                                                float* __restrict__ out32, int dim, lbh_memo memo) {
    __shared__ double lut[256];
    __shared__ int tok_start[EMB_WAVES][TOK_RING];
    __shared__ int tok_end[EMB_WAVES][TOK_RING];
    __shared__ int tok_claim[EMB_WAVES][TOK_RING];            // which claim of the group (0..EMB_GROUP-1) a token belongs to
    __shared__ int64_t claim_off[EMB_WAVES][EMB_GROUP + 1];   // text offsets of the group's claims (and the end of the last)
    __shared__ __attribute__((aligned(8))) uint8_t head[EMB_WAVES][EMB_GROUP][EMB_HEAD + 8];  // first bytes of each claim: the classifier's 64-byte steps
                                                              // would otherwise be a chain of dependent HBM round trips
    __shared__ uint32_t dig[EMB_WAVES][BATCH][16];            // 64 digest bytes per token, digest byte order

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    // lb/latent.py:52  (b / 127.5) - 1.0, evaluated once per block with IEEE fp64 divide
    lut[threadIdx.x & 255] = ((double)(threadIdx.x & 255) / 127.5) - 1.0;
    __syncthreads();
#ifdef LBH_K1_TIMING
    unsigned long long k1_t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    int* ts = tok_start[wave];
    int* te = tok_end[wave];
    int* tc = tok_claim[wave];
    const uint8_t* digb = reinterpret_cast<const uint8_t*>(&dig[wave][0][0]);

    for (int64_t g0 = ((int64_t)blockIdx.x * EMB_WAVES + wave) * EMB_GROUP; g0 < n;
         g0 += (int64_t)gridDim.x * EMB_WAVES * EMB_GROUP) {
        K1_T(c0);
        const int gn = (int)(n - g0 < EMB_GROUP ? n - g0 : EMB_GROUP);
        int64_t* cb = claim_off[wave];
        if (lane <= EMB_GROUP) cb[lane] = offsets[g0 + (lane <= gn ? lane : gn)];
        __builtin_amdgcn_wave_barrier();
        {
            uint32_t hb[EMB_GROUP][EMB_HEAD / 64];
#pragma unroll
            for (int c = 0; c < EMB_GROUP; ++c) {
                const int64_t o = cb[c < gn ? c : gn], e = cb[c < gn ? c + 1 : gn];
#pragma unroll
                for (int u = 0; u < EMB_HEAD / 64; ++u) hb[c][u] = (o + 64 * u + lane < e) ? utf8[o + 64 * u + lane] : 0u;
            }
#pragma unroll
            for (int c = 0; c < EMB_GROUP; ++c) {
#pragma unroll
                for (int u = 0; u < EMB_HEAD / 64; ++u) head[wave][c][64 * u + lane] = (uint8_t)hb[c][u];
            }
            __builtin_amdgcn_wave_barrier();
        }
        uint32_t has_tok = 0;    // bit c: claim c of the group has at least one token (an empty claim embeds to zeros)
        double acc[EMB_GROUP];   // lane = dimension
#pragma unroll
        for (int c = 0; c < EMB_GROUP; ++c) acc[c] = 0.0;
        int n_started = 0, n_ended = 0, n_done = 0;

        auto hash_batch = [&](int nb) {
            K1_T(h0);
            const int t = lane >> 1, v = lane & 1;
            const bool act = t < nb;
            const int slot = (n_done + t) & (TOK_RING - 1);
            const int s = act ? ts[slot] : 0;
            const int tl = act ? te[slot] - s : 0;
            const int ci = act ? tc[slot] : 0;
            const int64_t base = cb[ci];
            const int total = tl + v;
            const int nblk = act ? ((total + 9 + 63) >> 6) : 0;
            const uint8_t* tok = utf8 + base + s;
            uint32_t st[8] = {0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au,
                              0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u};
            // Fast path, taken when every token of the batch is a one-block message (<= 55 bytes with its NUL) that lies
            // inside its claim's prefetched head: the padded block is put together a 32-bit word at a time from two
            // aligned LDS dwords (v_alignbyte), lower-cased four bytes at once, instead of byte by byte from global
            // memory (593 instructions and a dependent load per byte in the general path below).
            const bool one_block_in_head = !act || (nblk == 1 && s + tl <= EMB_HEAD);
            if (__all(one_block_in_head)) {
                uint32_t w[16];
                const uint32_t* h32 = reinterpret_cast<const uint32_t*>(&head[wave][ci][0]);
                const int nw_mine = act ? (total >> 2) + 1 : 0;           // words that hold token bytes or the 0x80 mark
                int nw = nw_mine;                                         // wave maximum (uniform): the rest is zero
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) { const int o2 = __shfl_xor(nw, off); nw = o2 > nw ? o2 : nw; }
                nw = __builtin_amdgcn_readfirstlane(nw);
                uint32_t f1 = 0x811C9DC5u ^ (uint32_t)tl, f2 = 0x9E3779B9u + (uint32_t)tl;  // two 32-bit hashes of the lowered token (memo)
#pragma unroll
                for (int i = 0; i < 14; ++i) {
                    uint32_t x = 0;
                    if (i < nw) {
                        // byte offset of this word in the claim's head; a lane whose own message ends earlier reads
                        // its first word again (everything is masked off below), so no read leaves the padded row
                        const int b0 = (i <= (total >> 2)) ? s + 4 * i : s;
                        const uint32_t lo = h32[b0 >> 2], hi = h32[(b0 >> 2) + 1];
                        x = __builtin_amdgcn_alignbyte(hi, lo, (uint32_t)(b0 & 3));  // tok[4i..4i+3], little-endian
                        const int nv = tl - 4 * i;                        // token bytes in this word
                        x = nv >= 4 ? x : (nv <= 0 ? 0u : (x & ((1u << (8 * nv)) - 1u)));
                        // lower-case: bytes 'A'..'Z' (0x41..0x5A) get +0x20; token bytes are ASCII (< 0x80)
                        const uint32_t ge_A = (x & 0x7F7F7F7Fu) + 0x3F3F3F3Fu;
                        const uint32_t gt_Z = (x & 0x7F7F7F7Fu) + 0x25252525u;
                        x += ((ge_A & ~gt_Z & ~x) & 0x80808080u) >> 2;
                        if (memo.table && nv > 0) {  // words that hold token bytes only (the loop runs to the WAVE's longest message); the same in both lanes of a token
                            f1 = (f1 ^ x) * 16777619u;
                            f2 = (f2 + x) * 0x85EBCA6Bu;
                            f2 = (f2 << 13) | (f2 >> 19);
                        }
                        const int mk = total - 4 * i;                     // position of the 0x80 mark relative to this word
                        if (mk >= 0 && mk < 4) x |= 0x80u << (8 * mk);
                        x = __builtin_bswap32(x);                         // message words are big-endian
                    }
                    w[i] = x;
                }
                w[14] = 0u;
                w[15] = (uint32_t)total * 8u;
                // ---- token memo (lb/latent.py:47-49 is a pure function of the lower-cased token): a table in HBM, kept for
                // the life of the context, maps a token to its 64 digest bytes.  Entry = 128 B: tag (launch epoch << 40 |
                // 40-bit fingerprint; 0 = empty), the 56 bytes of the token's padded one-block message (the exact key),
                // the two digests.  Buckets of four slots, filled front to back and never emptied, so a lookup stops at
                // the first empty slot.  An entry counts only when it was written by an EARLIER launch (tag epoch < this
                // launch's): it was complete before this kernel started, whichever XCD wrote it.  Entries of the running
                // launch are neither read (they may be half written, and another XCD's L2 may hold stale lines of them) nor
                // written twice (the tag is taken with an atomic compare-and-swap).  The saving comes when EVERY token of
                // a batch is found: the wave then copies 32 bytes per lane instead of running the compression.
                int hit = -1, vacant = -1;
                uint64_t fp = 0;
                uint32_t bucket = 0;
                bool all_hit = false;
                if (memo.table) {
                    fp = (((uint64_t)(f2 & 0xFFu)) << 32) | f1;
                    fp += fp == 0;
                    bucket = (f2 >> 8) & memo.mask & ~3u;
                    // Two rounds of independent loads (the wave is latency-bound here: two waves per SIMD, and the table
                    // is larger than an XCD's L2): (1) the bucket's four tags; (2) key and digest half of the slot whose
                    // fingerprint matches -- used only if the key then compares equal.
                    uint4 d0 = {0u, 0u, 0u, 0u}, d1 = {0u, 0u, 0u, 0u};
                    if (act) {
                        const uint64_t* e0 = memo.table + (size_t)bucket * 16;
                        uint64_t tg[4];
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) tg[jj] = e0[jj * 16];
                        int cand = -1;
#pragma unroll
                        for (int jj = 3; jj >= 0; --jj) {  // back to front: the first empty slot and the first match win
                            if (tg[jj] == 0) { vacant = jj; cand = -1; }
                            else if ((tg[jj] & 0xFFFFFFFFFFull) == fp) {
                                if ((uint32_t)(tg[jj] >> 40) < memo.epoch) cand = jj;
                                else { vacant = -2; cand = -1; }  // being written by this launch: neither usable nor ours to write
                            }
                        }
                        if (cand >= 0) {
                            const uint64_t* e = e0 + cand * 16;
                            const uint2 k0 = *reinterpret_cast<const uint2*>(e + 1);
                            const uint4 k1 = *reinterpret_cast<const uint4*>(e + 2);
                            const uint4 k2 = *reinterpret_cast<const uint4*>(e + 4);
                            const uint4 k3 = *reinterpret_cast<const uint4*>(e + 6);
                            const uint4* dg = reinterpret_cast<const uint4*>(e + 8) + v * 2;
                            d0 = dg[0];
                            d1 = dg[1];
                            const bool same = k0.x == w[0] && k0.y == w[1] && k1.x == w[2] && k1.y == w[3] && k1.z == w[4] && k1.w == w[5] &&
                                              k2.x == w[6] && k2.y == w[7] && k2.z == w[8] && k2.w == w[9] && k3.x == w[10] && k3.y == w[11] &&
                                              k3.z == w[12] && k3.w == w[13];
                            if (same) hit = cand;  // (lane (t, 1) compares against ITS message, which differs: it takes lane (t, 0)'s verdict below)
                            else vacant = -2;      // a fingerprint collision: this token is simply not memoised
                        }
                    }
                    hit = __shfl(hit, lane & ~1);  // lane (t, 1) learns what lane (t, 0) found
                    vacant = __shfl(vacant, lane & ~1);
                    all_hit = __all(!act || hit >= 0);
                    if (all_hit && act) {
                        uint32_t* dst = &dig[wave][t][v * 8];
                        dst[0] = d0.x; dst[1] = d0.y; dst[2] = d0.z; dst[3] = d0.w;
                        dst[4] = d1.x; dst[5] = d1.y; dst[6] = d1.z; dst[7] = d1.w;
                    }
                }
                K1_ADD(6, 0, 1);                  // batches on the one-block fast path
                if (all_hit) K1_ADD(7, 0, 1);     // ... of which every token came from the memo
                if (!all_hit) {
                    // a token nobody has entered yet: take its slot and store the key NOW (the compression below turns w[]
                    // into the message schedule), the digests once both variants are in LDS
                    unsigned long long* mine = nullptr;
                    if (memo.table && act && v == 0 && hit < 0 && vacant >= 0) {
                        unsigned long long* e = reinterpret_cast<unsigned long long*>(memo.table + (size_t)(bucket + vacant) * 16);
                        // (device-coherent re-check first: a frequent token is seen as missing by thousands of waves of the
                        // first launch, and that many compare-and-swaps on one address would queue up for milliseconds)
                        if (__hip_atomic_load(e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0ull &&
                            atomicCAS(e, 0ull, ((unsigned long long)memo.epoch << 40) | fp) == 0ull) {
                            mine = e;
                            uint32_t* kw = reinterpret_cast<uint32_t*>(e + 1);
#pragma unroll
                            for (int i = 0; i < 14; ++i) kw[i] = w[i];
                        }
                    }
                    if (act) sha256_compress(st, w);
                    if (act) {
#pragma unroll
                        for (int i = 0; i < 8; ++i) dig[wave][t][v * 8 + i] = __builtin_bswap32(st[i]);
                    }
                    if (memo.table) {
                        __builtin_amdgcn_wave_barrier();  // the neighbour lane's digest is in LDS
                        if (mine) {
                            uint32_t* dd = reinterpret_cast<uint32_t*>(mine + 8);
#pragma unroll
                            for (int i = 0; i < 16; ++i) dd[i] = dig[wave][t][i];
                        }
                    }
                }
                __builtin_amdgcn_wave_barrier();
            } else
            for (int blk = 0; __any(blk < nblk); ++blk) {
                if (blk < nblk) {
                    uint32_t w[16];
                    const int p0 = blk * 64;
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int p = p0 + 4 * i;
                        w[i] = (msg_byte(tok, tl, total, p) << 24) | (msg_byte(tok, tl, total, p + 1) << 16) |
                               (msg_byte(tok, tl, total, p + 2) << 8) | msg_byte(tok, tl, total, p + 3);
                    }
                    if (blk == nblk - 1) {
                        w[14] = (uint32_t)(((uint64_t)total * 8ull) >> 32);
                        w[15] = (uint32_t)((uint64_t)total * 8ull);
                    }
                    sha256_compress(st, w);
                }
            }
            if (!__all(one_block_in_head) && act) {  // general path: its digests (the fast path stored or copied its own above)
#pragma unroll
                for (int i = 0; i < 8; ++i) dig[wave][t][v * 8 + i] = __builtin_bswap32(st[i]);
            }
            __builtin_amdgcn_wave_barrier();
            K1_T(h1);
            // token order, lane = dimension.  A batch holds its tokens claim by claim (the ring is filled in claim order),
            // so each claim owns one contiguous run of it: the same tight loop as for a single claim, once per claim
            const int my_ci = lane < nb ? tc[(n_done + lane) & (TOK_RING - 1)] : -1;
#pragma unroll
            for (int c = 0; c < EMB_GROUP; ++c) {
                const uint64_t mk = __ballot(my_ci == c);
                if (mk) {
                    const int lo = __ffsll((long long)mk) - 1, hi = lo + __popcll(mk);
                    for (int i = lo; i < hi; ++i) acc[c] = acc[c] + lut[digb[i * 64 + lane]];
                }
            }
            __builtin_amdgcn_wave_barrier();
            K1_T(h2);
            K1_ADD(1, h0, h1);
            K1_ADD(2, h1, h2);
            K1_ADD(5, 0, 1);  // hash batches
        };

        for (int c = 0; c < gn; ++c) {
            const uint8_t* text = utf8 + cb[c];
            const int len = (int)(cb[c + 1] - cb[c]);
            const int ended_before = n_ended;
            uint64_t carry = 0;
            for (int base = 0; base <= len; base += 64) {
                const int p = base + lane;
                const uint32_t ch = (p < len) ? (p < EMB_HEAD ? (uint32_t)head[wave][c][p] : (uint32_t)text[p]) : 0u;
                const uint64_t m = __ballot(is_word(ch));
                const uint64_t prev = (m << 1) | carry;
                const uint64_t S = m & ~prev;   // token starts at this byte
                const uint64_t E = ~m & prev;   // token ended just before this byte (exclusive end)
                carry = m >> 63;
                if ((S >> lane) & 1) ts[(n_started + __popcll(S & lt_mask)) & (TOK_RING - 1)] = p;
                if ((E >> lane) & 1) {
                    const int slot = (n_ended + __popcll(E & lt_mask)) & (TOK_RING - 1);
                    te[slot] = p;
                    tc[slot] = c;
                }
                n_started += __popcll(S);
                n_ended += __popcll(E);
                __builtin_amdgcn_wave_barrier();
                while (n_ended - n_done >= BATCH) {
                    hash_batch(BATCH);
                    n_done += BATCH;
                }
            }
            if (n_ended > ended_before) has_tok |= 1u << c;
        }
        if (n_ended > n_done) hash_batch(n_ended - n_done);
        K1_T(c1);
        // lb/latent.py:54-57, claim by claim
#pragma unroll
        for (int c = 0; c < EMB_GROUP; ++c) {
            if (c < gn) {
                double ss = 0.0;
                if (dim == 64) {
#pragma unroll
                    for (int d = 0; d < 64; ++d) {
                        const double x = readlane_f64(acc[c], d);
                        ss = ss + x * x;
                    }
                } else {  // embed(text, dim) with dim < 64: the norm runs over the first `dim` components only (:54)
#pragma unroll
                    for (int d = 0; d < 64; ++d) {
                        const double x = readlane_f64(acc[c], d);
                        if (d < dim) ss = ss + x * x;
                    }
                }
                const double nrm = __builtin_sqrt(ss);
                double o = 0.0;
                if (((has_tok >> c) & 1u) && nrm != 0.0 && lane < dim) o = acc[c] / nrm;
                out64[(g0 + c) * 64 + lane] = o;
                if (out32) out32[(g0 + c) * 64 + lane] = (float)o;
            }
        }
        K1_T(c2);
        K1_ADD(0, c0, c1);
        K1_ADD(3, c1, c2);
        k1_count(gn);
    }
#ifdef LBH_K1_TIMING
    if (lane == 0)
        for (int i = 0; i < 8; ++i) atomicAdd(&lbh_k1_dbg[i], k1_t[i]);
#endif
}

// lb/latent.py:63  sum(x * y for x, y in zip(a, b)): one lane, strictly sequential, no FMA.
__global__ void cosine_kernel(const double* __restrict__ a, const double* __restrict__ b, int64_t n,
                              double* __restrict__ out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double acc = 0.0;
        for (int64_t i = 0; i < n; ++i) acc = acc + a[i] * b[i];
        *out = acc;
    }
}

__global__ void f64_to_f32_kernel(const double* __restrict__ in, float* __restrict__ out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = (float)in[i];
}

// fp16 copy of the matrix (fp16 scans): rows [row0, row1) of emb64 -> tiled emb16 (layout: lbh_emb16_off).  One
// rounding to nearest even straight from the stored fp64 value (the scans' error bound assumes it; a detour through
// fp32 would round twice).  One thread per (row, group of 8 dims): 64 B in, one 16-byte store out.
typedef _Float16 e_f16x8 __attribute__((ext_vector_type(8)));
__global__ void f64_to_f16_tiled_kernel(const double* __restrict__ emb64, _Float16* __restrict__ emb16, int64_t row0, int64_t row1) {
    const int64_t total = (row1 - row0) * 8;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = row0 + (i >> 3);
        const int g = (int)(i & 7);
        const double* src = emb64 + row * 64 + g * 8;
        e_f16x8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (_Float16)src[e];
        *reinterpret_cast<e_f16x8*>(emb16 + lbh_emb16_off(row, g * 8)) = v;
    }
}

// int8 copy, step 1: the largest |element| of rows [row0, row1) -> atomicMax on its float bits (non-negative floats order
// like their bit patterns)
__global__ void absmax_f64_kernel(const double* __restrict__ emb64, int64_t e0, int64_t e1, uint32_t* __restrict__ out_bits) {
    float m = 0.f;
    for (int64_t i = e0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < e1; i += (int64_t)gridDim.x * blockDim.x) {
        const float a = (float)__builtin_fabs(emb64[i]);
        // round UP to float when the conversion rounded down: the scale derived from this must cover the double
        const float up = ((double)a < __builtin_fabs(emb64[i])) ? lbh_next_up(a) : a;
        m = up > m ? up : m;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float t = __shfl_xor(m, o);
        m = t > m ? t : m;
    }
    if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(out_bits, __builtin_bit_cast(uint32_t, m));
}

// int8 copy, step 2: X = rint(x * inv_scale), clamped to [-127, 127] (the scale covers every element, so the clamp only
// guards against a caller's misuse), written in fragment order: one thread = 16 consecutive dims of one row = one 16-byte chunk
__global__ void f64_to_i8_tiled_kernel(const double* __restrict__ emb64, int8_t* __restrict__ emb8, int64_t row0, int64_t row1, double inv_scale) {
    const int64_t total = (row1 - row0) * 4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = row0 + (i >> 2);
        const int c = (int)(i & 3);  // dims 16 c .. 16 c + 15
        const double* src = emb64 + row * 64 + c * 16;
        union { int8_t b[16]; uint4 v; } u;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            int q = __double2int_rn(src[e] * inv_scale);
            q = q > 127 ? 127 : (q < -127 ? -127 : q);
            u.b[e] = (int8_t)q;
        }
        *reinterpret_cast<uint4*>(emb8 + lbh_emb8_off(row, c * 16)) = u.v;
    }
}

// swap-delete support: copy row `src` over row `dst` in every per-claim array
__global__ void move_row_kernel(float* emb32, double* emb64, uint16_t* emb16, int8_t* emb8, int64_t* created, uint64_t* tags, uint64_t* keys,
                                uint64_t* order, uint64_t* ids, int64_t dst, int64_t src) {
    const int l = threadIdx.x;
    emb32[dst * 64 + l] = emb32[src * 64 + l];
    if (emb16) emb16[lbh_emb16_off(dst, l)] = emb16[lbh_emb16_off(src, l)];
    if (emb8) emb8[lbh_emb8_off(dst, l)] = emb8[lbh_emb8_off(src, l)];
    emb64[dst * 64 + l] = emb64[src * 64 + l];
    if (l < 4) keys[dst * 4 + l] = keys[src * 4 + l];
    if (l == 4) created[dst] = created[src];
    if (l == 5) tags[dst] = tags[src];
    if (l == 6) order[dst] = order[src];
    if (l == 7) ids[dst] = ids[src];
}

// batched swap-delete (lbh_index_retract_many): pair p copies row pairs[2p+1] (a survivor from the tail that is cut
// off) over row pairs[2p] (a hole below the new end).  Sources and destinations are disjoint, so pairs are independent.
__global__ void move_rows_kernel(float* emb32, double* emb64, uint16_t* emb16, int8_t* emb8, int64_t* created, uint64_t* tags, uint64_t* keys,
                                 uint64_t* order, uint64_t* ids, const int64_t* __restrict__ pairs, int64_t n_pairs) {
    const int l = threadIdx.x & 63;
    const int64_t p = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (p >= n_pairs) return;
    const int64_t dst = pairs[2 * p], src = pairs[2 * p + 1];
    emb32[dst * 64 + l] = emb32[src * 64 + l];
    if (emb16) emb16[lbh_emb16_off(dst, l)] = emb16[lbh_emb16_off(src, l)];
    if (emb8) emb8[lbh_emb8_off(dst, l)] = emb8[lbh_emb8_off(src, l)];
    emb64[dst * 64 + l] = emb64[src * 64 + l];
    if (l < 4) keys[dst * 4 + l] = keys[src * 4 + l];
    if (l == 4) created[dst] = created[src];
    if (l == 5) tags[dst] = tags[src];
    if (l == 6) order[dst] = order[src];
    if (l == 7) ids[dst] = ids[src];
}

// out[i] = emb64 row rows[i] (batched _embedding_cache reads)
__global__ void gather_rows_f64_kernel(const double* __restrict__ emb64, const int64_t* __restrict__ rows, int64_t n,
                                       double* __restrict__ out) {
    const int l = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (i < n) out[i * 64 + l] = emb64[rows[i] * 64 + l];
}

// order[r0 + i] = first + i: rows appended without an explicit order sort by insertion
__global__ void iota_u64_kernel(uint64_t* dst, uint64_t first, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = first + (uint64_t)i;
}

// in place: the 32 key bytes of a claim as uploaded -> their four big-endian words (word 0 most significant)
__global__ void bswap_u64_kernel(uint64_t* p, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        p[i] = __builtin_bswap64(p[i]);
}

// Metadata of a SMALL batch (lbh_index_add, n <= 256 = one block) straight from device-mapped host memory: timestamps, tag masks, the
// 32 key bytes as four big-endian words, order stamps and claim ids of rows r0 .. r0 + n - 1 -- one launch where the
// bulk path makes three copies and three kernels.
__global__ void small_meta_kernel(const int64_t* __restrict__ created_src, const uint64_t* __restrict__ tags_src,
                                  const uint64_t* __restrict__ keys_src, int64_t* created, uint64_t* tags, uint64_t* keys,
                                  uint64_t* order, uint64_t* ids, int64_t r0, uint64_t id0, int n) {
    const int t = threadIdx.x;
    for (int i = t; i < n * 4; i += blockDim.x) keys[r0 * 4 + i] = __builtin_bswap64(keys_src[i]);
    if (t < n) {
        created[r0 + t] = created_src[t];
        tags[r0 + t] = tags_src ? tags_src[t] : 0ull;
        order[r0 + t] = id0 + (uint64_t)t;
        ids[r0 + t] = id0 + (uint64_t)t;
    }
}

// dst[rows[i]] = vals[i]  (rows are distinct: one claim id maps to one row)
__global__ void scatter_u64_kernel(uint64_t* dst, const int64_t* __restrict__ rows, const uint64_t* __restrict__ vals, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[rows[i]] = vals[i];
}

}  // namespace

hipError_t lbh_launch_embed(const uint8_t* utf8_dev, const int64_t* offsets_dev, int64_t n, double* out64_dev,
                            float* out32_dev, hipStream_t s, int dim, const lbh_memo* memo) {
    if (n <= 0) return hipSuccess;
    int64_t blocks = (n + EMB_WAVES * EMB_GROUP - 1) / (EMB_WAVES * EMB_GROUP);
    if (blocks > 256 * 32) blocks = 256 * 32;
    embed_kernel<<<dim3((unsigned)blocks), dim3(EMB_WAVES * 64), 0, s>>>(utf8_dev, offsets_dev, n, out64_dev, out32_dev, dim,
                                                                         memo ? *memo : lbh_memo{nullptr, 0, 0});
    return hipGetLastError();
}

hipError_t lbh_launch_cosine(const double* a_dev, const double* b_dev, int64_t n, double* out_dev, hipStream_t s) {
    cosine_kernel<<<1, 64, 0, s>>>(a_dev, b_dev, n, out_dev);
    return hipGetLastError();
}

hipError_t lbh_launch_f64_to_f32(const double* in_dev, float* out_dev, int64_t n_elems, hipStream_t s) {
    if (n_elems <= 0) return hipSuccess;
    int64_t blocks = (n_elems + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    f64_to_f32_kernel<<<dim3((unsigned)blocks), 256, 0, s>>>(in_dev, out_dev, n_elems);
    return hipGetLastError();
}

hipError_t lbh_launch_move_row(float* emb32, double* emb64, uint16_t* emb16, int8_t* emb8, int64_t* created, uint64_t* tags, uint64_t* keys,
                               uint64_t* order, uint64_t* ids, int64_t dst, int64_t src, hipStream_t s) {
    move_row_kernel<<<1, 64, 0, s>>>(emb32, emb64, emb16, emb8, created, tags, keys, order, ids, dst, src);
    return hipGetLastError();
}

hipError_t lbh_launch_move_rows(float* emb32, double* emb64, uint16_t* emb16, int8_t* emb8, int64_t* created, uint64_t* tags, uint64_t* keys,
                                uint64_t* order, uint64_t* ids, const int64_t* pairs_dev, int64_t n_pairs, hipStream_t s) {
    if (n_pairs <= 0) return hipSuccess;
    move_rows_kernel<<<dim3((unsigned)((n_pairs + 3) / 4)), 256, 0, s>>>(emb32, emb64, emb16, emb8, created, tags, keys, order, ids, pairs_dev, n_pairs);
    return hipGetLastError();
}

hipError_t lbh_launch_small_meta(const int64_t* created_src, const uint64_t* tags_src, const uint64_t* keys_src, int64_t* created,
                                 uint64_t* tags, uint64_t* keys, uint64_t* order, uint64_t* ids, int64_t r0, uint64_t id0, int n,
                                 hipStream_t s) {
    small_meta_kernel<<<1, 256, 0, s>>>(created_src, tags_src, keys_src, created, tags, keys, order, ids, r0, id0, n);
    return hipGetLastError();
}

hipError_t lbh_launch_absmax_f64(const double* emb64_dev, int64_t row0, int64_t row1, uint32_t* out_bits_dev, hipStream_t s) {
    if (row1 <= row0) return hipSuccess;
    const int64_t e0 = row0 * 64, e1 = row1 * 64;
    int64_t blocks = (e1 - e0 + 256 * 8 - 1) / (256 * 8);
    if (blocks > 4096) blocks = 4096;
    absmax_f64_kernel<<<dim3((unsigned)blocks), 256, 0, s>>>(emb64_dev, e0, e1, out_bits_dev);
    return hipGetLastError();
}

hipError_t lbh_launch_f64_to_i8_tiled(const double* emb64_dev, int8_t* emb8_dev, int64_t row0, int64_t row1, double inv_scale, hipStream_t s) {
    if (row1 <= row0) return hipSuccess;
    int64_t blocks = ((row1 - row0) * 4 + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    f64_to_i8_tiled_kernel<<<dim3((unsigned)blocks), 256, 0, s>>>(emb64_dev, emb8_dev, row0, row1, inv_scale);
    return hipGetLastError();
}

hipError_t lbh_launch_gather_rows_f64(const double* emb64, const int64_t* rows_dev, int64_t n, double* out_dev, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    gather_rows_f64_kernel<<<dim3((unsigned)((n + 3) / 4)), 256, 0, s>>>(emb64, rows_dev, n, out_dev);
    return hipGetLastError();
}

hipError_t lbh_launch_f64_to_f16_tiled(const double* emb64_dev, uint16_t* emb16_dev, int64_t row0, int64_t row1, hipStream_t s) {
    if (row1 <= row0) return hipSuccess;
    int64_t blocks = ((row1 - row0) * 8 + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    f64_to_f16_tiled_kernel<<<dim3((unsigned)blocks), 256, 0, s>>>(emb64_dev, reinterpret_cast<_Float16*>(emb16_dev), row0, row1);
    return hipGetLastError();
}

hipError_t lbh_launch_bswap_u64(uint64_t* p, int64_t n, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    bswap_u64_kernel<<<dim3((unsigned)blocks), 256, 0, s>>>(p, n);
    return hipGetLastError();
}

hipError_t lbh_launch_iota_u64(uint64_t* dst, uint64_t first, int64_t n, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    iota_u64_kernel<<<dim3((unsigned)((n + 255) / 256)), 256, 0, s>>>(dst, first, n);
    return hipGetLastError();
}

hipError_t lbh_launch_scatter_u64(uint64_t* dst, const int64_t* rows_dev, const uint64_t* vals_dev, int64_t n, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    scatter_u64_kernel<<<dim3((unsigned)((n + 255) / 256)), 256, 0, s>>>(dst, rows_dev, vals_dev, n);
    return hipGetLastError();
}

#ifdef LBH_K1_TIMING
extern "C" int lbh_debug_k1_times(unsigned long long* out, int reset) {  // instrumented build only
    if (reset) {
        unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        return (int)hipMemcpyToSymbol(HIP_SYMBOL(lbh_k1_dbg), z, sizeof z);
    }
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(lbh_k1_dbg), sizeof(unsigned long long) * 8);
}
#endif
