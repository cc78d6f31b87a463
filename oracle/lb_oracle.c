/*
 * lb_oracle.c -- CPU restatement of the reference's latent-retrieval path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity oracle: it may be used by
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, and by
 * nothing else.  The product (lbm_amd/ + liblbhip.so) never links, loads or
 * calls it.
 *
 * Parity status: PINNED.  Every function below is checked bit-for-bit against
 * golden vectors produced by importing the reference itself
 * (tests/golden/make_golden.py, run in the build container under CPython
 * 3.10.12) -- see tests/test_oracle_golden.py.
 *
 * What it restates (paths relative to the reference tree):
 *   lbo_tokenize      lb/latent.py:34-38   _WORD_RE / _tokenize
 *   lbo_sha256        hashlib.sha256 as used at lb/latent.py:47-48 (FIPS 180-4)
 *   lbo_embed         lb/latent.py:41-57   embed
 *   lbo_cosine        lb/latent.py:60-63   cosine
 *   lbo_compile       lb/context_graph.py:104-121  filter + score + sort + cut
 *
 * Arithmetic rules that make this bit-exact with CPython 3.10 floats:
 *   - IEEE binary64 everywhere, no FMA contraction (build with
 *     -ffp-contract=off; no -ffast-math), strictly left-to-right sums that
 *     start from 0.0 (CPython's sum() starts from int 0; 0 + x == 0.0 + x).
 *   - (b / 127.5) - 1.0 is evaluated exactly as written (lb/latent.py:52).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define LBO_DIM 64

/* ---------------------------------------------------------------- SHA-256 */

static const uint32_t K256[64] = {
    0x428a2f98u, 0x71374491u, 0xb5c0fbcfu, 0xe9b5dba5u, 0x3956c25bu, 0x59f111f1u, 0x923f82a4u, 0xab1c5ed5u,
    0xd807aa98u, 0x12835b01u, 0x243185beu, 0x550c7dc3u, 0x72be5d74u, 0x80deb1feu, 0x9bdc06a7u, 0xc19bf174u,
    0xe49b69c1u, 0xefbe4786u, 0x0fc19dc6u, 0x240ca1ccu, 0x2de92c6fu, 0x4a7484aau, 0x5cb0a9dcu, 0x76f988dau,
    0x983e5152u, 0xa831c66du, 0xb00327c8u, 0xbf597fc7u, 0xc6e00bf3u, 0xd5a79147u, 0x06ca6351u, 0x14292967u,
    0x27b70a85u, 0x2e1b2138u, 0x4d2c6dfcu, 0x53380d13u, 0x650a7354u, 0x766a0abbu, 0x81c2c92eu, 0x92722c85u,
    0xa2bfe8a1u, 0xa81a664bu, 0xc24b8b70u, 0xc76c51a3u, 0xd192e819u, 0xd6990624u, 0xf40e3585u, 0x106aa070u,
    0x19a4c116u, 0x1e376c08u, 0x2748774cu, 0x34b0bcb5u, 0x391c0cb3u, 0x4ed8aa4au, 0x5b9cca4fu, 0x682e6ff3u,
    0x748f82eeu, 0x78a5636fu, 0x84c87814u, 0x8cc70208u, 0x90befffau, 0xa4506cebu, 0xbef9a3f7u, 0xc67178f2u};

static inline uint32_t rotr32(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }

static void sha256_block(uint32_t st[8], const uint8_t blk[64]) {
    uint32_t w[64];
    for (int i = 0; i < 16; ++i)
        w[i] = ((uint32_t)blk[4 * i] << 24) | ((uint32_t)blk[4 * i + 1] << 16) | ((uint32_t)blk[4 * i + 2] << 8) |
               (uint32_t)blk[4 * i + 3];
    for (int i = 16; i < 64; ++i) {
        uint32_t s0 = rotr32(w[i - 15], 7) ^ rotr32(w[i - 15], 18) ^ (w[i - 15] >> 3);
        uint32_t s1 = rotr32(w[i - 2], 17) ^ rotr32(w[i - 2], 19) ^ (w[i - 2] >> 10);
        w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    uint32_t a = st[0], b = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6], h = st[7];
    for (int i = 0; i < 64; ++i) {
        uint32_t S1 = rotr32(e, 6) ^ rotr32(e, 11) ^ rotr32(e, 25);
        uint32_t ch = (e & f) ^ (~e & g);
        uint32_t t1 = h + S1 + ch + K256[i] + w[i];
        uint32_t S0 = rotr32(a, 2) ^ rotr32(a, 13) ^ rotr32(a, 22);
        uint32_t mj = (a & b) ^ (a & c) ^ (b & c);
        uint32_t t2 = S0 + mj;
        h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    st[0] += a; st[1] += b; st[2] += c; st[3] += d; st[4] += e; st[5] += f; st[6] += g; st[7] += h;
}

/* SHA-256 of msg[0..len) followed by `extra_nul` (0 or 1) 0x00 bytes; bytes are
 * passed through `lower` (ASCII A-Z -> a-z) first when lower != 0.  The extra
 * NUL and the lowering restate  t.lower()  and  (t + "\x00")  of
 * lb/latent.py:38,47-48 without materialising the strings. */
static void sha256_token(const uint8_t* msg, size_t len, int extra_nul, int lower, uint8_t out[32]) {
    uint32_t st[8] = {0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au,
                      0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u};
    size_t total = len + (size_t)(extra_nul ? 1 : 0);
    uint8_t blk[64];
    size_t pos = 0; /* position in the (virtual) message */
    int done = 0, wrote80 = 0;
    while (!done) {
        int fill = 0;
        while (fill < 64 && pos < total) {
            uint8_t c = (pos < len) ? msg[pos] : 0x00;
            if (lower && c >= 'A' && c <= 'Z') c = (uint8_t)(c + 32);
            blk[fill++] = c;
            ++pos;
        }
        if (fill < 64 && !wrote80) { blk[fill++] = 0x80; wrote80 = 1; }
        if (wrote80 && fill <= 56) {
            while (fill < 56) blk[fill++] = 0;
            uint64_t bits = (uint64_t)total * 8u;
            for (int i = 0; i < 8; ++i) blk[56 + i] = (uint8_t)(bits >> (56 - 8 * i));
            done = 1;
        } else {
            while (fill < 64) blk[fill++] = 0;
        }
        sha256_block(st, blk);
    }
    for (int i = 0; i < 8; ++i) {
        out[4 * i] = (uint8_t)(st[i] >> 24); out[4 * i + 1] = (uint8_t)(st[i] >> 16);
        out[4 * i + 2] = (uint8_t)(st[i] >> 8); out[4 * i + 3] = (uint8_t)st[i];
    }
}

void lbo_sha256(const uint8_t* msg, int64_t len, uint8_t out[32]) { sha256_token(msg, (size_t)len, 0, 0, out); }

/* -------------------------------------------------------------- tokeniser */

/* lb/latent.py:34  _WORD_RE = [A-Za-z0-9_]+ : an explicit ASCII class, so every
 * byte >= 0x80 of the UTF-8 encoding is a separator. */
static inline int is_word(uint8_t c) {
    return (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || (c >= '0' && c <= '9') || c == '_';
}

/* Writes up to `max_tokens` (start, length) pairs; returns the TOTAL number of
 * tokens in the text (may exceed max_tokens). */
int64_t lbo_tokenize(const uint8_t* utf8, int64_t len, int64_t* starts, int64_t* lens, int64_t max_tokens) {
    int64_t n = 0, i = 0;
    while (i < len) {
        while (i < len && !is_word(utf8[i])) ++i;
        if (i >= len) break;
        int64_t s = i;
        while (i < len && is_word(utf8[i])) ++i;
        if (n < max_tokens) { starts[n] = s; lens[n] = i - s; }
        ++n;
    }
    return n;
}

/* ------------------------------------------------------------------ embed */

static double LUT[256];
static int lut_ready = 0;
static void lut_init(void) {
    if (lut_ready) return;
    for (int b = 0; b < 256; ++b) LUT[b] = ((double)b / 127.5) - 1.0; /* lb/latent.py:52 */
    lut_ready = 1;
}

double lbo_lut(int b) { lut_init(); return LUT[b & 255]; }

/* lb/latent.py:41-57 for 1 <= dim <= 64: only the first `dim` digest bytes are accumulated (:50-52) and the norm runs
 * over those `dim` components (:54); out receives `dim` doubles.  (dim > 64 is the reference's IndexError and dim = 0
 * its empty list: the Python front-end answers both.) */
void lbo_embed_dim(const uint8_t* utf8, int64_t len, int dim, double* out) {
    lut_init();
    double v[LBO_DIM];
    for (int d = 0; d < dim; ++d) v[d] = 0.0;
    int64_t i = 0, ntok = 0;
    uint8_t b[64];
    while (i < len) {
        while (i < len && !is_word(utf8[i])) ++i;
        if (i >= len) break;
        int64_t s = i;
        while (i < len && is_word(utf8[i])) ++i;
        sha256_token(utf8 + s, (size_t)(i - s), 0, 1, b);      /* :47 */
        sha256_token(utf8 + s, (size_t)(i - s), 1, 1, b + 32); /* :48 */
        for (int d = 0; d < dim; ++d) v[d] += LUT[b[d]];       /* :50-52 */
        ++ntok;
    }
    if (ntok == 0) { for (int d = 0; d < dim; ++d) out[d] = 0.0; return; } /* :43-44 */
    double acc = 0.0;
    for (int d = 0; d < dim; ++d) acc = acc + v[d] * v[d]; /* :54 sum(x*x) */
    double n = sqrt(acc);
    if (n == 0.0) { for (int d = 0; d < dim; ++d) out[d] = 0.0; return; } /* :55-56 */
    for (int d = 0; d < dim; ++d) out[d] = v[d] / n; /* :57 */
}

/* lb/latent.py:41-57 with dim = 64 (the only value any caller uses). */
void lbo_embed(const uint8_t* utf8, int64_t len, double out[LBO_DIM]) {
    lut_init();
    double v[LBO_DIM];
    for (int d = 0; d < LBO_DIM; ++d) v[d] = 0.0;
    int64_t i = 0, ntok = 0;
    uint8_t b[64];
    while (i < len) {
        while (i < len && !is_word(utf8[i])) ++i;
        if (i >= len) break;
        int64_t s = i;
        while (i < len && is_word(utf8[i])) ++i;
        sha256_token(utf8 + s, (size_t)(i - s), 0, 1, b);      /* :47 */
        sha256_token(utf8 + s, (size_t)(i - s), 1, 1, b + 32); /* :48 */
        for (int d = 0; d < LBO_DIM; ++d) v[d] += LUT[b[d]];   /* :50-52 */
        ++ntok;
    }
    if (ntok == 0) { for (int d = 0; d < LBO_DIM; ++d) out[d] = 0.0; return; } /* :43-44 */
    double acc = 0.0;
    for (int d = 0; d < LBO_DIM; ++d) acc = acc + v[d] * v[d]; /* :54 sum(x*x) */
    double n = sqrt(acc);
    if (n == 0.0) { for (int d = 0; d < LBO_DIM; ++d) out[d] = 0.0; return; } /* :55-56 */
    for (int d = 0; d < LBO_DIM; ++d) out[d] = v[d] / n; /* :57 */
}

void lbo_embed_batch(const uint8_t* utf8, const int64_t* offsets, int64_t n, double* out) {
    for (int64_t i = 0; i < n; ++i) lbo_embed(utf8 + offsets[i], offsets[i + 1] - offsets[i], out + i * LBO_DIM);
}

/* lb/latent.py:60-63 (length check is the caller's: -1 signals the mismatch). */
int lbo_cosine(const double* a, int64_t na, const double* b, int64_t nb, double* out) {
    if (na != nb) return -1;
    double acc = 0.0;
    for (int64_t i = 0; i < na; ++i) acc = acc + a[i] * b[i];
    *out = acc;
    return 0;
}

/* ---------------------------------------------------------------- compile */

typedef struct {
    double neg_s; /* -score, the reference's primary sort key (lb/context_graph.py:120) */
    int64_t idx;
} lbo_scored;

static const uint8_t* g_keys;
static const int64_t* g_key_off;

/* Python str ordering == lexicographic order of the UTF-8 bytes (UTF-8
 * preserves code-point order), shorter prefix first. */
static int key_cmp(int64_t a, int64_t b) {
    int64_t la = g_key_off[a + 1] - g_key_off[a], lb = g_key_off[b + 1] - g_key_off[b];
    int64_t m = la < lb ? la : lb;
    int c = memcmp(g_keys + g_key_off[a], g_keys + g_key_off[b], (size_t)m);
    if (c) return c;
    return (la > lb) - (la < lb);
}

static int scored_cmp(const void* pa, const void* pb) {
    const lbo_scored* a = (const lbo_scored*)pa;
    const lbo_scored* b = (const lbo_scored*)pb;
    if (a->neg_s < b->neg_s) return -1;
    if (a->neg_s > b->neg_s) return 1;
    return key_cmp(a->idx, b->idx);
}

/*
 * lb/context_graph.py:104-121 for one query vector.
 *   emb        n x 64 doubles (row i = cached embedding of claim i)
 *   created_ms n int64; retracted n bytes (non-zero = skip, :107)
 *   tagmask    n u64 or NULL; tag_filter 0 = no tag filter (extension; the
 *              reference has none -- SURVEY.md fact 2)
 *   keys/key_off  claim-hash strings as UTF-8, CSR offsets (n+1)
 *   has_since/since_ms  :110  (since_ms is not None and created_ms < since_ms)
 *   top_k      :121  max(0, int(top_k))
 * Writes min(top_k, survivors) indices + scores; returns that count.
 * Not re-entrant (qsort comparator uses file-scope pointers).
 */
int64_t lbo_compile(const double* emb, const int64_t* created_ms, const uint8_t* retracted,
                    const uint64_t* tagmask, uint64_t tag_filter, const uint8_t* keys,
                    const int64_t* key_off, int64_t n, const double* q, int has_since, int64_t since_ms,
                    int64_t top_k, int64_t* out_idx, double* out_scores) {
    lbo_scored* sc = (lbo_scored*)malloc(sizeof(lbo_scored) * (size_t)(n > 0 ? n : 1));
    int64_t m = 0;
    for (int64_t i = 0; i < n; ++i) {
        if (retracted && retracted[i]) continue;
        if (has_since && created_ms[i] < since_ms) continue;
        if (tag_filter && !(tagmask && (tagmask[i] & tag_filter))) continue;
        const double* cv = emb + i * LBO_DIM;
        double acc = 0.0;
        for (int d = 0; d < LBO_DIM; ++d) acc = acc + q[d] * cv[d]; /* cosine(qv, cv): x from qv */
        sc[m].neg_s = -acc;
        sc[m].idx = i;
        ++m;
    }
    g_keys = keys;
    g_key_off = key_off;
    qsort(sc, (size_t)m, sizeof(lbo_scored), scored_cmp);
    int64_t k = top_k < 0 ? 0 : top_k;
    if (k > m) k = m;
    for (int64_t j = 0; j < k; ++j) {
        out_idx[j] = sc[j].idx;
        out_scores[j] = -sc[j].neg_s;
    }
    free(sc);
    return k;
}
