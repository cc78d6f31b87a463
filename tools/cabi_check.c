/* Standalone C program over the C ABI (include/lbhip.h): what a maintainer binding liblbhip from C would write, and the
 * host-side memory-safety check of the library -- built with AddressSanitizer on the HOST code only
 * (tools/scripts/cabi_asan.sh) it walks every entry-point family, including searches from several threads next to a writer.
 * Results are checked for self-consistency (the GPU parity tests compare against the oracle; this program looks for
 * host-side memory errors, leaks of logic between calls and error-path handling).
 *   usage: cabi_check [claims]        exit code 0 = all checks passed */
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "lbhip.h"

#define CHECK(cond)                                                                        \
    do {                                                                                   \
        if (!(cond)) {                                                                     \
            fprintf(stderr, "FAILED %s:%d: %s   (last error: %s)\n", __FILE__, __LINE__, #cond, lbh_last_error()); \
            exit(1);                                                                       \
        }                                                                                  \
    } while (0)
#define OK(call) CHECK((call) == LBH_OK)

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint32_t rnd(void) {
    rng_state = rng_state * 6364136223846793005ull + 1442695040888963407ull;
    return (uint32_t)(rng_state >> 33);
}

/* n synthetic claims "wNNNNN wNNNNN ..." (3..10 tokens) packed like lbh_index_add wants them */
static void make_texts(int64_t n, uint8_t** blob, int64_t** off) {
    *off = (int64_t*)malloc((size_t)(n + 1) * sizeof(int64_t));
    *blob = (uint8_t*)malloc((size_t)n * 80 + 1);
    int64_t pos = 0;
    for (int64_t i = 0; i < n; ++i) {
        (*off)[i] = pos;
        const int nt = 3 + (int)(rnd() % 8);
        for (int t = 0; t < nt; ++t) pos += sprintf((char*)*blob + pos, "%sw%05u", t ? " " : "", rnd() % 3000u);
    }
    (*off)[n] = pos;
}

typedef struct {
    lbh_index* ix;
    const double* q;      /* nq x 64 */
    const uint64_t* want; /* nq x 8 ids */
    int nq, reps, id;
    int failures;
} reader_arg;

static void* reader(void* p) {
    reader_arg* a = (reader_arg*)p;
    uint64_t ids[8];
    double sc[8];
    uint8_t keys[8 * LBH_KEY_BYTES];
    for (int r = 0; r < a->reps; ++r) {
        const int j = (r * 3 + a->id) % a->nq;
        int64_t cnt = 0;
        if (lbh_index_search_shard(a->ix, a->q + j * 64, 8, LBH_SINCE_NONE, 0, ids, sc, keys, &cnt, NULL) != LBH_OK || cnt != 8 ||
            memcmp(ids, a->want + j * 8, sizeof ids) != 0)
            a->failures++;
    }
    return NULL;
}

typedef struct {
    lbh_index* ix;
    volatile int stop;
    int rounds;
} writer_arg;

static void* writer(void* p) { /* zero vectors (score 0): never in a top-8 whose scores are positive */
    writer_arg* a = (writer_arg*)p;
    enum { M = 37 };
    double* z = (double*)calloc((size_t)M * 64, sizeof(double));
    int64_t created[M];
    uint8_t keys[M * LBH_KEY_BYTES];
    uint64_t ids[M];
    while (!a->stop) {
        for (int i = 0; i < M; ++i) {
            created[i] = 1;
            for (int b = 0; b < LBH_KEY_BYTES; ++b) keys[i * LBH_KEY_BYTES + b] = (uint8_t)rnd();
        }
        if (lbh_index_add(a->ix, M, z, NULL, NULL, created, NULL, keys, ids) != LBH_OK) break;
        if (lbh_index_retract(a->ix, ids[0]) != LBH_OK) break;
        if (lbh_index_retract_many(a->ix, M - 1, ids + 1) != LBH_OK) break;
        a->rounds++;
    }
    free(z);
    return NULL;
}

int main(int argc, char** argv) {
    const int64_t n = argc > 1 ? atoll(argv[1]) : 50000;
    lbh_ctx* ctx = NULL;
    int dev = 0;
    OK(lbh_init(&dev, 1, &ctx));
    char name[128];
    OK(lbh_device_name(ctx, name, (int)sizeof name));
    printf("device %s, ABI version %d\n", name, lbh_version());

    /* error paths first: they must not leave anything behind */
    {
        lbh_ctx* none = NULL;
        CHECK(lbh_init(&dev, 2, &none) == LBH_E_INVALID && none == NULL);
    }
    CHECK(lbh_index_create(NULL, NULL) == LBH_E_INVALID);
    CHECK(lbh_embed_batch(ctx, NULL, NULL, 1, NULL) != LBH_OK);
    CHECK(strlen(lbh_last_error()) > 0);

    /* embed + cosine */
    const char* t2 = "hellohello world";
    const int64_t off2[3] = {0, 5, 16};
    double e2[2 * 64], cs = 0.0, e16[64];
    OK(lbh_embed_batch(ctx, (const uint8_t*)t2, off2, 2, e2));
    OK(lbh_cosine(ctx, e2, 64, e2, 64, &cs));
    CHECK(cs > 0.999999 && cs < 1.000001);
    CHECK(lbh_cosine(ctx, e2, 64, e2, 63, &cs) == LBH_E_INVALID);
    OK(lbh_embed_batch_dim(ctx, (const uint8_t*)t2, off2, 1, 16, e16));
    CHECK(e16[15] != 0.0 && e16[16] == 0.0);

    /* an index from texts, in pinned buffers for half of it */
    uint8_t* blob;
    int64_t* off;
    make_texts(n, &blob, &off);
    int64_t* created = (int64_t*)malloc((size_t)n * 8);
    uint64_t* tagm = (uint64_t*)malloc((size_t)n * 8);
    uint8_t* keys = (uint8_t*)malloc((size_t)n * LBH_KEY_BYTES);
    uint64_t* ids = (uint64_t*)malloc((size_t)n * 8);
    for (int64_t i = 0; i < n; ++i) {
        created[i] = 1000 + i;
        tagm[i] = 1ull << (rnd() % 64);
        for (int b = 0; b < LBH_KEY_BYTES; ++b) keys[i * LBH_KEY_BYTES + b] = (uint8_t)rnd();
    }
    lbh_index* ix = NULL;
    OK(lbh_index_create(ctx, &ix));
    const int64_t half = n / 2;
    OK(lbh_index_add(ix, half, NULL, blob, off, created, tagm, keys, ids));
    void* pin = NULL;
    OK(lbh_host_alloc(ctx, (n - half) * 8, &pin));
    memcpy(pin, created + half, (size_t)(n - half) * 8);
    OK(lbh_index_add(ix, n - half, NULL, blob, off + half, (const int64_t*)pin, NULL, keys + half * LBH_KEY_BYTES, ids + half));
    OK(lbh_host_free(ctx, pin));
    int64_t live = 0;
    OK(lbh_index_size(ix, &live));
    CHECK(live == n);
    CHECK(lbh_index_add(ix, 1, NULL, NULL, NULL, created, NULL, keys, ids) == LBH_E_INVALID);

    /* queries: text path == vector path; batched == single; paging beyond 64 */
    enum { NQ = 12 };
    uint8_t* qblob;
    int64_t* qoff;
    make_texts(NQ, &qblob, &qoff);
    double* qv = (double*)malloc(NQ * 64 * sizeof(double));
    OK(lbh_embed_batch(ctx, qblob, qoff, NQ, qv));
    uint64_t want[NQ * 8];
    double want_sc[NQ * 8];
    lbh_stats st;
    memset(&st, 0, sizeof st);
    st.want_timing = 2;
    for (int j = 0; j < NQ; ++j) {
        int64_t cnt = 0;
        uint64_t a[8];
        double sa[8], qvec[64];
        OK(lbh_index_search_text(ix, qblob + qoff[j], qoff[j + 1] - qoff[j], 8, LBH_SINCE_NONE, 0, a, sa, &cnt, qvec, &st));
        CHECK(cnt == 8 && st.rows_scanned == n && memcmp(qvec, qv + j * 64, sizeof qvec) == 0);
        uint8_t kk[8 * LBH_KEY_BYTES];
        OK(lbh_index_search_shard(ix, qv + j * 64, 8, LBH_SINCE_NONE, 0, want + j * 8, want_sc + j * 8, kk, &cnt, NULL));
        CHECK(cnt == 8 && memcmp(a, want + j * 8, sizeof a) == 0 && memcmp(sa, want_sc + j * 8, sizeof sa) == 0);
        CHECK(want_sc[j * 8] >= want_sc[j * 8 + 7] && want_sc[j * 8 + 7] > 0.0);
    }
    {
        uint64_t bi[NQ * 8];
        double bs[NQ * 8];
        int64_t bc[NQ];
        OK(lbh_index_search(ix, qv, NQ, 8, LBH_SINCE_NONE, 0, bi, bs, bc, &st));
        CHECK(st.path == 2 && memcmp(bi, want, sizeof bi) == 0 && memcmp(bs, want_sc, sizeof bs) == 0);
        uint64_t big[200];
        double bigs[200];
        int64_t c1 = 0;
        OK(lbh_index_search(ix, qv, 1, 200, LBH_SINCE_NONE, 0, big, bigs, &c1, NULL));
        CHECK(c1 == 200 && memcmp(big, want, 8 * sizeof(uint64_t)) == 0);
        for (int i = 1; i < 200; ++i) CHECK(bigs[i - 1] >= bigs[i]);
        OK(lbh_index_search(ix, qv, 1, 8, created[n / 2], 1ull << 3, big, bigs, &c1, &st)); /* both fused masks */
        CHECK(st.bytes_per_row == 272 && c1 <= 8);
        OK(lbh_index_search(ix, qv, 0, 8, LBH_SINCE_NONE, 0, NULL, NULL, NULL, NULL));
    }

    /* recency, metadata, embeddings, tombstones */
    {
        uint64_t rid[64];
        int64_t rcr[64], rc = 0, matched = 0;
        OK(lbh_index_recent(ix, created[n - 10], 64, rid, rcr, &rc, &matched, NULL));
        CHECK(rc == 10 && matched == 10 && rcr[0] == created[n - 1] && rid[0] == ids[n - 1]);
        double one[64], many[3 * 64];
        OK(lbh_index_get_embedding(ix, ids[5], one));
        const uint64_t three[3] = {ids[7], ids[5], ids[n - 1]};
        OK(lbh_index_get_embeddings(ix, 3, three, many));
        CHECK(memcmp(one, many + 64, sizeof one) == 0);
        OK(lbh_index_update_meta(ix, ids[5], 5, 0, keys + 5 * LBH_KEY_BYTES));
        const uint64_t ord[2] = {7, 3};
        OK(lbh_index_set_order(ix, 2, three, ord));
        OK(lbh_index_set_keys(ix, 1, three, keys + 7 * LBH_KEY_BYTES));
        CHECK(lbh_index_retract(ix, 1ull << 40) == LBH_E_NOTFOUND);
        const uint64_t dup[2] = {ids[9], ids[9]};
        CHECK(lbh_index_retract_many(ix, 2, dup) == LBH_E_INVALID);
        OK(lbh_index_size(ix, &live));
        CHECK(live == n); /* a refused batch changes nothing */
        OK(lbh_index_retract_many(ix, 100, ids + 100));
        OK(lbh_index_retract(ix, ids[300]));
        OK(lbh_index_size(ix, &live));
        CHECK(live == n - 101);
        CHECK(lbh_index_get_embedding(ix, ids[150], one) == LBH_E_NOTFOUND);
        OK(lbh_index_add(ix, 100, NULL, blob, off + 100, created + 100, tagm + 100, keys + 100 * LBH_KEY_BYTES, ids + 100)); /* same claims, new ids */
        OK(lbh_index_add(ix, 1, NULL, blob, off + 300, created + 300, tagm + 300, keys + 300 * LBH_KEY_BYTES, ids + 300));
        for (int j = 0; j < NQ; ++j) { /* the expected ids of retracted-and-re-added claims changed: recompute */
            int64_t cnt = 0;
            uint8_t kk[8 * LBH_KEY_BYTES];
            double s8[8];
            OK(lbh_index_search_shard(ix, qv + j * 64, 8, LBH_SINCE_NONE, 0, want + j * 8, s8, kk, &cnt, NULL));
            CHECK(cnt == 8 && memcmp(s8, want_sc + j * 8, sizeof s8) == 0); /* the scores did not */
        }
    }

    /* both scan operands and every batched precision give the same answer */
    for (int mode = 0; mode < 2; ++mode) {
        OK(lbh_set_option(ctx, "scan_mode", mode));
        for (int bq = 0; bq < 3; ++bq) {
            OK(lbh_set_option(ctx, "bq_mode", bq));
            uint64_t bi[NQ * 8];
            double bs[NQ * 8];
            int64_t bc[NQ];
            OK(lbh_index_search(ix, qv, NQ, 8, LBH_SINCE_NONE, 0, bi, bs, bc, NULL));
            CHECK(memcmp(bi, want, sizeof bi) == 0 && memcmp(bs, want_sc, sizeof bs) == 0);
        }
        uint64_t a[8];
        double sa[8];
        int64_t cnt = 0;
        OK(lbh_index_search(ix, qv, 1, 8, LBH_SINCE_NONE, 0, a, sa, &cnt, NULL));
        CHECK(memcmp(a, want, sizeof a) == 0);
    }
    OK(lbh_set_option(ctx, "scan_mode", 0));
    OK(lbh_set_option(ctx, "bq_mode", 2));
    CHECK(lbh_set_option(ctx, "no_such_option", 1) != LBH_OK);

    /* six reader threads (more than the index has search slots) next to one writer */
    {
        enum { NT = 6 };
        pthread_t th[NT], wt;
        reader_arg ra[NT];
        writer_arg wa = {ix, 0, 0};
        pthread_create(&wt, NULL, writer, &wa);
        for (int t = 0; t < NT; ++t) {
            ra[t] = (reader_arg){ix, qv, want, NQ, 150, t, 0};
            pthread_create(&th[t], NULL, reader, &ra[t]);
        }
        for (int t = 0; t < NT; ++t) {
            pthread_join(th[t], NULL);
            CHECK(ra[t].failures == 0);
        }
        wa.stop = 1;
        pthread_join(wt, NULL);
        CHECK(wa.rounds > 0);
        OK(lbh_index_size(ix, &live));
        CHECK(live == n);
        printf("threads: %d readers x 150 searches, writer completed %d add/retract rounds\n", NT, wa.rounds);
    }

    /* the exchange step with a one-rank communicator, and the host-side merge */
    {
        uint8_t uid[LBH_COMM_ID_BYTES];
        lbh_comm* cm = NULL;
        if (lbh_comm_unique_id(uid) == LBH_OK) {
            OK(lbh_comm_init(ctx, uid, 0, 1, &cm));
            for (int j = 0; j < NQ; ++j) {
                uint64_t g[8];
                double s8[8];
                uint8_t kk[8 * LBH_KEY_BYTES];
                int64_t cnt = 0;
                OK(lbh_index_search_sharded(ix, cm, qv + j * 64, 8, LBH_SINCE_NONE, 0, 1, 0, g, s8, kk, &cnt, NULL));
                CHECK(cnt == 8 && memcmp(g, want + j * 8, sizeof g) == 0 && memcmp(s8, want_sc + j * 8, sizeof s8) == 0);
            }
            OK(lbh_comm_destroy(cm));
        } else {
            printf("librccl not available: exchange step skipped (%s)\n", lbh_last_error());
        }
        int64_t lists[2 * 3 * 6], outl[2 * 6], oc = 0;
        memset(lists, 0, sizeof lists);
        const double s1 = 0.5, s2 = 0.25;
        lists[0] = 1; memcpy(&lists[6], &s1, 8); lists[7] = 11; lists[8] = 2;
        lists[18] = 2; memcpy(&lists[24], &s1, 8); lists[25] = 22; lists[26] = 1; memcpy(&lists[30], &s2, 8); lists[31] = 23;
        OK(lbh_merge_topk(lists, 2, 2, outl, &oc));
        CHECK(oc == 2 && outl[1] == 22 && outl[7] == 11); /* equal scores: the smaller key first */
    }

    OK(lbh_index_destroy(ix));
    OK(lbh_shutdown(ctx));
    free(blob); free(off); free(created); free(tagm); free(keys); free(ids); free(qblob); free(qoff); free(qv);
    printf("cabi_check: all checks passed (%lld claims)\n", (long long)n);
    return 0;
}
