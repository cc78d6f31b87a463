/*
 * lbhip.h -- C ABI of liblbhip.so, the MI355X (gfx950) latent-retrieval backend.
 *
 * The reference (amazedsaint/lbm) has no FFI for this path: the path is plain
 * Python (lb/latent.py, lb/context_graph.py, lb/node.py:718-734).  Each entry
 * point below therefore cites the reference *Python* interface it replaces;
 * INTEGRATION.md shows the ctypes stub a maintainer adds on the reference side.
 *
 * Conventions
 *   - every function returns 0 on success, a negative LBH_E_* code on failure
 *     and never throws; lbh_last_error() returns a thread-local message.
 *   - all array arguments are caller-owned, caller-sized, contiguous,
 *     little-endian.  Pointers are HOST pointers unless the name ends in _dev.
 *   - handles are released only through lbh_shutdown / lbh_index_destroy.
 *   - one context = one GPU (one process per GPU); an index belongs to the
 *     context that created it.  Any thread may call: searches (lbh_index_search*,
 *     lbh_index_recent, lbh_index_size) on one index run concurrently -- up to four
 *     at a time, further ones wait -- while add / retract / metadata edits are
 *     exclusive (the reference's compile_context takes no lock, its writers hold
 *     BatteryNode._lock: lb/node.py:718 vs :450).  lbh_index_search_sharded is a
 *     collective: one at a time per communicator, in the same order on every rank.
 *   - there is NO CPU fallback: without a usable gfx950 device lbh_init fails.
 */
#ifndef LBHIP_H
#define LBHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LBH_DIM 64          /* embedding width; lb/latent.py:41 default dim, the only one callers use */
#define LBH_KEY_BYTES 32    /* tie-break key: raw bytes of the 64-hex claim hash (big-endian order) */
#define LBH_MAX_PAGE 64     /* results produced per device pass; larger k is paged internally */

#define LBH_OK 0
#define LBH_E_INVALID (-1)  /* bad argument */
#define LBH_E_DEVICE (-2)   /* HIP runtime error / no gfx950 device */
#define LBH_E_NOMEM (-3)
#define LBH_E_NOTFOUND (-4) /* unknown claim id */
#define LBH_E_INTERNAL (-5)

#define LBH_SINCE_NONE INT64_MIN /* since_ms value meaning "no time filter" (Python None) */

typedef struct lbh_ctx lbh_ctx;
typedef struct lbh_index lbh_index;
typedef struct lbh_comm lbh_comm;   /* one rank's end of the claim-sharded search's exchange (RCCL communicator + buffers) */
#define LBH_COMM_ID_BYTES 128       /* size of the rendezvous id (an RCCL unique id) */

/* Per-search statistics (all optional; pass NULL to skip).  Times are HIP-event
 * times on the library's own stream. */
typedef struct lbh_stats {
    int32_t want_timing;      /* in: 1 = HIP events around the scan kernel(s) (scan_ms); 2 = also around the whole call
                                 (total_ms).  Each event record costs a few microseconds of queue time. */
    int32_t path;             /* out: 0 = fp32 scan + exact re-rank, 1 = exact fp64 scan (fallback / paging), 2 = batched MFMA */
    int64_t rows_scanned;     /* out: live rows the scoring kernel read */
    int64_t bytes_per_row;    /* out: algorithmic bytes per row for that kernel (256/264/272; 128+.. with the fp16 copy; 512+.. exact path) */
    int64_t candidates;       /* out: rows that reached the exact fp64 re-rank */
    int32_t scan_launches;    /* out: number of scoring-kernel launches this call made */
    int32_t fallbacks;        /* out: times the fp32 path overflowed into the exact path */
    float scan_ms;            /* out: total HIP-event time of the scoring kernel launches */
    float total_ms;           /* out: HIP-event time of everything enqueued by the call (want_timing >= 2, else 0) */
    float max_score_err;      /* out: largest |scan score - exact fp64 score| over the rows that were re-scored: a live
                                 check of the scan's error bound (must stay below half the candidate margin) */
    float margin;             /* out, batched searches: the candidate margin the scan kernels worked with (2 x their error bound;
                                 a constant per scan precision, derived from the index and the queries in the int8 mode) */
} lbh_stats;

/* ---- context -------------------------------------------------------------- */

/* Bind a context to devices[0] (n_devices must be 1: one process per GPU). */
int lbh_init(const int* devices, int n_devices, lbh_ctx** out_ctx);
int lbh_shutdown(lbh_ctx* ctx);
const char* lbh_last_error(void);
int lbh_version(void);
/* Name of the device the context runs on, e.g. "gfx950:sramecc+:xnack-". */
int lbh_device_name(lbh_ctx* ctx, char* buf, int buflen);

/* Tuning knobs (no reference counterpart).  name: "scan_blocks_per_cu" (1..8; 0 = by index size, the default),
 * "scan_nt" (0/1: non-temporal loads in the single-query scan), "scan_mode" (0 = scan the fp32
 * matrix, 256 B/claim, default; 1 = scan a 128 B/claim fp16 copy that the index builds on first
 * use, wider candidate margin, unit-scale data only - results are the same either way, the
 * re-rank is always exact fp64), "bq_mode" (batched scan precision: 0 fp32 MFMA, 1 split bf16,
 * 2 fp16 = default, 3 int8 on a 64 B/claim copy, margin derived per batch from the index's and the queries' scales), "scan_fuse" (0 / 1 / 2: the single-query scan's last block runs the select step itself -- never /
 * for grids of one block per CU on indexes up to 4 M claims, the default / always), "scan_xb_k" (smallest top_k for
 * which the scan's waves share a cross-block threshold at their compactions, default 24, 0 = never), "k1_memo_log2"
 * (log2 of the entries of the vectoriser's token -> digest table, default 20 = 128 MiB of HBM, 0 = off),
 * "stage_helpers" (1..8, default 2: threads that copy PAGEABLE caller buffers of lbh_index_add into its pinned staging
 * ring; pinned buffers -- lbh_host_alloc -- need none), "spin_us" (completion-ticket polling budget), "coll_timeout_ms" (longest wait for a
 * claim-sharded search's all-gather; also LBH_COLL_TIMEOUT_MS), "force_exact" (0/1, diagnostic: single-query searches rank with the exact fp64 scan only -- the
 * path that otherwise serves mass ties and result pages beyond the first), "inject_fault" (test hook, one-shot: 1 = the next
 * sharded search's local part fails on this rank, 2 = its stream is held up past the timeout).  Also read from the
 * environment at lbh_init as LBH_SCAN_BLOCKS_PER_CU / LBH_SCAN_NT / LBH_SCAN_MODE /
 * LBH_BQ_MODE / LBH_SPIN_US / LBH_SCAN_FUSE / LBH_SCAN_XB_K / LBH_K1_MEMO_LOG2; environment only: LBH_SCAN_SPLIT ("a,b,c,d": tile shares of the scan's
 * four block classes, "0": grid-stride loop). */
int lbh_set_option(lbh_ctx* ctx, const char* name, int64_t value);

/* Pinned (page-locked) host memory for the arrays handed to lbh_index_add / lbh_embed_batch (no reference counterpart:
 * an ingest-rate helper).  From pinned buffers the text, offsets, timestamps, tag masks and keys cross PCIe by
 * asynchronous DMA that overlaps the vectorising kernel; ordinary (pageable) buffers work everywhere -- lbh_index_add
 * stages them itself through a ring of pinned pieces filled by helper threads (4.1 against 3.4 ms per million claims). */
int lbh_host_alloc(lbh_ctx* ctx, int64_t bytes, void** out_ptr);
int lbh_host_free(lbh_ctx* ctx, void* ptr);

/* ---- lb/latent.py ----------------------------------------------------------- */

/* replaces lb/latent.py:41-57  embed(text, dim=64), batched.
 * utf8 + offsets[n+1] is a CSR pack of n texts; out_f64 receives n x 64 doubles,
 * bit-identical to the reference's Python floats.  Tokenisation (lb/latent.py:34-38),
 * both SHA-256 digests per token (:47-48), the in-order fp64 accumulation (:50-52)
 * and the normalisation (:54-57) all run on the GPU. */
int lbh_embed_batch(lbh_ctx* ctx, const uint8_t* utf8, const int64_t* offsets, int64_t n, double* out_f64);
/* embed(text, dim) for 1 <= dim <= 64 (lb/latent.py:41-57 with a caller-chosen dim): only the first `dim` digest
 * bytes of a token are accumulated (:50-52) and the norm runs over those `dim` components (:54).  out_f64 still
 * receives n x 64 doubles; components >= dim are 0.  dim = 64 is lbh_embed_batch. */
int lbh_embed_batch_dim(lbh_ctx* ctx, const uint8_t* utf8, const int64_t* offsets, int64_t n, int dim, double* out_f64);
/* Same as lbh_embed_batch, device pointers in and out (text already resident in HBM); out_f32_dev may be NULL. */
int lbh_embed_batch_dev(lbh_ctx* ctx, const uint8_t* utf8_dev, const int64_t* offsets_dev, int64_t n,
                        int64_t total_bytes, double* out_f64_dev, float* out_f32_dev);

/* replaces lb/latent.py:60-63  cosine(a, b): sequential fp64 dot product, computed on
 * the GPU.  Returns LBH_E_INVALID (message "dimension mismatch") when na != nb,
 * which the Python mirror turns into the reference's ValueError. */
int lbh_cosine(lbh_ctx* ctx, const double* a, int64_t na, const double* b, int64_t nb, double* out);

/* ---- lb/context_graph.py ---------------------------------------------------- */

/* replaces ContextGraph.__init__ (lb/context_graph.py:56-58): an empty claim index in HBM. */
int lbh_index_create(lbh_ctx* ctx, lbh_index** out_idx);
int lbh_index_destroy(lbh_index* idx);
/* Pre-size the HBM arrays for `capacity` claims (optional; add grows geometrically). */
int lbh_index_reserve(lbh_index* idx, int64_t capacity);

/* replaces ContextGraph.add_claim (lb/context_graph.py:60-82) / from_snapshot's re-embedding
 * loop (:142-147), batched.  Exactly one of (utf8, offsets) or emb_f64 supplies the vectors:
 * texts are embedded on the GPU (same kernel as lbh_embed_batch); emb_f64 is n x 64 doubles.
 * created_ms[n]; tagmask[n] may be NULL (= 0); hashkey is n x 32 bytes compared as big-endian
 * integers for the (-score, claim_hash) tie-break of :120.  out_ids[n] receives stable claim ids. */
int lbh_index_add(lbh_index* idx, int64_t n, const double* emb_f64, const uint8_t* utf8, const int64_t* offsets,
                  const int64_t* created_ms, const uint64_t* tagmask, const uint8_t* hashkey, uint64_t* out_ids);

/* replaces ContextGraph.retract (lb/context_graph.py:84-89): the row is removed from the
 * scanned arrays (the reference drops the embedding and skips the claim, :89,:107). */
int lbh_index_retract(lbh_index* idx, uint64_t id);
/* Batched form of lbh_index_retract: what a loader or a reconciling mirror issues after many ContextGraph.retract
 * calls (lb/context_graph.py:84-89 in a loop).  The n rows leave the scanned arrays in ONE compaction kernel
 * (surviving rows of the cut-off tail fill the holes).  Unknown or repeated ids fail the call before anything changes. */
int lbh_index_retract_many(lbh_index* idx, int64_t n, const uint64_t* ids);
/* Rewrites the per-claim metadata that compile() filters on (a caller mutated Claim.created_ms / tags). */
int lbh_index_update_meta(lbh_index* idx, uint64_t id, int64_t created_ms, uint64_t tagmask, const uint8_t* hashkey);
/* Bulk rewrite of the tie-break keys of n live claims (ids[n], hashkey n x 32 bytes).  Used by the
 * Python mirror when claim hashes are not 64-hex strings: it then ranks the hash strings with
 * Python's own str order (what lb/context_graph.py:120 compares) and stores the ranks as keys. */
int lbh_index_set_keys(lbh_index* idx, int64_t n, const uint64_t* ids, const uint8_t* hashkey);
int lbh_index_size(lbh_index* idx, int64_t* out_live);
/* Copies the cached fp64 embedding of a claim (the reference's _embedding_cache[h]). */
int lbh_index_get_embedding(lbh_index* idx, uint64_t id, double* out64);
/* Same for n claims in one gather kernel + one copy: out_f64 receives n x 64 doubles. */
int lbh_index_get_embeddings(lbh_index* idx, int64_t n, const uint64_t* ids, double* out_f64);

/* replaces the scoring loop + sort + cut of ContextGraph.compile (lb/context_graph.py:104-121)
 * for nq query vectors (q_f64 = nq x 64 doubles, e.g. from lbh_embed_batch).
 *   k         max(0, int(top_k)) of :121; any value (paged internally beyond LBH_MAX_PAGE)
 *   since_ms  LBH_SINCE_NONE = None; otherwise rows with created_ms < since_ms are skipped (:110)
 *   tag_mask  0 = no tag filter (reference behaviour); else rows need (tagmask & tag_mask) != 0
 * Outputs, per query q: out_counts[q] results in out_ids[q*k ..] / out_scores[q*k ..], ordered
 * exactly like the reference: score descending, then hash key ascending.  Scores are the
 * reference's fp64 values bit-for-bit.  stats may be NULL.
 * Three or more queries with 1 <= k <= 256 (<= 16 unless the scan runs in its default fp16 mode on
 * unit-scale vectors; above 64 the ranking step runs once per 64 results) share ONE pass over the claim matrix on the matrix cores, 256 queries per
 * launch; results do not depend on which path served a query. */
int lbh_index_search(lbh_index* idx, const double* q_f64, int64_t nq, int64_t k, int64_t since_ms,
                     uint64_t tag_mask, uint64_t* out_ids, double* out_scores, int64_t* out_counts,
                     lbh_stats* stats);

/* Convenience: embed(query) (:104) and the search in one call, one round trip. */
int lbh_index_search_text(lbh_index* idx, const uint8_t* utf8, int64_t len, int64_t k, int64_t since_ms,
                          uint64_t tag_mask, uint64_t* out_ids, double* out_scores, int64_t* out_count,
                          double* out_qvec64, lbh_stats* stats);

/* Local top-k for a claim-sharded corpus (one shard per GPU/rank): like lbh_index_search for one
 * query, and additionally returns the 32-byte hash keys so that ranks can merge their lists with
 * the same (-score, key) comparator after an all-gather. */
int lbh_index_search_shard(lbh_index* idx, const double* q_f64, int64_t k, int64_t since_ms, uint64_t tag_mask,
                           uint64_t* out_ids, double* out_scores, uint8_t* out_keys, int64_t* out_count,
                           lbh_stats* stats);

/* No counterpart in the reference.  For callers that cannot allocate device memory -- tight serving loops, harnesses that
 * forbid hipMalloc, code that runs between the captures of a HIP graph (the call itself cannot be captured: it runs on the
 * library's own stream and ends in a hipStreamSynchronize): a variant of lbh_index_search_shard that calls no hipMalloc /
 * hipFree / allocating routine at all.  The call is NOT ordered against any other stream: inputs must be complete, and
 * the output buffers idle, on every stream of the caller's before it is made.  The caller owns the claim arrays (device pointers, the index's layout: emb32 = n x 64
 * floats, exactly (float) of emb64 = n x 64 doubles; created_ms / tagmask = n words, needed only when the filter is
 * used; keywords = n x 4 uint64, word w = the big-endian value of key bytes 8w..8w+7) and the outputs (device
 * pointers: out_rows = row indices, out_scores, each k entries; *out_count = number of results, or -1 if more than 2048
 * rows tie on score AND key).  Everything the kernels need beyond that -- candidate lists, counters, the result block,
 * LBH_NOALLOC_SCRATCH_BYTES in all -- lives in ONE file-scope `static __device__` array of the library, of which every
 * GPU holds its own copy for the life of the process.  max_row_norm: an upper bound on the row norms (1.0 for vectors
 * from lbh_embed_batch); the candidate margin scales with it.  Limits: n <= LBH_NOALLOC_MAX_ROWS, k <= LBH_MAX_PAGE
 * (larger values are rejected, not paged).  The query travels as a kernel argument; the call enqueues its kernels on
 * the context's stream and returns after one hipStreamSynchronize.  Calls on one GPU share that GPU's array: they are
 * serialised per context, and two contexts (or processes' threads) on the same GPU must not overlap such calls. */
#define LBH_NOALLOC_SCRATCH_BYTES 1445376
#define LBH_NOALLOC_MAX_ROWS 4294967280LL
int lbh_search_dev_noalloc(lbh_ctx* ctx, const float* emb32_dev, const double* emb64_dev, const int64_t* created_ms_dev,
                           const uint64_t* tagmask_dev, const uint64_t* keywords_dev, int64_t n, double max_row_norm,
                           const double* q_f64, int64_t k, int64_t since_ms, uint64_t tag_mask, int64_t* out_rows_dev,
                           double* out_scores_dev, int64_t* out_count_dev);

/* ---- claim-sharded search over the GPUs of a node: one process per GPU, one index (shard) per process.  The reference
 * ranks by a total order on (-score, claim_hash) with unique hashes (lb/context_graph.py:120) and a claim's score depends
 * on that claim and the query alone, so the global top-k is inside the union of the shards' top-k lists.
 *
 * lbh_comm_unique_id   rank 0 draws the rendezvous id and hands it to the other ranks by any channel it likes (a file,
 *                      MPI, torch.distributed); needs librccl (opened on first use), no GPU work.
 * lbh_comm_init        every rank: joins the communicator (collective: returns when all `world` ranks have called it).
 * lbh_index_search_sharded   every rank, same query / k / filters (collective): exact local top-k on this rank's shard
 *                      (the kernels of lbh_index_search_shard), ONE all-gather of the ranks' lists -- (k + 1) x 48 bytes per
 *                      rank, read from the device memory the final kernel wrote them to, RCCL over xGMI on the library's
 *                      stream -- and the merge with the reference's comparator (lbh_merge_topk).  Every rank returns the
 *                      same global result: out_gids[i] = local claim id * gid_mul + gid_add of the owning rank (e.g.
 *                      gid_mul = world, gid_add = rank for claims dealt round-robin), fp64 scores bit-identical to the
 *                      reference's, the 32-byte hash keys (out_keys may be NULL).  top_k <= LBH_MAX_PAGE takes one host
 *                      wait per query; larger k is paged locally first and exchanged through staging buffers. */
int lbh_comm_unique_id(uint8_t* out_id /* LBH_COMM_ID_BYTES */);
int lbh_comm_init(lbh_ctx* ctx, const uint8_t* id, int rank, int world, lbh_comm** out_comm);
int lbh_comm_destroy(lbh_comm* comm);
int lbh_index_search_sharded(lbh_index* idx, lbh_comm* comm, const double* q_f64, int64_t k, int64_t since_ms,
                             uint64_t tag_mask, uint64_t gid_mul, uint64_t gid_add, uint64_t* out_gids, double* out_scores,
                             uint8_t* out_keys, int64_t* out_count, lbh_stats* stats);

/* Batched form (config 3 x config 4: many queries over a claim-sharded corpus; the reference's semantics per query are
 * lb/context_graph.py:104-121, as for lbh_index_search).  Every rank, same queries / k / filters (collective): the local
 * part is lbh_index_search on this rank's shard -- three or more queries share ONE pass over the shard on the matrix
 * cores -- then ONE all-gather of the nq lists of (k + 1) x 48 bytes per rank and nq merges with the reference's
 * comparator.  Outputs as lbh_index_search (row q at q * k), global ids as lbh_index_search_sharded, out_keys = nq x k x 32
 * bytes or NULL.  Any k (the local search pages internally); nq x (k + 1) x 48 bytes <= 1 GiB.
 *
 * Failure containment (both sharded searches).  A rank whose LOCAL search fails still takes part in the all-gather: its
 * list's count word carries the negative error code, and every rank returns an error for that call -- nobody is left
 * waiting.  A rank that cannot reach the all-gather at all (it died) is noticed by its peers' bounded wait: after
 * "coll_timeout_ms" (lbh_set_option / LBH_COLL_TIMEOUT_MS, default 60 000) the call fails with LBH_E_DEVICE and the
 * communicator refuses further calls; the process should then exit (its stream still holds the unfinished collective). */
int lbh_index_search_sharded_many(lbh_index* idx, lbh_comm* comm, const double* q_f64, int64_t nq, int64_t k, int64_t since_ms,
                                  uint64_t tag_mask, uint64_t gid_mul, uint64_t gid_add, uint64_t* out_gids, double* out_scores,
                                  uint8_t* out_keys, int64_t* out_counts, lbh_stats* stats);

/* Plain all-gather of `bytes` bytes per rank through the communicator (host buffers in and out; recv = world x bytes,
 * rank-major): what the claim-sharded graph uses to hand the winners' text and tags (lb/context_graph.py:122-133 formats
 * them) from the owning rank to every rank.  Collective; same bounded wait as the searches. */
int lbh_comm_allgather(lbh_comm* comm, const void* send, int64_t bytes, void* recv);

/* Merge step of the claim-sharded search (host code, needs no GPU): `lists` = n_lists blocks of (k + 1) x 6
 * int64, as exchanged by the all-gather -- row 0 word 0 = entry count (<= k), rows 1.. = {fp64 score bits,
 * global claim id, 4 big-endian key words}.  Writes the global top-k entries (6 words each) in the
 * reference's order (lb/context_graph.py:120: score descending, hash key ascending) to out_entries. */
int lbh_merge_topk(const int64_t* lists, int64_t n_lists, int64_t k, int64_t* out_entries, int64_t* out_count);

/* ---- next row N3: recency query ------------------------------ */

/* replaces the loop + sort + slice of BatteryNode.get_recent_claims (lb/node.py:750-762): live claims
 * with created_ms >= since_ms (LBH_SINCE_NONE = no bound), newest first; claims that share a
 * created_ms come in ascending `order` (the reference's list.sort is stable, so ties keep the claims
 * dict's insertion order).  limit = max(0, limit) of the reference's claims[:limit]; any value (paged
 * internally beyond LBH_MAX_PAGE).  out_ids / out_created (may be NULL) receive out_count entries;
 * out_matched (may be NULL) receives the number of live claims that pass the time bound, whatever the
 * limit.  stats->path is 3, bytes_per_row 8. */
int lbh_index_recent(lbh_index* idx, int64_t since_ms, int64_t limit, uint64_t* out_ids, int64_t* out_created,
                     int64_t* out_count, int64_t* out_matched, lbh_stats* stats);

/* Insertion-order keys used to break created_ms ties in lbh_index_recent.  A claim's order defaults
 * to its claim id (the order of lbh_index_add calls); a host whose `claims` dict order differs
 * (re-added hashes keep their old dict position, lb/context_graph.py:72) overrides it here. */
int lbh_index_set_order(lbh_index* idx, int64_t n, const uint64_t* ids, const uint64_t* order);

#ifdef __cplusplus
}
#endif
#endif /* LBHIP_H */
