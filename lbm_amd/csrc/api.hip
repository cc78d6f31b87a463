// api.hip -- host side of liblbhip.so: context, HBM-resident claim index, and the extern "C" entry
// points declared in include/lbhip.h.  No compute happens here: every number the caller sees was
// produced by a kernel in embed.hip / search.hip.  There is no CPU fallback.
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <ctime>
#include <dlfcn.h>
#include <exception>
#include <condition_variable>
#include <mutex>
#include <shared_mutex>
#include <new>
#include <string>
#include <system_error>
#include <thread>
#include <atomic>
#include <memory>
#include <limits>
#include <vector>

#include "lbh_internal.h"

// No exception may cross the C ABI: entry points that grow host containers are function-try-blocks ending in this.
#define LBH_CATCH_ALLOC catch (const std::bad_alloc&) { return fail(LBH_E_NOMEM, "out of host memory"); } \
    catch (const std::exception& e) { return fail(LBH_E_INTERNAL, "%s", e.what()); }

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                             \
    do {                                                                                          \
        hipError_t e__ = (expr);                                                                  \
        if (e__ != hipSuccess)                                                                    \
            return fail(e__ == hipErrorOutOfMemory ? LBH_E_NOMEM : LBH_E_DEVICE, "%s: %s", #expr, \
                        hipGetErrorString(e__));                                                  \
    } while (0)

}  // namespace

#define LBH_QUERY_ZERO_COPY_MAX 8192  // lbh_index_search_text: queries up to this many bytes are read by K1 straight from mapped host memory
#define LBH_SMALL_ADD 256                     // lbh_index_add: text batches up to this many claims (one block of small_meta_kernel) ...
#define LBH_SMALL_ADD_TEXT 40960              // ... and this many text bytes take the two-launch path (64 KiB staging block)
#define LBH_STAGE_NBUF 12                     // pinned pieces of the ingest staging ring
#define LBH_STAGE_PIECE ((size_t)8 << 20)     // bytes per piece
#define LBH_STAGE_HELPERS 8                   // most threads copying the caller's pageable buffers into the ring (option "stage_helpers")
#define LBH_ADD_SLICES 4  // most slices one lbh_index_add call is cut into

struct lbh_ctx {
    int device = 0;
    int n_cu = 256;
    hipStream_t stream = nullptr;
    hipStream_t copy_stream = nullptr;            // H2D copies of lbh_index_add's text slices (overlap with K1 on `stream`)
    // staging ring of lbh_index_add for PAGEABLE caller buffers (StagedCopier): pinned pieces filled by helper threads
    std::mutex stage_mu;
    void* stage_buf[LBH_STAGE_NBUF] = {};
    hipEvent_t stage_ev[LBH_STAGE_NBUF] = {};
    hipEvent_t copy_ev[LBH_ADD_SLICES + 1] = {};  // one per slice of text + one for the metadata columns
    std::string arch;
    int scan_blocks_per_cu = 0;  // tuning knobs (env LBH_SCAN_BLOCKS_PER_CU / LBH_SCAN_NT); 0 = by index size (lbh_scan_grid)
    int scan_nt = 1;
    int scan_mode = 0;   // single-query scan: 0 = fp32 matrix (256 B/claim), 1 = fp16 copy (128 B/claim, margin 2.2e-3; opt-in)
    int bq_mode = 2;     // batched scan: 0 = exact fp32 MFMA, 1 = split-bf16 products (3 MFMAs at 16x the fp32 rate),
                         // 2 = one fp16 product (a third of the matrix work, margin 2.2e-3)
    int spin_us = 5000;  // poll the completion ticket this long before blocking in hipStreamSynchronize (0 = always block)
    int coll_timeout_ms = 60000;  // claim-sharded searches: longest wait for the all-gather (a peer that never arrives) before the call fails
    // K1's token memo (embed.hip): token -> its two SHA-256 digests, kept for the life of the context
    int memo_log2 = 20;            // option "k1_memo_log2": 2^n entries of 128 B (default 2^20 = 128 MiB; 0 = off); allocated at the first large batch
    uint64_t* memo_table = nullptr;
    uint32_t memo_mask = 0;
    uint32_t memo_epoch = 0;       // launches that used the table so far (a launch trusts only entries of earlier ones)
    int stage_helpers = 2; // option "stage_helpers": threads that copy pageable caller buffers into the pinned staging ring (1..8)
    int add_cuts = 0;      // option "add_cuts" (experiments): cuts of lbh_index_add's four text slices, a * 10000 + b * 100 + c sixteenths; 0 = built-in
    int scan_xb_k = 24;    // option "scan_xb_k": smallest top_k for which the scan's waves share a cross-block bound at their compactions (0 = never)
    int scan_fuse = 1;     // option "scan_fuse": the scan's last block runs the select step itself -- 0 never, 1 = when the grid is one block per CU (default), 2 always
    int force_exact = 0;   // option "force_exact" (diagnostic / parity tests): single-query searches skip the fp32 scan and rank with the exact fp64 scan
    int inject_fault = 0;  // TEST HOOK (option "inject_fault", one-shot): 1 = this rank's local search of the next sharded call fails;
                           // 2 = the next sharded call's stream is held up for 4 x coll_timeout_ms by a sleeping host function
    void* noalloc_scratch = nullptr;  // THIS device's address of search.hip's file-scope scratch array (lbh_search_dev_noalloc)
    std::mutex noalloc_mu;            // calls that share the array on one GPU would race: serialised per context
    std::mutex mu;  // serialises use of the scratch below
    // small scratch for embed / cosine calls
    uint8_t* d_text = nullptr;
    int64_t d_text_cap = 0;
    int64_t* d_off = nullptr;
    int64_t d_off_cap = 0;
    double* d_vec = nullptr;  // embed output / cosine operands
    int64_t d_vec_cap = 0;
};

#define LBH_SLOTS 4  // searches that may run on one index at the same time (each has its own workspace and stream)

// Everything ONE search in flight needs besides the index's arrays: workspace, result block, query, events, stream.
struct lbh_slot {
    lbh_workspace ws{};
    lbh_qarg qarg{};              // current query when it came from the host (kernel argument)
    const double* qptr = nullptr;  // current query when it was embedded on the GPU (ws.q64), else NULL
    double qnorm = 1.0;           // |q| of the current query (1 for a query embedded from text)
    int32_t seq = 0;              // completion ticket of the last single-query launch
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    hipStream_t stream = nullptr;  // slot 0: the context's stream; further slots: their own
    bool own_stream = false;
    void* staging = nullptr;      // pinned host staging of the query text (device-mapped, coherent: K1 reads a short query from it directly)
    void* staging_dev = nullptr;  // the same memory as the device sees it
    int64_t staging_cap = 0;
    uint8_t* d_text = nullptr;    // query text on the device (search_text)
    int64_t d_text_cap = 0;
    int64_t* d_off = nullptr;     // its two offsets
    uint64_t seen_gen = 0;        // last mutation this slot's own stream was ordered behind (lbh_index::mut_gen)
    bool busy = false;
};

struct lbh_index {
    lbh_ctx* ctx = nullptr;
    // Readers and writers (the reference's compile_context takes no lock while publishers hold BatteryNode._lock,
    // lb/node.py:718 vs :450): searches hold `rw` shared and each leases one slot, so up to LBH_SLOTS of them overlap on
    // one index; add / retract / metadata edits hold it exclusively and leave their stream drained before they release it.
    std::shared_mutex rw;
    std::mutex slot_mu;
    std::condition_variable slot_cv;
    lbh_slot* slots[LBH_SLOTS] = {};
    int64_t n = 0, cap = 0;
    float* emb32 = nullptr;
    double* emb64 = nullptr;
    int64_t* created = nullptr;
    uint64_t* tags = nullptr;
    uint64_t* keys = nullptr;
    uint64_t* order = nullptr;    // per-row insertion-order key of the recency query (defaults to the claim id)
    uint64_t* ids = nullptr;      // per-row claim id (device mirror of id_of_row: the sharded search emits global ids on the GPU)
    uint16_t* emb16 = nullptr;    // fp16 copy in MFMA-fragment order (lbh_emb16_off), built on first use; rows [0, emb16_rows) are valid
    int64_t emb16_rows = 0, emb16_cap = 0;
    void* small_stage = nullptr;      // device-mapped pinned block for small text batches (lbh_index_add, n <= LBH_SMALL_ADD)
    void* small_stage_dev = nullptr;
    int8_t* emb8 = nullptr;       // int8 copy in MFMA-fragment order (lbh_emb8_off) for the int8 batched scan; rows [0, emb8_rows) are valid
    int64_t emb8_rows = 0, emb8_cap = 0;
    double emb8_scale = 0.0;      // x ~ emb8_scale * X for every element of every converted row (|x| <= 127 emb8_scale)
    uint32_t* emb8_absmax = nullptr;  // device word: float bits of the largest |element| of the rows being converted
    std::vector<uint64_t> id_of_row;
    std::vector<int64_t> row_of_id;
    double max_norm = 0.0;    // largest |x| ever added: the scans' error bounds (hence their candidate margins) scale with it
    void* staging = nullptr;  // pinned host staging of the mutators (held under the exclusive lock)
    int64_t staging_cap = 0;
    void* dscratch = nullptr;  // device scratch of the batched mutators (row lists, pairs, gathered rows); grow-only
    int64_t dscratch_cap = 0;
    hipEvent_t mut_ev = nullptr;  // recorded on the context's stream by every mutator: a slot with a stream of its own waits on it
    uint64_t mut_gen = 0;
};

namespace {

int ensure_ctx_scratch(lbh_ctx* c, int64_t text_bytes, int64_t n_off, int64_t n_vec) {
    HIP_TRY(hipSetDevice(c->device));
    if (text_bytes > c->d_text_cap) {
        if (c->d_text) (void)hipFree(c->d_text);
        c->d_text = nullptr;
        int64_t cap = std::max<int64_t>(text_bytes + text_bytes / 8, 1 << 16);  // headroom: the next batch of about this size must not re-allocate
        HIP_TRY(hipMalloc((void**)&c->d_text, (size_t)cap));
        c->d_text_cap = cap;
    }
    if (n_off > c->d_off_cap) {
        if (c->d_off) (void)hipFree(c->d_off);
        c->d_off = nullptr;
        int64_t cap = std::max<int64_t>(n_off, 1024);
        HIP_TRY(hipMalloc((void**)&c->d_off, (size_t)cap * 8));
        c->d_off_cap = cap;
    }
    if (n_vec > c->d_vec_cap) {
        if (c->d_vec) (void)hipFree(c->d_vec);
        c->d_vec = nullptr;
        int64_t cap = std::max<int64_t>(n_vec, 1024);
        HIP_TRY(hipMalloc((void**)&c->d_vec, (size_t)cap * 8));
        c->d_vec_cap = cap;
    }
    return LBH_OK;
}

// The memo argument of the next K1 launch on the context's stream (callers hold c->mu or the index's exclusive lock:
// launches that use the table are stream-ordered one behind the other).  Small batches -- a query's text, a single
// publish -- go without it: they could overlap a bulk launch on another stream, and have nothing to gain.
constexpr int64_t LBH_MEMO_MIN_BATCH = 2048;
lbh_memo next_memo(lbh_ctx* c, int64_t n) {
    lbh_memo none{nullptr, 0, 0};
    if (c->memo_log2 <= 0 || n < LBH_MEMO_MIN_BATCH) return none;
    if (!c->memo_table) {
        const int lg = std::max(8, std::min(24, c->memo_log2));
        const size_t bytes = ((size_t)1 << lg) * 128;
        uint64_t* t = nullptr;
        if (hipMalloc((void**)&t, bytes) != hipSuccess) {  // no room: vectorise without it
            (void)hipGetLastError();
            c->memo_log2 = 0;
            return none;
        }
        if (hipMemsetAsync(t, 0, bytes, c->stream) != hipSuccess) {
            (void)hipFree(t);
            c->memo_log2 = 0;
            return none;
        }
        c->memo_table = t;
        c->memo_mask = (uint32_t)(((size_t)1 << lg) - 1);
        c->memo_epoch = 0;
    }
    if (c->memo_epoch >= (1u << 24) - 2) {  // the tag's epoch field is about to wrap: start over with an empty table
        if (hipMemsetAsync(c->memo_table, 0, ((size_t)c->memo_mask + 1) * 128, c->stream) != hipSuccess) return none;
        c->memo_epoch = 0;
    }
    c->memo_epoch += 1;
    return lbh_memo{c->memo_table, c->memo_mask, c->memo_epoch};
}

template <class T>
int grow_array(T*& p, int64_t old_n, int64_t new_cap, int64_t per_row, hipStream_t s) {
    T* q = nullptr;
    HIP_TRY(hipMalloc((void**)&q, (size_t)new_cap * per_row * sizeof(T)));
    if (p && old_n > 0) HIP_TRY(hipMemcpyAsync(q, p, (size_t)old_n * per_row * sizeof(T), hipMemcpyDeviceToDevice, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (p) (void)hipFree(p);
    p = q;
    return LBH_OK;
}

int index_reserve_locked(lbh_index* ix, int64_t want) {
    if (want <= ix->cap) return LBH_OK;
    if (want >= (int64_t)0xFFFFFFF0ll) return fail(LBH_E_INVALID, "index capacity %lld exceeds 2^32 rows per shard", (long long)want);
    HIP_TRY(hipSetDevice(ix->ctx->device));
    hipStream_t s = ix->ctx->stream;
    int rc;
    if ((rc = grow_array(ix->emb32, ix->n, want, 64, s))) return rc;
    if ((rc = grow_array(ix->emb64, ix->n, want, 64, s))) return rc;
    if ((rc = grow_array(ix->created, ix->n, want, 1, s))) return rc;
    if ((rc = grow_array(ix->tags, ix->n, want, 1, s))) return rc;
    if ((rc = grow_array(ix->keys, ix->n, want, 4, s))) return rc;
    if ((rc = grow_array(ix->order, ix->n, want, 1, s))) return rc;
    if ((rc = grow_array(ix->ids, ix->n, want, 1, s))) return rc;
    if (ix->emb16) {  // rebuilt lazily at the next fp16 scan
        (void)hipFree(ix->emb16);
        ix->emb16 = nullptr;
        ix->emb16_rows = ix->emb16_cap = 0;
    }
    if (ix->emb8) {  // likewise
        (void)hipFree(ix->emb8);
        ix->emb8 = nullptr;
        ix->emb8_rows = ix->emb8_cap = 0;
    }
    try {  // the host-side id maps grow with the device columns, so that no ingest call pays a vector re-allocation
        // ... nor the first-touch page faults of the reserved space (a million new ids = 4 000 faults = 2-3 ms of the call):
        // the space is touched here, once, by growing and shrinking the vector
        auto reserve_touched = [](auto& v, size_t cap) {
            if (v.capacity() >= cap) return;
            const size_t sz = v.size();
            v.resize(cap);
            v.resize(sz);
        };
        reserve_touched(ix->id_of_row, (size_t)want);
        if (ix->row_of_id.capacity() < ix->row_of_id.size() + (size_t)(want - ix->n))
            reserve_touched(ix->row_of_id, std::max(ix->row_of_id.size() + (size_t)(want - ix->n), 2 * ix->row_of_id.capacity()));
    } catch (const std::bad_alloc&) {
        return fail(LBH_E_NOMEM, "out of host memory");
    }
    ix->cap = want;
    return LBH_OK;
}

int ensure_staging(lbh_index* ix, int64_t bytes) {
    if (bytes <= ix->staging_cap) return LBH_OK;
    if (ix->staging) (void)hipHostFree(ix->staging);
    ix->staging = nullptr;
    int64_t cap = std::max<int64_t>(bytes, 1 << 20);
    HIP_TRY(hipHostMalloc(&ix->staging, (size_t)cap, hipHostMallocDefault));
    ix->staging_cap = cap;
    return LBH_OK;
}

int ensure_dscratch(lbh_index* ix, int64_t bytes) {
    if (bytes <= ix->dscratch_cap) return LBH_OK;
    if (ix->dscratch) (void)hipFree(ix->dscratch);
    ix->dscratch = nullptr;
    ix->dscratch_cap = 0;
    const int64_t cap = std::max<int64_t>(bytes + bytes / 2, 1 << 16);
    HIP_TRY(hipMalloc(&ix->dscratch, (size_t)cap));
    ix->dscratch_cap = cap;
    return LBH_OK;
}

// ---- slots and locks -------------------------------------------------------------------------------------------
constexpr int LBH_NEED_EMB8 = -1001;   // ... the int8 copy
constexpr int LBH_NEED_EMB16 = -1000;  // internal: a search found the fp16 copy out of date; rebuilt under the exclusive lock, then retried

int slot_create(lbh_index* ix, int which, lbh_slot** out) {
    lbh_slot* sl = new (std::nothrow) lbh_slot();
    if (!sl) return fail(LBH_E_NOMEM, "out of host memory");
    *out = sl;  // the caller destroys it on failure
    if (which == 0) sl->stream = ix->ctx->stream;
    else {
        HIP_TRY(hipStreamCreateWithFlags(&sl->stream, hipStreamNonBlocking));
        sl->own_stream = true;
    }
    lbh_workspace& w = sl->ws;
    HIP_TRY(hipMalloc((void**)&w.q64, 512));
    HIP_TRY(hipMalloc((void**)&w.cand, (size_t)LBH_MAX_BLOCKS * LBH_LIST_CAP * 8));
    HIP_TRY(hipMalloc((void**)&w.xcand_ord, (size_t)LBH_MAX_BLOCKS * LBH_SCAN_WAVES * LBH_MAX_PAGE * 8));
    HIP_TRY(hipMalloc((void**)&w.xcand_row, (size_t)LBH_MAX_BLOCKS * LBH_SCAN_WAVES * LBH_MAX_PAGE * 4));
    HIP_TRY(hipMalloc((void**)&w.flags, 512));
    HIP_TRY(hipMalloc((void**)&w.bmax, (size_t)LBH_MAX_BLOCKS * 4 + (size_t)LBH_XB_SLOTS * 8));
    HIP_TRY(hipMemset(w.bmax, 0, (size_t)LBH_MAX_BLOCKS * 4 + (size_t)LBH_XB_SLOTS * 8));  // launch numbers start at 1: a zero slot is empty
    HIP_TRY(hipMalloc((void**)&sl->d_off, 16));
    HIP_TRY(hipHostMalloc((void**)&w.result_host, sizeof(lbh_result), hipHostMallocMapped | hipHostMallocCoherent));
    HIP_TRY(hipHostGetDevicePointer((void**)&w.result_dev, w.result_host, 0));
    memset(w.result_host, 0, sizeof(lbh_result));
    HIP_TRY(hipMemset(w.flags, 0, 512));
    for (auto& ev : sl->ev) HIP_TRY(hipEventCreate(&ev));
    return LBH_OK;
}

void slot_destroy(lbh_slot* sl) {
    if (!sl) return;
    if (sl->stream) (void)hipStreamSynchronize(sl->stream);
    lbh_workspace& w = sl->ws;
    void* dev[] = {w.rc_a, w.rc_b, w.rc_row, w.rc_cnt, w.q64, w.cand, w.xcand_ord, w.xcand_row, w.flags, w.bmax,
                   w.bq_qmat, w.bq_cand, w.bq_total, w.bq_seed, w.bq_bfrag, w.bq_qscale, sl->d_text, sl->d_off};
    for (void* p : dev)
        if (p) (void)hipFree(p);
    for (auto& ev : sl->ev)
        if (ev) (void)hipEventDestroy(ev);
    if (w.result_host) (void)hipHostFree(w.result_host);
    if (w.bq_result_host) (void)hipHostFree(w.bq_result_host);
    if (w.bq_qhost) (void)hipHostFree(w.bq_qhost);
    if (w.rc_result_host) (void)hipHostFree(w.rc_result_host);
    if (sl->staging) (void)hipHostFree(sl->staging);
    if (sl->own_stream && sl->stream) (void)hipStreamDestroy(sl->stream);
    delete sl;
}

// query text of lbh_index_search_text: pinned staging + device copy, per slot (two searches in flight must not share them)
int slot_text_buffers(lbh_slot* sl, int64_t bytes) {
    if (bytes > sl->staging_cap) {
        if (sl->staging) (void)hipHostFree(sl->staging);
        sl->staging = sl->staging_dev = nullptr;
        sl->staging_cap = 0;
        const int64_t cap = std::max<int64_t>(bytes, 1 << 16);
        HIP_TRY(hipHostMalloc(&sl->staging, (size_t)cap, hipHostMallocMapped | hipHostMallocCoherent));
        HIP_TRY(hipHostGetDevicePointer(&sl->staging_dev, sl->staging, 0));
        sl->staging_cap = cap;
    }
    if (bytes > sl->d_text_cap) {
        if (sl->d_text) (void)hipFree(sl->d_text);
        sl->d_text = nullptr;
        sl->d_text_cap = 0;
        const int64_t cap = std::max<int64_t>(bytes, 1 << 16);
        HIP_TRY(hipMalloc((void**)&sl->d_text, (size_t)cap));
        sl->d_text_cap = cap;
    }
    return LBH_OK;
}

// A reader's hold on the index: the shared side of `rw` plus one slot.  Slot 0 exists from lbh_index_create; the others
// are made the first time two searches overlap (their workspaces cost device memory only then).
class Lease {
  public:
    explicit Lease(lbh_index* ix) : ix_(ix), lk_(ix->rw) {}
    Lease(const Lease&) = delete;
    Lease& operator=(const Lease&) = delete;
    ~Lease() {
        if (!sl_) return;
        {
            std::lock_guard<std::mutex> g(ix_->slot_mu);
            sl_->busy = false;
        }
        ix_->slot_cv.notify_one();
    }
    int acquire() {
        std::unique_lock<std::mutex> g(ix_->slot_mu);
        for (;;) {
            int vacant = -1;
            for (int i = 0; i < LBH_SLOTS; ++i) {
                lbh_slot* sl = ix_->slots[i];
                if (sl && !sl->busy) {
                    sl->busy = true;
                    sl_ = sl;
                    break;
                }
                if (!sl && vacant < 0) vacant = i;
            }
            if (sl_) break;
            if (vacant >= 0) {
                lbh_slot* sl = nullptr;
                const int rc = slot_create(ix_, vacant, &sl);
                if (rc) {
                    slot_destroy(sl);
                    return rc;
                }
                sl->busy = true;
                ix_->slots[vacant] = sl_ = sl;
                break;
            }
            ix_->slot_cv.wait(g);
        }
        g.unlock();
        // a slot with a stream of its own runs behind the last mutation (mutators work on the context's stream)
        if (sl_->own_stream && sl_->seen_gen != ix_->mut_gen) {
            HIP_TRY(hipStreamWaitEvent(sl_->stream, ix_->mut_ev, 0));
            sl_->seen_gen = ix_->mut_gen;
        }
        return LBH_OK;
    }
    lbh_slot* slot() const { return sl_; }

  private:
    lbh_index* ix_;
    std::shared_lock<std::shared_mutex> lk_;
    lbh_slot* sl_ = nullptr;
};

// every mutator (and the bulk readers that borrow the index's own staging) ends with this, still under the exclusive
// lock: searches on a slot with a stream of its own order themselves behind the event
int mutation_done(lbh_index* ix) {
    ix->mut_gen += 1;
    HIP_TRY(hipSetDevice(ix->ctx->device));
    HIP_TRY(hipEventRecord(ix->mut_ev, ix->ctx->stream));
    return LBH_OK;
}

struct Exclusive {
    lbh_index* ix;
    std::unique_lock<std::shared_mutex> lk;
    explicit Exclusive(lbh_index* i) : ix(i), lk(i->rw) {}
    ~Exclusive() {
        const std::string keep = g_err;  // the event record must not replace the caller's error text
        if (mutation_done(ix) != LBH_OK) g_err = keep;
    }
};

int ensure_emb16(lbh_index* ix);
int ensure_emb8(lbh_index* ix);

// run a search body under a lease; when it reports the fp16 copy out of date, bring it up to date exclusively and retry
template <class Body>
int with_lease(lbh_index* ix, Body&& body) {
    // the calling thread's current device is its own business (a fresh thread starts on device 0): everything below --
    // a slot created on first overlap included -- belongs on the index's device
    HIP_TRY(hipSetDevice(ix->ctx->device));
    for (int attempt = 0;; ++attempt) {
        int rc;
        {
            Lease lease(ix);
            rc = lease.acquire();
            if (rc) return rc;
            rc = body(lease.slot());
        }
        if (rc != LBH_NEED_EMB16 && rc != LBH_NEED_EMB8) return rc;
        if (attempt >= 8) return fail(LBH_E_INTERNAL, "the fp16 / int8 copy kept going out of date under concurrent writers");
        const bool want8 = rc == LBH_NEED_EMB8;
        Exclusive w(ix);
        HIP_TRY(hipSetDevice(ix->ctx->device));
        rc = want8 ? ensure_emb8(ix) : ensure_emb16(ix);
        if (rc) return rc;
        HIP_TRY(hipStreamSynchronize(ix->ctx->stream));
    }
}

inline bool emb16_current(const lbh_index* ix) { return ix->emb16 && ix->emb16_rows == ix->n && ix->emb16_cap >= ix->cap; }
inline bool emb8_current(const lbh_index* ix) { return ix->emb8 && ix->emb8_rows == ix->n && ix->emb8_cap >= ix->cap; }

lbh_dev_index dev_view(const lbh_index* ix) {
    lbh_dev_index d;
    d.emb32 = ix->emb32;
    d.emb64 = ix->emb64;
    d.created = ix->created;
    d.tags = ix->tags;
    d.keys = ix->keys;
    d.ids = ix->ids;
    d.n = ix->n;
    return d;
}

inline uint64_t load_be64(const uint8_t* p) {
    uint64_t v;
    memcpy(&v, p, 8);
    return __builtin_bswap64(v);
}
inline void store_be64(uint8_t* p, uint64_t v) {
    v = __builtin_bswap64(v);
    memcpy(p, &v, 8);
}

double vec_norm(const double* v) {
    double s = 0.0;
    for (int d = 0; d < LBH_DIM; ++d) s += v[d] * v[d];
    return std::sqrt(s);
}

// |scan score - exact score| <= eps * |x| * |q|; the reference's embeddings are unit vectors (or zero), but vectors
// handed in directly (lbh_index_add with emb_f64, ContextGraph._embedding_cache[h] = v) and query vectors need not be
float norm_scale(const lbh_index* ix, double qnorm) {
    const double s = ix->max_norm * qnorm * (1.0 + 1e-6);
    return (float)std::min(std::max(s, 1e-30), 1e30);
}

// Latency path: the final kernel of a query writes its result block into mapped host memory and
// publishes the ticket last, so polling it skips the runtime's completion hand-off.  Bounded: past
// spin_us (a hung or faulted kernel, or a very long scan) fall back to the stream, which also reports
// device errors.
hipError_t wait_ticket(lbh_ctx* c, hipStream_t stream, const int32_t* ticket, int32_t seq) {
    if (c->spin_us > 0) {
        const auto t0 = std::chrono::steady_clock::now();
        for (int it = 0;; ++it) {
            if (__atomic_load_n(ticket, __ATOMIC_ACQUIRE) == seq) return hipSuccess;
            __builtin_ia32_pause();
            if ((it & 255) == 255 &&
                std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() > c->spin_us)
                break;
        }
    }
    return hipStreamSynchronize(stream);
}

// The same for a call whose stream carries a collective: a peer rank that never arrives (it died, or it failed before
// its all-gather) must not hang this one, so the wait is bounded -- poll the ticket (NULL: the stream alone), then the
// stream, for at most coll_timeout_ms.  A timeout leaves the collective queued: the caller marks its communicator broken.
int wait_collective(lbh_ctx* c, hipStream_t stream, const int32_t* ticket, int32_t seq) {
    const auto t0 = std::chrono::steady_clock::now();
    auto elapsed_us = [&] { return std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count(); };
    if (ticket && c->spin_us > 0) {
        for (int it = 0;; ++it) {
            if (__atomic_load_n(ticket, __ATOMIC_ACQUIRE) == seq) return LBH_OK;
            __builtin_ia32_pause();
            if ((it & 255) == 255 && elapsed_us() > c->spin_us) break;
        }
    }
    for (;;) {
        if (ticket && __atomic_load_n(ticket, __ATOMIC_ACQUIRE) == seq) return LBH_OK;
        const hipError_t e = hipStreamQuery(stream);
        if (e == hipSuccess) {
            if (!ticket || __atomic_load_n(ticket, __ATOMIC_ACQUIRE) == seq) return LBH_OK;
            return fail(LBH_E_INTERNAL, "the stream drained without publishing the completion ticket");
        }
        if (e != hipErrorNotReady) return fail(LBH_E_DEVICE, "hipStreamQuery: %s", hipGetErrorString(e));
        if (elapsed_us() > (int64_t)c->coll_timeout_ms * 1000)
            return fail(LBH_E_DEVICE, "claim-sharded search: no completion after %d ms -- a peer rank never reached the all-gather "
                                      "(or the device hangs); this communicator is unusable from here on", c->coll_timeout_ms);
        struct timespec ts = {0, 50000};
        nanosleep(&ts, nullptr);
    }
}

// fp16 copy (tiled, lbh_emb16_off) for the fp16 scans: allocate with the index capacity rounded up to whole 32-row
// tiles, zero-filled (the pad rows of the last tile are read and masked, never trusted), convert the rows that are
// not in it yet
int ensure_emb16(lbh_index* ix) {
    hipStream_t s = ix->ctx->stream;
    if (!ix->emb16 || ix->emb16_cap < ix->cap) {
        if (ix->emb16) (void)hipFree(ix->emb16);
        ix->emb16 = nullptr;
        ix->emb16_rows = ix->emb16_cap = 0;
        const size_t bytes = (size_t)((ix->cap + 63) / 64) * 64 * 128;
        HIP_TRY(hipMalloc((void**)&ix->emb16, bytes));
        HIP_TRY(hipMemsetAsync(ix->emb16, 0, bytes, s));
        ix->emb16_cap = ix->cap;
    }
    if (ix->emb16_rows < ix->n) {
        HIP_TRY(lbh_launch_f64_to_f16_tiled(ix->emb64, ix->emb16, ix->emb16_rows, ix->n, s));
        ix->emb16_rows = ix->n;
    }
    return LBH_OK;
}

// int8 copy (tiled, lbh_emb8_off) for the int8 batched scan.  ONE scale serves the whole index: the largest |element| of
// the rows to convert is found first; while it fits the current scale only the new rows are converted, else every row is
// converted again under a scale with 1/16 of headroom (after the first batch of a corpus this does not happen again in
// practice: element magnitudes are a property of the embedding, not of the claim).  Called under the exclusive lock.
int ensure_emb8(lbh_index* ix) {
    hipStream_t s = ix->ctx->stream;
    if (!ix->emb8 || ix->emb8_cap < ix->cap) {
        if (ix->emb8) (void)hipFree(ix->emb8);
        ix->emb8 = nullptr;
        ix->emb8_rows = ix->emb8_cap = 0;
        const size_t bytes = (size_t)((ix->cap + 31) / 32) * 32 * 64;
        HIP_TRY(hipMalloc((void**)&ix->emb8, bytes));
        HIP_TRY(hipMemsetAsync(ix->emb8, 0, bytes, s));
        ix->emb8_cap = ix->cap;
    }
    if (!ix->emb8_absmax) HIP_TRY(hipMalloc((void**)&ix->emb8_absmax, 4));
    if (ix->emb8_rows < ix->n) {
        auto absmax = [&](int64_t r0, int64_t r1, float* out) -> int {
            uint32_t bits = 0;
            HIP_TRY(hipMemsetAsync(ix->emb8_absmax, 0, 4, s));
            HIP_TRY(lbh_launch_absmax_f64(ix->emb64, r0, r1, ix->emb8_absmax, s));
            HIP_TRY(hipMemcpyAsync(&bits, ix->emb8_absmax, 4, hipMemcpyDeviceToHost, s));
            HIP_TRY(hipStreamSynchronize(s));
            memcpy(out, &bits, 4);
            return LBH_OK;
        };
        float mx = 0.f;
        int rc = absmax(ix->emb8_rows, ix->n, &mx);
        if (rc) return rc;
        if (!(mx <= 3.0e38f)) return fail(LBH_E_INTERNAL, "int8 copy: non-finite element in the matrix");
        int64_t from = ix->emb8_rows;
        if (ix->emb8_rows == 0 || (double)mx > 127.0 * ix->emb8_scale) {
            if (ix->emb8_rows > 0) {  // the old rows count too
                float mx_old = 0.f;
                rc = absmax(0, ix->emb8_rows, &mx_old);
                if (rc) return rc;
                mx = std::max(mx, mx_old);
            }
            ix->emb8_scale = mx > 0.f ? (double)mx * 1.0625 / 127.0 : 1.0;
            from = 0;
        }
        HIP_TRY(lbh_launch_f64_to_i8_tiled(ix->emb64, ix->emb8, from, ix->n, 1.0 / ix->emb8_scale, s));
        ix->emb8_rows = ix->n;
    }
    return LBH_OK;
}

struct PageOut {
    int count = 0;
    uint32_t rows[LBH_MAX_PAGE];
    double scores[LBH_MAX_PAGE];
    uint64_t keys[LBH_MAX_PAGE][4];
};

struct SearchAcc {
    int path = 0;
    int64_t rows_scanned = 0, bytes_per_row = 0, candidates = 0;
    int scan_launches = 0, fallbacks = 0;
    float scan_ms = 0.f;
    float max_err = 0.f;
};

// one device pass producing up to `page` (<= 64) results; q32/q64 must already be in the workspace
int run_page(lbh_index* ix, lbh_slot* sl, int page, int64_t since_ms, uint64_t tag_mask, bool allow_fast, const lbh_cursor& cur,
             bool timing, PageOut* out, SearchAcc* acc) {
    lbh_ctx* c = ix->ctx;
    hipStream_t s = sl->stream;
    const lbh_dev_index dv = dev_view(ix);
    const int has_since = since_ms != LBH_SINCE_NONE;
    const int grid = lbh_scan_grid(ix->n, c->n_cu, c->scan_blocks_per_cu);
    bool fast = allow_fast && !cur.active;
    for (;;) {
        hipEvent_t e0 = timing ? sl->ev[0] : nullptr, e1 = timing ? sl->ev[1] : nullptr;
        const int32_t seq = sl->seq = (sl->seq == INT32_MAX) ? 1 : sl->seq + 1;
        // the fp16 scan is for unit-scale data (its range and its error bound); anything else takes the fp32 matrix
        const float nscale = norm_scale(ix, sl->qnorm);
        const bool half = fast && c->scan_mode == 1 && nscale <= 1.001f && nscale >= 0.25f;
        if (half && !emb16_current(ix)) return LBH_NEED_EMB16;
        if (fast)
            HIP_TRY(lbh_launch_search_fast(dv, half ? ix->emb16 : nullptr, sl->ws, sl->qptr, sl->qarg, page, has_since, since_ms,
                                           tag_mask, (half && !has_since && !tag_mask) ? lbh_scan_grid(ix->n, c->n_cu, c->scan_blocks_per_cu, true) : grid, c->n_cu, c->scan_nt, seq, (half ? LBH_MARGIN_F16 : LBH_MARGIN) * nscale, s, e0, e1, nullptr, c->scan_fuse, c->scan_xb_k));
        else {
            const int xgrid = std::min(grid, 512);
            HIP_TRY(lbh_launch_search_exact(dv, sl->ws, sl->qptr, sl->qarg, page, has_since, since_ms, tag_mask, cur, xgrid, seq, s, e0, e1));
        }
        const lbh_result* r = sl->ws.result_host;
        HIP_TRY(wait_ticket(c, s, &r->seq, seq));
        if (timing) HIP_TRY(hipEventSynchronize(sl->ev[1]));
        acc->scan_launches += 1;
        acc->rows_scanned = ix->n;
        if (timing) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, sl->ev[0], sl->ev[1]) == hipSuccess) acc->scan_ms += ms;
        }
        if (fast) {
            acc->path = 0;
            acc->bytes_per_row = (half ? 128 : 256) + (has_since ? 8 : 0) + (tag_mask ? 8 : 0);
        } else {
            acc->path = 1;
            acc->bytes_per_row = 512 + (has_since ? 8 : 0) + (tag_mask ? 8 : 0);
        }
        if (r->overflow == 1 && fast) {  // candidate set unbounded in fp32 (mass near-ties): exact pass
            fast = false;
            acc->fallbacks += 1;
            continue;
        }
        if (r->overflow)
            return fail(LBH_E_INTERNAL, "exact path: more than %d claims tie on score AND carry the same 32-byte hash key (keys must be unique)", LBH_MCAP);
        acc->candidates += r->candidates;
        if (fast) acc->max_err = std::max(acc->max_err, r->max_err);
        out->count = r->count;
        memcpy(out->rows, r->rows, sizeof(uint32_t) * r->count);
        memcpy(out->scores, r->scores, sizeof(double) * r->count);
        memcpy(out->keys, r->keys, sizeof(uint64_t) * 4 * r->count);
        return LBH_OK;
    }
}

// full search for one query whose vectors are already in ws.q32 / ws.q64
int search_loaded(lbh_index* ix, lbh_slot* sl, int64_t k, int64_t since_ms, uint64_t tag_mask, uint64_t* out_ids, double* out_scores,
                  uint8_t* out_keys, int64_t* out_count, lbh_stats* stats) {
    const bool timing = stats && stats->want_timing;
    SearchAcc acc;
    int64_t got = 0;
    const bool timing_total = timing && stats->want_timing >= 2;  // two more event records per call
    if (timing_total) HIP_TRY(hipEventRecord(sl->ev[2], sl->stream));
    if (k > 0 && ix->n > 0) {
        lbh_cursor cur;
        memset(&cur, 0, sizeof cur);
        int64_t remaining = std::min<int64_t>(k, ix->n);
        while (remaining > 0) {
            const int page = (int)std::min<int64_t>(remaining, LBH_MAX_PAGE);
            PageOut po;
            int rc = run_page(ix, sl, page, since_ms, tag_mask, !ix->ctx->force_exact, cur, timing, &po, &acc);
            if (rc) return rc;
            for (int i = 0; i < po.count; ++i) {
                out_ids[got + i] = ix->id_of_row[po.rows[i]];
                out_scores[got + i] = po.scores[i];
                if (out_keys)
                    for (int w = 0; w < 4; ++w) store_be64(out_keys + (got + i) * 32 + 8 * w, po.keys[i][w]);
            }
            got += po.count;
            remaining -= po.count;
            if (po.count < page) break;
            cur.active = 1;
            cur.ord = lbh_ord64(po.scores[po.count - 1]);
            for (int w = 0; w < 4; ++w) cur.key[w] = po.keys[po.count - 1][w];
        }
    }
    *out_count = got;
    if (stats) {
        float total = 0.f;
        if (timing_total) {
            HIP_TRY(hipEventRecord(sl->ev[3], sl->stream));
            HIP_TRY(hipEventSynchronize(sl->ev[3]));
            (void)hipEventElapsedTime(&total, sl->ev[2], sl->ev[3]);
        }
        stats->path = acc.path;
        stats->rows_scanned = acc.rows_scanned;
        stats->bytes_per_row = acc.bytes_per_row;
        stats->candidates = acc.candidates;
        stats->max_score_err = acc.max_err;
        stats->scan_launches = acc.scan_launches;
        stats->fallbacks = acc.fallbacks;
        stats->scan_ms = acc.scan_ms;
        stats->total_ms = total;
    }
    return LBH_OK;
}

int ensure_batched_ws(lbh_slot* sl) {
    lbh_workspace& w = sl->ws;
    if (w.bq_qhost_dev) return LBH_OK;  // the last pointer set: a call that failed half-way is completed by the next one
    if (!w.bq_qmat) HIP_TRY(hipMalloc((void**)&w.bq_qmat, (size_t)LBH_BQ_MAXQ * 512));
    if (!w.bq_cand) HIP_TRY(hipMalloc((void**)&w.bq_cand, (size_t)LBH_BQ_MAXQ * LBH_BQ_PERQ * 8));
    if (!w.bq_total) HIP_TRY(hipMalloc((void**)&w.bq_total, (size_t)LBH_BQ_MAXQ * LBH_BQ_QT_STRIDE * 4));
    if (!w.bq_seed) HIP_TRY(hipMalloc((void**)&w.bq_seed, (size_t)LBH_BQ_MAXQ * 4));
    if (!w.bq_bfrag) HIP_TRY(hipMalloc((void**)&w.bq_bfrag, (size_t)LBH_BQ_MAXQ * 128));
    if (!w.bq_qscale) HIP_TRY(hipMalloc((void**)&w.bq_qscale, (size_t)LBH_BQ_MAXQ * 4));
    if (!w.bq_result_host) {
        HIP_TRY(hipHostMalloc((void**)&w.bq_result_host, sizeof(lbh_result) * LBH_BQ_MAXQ, hipHostMallocMapped | hipHostMallocCoherent));
        memset(w.bq_result_host, 0, sizeof(lbh_result) * LBH_BQ_MAXQ);
    }
    HIP_TRY(hipHostGetDevicePointer((void**)&w.bq_result_dev, w.bq_result_host, 0));
    if (!w.bq_qhost) HIP_TRY(hipHostMalloc((void**)&w.bq_qhost, (size_t)LBH_BQ_MAXQ * 512, hipHostMallocMapped));
    HIP_TRY(hipHostGetDevicePointer((void**)&w.bq_qhost_dev, w.bq_qhost, 0));
    return LBH_OK;
}

int ensure_recent_ws(lbh_slot* sl) {
    lbh_workspace& w = sl->ws;
    if (w.rc_result_host) return LBH_OK;
    const size_t cap = (size_t)LBH_RC_MAX_BLOCKS * LBH_MAX_PAGE;
    if (!w.rc_a) HIP_TRY(hipMalloc((void**)&w.rc_a, cap * 8));
    if (!w.rc_b) HIP_TRY(hipMalloc((void**)&w.rc_b, cap * 8));
    if (!w.rc_row) HIP_TRY(hipMalloc((void**)&w.rc_row, cap * 4));
    if (!w.rc_cnt) HIP_TRY(hipMalloc((void**)&w.rc_cnt, 64));
    HIP_TRY(hipMemset(w.rc_cnt, 0, 64));
    lbh_recent_result* h = nullptr;
    HIP_TRY(hipHostMalloc((void**)&h, sizeof(lbh_recent_result), hipHostMallocMapped | hipHostMallocCoherent));
    memset(h, 0, sizeof *h);
    HIP_TRY(hipHostGetDevicePointer((void**)&w.rc_result_dev, h, 0));
    w.rc_result_host = h;
    return LBH_OK;
}

int load_query(lbh_slot* sl, const double* q) {
    const double nrm = vec_norm(q);
    if (!(nrm <= 1e15)) return fail(LBH_E_INVALID, "query vector is not finite (or its norm exceeds 1e15)");
    memcpy(sl->qarg.q, q, sizeof sl->qarg.q);  // travels to the GPU inside the kernel-argument segment
    sl->qptr = nullptr;
    sl->qnorm = nrm;
    return LBH_OK;
}

// merge step of the claim-sharded search over lists that need not be contiguous: bases[l] = (k + 1) x 6 int64 (lbhip.h, lbh_merge_topk)
int merge_lists(const int64_t* const* bases, int64_t n_lists, int64_t k, int64_t* out_entries, int64_t* out_count) {
    struct Ref { double s; const int64_t* e; };
    std::vector<Ref> all;
    all.reserve((size_t)(n_lists * k));
    for (int64_t l = 0; l < n_lists; ++l) {
        const int64_t* base = bases[l];
        const int64_t cnt = base[0];
        if (cnt < 0 || cnt > k) return fail(LBH_E_INVALID, "list %lld announces %lld entries, k = %lld", (long long)l, (long long)cnt, (long long)k);
        for (int64_t i = 1; i <= cnt; ++i) {
            double sc;
            memcpy(&sc, base + i * 6, 8);
            all.push_back({sc, base + i * 6});
        }
    }
    // lb/context_graph.py:120: score descending (-0.0 == 0.0, as in Python), then hash ascending
    auto before = [](const Ref& a, const Ref& b) {
        if (a.s != b.s) return a.s > b.s;
        for (int w = 2; w < 6; ++w) {
            const uint64_t x = (uint64_t)a.e[w], y = (uint64_t)b.e[w];
            if (x != y) return x < y;
        }
        return false;
    };
    const size_t take = std::min<size_t>((size_t)k, all.size());
    std::partial_sort(all.begin(), all.begin() + take, all.end(), before);
    for (size_t i = 0; i < take; ++i) memcpy(out_entries + i * 6, all[i].e, 48);
    *out_count = (int64_t)take;
    return LBH_OK;
}

}  // namespace

extern "C" {

const char* lbh_last_error(void) { return g_err.c_str(); }

int lbh_version(void) { return 1; }

int lbh_init(const int* devices, int n_devices, lbh_ctx** out_ctx) {
    if (!out_ctx) return fail(LBH_E_INVALID, "out_ctx is NULL");
    *out_ctx = nullptr;
    if (n_devices != 1 || !devices) return fail(LBH_E_INVALID, "one context drives exactly one GPU (one process per GPU)");
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(LBH_E_DEVICE, "no HIP device available (%s); liblbhip has no CPU fallback",
                    e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    if (devices[0] < 0 || devices[0] >= count) return fail(LBH_E_INVALID, "device %d out of range [0,%d)", devices[0], count);
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, devices[0]));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(LBH_E_DEVICE, "device %d is %s; liblbhip is built for gfx950 (MI355X) only", devices[0], prop.gcnArchName);
    lbh_ctx* c = new (std::nothrow) lbh_ctx();
    if (!c) return fail(LBH_E_NOMEM, "out of host memory");
    c->device = devices[0];
    c->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    c->arch = prop.gcnArchName;
    if (const char* e1 = getenv("LBH_SCAN_BLOCKS_PER_CU")) c->scan_blocks_per_cu = std::max(0, std::min(8, atoi(e1)));
    if (const char* e2 = getenv("LBH_SCAN_NT")) c->scan_nt = atoi(e2) != 0;
    if (const char* e5 = getenv("LBH_SCAN_MODE")) c->scan_mode = atoi(e5) != 0;
    if (const char* e6 = getenv("LBH_SCAN_SPLIT")) {  // "a,b,c,d": tile shares of the scan's four block classes (lbh_split); "0" = grid-stride
        unsigned w[4] = {0, 0, 0, 0};
        const int got = sscanf(e6, "%u,%u,%u,%u", &w[0], &w[1], &w[2], &w[3]);
        const uint32_t ww[4] = {std::min(w[0], 65535u), std::min(w[1], 65535u), std::min(w[2], 65535u), std::min(w[3], 65535u)};
        if (got == 4 || got == 1) lbh_set_scan_split(ww);
    }
    if (const char* e4 = getenv("LBH_BQ_MODE")) c->bq_mode = std::max(0, std::min(3, atoi(e4)));
    if (const char* e3 = getenv("LBH_SPIN_US")) c->spin_us = std::max(0, std::min(1000000, atoi(e3)));
    if (const char* e9 = getenv("LBH_K1_MEMO_LOG2")) c->memo_log2 = atoi(e9) <= 0 ? 0 : std::max(8, std::min(24, atoi(e9)));
    if (const char* e9 = getenv("LBH_SCAN_XB_K")) c->scan_xb_k = std::max(0, atoi(e9));
    if (const char* e8 = getenv("LBH_SCAN_FUSE")) c->scan_fuse = std::max(0, std::min(2, atoi(e8)));
    if (const char* e7 = getenv("LBH_COLL_TIMEOUT_MS")) c->coll_timeout_ms = std::max(1, std::min(3600000, atoi(e7)));
    hipError_t e0 = hipSetDevice(c->device);
    if (e0 == hipSuccess) e0 = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e0 == hipSuccess) e0 = hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking);
    for (auto& ev : c->copy_ev)
        if (e0 == hipSuccess) e0 = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    // with this device current: where ITS copy of the allocation-free entry point's scratch array lives
    if (e0 == hipSuccess) e0 = lbh_noalloc_init(&c->noalloc_scratch);
    if (e0 != hipSuccess || !c->noalloc_scratch) {
        if (c->stream) (void)hipStreamDestroy(c->stream);
        delete c;
        return fail(LBH_E_DEVICE, "context set-up on device %d failed: %s", devices[0], hipGetErrorString(e0));
    }
    *out_ctx = c;
    return LBH_OK;
}

int lbh_shutdown(lbh_ctx* c) {
    if (!c) return LBH_OK;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->d_text) (void)hipFree(c->d_text);
    if (c->d_off) (void)hipFree(c->d_off);
    if (c->d_vec) (void)hipFree(c->d_vec);
    if (c->memo_table) (void)hipFree(c->memo_table);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    for (int b = 0; b < LBH_STAGE_NBUF; ++b) {
        if (c->stage_buf[b]) (void)hipHostFree(c->stage_buf[b]);
        if (c->stage_ev[b]) (void)hipEventDestroy(c->stage_ev[b]);
    }
    for (auto& ev : c->copy_ev)
        if (ev) (void)hipEventDestroy(ev);
    delete c;
    return LBH_OK;
}

int lbh_device_name(lbh_ctx* c, char* buf, int buflen) {
    if (!c || !buf || buflen <= 0) return fail(LBH_E_INVALID, "bad argument");
    snprintf(buf, (size_t)buflen, "%s", c->arch.c_str());
    return LBH_OK;
}

int lbh_set_option(lbh_ctx* c, const char* name, int64_t value) {
    if (!c || !name) return fail(LBH_E_INVALID, "bad argument");
    std::lock_guard<std::mutex> lk(c->mu);
    if (!strcmp(name, "scan_blocks_per_cu")) c->scan_blocks_per_cu = (int)std::max<int64_t>(0, std::min<int64_t>(8, value));
    else if (!strcmp(name, "scan_nt")) c->scan_nt = value != 0;
    else if (!strcmp(name, "scan_mode")) c->scan_mode = value != 0;
    else if (!strcmp(name, "bq_mode")) c->bq_mode = (int)std::max<int64_t>(0, std::min<int64_t>(3, value));
    else if (!strcmp(name, "spin_us")) c->spin_us = (int)std::max<int64_t>(0, std::min<int64_t>(1000000, value));
    else if (!strcmp(name, "coll_timeout_ms")) c->coll_timeout_ms = (int)std::max<int64_t>(1, std::min<int64_t>(3600000, value));
    else if (!strcmp(name, "k1_memo_log2")) {
        const int lg = value <= 0 ? 0 : (int)std::max<int64_t>(8, std::min<int64_t>(24, value));
        if (c->memo_table && lg != c->memo_log2) {  // a new size (or off): drop the table once the stream is idle
            (void)hipSetDevice(c->device);
            (void)hipStreamSynchronize(c->stream);
            (void)hipFree(c->memo_table);
            c->memo_table = nullptr;
        }
        c->memo_log2 = lg;
    }
    else if (!strcmp(name, "stage_helpers")) c->stage_helpers = (int)std::max<int64_t>(1, std::min<int64_t>(LBH_STAGE_HELPERS, value));
    else if (!strcmp(name, "add_cuts")) {
        const int a = (int)(value / 10000 % 100), b = (int)(value / 100 % 100), d = (int)(value % 100);
        if (value != 0 && !(0 < a && a < b && b < d && d < 16)) return fail(LBH_E_INVALID, "add_cuts: need 0 < a < b < c < 16 as a * 10000 + b * 100 + c");
        c->add_cuts = (int)value;
    }
    else if (!strcmp(name, "scan_xb_k")) c->scan_xb_k = (int)std::max<int64_t>(0, std::min<int64_t>(1 << 20, value));
    else if (!strcmp(name, "scan_fuse")) c->scan_fuse = (int)std::max<int64_t>(0, std::min<int64_t>(2, value));
    else if (!strcmp(name, "force_exact")) c->force_exact = value != 0;
    else if (!strcmp(name, "inject_fault")) c->inject_fault = (int)std::max<int64_t>(0, std::min<int64_t>(2, value));
    else return fail(LBH_E_INVALID, "unknown option %s", name);
    return LBH_OK;
}

int lbh_host_alloc(lbh_ctx* c, int64_t bytes, void** out_ptr) {
    if (!c || !out_ptr || bytes < 0) return fail(LBH_E_INVALID, "bad argument");
    *out_ptr = nullptr;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipHostMalloc(out_ptr, (size_t)std::max<int64_t>(bytes, 1), hipHostMallocDefault));
    return LBH_OK;
}

int lbh_host_free(lbh_ctx* c, void* ptr) {
    if (!c) return fail(LBH_E_INVALID, "bad argument");
    if (ptr) HIP_TRY(hipHostFree(ptr));
    return LBH_OK;
}

int lbh_embed_batch(lbh_ctx* c, const uint8_t* utf8, const int64_t* offsets, int64_t n, double* out_f64) {
    return lbh_embed_batch_dim(c, utf8, offsets, n, LBH_DIM, out_f64);
}

int lbh_embed_batch_dim(lbh_ctx* c, const uint8_t* utf8, const int64_t* offsets, int64_t n, int dim, double* out_f64) try {
    if (!c || n < 0 || (n > 0 && (!offsets || !out_f64))) return fail(LBH_E_INVALID, "bad argument");
    if (dim < 1 || dim > LBH_DIM) return fail(LBH_E_INVALID, "dim must be in 1..64 (a token only has 64 digest bytes, lb/latent.py:49-52)");
    if (n == 0) return LBH_OK;
    const int64_t base = offsets[0], bytes = offsets[n] - base;
    if (bytes < 0 || (bytes > 0 && !utf8)) return fail(LBH_E_INVALID, "bad offsets");
    for (int64_t i = 0; i < n; ++i) {
        const int64_t l = offsets[i + 1] - offsets[i];
        if (l < 0 || l >= (int64_t)0x7FFFFF00) return fail(LBH_E_INVALID, "text %lld has invalid length %lld", (long long)i, (long long)l);
    }
    std::lock_guard<std::mutex> lk(c->mu);
    int rc = ensure_ctx_scratch(c, bytes + 64, n + 1, n * 64);
    if (rc) return rc;
    hipStream_t s = c->stream;
    std::vector<int64_t> rel;
    const int64_t* offp = offsets;
    if (base != 0) {
        rel.resize((size_t)n + 1);
        for (int64_t i = 0; i <= n; ++i) rel[(size_t)i] = offsets[i] - base;
        offp = rel.data();
    }
    if (bytes > 0) HIP_TRY(hipMemcpyAsync(c->d_text, utf8 + base, (size_t)bytes, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(c->d_off, offp, (size_t)(n + 1) * 8, hipMemcpyHostToDevice, s));
    const lbh_memo memo = next_memo(c, n);
    HIP_TRY(lbh_launch_embed(c->d_text, c->d_off, n, c->d_vec, nullptr, s, dim, &memo));
    HIP_TRY(hipMemcpyAsync(out_f64, c->d_vec, (size_t)n * 64 * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return LBH_OK;
} LBH_CATCH_ALLOC

int lbh_embed_batch_dev(lbh_ctx* c, const uint8_t* utf8_dev, const int64_t* offsets_dev, int64_t n, int64_t total_bytes,
                        double* out_f64_dev, float* out_f32_dev) {
    if (!c || n < 0 || total_bytes < 0 || (n > 0 && (!offsets_dev || !out_f64_dev))) return fail(LBH_E_INVALID, "bad argument");
    if (n == 0) return LBH_OK;
    HIP_TRY(hipSetDevice(c->device));
    lbh_memo memo;
    {
        std::lock_guard<std::mutex> lk(c->mu);
        memo = next_memo(c, n);
    }
    HIP_TRY(lbh_launch_embed(utf8_dev, offsets_dev, n, out_f64_dev, out_f32_dev, c->stream, LBH_DIM, &memo));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return LBH_OK;
}

int lbh_cosine(lbh_ctx* c, const double* a, int64_t na, const double* b, int64_t nb, double* out) {
    if (!c || !out || na < 0 || nb < 0) return fail(LBH_E_INVALID, "bad argument");
    if (na != nb) return fail(LBH_E_INVALID, "dimension mismatch");
    std::lock_guard<std::mutex> lk(c->mu);
    int rc = ensure_ctx_scratch(c, 0, 0, 2 * na + 8);
    if (rc) return rc;
    hipStream_t s = c->stream;
    if (na > 0) {
        HIP_TRY(hipMemcpyAsync(c->d_vec, a, (size_t)na * 8, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(c->d_vec + na, b, (size_t)na * 8, hipMemcpyHostToDevice, s));
    }
    HIP_TRY(lbh_launch_cosine(c->d_vec, c->d_vec + na, na, c->d_vec + 2 * na, s));
    HIP_TRY(hipMemcpyAsync(out, c->d_vec + 2 * na, 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return LBH_OK;
}

int lbh_index_create(lbh_ctx* c, lbh_index** out_idx) {
    if (!c || !out_idx) return fail(LBH_E_INVALID, "bad argument");
    *out_idx = nullptr;
    HIP_TRY(hipSetDevice(c->device));
    lbh_index* ix = new (std::nothrow) lbh_index();
    if (!ix) return fail(LBH_E_NOMEM, "out of host memory");
    ix->ctx = c;
    int rc = slot_create(ix, 0, &ix->slots[0]);
    if (!rc && hipEventCreateWithFlags(&ix->mut_ev, hipEventDisableTiming) != hipSuccess) rc = fail(LBH_E_DEVICE, "hipEventCreate failed");
    if (!rc) rc = mutation_done(ix);
    if (rc) {
        const std::string why = g_err;  // lbh_index_destroy must not hide the reason
        lbh_index_destroy(ix);
        g_err = why;
        return rc;
    }
    *out_idx = ix;
    return LBH_OK;
}

int lbh_index_destroy(lbh_index* ix) {
    if (!ix) return LBH_OK;
    (void)hipSetDevice(ix->ctx->device);
    (void)hipStreamSynchronize(ix->ctx->stream);
    for (auto& sl : ix->slots) {
        slot_destroy(sl);
        sl = nullptr;
    }
    void* dev[] = {ix->emb16, ix->emb8, ix->emb8_absmax, ix->emb32, ix->emb64, ix->created, ix->tags, ix->keys, ix->order, ix->ids};
    for (void* p : dev)
        if (p) (void)hipFree(p);
    if (ix->mut_ev) (void)hipEventDestroy(ix->mut_ev);
    if (ix->staging) (void)hipHostFree(ix->staging);
    if (ix->small_stage) (void)hipHostFree(ix->small_stage);
    if (ix->dscratch) (void)hipFree(ix->dscratch);
    delete ix;
    return LBH_OK;
}

int lbh_index_reserve(lbh_index* ix, int64_t capacity) {
    if (!ix || capacity < 0) return fail(LBH_E_INVALID, "bad argument");
    Exclusive lk(ix);
    return index_reserve_locked(ix, capacity);
}

}  // extern "C"

namespace {
// Host -> device copies of one lbh_index_add call.  Pinned sources (lbh_host_alloc) go to the copy engine as they are.
// PAGEABLE sources -- ordinary numpy arrays, std::vectors: what the drop-in Python API hands over -- would be staged by
// the runtime through its own bounce buffers by ONE thread (~20 GB/s: 8.3 ms for the 168 MB of a 1 M-claim batch, where
// the link needs 2.9); here two helper threads copy them piece by piece (8 MiB) into a ring of pinned buffers (45 GB/s
// between them, tools/stage_rate.py) while the calling thread hands finished pieces to the copy engine in order, so the
// K1 slices still start as soon as their text has crossed.  Use: plan() every copy of the call in order, start(), then
// copy() them in the same order; the destructor stops the helpers (also on an error return half-way).
class StagedCopier {
  public:
    explicit StagedCopier(lbh_ctx* c) : c_(c) {}
    StagedCopier(const StagedCopier&) = delete;
    StagedCopier& operator=(const StagedCopier&) = delete;
    ~StagedCopier() { stop(); }

    void plan(const void* src, size_t bytes) {
        Op op{src, bytes, pieces_.size(), 0};
        if (bytes >= 65536 && pageable(src)) {
            for (size_t off = 0; off < bytes; off += LBH_STAGE_PIECE) {
                pieces_.push_back({static_cast<const uint8_t*>(src) + off, std::min(LBH_STAGE_PIECE, bytes - off)});
                staged_bytes_ += pieces_.back().bytes;
            }
            op.n_pieces = pieces_.size() - op.first;
        }
        ops_.push_back(op);
    }

    int start() {
        if (staged_bytes_ < ((size_t)4 << 20)) {  // small calls: the runtime's own staging is fine
            for (auto& o : ops_) o.n_pieces = 0;
            pieces_.clear();
            return LBH_OK;
        }
        lk_ = std::unique_lock<std::mutex>(c_->stage_mu);
        for (int b = 0; b < LBH_STAGE_NBUF; ++b) {
            if (!c_->stage_buf[b]) HIP_TRY(hipHostMalloc(&c_->stage_buf[b], LBH_STAGE_PIECE, hipHostMallocDefault));
            if (!c_->stage_ev[b]) HIP_TRY(hipEventCreateWithFlags(&c_->stage_ev[b], hipEventDisableTiming));
        }
        ready_.reset(new std::atomic<int>[pieces_.size()]);
        issued_.reset(new std::atomic<int>[pieces_.size()]);
        for (size_t i = 0; i < pieces_.size(); ++i) { ready_[i].store(0); issued_[i].store(0); }
        nh_ = std::max(1, std::min(LBH_STAGE_HELPERS, c_->stage_helpers));
        started_ = true;
        try {
            for (int h = 0; h < nh_; ++h) helpers_[h] = std::thread([this, h] { run_helper(h); });
        } catch (const std::system_error&) {  // no thread to be had: stop the ones that started, let the runtime stage the copies
            stop();
            abort_.store(false);
            for (auto& o : ops_) o.n_pieces = 0;
            pieces_.clear();
            lk_.unlock();
        }
        return LBH_OK;
    }

    int copy(void* dst, const void* src, size_t bytes, hipStream_t cs) {
        if (next_op_ >= ops_.size() || ops_[next_op_].src != src || ops_[next_op_].bytes != bytes)
            return fail(LBH_E_INTERNAL, "StagedCopier: copy %zu does not match the plan", next_op_);
        const Op& op = ops_[next_op_++];
        if (bytes == 0) return LBH_OK;
        if (op.n_pieces == 0) {
            HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, cs));
            return LBH_OK;
        }
        size_t off = 0;
        for (size_t i = op.first; i < op.first + op.n_pieces; ++i) {
            const auto t0 = std::chrono::steady_clock::now();
            while (!ready_[i].load(std::memory_order_acquire)) {
                if (failed_.load()) return fail(LBH_E_DEVICE, "StagedCopier: a staging thread failed");
                std::this_thread::yield();
            }
            const auto t1 = std::chrono::steady_clock::now();
            const int b = (int)(i % LBH_STAGE_NBUF);
            HIP_TRY(hipMemcpyAsync(static_cast<uint8_t*>(dst) + off, c_->stage_buf[b], pieces_[i].bytes, hipMemcpyHostToDevice, cs));
            HIP_TRY(hipEventRecord(c_->stage_ev[b], cs));
            issued_[i].store(1, std::memory_order_release);
            off += pieces_[i].bytes;
            wait_us += std::chrono::duration_cast<std::chrono::microseconds>(t1 - t0).count();
            issue_us += std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t1).count();
        }
        return LBH_OK;
    }

    // after the call's final synchronisation (or on the way out of a failed call, after the streams were drained by the
    // caller's guard -- which is declared BEFORE this object, so it runs after this destructor: the helpers are stopped
    // first, then the streams drain, then the ring is free for the next call)
    void stop() {
        abort_.store(true);
        if (started_) {
            for (auto& t : helpers_)
                if (t.joinable()) t.join();
            started_ = false;
        }
    }
    // the ring must not be handed to the next call before this call's copies out of it have completed
    void release_after_sync() {
        stop();
        if (lk_.owns_lock()) lk_.unlock();
    }

    long long wait_us = 0, issue_us = 0;  // (LBH_ADD_TRACE) time the calling thread spent waiting for staged pieces / inside the HIP calls

  private:
    struct Piece { const uint8_t* src; size_t bytes; };
    struct Op { const void* src; size_t bytes; size_t first, n_pieces; };

    static bool pageable(const void* p) {
        hipPointerAttribute_t a;
        const hipError_t e = hipPointerGetAttributes(&a, p);
        if (e != hipSuccess) {
            (void)hipGetLastError();  // an unregistered pointer is reported as an error by older runtimes
            return true;
        }
        return a.type == hipMemoryTypeUnregistered;
    }

    void run_helper(int h) {
        (void)hipSetDevice(c_->device);
        for (size_t i = (size_t)h; i < pieces_.size(); i += (size_t)nh_) {
            const int b = (int)(i % LBH_STAGE_NBUF);
            if (i >= LBH_STAGE_NBUF) {  // the piece that used this buffer last must have left it
                while (!issued_[i - LBH_STAGE_NBUF].load(std::memory_order_acquire)) {
                    if (abort_.load()) return;
                    std::this_thread::yield();
                }
                if (hipEventSynchronize(c_->stage_ev[b]) != hipSuccess) { failed_.store(true); return; }
            }
            if (abort_.load()) return;
            memcpy(c_->stage_buf[b], pieces_[i].src, pieces_[i].bytes);
            ready_[i].store(1, std::memory_order_release);
        }
    }

    lbh_ctx* c_;
    std::vector<Piece> pieces_;
    std::vector<Op> ops_;
    size_t staged_bytes_ = 0, next_op_ = 0;
    std::unique_ptr<std::atomic<int>[]> ready_, issued_;
    std::atomic<bool> abort_{false}, failed_{false};
    std::thread helpers_[LBH_STAGE_HELPERS];
    std::unique_lock<std::mutex> lk_;
    bool started_ = false;
    int nh_ = 1;
};
}  // namespace

extern "C" {

int lbh_index_add(lbh_index* ix, int64_t n, const double* emb_f64, const uint8_t* utf8, const int64_t* offsets,
                  const int64_t* created_ms, const uint64_t* tagmask, const uint8_t* hashkey, uint64_t* out_ids) try {
    if (!ix || n < 0) return fail(LBH_E_INVALID, "bad argument");
    if (n == 0) return LBH_OK;
    if ((emb_f64 != nullptr) == (offsets != nullptr)) return fail(LBH_E_INVALID, "pass either emb_f64 or (utf8, offsets)");
    if (!created_ms || !hashkey) return fail(LBH_E_INVALID, "created_ms and hashkey are required");
    const auto t_enter = std::chrono::steady_clock::now();
    Exclusive lk(ix);
    lbh_ctx* c = ix->ctx;
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    if (ix->n + n > ix->cap) {
        int64_t want = std::max<int64_t>(ix->n + n, std::max<int64_t>(1024, ix->cap + ix->cap / 2));
        int rc = index_reserve_locked(ix, want);
        if (rc) return rc;
    }
    // host id maps first: nothing below may fail for want of host memory once GPU work has been queued
    // (geometric: reserve() to the exact size would re-allocate and copy the ever-growing id map on every call)
    if (ix->row_of_id.capacity() < ix->row_of_id.size() + (size_t)n)
        ix->row_of_id.reserve(std::max(ix->row_of_id.size() + (size_t)n, 2 * ix->row_of_id.capacity()));
    if (ix->id_of_row.capacity() < (size_t)(ix->n + n))
        ix->id_of_row.reserve(std::max((size_t)(ix->n + n), 2 * ix->id_of_row.capacity()));
    const int64_t r0 = ix->n;
    // Per-claim metadata goes straight from the caller's arrays to their device columns (asynchronous DMA when the
    // arrays are pinned -- lbh_host_alloc -- else the runtime stages them); the 32 key bytes are decoded into their four
    // big-endian words by a kernel, not by a host loop.
    // Every host -> device copy of the call goes through h2d(): the call's body runs twice, first to PLAN the copies
    // (so that the staging threads of StagedCopier can run ahead through pageable buffers), then for real.
    bool planning = true;
    StagedCopier* scp = nullptr;
    auto h2d = [&](void* dst, const void* src, size_t bytes, hipStream_t cs) -> int {
        if (planning) {
            scp->plan(src, bytes);
            return LBH_OK;
        }
        return scp->copy(dst, src, bytes, cs);
    };
    auto copy_metadata = [&](hipStream_t cs) -> int {
        int rc = h2d(ix->created + r0, created_ms, (size_t)n * 8, cs);
        if (rc) return rc;
        if (tagmask) rc = h2d(ix->tags + r0, tagmask, (size_t)n * 8, cs);
        else if (!planning) HIP_TRY(hipMemsetAsync(ix->tags + r0, 0, (size_t)n * 8, cs));
        if (rc) return rc;
        return h2d(ix->keys + r0 * 4, hashkey, (size_t)n * 32, cs);
    };
    const uint64_t id0 = (uint64_t)ix->row_of_id.size();  // first claim id of the batch
    auto finish_metadata = [&]() -> int {
        HIP_TRY(lbh_launch_bswap_u64(ix->keys + r0 * 4, n * 4, s));
        HIP_TRY(lbh_launch_iota_u64(ix->order + r0, id0, n, s));
        HIP_TRY(lbh_launch_iota_u64(ix->ids + r0, id0, n, s));
        return LBH_OK;
    };
    double batch_norm = 1.0;  // K1 writes unit vectors (or zero)
    std::vector<int64_t> rel;  // must outlive the copies below: freed after the final synchronise
    std::unique_lock<std::mutex> text_lk(c->mu, std::defer_lock);  // the context's text scratch, held until the final synchronise
    // Every exit below that leaves copies or kernels queued -- a failed enqueue half-way through the slices -- drains both
    // streams first: the copies read the caller's arrays and `rel`, and the text scratch is released with text_lk.
    // (Declared after rel and text_lk, so it runs before they go.)  The normal path synchronises once, further down.
    struct Drain {
        lbh_ctx* c;
        bool armed = true;
        ~Drain() {
            if (!armed) return;
            (void)hipStreamSynchronize(c->copy_stream);
            (void)hipStreamSynchronize(c->stream);
        }
    } drain{c};
    // SMALL text batches (a publish between two queries: N1): everything the kernels need is packed into ONE block of
    // device-mapped pinned memory -- offsets, timestamps, tag masks, keys, text -- and two launches read it from there
    // (K1 the text, small_meta_kernel the rest): no copy-engine hops, no context scratch, where the bulk path below
    // queues five copies and four kernels (~90 us for one claim, ~45 this way).
    if (!emb_f64 && n <= LBH_SMALL_ADD && offsets[n] - offsets[0] >= 0 && offsets[n] - offsets[0] <= LBH_SMALL_ADD_TEXT) {
        const int64_t base = offsets[0], bytes = offsets[n] - base;
        bool bad = bytes > 0 && !utf8;
        for (int64_t i = 0; i < n && !bad; ++i) bad = offsets[i + 1] < offsets[i];
        if (bad) return fail(LBH_E_INVALID, "bad offsets (a text has a negative length or 2 GiB and more)");
        if (!ix->small_stage) {
            HIP_TRY(hipHostMalloc(&ix->small_stage, 65536, hipHostMallocMapped | hipHostMallocCoherent));
            HIP_TRY(hipHostGetDevicePointer(&ix->small_stage_dev, ix->small_stage, 0));
        }
        // layout (8-byte units first): offsets[n + 1] | created[n] | tags[n] | keys[n x 4] | text
        int64_t* h_off = static_cast<int64_t*>(ix->small_stage);
        int64_t* h_created = h_off + (n + 1);
        uint64_t* h_tags = reinterpret_cast<uint64_t*>(h_created + n);
        uint64_t* h_keys = h_tags + n;
        uint8_t* h_text = reinterpret_cast<uint8_t*>(h_keys + 4 * n);
        for (int64_t i = 0; i <= n; ++i) h_off[i] = offsets[i] - base;
        memcpy(h_created, created_ms, (size_t)n * 8);
        if (tagmask) memcpy(h_tags, tagmask, (size_t)n * 8);
        memcpy(h_keys, hashkey, (size_t)n * 32);
        if (bytes) memcpy(h_text, utf8 + base, (size_t)bytes);
        const char* d0 = static_cast<const char*>(ix->small_stage_dev);
        auto dev = [&](const void* host_ptr) { return d0 + (static_cast<const char*>(host_ptr) - static_cast<const char*>(ix->small_stage)); };
        HIP_TRY(lbh_launch_embed(reinterpret_cast<const uint8_t*>(dev(h_text)), reinterpret_cast<const int64_t*>(dev(h_off)), n,
                                 ix->emb64 + r0 * 64, ix->emb32 + r0 * 64, s));
        HIP_TRY(lbh_launch_small_meta(reinterpret_cast<const int64_t*>(dev(h_created)), tagmask ? reinterpret_cast<const uint64_t*>(dev(h_tags)) : nullptr,
                                      reinterpret_cast<const uint64_t*>(dev(h_keys)), ix->created, ix->tags, ix->keys, ix->order, ix->ids, r0, id0,
                                      (int)n, s));
        ix->row_of_id.resize((size_t)id0 + (size_t)n);
        ix->id_of_row.resize((size_t)(r0 + n));
        for (int64_t i = 0; i < n; ++i) {
            ix->row_of_id[(size_t)(id0 + i)] = r0 + i;
            ix->id_of_row[(size_t)(r0 + i)] = id0 + (uint64_t)i;
            if (out_ids) out_ids[i] = id0 + (uint64_t)i;
        }
        const hipError_t e_small = hipStreamSynchronize(s);  // the block is reused by the next small batch; errors surface here
        drain.armed = false;
        if (e_small != hipSuccess) {
            ix->row_of_id.resize((size_t)id0);
            ix->id_of_row.resize((size_t)r0);
            return fail(LBH_E_DEVICE, "lbh_index_add: %s", hipGetErrorString(e_small));
        }
        ix->n += n;
        ix->max_norm = std::max(ix->max_norm, 1.0);
        return LBH_OK;
    }
    StagedCopier sc(c);  // (after `drain`: its helpers stop before the streams are drained on an error return)
    scp = &sc;
    // The host-side id maps (a million entries per large batch: 1-2 ms) are brought up to date by a thread of their own
    // while this one feeds the copy engine -- with pageable buffers it is busy handing over staged pieces until the last
    // byte has left, and the maps used to wait for that.  The vectors were reserved above (no re-allocation here); every
    // return before the call is committed shrinks them again.
    struct IdMapJob {
        lbh_index* ix;
        int64_t n, r0;
        uint64_t id0;
        uint64_t* out_ids;
        std::thread th;
        bool started = false, committed = false;
        void start() {
            started = true;
            auto fill = [this] {
                ix->row_of_id.resize((size_t)id0 + (size_t)n);  // (capacity reserved by the caller: no allocation, no exception)
                ix->id_of_row.resize((size_t)(r0 + n));
                for (int64_t i = 0; i < n; ++i) {
                    ix->row_of_id[(size_t)(id0 + i)] = r0 + i;
                    ix->id_of_row[(size_t)(r0 + i)] = id0 + (uint64_t)i;
                    if (out_ids) out_ids[i] = id0 + (uint64_t)i;
                }
            };
            bool own_thread = n >= 65536;
            if (own_thread) {
                try { th = std::thread(fill); } catch (const std::system_error&) { own_thread = false; }
            }
            if (!own_thread) fill();
        }
        void join() { if (th.joinable()) th.join(); }
        ~IdMapJob() {
            join();
            if (started && !committed) {
                ix->row_of_id.resize((size_t)id0);
                ix->id_of_row.resize((size_t)r0);
            }
        }
    } idmap{ix, n, r0, id0, out_ids, {}};
    if (emb_f64) {
        batch_norm = 0.0;
        {   // largest norm of the batch (non-finite values make it NaN or inf): a million rows take 30 ms on one thread
            const int nt = n >= 65536 ? 4 : 1;
            double part[4] = {0.0, 0.0, 0.0, 0.0};
            auto work = [&](int t) {
                double m = 0.0;
                bool bad = false;
                for (int64_t i = n * t / nt; i < n * (t + 1) / nt; ++i) {
                    const double v = vec_norm(emb_f64 + i * 64);
                    if (!(v <= 1e15)) bad = true;
                    m = std::max(m, v);
                }
                part[t] = bad ? std::numeric_limits<double>::infinity() : m;
            };
            std::thread th[3];
            bool inl[3] = {false, false, false};
            for (int t = 1; t < nt; ++t) {
                try { th[t - 1] = std::thread(work, t); } catch (const std::system_error&) { inl[t - 1] = true; }
            }
            work(0);
            for (int t = 1; t < nt; ++t) {
                if (inl[t - 1]) work(t);
                else th[t - 1].join();
            }
            for (int t = 0; t < nt; ++t) batch_norm = std::max(batch_norm, part[t]);
        }
        if (!(batch_norm <= 1e15)) return fail(LBH_E_INVALID, "a vector is not finite (or its norm exceeds 1e15)");
        for (int pass = 0; pass < 2; ++pass) {
            planning = pass == 0;
            int rc = copy_metadata(s);
            if (rc) return rc;
            rc = h2d(ix->emb64 + r0 * 64, emb_f64, (size_t)n * 512, s);
            if (rc) return rc;
            if (planning) {
                rc = sc.start();
                if (rc) return rc;
                idmap.start();
            }
        }
        HIP_TRY(lbh_launch_f64_to_f32(ix->emb64 + r0 * 64, ix->emb32 + r0 * 64, n * 64, s));
        int rc = finish_metadata();
        if (rc) return rc;
    } else {
        const int64_t base = offsets[0], bytes = offsets[n] - base;
        bool bad = bytes < 0 || (bytes > 0 && !utf8);
        for (int64_t i = 0; i < n && !bad; ++i) {
            const int64_t l = offsets[i + 1] - offsets[i];
            bad = l < 0 || l >= (int64_t)0x7FFFFF00;
        }
        if (bad) return fail(LBH_E_INVALID, "bad offsets (a text has a negative length or 2 GiB and more)");
        text_lk.lock();
        int rc = ensure_ctx_scratch(c, bytes + 64, n + 1, 0);
        if (rc) return rc;
        const int64_t* offp = offsets;
        if (base != 0) {
            rel.resize((size_t)n + 1);
            for (int64_t i = 0; i <= n; ++i) rel[(size_t)i] = offsets[i] - base;
            offp = rel.data();
        }
        // The batch is cut into slices: slice i + 1's text and offsets cross PCIe on the copy stream while K1 vectorises
        // slice i on the compute stream (offsets stay absolute into the one text blob, so a slice is just a pointer pair).
        // K1 needs none of the metadata columns, so those follow the LAST slice of text on the copy stream and arrive under
        // K1 as well (ahead of the text they kept K1 waiting for 48 bytes per claim: 1.1 of 6.4 ms per million claims).
        // Few, large launches, because each K1 launch pays its own ramp and tail (8 equal slices of a 1 M-claim batch took
        // 8 x 0.61 ms against 4.1 ms in one launch).  A short first slice lets K1 start early; the LAST slice must not be the
        // largest any more: since the token memo K1 outruns the link (2.3 against 2.9 ms per million claims), so what
        // matters is how much K1 work is left when the last text byte has arrived -- 5/16 of the batch hides under the
        // metadata copies, the 9/16 of the round-2 cuts (1, 3, 7 sixteenths: made for a K1 slower than the link) did not:
        // 3.91 -> 3.45-3.85 ms per 1 M-claim call, and 4.60 -> 4.50 with the memo off (tools/ingest_ab.py).
        int cut16[LBH_ADD_SLICES][LBH_ADD_SLICES + 1] = {{0, 16}, {0, 4, 16}, {0, 2, 6, 16}, {0, 2, 6, 11, 16}};
        if (c->add_cuts > 0) {  // option "add_cuts" = a * 10000 + b * 100 + c: the four-slice cuts in sixteenths
            cut16[3][1] = c->add_cuts / 10000 % 100;
            cut16[3][2] = c->add_cuts / 100 % 100;
            cut16[3][3] = c->add_cuts % 100;
        }
        const int n_sl = (int)std::max<int64_t>(1, std::min<int64_t>(LBH_ADD_SLICES, n / 65536));
        for (int pass = 0; pass < 2; ++pass) {
            planning = pass == 0;
            for (int sl = 0; sl < n_sl; ++sl) {
                const int64_t a = n * cut16[n_sl - 1][sl] / 16, b2 = n * cut16[n_sl - 1][sl + 1] / 16;
                const int64_t t0 = offp[a], t1 = offp[b2];
                if (t1 > t0) {
                    rc = h2d(c->d_text + t0, utf8 + base + t0, (size_t)(t1 - t0), c->copy_stream);
                    if (rc) return rc;
                }
                rc = h2d(c->d_off + a, offp + a, (size_t)(b2 - a + 1) * 8, c->copy_stream);
                if (rc) return rc;
                if (!planning) {
                    HIP_TRY(hipEventRecord(c->copy_ev[sl], c->copy_stream));
                    HIP_TRY(hipStreamWaitEvent(s, c->copy_ev[sl], 0));
                    const lbh_memo memo = next_memo(c, b2 - a);  // (text_lk holds c->mu)
                    HIP_TRY(lbh_launch_embed(c->d_text, c->d_off + a, b2 - a, ix->emb64 + (r0 + a) * 64, ix->emb32 + (r0 + a) * 64, s, LBH_DIM, &memo));
                }
                if (sl == n_sl - 1) {
                    rc = copy_metadata(c->copy_stream);
                    if (rc) return rc;
                    if (!planning) {
                        HIP_TRY(hipEventRecord(c->copy_ev[LBH_ADD_SLICES], c->copy_stream));
                        HIP_TRY(hipStreamWaitEvent(s, c->copy_ev[LBH_ADD_SLICES], 0));
                    }
                }
            }
            if (planning) {
                rc = sc.start();
                if (rc) return rc;
                idmap.start();
            }
        }
        rc = finish_metadata();
        if (rc) return rc;
    }
    static const bool add_trace = getenv("LBH_ADD_TRACE") != nullptr;  // debugging aid: where a call's time goes
    const auto t_issued = std::chrono::steady_clock::now();
    idmap.join();
    const auto t_joined = std::chrono::steady_clock::now();
    ix->n += n;
    ix->max_norm = std::max(ix->max_norm, batch_norm);
    // ONE synchronisation per call, after the host-side id maps were brought up to date under the GPU's work: the
    // caller's buffers (and `rel`) are free again when this returns
    const hipError_t e_sync = hipStreamSynchronize(s);
    if (add_trace) {
        const auto t_synced = std::chrono::steady_clock::now();
        auto us = [&](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
            return (long long)std::chrono::duration_cast<std::chrono::microseconds>(b - a).count();
        };
        fprintf(stderr, "lbh_index_add n=%lld: copies issued +%lld us (waiting for staged pieces %lld, in HIP calls %lld), id maps joined +%lld us, streams idle +%lld us\n",
                (long long)n, us(t_enter, t_issued), sc.wait_us, sc.issue_us, us(t_issued, t_joined), us(t_joined, t_synced));
    }
    drain.armed = false;  // `s` waited on every copy-stream event: both streams are idle
    sc.release_after_sync();  // ... and so is the staging ring
    if (e_sync != hipSuccess) {  // the rows are not usable: take them back (idmap's destructor shrinks the maps again)
        ix->n -= n;
        return fail(LBH_E_DEVICE, "lbh_index_add: %s", hipGetErrorString(e_sync));
    }
    idmap.committed = true;
    return LBH_OK;
} LBH_CATCH_ALLOC

int lbh_index_retract(lbh_index* ix, uint64_t id) {
    if (!ix) return fail(LBH_E_INVALID, "bad argument");
    Exclusive lk(ix);
    if (id >= ix->row_of_id.size() || ix->row_of_id[(size_t)id] < 0) return fail(LBH_E_NOTFOUND, "unknown claim id %llu", (unsigned long long)id);
    HIP_TRY(hipSetDevice(ix->ctx->device));
    const int64_t r = ix->row_of_id[(size_t)id], last = ix->n - 1;
    if (r != last) {
        HIP_TRY(lbh_launch_move_row(ix->emb32, ix->emb64, (ix->emb16 && ix->emb16_rows == ix->n) ? ix->emb16 : nullptr,
                                    (ix->emb8 && ix->emb8_rows == ix->n) ? ix->emb8 : nullptr, ix->created, ix->tags, ix->keys, ix->order, ix->ids, r, last, ix->ctx->stream));
        const uint64_t moved = ix->id_of_row[(size_t)last];
        ix->id_of_row[(size_t)r] = moved;
        ix->row_of_id[(size_t)moved] = r;
    }
    ix->row_of_id[(size_t)id] = -1;
    ix->id_of_row.pop_back();
    if (ix->emb16_rows == ix->n) ix->emb16_rows = ix->n - 1;  // the copy followed the swap-delete
    else ix->emb16_rows = 0;                                   // rows were pending conversion: rebuild at next use
    if (ix->emb8_rows == ix->n) ix->emb8_rows = ix->n - 1;
    else ix->emb8_rows = 0;
    ix->n -= 1;
    return LBH_OK;
}

int lbh_index_retract_many(lbh_index* ix, int64_t n, const uint64_t* ids) try {
    if (!ix || n < 0 || (n > 0 && !ids)) return fail(LBH_E_INVALID, "bad argument");
    if (n == 0) return LBH_OK;
    Exclusive lk(ix);
    // rows to delete, ascending; an unknown or repeated id fails the whole call before anything changes
    std::vector<int64_t> dead((size_t)n);
    for (int64_t i = 0; i < n; ++i) {
        if (ids[i] >= ix->row_of_id.size() || ix->row_of_id[(size_t)ids[i]] < 0)
            return fail(LBH_E_NOTFOUND, "unknown claim id %llu", (unsigned long long)ids[i]);
        dead[(size_t)i] = ix->row_of_id[(size_t)ids[i]];
    }
    std::sort(dead.begin(), dead.end());
    for (int64_t i = 1; i < n; ++i)
        if (dead[(size_t)i] == dead[(size_t)i - 1]) return fail(LBH_E_INVALID, "a claim id appears twice in the batch");
    lbh_ctx* c = ix->ctx;
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    const int64_t n_old = ix->n, n_new = n_old - n;
    // holes = deleted rows below the new end; movers = surviving rows at or above it (as many as holes)
    const int64_t n_holes = std::lower_bound(dead.begin(), dead.end(), n_new) - dead.begin();
    int rc = ensure_staging(ix, std::max<int64_t>(n_holes, 1) * 16);
    if (rc) return rc;
    int64_t* pairs = reinterpret_cast<int64_t*>(ix->staging);
    {
        int64_t np = 0;
        size_t di = (size_t)n_holes;  // first deleted row in the tail
        for (int64_t r = n_new; r < n_old && np < n_holes; ++r) {
            if (di < dead.size() && dead[di] == r) { ++di; continue; }
            pairs[2 * np] = dead[(size_t)np];
            pairs[2 * np + 1] = r;
            ++np;
        }
        if (np != n_holes) return fail(LBH_E_INTERNAL, "retract_many: %lld movers for %lld holes", (long long)np, (long long)n_holes);
    }
    if (n_holes > 0) {
        if (ix->emb16 && ix->emb16_rows < ix->n) {  // bring the fp16 copy up to date so that it can follow the moves
            rc = ensure_emb16(ix);
            if (rc) return rc;
        }
        if (ix->emb8 && ix->emb8_rows < ix->n) {  // likewise the int8 copy
            rc = ensure_emb8(ix);
            if (rc) return rc;
        }
        rc = ensure_dscratch(ix, n_holes * 16);
        if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(ix->dscratch, pairs, (size_t)n_holes * 16, hipMemcpyHostToDevice, s));
        HIP_TRY(lbh_launch_move_rows(ix->emb32, ix->emb64, (ix->emb16 && ix->emb16_rows == ix->n) ? ix->emb16 : nullptr,
                                     (ix->emb8 && ix->emb8_rows == ix->n) ? ix->emb8 : nullptr, ix->created,
                                     ix->tags, ix->keys, ix->order, ix->ids, reinterpret_cast<const int64_t*>(ix->dscratch), n_holes, s));
        HIP_TRY(hipStreamSynchronize(s));  // the staging area is reused by the next call
    }
    for (int64_t i = 0; i < n; ++i) ix->row_of_id[(size_t)ids[i]] = -1;
    for (int64_t p = 0; p < n_holes; ++p) {
        const uint64_t moved = ix->id_of_row[(size_t)pairs[2 * p + 1]];
        ix->id_of_row[(size_t)pairs[2 * p]] = moved;
        ix->row_of_id[(size_t)moved] = pairs[2 * p];
    }
    ix->id_of_row.resize((size_t)n_new);
    if (ix->emb16_rows == n_old) ix->emb16_rows = n_new;  // the copy followed the moves
    else ix->emb16_rows = std::min(ix->emb16_rows, n_new);  // only reached without moves: the converted prefix stays valid
    if (ix->emb8_rows == n_old) ix->emb8_rows = n_new;
    else ix->emb8_rows = std::min(ix->emb8_rows, n_new);
    ix->n = n_new;
    return LBH_OK;
} LBH_CATCH_ALLOC

int lbh_index_get_embeddings(lbh_index* ix, int64_t n, const uint64_t* ids, double* out_f64) {
    if (!ix || n < 0 || (n > 0 && (!ids || !out_f64))) return fail(LBH_E_INVALID, "bad argument");
    if (n == 0) return LBH_OK;
    Exclusive lk(ix);
    HIP_TRY(hipSetDevice(ix->ctx->device));
    hipStream_t s = ix->ctx->stream;
    int rc = ensure_staging(ix, n * 8);
    if (rc) return rc;
    int64_t* rows = reinterpret_cast<int64_t*>(ix->staging);
    for (int64_t i = 0; i < n; ++i) {
        if (ids[i] >= ix->row_of_id.size() || ix->row_of_id[(size_t)ids[i]] < 0)
            return fail(LBH_E_NOTFOUND, "unknown claim id %llu", (unsigned long long)ids[i]);
        rows[i] = ix->row_of_id[(size_t)ids[i]];
    }
    const int64_t rows_bytes = (n * 8 + 511) / 512 * 512;
    rc = ensure_dscratch(ix, rows_bytes + n * 512);
    if (rc) return rc;
    int64_t* d_rows = reinterpret_cast<int64_t*>(ix->dscratch);
    double* d_out = reinterpret_cast<double*>(reinterpret_cast<char*>(ix->dscratch) + rows_bytes);
    HIP_TRY(hipMemcpyAsync(d_rows, rows, (size_t)n * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(lbh_launch_gather_rows_f64(ix->emb64, d_rows, n, d_out, s));
    HIP_TRY(hipMemcpyAsync(out_f64, d_out, (size_t)n * 512, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return LBH_OK;
}

int lbh_index_update_meta(lbh_index* ix, uint64_t id, int64_t created_ms, uint64_t tagmask, const uint8_t* hashkey) {
    if (!ix || !hashkey) return fail(LBH_E_INVALID, "bad argument");
    Exclusive lk(ix);
    if (id >= ix->row_of_id.size() || ix->row_of_id[(size_t)id] < 0) return fail(LBH_E_NOTFOUND, "unknown claim id %llu", (unsigned long long)id);
    HIP_TRY(hipSetDevice(ix->ctx->device));
    const int64_t r = ix->row_of_id[(size_t)id];
    uint64_t kw[4];
    for (int w = 0; w < 4; ++w) kw[w] = load_be64(hashkey + 8 * w);
    hipStream_t s = ix->ctx->stream;
    HIP_TRY(hipMemcpyAsync(ix->created + r, &created_ms, 8, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(ix->tags + r, &tagmask, 8, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(ix->keys + r * 4, kw, 32, hipMemcpyHostToDevice, s));
    HIP_TRY(hipStreamSynchronize(s));
    return LBH_OK;
}

int lbh_index_set_keys(lbh_index* ix, int64_t n, const uint64_t* ids, const uint8_t* hashkey) try {
    if (!ix || n < 0 || (n > 0 && (!ids || !hashkey))) return fail(LBH_E_INVALID, "bad argument");
    if (n == 0) return LBH_OK;
    Exclusive lk(ix);
    HIP_TRY(hipSetDevice(ix->ctx->device));
    hipStream_t s = ix->ctx->stream;
    // download, patch, upload: the key table is 32 B/claim and this path is for small graphs
    std::vector<uint64_t> host((size_t)ix->n * 4);
    HIP_TRY(hipMemcpyAsync(host.data(), ix->keys, host.size() * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    for (int64_t i = 0; i < n; ++i) {
        if (ids[i] >= ix->row_of_id.size() || ix->row_of_id[(size_t)ids[i]] < 0)
            return fail(LBH_E_NOTFOUND, "unknown claim id %llu", (unsigned long long)ids[i]);
        const int64_t r = ix->row_of_id[(size_t)ids[i]];
        for (int w = 0; w < 4; ++w) host[(size_t)r * 4 + w] = load_be64(hashkey + i * 32 + 8 * w);
    }
    HIP_TRY(hipMemcpyAsync(ix->keys, host.data(), host.size() * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(hipStreamSynchronize(s));
    return LBH_OK;
} LBH_CATCH_ALLOC

int lbh_index_set_order(lbh_index* ix, int64_t n, const uint64_t* ids, const uint64_t* order) {
    if (!ix || n < 0 || (n > 0 && (!ids || !order))) return fail(LBH_E_INVALID, "bad argument");
    if (n == 0) return LBH_OK;
    Exclusive lk(ix);
    HIP_TRY(hipSetDevice(ix->ctx->device));
    hipStream_t s = ix->ctx->stream;
    int rc = ensure_staging(ix, n * 16);
    if (rc) return rc;
    int64_t* h_rows = reinterpret_cast<int64_t*>(ix->staging);
    uint64_t* h_vals = reinterpret_cast<uint64_t*>(h_rows + n);
    for (int64_t i = 0; i < n; ++i) {
        if (ids[i] >= ix->row_of_id.size() || ix->row_of_id[(size_t)ids[i]] < 0)
            return fail(LBH_E_NOTFOUND, "unknown claim id %llu", (unsigned long long)ids[i]);
        h_rows[i] = ix->row_of_id[(size_t)ids[i]];
        h_vals[i] = order[i];
    }
    rc = ensure_dscratch(ix, n * 16);
    if (rc) return rc;
    void* d = ix->dscratch;
    HIP_TRY(hipMemcpyAsync(d, ix->staging, (size_t)n * 16, hipMemcpyHostToDevice, s));
    HIP_TRY(lbh_launch_scatter_u64(ix->order, reinterpret_cast<const int64_t*>(d), reinterpret_cast<const uint64_t*>(d) + n, n, s));
    HIP_TRY(hipStreamSynchronize(s));
    return LBH_OK;
}

int lbh_index_recent(lbh_index* ix, int64_t since_ms, int64_t limit, uint64_t* out_ids, int64_t* out_created,
                     int64_t* out_count, int64_t* out_matched, lbh_stats* stats) {
    if (!ix || !out_count) return fail(LBH_E_INVALID, "bad argument");
    if (limit < 0) limit = 0;
    if (limit > 0 && !out_ids) return fail(LBH_E_INVALID, "output arrays are NULL");
    return with_lease(ix, [&](lbh_slot* sl) -> int {
    lbh_ctx* c = ix->ctx;
    HIP_TRY(hipSetDevice(c->device));
    *out_count = 0;
    if (out_matched) *out_matched = 0;
    if (stats) {
        const int32_t wt = stats->want_timing;
        memset(stats, 0, sizeof *stats);
        stats->want_timing = wt;
        stats->path = 3;
    }
    if (ix->n == 0) return LBH_OK;
    int rc = ensure_recent_ws(sl);
    if (rc) return rc;
    const bool timing = stats && stats->want_timing;
    const int has_since = since_ms != LBH_SINCE_NONE;
    lbh_rcursor cur;
    memset(&cur, 0, sizeof cur);
    int64_t got = 0, remaining = std::min<int64_t>(limit, ix->n);
    const lbh_recent_result* r = sl->ws.rc_result_host;
    bool first = true;
    int64_t anchor = ix->n;  // where the threshold seed looks: the index tail, then just below the last page's rows
    while (first || remaining > 0) {  // one pass even for limit 0: it still reports `matched`
        const int page = (int)std::max<int64_t>(1, std::min<int64_t>(remaining, LBH_MAX_PAGE));
        const int32_t seq = sl->seq = (sl->seq == INT32_MAX) ? 1 : sl->seq + 1;
        HIP_TRY(lbh_launch_recent(ix->created, ix->order, ix->n, sl->ws, page, has_since, since_ms, cur, anchor, c->n_cu, seq,
                                  sl->stream, timing ? sl->ev[0] : nullptr, timing ? sl->ev[1] : nullptr));
        HIP_TRY(wait_ticket(c, sl->stream, &r->seq, seq));
        if (timing) {
            HIP_TRY(hipEventSynchronize(sl->ev[1]));
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, sl->ev[0], sl->ev[1]) == hipSuccess) stats->scan_ms += ms;
        }
        if (stats) {
            stats->scan_launches += 1;
            stats->rows_scanned = ix->n;
            stats->bytes_per_row = 8;
        }
        if (first && out_matched) *out_matched = r->matched;
        first = false;
        const int take = (int)std::min<int64_t>(r->count, remaining);
        for (int i = 0; i < take; ++i) {
            out_ids[got + i] = ix->id_of_row[r->rows[i]];
            if (out_created) out_created[got + i] = r->created[i];
        }
        got += take;
        remaining -= take;
        if (r->count < page || take == 0) break;
        for (int i = 0; i < r->count; ++i) anchor = std::min<int64_t>(anchor, r->rows[i]);
        cur.active = 1;
        cur.a = (uint64_t)r->created[r->count - 1] ^ 0x8000000000000000ull;
        cur.b = ~r->order[r->count - 1];
    }
    *out_count = got;
    if (stats) stats->candidates = got;
    return LBH_OK;
    });
}

int lbh_merge_topk(const int64_t* lists, int64_t n_lists, int64_t k, int64_t* out_entries, int64_t* out_count) try {
    if (n_lists < 0 || k < 0 || (n_lists > 0 && !lists) || !out_count || (k > 0 && !out_entries))
        return fail(LBH_E_INVALID, "bad argument");
    std::vector<const int64_t*> bases((size_t)n_lists);
    for (int64_t l = 0; l < n_lists; ++l) bases[(size_t)l] = lists + l * (k + 1) * 6;
    return merge_lists(bases.data(), n_lists, k, out_entries, out_count);
} LBH_CATCH_ALLOC

int lbh_index_size(lbh_index* ix, int64_t* out_live) {
    if (!ix || !out_live) return fail(LBH_E_INVALID, "bad argument");
    std::shared_lock<std::shared_mutex> lk(ix->rw);
    *out_live = ix->n;
    return LBH_OK;
}

int lbh_index_get_embedding(lbh_index* ix, uint64_t id, double* out64) {
    if (!ix || !out64) return fail(LBH_E_INVALID, "bad argument");
    Exclusive lk(ix);
    if (id >= ix->row_of_id.size() || ix->row_of_id[(size_t)id] < 0) return fail(LBH_E_NOTFOUND, "unknown claim id %llu", (unsigned long long)id);
    HIP_TRY(hipSetDevice(ix->ctx->device));
    const int64_t r = ix->row_of_id[(size_t)id];
    HIP_TRY(hipMemcpyAsync(out64, ix->emb64 + r * 64, 512, hipMemcpyDeviceToHost, ix->ctx->stream));
    HIP_TRY(hipStreamSynchronize(ix->ctx->stream));
    return LBH_OK;
}

}  // extern "C"

namespace {
// nq queries against one index under a lease (the body of lbh_index_search; lbh_index_search_sharded_many runs it as its
// local part and also wants the hash keys: out_keys = nq x k x 32 bytes, or NULL)
int search_many(lbh_index* ix, lbh_slot* sl, const double* q_f64, int64_t nq, int64_t k, int64_t since_ms, uint64_t tag_mask,
                uint64_t* out_ids, double* out_scores, uint8_t* out_keys, int64_t* out_counts, lbh_stats* stats) {
    HIP_TRY(hipSetDevice(ix->ctx->device));
    lbh_stats total;
    memset(&total, 0, sizeof total);
    const bool timing = stats && stats->want_timing;
    const int64_t keff = std::min<int64_t>(k, ix->n);
    std::vector<char> done((size_t)nq, 0);
    // K3: three or more queries with a small k share ONE pass over the claim matrix on the matrix
    // cores (256 queries per launch).  A query whose candidate set could not be bounded there is
    // re-run below on the single-query path, so results never depend on which path served them.
    // top_k: the fp16 kernel keeps no per-query state in registers, so it serves any k of one result page (<= 64); the
    // fp32 / split-bf16 kernels keep k-entry lists per lane and stop at LBH_BQ_KMAX (larger k: single-query path below)
    // (round 3: the tiled kernels also serve top_k up to LBH_BQ_PAGED_KMAX -- the pass lists every row within the margin of
    // a query's k-th best whatever k is, and the select step is run once per result page of LBH_MAX_PAGE rows)
    if (nq >= 3 && keff >= 1 && keff <= (ix->ctx->bq_mode >= 2 ? LBH_BQ_PAGED_KMAX : LBH_BQ_KMAX) && ix->n > 0) {
        int rc = ensure_batched_ws(sl);
        if (rc) return rc;
        lbh_ctx* c = ix->ctx;
        hipStream_t s = sl->stream;
        const lbh_dev_index dv = dev_view(ix);
        const int has_since = since_ms != LBH_SINCE_NONE;
        bool any16 = false, any8 = false;
        for (int64_t q0 = 0; q0 < nq; q0 += LBH_BQ_MAXQ) {
            const int nb = (int)std::min<int64_t>(LBH_BQ_MAXQ, nq - q0);
            double qn = 0.0;
            for (int q = 0; q < nb; ++q) qn = std::max(qn, vec_norm(q_f64 + (q0 + q) * 64));
            if (!(qn <= 1e15)) return fail(LBH_E_INVALID, "a query vector is not finite (or its norm exceeds 1e15)");
            const float nscale = norm_scale(ix, qn);
            // int8 mode (bq_mode 3): the int8 copy (64 B/claim) on v_mfma_i32_32x32x32_i8.  Its candidate margin comes from
            // the error bound for THIS index and THESE queries (lbh_internal.h): with s the copy's scale and t_q = max|q_i| / 127,
            // eps_q = (s/2)(|q|_1 + 32 t_q) + (t_q/2)(8 max|x| + 32 s) + 16 s t_q; margin = 2 max_q eps_q (+ 0.1 % and 1e-6 for the
            // float arithmetic around it).  The scale is only known once the copy exists, so the copy comes first.
            bool src8 = false;
            float margin8 = 0.f;
            if (c->bq_mode == 3 && qn > 0.0) {
                if (!emb8_current(ix)) return LBH_NEED_EMB8;
                const double sx = ix->emb8_scale;
                double eps = 0.0;
                for (int q = 0; q < nb; ++q) {
                    const double* v = q_f64 + (q0 + q) * 64;
                    double l1 = 0.0, mxq = 0.0;
                    for (int d = 0; d < 64; ++d) { l1 += std::fabs(v[d]); mxq = std::max(mxq, std::fabs(v[d])); }
                    if (mxq == 0.0) continue;  // a zero query scores every claim exactly 0 in either arithmetic
                    const double t = mxq / 127.0;
                    eps = std::max(eps, 0.5 * sx * (l1 + 32.0 * t) + 0.5 * t * (8.0 * ix->max_norm + 32.0 * sx) + 16.0 * sx * t);
                }
                margin8 = (float)(2.0 * eps * 1.001 + 1e-6);
                src8 = margin8 <= LBH_I8_MARGIN_MAX * nscale;  // odd data (one huge component): the fp16 kernel's bound is tighter
            }
            // the fp16 kernel streams the fp16 copy of the matrix (128 B/claim, MFMA-fragment order; built on first use)
            const bool src16 = !src8 && c->bq_mode >= 2 && nscale >= 0.25f && nscale <= 1.001f;
            if (!src8 && !src16 && keff > LBH_BQ_KMAX) continue;  // data outside fp16's range with a large k: single-query path
            const float margin_used = src8 ? margin8 : LBH_MARGIN_F16 * nscale;  // (of the tiled kernels: what a further page's select needs)
            if (src16 && !emb16_current(ix)) return LBH_NEED_EMB16;
            any16 = any16 || src16;
            any8 = any8 || src8;
            total.margin = std::max(total.margin, src8 ? margin8 : (src16 ? LBH_MARGIN_F16 : c->bq_mode >= 1 ? LBH_MARGIN_BF16X3 : LBH_MARGIN) * nscale);
            // The queries: the fp16 mode's first kernel reads them straight from device-mapped pinned memory (one host
            // memcpy, no copy-engine hop, no staging of pageable memory); the other modes take a device copy first.
            const double* q_src = sl->ws.bq_qmat;
            if (src16 || src8) {
                memcpy(sl->ws.bq_qhost, q_f64 + q0 * 64, (size_t)nb * 512);
                q_src = sl->ws.bq_qhost_dev;
            } else {
                HIP_TRY(hipMemcpyAsync(sl->ws.bq_qmat, q_f64 + q0 * 64, (size_t)nb * 512, hipMemcpyHostToDevice, s));
            }
            const int32_t seq = sl->seq = (sl->seq == INT32_MAX) ? 1 : sl->seq + 1;
            HIP_TRY(lbh_launch_search_batched(dv, src16 ? ix->emb16 : nullptr, sl->ws, q_src, nb, (int)keff, has_since, since_ms, tag_mask, c->n_cu,
                                              src8 ? 3 : std::min(c->bq_mode, 2), nscale, seq, s, timing ? sl->ev[0] : nullptr, timing ? sl->ev[1] : nullptr,
                                              src8 ? ix->emb8 : nullptr, (float)ix->emb8_scale, margin8));
            // every query's select block publishes its own ticket into its result block (mapped host memory)
            for (int i = 0; i < nb; ++i) HIP_TRY(wait_ticket(c, s, &sl->ws.bq_result_host[i].seq, seq));
            if (timing) HIP_TRY(hipEventSynchronize(sl->ev[1]));
            total.scan_launches += 1;
            total.rows_scanned += ix->n;
            if (timing) {
                float ms = 0.f;
                if (hipEventElapsedTime(&ms, sl->ev[0], sl->ev[1]) == hipSuccess) total.scan_ms += ms;
            }
            std::vector<char> over((size_t)nb, 0);
            for (int64_t off = 0; off < keff; off += LBH_MAX_PAGE) {
                if (off > 0) {  // a further page of every query: the select step alone, on the lists the pass left
                    const int32_t pseq = sl->seq = (sl->seq == INT32_MAX) ? 1 : sl->seq + 1;
                    HIP_TRY(lbh_launch_select_batched_page(dv, sl->ws, nb, (int)keff, margin_used, pseq, (int)off, s));
                    for (int i = 0; i < nb; ++i) HIP_TRY(wait_ticket(c, s, &sl->ws.bq_result_host[i].seq, pseq));
                }
                for (int i = 0; i < nb; ++i) {
                    const lbh_result* r = sl->ws.bq_result_host + i;
                    if (r->overflow) over[(size_t)i] = 1;
                    if (over[(size_t)i]) continue;  // served by the single-query path below
                    const int64_t q = q0 + i;
                    for (int j = 0; j < r->count; ++j) {
                        out_ids[q * k + off + j] = ix->id_of_row[r->rows[j]];
                        out_scores[q * k + off + j] = r->scores[j];
                        if (out_keys)
                            for (int w = 0; w < 4; ++w) store_be64(out_keys + (q * k + off + j) * 32 + 8 * w, r->keys[j][w]);
                    }
                    if (off == 0 || r->count > 0) out_counts[q] = off + r->count;  // (a page past the last result is empty)
                    if (off == 0) {
                        total.candidates += r->candidates;
                        total.max_score_err = std::max(total.max_score_err, r->max_err);
                    }
                }
            }
            for (int i = 0; i < nb; ++i)
                if (!over[(size_t)i]) done[(size_t)(q0 + i)] = 1;
        }
        total.path = 2;
        total.bytes_per_row = (any8 ? 64 : any16 ? 128 : 256) + (has_since ? 8 : 0) + (tag_mask ? 8 : 0);
    }
    for (int64_t q = 0; q < nq; ++q) {
        if (done[(size_t)q]) continue;
        int rc = load_query(sl, q_f64 + q * 64);
        if (rc) return rc;
        lbh_stats st;
        memset(&st, 0, sizeof st);
        st.want_timing = stats ? stats->want_timing : 0;
        rc = search_loaded(ix, sl, k, since_ms, tag_mask, out_ids + q * k, out_scores + q * k, out_keys ? out_keys + q * k * 32 : nullptr,
                           out_counts + q, stats ? &st : nullptr);
        if (rc) return rc;
        if (stats) {
            if (total.path != 2) {
                total.path = st.path;
                total.bytes_per_row = st.bytes_per_row;
            } else {
                total.fallbacks += 1;
            }
            total.rows_scanned += st.rows_scanned;
            total.candidates += st.candidates;
            total.scan_launches += st.scan_launches;
            total.fallbacks += st.fallbacks;
            total.scan_ms += st.scan_ms;
            total.total_ms += st.total_ms;
            total.max_score_err = std::max(total.max_score_err, st.max_score_err);
        }
    }
    if (stats) {
        total.want_timing = stats->want_timing;
        *stats = total;
    }
    return LBH_OK;
}
}  // namespace

extern "C" {

int lbh_index_search(lbh_index* ix, const double* q_f64, int64_t nq, int64_t k, int64_t since_ms, uint64_t tag_mask,
                     uint64_t* out_ids, double* out_scores, int64_t* out_counts, lbh_stats* stats) try {
    if (!ix || nq < 0 || (nq > 0 && (!q_f64 || !out_counts))) return fail(LBH_E_INVALID, "bad argument");
    if (k < 0) k = 0;
    if (k > 0 && nq > 0 && (!out_ids || !out_scores)) return fail(LBH_E_INVALID, "output arrays are NULL");
    return with_lease(ix, [&](lbh_slot* sl) -> int {
        return search_many(ix, sl, q_f64, nq, k, since_ms, tag_mask, out_ids, out_scores, nullptr, out_counts, stats);
    });
} LBH_CATCH_ALLOC

int lbh_index_search_text(lbh_index* ix, const uint8_t* utf8, int64_t len, int64_t k, int64_t since_ms, uint64_t tag_mask,
                          uint64_t* out_ids, double* out_scores, int64_t* out_count, double* out_qvec64, lbh_stats* stats) {
    if (!ix || len < 0 || (len > 0 && !utf8) || !out_count) return fail(LBH_E_INVALID, "bad argument");
    if (len >= (int64_t)0x7FFFFF00) return fail(LBH_E_INVALID, "query too long");
    if (k < 0) k = 0;
    if (k > 0 && (!out_ids || !out_scores)) return fail(LBH_E_INVALID, "output arrays are NULL");
    return with_lease(ix, [&](lbh_slot* sl) -> int {
    lbh_ctx* c = ix->ctx;
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = sl->stream;
    int rc = slot_text_buffers(sl, len + 64);
    if (rc) return rc;
    int64_t* ho = reinterpret_cast<int64_t*>(sl->staging);
    ho[0] = 0;
    ho[1] = len;
    uint8_t* ht = reinterpret_cast<uint8_t*>(ho + 2);
    if (len) memcpy(ht, utf8, (size_t)len);
    // embed(query) (lb/context_graph.py:104) straight into the search workspace.  A query of ordinary length is read by K1
    // from the mapped staging area itself (a few hundred bytes over PCIe inside the kernel: ~2 us) -- two copy-engine
    // hops ahead of the kernel cost the text entry ~10 us of stream time; long texts take the device copy.
    if (len <= LBH_QUERY_ZERO_COPY_MAX) {
        const int64_t* d_ho = reinterpret_cast<const int64_t*>(sl->staging_dev);
        HIP_TRY(lbh_launch_embed(reinterpret_cast<const uint8_t*>(d_ho + 2), d_ho, 1, sl->ws.q64, nullptr, s));
    } else {
        HIP_TRY(hipMemcpyAsync(sl->d_text, ht, (size_t)len, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(sl->d_off, ho, 16, hipMemcpyHostToDevice, s));
        HIP_TRY(lbh_launch_embed(sl->d_text, sl->d_off, 1, sl->ws.q64, nullptr, s));
    }
    sl->qptr = sl->ws.q64;
    sl->qnorm = 1.0;  // K1 output: a unit vector (or zero)
    if (out_qvec64) {
        HIP_TRY(hipMemcpyAsync(out_qvec64, sl->ws.q64, 512, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
    }
    // no sync otherwise: the scan is stream-ordered behind the embed kernel, and the slot's staging area is only
    // rewritten by its next holder, after this search's final wait
    return search_loaded(ix, sl, k, since_ms, tag_mask, out_ids, out_scores, nullptr, out_count, stats);
    });
}

int lbh_search_dev_noalloc(lbh_ctx* c, const float* emb32_dev, const double* emb64_dev, const int64_t* created_ms_dev,
                           const uint64_t* tagmask_dev, const uint64_t* keywords_dev, int64_t n, double max_row_norm,
                           const double* q_f64, int64_t k, int64_t since_ms, uint64_t tag_mask, int64_t* out_rows_dev,
                           double* out_scores_dev, int64_t* out_count_dev) {
    if (!c || !q_f64 || !out_count_dev || n < 0) return fail(LBH_E_INVALID, "bad argument");
    if (n > LBH_NOALLOC_MAX_ROWS)
        return fail(LBH_E_INVALID, "lbh_search_dev_noalloc serves at most %lld rows per call (got %lld)", (long long)LBH_NOALLOC_MAX_ROWS, (long long)n);
    if (k < 0) k = 0;
    if (k > LBH_MAX_PAGE)
        return fail(LBH_E_INVALID, "lbh_search_dev_noalloc serves top_k <= %d (got %lld): its fixed scratch holds one result page", LBH_MAX_PAGE, (long long)k);
    if (n > 0 && (!emb32_dev || !emb64_dev || !keywords_dev)) return fail(LBH_E_INVALID, "the matrix and key arrays are required");
    if (k > 0 && (!out_rows_dev || !out_scores_dev)) return fail(LBH_E_INVALID, "output arrays are NULL");
    const int has_since = since_ms != LBH_SINCE_NONE;
    if ((has_since && !created_ms_dev) || (tag_mask && !tagmask_dev)) return fail(LBH_E_INVALID, "a filter needs its column");
    if (!(max_row_norm >= 0.0) || !(max_row_norm <= 1e15)) return fail(LBH_E_INVALID, "max_row_norm must be in [0, 1e15]");
    const double qn = vec_norm(q_f64);
    if (!(qn <= 1e15)) return fail(LBH_E_INVALID, "query vector is not finite (or its norm exceeds 1e15)");
    std::lock_guard<std::mutex> lk(c->noalloc_mu);
    HIP_TRY(hipSetDevice(c->device));
    lbh_dev_index dv;
    dv.emb32 = emb32_dev; dv.emb64 = emb64_dev; dv.created = created_ms_dev; dv.tags = tagmask_dev; dv.keys = keywords_dev; dv.ids = nullptr; dv.n = n;
    if (n == 0 || k == 0) {  // nothing to scan: only the count is written (stream-ordered, like everything else here)
        HIP_TRY(hipMemsetAsync(out_count_dev, 0, sizeof(int64_t), c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        return LBH_OK;
    }
    lbh_qarg qa;
    memcpy(qa.q, q_f64, sizeof qa.q);  // travels inside the kernel-argument segment: no H2D copy
    const double sc = std::min(std::max(max_row_norm * qn * (1.0 + 1e-6), 1e-30), 1e30);
    HIP_TRY(lbh_launch_search_noalloc(c->noalloc_scratch, dv, qa, (int)std::min<int64_t>(k, n), has_since, since_ms, tag_mask, c->n_cu, c->scan_nt,
                                      LBH_MARGIN * (float)sc, out_rows_dev, out_scores_dev, out_count_dev, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return LBH_OK;
}

int lbh_index_search_shard(lbh_index* ix, const double* q_f64, int64_t k, int64_t since_ms, uint64_t tag_mask,
                           uint64_t* out_ids, double* out_scores, uint8_t* out_keys, int64_t* out_count, lbh_stats* stats) {
    if (!ix || !q_f64 || !out_count) return fail(LBH_E_INVALID, "bad argument");
    if (k < 0) k = 0;
    if (k > 0 && (!out_ids || !out_scores || !out_keys)) return fail(LBH_E_INVALID, "output arrays are NULL");
    return with_lease(ix, [&](lbh_slot* sl) -> int {
        HIP_TRY(hipSetDevice(ix->ctx->device));
        int rc = load_query(sl, q_f64);
        if (rc) return rc;
        return search_loaded(ix, sl, k, since_ms, tag_mask, out_ids, out_scores, out_keys, out_count, stats);
    });
}

}  // extern "C"

// ---- claim-sharded search: the exchange step behind the C ABI (RCCL all-gather over xGMI) ---------------------
// librccl is opened on first use (dlopen), so a single-GPU deployment needs no RCCL at all.
namespace {
struct RcclId128 { char b[128]; };  // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128), passed by value
struct RcclApi {
    void* lib = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, RcclId128, int) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*CommAbort)(void*) = nullptr;  // optional: tears a communicator down without waiting for its queued collectives
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
RcclApi g_rccl;
std::once_flag g_rccl_once;
std::string g_rccl_err;

int rccl_api() {
    std::call_once(g_rccl_once, [] {
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* nm : names) {
            g_rccl.lib = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
            if (g_rccl.lib) break;
        }
        if (!g_rccl.lib) { g_rccl_err = std::string("cannot open librccl: ") + dlerror(); return; }
        g_rccl.GetUniqueId = reinterpret_cast<decltype(g_rccl.GetUniqueId)>(dlsym(g_rccl.lib, "ncclGetUniqueId"));
        g_rccl.CommInitRank = reinterpret_cast<decltype(g_rccl.CommInitRank)>(dlsym(g_rccl.lib, "ncclCommInitRank"));
        g_rccl.CommDestroy = reinterpret_cast<decltype(g_rccl.CommDestroy)>(dlsym(g_rccl.lib, "ncclCommDestroy"));
        g_rccl.CommAbort = reinterpret_cast<decltype(g_rccl.CommAbort)>(dlsym(g_rccl.lib, "ncclCommAbort"));
        g_rccl.AllGather = reinterpret_cast<decltype(g_rccl.AllGather)>(dlsym(g_rccl.lib, "ncclAllGather"));
        g_rccl.GetErrorString = reinterpret_cast<decltype(g_rccl.GetErrorString)>(dlsym(g_rccl.lib, "ncclGetErrorString"));
        if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.CommDestroy || !g_rccl.AllGather || !g_rccl.GetErrorString)
            g_rccl_err = "librccl lacks an expected symbol";
    });
    if (!g_rccl_err.empty()) return fail(LBH_E_DEVICE, "%s", g_rccl_err.c_str());
    return LBH_OK;
}
#define RCCL_TRY(expr)                                                                                  \
    do {                                                                                                \
        const int r__ = (expr);                                                                         \
        if (r__ != 0) return fail(LBH_E_DEVICE, "%s: %s", #expr, g_rccl.GetErrorString(r__));          \
    } while (0)
constexpr int NCCL_INT8 = 0;  // ncclInt8 / ncclChar
}  // namespace

struct lbh_comm {
    lbh_ctx* ctx = nullptr;
    void* comm = nullptr;
    int rank = 0, world = 1;
    std::mutex mu;
    bool broken = false;               // a collective timed out (a peer never arrived): nothing more can be issued on this communicator
    int64_t* send_dev = nullptr;       // LBH_XCHG_WORDS: this rank's packed list, written by the select kernel
    int64_t* recv_dev = nullptr;       // world x LBH_XCHG_WORDS
    int64_t* recv_host = nullptr;      // the same, device-mapped pinned host memory, + one ticket word behind it
    int64_t* recv_host_dev = nullptr;
    int32_t seq = 0;
    // staged exchange (top_k > LBH_MAX_PAGE, batched queries, lbh_comm_allgather): lists cross through pinned host buffers
    int64_t* big_send_host = nullptr;  // pinned
    int64_t* big_recv_host = nullptr;  // pinned
    int64_t* big_send_dev = nullptr;
    int64_t* big_recv_dev = nullptr;
    int64_t big_bytes = 0;             // capacity of big_send_* (big_recv_*: world times as much)
};

namespace {

// one rank's failed local search travels as a NEGATIVE count word of its (first) list: every rank takes part in the
// all-gather, every rank sees the code, every rank returns an error -- nobody is left waiting inside the collective
int report_peer_failure(const lbh_comm* cm, int r, int64_t code, const std::string& own_error) {
    if (r == cm->rank && !own_error.empty()) {
        g_err = own_error;
        return (int)code;
    }
    if (code == LBH_E_INTERNAL)
        return fail(LBH_E_INTERNAL, "rank %d: its local search failed (e.g. more than %d claims tie on score AND carry the same 32-byte hash key)", r, LBH_MCAP);
    return fail(LBH_E_DEVICE, "rank %d: its local search failed with code %lld; every rank fails this query together", r, (long long)code);
}

int ensure_staged(lbh_comm* cm, int64_t bytes) {
    if (bytes <= cm->big_bytes) return LBH_OK;
    void* dev[] = {cm->big_send_dev, cm->big_recv_dev};
    for (void* p : dev)
        if (p) (void)hipFree(p);
    void* host[] = {cm->big_send_host, cm->big_recv_host};
    for (void* p : host)
        if (p) (void)hipHostFree(p);
    cm->big_send_dev = cm->big_recv_dev = cm->big_send_host = cm->big_recv_host = nullptr;
    cm->big_bytes = 0;
    const size_t cap = (size_t)std::max<int64_t>(bytes, 1 << 16), W = (size_t)cm->world;
    HIP_TRY(hipMalloc((void**)&cm->big_send_dev, cap));
    HIP_TRY(hipMalloc((void**)&cm->big_recv_dev, cap * W));
    HIP_TRY(hipHostMalloc((void**)&cm->big_send_host, cap, hipHostMallocDefault));
    HIP_TRY(hipHostMalloc((void**)&cm->big_recv_host, cap * W, hipHostMallocDefault));
    cm->big_bytes = (int64_t)cap;
    return LBH_OK;
}

void sleepy_host_fn(void* ms) {  // test hook "inject_fault" = 2: holds the stream up like a peer that is late
    struct timespec ts = {(time_t)((intptr_t)ms / 1000), (long)((intptr_t)ms % 1000) * 1000000L};
    nanosleep(&ts, nullptr);
}

// big_send_host (bytes, already filled) -> all ranks' copies in big_recv_host, through device buffers; bounded wait
int staged_allgather(lbh_comm* cm, hipStream_t s, size_t bytes) {
    lbh_ctx* c = cm->ctx;
    if (c->inject_fault == 2) {
        c->inject_fault = 0;
        HIP_TRY(hipLaunchHostFunc(s, sleepy_host_fn, (void*)(intptr_t)(4 * (int64_t)c->coll_timeout_ms)));
    }
    HIP_TRY(hipMemcpyAsync(cm->big_send_dev, cm->big_send_host, bytes, hipMemcpyHostToDevice, s));
    RCCL_TRY(g_rccl.AllGather(cm->big_send_dev, cm->big_recv_dev, bytes, NCCL_INT8, cm->comm, s));
    HIP_TRY(hipMemcpyAsync(cm->big_recv_host, cm->big_recv_dev, bytes * (size_t)cm->world, hipMemcpyDeviceToHost, s));
    const int rc = wait_collective(c, s, nullptr, 0);
    if (rc) cm->broken = true;
    return rc;
}

}  // namespace

extern "C" {

int lbh_comm_unique_id(uint8_t* out_id) {
    if (!out_id) return fail(LBH_E_INVALID, "bad argument");
    int rc = rccl_api();
    if (rc) return rc;
    RCCL_TRY(g_rccl.GetUniqueId(out_id));
    return LBH_OK;
}

int lbh_comm_destroy(lbh_comm* cm) {
    if (!cm) return LBH_OK;
    (void)hipSetDevice(cm->ctx->device);
    if (cm->broken) {
        // a collective is still queued behind a peer that never came: neither the stream nor ncclCommDestroy would return.
        // Abort the communicator where RCCL offers it and leave the buffers to the process's exit.
        if (cm->comm && g_rccl.CommAbort) (void)g_rccl.CommAbort(cm->comm);
        delete cm;
        return LBH_OK;
    }
    (void)hipStreamSynchronize(cm->ctx->stream);
    if (cm->comm) (void)g_rccl.CommDestroy(cm->comm);
    void* dev[] = {cm->send_dev, cm->recv_dev, cm->big_send_dev, cm->big_recv_dev};
    for (void* p : dev)
        if (p) (void)hipFree(p);
    void* host[] = {cm->recv_host, cm->big_send_host, cm->big_recv_host};
    for (void* p : host)
        if (p) (void)hipHostFree(p);
    delete cm;
    return LBH_OK;
}

int lbh_comm_init(lbh_ctx* c, const uint8_t* id, int rank, int world, lbh_comm** out_comm) {
    if (!c || !id || !out_comm || world < 1 || rank < 0 || rank >= world) return fail(LBH_E_INVALID, "bad argument");
    *out_comm = nullptr;
    int rc = rccl_api();
    if (rc) return rc;
    HIP_TRY(hipSetDevice(c->device));
    lbh_comm* cm = new (std::nothrow) lbh_comm();
    if (!cm) return fail(LBH_E_NOMEM, "out of host memory");
    cm->ctx = c;
    cm->rank = rank;
    cm->world = world;
    auto bail = [&](int code) {
        const std::string why = g_err;
        lbh_comm_destroy(cm);
        g_err = why;
        return code;
    };
    // everything that can fail on ONE rank alone comes before the collective call, so that a rank which cannot take part
    // fails here, before its peers are inside ncclCommInitRank waiting for it (callers agree on these results first:
    // bench.py)
    const size_t words = (size_t)LBH_XCHG_WORDS;
    hipError_t e = hipMalloc((void**)&cm->send_dev, words * 8);
    if (e == hipSuccess) e = hipMalloc((void**)&cm->recv_dev, words * 8 * (size_t)world);
    if (e == hipSuccess) e = hipHostMalloc((void**)&cm->recv_host, words * 8 * (size_t)world + 64, hipHostMallocMapped | hipHostMallocCoherent);
    if (e == hipSuccess) e = hipHostGetDevicePointer((void**)&cm->recv_host_dev, cm->recv_host, 0);
    if (e == hipSuccess) e = hipMemset(cm->send_dev, 0, words * 8);
    if (e != hipSuccess) return bail(fail(e == hipErrorOutOfMemory ? LBH_E_NOMEM : LBH_E_DEVICE, "exchange buffers: %s", hipGetErrorString(e)));
    memset(cm->recv_host, 0, words * 8 * (size_t)world + 64);
    RcclId128 uid;
    memcpy(uid.b, id, sizeof uid.b);
    {
        const int r = g_rccl.CommInitRank(&cm->comm, world, uid, rank);
        if (r != 0) {
            cm->comm = nullptr;
            return bail(fail(LBH_E_DEVICE, "ncclCommInitRank(rank %d of %d): %s", rank, world, g_rccl.GetErrorString(r)));
        }
    }
    *out_comm = cm;
    return LBH_OK;
}

int lbh_comm_allgather(lbh_comm* cm, const void* send, int64_t bytes, void* recv) {
    if (!cm || bytes < 0 || (bytes > 0 && (!send || !recv))) return fail(LBH_E_INVALID, "bad argument");
    if (bytes == 0) return LBH_OK;
    if (bytes > ((int64_t)1 << 30)) return fail(LBH_E_INVALID, "lbh_comm_allgather moves at most 1 GiB per rank");
    std::lock_guard<std::mutex> lk(cm->mu);
    if (cm->broken) return fail(LBH_E_DEVICE, "this communicator is unusable: an earlier collective timed out");
    lbh_ctx* c = cm->ctx;
    HIP_TRY(hipSetDevice(c->device));
    int rc = ensure_staged(cm, bytes);
    if (rc) return rc;
    memcpy(cm->big_send_host, send, (size_t)bytes);
    rc = staged_allgather(cm, c->stream, (size_t)bytes);
    if (rc) return rc;
    memcpy(recv, cm->big_recv_host, (size_t)bytes * (size_t)cm->world);
    return LBH_OK;
}

int lbh_index_search_sharded(lbh_index* ix, lbh_comm* cm, const double* q_f64, int64_t k, int64_t since_ms, uint64_t tag_mask,
                             uint64_t gid_mul, uint64_t gid_add, uint64_t* out_gids, double* out_scores, uint8_t* out_keys,
                             int64_t* out_count, lbh_stats* stats) try {
    if (!ix || !cm || !q_f64 || !out_count) return fail(LBH_E_INVALID, "bad argument");
    if (cm->ctx != ix->ctx) return fail(LBH_E_INVALID, "the communicator and the index belong to different contexts");
    if (k < 0) k = 0;
    if (k > 0 && (!out_gids || !out_scores)) return fail(LBH_E_INVALID, "output arrays are NULL");
    *out_count = 0;
    // collectives of one communicator are issued in the same order on every rank: one sharded search at a time per
    // communicator (taken before the index lease; mutators never take it)
    std::lock_guard<std::mutex> lk2(cm->mu);
    if (cm->broken) return fail(LBH_E_DEVICE, "this communicator is unusable: an earlier collective timed out");
    return with_lease(ix, [&](lbh_slot* sl) -> int {
    lbh_ctx* c = ix->ctx;
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = sl->stream;
    // Argument errors (a non-finite query) are the same on every rank -- the call is collective, same arguments
    // everywhere -- so returning here leaves nobody behind.
    int rc = load_query(sl, q_f64);
    if (rc) return rc;
    if (stats) {
        const int32_t wt = stats->want_timing;
        memset(stats, 0, sizeof *stats);
        stats->want_timing = wt;
    }
    if (k == 0) return LBH_OK;  // every rank returns here: no collective for an empty request
    const bool timing = stats && stats->want_timing;
    const int W = cm->world;
    std::vector<int64_t> merged;
    std::string own_error;  // what went wrong in THIS rank's local part (the exchange still runs)
    const int64_t* lists = nullptr;
    if (k <= LBH_MAX_PAGE) {
        // Stream-ordered, one host wait: fp32 scan -> select (writes the packed list to cm->send_dev) -> exact fp64 scan
        // and select, which return at once unless the fast path reported an unbounded candidate set -> all-gather of the
        // device-resident lists over xGMI -> copy into mapped host memory + ticket.
        const lbh_dev_index dv = dev_view(ix);
        const int has_since = since_ms != LBH_SINCE_NONE;
        const int page = (int)k;
        const lbh_xchg xc{cm->send_dev, gid_mul, gid_add};
        const float nscale = norm_scale(ix, sl->qnorm);
        const int32_t seq = sl->seq = (sl->seq == INT32_MAX) ? 1 : sl->seq + 1;
        const bool half = ix->n > 0 && c->scan_mode == 1 && nscale <= 1.001f && nscale >= 0.25f;
        if (half && !emb16_current(ix)) return LBH_NEED_EMB16;  // nothing queued yet: rebuilt under the exclusive lock, then retried
        // From the first enqueue on, no rank may leave before the all-gather: a local failure is recorded, the list's
        // count word carries its (negative) code, and the exchange runs regardless.
        auto local = [&]() -> int {
            if (c->inject_fault == 1) {
                c->inject_fault = 0;
                return fail(LBH_E_DEVICE, "injected fault (test hook): the local search of this rank fails");
            }
            HIP_TRY(hipMemsetAsync(cm->send_dev, 0, 48, s));  // count 0 if this shard is empty / nothing passes
            if (ix->n == 0) return LBH_OK;
            const int grid = lbh_scan_grid(ix->n, c->n_cu, c->scan_blocks_per_cu);
            HIP_TRY(lbh_launch_search_fast(dv, half ? ix->emb16 : nullptr, sl->ws, nullptr, sl->qarg, page, has_since, since_ms, tag_mask,
                                           (half && !has_since && !tag_mask) ? lbh_scan_grid(ix->n, c->n_cu, c->scan_blocks_per_cu, true) : grid,
                                           c->n_cu, c->scan_nt, seq, (half ? LBH_MARGIN_F16 : LBH_MARGIN) * nscale, s, timing ? sl->ev[0] : nullptr,
                                           timing ? sl->ev[1] : nullptr, &xc, c->scan_fuse, c->scan_xb_k));
            lbh_cursor cur;
            memset(&cur, 0, sizeof cur);
            HIP_TRY(lbh_launch_search_exact(dv, sl->ws, nullptr, sl->qarg, page, has_since, since_ms, tag_mask, cur, std::min(grid, 512), seq, s,
                                            nullptr, nullptr, &xc, &sl->ws.result_dev->overflow));
            if (stats) {
                stats->rows_scanned = ix->n;
                stats->bytes_per_row = (half ? 128 : 256) + (has_since ? 8 : 0) + (tag_mask ? 8 : 0);
                stats->scan_launches = 1;
            }
            return LBH_OK;
        };
        const int lrc = local();
        if (lrc) {
            own_error = g_err;
            static const int64_t code_words[4] = {LBH_E_DEVICE, 0, 0, 0};  // any local failure travels as LBH_E_DEVICE
            // pageable source: the runtime stages these 8 bytes before the call returns
            if (hipMemcpyAsync(cm->send_dev, code_words, 8, hipMemcpyHostToDevice, s) != hipSuccess) {
                cm->broken = true;  // this rank cannot even announce its failure: its peers run into their timeout
                g_err = own_error;
                return lrc;
            }
        }
        if (c->inject_fault == 2) {
            c->inject_fault = 0;
            HIP_TRY(hipLaunchHostFunc(s, sleepy_host_fn, (void*)(intptr_t)(4 * (int64_t)c->coll_timeout_ms)));
        }
        {
            const int r = g_rccl.AllGather(cm->send_dev, cm->recv_dev, (size_t)LBH_XCHG_WORDS * 8, NCCL_INT8, cm->comm, s);
            if (r != 0) {
                cm->broken = true;
                return fail(LBH_E_DEVICE, "ncclAllGather: %s", g_rccl.GetErrorString(r));
            }
        }
        const int32_t xseq = cm->seq = (cm->seq == INT32_MAX) ? 1 : cm->seq + 1;
        int32_t* ticket_host = reinterpret_cast<int32_t*>(cm->recv_host + (size_t)LBH_XCHG_WORDS * W);
        int32_t* ticket_dev = reinterpret_cast<int32_t*>(cm->recv_host_dev + (size_t)LBH_XCHG_WORDS * W);
        hipError_t pe = lbh_launch_xchg_publish(cm->recv_dev, cm->recv_host_dev, LBH_XCHG_WORDS * W, ticket_dev, xseq, s);
        if (pe != hipSuccess) {
            cm->broken = true;
            return fail(LBH_E_DEVICE, "xchg_publish_kernel: %s", hipGetErrorString(pe));
        }
        rc = wait_collective(c, s, ticket_host, xseq);
        if (rc) {
            cm->broken = true;
            return rc;
        }
        for (int r = 0; r < W; ++r) {
            const int64_t head = cm->recv_host[(size_t)r * LBH_XCHG_WORDS];
            if (head < 0) return report_peer_failure(cm, r, head == -1 ? LBH_E_INTERNAL : head, own_error);  // -1: the select kernels' duplicate-key report
        }
        if (timing && ix->n > 0) {
            HIP_TRY(hipEventSynchronize(sl->ev[1]));
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, sl->ev[0], sl->ev[1]) == hipSuccess) stats->scan_ms = ms;
        }
        if (stats && ix->n > 0) {
            const lbh_result* r = sl->ws.result_host;
            stats->candidates = r->candidates;
        }
        // lists are LBH_MAX_PAGE + 1 rows apart; the merge wants (k + 1) rows: compact on the host copy
        merged.resize((size_t)W * (size_t)(k + 1) * 6);
        for (int r = 0; r < W; ++r)
            memcpy(merged.data() + (size_t)r * (size_t)(k + 1) * 6, cm->recv_host + (size_t)r * LBH_XCHG_WORDS, (size_t)(k + 1) * 48);
        lists = merged.data();
    } else {
        // top_k beyond one device page: the local pages are collected on the host (fast page + exact pages), then exchanged
        // through staging buffers.  Rare, and dominated by the paged scans anyway.
        if (k > ((int64_t)1 << 24)) return fail(LBH_E_INVALID, "claim-sharded search: top_k beyond 2^24");  // the same on every rank
        const size_t bytes = (size_t)(k + 1) * 48;
        rc = ensure_staged(cm, (int64_t)bytes);  // an allocation failure here leaves the peers to their timeout
        if (rc) {
            cm->broken = true;
            return rc;
        }
        std::vector<uint64_t> ids((size_t)k);
        std::vector<double> sc((size_t)k);
        std::vector<uint8_t> keys((size_t)k * 32);
        int64_t got = 0;
        int lrc = LBH_OK;
        if (c->inject_fault == 1) {
            c->inject_fault = 0;
            lrc = fail(LBH_E_DEVICE, "injected fault (test hook): the local search of this rank fails");
        } else {
            lrc = search_loaded(ix, sl, k, since_ms, tag_mask, ids.data(), sc.data(), keys.data(), &got, stats);
        }
        int64_t* sh = cm->big_send_host;
        memset(sh, 0, bytes);
        if (lrc) {
            own_error = g_err;
            sh[0] = lrc;
        } else {
            sh[0] = got;
            for (int64_t i = 0; i < got; ++i) {
                int64_t* e = sh + 6 * (i + 1);
                memcpy(&e[0], &sc[(size_t)i], 8);
                e[1] = (int64_t)(ids[(size_t)i] * gid_mul + gid_add);
                for (int w = 0; w < 4; ++w) e[2 + w] = (int64_t)load_be64(keys.data() + i * 32 + 8 * w);
            }
        }
        rc = staged_allgather(cm, s, bytes);
        if (rc) return rc;
        lists = cm->big_recv_host;
        for (int r = 0; r < W; ++r) {
            const int64_t head = lists[(size_t)r * (size_t)(k + 1) * 6];
            if (head < 0) return report_peer_failure(cm, r, head, own_error);
        }
    }
    std::vector<int64_t> top((size_t)std::max<int64_t>(k, 1) * 6);
    int64_t cnt = 0;
    rc = lbh_merge_topk(lists, W, k, top.data(), &cnt);
    if (rc) return rc;
    for (int64_t i = 0; i < cnt; ++i) {
        const int64_t* e = top.data() + i * 6;
        memcpy(&out_scores[i], &e[0], 8);
        out_gids[i] = (uint64_t)e[1];
        if (out_keys)
            for (int w = 0; w < 4; ++w) store_be64(out_keys + i * 32 + 8 * w, (uint64_t)e[2 + w]);
    }
    *out_count = cnt;
    return LBH_OK;
    });
} LBH_CATCH_ALLOC

int lbh_index_search_sharded_many(lbh_index* ix, lbh_comm* cm, const double* q_f64, int64_t nq, int64_t k, int64_t since_ms,
                                  uint64_t tag_mask, uint64_t gid_mul, uint64_t gid_add, uint64_t* out_gids, double* out_scores,
                                  uint8_t* out_keys, int64_t* out_counts, lbh_stats* stats) try {
    if (!ix || !cm || nq < 0 || (nq > 0 && (!q_f64 || !out_counts))) return fail(LBH_E_INVALID, "bad argument");
    if (cm->ctx != ix->ctx) return fail(LBH_E_INVALID, "the communicator and the index belong to different contexts");
    if (k < 0) k = 0;
    if (k > 0 && nq > 0 && (!out_gids || !out_scores)) return fail(LBH_E_INVALID, "output arrays are NULL");
    for (int64_t q = 0; q < nq; ++q) out_counts[q] = 0;
    if (stats) {
        const int32_t wt = stats->want_timing;
        memset(stats, 0, sizeof *stats);
        stats->want_timing = wt;
    }
    if (nq == 0 || k == 0) return LBH_OK;  // on every rank alike: no collective for an empty request
    const int64_t per = (k + 1) * 6;  // int64 words of one query's list
    if (nq > ((int64_t)1 << 20) || k > ((int64_t)1 << 24) || nq * per * 8 > ((int64_t)1 << 30))
        return fail(LBH_E_INVALID, "claim-sharded batch: nq x (top_k + 1) x 48 bytes exceeds 1 GiB per rank");
    std::lock_guard<std::mutex> lk2(cm->mu);
    if (cm->broken) return fail(LBH_E_DEVICE, "this communicator is unusable: an earlier collective timed out");
    return with_lease(ix, [&](lbh_slot* sl) -> int {
    lbh_ctx* c = ix->ctx;
    HIP_TRY(hipSetDevice(c->device));
    const int W = cm->world;
    const size_t bytes = (size_t)(nq * per) * 8;
    // local part: the batched (matrix-core) search of lbh_index_search on this rank's shard, keys included
    std::vector<uint64_t> ids((size_t)(nq * k));
    std::vector<double> sc((size_t)(nq * k));
    std::vector<uint8_t> keys((size_t)(nq * k) * 32);
    std::vector<int64_t> cnts((size_t)nq, 0);
    int lrc;
    if (c->inject_fault == 1) {
        c->inject_fault = 0;
        lrc = fail(LBH_E_DEVICE, "injected fault (test hook): the local search of this rank fails");
    } else {
        lrc = search_many(ix, sl, q_f64, nq, k, since_ms, tag_mask, ids.data(), sc.data(), keys.data(), cnts.data(), stats);
    }
    if (lrc == LBH_NEED_EMB16 || lrc == LBH_NEED_EMB8) return lrc;  // nothing of the exchange has been queued: fp16 copy rebuilt, body retried
    int rc = ensure_staged(cm, (int64_t)bytes);  // an allocation failure here leaves the peers to their timeout
    if (rc) {
        cm->broken = true;
        return rc;
    }
    // A local failure does not excuse this rank from the all-gather: the first list's count word carries the code and
    // every rank fails the batch together.
    std::string own_error;
    int64_t* sh = cm->big_send_host;
    memset(sh, 0, bytes);
    if (lrc) {
        own_error = g_err;
        sh[0] = lrc;
    } else {
        for (int64_t q = 0; q < nq; ++q) {
            int64_t* base = sh + q * per;
            base[0] = cnts[(size_t)q];
            for (int64_t i = 0; i < cnts[(size_t)q]; ++i) {
                int64_t* e = base + 6 * (i + 1);
                memcpy(&e[0], &sc[(size_t)(q * k + i)], 8);
                e[1] = (int64_t)(ids[(size_t)(q * k + i)] * gid_mul + gid_add);
                for (int w = 0; w < 4; ++w) e[2 + w] = (int64_t)load_be64(keys.data() + (size_t)(q * k + i) * 32 + 8 * w);
            }
        }
    }
    rc = staged_allgather(cm, sl->stream, bytes);  // ONE all-gather for the whole batch
    if (rc) return rc;
    const int64_t* all = cm->big_recv_host;
    for (int r = 0; r < W; ++r) {
        const int64_t head = all[(size_t)r * (size_t)(nq * per)];
        if (head < 0) return report_peer_failure(cm, r, head, own_error);
    }
    std::vector<const int64_t*> bases((size_t)W);
    std::vector<int64_t> top((size_t)k * 6);
    for (int64_t q = 0; q < nq; ++q) {
        for (int r = 0; r < W; ++r) bases[(size_t)r] = all + (size_t)r * (size_t)(nq * per) + (size_t)(q * per);
        int64_t cnt = 0;
        rc = merge_lists(bases.data(), W, k, top.data(), &cnt);
        if (rc) return rc;
        for (int64_t i = 0; i < cnt; ++i) {
            const int64_t* e = top.data() + i * 6;
            memcpy(&out_scores[q * k + i], &e[0], 8);
            out_gids[q * k + i] = (uint64_t)e[1];
            if (out_keys)
                for (int w = 0; w < 4; ++w) store_be64(out_keys + (size_t)(q * k + i) * 32 + 8 * w, (uint64_t)e[2 + w]);
        }
        out_counts[q] = cnt;
    }
    return LBH_OK;
    });
} LBH_CATCH_ALLOC

}  // extern "C"
