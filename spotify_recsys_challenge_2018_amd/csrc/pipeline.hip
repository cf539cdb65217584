// pipeline.hip -- the drivers' loop of the reference (main_challenge.py:72-93, main_train.py:62-96: read a batch, feed it,
// fetch y_pred, rank, next batch) as a C-side streaming pipeline: HOST feeds in, HOST top-k lists out.
//
// What the Python loop of round 3 (models/DAEs.py recommend_iter) did per launch -- stage the COO feed in pinned memory,
// upload, build the CSR and the seed lists on the device, score, fetch -- took ~0.3 ms of interpreter time against 0.05 -
// 0.2 ms of kernels, and threads did not help it (the GIL).  Here the caller's thread only copies a feed into a pinned
// staging buffer (dae_pipeline_submit) and picks finished lists up (dae_pipeline_poll: pointers into pinned result blocks,
// no copy); a library-owned thread issues every launch -- H2D, the feed -> CSR + seed lists, dae_score_topk, the lists' stores --
// on one of `lanes` contexts / streams that take the launches in turn and share ONE packed decoder image.  Consecutive
// feeds are scored in one launch of up to `group_rows` rows: rows are scored independently, so every feed gets the bits
// dae_score_topk returns for it alone.  Seeds are the playlist's own tracks (what both reference drivers pass).
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>

#include "dae_internal.h"

namespace {

struct Feed { uint64_t ticket; int row0, n_rows; };

struct OutBlock {                 // pinned result block of one launch: idx [rows][k] (+ score), handed out by pointer
    int32_t* idx = nullptr; float* score = nullptr;
    int refs = 0;                 // feeds handed out and not yet released (+1 while its launch is pending)
};

struct Lane {                     // a context + stream + the device buffers its launches reuse in stream order
    dae_ctx* ctx = nullptr;
    hipStream_t stream = nullptr;
    float* d_score = nullptr;     // scores nobody fetches
    bool has_f32 = false;         // (guard fallback) an fp32 image of its own
    // titled pipelines (dae_pipeline_create_titled): the title scorer's context on the same stream, its guard words as the
    // launch left them, the count the previous polled launch of this lane saw (the words are cumulative)
    dae_ctx* tctx = nullptr;
    int32_t* d_guard = nullptr;
    int32_t guard_seen = 0;
    bool t_has_f32 = false;
};

struct TitleW {                   // the title scorer's variables (caller-owned device arrays) and shapes
    const float *emb = nullptr, *conv_w = nullptr, *conv_b = nullptr, *out_WT = nullptr, *out_b = nullptr;
    int n_char = 0, E = 0, n_sizes = 0, F = 0, ld_feat = 0, L = 0;
    std::vector<int32_t> fs;
};

struct Slot {                     // one launch from staging to its last polled feed; more slots than lanes, so that the caller
    int64_t* h_pos = nullptr;     // stages launch n + lanes while the lanes still score / hand out the ones before
    float* h_val = nullptr;       // (pinned staging of the feed)
    int32_t* h_flags = nullptr;   // pinned: {csr status, guard violations, guard column}
    int64_t* d_pos = nullptr; float* d_val = nullptr;      // the feed on the device (uploaded on the copy stream, ahead of the lane)
    // the launch's CSR and seed lists (plain launches), built on the PREP stream while the lane still scores its previous launch
    int32_t *d_rp = nullptr, *d_col = nullptr, *d_srp = nullptr, *d_scol = nullptr, *d_status = nullptr; float* d_cval = nullptr;
    float *t_h = nullptr, *t_feat = nullptr, *t_wt = nullptr, *t_wp = nullptr;   // titled: hidden rows, title features, mixing weights
    hipEvent_t ev_prep = nullptr;
    int32_t* d_idx = nullptr; float* d_score = nullptr;    // the launch's lists on the device (moved out on the OUT stream, round 6)
    int32_t* d_flags = nullptr;                            // {csr status, guard violations, guard column} as the launch left them
    hipEvent_t ev_scored = nullptr;                        // the scoring call's last kernel (the out stream waits for it)
    bool sync_fetch = true;                                // ev_fetch was recorded by issue() (modes 0, 1): the wait ends on it
    int32_t* h_titles = nullptr; float* h_use = nullptr;   // titled pipelines: [group_rows][L] characters, [group_rows] titles_use
    int32_t* d_titles = nullptr; float* d_use = nullptr;
    int titled = 0;               // this launch ranks the title-mixed score (its feeds came through dae_pipeline_submit_titled)
    hipEvent_t dbg_t0 = nullptr, dbg_t1 = nullptr; bool dbg_used = false;      // experiments build (DAE_DBG_PIPE)
    unsigned polls = 0;           // non-waiting polls of this issue that found its word missing (every 256th asks the runtime)
    int32_t seq = 0;              // the sequence word this launch's last kernel writes into h_flags[3] (the caller's wait watches it)
    int ran_dtype = 0;            // arithmetic the launch was issued with (a paused exact mode issues DAE_DTYPE_F32)
    hipEvent_t ev_fetch = nullptr, ev_h2d = nullptr, ev_gate = nullptr;
    int state = 0;                // 0 free, 1 staging (open launch), 2 queued for the worker, 3 issued (ev_fetch recorded)
    int rows = 0; int64_t nnz = 0;
    int block = -1, lane = -1;
    std::vector<Feed> feeds;
    size_t next_feed = 0;         // first feed of the launch not handed out yet
};

}  // namespace

struct dae_pipeline {
    int device = 0, V = 0, H = 0, n_tracks = 0, dtype = 0, k = 0, group_rows = 0, want_scores = 0;
    int64_t max_nnz = 0;
    const float *W_enc = nullptr, *b_enc = nullptr, *W_dec = nullptr, *b_dec = nullptr;
    std::vector<Lane> lanes;
    std::vector<Slot> slots;
    std::vector<OutBlock> blocks;
    // the feed -> CSR + seed lists of launch n + 1 on a stream (and library context) of their own: five small launches that are
    // independent of the lane's previous launch -- in the lane's stream they were 170 us of a 600 us chain (queued behind the
    // other lanes' decode workgroups, each takes 5 - 10 x its time alone: profiles/r06_notes.md)
    hipStream_t prep_stream = nullptr;
    dae_ctx* prep_ctx = nullptr;
    dae_ctx* prep_tctx = nullptr;         // titled pipelines: the title scorer's context of the prep stream (its convolution table)
    hipStream_t out_stream = nullptr;     // the lists' way out (out_mode 2): the out thread's copies
    hipStream_t copy_stream = nullptr;    // uploads: hipMemcpyAsync on a stream that still has kernels queued blocks its caller
                                          // until they have run (measured: 0.43 ms per launch next to the fp32 decode) -- on a
                                          // stream of their own the uploads run ahead and the lane waits for their event
    std::mutex mu;                // states, queue, blocks
    std::mutex issue_mu;          // the lanes' contexts are used by one thread at a time (worker; poll's fp32 re-run)
    std::condition_variable cv_worker, cv_caller;
    std::deque<int> queue;        // slots waiting for the worker, in submission order
    std::thread worker;
    bool stop = false;
    // how a launch's lists reach the host (round 6, measured in profiles/r06_notes.md 5: M playlists/s through the loop, exact /
    // bf16): 0 = the scoring call's last kernel stores them straight into the pinned block (round 5: 7.9 - 8.1 / 8.8), 2 = the copy
    // ENGINE, driven by a thread of its own that makes no HIP call until the launch's "scored" word has arrived (8.3 / 9.05: the
    // last scoring kernel no longer sits on its CUs while the link takes 4 MB).  (A 32-workgroup copy kernel on a stream of its
    // own lost: 6.7 / 7.5.  Without any copy-out the loop runs at 9.9 / 11.4: the link is what the loop pays for.)
    int out_mode = 2;
    std::thread out_worker;
    std::deque<int> out_queue;
    std::mutex out_mu;
    std::condition_variable cv_out;
    bool out_stop = false;
    std::mutex err_mu;            // err / err_code: written by the library thread and by the caller, read by dae_pipeline_last_error
    std::string err;
    int err_code = 0;
    uint64_t next_ticket = 1;
    int open_slot = -1;           // slot of the launch being staged
    int next_slot = 0;            // slots take the launches in ring order ...
    int poll_slot = 0;            // ... and are polled in the same order
    int next_lane = 0;            // lanes take the launches in turn
    hipEvent_t last_gate = nullptr;   // fp32, several lanes: the gate event of the launch issued before (dae_set_decode_gate)
    uint64_t guard_fallbacks = 0, launches = 0, seq_counter = 0;
    bool has_title = false;       // dae_pipeline_create_titled
    TitleW tw;
    double dbg_dev_ms = 0.0; uint64_t dbg_dev_n = 0;
    std::vector<std::pair<int, uint64_t>> dbg_log;   // experiments build: (tag * 100 + slot, ns) host timeline
    uint64_t dbg_ns[4] = {0, 0, 0, 0};            // experiments build: cumulative stage stamps of issue()
    int exact_pause = 0, overflow_streak = 0;      // titled + exact: launches left on the fp32 kernels / overflow events in a row
    uint64_t issue_ns = 0, idle_ns = 0, submit_ns = 0, wait_ns = 0;      // where the host side of the loop spends its time (dae_pipeline_times)
};

namespace {

#ifdef DAE_EXPERIMENTS
#define PLOG(p, tag, slot) do { if (dae_exp_env("DAE_DBG_PIPE") && (p)->dbg_log.size() < 100000) (p)->dbg_log.emplace_back((tag) * 100 + (slot), \
    (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count()); } while (0)
#else
#define PLOG(p, tag, slot)
#endif

thread_local std::string g_pipe_err;

// pfail: an error of THIS call (message only); pfatal: the pipeline is broken from here on (every later call returns it)
int pfail(dae_pipeline* p, int code, const char* msg)
{
    if (p) { std::lock_guard<std::mutex> g(p->err_mu); p->err = msg; } else g_pipe_err = msg;
    return code;
}
int pfatal(dae_pipeline* p, int code, const char* msg)
{
    if (p) {
        std::lock_guard<std::mutex> g(p->err_mu);
        if (!p->err_code) { p->err = msg; p->err_code = code; }
    }
    return code;
}

// the calling thread's current device, put back when the call returns: the pipeline's entry points select ITS device for
// their HIP calls, and a multi-GPU process must not find its device changed behind its back
struct DeviceGuard {
    int prev = -1;
    explicit DeviceGuard(int dev) { if (hipGetDevice(&prev) != hipSuccess) prev = -1; if (prev != dev) (void)hipSetDevice(dev); else prev = -1; }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

#define PIPE_HIP(p, expr)                                                                          \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess) {                                                                    \
            char _b[400];                                                                          \
            snprintf(_b, sizeof(_b), "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return pfatal((p), DAE_ERR_HIP, _b);                                                   \
        }                                                                                          \
    } while (0)

// The caller's wait for a launch makes NO HIP call while the launch runs.  A thread inside hipEventSynchronize -- or asking
// hipEventQuery in a loop -- slows the library thread's enqueueing down (measured on the titled loop, one lane: 0.94 ms per
// launch = staging + issue + device + fetch one after the other, against 0.58 ms of kernels; a consumer that idled 80 us per
// feed OUTSIDE HIP made the loop faster; profiles/r05_notes.md).
// So the wait watches the launch's SEQUENCE WORD -- the last word flags_to_host_kernel, the last kernel of the launch, writes
// into the slot's pinned flags -- without any HIP call, and only then takes the (completed) event for the formal ordering.
hipError_t wait_launch(const Slot& S)
{
    const volatile int32_t* seq = S.h_flags + 3;
    for (int spins = 0; *seq != S.seq; ++spins) {
        if (spins < 200) std::this_thread::yield();
        else std::this_thread::sleep_for(std::chrono::microseconds(20));
        if ((spins & 1023) == 1023) {                        // (a failed launch never writes its word: ask the runtime now and then)
            // ev_scored: recorded by issue() behind the scoring call on the lane's stream (ev_fetch is recorded later in the copy-
            // engine mode: until then it still holds its previous, completed record).  Complete but no word yet = the lists are
            // on their way out: keep waiting
            const hipError_t q = hipEventQuery(S.ev_scored);
            if (q != hipErrorNotReady && q != hipSuccess) return q;
        }
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    return S.sync_fetch ? hipEventSynchronize(S.ev_fetch) : hipSuccess;
}

// {csr status, guard violations, guard column} as the launch's own stream leaves them -> the slot's device words (the lane's
// NEXT launch moves the cumulative guard words; the out stream reads this snapshot, not the live words)
__global__ void flags_snapshot_kernel(int32_t* dst, const int32_t* status, const int32_t* guard, int32_t* scored_host, int32_t seq)
{
    if (threadIdx.x == 0) {
        dst[0] = status[0];
        dst[1] = guard ? guard[0] : 0;
        dst[2] = guard ? guard[1] : -1;
        if (scored_host) { __threadfence_system(); *scored_host = seq; }         // (copy-engine mode: the out thread watches this word)
    }
}

// {csr status, guard violations, guard column} of a launch -> its pinned flag words
__global__ void flags_to_host_kernel(int32_t* dst, const int32_t* snap, int32_t seq)
{
    if (threadIdx.x == 0) {
        dst[0] = snap[0];
        dst[1] = snap[1];
        dst[2] = snap[2];
        __threadfence_system();
        dst[3] = seq;                                            // (last: the caller's wait watches this word)
    }
}

// fp32 images on a lane (the bound guard's fallback; a paused exact mode): the DAE's and, on a titled pipeline, the title
// scorer's -- lane 0 re-tiles them, the other lanes borrow lane 0's (issue_mu held)
int ensure_f32(dae_pipeline* p, int lane)
{
    Lane& L0 = p->lanes[0];
    Lane& L = p->lanes[lane];
    int rc = DAE_OK;
    if (!L0.has_f32) {
        rc = dae_prepack_decoder(L0.ctx, p->W_dec, p->b_dec, p->V, p->H, 0, p->V, DAE_DTYPE_F32);
        if (rc) return pfatal(p, rc, dae_last_error(L0.ctx));
        PIPE_HIP(p, hipStreamSynchronize(L0.stream));
        L0.has_f32 = true;
    }
    if (p->has_title && !L0.t_has_f32) {
        rc = dae_prepack_decoder(L0.tctx, p->tw.out_WT, p->tw.out_b, p->V, p->tw.ld_feat, 0, p->V, DAE_DTYPE_F32);
        if (rc) return pfatal(p, rc, dae_last_error(L0.tctx));
        PIPE_HIP(p, hipStreamSynchronize(L0.stream));
        L0.t_has_f32 = true;
    }
    if (lane != 0 && !L.has_f32) {
        rc = dae_share_decoder(L.ctx, L0.ctx, DAE_DTYPE_F32);
        if (rc) return pfatal(p, rc, dae_last_error(L.ctx));
        L.has_f32 = true;
    }
    if (lane != 0 && p->has_title && !L.t_has_f32) {
        rc = dae_share_decoder(L.tctx, L0.tctx, DAE_DTYPE_F32);
        if (rc) return pfatal(p, rc, dae_last_error(L.tctx));
        L.t_has_f32 = true;
    }
    return DAE_OK;
}

// everything of one launch, asynchronously on its lane's stream (issue_mu held)
int issue(dae_pipeline* p, Slot& S, int dtype)
{
    Lane& L = p->lanes[S.lane];
#ifdef DAE_EXPERIMENTS         // where a launch's issue time goes (DAE_DBG_PIPE=1 prints the sums when the pipeline is destroyed)
    static const bool dbg_pipe = dae_exp_env("DAE_DBG_PIPE") != nullptr;
    const auto t_a = std::chrono::steady_clock::now();
    auto lap = [&](int i) {
        if (dbg_pipe) p->dbg_ns[i] += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t_a).count();
    };
#else
    auto lap = [&](int) {};
#endif
    // (plain launches stage 32-bit (row, col) pairs -- submit_impl narrows them while it copies -- titled ones the int64 feed)
    PIPE_HIP(p, hipMemcpyAsync(S.d_pos, S.h_pos, (size_t)S.nnz * 2 * (S.titled ? sizeof(int64_t) : sizeof(int32_t)), hipMemcpyHostToDevice,
                               p->copy_stream));
    PIPE_HIP(p, hipMemcpyAsync(S.d_val, S.h_val, (size_t)S.nnz * sizeof(float), hipMemcpyHostToDevice, p->copy_stream));
    if (S.titled) {
        PIPE_HIP(p, hipMemcpyAsync(S.d_titles, S.h_titles, (size_t)S.rows * p->tw.L * sizeof(int32_t), hipMemcpyHostToDevice, p->copy_stream));
        PIPE_HIP(p, hipMemcpyAsync(S.d_use, S.h_use, (size_t)S.rows * sizeof(float), hipMemcpyHostToDevice, p->copy_stream));
    }
    PIPE_HIP(p, hipEventRecord(S.ev_h2d, p->copy_stream));
    if (p->dtype == DAE_DTYPE_F32 && p->lanes.size() > 1) {
        // the fp32 filter launch takes every CU, two of them in flight only queue behind each other: this launch's waits for
        // the one issued before it (on another lane) and announces its own end.  One event per launch SLOT: re-recording a
        // lane's event while its previous record was still pending made hipEventRecord wait for it (0.7 ms per launch).
        (void)dae_set_decode_gate(L.ctx, p->last_gate, S.ev_gate);
        p->last_gate = S.ev_gate;
    }
    int rc;
    S.ran_dtype = dtype;
#ifdef DAE_EXPERIMENTS
    if (dbg_pipe) {                                          // device time of the launch: a timing event pair on the lane's stream
        if (!S.dbg_t0) { (void)hipEventCreate(&S.dbg_t0); (void)hipEventCreate(&S.dbg_t1); }
        else if (S.dbg_used) { float ms = 0.f; if (hipEventElapsedTime(&ms, S.dbg_t0, S.dbg_t1) == hipSuccess) { p->dbg_dev_ms += ms; ++p->dbg_dev_n; } }
        (void)hipEventRecord(S.dbg_t0, L.stream);
        S.dbg_used = true;
    }
#endif
    // NO DOWNLOADS: the last kernel of a launch writes its lists straight into the launch's pinned result block (host memory
    // the device reaches over the link: 1.5 - 4 MB of coalesced stores per launch), and a one-wave kernel leaves the launch's
    // flags next to them.  A hipMemcpyAsync device-to-host blocks its caller until everything queued before it has run --
    // whichever stream it is put on (measured: on the lane's stream and on a fetch stream behind an event alike): the
    // library thread sat in those calls for the length of every launch and a second launch was never in flight
    // (0.77 of a titled launch's 0.9 ms of issue time; profiles/r05_notes.md).
    OutBlock& ob = p->blocks[S.block];
    const bool direct = p->out_mode == 0;
    int32_t* const out_idx = direct ? ob.idx : S.d_idx;
    float* const out_score = !p->want_scores ? L.d_score : direct ? ob.score : S.d_score;     // (scores nobody fetches stay on the lane)
    lap(0);
    if (S.titled) {
        // main_challenge.py:80-90 with DAE_title: the whole titled launch in one library call (api.hip dae_title_score)
        const TitleW& t = p->tw;
        if (dtype == DAE_DTYPE_F32) { rc = ensure_f32(p, S.lane); if (rc) return rc; }
        // the half that does not depend on the lane's previous launch -- title features, CSR + seed lists, hidden rows, mixing
        // weights -- on the prep stream (contexts of its own), the ranking on the lane (api.hip dae_title_prepare / _rank)
        dae_title_bufs tb;
        tb.rp = S.d_rp; tb.col = S.d_col; tb.srp = S.d_srp; tb.sc = S.d_scol; tb.val = S.d_cval;
        tb.h = S.t_h; tb.feat = S.t_feat; tb.wt = S.t_wt; tb.wp = S.t_wp;
        PIPE_HIP(p, hipStreamWaitEvent(p->prep_stream, S.ev_h2d, 0));
        rc = dae_title_prepare(p->prep_tctx, p->prep_ctx, S.d_pos, S.d_val, 0, S.nnz, S.rows, p->V, p->W_enc, p->b_enc, p->H, S.d_titles,
                               t.L, t.emb, t.n_char, t.E, t.conv_w, t.conv_b, t.fs.data(), t.n_sizes, t.F, t.ld_feat, S.d_use,
                               p->n_tracks, tb, S.d_status);
        if (rc) return pfatal(p, rc, dae_last_error(p->prep_tctx));
        PIPE_HIP(p, hipEventRecord(S.ev_prep, p->prep_stream));
        PIPE_HIP(p, hipStreamWaitEvent(L.stream, S.ev_prep, 0));
        rc = dae_title_rank(L.tctx, L.ctx, dtype, S.rows, p->V, p->H, t.ld_feat, tb, p->n_tracks, p->k, out_score, out_idx, L.d_guard);
        if (rc) return pfatal(p, rc, dae_last_error(L.tctx));
        lap(1);
    } else {
        // the feed -> CSR and the seed lists (the playlist's own tracks) in ONE group of four launches (round 6: six + a wider feed)
        // ... on the prep stream, behind the upload; the lane's stream only waits for the finished CSR
        PIPE_HIP(p, hipStreamWaitEvent(p->prep_stream, S.ev_h2d, 0));
        rc = dae_launch_coo32_to_csr_seeds(p->prep_ctx, reinterpret_cast<const int32_t*>(S.d_pos), S.d_val, 0, S.nnz, S.rows, p->V, S.d_rp,
                                           S.d_col, S.d_cval, S.d_status, p->n_tracks, S.d_srp, S.d_scol);
        if (rc) return pfatal(p, rc, dae_last_error(p->prep_ctx));
        PIPE_HIP(p, hipEventRecord(S.ev_prep, p->prep_stream));
        PIPE_HIP(p, hipStreamWaitEvent(L.stream, S.ev_prep, 0));
        rc = dae_score_topk(L.ctx, S.d_rp, S.d_col, S.d_cval, p->W_enc, p->b_enc, p->V, p->H, S.rows, dtype, p->n_tracks,
                            S.d_srp, S.d_scol, p->k, DAE_OUT_SCORE, out_score, out_idx);
        if (rc) return pfatal(p, rc, dae_last_error(L.ctx));
    }
    // The downloads go to the lane's FETCH stream, behind an event of the launch: a hipMemcpyAsync on a stream that still has
    // kernels queued blocks its caller until they have run (as for the uploads above) -- on the lane's own stream the library
    // thread sat in these calls for the length of the launch (0.77 ms of a titled launch's 0.9 ms issue time, stage stamps of
    // issue(): profiles/r05_notes.md), and no second launch was in flight.  The small words first pass through one device
    // block (flags: status, guard words), so a launch costs two or three copies.
    const int32_t* gw = nullptr;                             // the guard words as this launch left them (titled fp32: zeros)
    if (S.titled) {
        gw = L.d_guard;
    } else if (dtype == DAE_DTYPE_BF16_EXACT) {
        rc = dae_exact_guard_words(L.ctx, &gw);
        if (rc) return pfatal(p, rc, dae_last_error(L.ctx));
    }
    // (odd, hence never 0 -- the word's idle value -- and different for every issue: a guard fallback re-issues a slot right
    // after its first issue, whose word is still in h_flags[3]; the counter wraps after 2^30 launches)
    S.seq = (int32_t)(((++p->seq_counter << 1) | 1u) & 0x7FFFFFFFu);
    S.polls = 0;
    hipLaunchKernelGGL(flags_snapshot_kernel, dim3(1), dim3(64), 0, L.stream, S.d_flags, S.d_status, gw,
                       p->out_mode == 2 ? S.h_flags + 4 : nullptr, S.seq);
    PIPE_HIP(p, hipGetLastError());
#ifdef DAE_EXPERIMENTS
    if (dbg_pipe) (void)hipEventRecord(S.dbg_t1, L.stream);
#endif
    PIPE_HIP(p, hipEventRecord(S.ev_scored, L.stream));
    if (p->out_mode == 2) {
        // the copy engine, from the out thread: nothing more to enqueue here
        S.sync_fetch = false;
        {
            std::lock_guard<std::mutex> g(p->out_mu);
            p->out_queue.push_back((int)(&S - p->slots.data()));
        }
        p->cv_out.notify_one();
        lap(2);
        return DAE_OK;
    }
    S.sync_fetch = true;
    hipStream_t fs = L.stream;
    hipLaunchKernelGGL(flags_to_host_kernel, dim3(1), dim3(64), 0, fs, S.h_flags, S.d_flags, S.seq);
    PIPE_HIP(p, hipGetLastError());
    PIPE_HIP(p, hipEventRecord(S.ev_fetch, fs));
    lap(2);
    return DAE_OK;
}

// out_mode 2: the lists of every issued launch, in issue order, through the copy engine.  The thread makes no HIP call until the
// launch's "scored" word (written by flags_snapshot_kernel, the scoring call's last kernel) has arrived -- a thread parked inside
// hipEventSynchronize slowed the issuing thread's enqueueing down (profiles/r05_notes.md) -- then three asynchronous copies on the
// out stream (nothing is queued there: they do not block), one synchronize, and the sequence word the caller's wait watches.
void out_worker_main(dae_pipeline* p)
{
    (void)hipSetDevice(p->device);
    for (;;) {
        int si;
        {
            std::unique_lock<std::mutex> lk(p->out_mu);
            p->cv_out.wait(lk, [&] { return p->out_stop || !p->out_queue.empty(); });
            if (p->out_queue.empty()) return;                // (stop, and nothing left to move)
            si = p->out_queue.front();
            p->out_queue.pop_front();
        }
        Slot& S = p->slots[si];
        const int32_t seq = S.seq;
        const volatile int32_t* scored = S.h_flags + 4;
        hipError_t e = hipSuccess;
        for (int spins = 0; *scored != seq; ++spins) {
            if (spins < 200) std::this_thread::yield();
            else std::this_thread::sleep_for(std::chrono::microseconds(20));
            if ((spins & 1023) == 1023) {                    // (a faulted launch never writes its word)
                const hipError_t q = hipEventQuery(S.ev_scored);
                if (q != hipErrorNotReady && q != hipSuccess) { e = q; break; }
            }
        }
        const OutBlock& ob = p->blocks[S.block];
        const size_t nb = (size_t)S.rows * p->k;
        if (e == hipSuccess) e = hipMemcpyAsync(ob.idx, S.d_idx, nb * sizeof(int32_t), hipMemcpyDeviceToHost, p->out_stream);
        if (e == hipSuccess && p->want_scores) e = hipMemcpyAsync(ob.score, S.d_score, nb * sizeof(float), hipMemcpyDeviceToHost, p->out_stream);
        if (e == hipSuccess) e = hipMemcpyAsync(S.h_flags, S.d_flags, 3 * sizeof(int32_t), hipMemcpyDeviceToHost, p->out_stream);
        if (e == hipSuccess) e = hipStreamSynchronize(p->out_stream);
        if (e != hipSuccess) {
            (void)pfatal(p, DAE_ERR_HIP, hipGetErrorString(e));
            // (the caller's wait ends on the word all the same: it finds err_code set)
        }
        __atomic_thread_fence(__ATOMIC_RELEASE);
        *(volatile int32_t*)(S.h_flags + 3) = seq;
    }
}

void worker_main(dae_pipeline* p)
{
    (void)hipSetDevice(p->device);
    std::unique_lock<std::mutex> lk(p->mu);
    for (;;) {
        const auto t_idle = std::chrono::steady_clock::now();
        p->cv_worker.wait(lk, [&] { return p->stop || !p->queue.empty(); });
        p->idle_ns += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t_idle).count();
        if (p->stop && p->queue.empty()) return;
        const int si = p->queue.front();
        p->queue.pop_front();
        Slot& S = p->slots[si];
        const bool broken = p->err_code != 0;
        lk.unlock();
        const auto t_iss = std::chrono::steady_clock::now();
        if (!broken) {                                       // after an error: drain without touching the device
            std::lock_guard<std::mutex> g(p->issue_mu);
            int dt = p->dtype;
            if (S.titled && dt == DAE_DTYPE_BF16_EXACT && p->exact_pause > 0) { --p->exact_pause; dt = DAE_DTYPE_F32; }
            PLOG(p, 2, si);
            (void)issue(p, S, dt);
            PLOG(p, 3, si);
        }
        lk.lock();
        p->issue_ns += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t_iss).count();
        S.state = 3;
        ++p->launches;
        p->cv_caller.notify_all();
    }
}

// close the open launch: a result block and a lane for it, then over to the worker (caller thread, lock held)
int close_open(dae_pipeline* p)
{
    if (p->open_slot < 0) return DAE_OK;
    Slot& S = p->slots[p->open_slot];
    int b = -1;
    for (size_t i = 0; i < p->blocks.size(); ++i) if (p->blocks[i].refs == 0) { b = (int)i; break; }
    if (b < 0) return pfail(p, DAE_PIPE_BUSY, "dae_pipeline: every result block is still held by the caller (dae_pipeline_release)");
    p->blocks[b].refs = 1;                                  // the launch's own reference, dropped when its last feed went out
    S.block = b;
    S.lane = p->next_lane;
    p->next_lane = (p->next_lane + 1) % (int)p->lanes.size();
    S.state = 2;
    S.next_feed = 0;
    PLOG(p, 1, p->open_slot);
    p->queue.push_back(p->open_slot);
    p->open_slot = -1;
    p->cv_worker.notify_one();
    return DAE_OK;
}

}  // namespace

extern "C" {

const char* dae_pipeline_last_error(const dae_pipeline* p)
{
    if (!p) return g_pipe_err.c_str();
    thread_local std::string copy;                           // (the library thread may replace p->err at any time)
    {
        std::lock_guard<std::mutex> g(const_cast<dae_pipeline*>(p)->err_mu);
        copy = p->err;
    }
    return copy.c_str();
}

// Lane and copy streams are taken from a process-wide pool and RETURNED to it, never destroyed: the runtime binds a stream to a
// hardware queue when it is created, and after a pipeline's streams were destroyed the next pipeline's fresh ones came out sharing
// queues -- its lanes' launches queued behind each other (bench.py's host-loop rows run fp32, then exact_bf16, on one model: the
// second pipeline measured 5.8 M playlists/s against 7.3 M as the process's first; scripts/probe/row_diag.py)
// Round 6: the pool is keyed by ROLE as well.  It was first-in first-out over all of a pipeline's streams, so every other pipeline
// of a process got its LANES on streams that had been the copy / out / prep streams before (and the other way round): such a
// pipeline ran 25 - 30 % slower -- 6.1 against 8.4 M playlists/s, instance after instance in the pattern fast, fast, slow, slow
// (scripts/gpu_r6_t14.sh; bench.py's host-loop rows and scripts/bench_loop.py disagreed by exactly that).  A stream now returns to
// the role it was created for: 0 = a lane (kernels of the scoring call), 1 = copies (upload / out), 2 = the CSR build.
static std::mutex g_stream_pool_mu;
struct PooledStream { int device, role; hipStream_t s; };
static std::vector<PooledStream> g_stream_pool;
static hipStream_t pool_take_stream(int device, int role)
{
    {
        std::lock_guard<std::mutex> lk(g_stream_pool_mu);
        for (size_t i = g_stream_pool.size(); i-- > 0;)                        // (last in, first out: a lane gets a lane's stream back)
            if (g_stream_pool[i].device == device && g_stream_pool[i].role == role) {
                hipStream_t s = g_stream_pool[i].s;
                g_stream_pool.erase(g_stream_pool.begin() + (long)i);
                return s;
            }
    }
    hipStream_t s = nullptr;
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return nullptr;
    return s;
}
static void pool_give_stream(int device, hipStream_t s, int role)
{
    if (!s) return;
    (void)hipStreamSynchronize(s);
    std::lock_guard<std::mutex> lk(g_stream_pool_mu);
    g_stream_pool.push_back(PooledStream{device, role, s});
}

int dae_pipeline_destroy(dae_pipeline* p)
{
    if (!p) return DAE_OK;
    {
        std::lock_guard<std::mutex> g(p->mu);
        p->stop = true;
    }
    p->cv_worker.notify_all();
    if (p->worker.joinable()) p->worker.join();
    {
        std::lock_guard<std::mutex> g(p->out_mu);
        p->out_stop = true;
    }
    p->cv_out.notify_all();
    if (p->out_worker.joinable()) p->out_worker.join();
    DeviceGuard dev_guard(p->device);
#ifdef DAE_EXPERIMENTS
    if (dae_exp_env("DAE_DBG_PIPE"))
        fprintf(stderr, "PIPE issue(): uploads+gate %.2f ms | + scoring calls %.2f ms | + downloads %.2f ms (cumulative, %llu launches)\n",
                p->dbg_ns[0] / 1e6, p->dbg_ns[1] / 1e6, p->dbg_ns[2] / 1e6, (unsigned long long)p->launches);
    if (dae_exp_env("DAE_DBG_PIPE") && p->dbg_log.size() > 400) {
        uint64_t prev = 0;                                   // intervals between consecutive "got" events (launch completions seen by the caller)
        fprintf(stderr, "PLOG got-intervals (us):");
        for (const auto& e : p->dbg_log)
            if (e.first / 100 == 4) { if (prev) fprintf(stderr, " %.0f", (e.second - prev) / 1e3); prev = e.second; }
        fprintf(stderr, "\n");
        {   // the events around the longest interval
            size_t worst = 0; uint64_t wl = 0, pv = 0;
            for (size_t i = 0; i < p->dbg_log.size(); ++i)
                if (p->dbg_log[i].first / 100 == 4) { if (pv && i > 25 && p->dbg_log[i].second - pv > wl) { wl = p->dbg_log[i].second - pv; worst = i; } pv = p->dbg_log[i].second; }
            const size_t a0 = worst > 14 ? worst - 14 : 0;
            const uint64_t t0 = p->dbg_log[a0].second;
            for (size_t i = a0; i < worst + 6 && i < p->dbg_log.size(); ++i)
                fprintf(stderr, "PLOG %s slot %d  t=%.0f us\n", (const char*[]){"?", "staged", "issue>", "issue<", "got", "wait>", "freed"}[p->dbg_log[i].first / 100],
                        p->dbg_log[i].first % 100, (p->dbg_log[i].second - t0) / 1e3);
        }
    }
    if (dae_exp_env("DAE_DBG_PIPE") && p->dbg_dev_n)
        fprintf(stderr, "PIPE device time per launch (event pair on the lane's stream): %.3f ms over %llu launches\n",
                p->dbg_dev_ms / (double)p->dbg_dev_n, (unsigned long long)p->dbg_dev_n);
#endif
    for (Lane& L : p->lanes) {
        if (L.stream) (void)hipStreamSynchronize(L.stream);
        if (L.tctx) (void)dae_destroy(L.tctx);                  // (borrows lane 0's images: never frees them)
        if (L.ctx) { (void)dae_set_decode_gate(L.ctx, nullptr, nullptr); (void)dae_destroy(L.ctx); }
        pool_give_stream(p->device, L.stream, 0);
        void* dev[] = {L.d_score, L.d_guard};
        for (void* q : dev) if (q) (void)hipFree(q);
    }
    pool_give_stream(p->device, p->copy_stream, 1);
    if (p->out_stream) { (void)hipStreamSynchronize(p->out_stream); pool_give_stream(p->device, p->out_stream, 1); }
    if (p->prep_stream) (void)hipStreamSynchronize(p->prep_stream);
    if (p->prep_tctx) (void)dae_destroy(p->prep_tctx);
    if (p->prep_ctx) (void)dae_destroy(p->prep_ctx);
    pool_give_stream(p->device, p->prep_stream, 2);
    for (Slot& S : p->slots) {
        if (S.ev_scored) (void)hipEventDestroy(S.ev_scored);
        if (S.ev_prep) (void)hipEventDestroy(S.ev_prep);
        void* outs[] = {S.d_idx, S.d_score, S.d_flags, S.d_rp, S.d_col, S.d_srp, S.d_scol, S.d_status, S.d_cval, S.t_h, S.t_feat, S.t_wt, S.t_wp};
        for (void* q : outs) if (q) (void)hipFree(q);
        if (S.ev_fetch) (void)hipEventDestroy(S.ev_fetch);
        if (S.ev_h2d) (void)hipEventDestroy(S.ev_h2d);
        if (S.ev_gate) (void)hipEventDestroy(S.ev_gate);
        if (S.d_pos) (void)hipFree(S.d_pos);
        if (S.d_val) (void)hipFree(S.d_val);
        if (S.d_titles) (void)hipFree(S.d_titles);
        if (S.d_use) (void)hipFree(S.d_use);
        void* host[] = {S.h_pos, S.h_val, S.h_flags, S.h_titles, S.h_use};
        for (void* q : host) if (q) (void)hipHostFree(q);
    }
    for (OutBlock& b : p->blocks) {
        if (b.idx) (void)hipHostFree(b.idx);
        if (b.score) (void)hipHostFree(b.score);
    }
    delete p;
    return DAE_OK;
}

static int pipeline_create(int device, const float* W_enc, const float* b_enc, const float* W_dec, const float* b_dec,
                           int V, int H, int n_tracks, int dtype, int k, int group_rows, int64_t max_nnz, int lanes,
                           int want_scores, int result_blocks, const TitleW* tw, dae_pipeline** out)
{
    if (!out) return pfail(nullptr, DAE_ERR_ARG, "out is null");
    if (!W_enc || !b_enc || !W_dec || !b_dec) return pfail(nullptr, DAE_ERR_ARG, "null pointer");
    if (V <= 0 || H <= 0 || n_tracks <= 0 || n_tracks > V || k < 1 || k > DAE_MAX_K || group_rows < 1 || group_rows > 16384 ||
        max_nnz < 1 || max_nnz >= ((int64_t)1 << 31) || lanes < 1 || lanes > 8)
        return pfail(nullptr, DAE_ERR_ARG, "dae_pipeline_create: bad shape / sizes");
    if (dtype != DAE_DTYPE_F32 && dtype != DAE_DTYPE_BF16 && dtype != DAE_DTYPE_BF16_EXACT)
        return pfail(nullptr, DAE_ERR_ARG, "dae_pipeline_create: unknown dtype");
    const int n_slots = 2 * lanes + 2;                       // launches staged, queued, running or waiting to be polled
    if (result_blocks < n_slots) result_blocks = n_slots;
    dae_pipeline* p = new dae_pipeline();
    p->device = device; p->V = V; p->H = H; p->n_tracks = n_tracks; p->dtype = dtype; p->k = k; p->group_rows = group_rows;
    p->max_nnz = max_nnz; p->want_scores = want_scores ? 1 : 0;
    p->W_enc = W_enc; p->b_enc = b_enc; p->W_dec = W_dec; p->b_dec = b_dec;
    if (tw) { p->has_title = true; p->tw = *tw; }
    p->lanes.resize(lanes);
    p->slots.resize(n_slots);
    p->blocks.resize(result_blocks);
    auto bail = [&](int rc, const char* msg) { const std::string m(msg); dae_pipeline_destroy(p); g_pipe_err = m; return rc; };
    DeviceGuard dev_guard(device);
    { int cur = -1; if (hipGetDevice(&cur) != hipSuccess || cur != device) return bail(DAE_ERR_HIP, "hipSetDevice failed"); }
    if (!(p->copy_stream = pool_take_stream(device, 1))) return bail(DAE_ERR_HIP, "stream creation failed");
    if (!(p->out_stream = pool_take_stream(device, 1))) return bail(DAE_ERR_HIP, "stream creation failed");
    if (!(p->prep_stream = pool_take_stream(device, 2))) return bail(DAE_ERR_HIP, "stream creation failed");
    {
        int rc_p = dae_create(device, &p->prep_ctx);
        if (!rc_p) rc_p = dae_set_stream(p->prep_ctx, p->prep_stream);
        if (rc_p) return bail(rc_p, dae_last_error(p->prep_ctx));
    }
    const size_t rows = (size_t)group_rows, nz = (size_t)max_nnz, kk = (size_t)k;
    for (int i = 0; i < lanes; ++i) {
        Lane& L = p->lanes[i];
        int rc = dae_create(device, &L.ctx);
        if (rc) return bail(rc, dae_last_error(nullptr));
        L.stream = pool_take_stream(device, 0);
        bool ok = L.stream != nullptr &&
                  hipMalloc(reinterpret_cast<void**>(&L.d_score), (rows * kk * sizeof(float) + 15) / 16 * 16) == hipSuccess;
        if (!ok) return bail(DAE_ERR_NOMEM, "dae_pipeline_create: allocation failed");
        rc = dae_set_stream(L.ctx, L.stream);
        if (!rc) rc = i == 0 ? dae_prepack_decoder(L.ctx, W_dec, b_dec, V, H, 0, V, dtype) : DAE_OK;
        if (rc) return bail(rc, dae_last_error(L.ctx));
        if (i == 0 && hipStreamSynchronize(L.stream) != hipSuccess) return bail(DAE_ERR_HIP, "prepack failed");
        if (i > 0) {
            rc = dae_share_decoder(L.ctx, p->lanes[0].ctx, dtype);
            if (rc) return bail(rc, dae_last_error(L.ctx));
        }
        (void)dae_set_overlap_hint(L.ctx, lanes);
        if (tw) {
            // the title scorer next to the DAE: its own context on the lane's stream, its "decoder" = Output_W^T / Output_b
            // (lane 0 re-tiles it, the others borrow the image)
            rc = dae_create(device, &L.tctx);
            if (rc) return bail(rc, dae_last_error(nullptr));
            if (hipMalloc(reinterpret_cast<void**>(&L.d_guard), DAE_GUARD_BYTES) != hipSuccess ||
                hipMemsetAsync(L.d_guard, 0, DAE_GUARD_BYTES, L.stream) != hipSuccess)
                return bail(DAE_ERR_NOMEM, "dae_pipeline_create: allocation failed");
            rc = dae_set_stream(L.tctx, L.stream);
            if (!rc) rc = i == 0 ? dae_prepack_decoder(L.tctx, tw->out_WT, tw->out_b, V, tw->ld_feat, 0, V, dtype)
                                 : dae_share_decoder(L.tctx, p->lanes[0].tctx, dtype);
            // (the scorer is frozen for the life of the pipeline: its convolutions as a table, dae_title_prepack_features)
            if (!rc) rc = dae_title_prepack_features(L.tctx, tw->emb, tw->n_char, tw->E, tw->conv_w, tw->fs.data(), tw->n_sizes, tw->F);
            if (rc) return bail(rc, dae_last_error(L.tctx));
            if (i == 0 && hipStreamSynchronize(L.stream) != hipSuccess) return bail(DAE_ERR_HIP, "prepack failed");
            (void)dae_set_overlap_hint(L.tctx, lanes);
            if (dtype == DAE_DTYPE_F32) { L.has_f32 = true; L.t_has_f32 = true; }
        }
    }
    if (tw) {
        int rc_t = dae_create(device, &p->prep_tctx);
        if (!rc_t) rc_t = dae_set_stream(p->prep_tctx, p->prep_stream);
        if (!rc_t) rc_t = dae_title_prepack_features(p->prep_tctx, tw->emb, tw->n_char, tw->E, tw->conv_w, tw->fs.data(), tw->n_sizes, tw->F);
        if (rc_t) return bail(rc_t, dae_last_error(p->prep_tctx));
    }
    for (Slot& S : p->slots) {
        bool ok = hipEventCreateWithFlags(&S.ev_fetch, hipEventDisableTiming) == hipSuccess &&
                  hipEventCreateWithFlags(&S.ev_scored, hipEventDisableTiming) == hipSuccess &&
                  hipMalloc(reinterpret_cast<void**>(&S.d_idx), (rows * kk * sizeof(int32_t) + 15) / 16 * 16) == hipSuccess &&
                  (!want_scores || hipMalloc(reinterpret_cast<void**>(&S.d_score), (rows * kk * sizeof(float) + 15) / 16 * 16) == hipSuccess) &&
                  hipMalloc(reinterpret_cast<void**>(&S.d_flags), 4 * sizeof(int32_t)) == hipSuccess &&
                  hipEventCreateWithFlags(&S.ev_prep, hipEventDisableTiming) == hipSuccess &&
                  hipMalloc(reinterpret_cast<void**>(&S.d_rp), (rows + 1) * sizeof(int32_t)) == hipSuccess &&
                  hipMalloc(reinterpret_cast<void**>(&S.d_col), nz * sizeof(int32_t)) == hipSuccess &&
                  hipMalloc(reinterpret_cast<void**>(&S.d_cval), nz * sizeof(float)) == hipSuccess &&
                  hipMalloc(reinterpret_cast<void**>(&S.d_status), sizeof(int32_t)) == hipSuccess &&
                  hipMalloc(reinterpret_cast<void**>(&S.d_srp), (rows + 1) * sizeof(int32_t)) == hipSuccess &&
                  hipMalloc(reinterpret_cast<void**>(&S.d_scol), nz * sizeof(int32_t)) == hipSuccess &&
                  hipEventCreateWithFlags(&S.ev_h2d, hipEventDisableTiming) == hipSuccess &&
                  hipEventCreateWithFlags(&S.ev_gate, hipEventDisableTiming) == hipSuccess &&
                  hipMalloc(reinterpret_cast<void**>(&S.d_pos), nz * 2 * sizeof(int64_t)) == hipSuccess &&
                  hipMalloc(reinterpret_cast<void**>(&S.d_val), nz * sizeof(float)) == hipSuccess &&
                  hipHostMalloc(reinterpret_cast<void**>(&S.h_pos), nz * 2 * sizeof(int64_t)) == hipSuccess &&
                  hipHostMalloc(reinterpret_cast<void**>(&S.h_val), nz * sizeof(float)) == hipSuccess &&
                  hipHostMalloc(reinterpret_cast<void**>(&S.h_flags), 8 * sizeof(int32_t)) == hipSuccess;
        if (ok && tw)
            ok = hipMalloc(reinterpret_cast<void**>(&S.t_h), rows * (size_t)H * sizeof(float)) == hipSuccess &&
                 hipMalloc(reinterpret_cast<void**>(&S.t_feat), rows * (size_t)tw->ld_feat * sizeof(float)) == hipSuccess &&
                 hipMalloc(reinterpret_cast<void**>(&S.t_wt), rows * sizeof(float)) == hipSuccess &&
                 hipMalloc(reinterpret_cast<void**>(&S.t_wp), rows * sizeof(float)) == hipSuccess &&
                 hipMalloc(reinterpret_cast<void**>(&S.d_titles), rows * (size_t)tw->L * sizeof(int32_t)) == hipSuccess &&
                 hipMalloc(reinterpret_cast<void**>(&S.d_use), rows * sizeof(float)) == hipSuccess &&
                 hipHostMalloc(reinterpret_cast<void**>(&S.h_titles), rows * (size_t)tw->L * sizeof(int32_t)) == hipSuccess &&
                 hipHostMalloc(reinterpret_cast<void**>(&S.h_use), rows * sizeof(float)) == hipSuccess;
        if (!ok) return bail(DAE_ERR_NOMEM, "dae_pipeline_create: pinned allocation failed");
        S.h_flags[0] = S.h_flags[1] = 0; S.h_flags[2] = -1; S.h_flags[3] = 0; S.h_flags[4] = 0;
    }
    for (OutBlock& b : p->blocks) {
        bool ok = hipHostMalloc(reinterpret_cast<void**>(&b.idx), rows * kk * sizeof(int32_t) + 16) == hipSuccess &&
                  (!want_scores || hipHostMalloc(reinterpret_cast<void**>(&b.score), rows * kk * sizeof(float) + 16) == hipSuccess);
        if (!ok) return bail(DAE_ERR_NOMEM, "dae_pipeline_create: pinned allocation failed");
    }
    for (Lane& L : p->lanes) if (hipStreamSynchronize(L.stream) != hipSuccess) return bail(DAE_ERR_HIP, "setup failed");
    if (const char* om = dae_exp_env("DAE_PIPE_OUT")) p->out_mode = atoi(om) == 0 ? 0 : 2;       // A/B (experiments build)
    if (p->out_mode == 2) p->out_worker = std::thread(out_worker_main, p);
    p->worker = std::thread(worker_main, p);
    *out = p;
    return DAE_OK;
}

int dae_pipeline_create(int device, const float* W_enc, const float* b_enc, const float* W_dec, const float* b_dec,
                        int V, int H, int n_tracks, int dtype, int k, int group_rows, int64_t max_nnz, int lanes,
                        int want_scores, int result_blocks, dae_pipeline** out)
{
    return pipeline_create(device, W_enc, b_enc, W_dec, b_dec, V, H, n_tracks, dtype, k, group_rows, max_nnz, lanes, want_scores,
                           result_blocks, nullptr, out);
}

int dae_pipeline_create_titled(int device, const float* W_enc, const float* b_enc, const float* W_dec, const float* b_dec,
                               int V, int H, int n_tracks, const float* emb, int n_char, int E, const float* conv_w,
                               const float* conv_b, const int32_t* filter_sizes, int n_sizes, int F, const float* Output_WT,
                               const float* Output_b, int ld_feat, int L, int dtype, int k, int group_rows, int64_t max_nnz,
                               int lanes, int want_scores, int result_blocks, dae_pipeline** out)
{
    if (!emb || !conv_w || !conv_b || !filter_sizes || !Output_WT || !Output_b) return pfail(nullptr, DAE_ERR_ARG, "null pointer");
    if (n_char < 1 || E < 1 || n_sizes < 1 || n_sizes > DAE_TITLE_MAX_SIZES || F < 1 || L < 1 || ld_feat < n_sizes * F || group_rows > 4096)
        return pfail(nullptr, DAE_ERR_ARG, "dae_pipeline_create_titled: bad title shapes (a titled launch holds at most 4096 rows)");
    TitleW tw;
    tw.emb = emb; tw.conv_w = conv_w; tw.conv_b = conv_b; tw.out_WT = Output_WT; tw.out_b = Output_b;
    tw.n_char = n_char; tw.E = E; tw.n_sizes = n_sizes; tw.F = F; tw.ld_feat = ld_feat; tw.L = L;
    tw.fs.assign(filter_sizes, filter_sizes + n_sizes);
    return pipeline_create(device, W_enc, b_enc, W_dec, b_dec, V, H, n_tracks, dtype, k, group_rows, max_nnz, lanes, want_scores,
                           result_blocks, &tw, out);
}

int dae_pipeline_flush(dae_pipeline* p)
{
    if (!p) return DAE_ERR_ARG;
    std::unique_lock<std::mutex> lk(p->mu);
    if (p->err_code) return p->err_code;
    return close_open(p);
}

static int submit_impl(dae_pipeline* p, const int64_t* positions, const float* values, int values_broadcast, int64_t nnz,
                       int n_rows, const int32_t* titles, const float* titles_use, uint64_t* ticket_out)
{
    if (!p) return DAE_ERR_ARG;
    const int titled = titles != nullptr ? 1 : 0;
    if (n_rows < 1 || n_rows > p->group_rows || nnz < 0 || nnz > p->max_nnz || (nnz > 0 && (!positions || !values)))
        return pfail(p, DAE_ERR_ARG, "dae_pipeline_submit: a feed must fit one launch (rows <= group_rows, nnz <= max_nnz)");
    const auto t_sub = std::chrono::steady_clock::now();
    // a row index outside the feed would land in ANOTHER feed's rows of the launch: checked here (the device flags only
    // what leaves the launch).  Titled feeds: a pass of its own; plain feeds: inside the narrowing copy below (ONE pass over the
    // feed -- the caller's thread is what bounds the loop: 17 us of its 33 us per 256-row feed were these two passes, round 6)
    if (titled)
        for (int64_t i = 0; i < nnz; ++i)
            if (positions[2 * i] < 0 || positions[2 * i] >= n_rows)
                return pfail(p, DAE_ERR_ARG, "dae_pipeline_submit: a row index of the feed is outside [0, n_rows)");
    std::unique_lock<std::mutex> lk(p->mu);
    if (p->err_code) return p->err_code;
    if (p->open_slot >= 0) {
        Slot& O = p->slots[p->open_slot];
        // (a launch ranks EITHER the plain logits or the title-mixed score: feeds of the other kind start a new one)
        if (O.rows + n_rows > p->group_rows || O.nnz + nnz > p->max_nnz || O.titled != titled) {
            const int rc = close_open(p);
            if (rc) return rc;
        }
    }
    if (p->open_slot < 0) {
        // the next slot of the ring; it is free once every feed of its previous launch has been handed out
        Slot& N = p->slots[p->next_slot];
        if (N.state != 0) return pfail(p, DAE_PIPE_BUSY, "dae_pipeline_submit: every launch slot holds results that were not polled "
                                                           "yet (poll before submitting more)");
        p->open_slot = p->next_slot;
        p->next_slot = (p->next_slot + 1) % (int)p->slots.size();
        N.state = 1; N.rows = 0; N.nnz = 0; N.feeds.clear(); N.next_feed = 0; N.titled = titled;
    }
    Slot& S = p->slots[p->open_slot];
    const int row0 = S.rows;
    const int64_t off = S.nnz;
    const uint64_t ticket = p->next_ticket++;
    S.feeds.push_back(Feed{ticket, row0, n_rows});
    S.rows += n_rows; S.nnz += nnz;
    lk.unlock();                                            // the copy runs outside the lock (the worker never touches an open slot)
    if (titled) {
        int64_t* dp = S.h_pos + 2 * off;
        if (row0 == 0) {
            memcpy(dp, positions, (size_t)nnz * 2 * sizeof(int64_t));
        } else {
            for (int64_t i = 0; i < nnz; ++i) { dp[2 * i] = positions[2 * i] + row0; dp[2 * i + 1] = positions[2 * i + 1]; }
        }
    } else {
        // 32-bit pairs: the row shifted to its place in the launch, a column that does not fit 32 bits becomes -1 (out of range for
        // the device's own check, like any column >= V); the row check of the feed rides along
        int32_t* dp = reinterpret_cast<int32_t*>(S.h_pos) + 2 * off;
        uint64_t bad = 0;
        const uint64_t nr = (uint64_t)n_rows;
        for (int64_t i = 0; i < nnz; ++i) {
            const int64_t r = positions[2 * i], c = positions[2 * i + 1];
            bad |= (uint64_t)((uint64_t)r >= nr);
            dp[2 * i] = (int32_t)r + row0;
            dp[2 * i + 1] = ((uint64_t)c > (uint64_t)INT32_MAX) ? -1 : (int32_t)c;
        }
        if (bad) {                                           // nothing of this feed stays: the slot is as it was before the call
            lk.lock();
            S.feeds.pop_back(); S.rows -= n_rows; S.nnz -= nnz; --p->next_ticket;
            if (S.feeds.empty()) { S.state = 0; p->next_slot = p->open_slot; p->open_slot = -1; }      // (it had opened the slot)
            return pfail(p, DAE_ERR_ARG, "dae_pipeline_submit: a row index of the feed is outside [0, n_rows)");
        }
    }
    float* dv = S.h_val + off;
    if (values_broadcast) { const float v = values[0]; for (int64_t i = 0; i < nnz; ++i) dv[i] = v; }
    else memcpy(dv, values, (size_t)nnz * sizeof(float));
    if (titled) {
        memcpy(S.h_titles + (size_t)row0 * p->tw.L, titles, (size_t)n_rows * p->tw.L * sizeof(int32_t));
        memcpy(S.h_use + row0, titles_use, (size_t)n_rows * sizeof(float));
    }
    if (ticket_out) *ticket_out = ticket;
    lk.lock();
    if (S.rows + n_rows > p->group_rows) (void)close_open(p);   // the next feed of this size would not fit: off it goes
    p->submit_ns += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t_sub).count();
    return DAE_OK;
}

int dae_pipeline_submit(dae_pipeline* p, const int64_t* positions, const float* values, int values_broadcast, int64_t nnz,
                        int n_rows, uint64_t* ticket_out)
{
    return submit_impl(p, positions, values, values_broadcast, nnz, n_rows, nullptr, nullptr, ticket_out);
}

int dae_pipeline_submit_titled(dae_pipeline* p, const int64_t* positions, const float* values, int values_broadcast, int64_t nnz,
                               int n_rows, const int32_t* titles, const float* titles_use, uint64_t* ticket_out)
{
    if (!p) return DAE_ERR_ARG;
    if (!p->has_title) return pfail(p, DAE_ERR_STATE, "dae_pipeline_submit_titled: not a titled pipeline (dae_pipeline_create_titled)");
    if (!titles || !titles_use) return pfail(p, DAE_ERR_ARG, "null pointer");
    return submit_impl(p, positions, values, values_broadcast, nnz, n_rows, titles, titles_use, ticket_out);
}

int dae_pipeline_poll(dae_pipeline* p, int wait, uint64_t* ticket, const int32_t** idx, const float** score, int* n_rows,
                      int* block)
{
    if (!p) return DAE_ERR_ARG;
    if (!idx || !n_rows || !block) return pfail(p, DAE_ERR_ARG, "null pointer");
    std::unique_lock<std::mutex> lk(p->mu);
    *n_rows = 0; *idx = nullptr; *block = -1;
    if (score) *score = nullptr;
    Slot& S = p->slots[p->poll_slot];
    if (S.state == 0) return p->err_code;                    // nothing pending (0 rows), or the worker's error
    if (S.state == 1) {
        if (!wait) return DAE_OK;
        const int rc = close_open(p);                        // the caller waits for a launch that is still open: close it
        if (rc) return rc;
    }
    if (S.state == 2) {
        if (!wait) return DAE_OK;
        p->cv_caller.wait(lk, [&] { return S.state == 3; });
    }
    if (p->err_code) return p->err_code;
    if (S.next_feed == 0) {                                  // first feed of the launch: its results have to be here
        lk.unlock();
        if (!wait) {
            if (*(const volatile int32_t*)(S.h_flags + 3) != S.seq) {                   // (no HIP call while the launch runs ...
                if ((++S.polls & 255) == 0) {                    // ... but a launch that faulted never writes its word: ask now and then)
                    const hipError_t q = hipEventQuery(S.ev_scored);
                    if (q != hipErrorNotReady && q != hipSuccess) { lk.lock(); return pfatal(p, DAE_ERR_HIP, hipGetErrorString(q)); }
                }
                return DAE_OK;
            }
            __atomic_thread_fence(__ATOMIC_ACQUIRE);
            const hipError_t q = S.sync_fetch ? hipEventSynchronize(S.ev_fetch) : hipSuccess;
            if (q != hipSuccess) { lk.lock(); return pfatal(p, DAE_ERR_HIP, hipGetErrorString(q)); }
        } else {
            const auto t_w = std::chrono::steady_clock::now();
            PLOG(p, 5, p->poll_slot);
            const hipError_t e = wait_launch(S);
            PLOG(p, 4, p->poll_slot);
            p->wait_ns += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t_w).count();
            if (e != hipSuccess) { lk.lock(); return pfatal(p, DAE_ERR_HIP, hipGetErrorString(e)); }
        }
        if (S.h_flags[0] != 0) {
            lk.lock();
            return pfatal(p, DAE_ERR_ARG, "dae_pipeline: a feed of the launch holds a column index out of range");
        }
        if (S.titled && S.ran_dtype == DAE_DTYPE_BF16_EXACT && S.h_flags[1] != p->lanes[S.lane].guard_seen) {
            // the exact title mix (dae_mix_topk_exact): the title context's guard words moved under this launch -- a recomputed
            // logit left its promised interval (column >= 0), or rows hold more survivors than the refine launch lists (column
            // -2 / -3: scores no bound can tell apart; such rows came back without recommendations) -> the launch again on the
            // fp32 kernels.  Two overflow launches in a row pause the mode for 64 launches (a model whose rows keep
            // overflowing would otherwise pay both passes every time).
            int rc;
            bool ok;
            {
                std::lock_guard<std::mutex> g(p->issue_mu);
                DeviceGuard dev_guard(p->device);
                Lane& L = p->lanes[S.lane];
                L.guard_seen = S.h_flags[1];
                if (S.h_flags[2] == -2 || S.h_flags[2] == -3) {
                    if (++p->overflow_streak >= 2) { p->exact_pause = 64; p->overflow_streak = 0; }
                } else {
                    p->overflow_streak = 0;
                }
                rc = issue(p, S, DAE_DTYPE_F32);
                ok = !rc && wait_launch(S) == hipSuccess;
            }
            lk.lock();
            if (!ok) return pfatal(p, rc ? rc : DAE_ERR_HIP, "dae_pipeline: the fp32 re-run after a bound-guard hit failed");
            ++p->guard_fallbacks;
        } else if (!S.titled && p->dtype == DAE_DTYPE_BF16_EXACT && S.h_flags[1] != 0) {
            // BOUND GUARD (include/dae_hip.h dae_exact_guard_read): a recomputed survivor left its promised interval, the
            // lists of this launch are unproven -> the same launch again on the fp32 kernels, here and now (its feed is
            // still in the slot's staging buffers; the lane's context is shared with the worker: issue_mu)
            int rc;
            bool ok;
            {
                std::lock_guard<std::mutex> g(p->issue_mu);
                DeviceGuard dev_guard(p->device);
                Lane& L = p->lanes[S.lane];
                int32_t nv = 0, col = -1;
                rc = dae_exact_guard_read(L.ctx, &nv, &col);       // (synchronises the lane, resets the words)
                if (!rc) rc = ensure_f32(p, S.lane);
                if (!rc) rc = issue(p, S, DAE_DTYPE_F32);
                ok = !rc && wait_launch(S) == hipSuccess;
            }
            lk.lock();
            if (!ok) return pfatal(p, rc ? rc : DAE_ERR_HIP, "dae_pipeline: the fp32 re-run after a bound-guard hit failed");
            S.h_flags[1] = 0;
            ++p->guard_fallbacks;
        } else {
            if (S.titled && S.ran_dtype == DAE_DTYPE_BF16_EXACT) p->overflow_streak = 0;       // (only the caller's thread touches it here)
            lk.lock();
        }
    }
    const Feed& f = S.feeds[S.next_feed];
    OutBlock& ob = p->blocks[S.block];
    if (ticket) *ticket = f.ticket;
    *idx = ob.idx + (size_t)f.row0 * p->k;
    if (score) *score = p->want_scores ? ob.score + (size_t)f.row0 * p->k : nullptr;
    *n_rows = f.n_rows;
    *block = S.block;
    ++ob.refs;                                               // the caller's reference to the block (dae_pipeline_release)
    if (++S.next_feed == S.feeds.size()) {                   // last feed of the launch: the slot is free again
        --ob.refs;                                           // (the launch's own reference)
        PLOG(p, 6, p->poll_slot);
        S.state = 0; S.block = -1;
        p->poll_slot = (p->poll_slot + 1) % (int)p->slots.size();
    }
    return DAE_OK;
}

int dae_pipeline_release(dae_pipeline* p, int block)
{
    if (!p) return DAE_ERR_ARG;
    std::lock_guard<std::mutex> g(p->mu);
    if (block < 0 || block >= (int)p->blocks.size() || p->blocks[block].refs <= 0)
        return pfail(p, DAE_ERR_ARG, "dae_pipeline_release: not a block that is out");
    --p->blocks[block].refs;
    return DAE_OK;
}

int dae_pipeline_exact_margin(dae_pipeline* p, float scale)
{
    if (!p) return DAE_ERR_ARG;
    std::unique_lock<std::mutex> lk(p->mu);
    if (p->err_code) return p->err_code;
    if (p->dtype != DAE_DTYPE_BF16_EXACT) return pfail(p, DAE_ERR_ARG, "dae_pipeline_exact_margin: not an exact_bf16 pipeline");
    for (Slot& S : p->slots)
        if (S.state != 0) return pfail(p, DAE_ERR_STATE, "dae_pipeline_exact_margin: feeds are in flight");
    lk.unlock();
    std::lock_guard<std::mutex> g(p->issue_mu);
    DeviceGuard dev_guard(p->device);
    Lane& L0 = p->lanes[0];
    for (Lane& L : p->lanes) (void)hipStreamSynchronize(L.stream);
    int rc = dae_set_exact_margin(L0.ctx, scale);
    if (!rc) rc = dae_prepack_decoder(L0.ctx, p->W_dec, p->b_dec, p->V, p->H, 0, p->V, p->dtype);
    if (rc) return pfail(p, rc, dae_last_error(L0.ctx));
    if (hipStreamSynchronize(L0.stream) != hipSuccess) return pfatal(p, DAE_ERR_HIP, "prepack failed");
    for (size_t i = 1; i < p->lanes.size(); ++i) {
        rc = dae_share_decoder(p->lanes[i].ctx, L0.ctx, p->dtype);
        if (rc) return pfail(p, rc, dae_last_error(p->lanes[i].ctx));
    }
    if (p->has_title) {                                      // the title scorer's bounds as well (alpha_c, beta_c scale with the margin)
        rc = dae_set_exact_margin(L0.tctx, scale);
        if (!rc) rc = dae_prepack_decoder(L0.tctx, p->tw.out_WT, p->tw.out_b, p->V, p->tw.ld_feat, 0, p->V, p->dtype);
        if (rc) return pfail(p, rc, dae_last_error(L0.tctx));
        if (hipStreamSynchronize(L0.stream) != hipSuccess) return pfatal(p, DAE_ERR_HIP, "prepack failed");
        for (size_t i = 1; i < p->lanes.size(); ++i) {
            rc = dae_share_decoder(p->lanes[i].tctx, L0.tctx, p->dtype);
            if (rc) return pfail(p, rc, dae_last_error(p->lanes[i].tctx));
        }
        p->exact_pause = 0; p->overflow_streak = 0;
    }
    return DAE_OK;
}

int dae_pipeline_times(dae_pipeline* p, uint64_t out4[4])
{
    if (!p || !out4) return DAE_ERR_ARG;
    std::lock_guard<std::mutex> g(p->mu);
    out4[0] = p->issue_ns; out4[1] = p->idle_ns; out4[2] = p->submit_ns; out4[3] = p->wait_ns;
    return DAE_OK;
}

int dae_pipeline_stats(dae_pipeline* p, uint64_t out3[3])
{
    if (!p || !out3) return DAE_ERR_ARG;
    std::lock_guard<std::mutex> g(p->mu);
    out3[0] = p->launches; out3[1] = p->next_ticket - 1; out3[2] = p->guard_fallbacks;
    return DAE_OK;
}

}  // extern "C"
