// csr.hip -- the feed of the reference graph on the device: COO (row, col) -> value entries in FEED
// ORDER, duplicates allowed, scattered by ASSIGNMENT into a dense [n_batch, n_input] matrix
// (models/DAEs.py:33-35, tf.sparse_tensor_to_dense(validate_indices=False): the LAST occurrence of a
// (row, col) wins) -> the CSR the kernels consume: columns ascending per
// row, one entry per (row, col), explicit zeros dropped.  Same result as the host restatement
// models/DAEs.py:coo_to_csr of this repo, entry for entry (tests/test_gpu_csr.py).
//
// Integer / byte work, HBM- and latency-bound, a few tens of KB per batch: six small launches
//   count rows -> scan -> scatter into row buckets -> per-row order + last-wins dedup -> scan -> compact
// The per-row step ranks by counting inside LDS (rows of a playlist batch hold <= a few hundred
// entries; the quadratic count is cheaper than a sorting network at that size and needs no padding);
// rows longer than the LDS buffer take the same code over global memory.
#include "dae_internal.h"

namespace {

constexpr int CSR_ROW_CAP = 4096;      // entries of one row held in LDS (16 B each)

// PT: the feed's index type -- int64_t (the reference's feed: models/DAEs.py:23 x_positions is tf.int64) or int32_t (what
// dae_pipeline stages: its copy of a feed narrows the pairs, half the pinned bytes and half the upload)
template <typename PT>
__global__ __launch_bounds__(256) void csr_count_kernel(const PT* __restrict__ pos, int64_t nnz,
                                                        int n_rows, int n_cols, int* __restrict__ cnt,
                                                        int* __restrict__ status)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nnz; i += (int64_t)gridDim.x * 256) {
        const int64_t r = pos[2 * i], c = pos[2 * i + 1];
        if (r < 0 || r >= n_rows || c < 0 || c >= n_cols) { atomicOr(status, 1); continue; }
        atomicAdd(&cnt[r], 1);
    }
}

// out[0..n] = exclusive prefix sums of in[0..n) (out[n] = total); one workgroup
__global__ __launch_bounds__(1024) void csr_scan_kernel(const int* __restrict__ in, int n,
                                                        int* __restrict__ out)
{
    __shared__ int wsum[16];
    __shared__ int carry;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        const int i = base + tid;
        const int v = i < n ? in[i] : 0;
        int x = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int y = __shfl_up(x, d); if (lane >= d) x += y; }
        if (lane == 63) wsum[wave] = x;
        __syncthreads();
        int woff = 0;
        for (int w = 0; w < wave; ++w) woff += wsum[w];
        const int c0 = carry;
        if (i < n) out[i] = c0 + woff + x - v;
        __syncthreads();
        if (tid == 1023) carry = c0 + woff + x;
        __syncthreads();
    }
    if (tid == 0) out[n] = carry;
}

template <typename PT>
__global__ __launch_bounds__(256) void csr_scatter_kernel(const PT* __restrict__ pos,
                                                          const float* __restrict__ val, int val_bcast,
                                                          int64_t nnz, int n_rows, int n_cols,
                                                          const int* __restrict__ bptr, int* __restrict__ cursor,
                                                          int* __restrict__ t_col, int* __restrict__ t_feed,
                                                          float* __restrict__ t_val)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nnz; i += (int64_t)gridDim.x * 256) {
        const int64_t r = pos[2 * i], c = pos[2 * i + 1];
        if (r < 0 || r >= n_rows || c < 0 || c >= n_cols) continue;
        const int slot = bptr[r] + atomicAdd(&cursor[r], 1);
        t_col[slot] = (int)c;
        t_feed[slot] = (int)i;                       // feed order (nnz < 2^31 checked by the caller)
        t_val[slot] = val[val_bcast ? 0 : i];
    }
}

// One workgroup per row.  Entry i is KEPT iff its value is non-zero and no entry of the row with the
// same column comes later in the feed; a kept entry's output slot is the number of kept entries with
// a smaller column.  Kept entries go to k_col / k_val at the row's bucket offset, their number to kcnt.
// The two cases are separate kernels: picking `s_col` or `t_col + b` through one pointer makes every access a FLAT
// load (25 us per launch for rows of ~100 entries; 256 workgroups), while the LDS-only body is plain ds_reads.
// <CAP, NT>: entries of a row held in LDS / threads per row.  <4096, 256> takes any batch; <512, 64> -- one wave and 4.6 KB per
// row -- is the drivers' loop's: a playlist holds at most 250 tracks + 250 artists (utils/spotify_reader.py drops longer ones),
// and with 36 KB of LDS per row the launch got ONE row per CU at a time next to the other lanes' decode workgroups: 70 us for
// 2 048 rows in the loop's timeline against 9 alone (profiles/r06_notes.md).  Longer rows take the global-memory path below.
template <int CAP, int NT>
__global__ __launch_bounds__(NT) void csr_row_kernel(const int* __restrict__ bptr,
                                                      const int* __restrict__ t_col,
                                                      const int* __restrict__ t_feed,
                                                      const float* __restrict__ t_val,
                                                      int* __restrict__ k_col, float* __restrict__ k_val,
                                                      int* __restrict__ kcnt, int n_tracks, int* __restrict__ scnt)
{
    // scnt (nullable): kept entries of the row with column < n_tracks -- the row's seed list when the seeds are the playlist's own
    // tracks (seeds_from_csr_kernel below); columns ascend, so they are the row's FIRST scnt[row] kept entries
    __shared__ int s_col[CAP];
    __shared__ int s_feed[CAP];
    __shared__ unsigned char s_keep[CAP];
    __shared__ int s_n, s_ns;
    const int row = blockIdx.x, tid = threadIdx.x;
    const int b = bptr[row], n = bptr[row + 1] - b;
    if (tid == 0) { s_n = 0; s_ns = 0; }
    if (n <= CAP) {
        // ---- the row fits the LDS buffers (always, for playlist batches) ----
        float my_val[CAP / NT];
#pragma unroll
        for (int u = 0; u < CAP / NT; ++u) {
            const int i = tid + u * NT;
            if (i < n) { s_col[i] = t_col[b + i]; s_feed[i] = t_feed[b + i]; my_val[u] = t_val[b + i]; }
        }
        __syncthreads();
        // pass 1: keep flags
        for (int i = tid; i < n; i += NT) {
            const int c = s_col[i], f = s_feed[i];
            int later = 0;
            for (int j = 0; j < n; ++j) later |= (s_col[j] == c) & (s_feed[j] > f);
            s_keep[i] = 0;                         // (written below once the value is known)
            if (!later) s_keep[i] = 2;             // provisional: survives the duplicate rule
        }
#pragma unroll
        for (int u = 0; u < CAP / NT; ++u) {
            const int i = tid + u * NT;
            if (i < n) s_keep[i] = (s_keep[i] == 2 && my_val[u] != 0.0f) ? 1 : 0;
        }
        __syncthreads();
        // pass 2: slot among the kept entries, output written directly
#pragma unroll
        for (int u = 0; u < CAP / NT; ++u) {
            const int i = tid + u * NT;
            if (i >= n || !s_keep[i]) continue;
            const int c = s_col[i];
            int slot = 0;
            for (int j = 0; j < n; ++j) slot += (s_keep[j] != 0) & (s_col[j] < c);
            atomicAdd(&s_n, 1);
            if (c < n_tracks) atomicAdd(&s_ns, 1);
            k_col[b + slot] = c; k_val[b + slot] = my_val[u];
        }
        __syncthreads();
        if (tid == 0) { kcnt[row] = s_n; if (scnt) scnt[row] = s_ns; }
        return;
    }
    // ---- a row longer than the LDS buffers: the same steps over global memory ----
    __syncthreads();
    const int* colp = t_col + b;
    const int* feedp = t_feed + b;
    // pass 1: keep flags, kept in the low bit of k_col's slot until every reader is done
    for (int i = tid; i < n; i += NT) {
        const int c = colp[i], f = feedp[i];
        bool later = false;
        for (int j = 0; j < n; ++j) later = later || (colp[j] == c && feedp[j] > f);
        k_col[b + i] = (!later && t_val[b + i] != 0.0f) ? 1 : 0;      // scratch use; rewritten below after a barrier
    }
    __syncthreads();
    // pass 2: slot among the kept entries; the SOURCE INDEX of slot s is parked in k_val[s]
    for (int i = tid; i < n; i += NT) {
        if (k_col[b + i] == 0) continue;
        const int c = colp[i];
        int slot = 0;
        for (int j = 0; j < n; ++j) slot += (k_col[b + j] != 0 && colp[j] < c) ? 1 : 0;
        atomicAdd(&s_n, 1);
        if (c < n_tracks) atomicAdd(&s_ns, 1);
        k_val[b + slot] = __int_as_float(i);
    }
    __syncthreads();
    const int m = s_n;
    for (int s = tid; s < m; s += NT) {
        const int i = __float_as_int(k_val[b + s]);
        k_col[b + s] = -1 - i;                   // flags are dead now; mark as "source index"
    }
    __syncthreads();
    for (int s = tid; s < m; s += NT) {
        const int i = -1 - k_col[b + s];
        k_col[b + s] = t_col[b + i];
        k_val[b + s] = t_val[b + i];
    }
    if (tid == 0) { kcnt[row] = s_n; if (scnt) scnt[row] = s_ns; }
}

__global__ __launch_bounds__(256) void csr_compact_kernel(const int* __restrict__ bptr,
                                                          const int* __restrict__ row_ptr,
                                                          const int* __restrict__ k_col,
                                                          const float* __restrict__ k_val,
                                                          int32_t* __restrict__ col, float* __restrict__ val)
{
    const int row = blockIdx.x;
    const int b = bptr[row], o = row_ptr[row], m = row_ptr[row + 1] - o;
    for (int s = threadIdx.x; s < m; s += 256) { col[o + s] = k_col[b + s]; val[o + s] = k_val[b + s]; }
}

// ---- feeds of <= 4096 rows (every training / scoring batch): workgroup-private LDS counters ---------------------
// The global-atomic kernels above spend ~20 us each on a few hundred hot counters.  Here a workgroup counts its 1024
// entries per row in LDS and touches the global counter of a row ONCE (a feed is mostly row-ordered, so that is a
// handful of rows per workgroup); the scatter reserves a slot range per (workgroup, row) the same way and scans the
// row counts itself instead of waiting for a scan launch; the compaction sums the kept counts of the rows before its
// own.  4 launches per feed (count, scatter, per-row order + dedup, compact) instead of 6.
constexpr int CSR_SMALL_ROWS = 4096;     // 2 x 16 KiB of LDS counters in the scatter kernel

template <typename PT>
__global__ __launch_bounds__(1024) void csr_count_lds_kernel(const PT* __restrict__ pos, int64_t nnz,
                                                             int n_rows, int n_cols, int* __restrict__ cnt,
                                                             int* __restrict__ status)
{
    extern __shared__ int sh_rows[];                 // [n_rows]
    const int tid = threadIdx.x;
    for (int r = tid; r < n_rows; r += 1024) sh_rows[r] = 0;
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * 1024 + tid;
    if (i < nnz) {
        const int64_t r = pos[2 * i], c = pos[2 * i + 1];
        if (r < 0 || r >= n_rows || c < 0 || c >= n_cols) atomicOr(status, 1);
        else atomicAdd(&sh_rows[(int)r], 1);
    }
    __syncthreads();
    for (int r = tid; r < n_rows; r += 1024) {
        const int k = sh_rows[r];
        if (k) atomicAdd(&cnt[r], k);
    }
}

// exclusive scan of a[0..n) in LDS by one 1024-thread workgroup; returns the total (all threads)
__device__ int block_exclusive_scan_1024(int* a, int n, int* wsum /*[16]*/)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int carry = 0;
    for (int base = 0; base < n; base += 1024) {
        const int i = base + tid;
        const int v = i < n ? a[i] : 0;
        int x = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int y = __shfl_up(x, d); if (lane >= d) x += y; }
        if (lane == 63) wsum[wave] = x;
        __syncthreads();
        int woff = 0, tot = 0;
        for (int w = 0; w < 16; ++w) { if (w < wave) woff += wsum[w]; tot += wsum[w]; }
        if (i < n) a[i] = carry + woff + x - v;
        carry += tot;
        __syncthreads();
    }
    return carry;
}

template <typename PT>
__global__ __launch_bounds__(1024) void csr_scatter_lds_kernel(const PT* __restrict__ pos,
                                                               const float* __restrict__ val, int val_bcast,
                                                               int64_t nnz, int n_rows, int n_cols,
                                                               const int* __restrict__ cnt, int* __restrict__ cursor,
                                                               int* __restrict__ bptr, int* __restrict__ t_col,
                                                               int* __restrict__ t_feed, float* __restrict__ t_val)
{
    extern __shared__ int sh_rows[];                 // [n_rows] bucket starts | [n_rows] local counts -> slot bases
    __shared__ int wsum[16];
    int* start = sh_rows;
    int* loc = sh_rows + n_rows;
    const int tid = threadIdx.x;
    for (int r = tid; r < n_rows; r += 1024) { start[r] = cnt[r]; loc[r] = 0; }
    __syncthreads();
    const int total = block_exclusive_scan_1024(start, n_rows, wsum);
    if (blockIdx.x == 0) {                           // the bucket starts, for the per-row kernel
        for (int r = tid; r < n_rows; r += 1024) bptr[r] = start[r];
        if (tid == 0) bptr[n_rows] = total;
    }
    const int64_t i = (int64_t)blockIdx.x * 1024 + tid;
    int r = -1, c = 0, rank = 0;
    if (i < nnz) {
        const int64_t r64 = pos[2 * i], c64 = pos[2 * i + 1];
        if (!(r64 < 0 || r64 >= n_rows || c64 < 0 || c64 >= n_cols)) {
            r = (int)r64; c = (int)c64;
            rank = atomicAdd(&loc[r], 1);            // rank among this workgroup's entries of the row
        }
    }
    __syncthreads();
    for (int q = tid; q < n_rows; q += 1024) {       // one global reservation per (workgroup, row)
        const int k = loc[q];
        if (k) loc[q] = start[q] + atomicAdd(&cursor[q], k);
    }
    __syncthreads();
    if (r >= 0) {
        const int slot = loc[r] + rank;
        t_col[slot] = c;
        t_feed[slot] = (int)i;
        t_val[slot] = val[val_bcast ? 0 : i];
    }
}

// one workgroup per row: output start = kept entries of the rows before it (summed here: no scan launch)
__global__ __launch_bounds__(256) void csr_compact_sum_kernel(const int* __restrict__ bptr,
                                                              const int* __restrict__ kcnt, int n_rows,
                                                              const int* __restrict__ k_col,
                                                              const float* __restrict__ k_val,
                                                              int32_t* __restrict__ row_ptr,
                                                              int32_t* __restrict__ col, float* __restrict__ val,
                                                              const int* __restrict__ sflag,
                                                              int32_t* __restrict__ status,
                                                              const int* __restrict__ scnt,
                                                              int32_t* __restrict__ seed_row_ptr,
                                                              int32_t* __restrict__ seed_col)
{
    // scnt / seed_row_ptr / seed_col (all or none): the seed lists of the batch from the same launch (round 6: the two launches of
    // dae_launch_seeds_from_csr folded in -- the drivers' loop builds both for every launch) -- a row's seeds are its first scnt[row]
    // kept entries, their offset the sum of the counts before it
    __shared__ int ws[4], wq[4];
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    int s = 0, q = 0;
    for (int r = tid; r < row; r += 256) { s += kcnt[r]; if (scnt) q += scnt[r]; }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { s += __shfl_xor(s, d); q += __shfl_xor(q, d); }
    if (lane == 0) { ws[tid >> 6] = s; wq[tid >> 6] = q; }
    __syncthreads();
    const int o = ws[0] + ws[1] + ws[2] + ws[3];
    const int so = wq[0] + wq[1] + wq[2] + wq[3];
    const int b = bptr[row], m = kcnt[row], sm = scnt ? scnt[row] : 0;
    if (tid == 0) {
        row_ptr[row] = o;
        if (row == n_rows - 1) row_ptr[n_rows] = o + m;
        if (scnt) {
            seed_row_ptr[row] = so;
            if (row == n_rows - 1) seed_row_ptr[n_rows] = so + sm;
        }
        if (row == 0) *status = *sflag;              // the caller's flag, written once
    }
    for (int s2 = tid; s2 < m; s2 += 256) {
        const int c = k_col[b + s2];
        col[o + s2] = c; val[o + s2] = k_val[b + s2];
        if (s2 < sm) seed_col[so + s2] = c;
    }
}

// The seed lists of a scoring call when the seeds ARE the playlist's own tracks (main_challenge.py:76-88: `seed` is
// playlists[i][0], the ids x_positions feeds; main_train.py:66-89 likewise): the track columns (< n_tracks) of every
// CSR row, as their own CSR.  Columns ascend within a row, so a row's tracks are its first entries: one binary search
// per row and a block scan in one workgroup (rows <= 16384: the scoring path works in slabs of 4096), then a copy.
constexpr int SEED_MAX_ROWS = 16384;
__global__ __launch_bounds__(1024) void seeds_from_csr_kernel(const int32_t* __restrict__ row_ptr,
                                                              const int32_t* __restrict__ col, int B, int n_tracks,
                                                              int32_t* __restrict__ seed_row_ptr,
                                                              int32_t* __restrict__ seed_col, const int32_t* base)
{
    extern __shared__ int s_off[];                                // [B + 1]
    __shared__ int wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int carry = base ? *base : 0;                                 // a later slab of a long batch continues the offsets
    for (int r0 = 0; r0 < B; r0 += 1024) {
        const int r = r0 + tid;
        int cnt = 0;
        if (r < B) {
            int lo = row_ptr[r], hi = row_ptr[r + 1];
            const int beg = lo;
            while (lo < hi) {                                     // first entry with col >= n_tracks
                const int mid = (lo + hi) >> 1;
                if (col[mid] < n_tracks) lo = mid + 1; else hi = mid;
            }
            cnt = lo - beg;
        }
        int v = cnt;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(v, d);
            if (lane >= d) v += o;
        }
        __syncthreads();
        if (lane == 63) wsum[wv] = v;
        __syncthreads();
        int pre = carry;
        for (int w = 0; w < wv; ++w) pre += wsum[w];
        if (r < B) s_off[r] = pre + v - cnt;
        int tot = 0;
        for (int w = 0; w < 16; ++w) tot += wsum[w];
        carry += tot;
    }
    if (tid == 0) s_off[B] = carry;
    __syncthreads();
    for (int r = tid; r <= B; r += 1024) seed_row_ptr[r] = s_off[r];
}

// ... and the copy, a wave per row over the whole chip (inside the scan's single workgroup it was 16 waves walking B / 16
// rows each, one dependent trip to memory per row: 42 us of a 1 024-row launch, 98 us next to another lane's decode)
__global__ __launch_bounds__(256) void seeds_copy_kernel(const int32_t* __restrict__ row_ptr,
                                                         const int32_t* __restrict__ col, int B,
                                                         const int32_t* __restrict__ seed_row_ptr,
                                                         int32_t* __restrict__ seed_col)
{
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= B) return;
    const int dst = seed_row_ptr[r], n = seed_row_ptr[r + 1] - dst, src = row_ptr[r];
    for (int i = lane; i < n; i += 64) seed_col[dst + i] = col[src + i];
}

}  // namespace

int dae_launch_seeds_from_csr(dae_ctx* ctx, const int32_t* row_ptr, const int32_t* col, int B, int n_tracks,
                              int32_t* seed_row_ptr, int32_t* seed_col)
{
    static const char attr_key = 0;          // (B + 1) offsets in LDS: above 64 KiB from B = 16 383 on
    if (dae_first_use(ctx, &attr_key))
        DAE_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&seeds_from_csr_kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize,
                                               (int)((SEED_MAX_ROWS + 1) * sizeof(int))));
    // any number of rows, in slabs of SEED_MAX_ROWS: a slab's offsets start where the one before ended (seed_row_ptr[r0],
    // read on the device: the launches are stream-ordered)
    for (int r0 = 0; r0 < B || r0 == 0; r0 += SEED_MAX_ROWS) {
        const int nb = B - r0 < SEED_MAX_ROWS ? B - r0 : SEED_MAX_ROWS;
        hipLaunchKernelGGL(seeds_from_csr_kernel, dim3(1), dim3(1024), (size_t)(nb + 1) * sizeof(int), ctx->stream, row_ptr + r0,
                           col, nb, n_tracks, seed_row_ptr + r0, seed_col, r0 ? seed_row_ptr + r0 : nullptr);
        DAE_CHECK_LAUNCH(ctx, "seeds_from_csr_kernel");
        if (B <= SEED_MAX_ROWS) break;
    }
    if (B > 0) {
        hipLaunchKernelGGL(seeds_copy_kernel, dim3((B + 3) / 4), dim3(256), 0, ctx->stream, row_ptr, col, B, seed_row_ptr,
                           seed_col);
        DAE_CHECK_LAUNCH(ctx, "seeds_copy_kernel");
    }
    return DAE_OK;
}

namespace {
template <typename PT>
int launch_coo_to_csr(dae_ctx* ctx, const PT* positions, const float* values, int values_broadcast,
                      int64_t nnz, int n_rows, int n_cols, int32_t* row_ptr, int32_t* col, float* val,
                      int32_t* status, int n_tracks, int32_t* seed_row_ptr, int32_t* seed_col)
{
    hipStream_t st = ctx->stream;
    int rc;
    const bool seeds = seed_row_ptr != nullptr;
    // scratch: cnt | cursor | bptr | kcnt | scnt  (n_rows + 1 each), then t_col | t_feed | t_val | k_col | k_val
    // (n_rows + 1 rounded up to a multiple of 4: the fill of cnt | cursor | flag below is then a whole number of 16-byte
    // words and ONE fill kernel -- 6 024 bytes for 750 rows went out as two, 5 us each on the launch's critical path)
    const size_t nr = ((size_t)n_rows + 1 + 3) & ~(size_t)3;
    const size_t ints = 5 * nr + 4 + 5 * (size_t)(nnz > 0 ? nnz : 1);
    if ((rc = dae_reserve(ctx, ctx->csr_tmp, ints * sizeof(int)))) return rc;
    int* cnt = static_cast<int*>(ctx->csr_tmp.p);
    int* cursor = cnt + nr;
    int* sflag = cursor + nr;                      // range flag of this build, cleared with cnt | cursor in ONE fill
    int* bptr = sflag + 4;
    int* kcnt = bptr + nr;
    int* scnt = kcnt + nr;
    int* t_col = scnt + nr;
    int* t_feed = t_col + nnz;
    float* t_val = reinterpret_cast<float*>(t_feed + nnz);
    int* k_col = reinterpret_cast<int*>(t_val + nnz);
    float* k_val = reinterpret_cast<float*>(k_col + nnz);
    static const bool no_small = dae_exp_env("DAE_CSR_GENERIC") != nullptr;            // A/B against the 6-launch path
    if (n_rows <= CSR_SMALL_ROWS && !no_small) {
        // cnt | cursor are adjacent, the caller's status word is cleared with them by one small kernel-free memset each
        DAE_HIP_CHECK(ctx, hipMemsetAsync(cnt, 0, (2 * nr + 4) * sizeof(int), st));
        const int blocks = (int)((nnz + 1023) / 1024) > 0 ? (int)((nnz + 1023) / 1024) : 1;
        const size_t lds = (size_t)n_rows * sizeof(int);
        if (nnz > 0) {
            hipLaunchKernelGGL(csr_count_lds_kernel<PT>, dim3(blocks), dim3(1024), lds, st, positions, nnz, n_rows, n_cols,
                               cnt, sflag);
            DAE_CHECK_LAUNCH(ctx, "csr_count_lds_kernel");
        }
        hipLaunchKernelGGL(csr_scatter_lds_kernel<PT>, dim3(blocks), dim3(1024), 2 * lds, st, positions, values,
                           values_broadcast, nnz, n_rows, n_cols, cnt, cursor, bptr, t_col, t_feed, t_val);
        DAE_CHECK_LAUNCH(ctx, "csr_scatter_lds_kernel");
        if (seeds)                 // (the drivers' loop: playlist rows)
            hipLaunchKernelGGL((csr_row_kernel<512, 64>), dim3(n_rows), dim3(64), 0, st, bptr, t_col, t_feed, t_val, k_col, k_val,
                               kcnt, n_tracks, scnt);
        else
            hipLaunchKernelGGL((csr_row_kernel<CSR_ROW_CAP, 256>), dim3(n_rows), dim3(256), 0, st, bptr, t_col, t_feed, t_val, k_col, k_val,
                               kcnt, n_tracks, static_cast<int*>(nullptr));
        DAE_CHECK_LAUNCH(ctx, "csr_row_kernel");
        hipLaunchKernelGGL(csr_compact_sum_kernel, dim3(n_rows), dim3(256), 0, st, bptr, kcnt, n_rows, k_col, k_val,
                           row_ptr, col, val, sflag, status, seeds ? scnt : nullptr, seed_row_ptr, seed_col);
        DAE_CHECK_LAUNCH(ctx, "csr_compact_sum_kernel");
        return DAE_OK;
    }
    DAE_HIP_CHECK(ctx, hipMemsetAsync(cnt, 0, 2 * nr * sizeof(int), st));          // cnt and cursor
    DAE_HIP_CHECK(ctx, hipMemsetAsync(status, 0, sizeof(int32_t), st));
    if (nnz > 0) {
        int blocks = (int)((nnz + 255) / 256);
        if (blocks > 1024) blocks = 1024;
        hipLaunchKernelGGL(csr_count_kernel<PT>, dim3(blocks), dim3(256), 0, st, positions, nnz, n_rows, n_cols, cnt,
                           status);
        DAE_CHECK_LAUNCH(ctx, "csr_count_kernel");
    }
    hipLaunchKernelGGL(csr_scan_kernel, dim3(1), dim3(1024), 0, st, cnt, n_rows, bptr);
    DAE_CHECK_LAUNCH(ctx, "csr_scan_kernel");
    if (nnz > 0) {
        int blocks = (int)((nnz + 255) / 256);
        if (blocks > 1024) blocks = 1024;
        hipLaunchKernelGGL(csr_scatter_kernel<PT>, dim3(blocks), dim3(256), 0, st, positions, values, values_broadcast,
                           nnz, n_rows, n_cols, bptr, cursor, t_col, t_feed, t_val);
        DAE_CHECK_LAUNCH(ctx, "csr_scatter_kernel");
    }
    hipLaunchKernelGGL((csr_row_kernel<CSR_ROW_CAP, 256>), dim3(n_rows), dim3(256), 0, st, bptr, t_col, t_feed, t_val, k_col, k_val,
                       kcnt, n_tracks, static_cast<int*>(nullptr));
    DAE_CHECK_LAUNCH(ctx, "csr_row_kernel");
    hipLaunchKernelGGL(csr_scan_kernel, dim3(1), dim3(1024), 0, st, kcnt, n_rows, row_ptr);
    DAE_CHECK_LAUNCH(ctx, "csr_scan_kernel");
    hipLaunchKernelGGL(csr_compact_kernel, dim3(n_rows), dim3(256), 0, st, bptr, row_ptr, k_col, k_val, col, val);
    DAE_CHECK_LAUNCH(ctx, "csr_compact_kernel");
    // (more than CSR_SMALL_ROWS rows: the seed lists by their own two launches)
    if (seeds) return dae_launch_seeds_from_csr(ctx, row_ptr, col, n_rows, n_tracks, seed_row_ptr, seed_col);
    return DAE_OK;
}
}  // namespace

int dae_launch_coo_to_csr(dae_ctx* ctx, const int64_t* positions, const float* values, int values_broadcast,
                          int64_t nnz, int n_rows, int n_cols, int32_t* row_ptr, int32_t* col, float* val,
                          int32_t* status)
{
    return launch_coo_to_csr<int64_t>(ctx, positions, values, values_broadcast, nnz, n_rows, n_cols, row_ptr, col, val, status,
                                      0, nullptr, nullptr);
}

// the drivers' loop (pipeline.hip): 32-bit (row, col) pairs as dae_pipeline stages them, and the seed lists (the playlist's own
// tracks: columns < n_tracks) from the same four launches
int dae_launch_coo32_to_csr_seeds(dae_ctx* ctx, const int32_t* positions, const float* values, int values_broadcast,
                                  int64_t nnz, int n_rows, int n_cols, int32_t* row_ptr, int32_t* col, float* val,
                                  int32_t* status, int n_tracks, int32_t* seed_row_ptr, int32_t* seed_col)
{
    return launch_coo_to_csr<int32_t>(ctx, positions, values, values_broadcast, nnz, n_rows, n_cols, row_ptr, col, val, status,
                                      n_tracks, seed_row_ptr, seed_col);
}

// ... and the same for the reference's own 64-bit feed (api.hip dae_title_score: a titled launch of the drivers' loop)
int dae_launch_coo64_to_csr_seeds(dae_ctx* ctx, const int64_t* positions, const float* values, int values_broadcast,
                                  int64_t nnz, int n_rows, int n_cols, int32_t* row_ptr, int32_t* col, float* val,
                                  int32_t* status, int n_tracks, int32_t* seed_row_ptr, int32_t* seed_col)
{
    return launch_coo_to_csr<int64_t>(ctx, positions, values, values_broadcast, nnz, n_rows, n_cols, row_ptr, col, val, status,
                                      n_tracks, seed_row_ptr, seed_col);
}
