// api.hip -- the C ABI of libdae_hip.so (include/dae_hip.h): context, scratch, and the launch
// sequences of the scoring path.  No kernel lives here.
#include <stdarg.h>

#include <climits>
#include <cmath>

#include "dae_internal.h"

thread_local std::string g_dae_create_err;

int dae_fail(dae_ctx* ctx, int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf; else g_dae_create_err = buf;
    return code;
}

int dae_reserve(dae_ctx* ctx, dae_buf& b, size_t bytes)
{
    if (bytes <= b.bytes && b.p) return DAE_OK;
    if (bytes == 0) bytes = 16;
    if (b.p) {
        // growing: drain the stream first, the old buffer may still be in use
        hipError_t e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) return dae_fail(ctx, DAE_ERR_HIP, "sync before regrow: %s", hipGetErrorString(e));
        (void)hipFree(b.p);
        ctx->scratch_total -= b.bytes;
        b.p = nullptr; b.bytes = 0;
    }
    bytes = (bytes + 255) & ~(size_t)255;
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess)
        return dae_fail(ctx, DAE_ERR_NOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    b.p = p; b.bytes = bytes;
    ctx->scratch_total += bytes;
    return DAE_OK;
}


namespace {

struct Plan {            // geometry of the last dae_decode_topk call (dae_last_plan)
    int R_TILE, n_rg, nb_rg, S, n_samp, n_other, fused, ntiles;
};
thread_local Plan g_plan = {0, 0, 0, 0, 0, 0, 0, 0};

int prof_begin(dae_ctx* ctx)
{
    if (!ctx->prof_on) return DAE_OK;
    if (ctx->prof_used + 2 > ctx->prof_ev.size()) {
        for (int i = 0; i < 2; ++i) {
            hipEvent_t ev;
            DAE_HIP_CHECK(ctx, hipEventCreate(&ev));
            ctx->prof_ev.push_back(ev);
        }
    }
    // The pair is handed to the next decode launch (hipExtLaunchKernelGGL start / stop events): it then
    // times the kernel itself, like rocprofv3's kernel trace.  Events recorded on the stream around the
    // launch would add the dispatch gap on both sides (measured 167 vs 154 us for the same launches).
    ctx->prof_armed = true;
    return DAE_OK;
}
int prof_end(dae_ctx* ctx)
{
    if (!ctx->prof_on) return DAE_OK;
    ctx->prof_armed = false;                 // consumed by the launch (prof_used advanced there)
    return DAE_OK;
}

dae_rowgeom geom_for(int dtype, int B, int Hp)
{
    return dtype == DAE_DTYPE_F32 ? dae_row_geometry(B, Hp) : dae_row_geometry_bf16(B, Hp);
}

bool known_dtype(int dtype) { return dtype == DAE_DTYPE_F32 || dtype == DAE_DTYPE_BF16 || dtype == DAE_DTYPE_BF16_EXACT; }

int pack_hidden(dae_ctx* ctx, int dtype, const float* h, int B, int H, const dae_rowgeom& g)
{
    if (dtype == DAE_DTYPE_BF16_EXACT) {
        // the bound behind the exact mode holds for hidden rows in [0, 1]: the packing pass flags the others
        int rc = dae_reserve(ctx, ctx->row_bad, (size_t)g.Bpad * sizeof(int));
        if (rc) return rc;
        return dae_launch_pack_h_bf16(ctx, h, B, H, g, static_cast<int*>(ctx->row_bad.p));
    }
    if (dtype == DAE_DTYPE_F32) {
        int rc = dae_launch_pack_h(ctx, h, B, H, g);          // rewrites the whole image incl. zero pads
        if (rc) return rc;
        ctx->h_geom_key = ((long long)B << 32) | ((long long)H << 12) | (long long)g.R_TILE;
        ctx->h_geom_ptr = ctx->h_packed.p;
        return DAE_OK;
    }
    return dae_launch_pack_h_bf16(ctx, h, B, H, g);
}

const dae_packed* packed_for(dae_ctx* ctx, int dtype, int H)
{
    const dae_packed* pk = dtype == DAE_DTYPE_F32 ? &ctx->pk_f32 : &ctx->pk_bf16;
    if (!pk->valid) { dae_fail(ctx, DAE_ERR_STATE, "decoder weights not prepacked for dtype %d", dtype); return nullptr; }
    if (dtype == DAE_DTYPE_BF16_EXACT && !pk->exact) {
        dae_fail(ctx, DAE_ERR_STATE, "decoder weights not prepacked with DAE_DTYPE_BF16_EXACT"); return nullptr;
    }
    if (pk->H != H) { dae_fail(ctx, DAE_ERR_ARG, "H=%d does not match prepacked H=%d", H, pk->H); return nullptr; }
    return pk;
}

}  // namespace

extern "C" {

int dae_version(void) { return 1000; }

int dae_create(int device, dae_ctx** out)
{
    if (!out) return dae_fail(nullptr, DAE_ERR_ARG, "out is null");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return dae_fail(nullptr, DAE_ERR_HIP, "no HIP device available (%s)", hipGetErrorString(e));
    if (device < 0 || device >= n)
        return dae_fail(nullptr, DAE_ERR_ARG, "device %d out of range [0,%d)", device, n);
    e = hipSetDevice(device);
    if (e != hipSuccess) return dae_fail(nullptr, DAE_ERR_HIP, "hipSetDevice: %s", hipGetErrorString(e));
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) return dae_fail(nullptr, DAE_ERR_HIP, "hipGetDeviceProperties: %s", hipGetErrorString(e));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return dae_fail(nullptr, DAE_ERR_HIP, "device %d is %s; this library is built for gfx950 only",
                        device, prop.gcnArchName);
    dae_ctx* c = new dae_ctx();
    c->device = device;
    *out = c;
    return DAE_OK;
}

int dae_destroy(dae_ctx* ctx)
{
    if (!ctx) return DAE_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (dae_packed* pk : {&ctx->pk_f32, &ctx->pk_bf16})
        if (pk->borrowed) pk->W = pk->bias = pk->bias16 = pk->bias16_lo = pk->bias16_hi = pk->eps = pk->W32 = pk->mix_alpha = pk->mix_beta = pk->mix16_lo = pk->mix16_hi = dae_buf{};
    dae_buf* bufs[] = {&ctx->pk_f32.W, &ctx->pk_f32.bias, &ctx->pk_bf16.W, &ctx->pk_bf16.bias, &ctx->pk_f32.order, &ctx->pk_bf16.order, &ctx->pk_bf16.bias16, &ctx->pk_f32.ident, &ctx->pk_bf16.ident,
                       &ctx->h_packed, &ctx->sample, &ctx->tau, &ctx->sample_top, &ctx->cand,
                       &ctx->cand_cnt, &ctx->gmax, &ctx->dense_tmp, &ctx->h_packed16, &ctx->h_scratch, &ctx->train_a, &ctx->train_b,
                       &ctx->train_c, &ctx->train_d, &ctx->csr_tmp, &ctx->row_bad, &ctx->guard, &ctx->refined, &ctx->refstat, &ctx->pk_bf16.eps, &ctx->pk_bf16.bias16_lo,
                       &ctx->pk_bf16.bias16_hi, &ctx->pk_bf16.W32, &ctx->pk_bf16.mix_alpha, &ctx->pk_bf16.mix_beta, &ctx->pk_bf16.mix16_lo,
                       &ctx->pk_bf16.mix16_hi, &ctx->mix_fhat, &ctx->title_scratch, &ctx->tile_band, &ctx->title_tab, &ctx->audit, &ctx->audit_stat, &ctx->title_y1};
    for (dae_buf* b : bufs)
        if (b->p) (void)hipFree(b->p);
    for (hipEvent_t ev : ctx->prof_ev) (void)hipEventDestroy(ev);
    delete ctx;
    return DAE_OK;
}

int dae_set_stream(dae_ctx* ctx, void* hip_stream)
{
    if (!ctx) return DAE_ERR_ARG;
    ctx->stream = static_cast<hipStream_t>(hip_stream);
    return DAE_OK;
}

const char* dae_last_error(const dae_ctx* ctx)
{
    return ctx ? ctx->err.c_str() : g_dae_create_err.c_str();
}

size_t dae_scratch_bytes(const dae_ctx* ctx) { return ctx ? ctx->scratch_total : 0; }

int dae_profile_enable(dae_ctx* ctx, int on)
{
    if (!ctx) return DAE_ERR_ARG;
    ctx->prof_on = on != 0;
    ctx->prof_used = 0;
    return DAE_OK;
}

int dae_profile_read(dae_ctx* ctx, double* ms_total, int* launches)
{
    if (!ctx) return DAE_ERR_ARG;
    double tot = 0.0;
    int n = 0;
    for (size_t i = 0; i + 1 < ctx->prof_used; i += 2) {
        DAE_HIP_CHECK(ctx, hipEventSynchronize(ctx->prof_ev[i + 1]));
        float ms = 0.f;
        DAE_HIP_CHECK(ctx, hipEventElapsedTime(&ms, ctx->prof_ev[i], ctx->prof_ev[i + 1]));
        tot += ms; ++n;
    }
    ctx->prof_used = 0;
    if (ms_total) *ms_total = tot;
    if (launches) *launches = n;
    return DAE_OK;
}

const char* dae_profile_kernel(const dae_ctx* ctx) { return ctx ? ctx->prof_kernel.c_str() : ""; }

namespace {
__global__ __launch_bounds__(256) void fill_f32_kernel(float* dst, int n, float v)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = v;
}

__global__ __launch_bounds__(64) void clock_probe_kernel(unsigned long long* out, unsigned long long ticks)
{
    if (threadIdx.x != 0) return;
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = r0;
    while (r1 - r0 < ticks) {
        __builtin_amdgcn_s_sleep(16);
        r1 = __builtin_amdgcn_s_memrealtime();
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    r1 = __builtin_amdgcn_s_memrealtime();
    out[0] = c1 - c0;
    out[1] = r1 - r0;
}
}  // namespace

int dae_clock_probe(dae_ctx* ctx, void* hip_stream, int window_us, uint64_t* out2_dev, int* wall_khz_out)
{
    if (!ctx || !out2_dev || window_us < 1 || window_us > 1000000) return dae_fail(ctx, DAE_ERR_ARG, "dae_clock_probe: bad arguments");
    int khz = 0;
    DAE_HIP_CHECK(ctx, hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, ctx->device));
    if (khz <= 0) return dae_fail(ctx, DAE_ERR_STATE, "dae_clock_probe: no wall clock rate");
    if (wall_khz_out) *wall_khz_out = khz;
    hipStream_t st = hip_stream ? static_cast<hipStream_t>(hip_stream) : ctx->stream;
    const unsigned long long ticks = (unsigned long long)window_us * (unsigned long long)khz / 1000ull;
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, st, reinterpret_cast<unsigned long long*>(out2_dev), ticks);
    DAE_CHECK_LAUNCH(ctx, "clock_probe_kernel");
    return DAE_OK;
}

/* geometry of the last dae_decode_topk on this thread:
 * {R_TILE, n_rg, nb_rg, S, n_sample_tiles, n_filter_tiles, fused(0/1), ntiles} */
int dae_last_plan(int32_t out[8])
{
    if (!out) return DAE_ERR_ARG;
    out[0] = g_plan.R_TILE; out[1] = g_plan.n_rg; out[2] = g_plan.nb_rg; out[3] = g_plan.S;
    out[4] = g_plan.n_samp; out[5] = g_plan.n_other; out[6] = g_plan.fused; out[7] = g_plan.ntiles;
    return DAE_OK;
}

int dae_coo_to_csr(dae_ctx* ctx, const int64_t* positions, const float* values, int values_broadcast,
                   int64_t nnz, int n_rows, int n_cols, int32_t* row_ptr, int32_t* col, float* val,
                   int32_t* status)
{
    if (!ctx) return DAE_ERR_ARG;
    if (!row_ptr || !status || (nnz > 0 && (!positions || !values || !col || !val)))
        return dae_fail(ctx, DAE_ERR_ARG, "null pointer");
    if (nnz < 0 || nnz >= (int64_t)1 << 31) return dae_fail(ctx, DAE_ERR_ARG, "nnz=%lld out of range", (long long)nnz);
    if (n_rows < 1 || n_cols < 1) return dae_fail(ctx, DAE_ERR_ARG, "bad shape %d x %d", n_rows, n_cols);
    return dae_launch_coo_to_csr(ctx, positions, values, values_broadcast, nnz, n_rows, n_cols, row_ptr, col, val,
                                 status);
}

int dae_seeds_from_csr(dae_ctx* ctx, const int32_t* row_ptr, const int32_t* col, int B, int n_tracks,
                       int32_t* seed_row_ptr, int32_t* seed_col)
{
    if (!ctx) return DAE_ERR_ARG;
    if (!row_ptr || !col || !seed_row_ptr || !seed_col) return dae_fail(ctx, DAE_ERR_ARG, "null pointer");
    if (B <= 0) return DAE_OK;
    return dae_launch_seeds_from_csr(ctx, row_ptr, col, B, n_tracks, seed_row_ptr, seed_col);
}

int dae_encode(dae_ctx* ctx, const int32_t* row_ptr, const int32_t* col, const float* val,
               const float* W_enc, const float* b_enc, int V, int H, int B,
               float ikp, float kp, uint32_t seed, float* h_out)
{
    if (!ctx) return DAE_ERR_ARG;
    if (!row_ptr || !W_enc || !b_enc || !h_out) return dae_fail(ctx, DAE_ERR_ARG, "null pointer");
    if (H <= 0 || (H % 4) != 0) return dae_fail(ctx, DAE_ERR_ARG, "H=%d must be a positive multiple of 4", H);
    if (B < 0 || V <= 0) return dae_fail(ctx, DAE_ERR_ARG, "bad shape B=%d V=%d", B, V);
    if (!(ikp > 0.f && ikp <= 1.f) || !(kp > 0.f && kp <= 1.f))
        return dae_fail(ctx, DAE_ERR_ARG, "keep probabilities must be in (0,1]");
    if ((reinterpret_cast<uintptr_t>(W_enc) | reinterpret_cast<uintptr_t>(b_enc) |
         reinterpret_cast<uintptr_t>(h_out)) % 16)
        return dae_fail(ctx, DAE_ERR_ARG, "W_enc, b_enc, h_out must be 16-byte aligned");
    return dae_launch_encode(ctx, row_ptr, col, val, W_enc, b_enc, V, H, B, ikp, kp, seed, h_out,
                             nullptr, 0, 0);
}

// a slot that borrows another context's image gets buffers of its own again before it is written
static void unborrow(dae_packed& pk)
{
    if (!pk.borrowed) return;
    pk.W = pk.bias = pk.bias16 = pk.bias16_lo = pk.bias16_hi = pk.eps = pk.W32 = pk.mix_alpha = pk.mix_beta = pk.mix16_lo = pk.mix16_hi = dae_buf{};
    pk.borrowed = false; pk.valid = false; pk.exact = false; pk.order_nrank = -1;
}

int dae_share_decoder(dae_ctx* dst, const dae_ctx* src, int dtype)
{
    if (!dst) return DAE_ERR_ARG;
    if (!src || src == dst) return dae_fail(dst, DAE_ERR_ARG, "dae_share_decoder: needs another context");
    if (dst->device != src->device) return dae_fail(dst, DAE_ERR_ARG, "dae_share_decoder: contexts on different devices");
    if (!known_dtype(dtype)) return dae_fail(dst, DAE_ERR_ARG, "unknown dtype %d", dtype);
    const dae_packed& sp = dtype == DAE_DTYPE_F32 ? src->pk_f32 : src->pk_bf16;
    dae_packed& dp = dtype == DAE_DTYPE_F32 ? dst->pk_f32 : dst->pk_bf16;
    if (!sp.valid || (dtype == DAE_DTYPE_BF16_EXACT && !sp.exact))
        return dae_fail(dst, DAE_ERR_STATE, "dae_share_decoder: the source context holds no such image");
    if (sp.borrowed) return dae_fail(dst, DAE_ERR_ARG, "dae_share_decoder: share from the context that owns the image");
    if (!dp.borrowed) {                                    // drop the own image of this slot (after its last use)
        hipError_t e = hipStreamSynchronize(dst->stream);
        if (e != hipSuccess) return dae_fail(dst, DAE_ERR_HIP, "sync: %s", hipGetErrorString(e));
        for (dae_buf* b : {&dp.W, &dp.bias, &dp.bias16, &dp.bias16_lo, &dp.bias16_hi, &dp.eps, &dp.W32, &dp.mix_alpha, &dp.mix_beta, &dp.mix16_lo, &dp.mix16_hi})
            if (b->p) { (void)hipFree(b->p); dst->scratch_total -= b->bytes; *b = dae_buf{}; }
    }
    const dae_buf order = dp.order, ident = dp.ident;      // the tile lists stay this context's own (small, built lazily)
    dp = sp;
    dp.order = order; dp.ident = ident; dp.order_nrank = -1; dp.order_nsamp = -1;
    dp.borrowed = true;
    int rc = dae_reserve(dst, dp.ident, (size_t)(dp.ntiles > 0 ? dp.ntiles : 1) * sizeof(int));
    if (rc) return rc;
    return dae_launch_tile_iota(dst, static_cast<int*>(dp.ident.p), dp.ntiles);
}

int dae_prepack_decoder(dae_ctx* ctx, const float* W_dec, const float* b_dec, int V, int H,
                        int col_lo, int col_hi, int dtype)
{
    if (!ctx) return DAE_ERR_ARG;
    if (!W_dec || !b_dec) return dae_fail(ctx, DAE_ERR_ARG, "null pointer");
    if (H <= 0 || V <= 0 || col_lo < 0 || col_hi > V || col_lo >= col_hi)
        return dae_fail(ctx, DAE_ERR_ARG, "bad shape V=%d H=%d cols=[%d,%d)", V, H, col_lo, col_hi);
    unborrow(dtype == DAE_DTYPE_F32 ? ctx->pk_f32 : ctx->pk_bf16);
    if (dtype == DAE_DTYPE_F32) return dae_launch_prepack_f32(ctx, W_dec, b_dec, V, H, col_lo, col_hi);
    if (dtype == DAE_DTYPE_BF16) return dae_launch_prepack_bf16(ctx, W_dec, b_dec, V, H, col_lo, col_hi);
    if (dtype == DAE_DTYPE_BF16_EXACT) return dae_launch_prepack_bf16(ctx, W_dec, b_dec, V, H, col_lo, col_hi, 1);
    return dae_fail(ctx, DAE_ERR_ARG, "unknown dtype %d", dtype);
}

int dae_prepack_decoder_rows(dae_ctx* ctx, const float* W_rows, const float* b_rows, int n_rows, int H, int col_lo,
                             int dtype)
{
    if (!ctx) return DAE_ERR_ARG;
    if (!W_rows || !b_rows) return dae_fail(ctx, DAE_ERR_ARG, "null pointer");
    if (n_rows <= 0 || H <= 0 || col_lo < 0 || (int64_t)col_lo + n_rows > INT_MAX)
        return dae_fail(ctx, DAE_ERR_ARG, "bad shape n_rows=%d H=%d col_lo=%d", n_rows, H, col_lo);
    // the prepack kernels index their arguments by GLOBAL column and read rows [col_lo, col_hi) only (prepack_tile_kernel,
    // exact_bounds_kernel, the W32 copy: each forms W + v * H with col_lo <= v < col_hi): the shifted base is never
    // dereferenced below row col_lo
    const float* W = W_rows - (size_t)col_lo * H;
    const float* b = b_rows - (size_t)col_lo;
    return dae_prepack_decoder(ctx, W, b, col_lo + n_rows, H, col_lo, col_lo + n_rows, dtype);
}

int dae_exact_bounds(dae_ctx* ctx, float* eps_out)
{
    if (!ctx) return DAE_ERR_ARG;
    if (!eps_out) return dae_fail(ctx, DAE_ERR_ARG, "null pointer");
    const dae_packed& pk = ctx->pk_bf16;
    if (!pk.valid || !pk.exact) return dae_fail(ctx, DAE_ERR_STATE, "decoder weights not prepacked with DAE_DTYPE_BF16_EXACT");
    DAE_HIP_CHECK(ctx, hipMemcpyAsync(eps_out, pk.eps.p, (size_t)(pk.col_hi - pk.col_lo) * sizeof(float),
                                      hipMemcpyDeviceToDevice, ctx->stream));
    return DAE_OK;
}

int dae_exact_stats_read(dae_ctx* ctx, uint64_t out3[3])
{
    if (!ctx) return DAE_ERR_ARG;
    if (!out3) return dae_fail(ctx, DAE_ERR_ARG, "null pointer");
    out3[0] = out3[1] = out3[2] = 0;
    const int n = ctx->refstat_rows;
    if (!ctx->refstat.p || n <= 0) return DAE_OK;
    std::vector<int> h((size_t)2 * n);
    DAE_HIP_CHECK(ctx, hipMemcpyAsync(h.data(), ctx->refstat.p, h.size() * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    DAE_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    out3[0] = (uint64_t)n;
    for (int r = 0; r < n; ++r) { out3[1] += (uint64_t)h[2 * r]; out3[2] += (uint64_t)h[2 * r + 1]; }
    return DAE_OK;
}

int dae_set_exact_margin(dae_ctx* ctx, float scale)
{
    if (!ctx) return DAE_ERR_ARG;
    if (!(scale > 0.0f) || !(scale <= 1024.0f)) return dae_fail(ctx, DAE_ERR_ARG, "dae_set_exact_margin: scale must be in (0, 1024]");
    ctx->exact_margin = scale;
    return DAE_OK;
}

int dae_exact_guard_read(dae_ctx* ctx, int32_t* violations, int32_t* column)
{
    if (!ctx) return DAE_ERR_ARG;
    int32_t w[2] = {0, -1};
    if (ctx->guard.p) {
        DAE_HIP_CHECK(ctx, hipMemcpyAsync(w, ctx->guard.p, sizeof(w), hipMemcpyDeviceToHost, ctx->stream));
        DAE_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        if (w[0] != 0) DAE_HIP_CHECK(ctx, hipMemsetAsync(ctx->guard.p, 0, sizeof(w), ctx->stream));
    }
    if (violations) *violations = w[0];
    if (column) *column = w[0] ? w[1] : -1;
    return DAE_OK;
}

int dae_exact_guard_words(dae_ctx* ctx, const int32_t** words_dev)
{
    if (!ctx) return DAE_ERR_ARG;
    if (!words_dev) return dae_fail(ctx, DAE_ERR_ARG, "null pointer");
    if (!ctx->guard.p) {
        int rc = dae_reserve(ctx, ctx->guard, DAE_GUARD_BYTES);
        if (rc) return rc;
        DAE_HIP_CHECK(ctx, hipMemsetAsync(ctx->guard.p, 0, DAE_GUARD_BYTES, ctx->stream));
    }
    *words_dev = static_cast<const int32_t*>(ctx->guard.p);
    return DAE_OK;
}

int dae_exact_guard_snapshot(dae_ctx* ctx, int32_t* words_out_dev)
{
    if (!ctx) return DAE_ERR_ARG;
    if (!words_out_dev) return dae_fail(ctx, DAE_ERR_ARG, "null pointer");
    const int32_t* gw = nullptr;
    const int rc = dae_exact_guard_words(ctx, &gw);
    if (rc) return rc;
    DAE_HIP_CHECK(ctx, hipMemcpyAsync(words_out_dev, gw, DAE_GUARD_BYTES, hipMemcpyDeviceToDevice, ctx->stream));
    return DAE_OK;
}

int dae_decode_dense(dae_ctx* ctx, const float* h, int B, int H, int dtype, int apply_sigmoid,
                     float* out, int64_t ld_out)
{
    if (!ctx) return DAE_ERR_ARG;
    if (!h || !out) return dae_fail(ctx, DAE_ERR_ARG, "null pointer");
    if (dtype != DAE_DTYPE_F32 && dtype != DAE_DTYPE_BF16) return dae_fail(ctx, DAE_ERR_ARG, "unknown dtype %d", dtype);
    const dae_packed* pk = packed_for(ctx, dtype, H);
    if (!pk) return DAE_ERR_STATE;
    const int ncols = pk->col_hi - pk->col_lo;
    if (ld_out < ncols) return dae_fail(ctx, DAE_ERR_ARG, "ld_out=%lld < %d columns", (long long)ld_out, ncols);
    if (B <= 0) return DAE_OK;
    const dae_rowgeom g = geom_for(dtype, B, pk->Hp);
    int rc = pack_hidden(ctx, dtype, h, B, H, g);
    if (rc) return rc;
    dae_tileset ts{pk->ntiles, 1, 0, static_cast<const int*>(pk->ident.p)};
    rc = prof_begin(ctx); if (rc) return rc;
    rc = dae_launch_decode_dense_f32(ctx, g, B, ts, apply_sigmoid, INT_MAX, out, ld_out, 0, dtype);
    if (rc) return rc;
    return prof_end(ctx);
}

static long long geom_key(int B, int H, int R_TILE)
{
    return ((long long)B << 32) | ((long long)H << 12) | (long long)R_TILE;
}

// ---- decode + rank with the hidden tile already packed in the context for geometry g, in two halves -----------------
// topk_phase_a: the plan, the threshold sample (phase A) and tau_select -> tau_dst[B] (a valid lower bound, per row, of
//   the k-th largest rankable non-seed logit among THIS image's columns); small problems: the dense logits, tau = -inf.
// topk_phase_b: the filter launch with tau_src[B] (the same values, or larger ones that are still lower bounds of the
//   row's k-th largest logit over ALL shards: dae_score_topk_finish), the exact mode's refine step, the final selection.
// What phase B needs from phase A travels in ctx->tk.
// dtype_in == DAE_DTYPE_BF16_EXACT: h32 = the fp32 hidden rows [B][H] the packed bf16 image was rounded from,
// row_bad (nullable) = rows of h32 outside [0, 1]
static int topk_phase_a(dae_ctx* ctx, const dae_packed* pk, const dae_rowgeom& g, int B, int n_tracks,
                        const int32_t* seed_row_ptr, int k, int dtype_in, float* tau_dst)
{
    int rc;
    dae_topk_state& tk = ctx->tk;
    tk.valid = false;
    const bool exact = dtype_in == DAE_DTYPE_BF16_EXACT;
    const int dtype = exact ? DAE_DTYPE_BF16 : dtype_in;          // the arithmetic of the GEMM launches
    const int ntiles = pk->ntiles;
    const int n_valid_col = n_tracks < pk->col_hi ? n_tracks : pk->col_hi;       // global bound
    int nrank = n_valid_col - pk->col_lo;                                         // ranked columns
    if (nrank < 0) nrank = 0;

    // ---- plan: how many tiles form the threshold sample (phase A) ---------------------------------
    // phase A decodes one tile per SIMD of the row group's workgroups (a full, short round on the
    // matrix pipes whatever the wave count), i.e. every S-th tile; at least ntiles/8 for a tight tau
    const int n_simd = g.nb_rg * 4;
    int rounds = (int)(((double)ntiles / 8.0) / n_simd + 0.5);
    if (rounds < 1) rounds = 1;
    static const int rounds_env = dae_exp_env("DAE_SAMPLE_ROUNDS") ? atoi(dae_exp_env("DAE_SAMPLE_ROUNDS")) : 0;   // experiments
    if (rounds_env > 0) rounds = rounds_env;
    int S = (ntiles + rounds * n_simd - 1) / (rounds * n_simd);
    static const int s_env = dae_exp_env("DAE_SAMPLE_S") ? atoi(dae_exp_env("DAE_SAMPLE_S")) : 0;                   // experiments
    if (s_env > 1) S = s_env;
    // exact mode: the same launches whatever the size (a small problem's "sample" is every tile: S = 1)
    if (exact && S < 2) S = 1;
    const bool fused = (S >= 2 || exact) && nrank > 0;
    const int n_samp = fused ? (ntiles + S - 1) / S : ntiles;
    const int n_other = ntiles - n_samp;
    g_plan = Plan{g.R_TILE, g.n_rg, g.nb_rg, fused ? S : 1, n_samp, n_other, fused ? 1 : 0, ntiles};

    // ---- the sample launch's OWN geometry (round 6) ------------------------------------------------------------------------------
    // Phase A decodes 1 / 11 of the tiles the filter launch decodes, yet on the filter launch's grid (a workgroup per CU) it held
    // every CU for 8 - 15 us: one or two tiles per wave behind a 64 KB hidden-tile fill, with registers / LDS that let nothing of
    // another batch in.  With several batches in flight the step is the SUM of such chip-wide launches (filter + sample + refine:
    // profiles/r06_notes.md).  So the sample takes fewer workgroups per row group -- ~4 tiles per wave slot of the per-wave-maxima
    // kernel (decode_bf16_h256_wavemax_kernel), 8 nbA x 32 maxima per row -- and leaves the other CUs to the other batches' launches.
    // Same sample tiles, same logits; the groups (the tiles one wave decodes, n_ws places apart in the bias order) change, i.e.
    // only how tight tau is.  Only where that kernel applies (bf16 image of hidden 256, 128-row groups, no title mix).
    const bool mixed = ctx->mixT != nullptr;               // dae_set_score_mix: the launches rank the MIXED score
    static const bool no_whole = dae_exp_env("DAE_BF16_KEEP_SAMPLE") != nullptr;       // A/B
    // does a launch of geometry gg take per-WAVE groups?  (the one predicate behind `wave_groups` below)
    auto takes_wave_groups = [&](const dae_rowgeom& gg) {
        const int n_ws = gg.nb_rg * gg.waves;
        const bool enough = (int64_t)((n_samp + n_ws - 1) / n_ws) * gg.nb_rg * 32 >= 4 * (int64_t)k;      // (else: one value per wave slot)
        return fused && enough && dtype == DAE_DTYPE_BF16 && !mixed && (!no_whole || exact) && dae_sample_wave_groups(gg, pk->Hp, n_samp);
    };
    dae_rowgeom gA = g;
    {
        // measured (profiles/r06_notes.md 2, four batches in flight / alone, M playlists/s, exact mode): 256 rows 6.28 -> 6.62 / 3.92 ->
        // 3.59 at 16 workgroups per row group; 1 024 rows 9.10 -> 9.86 / 7.10 -> 6.53 at 8; 2 048 rows 10.85 -> 11.50 / 8.0 -> 8.0 at 8
        // -- a gain only when other batches' launches can use the CUs: taken under dae_set_overlap_hint, ~4 tiles per wave slot for
        // launches of few row groups, ~8 from 8 row groups on
        const int per_slot = g.n_rg >= 8 ? 8 : 4;
        int nbA = ctx->overlap_hint ? ((n_samp + 8 * per_slot - 1) / (8 * per_slot) + DAE_NUM_XCD - 1) / DAE_NUM_XCD * DAE_NUM_XCD : g.nb_rg;
        static const int nba_env = dae_exp_env("DAE_SAMPLE_NB") ? atoi(dae_exp_env("DAE_SAMPLE_NB")) : 0;            // A/B (experiments)
        if (nba_env > 0) nbA = nba_env / DAE_NUM_XCD * DAE_NUM_XCD;
        if (nbA < DAE_NUM_XCD) nbA = DAE_NUM_XCD;
        if (nbA < g.nb_rg) {
            dae_rowgeom t = g;
            t.nb_rg = nbA; t.grid = g.n_rg * nbA;
            if (takes_wave_groups(t) && (int64_t)8 * nbA * 32 >= 4 * (int64_t)k) gA = t;
        }
    }

    // phase A (or the whole problem when it is small): dense logits of the sampled tiles
    const int64_t ld_s = (int64_t)n_samp * 32;
    rc = dae_reserve(ctx, ctx->sample, (size_t)B * ld_s * sizeof(float));
    if (rc) return rc;
    float* sample = static_cast<float*>(ctx->sample.p);
    const int* order = static_cast<const int*>(pk->ident.p);
    if (fused) {
        dae_packed& pkm = dtype == DAE_DTYPE_F32 ? ctx->pk_f32 : ctx->pk_bf16;
        rc = dae_launch_tile_order(ctx, pkm, nrank, n_samp, S);
        if (rc) return rc;
        order = static_cast<const int*>(pkm.order.p);
        // bf16 launches whose sample takes several rounds of the phase-A workgroups (many rows: few workgroups per row group): the
        // sample re-dealt so that a workgroup's tiles of a round come from different popularity bands (decode_f32.hip
        // tile_band_kernel); the list is this context's, rebuilt when the order or the geometry changes
        static const bool no_band = dae_exp_env("DAE_NO_BAND") != nullptr;                    // A/B (experiments build)
        const int n_ws_s = gA.nb_rg * gA.waves;
        // (not when the launch takes per-WAVE groups -- see wave_groups below: there the plain order IS band-dealt)
        const bool wg_early = dtype == DAE_DTYPE_BF16 && ctx->mixT == nullptr && dae_sample_wave_groups(gA, pk->Hp, n_samp) &&
                              ((int64_t)((n_samp + n_ws_s - 1) / n_ws_s) * gA.nb_rg * 32 >= 4 * (int64_t)k);
        if (dtype == DAE_DTYPE_BF16 && n_samp > n_ws_s && !no_band && !wg_early) {
            const void* band_was = ctx->tile_band.p;
            rc = dae_reserve(ctx, ctx->tile_band, (size_t)ntiles * sizeof(int));
            if (rc) return rc;
            if (ctx->tile_band.p != band_was) ctx->band_gen = -1;
            if (ctx->band_gen != pkm.order_gen || ctx->band_nsamp != n_samp || ctx->band_nbrg != gA.nb_rg || ctx->band_waves != gA.waves) {
                rc = dae_launch_tile_band(ctx, order, ntiles, n_samp, gA.nb_rg, gA.waves, static_cast<int*>(ctx->tile_band.p));
                if (rc) return rc;
                ctx->band_gen = pkm.order_gen; ctx->band_nsamp = n_samp; ctx->band_nbrg = gA.nb_rg; ctx->band_waves = gA.waves;
            }
            order = static_cast<const int*>(ctx->tile_band.p);
        }
    }
    dae_tileset tsA{n_samp, fused ? S : 1, fused ? 3 : 0, order};
    float* gmax = nullptr;
    // one maximum per (workgroup of the row group, round of sample tiles, position in the tile)
    const int n_ws_a = gA.nb_rg * gA.waves;
    int64_t ld_g = (int64_t)((n_samp + n_ws_a - 1) / n_ws_a) * gA.nb_rg * 32;
    // ... unless that leaves too few maxima for the rank tau needs (k + seeds): small samples -- vocabulary shards,
    // large batches -- keep one value per wave slot and position, i.e. every sample element
    int gmax_per_wave = ld_g < 4 * (int64_t)k ? 1 : 0;
    if (gmax_per_wave) ld_g *= gA.waves;
    if (mixed && exact) return dae_fail(ctx, DAE_ERR_ARG, "DAE_DTYPE_BF16_EXACT is not available with dae_set_score_mix");
    // the groups are the tiles ONE wave of the filter kernel's shape decodes (decode_bf16_h256_wavemax_kernel: no exchange
    // through LDS, two waves per SIMD) -- 8 nb_rg x 32 maxima per row
    const bool wave_groups = takes_wave_groups(gA);        // (== !gmax_per_wave && ...: the same `enough`)
    if (wave_groups) {
        // (fewer than four tiles per wave slot: waves w and w + 4 share a group -- value 4 -- so that a row has 4 nb_rg x 32 maxima:
        // 4 096 at 1 024 rows, the threshold kernel's 16-key shape)
        static const bool no_pair = dae_exp_env("DAE_WAVEMAX_NOPAIR") != nullptr;            // A/B (experiments build)
        const bool pair = n_samp < 4 * gA.nb_rg * 8 && (int64_t)4 * gA.nb_rg * 32 >= 4 * (int64_t)k && !no_pair;
        gmax_per_wave = pair ? 4 : 3;
        ld_g = (int64_t)(pair ? 4 : 8) * gA.nb_rg * 32;
    }
    if (fused || mixed) {                                  // (the mix lives in the GMAX / FILTER epilogues)
        rc = dae_reserve(ctx, ctx->gmax, (size_t)B * ld_g * sizeof(float));
        if (rc) return rc;
        gmax = static_cast<float*>(ctx->gmax.p);
    }
    // bf16: the filter launch decodes the sample tiles AGAIN instead of phase A storing their dense logits for the
    // threshold kernel to scan: 483 more tiles cost its matrix cores 1.5 us, the 15.8 MB dense buffer (written by phase
    // A through LDS, read back by the threshold kernel, its survivors compacted there) costs more.  Phase A then leaves
    // the group maxima only, the threshold kernel emits no survivors, and every candidate comes from the filter launch.
    // (fp32 keeps the buffer: the same tiles are 13.6 us of its matrix time.)
    // exact mode (DAE_DTYPE_BF16_EXACT): always so, on BOUNDS -- phase A decodes with the bias b - eps (its maxima are
    // lower bounds of fp32 logits, so tau is a valid threshold for the fp32 ranking), the filter launch with b + eps
    // (nothing whose fp32 logit reaches tau is dropped), and the refine step recomputes every survivor in fp32
    const bool whole_b = fused && dtype == DAE_DTYPE_BF16 && ((gmax_per_wave != 1 && !mixed && !no_whole) || exact);
    if (whole_b) g_plan.n_other = ntiles;
    if (!fused) { rc = prof_begin(ctx); if (rc) return rc; }
    // (gA != g only with per-wave groups: the generic kernels run on the filter launch's geometry)
    rc = dae_launch_decode_dense_f32(ctx, wave_groups ? gA : g, B, tsA, 0, n_valid_col, whole_b ? nullptr : sample, ld_s, 1, dtype, gmax, ld_g,
                                     gmax_per_wave, exact ? 1 : 0);
    if (rc) return rc;
    if (!fused) { rc = prof_end(ctx); if (rc) return rc; }

    tk.pk = pk; tk.g = g; tk.B = B; tk.k = k; tk.dtype = dtype; tk.exact = exact; tk.fused = fused; tk.mixed = mixed;
    tk.whole_b = whole_b; tk.S = S; tk.n_samp = n_samp; tk.n_other = n_other; tk.n_valid_col = n_valid_col; tk.nrank = nrank;
    tk.ld_s = ld_s; tk.order = order; tk.sample_cnt = nullptr;
    if (!fused) {                                          // phase B ranks the dense rows; no threshold exists
        if (tau_dst) {
            // (-inf: 0xFF800000 is not a byte pattern hipMemset can write)
            hipLaunchKernelGGL(fill_f32_kernel, dim3((B + 255) / 256), dim3(256), 0, ctx->stream, tau_dst, B, -__builtin_inff());
            DAE_CHECK_LAUNCH(ctx, "fill_f32_kernel");
        }
        tk.valid = true;
        return DAE_OK;
    }

    // tau: the (k + n_seeds)-th largest of the sample's group maxima (written by the phase-A launch: the maximum
    // over the tiles a workgroup decodes together, per position in the tile) -- a valid lower bound of the row's k-th
    // largest rankable non-seed logit -- and, from the same launch, the sample logits >= tau as one flat list per row.
    // No selection over the 15 k dense sample logits of a row happens any more.
    const int64_t pstride = ld_s;                          // worst case (tau = -inf): every sample logit survives
    rc = dae_reserve(ctx, ctx->sample_top, ((size_t)g.Bpad * pstride) * sizeof(uint2) + (size_t)g.Bpad * sizeof(int));
    if (rc) return rc;
    tk.sample_cnt = reinterpret_cast<int*>(static_cast<uint2*>(ctx->sample_top.p) + (size_t)g.Bpad * pstride);
    rc = dae_launch_tau_select(ctx, gmax, ld_g, (int)ld_g, whole_b ? gmax : sample, whole_b ? 0 : ld_s,
                               whole_b ? 0 : (int)ld_s, order, pk->col_lo, B, k,
                               seed_row_ptr, tau_dst, static_cast<uint2*>(ctx->sample_top.p),
                               pstride, tk.sample_cnt);
    if (rc) return rc;
    tk.valid = true;
    return DAE_OK;
}

static int topk_phase_b(dae_ctx* ctx, const float* tau_src, const int32_t* seed_row_ptr, const int32_t* seed_col,
                        int out_kind, float* out_score, int32_t* out_idx, const float* h32, const int* row_bad,
                        bool tau_is_foreign = false)
{
    int rc;
    dae_topk_state& tk = ctx->tk;
    if (!tk.valid) return dae_fail(ctx, DAE_ERR_STATE, "no scoring call in progress on this context");
    tk.valid = false;
    const dae_packed* pk = tk.pk;
    const dae_rowgeom& g = tk.g;
    const int B = tk.B, k = tk.k, dtype = tk.dtype;
    if (tk.mixed) out_kind = DAE_OUT_LOGIT;                // the mixed score is a probability already: it goes out as it is
    dae_topk_args ta;
    memset(&ta, 0, sizeof(ta));
    ta.B = B; ta.k = k;
    ta.bitmap_base = pk->col_lo; ta.bitmap_n = tk.nrank;
    ta.seed_row_ptr = seed_row_ptr; ta.seed_col = seed_col;
    ta.out_kind = out_kind; ta.out_score = out_score; ta.out_idx = out_idx;
    // an exchanged threshold (dae_score_topk_finish) also cuts what the sample left behind under the image's own, lower
    // one; with the own threshold nothing below it was ever kept.  (Exact mode: the lists hold recomputed fp32 logits
    // by then, and tau bounds the fp32 ranking.)
    ta.row_min = (tau_is_foreign && !tk.mixed) ? tau_src : nullptr;
    // batches in flight on other streams (dae_set_overlap_hint), bf16 arithmetic: the filter launch leaves ~112 registers
    // per SIMD lane and 94 KB of LDS on every CU -- the 256-thread selection fits there and runs UNDER the other batch's
    // launch (alone it is slower: 14 vs 11 us); the fp32 launches fill the LDS, nothing fits next to them
    ta.prefer_small = (ctx->overlap_hint && dtype == DAE_DTYPE_BF16) ? 1 : 0;
    if (!tk.fused) {
        dae_dense_src ds{static_cast<const float*>(ctx->sample.p), tk.ld_s, (int)tk.ld_s, pk->col_lo, 1, nullptr};
        return dae_launch_topk_dense(ctx, ds, ta);
    }

    // phase B: everything else through the threshold filter
    const int n_filter = tk.whole_b ? pk->ntiles : tk.n_other;
    const int cap = dae_filter_block_tiles(g, n_filter, dtype, pk->Hp, tk.mixed) * 32;    // worst case: everything passes
    rc = dae_reserve(ctx, ctx->cand, (size_t)g.nb_rg * g.Bpad * cap * sizeof(uint2));
    if (rc) return rc;
    rc = dae_reserve(ctx, ctx->cand_cnt, (size_t)g.nb_rg * g.Bpad * sizeof(int));
    if (rc) return rc;
    dae_tileset tsB{n_filter, tk.S, 3, tk.whole_b ? tk.order : tk.order + tk.n_samp};
    // dae_set_decode_gate: the dominant launch takes every CU, so two of them in flight on two streams only queue
    // behind each other; the gate makes this one wait for the other context's and announces its own end
    if (ctx->gate_wait) DAE_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->stream, ctx->gate_wait, 0));
    rc = prof_begin(ctx); if (rc) return rc;
    rc = dae_launch_decode_filter_f32(ctx, g, B, tsB, tau_src, tk.n_valid_col, static_cast<uint2*>(ctx->cand.p),
                                      static_cast<int*>(ctx->cand_cnt.p), cap, dtype, tk.exact ? 2 : 0);
    if (rc) return rc;
    rc = prof_end(ctx); if (rc) return rc;
    if (ctx->gate_record) DAE_HIP_CHECK(ctx, hipEventRecord(ctx->gate_record, ctx->stream));

    // final: exact top-k of (sample survivors) U (phase-B survivors), seeds removed
    dae_pair_group g0{static_cast<const uint2*>(ctx->sample_top.p), tk.sample_cnt, 0, tk.ld_s, 0, 1, 0};
    dae_pair_group g1{static_cast<const uint2*>(ctx->cand.p), static_cast<const int*>(ctx->cand_cnt.p),
                      (int64_t)g.Bpad * cap, cap, g.Bpad, g.nb_rg, 0};
    if (tk.exact) {
        if (!ctx->guard.p) {                                  // the guard words of this context, zero until a bound fails
            rc = dae_reserve(ctx, ctx->guard, DAE_GUARD_BYTES);
            if (rc) return rc;
            DAE_HIP_CHECK(ctx, hipMemsetAsync(ctx->guard.p, 0, DAE_GUARD_BYTES, ctx->stream));
        }
        dae_exact_src xs{h32, (int64_t)pk->H, pk->H, static_cast<const float*>(pk->W32.p),
                         static_cast<const float*>(pk->bias.p), pk->col_lo, row_bad,
                         static_cast<const float*>(pk->eps.p) + (size_t)pk->ntiles * 32,
                         static_cast<const float*>(pk->eps.p), static_cast<int*>(ctx->guard.p)};
        // the survivors of a row leave the refine launch as ONE compact list (group 0 of the selection: the threshold kernel
        // emits no sample survivors in this mode), the per-workgroup lists of the row are emptied
        rc = dae_reserve(ctx, ctx->refined, (size_t)g.Bpad * DAE_REFINED_CAP * sizeof(uint2) + (size_t)g.Bpad * sizeof(int));
        if (rc) return rc;
        uint2* rf = static_cast<uint2*>(ctx->refined.p);
        int* rf_cnt = reinterpret_cast<int*>(rf + (size_t)g.Bpad * DAE_REFINED_CAP);
        rc = dae_reserve(ctx, ctx->refstat, (size_t)g.Bpad * 2 * sizeof(int));
        if (rc) return rc;
        ctx->refstat_rows = B;
        // k <= 512: the refine launch ends the call itself -- a row's workgroup has its recomputed survivors in LDS, takes the
        // seeds out and orders the k best there (round 5: the selection launch that read them back was 10.6 / 24.0 us of the
        // step at 256 / 1024 rows); larger k keeps the two launches
        static const bool no_fuse = dae_exp_env("DAE_RF_NOFUSE") != nullptr;                  // A/B (experiments build)
        const bool fuse = !no_fuse && dae_exact_refine_can_fuse(ta);
        rc = dae_launch_exact_refine(ctx, g1, xs, B, k, seed_row_ptr, rf, rf_cnt, DAE_REFINED_CAP, static_cast<int*>(ctx->refstat.p),
                                     fuse ? &ta : nullptr);
        // every audit_every-th launch: a sample of the columns the filter launch DROPPED against its own promise (audit.hip) --
        // behind the refine launch, so that the lists are not held up; its verdict lands in the guard words the callers fetch
        if (!rc && ctx->audit_every > 0 && ctx->audit_tiles > 0 && (++ctx->audit_seq % (uint64_t)ctx->audit_every) == 0)
            rc = dae_launch_exact_audit(ctx, g, B, xs, tk.nrank, ctx->audit_tiles);
        if (rc || fuse) return rc;
        dae_pair_group gr{rf, rf_cnt, 0, DAE_REFINED_CAP, 0, 1, 0};
        return dae_launch_topk_pairs(ctx, gr, g1, ta);
    }
    return dae_launch_topk_pairs(ctx, g0, g1, ta);
}

static int decode_topk_core(dae_ctx* ctx, const dae_packed* pk, const dae_rowgeom& g, int B,
                            int n_tracks, const int32_t* seed_row_ptr, const int32_t* seed_col,
                            int k, int out_kind, float* out_score, int32_t* out_idx, int dtype_in,
                            const float* h32 = nullptr, const int* row_bad = nullptr)
{
    int rc = dae_reserve(ctx, ctx->tau, (size_t)g.Bpad * sizeof(float));
    if (rc) return rc;
    rc = topk_phase_a(ctx, pk, g, B, n_tracks, seed_row_ptr, k, dtype_in, static_cast<float*>(ctx->tau.p));
    if (rc) return rc;
    return topk_phase_b(ctx, static_cast<const float*>(ctx->tau.p), seed_row_ptr, seed_col, out_kind, out_score, out_idx,
                        h32, row_bad);
}

static int check_topk_args(dae_ctx* ctx, int dtype, int k, const int32_t* seed_row_ptr,
                           const int32_t* seed_col, const void* out_score, const void* out_idx)
{
    if (!out_score || !out_idx) return dae_fail(ctx, DAE_ERR_ARG, "null pointer");
    if (!known_dtype(dtype)) return dae_fail(ctx, DAE_ERR_ARG, "unknown dtype %d", dtype);
    if (k < 1 || k > DAE_MAX_K) return dae_fail(ctx, DAE_ERR_ARG, "k=%d out of [1,%d]", k, DAE_MAX_K);
    if ((seed_row_ptr == nullptr) != (seed_col == nullptr))
        return dae_fail(ctx, DAE_ERR_ARG, "seed_row_ptr and seed_col must both be given or both null");
    return DAE_OK;
}

// Batches are processed in slabs of DAE_ROW_SLAB rows: the worst-case candidate capacity grows with
// (rows x tiles per workgroup), and a slab keeps it at a few GB whatever batch the caller passes.
constexpr int DAE_ROW_SLAB = 4096;

static int decode_topk_slab(dae_ctx* ctx, const float* h, int B, int H, int dtype, int n_tracks,
                            const int32_t* seed_row_ptr, const int32_t* seed_col, int k, int out_kind,
                            float* out_score, int32_t* out_idx);
static int score_topk_slab(dae_ctx* ctx, const int32_t* row_ptr, const int32_t* col, const float* val,
                           const float* W_enc, const float* b_enc, int V, int H, int B, int dtype,
                           int n_tracks, const int32_t* seed_row_ptr, const int32_t* seed_col,
                           int k, int out_kind, float* out_score, int32_t* out_idx);

int dae_decode_topk(dae_ctx* ctx, const float* h, int B, int H, int dtype, int n_tracks,
                    const int32_t* seed_row_ptr, const int32_t* seed_col, int k, int out_kind,
                    float* out_score, int32_t* out_idx)
{
    for (int r0 = 0; r0 < B || r0 == 0; r0 += DAE_ROW_SLAB) {
        const int nb = B - r0 < DAE_ROW_SLAB ? B - r0 : DAE_ROW_SLAB;
        const int rc = decode_topk_slab(ctx, h ? h + (size_t)r0 * H : h, nb, H, dtype, n_tracks,
                                        seed_row_ptr ? seed_row_ptr + r0 : nullptr, seed_col, k, out_kind,
                                        out_score ? out_score + (size_t)r0 * k : out_score,
                                        out_idx ? out_idx + (size_t)r0 * k : out_idx);
        if (rc || B <= DAE_ROW_SLAB) return rc;
    }
    return DAE_OK;
}

int dae_score_topk(dae_ctx* ctx, const int32_t* row_ptr, const int32_t* col, const float* val,
                   const float* W_enc, const float* b_enc, int V, int H, int B, int dtype,
                   int n_tracks, const int32_t* seed_row_ptr, const int32_t* seed_col,
                   int k, int out_kind, float* out_score, int32_t* out_idx)
{
    // row_ptr / seed_row_ptr hold ABSOLUTE offsets into col / val / seed_col, so a slab is a pointer shift
    for (int r0 = 0; r0 < B || r0 == 0; r0 += DAE_ROW_SLAB) {
        const int nb = B - r0 < DAE_ROW_SLAB ? B - r0 : DAE_ROW_SLAB;
        const int rc = score_topk_slab(ctx, row_ptr ? row_ptr + r0 : row_ptr, col, val, W_enc, b_enc, V, H, nb, dtype,
                                       n_tracks, seed_row_ptr ? seed_row_ptr + r0 : nullptr, seed_col, k, out_kind,
                                       out_score ? out_score + (size_t)r0 * k : out_score,
                                       out_idx ? out_idx + (size_t)r0 * k : out_idx);
        if (rc || B <= DAE_ROW_SLAB) return rc;
    }
    return DAE_OK;
}

static int decode_topk_slab(dae_ctx* ctx, const float* h, int B, int H, int dtype, int n_tracks,
                            const int32_t* seed_row_ptr, const int32_t* seed_col, int k, int out_kind,
                            float* out_score, int32_t* out_idx)
{
    if (!ctx) return DAE_ERR_ARG;
    if (!h) return dae_fail(ctx, DAE_ERR_ARG, "null pointer");
    int rc = check_topk_args(ctx, dtype, k, seed_row_ptr, seed_col, out_score, out_idx);
    if (rc) return rc;
    const dae_packed* pk = packed_for(ctx, dtype, H);
    if (!pk) return DAE_ERR_STATE;
    if (B <= 0) return DAE_OK;
    const dae_rowgeom g = geom_for(dtype, B, pk->Hp);
    rc = pack_hidden(ctx, dtype, h, B, H, g);
    if (rc) return rc;
    return decode_topk_core(ctx, pk, g, B, n_tracks, seed_row_ptr, seed_col, k, out_kind,
                            out_score, out_idx, dtype, h, static_cast<const int*>(ctx->row_bad.p));
}

// encode a slab's rows straight into the packed hidden image of `dtype` (+ the fp32 rows in the exact mode)
static int score_encode(dae_ctx* ctx, const int32_t* row_ptr, const int32_t* col, const float* val,
                        const float* W_enc, const float* b_enc, int V, int H, int B, int dtype,
                        const dae_packed** pk_out, dae_rowgeom* g_out, const float** h32_out)
{
    if (!row_ptr || !W_enc || !b_enc) return dae_fail(ctx, DAE_ERR_ARG, "null pointer");
    if (H <= 0 || (H % 4) != 0) return dae_fail(ctx, DAE_ERR_ARG, "H=%d must be a positive multiple of 4", H);
    if ((reinterpret_cast<uintptr_t>(W_enc) | reinterpret_cast<uintptr_t>(b_enc)) % 16)
        return dae_fail(ctx, DAE_ERR_ARG, "W_enc, b_enc must be 16-byte aligned");
    const dae_packed* pk = packed_for(ctx, dtype, H);
    if (!pk) return DAE_ERR_STATE;
    *pk_out = pk; *h32_out = nullptr;
    int rc;
    if (dtype != DAE_DTYPE_F32) {
        // encode stays fp32 (north_star: bf16 decode GEMM + fp32 encode / top-k); the hidden rows leave the encode
        // kernel rounded to bf16, already in the MFMA operand order (no [B,H] round trip, no re-tiling launch)
        const dae_rowgeom g16 = dae_row_geometry_bf16(B, pk->Hp);
        const int NS = pk->Hp / 16, RB16 = g16.R_TILE / 32;
        const size_t bytes16 = (size_t)g16.n_rg * NS * RB16 * 64 * sizeof(uint4);
        rc = dae_reserve(ctx, ctx->h_packed16, bytes16);
        if (rc) return rc;
        const long long key16 = geom_key(B, H, g16.R_TILE);
        if (ctx->h16_geom_key != key16 || ctx->h16_geom_ptr != ctx->h_packed16.p) {
            // pad rows / pad k of the image are never written by the encode kernel: zero them once
            DAE_HIP_CHECK(ctx, hipMemsetAsync(ctx->h_packed16.p, 0, bytes16, ctx->stream));
            ctx->h16_geom_key = key16;
            ctx->h16_geom_ptr = ctx->h_packed16.p;
        }
        // exact mode: the fp32 rows as well -- the survivors of the bf16 filter are recomputed from them (the encoder's
        // sigmoid keeps them in [0, 1], the precondition of the bound: no row check needed)
        float* h32 = nullptr;
        if (dtype == DAE_DTYPE_BF16_EXACT) {
            rc = dae_reserve(ctx, ctx->h_scratch, (size_t)B * H * sizeof(float));
            if (rc) return rc;
            h32 = static_cast<float*>(ctx->h_scratch.p);
        }
        rc = dae_launch_encode(ctx, row_ptr, col, val, W_enc, b_enc, V, H, B, 1.0f, 1.0f, 0U, h32, nullptr, 0, RB16,
                               nullptr, nullptr, static_cast<unsigned short*>(ctx->h_packed16.p), NS);
        if (rc) return rc;
        *g_out = g16; *h32_out = h32;
        return DAE_OK;
    }
    const dae_rowgeom g = dae_row_geometry(B, pk->Hp);
    const int G = pk->Hp / DAE_KG, RB = g.R_TILE / 32;
    const size_t bytes = (size_t)g.n_rg * G * RB * 64 * sizeof(float4);
    rc = dae_reserve(ctx, ctx->h_packed, bytes);
    if (rc) return rc;
    const long long key = geom_key(B, H, g.R_TILE);
    if (ctx->h_geom_key != key || ctx->h_geom_ptr != ctx->h_packed.p) {
        // pad rows / pad k of the image are never written by the encode kernel: zero them once
        DAE_HIP_CHECK(ctx, hipMemsetAsync(ctx->h_packed.p, 0, bytes, ctx->stream));
        ctx->h_geom_key = key;
        ctx->h_geom_ptr = ctx->h_packed.p;
    }
    rc = dae_launch_encode(ctx, row_ptr, col, val, W_enc, b_enc, V, H, B, 1.0f, 1.0f, 0U, nullptr,
                           static_cast<float*>(ctx->h_packed.p), G, RB);
    if (rc) return rc;
    *g_out = g;
    return DAE_OK;
}

static int score_topk_slab(dae_ctx* ctx, const int32_t* row_ptr, const int32_t* col, const float* val,
                           const float* W_enc, const float* b_enc, int V, int H, int B, int dtype,
                           int n_tracks, const int32_t* seed_row_ptr, const int32_t* seed_col,
                           int k, int out_kind, float* out_score, int32_t* out_idx)
{
    if (!ctx) return DAE_ERR_ARG;
    int rc = check_topk_args(ctx, dtype, k, seed_row_ptr, seed_col, out_score, out_idx);
    if (rc) return rc;
    if (B <= 0) return packed_for(ctx, dtype, H) ? DAE_OK : DAE_ERR_STATE;
    const dae_packed* pk; dae_rowgeom g; const float* h32;
    rc = score_encode(ctx, row_ptr, col, val, W_enc, b_enc, V, H, B, dtype, &pk, &g, &h32);
    if (rc) return rc;
    return decode_topk_core(ctx, pk, g, B, n_tracks, seed_row_ptr, seed_col, k, out_kind,
                            out_score, out_idx, dtype, h32, nullptr);
}

int dae_score_topk_begin(dae_ctx* ctx, const int32_t* row_ptr, const int32_t* col, const float* val,
                         const float* W_enc, const float* b_enc, int V, int H, int B, int dtype,
                         int n_tracks, const int32_t* seed_row_ptr, int k, float* tau_out)
{
    if (!ctx) return DAE_ERR_ARG;
    if (!tau_out) return dae_fail(ctx, DAE_ERR_ARG, "null pointer");
    if (!known_dtype(dtype)) return dae_fail(ctx, DAE_ERR_ARG, "unknown dtype %d", dtype);
    if (k < 1 || k > DAE_MAX_K) return dae_fail(ctx, DAE_ERR_ARG, "k=%d out of [1,%d]", k, DAE_MAX_K);
    if (B < 1 || B > DAE_ROW_SLAB) return dae_fail(ctx, DAE_ERR_ARG, "dae_score_topk_begin takes 1..%d rows (B=%d)", DAE_ROW_SLAB, B);
    if (ctx->mixT) return dae_fail(ctx, DAE_ERR_ARG, "dae_score_topk_begin is not available with dae_set_score_mix");
    const dae_packed* pk; dae_rowgeom g; const float* h32;
    int rc = score_encode(ctx, row_ptr, col, val, W_enc, b_enc, V, H, B, dtype, &pk, &g, &h32);
    if (rc) return rc;
    rc = topk_phase_a(ctx, pk, g, B, n_tracks, seed_row_ptr, k, dtype, tau_out);
    if (rc) return rc;
    ctx->tk.pend_h32 = h32;
    return DAE_OK;
}

int dae_score_topk_finish(dae_ctx* ctx, const float* tau, const int32_t* seed_row_ptr, const int32_t* seed_col,
                          int out_kind, float* out_score, int32_t* out_idx)
{
    if (!ctx) return DAE_ERR_ARG;
    if (!tau || !out_score || !out_idx) return dae_fail(ctx, DAE_ERR_ARG, "null pointer");
    if ((seed_row_ptr == nullptr) != (seed_col == nullptr))
        return dae_fail(ctx, DAE_ERR_ARG, "seed_row_ptr and seed_col must both be given or both null");
    return topk_phase_b(ctx, tau, seed_row_ptr, seed_col, out_kind, out_score, out_idx, ctx->tk.pend_h32, nullptr, true);
}

int dae_decode_mix_term(dae_ctx* ctx, const float* h, int B, int H, int dtype, const float* row_scale, int n_cols,
                        float* outT, int64_t ldT)
{
    if (!ctx) return DAE_ERR_ARG;
    if (!h || !row_scale || !outT) return dae_fail(ctx, DAE_ERR_ARG, "null pointer");
    if (dtype != DAE_DTYPE_F32 && dtype != DAE_DTYPE_BF16) return dae_fail(ctx, DAE_ERR_ARG, "unknown dtype %d", dtype);
    const dae_packed* pk = packed_for(ctx, dtype, H);
    if (!pk) return DAE_ERR_STATE;
    if (ldT < B) return dae_fail(ctx, DAE_ERR_ARG, "ldT=%lld < %d rows", (long long)ldT, B);
    if (B <= 0) return DAE_OK;
    int n_loc = n_cols - pk->col_lo;                       // columns of the image below the global bound
    if (n_loc > pk->col_hi - pk->col_lo) n_loc = pk->col_hi - pk->col_lo;
    if (n_loc <= 0) return DAE_OK;
    const dae_rowgeom g = geom_for(dtype, B, pk->Hp);
    int rc = pack_hidden(ctx, dtype, h, B, H, g);
    if (rc) return rc;
    dae_tileset ts{(n_loc + 31) / 32, 1, 0, static_cast<const int*>(pk->ident.p)};
    return dae_launch_decode_scaled_T(ctx, g, B, ts, row_scale, outT, ldT, dtype);
}

int dae_set_score_mix(dae_ctx* ctx, const float* mixT, int64_t ld, int n_cols, const float* w_title)
{
    if (!ctx) return DAE_ERR_ARG;
    if ((mixT == nullptr) != (w_title == nullptr)) return dae_fail(ctx, DAE_ERR_ARG, "mixT and w_title go together");
    ctx->mixT = mixT; ctx->mix_ld = ld; ctx->mix_w = w_title; ctx->mix_ncols = mixT ? n_cols : 0;
    return DAE_OK;
}

int dae_mix_topk_exact(dae_ctx* title_ctx, dae_ctx* dae_ctx_, const float* feat, int64_t ld_feat, const float* h, int64_t ld_h,
                       int B, const float* w_title, const float* w_playlist, int n_tracks, const int32_t* seed_row_ptr,
                       const int32_t* seed_col, int k, float* out_score, int32_t* out_idx, int32_t* guard_out)
{
    if (!title_ctx) return DAE_ERR_ARG;
    if (!dae_ctx_ || dae_ctx_ == title_ctx) return dae_fail(title_ctx, DAE_ERR_ARG, "dae_mix_topk_exact: needs the DAE's context");
    if (title_ctx->device != dae_ctx_->device) return dae_fail(title_ctx, DAE_ERR_ARG, "dae_mix_topk_exact: contexts on different devices");
    if (!feat || !h || !w_title || !w_playlist) return dae_fail(title_ctx, DAE_ERR_ARG, "null pointer");
    if (B < 0 || n_tracks <= 0) return dae_fail(title_ctx, DAE_ERR_ARG, "bad shape B=%d n_tracks=%d", B, n_tracks);
    int rc = check_topk_args(title_ctx, DAE_DTYPE_BF16_EXACT, k, seed_row_ptr, seed_col, out_score, out_idx);
    if (rc) return rc;
    if (title_ctx->mixT) return dae_fail(title_ctx, DAE_ERR_STATE, "dae_mix_topk_exact takes the DAE's hidden rows itself: clear dae_set_score_mix");
    rc = dae_mix_topk_exact_impl(title_ctx, dae_ctx_, feat, ld_feat, h, ld_h, B, w_title, w_playlist, n_tracks, seed_row_ptr,
                                 seed_col, k, out_score, out_idx);
    if (rc) return rc;
    if (guard_out) {                                       // the guard words as they stand after this launch, in stream order
        if (!title_ctx->guard.p) {                         // (B == 0: nothing ran yet)
            rc = dae_reserve(title_ctx, title_ctx->guard, DAE_GUARD_BYTES);
            if (rc) return rc;
            DAE_HIP_CHECK(title_ctx, hipMemsetAsync(title_ctx->guard.p, 0, DAE_GUARD_BYTES, title_ctx->stream));
        }
        DAE_HIP_CHECK(title_ctx, hipMemcpyAsync(guard_out, title_ctx->guard.p, DAE_GUARD_BYTES, hipMemcpyDeviceToDevice, title_ctx->stream));
    }
    return DAE_OK;
}

int dae_topk_dense(dae_ctx* ctx, const float* logits, int64_t ld, int B, int ncols, int col_base,
                   const int32_t* seed_row_ptr, const int32_t* seed_col, int k, int out_kind,
                   float* out_score, int32_t* out_idx)
{
    if (!ctx) return DAE_ERR_ARG;
    if (!logits || !out_score || !out_idx) return dae_fail(ctx, DAE_ERR_ARG, "null pointer");
    if (ncols < 0 || ld < ncols) return dae_fail(ctx, DAE_ERR_ARG, "bad ncols/ld");
    if ((seed_row_ptr == nullptr) != (seed_col == nullptr))
        return dae_fail(ctx, DAE_ERR_ARG, "seed_row_ptr and seed_col must both be given or both null");
    dae_topk_args ta;
    memset(&ta, 0, sizeof(ta));
    ta.B = B; ta.k = k; ta.out_kind = out_kind;
    ta.bitmap_base = col_base; ta.bitmap_n = ncols;
    ta.seed_row_ptr = seed_row_ptr; ta.seed_col = seed_col;
    ta.out_score = out_score; ta.out_idx = out_idx;
    dae_dense_src ds{logits, ld, ncols, col_base, 1};
    return dae_launch_topk_dense(ctx, ds, ta);
}

int dae_topk_merge(dae_ctx* ctx, int G, int B, int k, const float* cand_logit,
                   const int32_t* cand_idx, int out_kind, float* out_score, int32_t* out_idx)
{
    if (!ctx) return DAE_ERR_ARG;
    if (!cand_logit || !cand_idx || !out_score || !out_idx) return dae_fail(ctx, DAE_ERR_ARG, "null pointer");
    if (G < 1) return dae_fail(ctx, DAE_ERR_ARG, "G=%d", G);
    dae_topk_args ta;
    memset(&ta, 0, sizeof(ta));
    ta.B = B; ta.k = k; ta.out_kind = out_kind;
    ta.out_score = out_score; ta.out_idx = out_idx;
    return dae_launch_topk_soa(ctx, G, cand_logit, cand_idx, ta);
}

int dae_set_train_dtype(dae_ctx* ctx, int dtype)
{
    if (!ctx) return DAE_ERR_ARG;
    if (dtype != DAE_DTYPE_F32 && dtype != DAE_DTYPE_BF16) return dae_fail(ctx, DAE_ERR_ARG, "unknown dtype %d", dtype);
    ctx->train_dtype = dtype;
    return DAE_OK;
}

int dae_train_forward_backward(dae_ctx* ctx,
        const int32_t* x_row_ptr, const int32_t* x_col, const float* x_val,
        const int32_t* y_row_ptr, const int32_t* y_col, const float* y_val,
        const float* W_enc, const float* b_enc, const float* W_dec, const float* b_dec,
        int V, int H, int B, int n_batch, int tied,
        float ikp, float kp, uint32_t seed, float reg_lambda,
        float* gW_enc, float* gb_enc, float* gW_dec, float* gb_dec, float* cost_out)
{
    if (!ctx) return DAE_ERR_ARG;
    if (!x_row_ptr || !y_row_ptr || !W_enc || !b_enc || !b_dec || !gW_enc || !gb_enc || !gb_dec || !cost_out)
        return dae_fail(ctx, DAE_ERR_ARG, "null pointer");
    if (!tied && (!W_dec || (!gW_dec && !ctx->arm_m)))
        return dae_fail(ctx, DAE_ERR_ARG, "untied model needs W_dec and gW_dec (or the armed decoder Adam)");
    if (V <= 0 || H <= 0 || B <= 0 || n_batch <= 0) return dae_fail(ctx, DAE_ERR_ARG, "bad shape");
    if (!(ikp > 0.f && ikp <= 1.f) || !(kp > 0.f && kp <= 1.f))
        return dae_fail(ctx, DAE_ERR_ARG, "keep probabilities must be in (0,1]");
    return dae_train_step_f32(ctx, x_row_ptr, x_col, x_val, y_row_ptr, y_col, y_val, W_enc, b_enc,
                              W_dec, b_dec, V, H, B, n_batch, tied, ikp, kp, seed, reg_lambda,
                              gW_enc, gb_enc, gW_dec, gb_dec, cost_out);
}

static int check_shard(dae_ctx* ctx, int col_lo, int col_hi, int H, int B, float ikp, float kp)
{
    if (col_lo < 0 || col_hi <= col_lo) return dae_fail(ctx, DAE_ERR_ARG, "bad shard [%d,%d)", col_lo, col_hi);
    if (H <= 0 || B <= 0) return dae_fail(ctx, DAE_ERR_ARG, "bad shape");
    if (!(ikp > 0.f && ikp <= 1.f) || !(kp > 0.f && kp <= 1.f))
        return dae_fail(ctx, DAE_ERR_ARG, "keep probabilities must be in (0,1]");
    return DAE_OK;
}

int dae_train_shard_encode(dae_ctx* ctx,
        const int32_t* x_row_ptr, const int32_t* x_col, const float* x_val,
        const float* W_enc_loc, int col_lo, int col_hi, int H, int B,
        float ikp, uint32_t seed, float* pre_partial)
{
    if (!ctx) return DAE_ERR_ARG;
    if (!x_row_ptr || !W_enc_loc || !pre_partial) return dae_fail(ctx, DAE_ERR_ARG, "null pointer");
    int rc = check_shard(ctx, col_lo, col_hi, H, B, ikp, 1.0f);
    if (rc) return rc;
    return dae_train_shard_encode_f32(ctx, x_row_ptr, x_col, x_val, W_enc_loc, col_lo, col_hi, H, B,
                                      ikp, seed, pre_partial);
}

int dae_train_shard_decode(dae_ctx* ctx, const float* pre, const float* b_enc,
        const int32_t* y_row_ptr, const int32_t* y_col, const float* y_val,
        const float* W_enc_loc, const float* W_dec_loc, const float* b_dec_loc,
        int col_lo, int col_hi, int H, int B, int n_batch, int tied,
        float kp, uint32_t seed, float reg_lambda,
        float* gW_out, float* gb_dec_loc, float* dh_partial, float* cost_partial)
{
    if (!ctx) return DAE_ERR_ARG;
    if (!pre || !b_enc || !y_row_ptr || !W_enc_loc || !b_dec_loc || !gW_out || !gb_dec_loc || !dh_partial ||
        !cost_partial)
        return dae_fail(ctx, DAE_ERR_ARG, "null pointer");
    if (!tied && !W_dec_loc) return dae_fail(ctx, DAE_ERR_ARG, "untied model needs W_dec_loc");
    if (n_batch <= 0) return dae_fail(ctx, DAE_ERR_ARG, "bad shape");
    int rc = check_shard(ctx, col_lo, col_hi, H, B, 1.0f, kp);
    if (rc) return rc;
    return dae_train_shard_decode_f32(ctx, pre, b_enc, y_row_ptr, y_col, y_val, W_enc_loc, W_dec_loc,
                                      b_dec_loc, col_lo, col_hi, H, B, n_batch, tied, kp, seed,
                                      reg_lambda, gW_out, gb_dec_loc, dh_partial, cost_partial);
}

int dae_train_shard_finish(dae_ctx* ctx, const float* dh,
        const int32_t* x_row_ptr, const int32_t* x_col, const float* x_val,
        const float* W_enc_loc, const float* b_enc, const float* W_dec_loc, const float* b_dec_loc,
        int col_lo, int col_hi, int H, int B, int tied,
        float ikp, float kp, uint32_t seed, float reg_lambda,
        float* gW_enc_loc, float* gb_enc, float* gW_dec_loc, float* gb_dec_loc)
{
    if (!ctx) return DAE_ERR_ARG;
    if (!dh || !x_row_ptr || !W_enc_loc || !b_enc || !b_dec_loc || !gW_enc_loc || !gb_enc || !gb_dec_loc)
        return dae_fail(ctx, DAE_ERR_ARG, "null pointer");
    if (!tied && (!W_dec_loc || !gW_dec_loc)) return dae_fail(ctx, DAE_ERR_ARG, "untied model needs W_dec_loc and gW_dec_loc");
    int rc = check_shard(ctx, col_lo, col_hi, H, B, ikp, kp);
    if (rc) return rc;
    return dae_train_shard_finish_f32(ctx, dh, x_row_ptr, x_col, x_val, W_enc_loc, b_enc, W_dec_loc,
                                      b_dec_loc, col_lo, col_hi, H, B, tied, ikp, kp, seed, reg_lambda,
                                      gW_enc_loc, gb_enc, gW_dec_loc, gb_dec_loc);
}

int dae_title_features(dae_ctx* ctx, const int32_t* titles, int B, int L, const float* emb, int n_char, int E,
                       const float* conv_w, const float* conv_b, const int32_t* filter_sizes, int n_sizes, int F,
                       float keep_prob, uint32_t seed, float* feat, int64_t ld, int32_t* argmax, float* feat_raw)
{
    if (!ctx) return DAE_ERR_ARG;
    if (!titles || !emb || !conv_w || !conv_b || !filter_sizes || !feat) return dae_fail(ctx, DAE_ERR_ARG, "null pointer");
    if (B <= 0) return DAE_OK;
    if (n_char < 1 || F < 1) return dae_fail(ctx, DAE_ERR_ARG, "bad shape");
    if (!(keep_prob > 0.f && keep_prob <= 1.f)) return dae_fail(ctx, DAE_ERR_ARG, "keep probability must be in (0,1]");
    return dae_launch_title_features(ctx, titles, B, L, emb, n_char, E, conv_w, conv_b, filter_sizes, n_sizes, F,
                                     keep_prob, seed, feat, ld, argmax, feat_raw);
}

int dae_title_prepack_features(dae_ctx* ctx, const float* emb, int n_char, int E, const float* conv_w,
                               const int32_t* filter_sizes, int n_sizes, int F)
{
    if (!ctx) return DAE_ERR_ARG;
    if (emb && (!conv_w || !filter_sizes)) return dae_fail(ctx, DAE_ERR_ARG, "null pointer");
    return dae_launch_title_table(ctx, emb, n_char, E, conv_w, filter_sizes, n_sizes, F);
}

}  // extern "C"

// dae_title_score in two halves (round 6): everything of a titled launch that does not depend on the lane's previous launch --
// title features, the feed -> CSR + seed lists, the DAE's hidden rows, the mixing weights -- and the ranking itself.
// dae_title_score runs them back to back on one stream; dae_pipeline runs the first half of launch n + 1 on its prep stream
// (contexts of its own) while the lane still ranks launch n.
int dae_title_prepare(dae_ctx* tc, dae_ctx* dc, const int64_t* positions, const float* values, int values_broadcast, int64_t nnz,
                      int B, int V, const float* W_enc, const float* b_enc, int H, const int32_t* titles, int L, const float* emb,
                      int n_char, int E, const float* conv_w, const float* conv_b, const int32_t* filter_sizes, int n_sizes, int F,
                      int ld_feat, const float* titles_use, int n_tracks, const dae_title_bufs& b, int32_t* csr_status)
{
    int rc = dae_title_features(tc, titles, B, L, emb, n_char, E, conv_w, conv_b, filter_sizes, n_sizes, F, 1.0f, 0u, b.feat, ld_feat,
                                nullptr, nullptr);
    if (rc) return rc;
    auto from_dc = [&](int r) { return r ? dae_fail(tc, r, "%s", dc->err.c_str()) : DAE_OK; };
    // the feed -> CSR AND the seed lists (the playlist's own tracks) from one group of four launches (round 6: csr.hip)
    rc = from_dc(dae_launch_coo64_to_csr_seeds(dc, positions, values, values_broadcast, nnz, B, V, b.rp, b.col, b.val, csr_status,
                                               n_tracks, b.srp, b.sc));
    if (rc) return rc;
    rc = from_dc(dae_encode(dc, b.rp, b.col, b.val, W_enc, b_enc, V, H, B, 1.0f, 1.0f, 0u, b.h));
    if (rc) return rc;
    return from_dc(dae_mix_weights(dc, b.rp, b.col, b.val, B, 1.0f, 0u, titles_use, b.wt, b.wp));
}

int dae_title_rank(dae_ctx* tc, dae_ctx* dc, int dtype, int B, int V, int H, int ld_feat, const dae_title_bufs& b, int n_tracks, int k,
                   float* out_score, int32_t* out_idx, int32_t* guard_out)
{
    if (dtype == DAE_DTYPE_BF16_EXACT)
        return dae_mix_topk_exact(tc, dc, b.feat, ld_feat, b.h, H, B, b.wt, b.wp, n_tracks, b.srp, b.sc, k, out_score, out_idx, guard_out);
    // fp32 / plain bf16: the fused mix of dae_set_score_mix -- the DAE term transposed, then the title context's threshold path
    // ranks sigmoid(z_title) * w_title + term (the operations and order of dae_mix_scores)
    auto from_dc = [&](int r) { return r ? dae_fail(tc, r, "%s", dc->err.c_str()) : DAE_OK; };
    const size_t nt32 = (size_t)((n_tracks + 31) / 32 * 32 < V ? (n_tracks + 31) / 32 * 32 : V);
    int rc = dae_reserve(tc, tc->title_y1, nt32 * (size_t)B * sizeof(float));
    if (rc) return rc;
    float* y1T = static_cast<float*>(tc->title_y1.p);
    rc = from_dc(dae_decode_mix_term(dc, b.h, B, H, dtype, b.wp, n_tracks, y1T, B));
    if (rc) return rc;
    rc = dae_set_score_mix(tc, y1T, B, (int)nt32, b.wt);
    if (rc) return rc;
    rc = dae_decode_topk(tc, b.feat, B, ld_feat, dtype, n_tracks, b.srp, b.sc, k, DAE_OUT_LOGIT, out_score, out_idx);
    (void)dae_set_score_mix(tc, nullptr, 0, 0, nullptr);
    if (rc) return rc;
    if (guard_out) DAE_HIP_CHECK(tc, hipMemsetAsync(guard_out, 0, DAE_GUARD_BYTES, tc->stream));      // (no bound to guard)
    return DAE_OK;
}

extern "C" {

int dae_title_score(dae_ctx* tc, dae_ctx* dc, int dtype, const int64_t* positions, const float* values, int values_broadcast,
                    int64_t nnz, int n_rows, int V, const float* W_enc, const float* b_enc, int H,
                    const int32_t* titles, int L, const float* emb, int n_char, int E, const float* conv_w,
                    const float* conv_b, const int32_t* filter_sizes, int n_sizes, int F, int ld_feat,
                    const float* titles_use, int n_tracks, int k, float* out_score, int32_t* out_idx,
                    int32_t* guard_out, int32_t* csr_status)
{
    if (!tc) return DAE_ERR_ARG;
    if (!dc || dc == tc) return dae_fail(tc, DAE_ERR_ARG, "dae_title_score: needs the DAE's context");
    if (!known_dtype(dtype)) return dae_fail(tc, DAE_ERR_ARG, "unknown dtype %d", dtype);
    if (!titles || !titles_use || !W_enc || !b_enc || !csr_status || (nnz > 0 && (!positions || !values)))
        return dae_fail(tc, DAE_ERR_ARG, "null pointer");
    if (n_rows <= 0) return DAE_OK;
    if (nnz < 0 || nnz >= (int64_t)1 << 31 || V < 1 || H < 1 || ld_feat < n_sizes * F)
        return dae_fail(tc, DAE_ERR_ARG, "bad shape");
    if (dc->stream != tc->stream) return dae_fail(tc, DAE_ERR_STATE, "both contexts must be bound to the same stream");
    const int B = n_rows;
    // the launch's intermediates, carved out of one buffer of the title context
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t nz = (size_t)(nnz > 0 ? nnz : 1);
    const size_t o_rp = 0, o_col = o_rp + up((size_t)(B + 1) * 4), o_val = o_col + up(nz * 4), o_srp = o_val + up(nz * 4),
                 o_sc = o_srp + up((size_t)(B + 1) * 4), o_h = o_sc + up(nz * 4), o_ft = o_h + up((size_t)B * H * 4),
                 o_wt = o_ft + up((size_t)B * ld_feat * 4), o_wp = o_wt + up((size_t)B * 4), total = o_wp + up((size_t)B * 4);
    int rc = dae_reserve(tc, tc->title_scratch, total);
    if (rc) return rc;
    char* base = static_cast<char*>(tc->title_scratch.p);
    dae_title_bufs b;
    b.rp = reinterpret_cast<int32_t*>(base + o_rp); b.col = reinterpret_cast<int32_t*>(base + o_col);
    b.val = reinterpret_cast<float*>(base + o_val); b.srp = reinterpret_cast<int32_t*>(base + o_srp);
    b.sc = reinterpret_cast<int32_t*>(base + o_sc); b.h = reinterpret_cast<float*>(base + o_h);
    b.feat = reinterpret_cast<float*>(base + o_ft); b.wt = reinterpret_cast<float*>(base + o_wt);
    b.wp = reinterpret_cast<float*>(base + o_wp);
    rc = dae_title_prepare(tc, dc, positions, values, values_broadcast, nnz, B, V, W_enc, b_enc, H, titles, L, emb, n_char, E, conv_w,
                           conv_b, filter_sizes, n_sizes, F, ld_feat, titles_use, n_tracks, b, csr_status);
    if (rc) return rc;
    return dae_title_rank(tc, dc, dtype, B, V, H, ld_feat, b, n_tracks, k, out_score, out_idx, guard_out);
}

int dae_title_score_exact(dae_ctx* tc, dae_ctx* dc, const int64_t* positions, const float* values, int values_broadcast,
                          int64_t nnz, int n_rows, int V, const float* W_enc, const float* b_enc, int H,
                          const int32_t* titles, int L, const float* emb, int n_char, int E, const float* conv_w,
                          const float* conv_b, const int32_t* filter_sizes, int n_sizes, int F, int ld_feat,
                          const float* titles_use, int n_tracks, int k, float* out_score, int32_t* out_idx,
                          int32_t* guard_out, int32_t* csr_status)
{
    return dae_title_score(tc, dc, DAE_DTYPE_BF16_EXACT, positions, values, values_broadcast, nnz, n_rows, V, W_enc, b_enc, H, titles,
                           L, emb, n_char, E, conv_w, conv_b, filter_sizes, n_sizes, F, ld_feat, titles_use, n_tracks, k, out_score,
                           out_idx, guard_out, csr_status);
}

int dae_mix_scores(dae_ctx* ctx, const float* title_score, int64_t ld_title, float* dae_score, int64_t ld_dae,
                   const float* w_title, const float* w_playlist, int B, int ncols)
{
    if (!ctx) return DAE_ERR_ARG;
    if (!title_score || !dae_score || !w_title || !w_playlist) return dae_fail(ctx, DAE_ERR_ARG, "null pointer");
    if (B <= 0 || ncols <= 0) return DAE_OK;
    if (ld_title < ncols || ld_dae < ncols) return dae_fail(ctx, DAE_ERR_ARG, "leading dimension < %d columns", ncols);
    return dae_launch_mix_scores(ctx, title_score, ld_title, dae_score, ld_dae, w_title, w_playlist, B, ncols);
}

int dae_row_sums(dae_ctx* ctx, const int32_t* row_ptr, const int32_t* col, const float* val, int B,
                 float input_keep_prob, uint32_t seed, float* out)
{
    if (!ctx) return DAE_ERR_ARG;
    if (!row_ptr || !out) return dae_fail(ctx, DAE_ERR_ARG, "null pointer");
    if (B <= 0) return DAE_OK;
    if (!(input_keep_prob > 0.f && input_keep_prob <= 1.f)) return dae_fail(ctx, DAE_ERR_ARG, "keep probability must be in (0,1]");
    return dae_launch_row_sums(ctx, row_ptr, col, val, B, input_keep_prob, seed, out);
}

int dae_mix_weights(dae_ctx* ctx, const int32_t* row_ptr, const int32_t* col, const float* val, int B, float input_keep_prob,
                    uint32_t seed, const float* titles_use, float* w_title, float* w_playlist)
{
    if (!ctx) return DAE_ERR_ARG;
    if (!row_ptr || !titles_use || !w_title || !w_playlist) return dae_fail(ctx, DAE_ERR_ARG, "null pointer");
    if (B <= 0) return DAE_OK;
    if (!(input_keep_prob > 0.f && input_keep_prob <= 1.f)) return dae_fail(ctx, DAE_ERR_ARG, "keep probability must be in (0,1]");
    return dae_launch_mix_weights(ctx, row_ptr, col, val, B, input_keep_prob, seed, titles_use, w_title, w_playlist);
}

int dae_title_loss_backward(dae_ctx* ctx, const float* title_logits, int64_t ld_z, const float* dae_score, int64_t ld_d,
                            const int32_t* y_row_ptr, const int32_t* y_col, const float* y_val,
                            const float* w_title, const float* w_playlist, int B, int V, int n_batch,
                            const float* feat, int ld, const float* Output_WT, float* gOutput_WT, float* gOutput_b,
                            float* dfeat, float* cost_out)
{
    if (!ctx) return DAE_ERR_ARG;
    if (!title_logits || !dae_score || !y_row_ptr || !w_title || !w_playlist || !feat || !Output_WT || !gOutput_WT ||
        !gOutput_b || !dfeat || !cost_out)
        return dae_fail(ctx, DAE_ERR_ARG, "null pointer");
    if (B <= 0 || V <= 0 || n_batch <= 0 || ld_z < V || ld_d < V) return dae_fail(ctx, DAE_ERR_ARG, "bad shape");
    return dae_launch_title_loss_backward(ctx, title_logits, ld_z, dae_score, ld_d, y_row_ptr, y_col, y_val, w_title,
                                          w_playlist, B, V, n_batch, feat, ld, Output_WT, gOutput_WT, gOutput_b,
                                          dfeat, cost_out);
}

int dae_title_conv_backward(dae_ctx* ctx, const int32_t* titles, int B, int L, const float* emb, int n_char, int E,
                            const float* conv_w, const int32_t* filter_sizes, int n_sizes, int F,
                            const int32_t* argmax, const float* feat_raw, const float* dfeat, int64_t ld,
                            float keep_prob, uint32_t seed, float* g_emb, float* g_conv_w, float* g_conv_b)
{
    if (!ctx) return DAE_ERR_ARG;
    if (!titles || !emb || !conv_w || !filter_sizes || !argmax || !feat_raw || !dfeat || !g_emb || !g_conv_w || !g_conv_b)
        return dae_fail(ctx, DAE_ERR_ARG, "null pointer");
    if (B <= 0) return DAE_OK;
    if (!(keep_prob > 0.f && keep_prob <= 1.f)) return dae_fail(ctx, DAE_ERR_ARG, "keep probability must be in (0,1]");
    return dae_launch_title_conv_backward(ctx, titles, B, L, emb, n_char, E, conv_w, filter_sizes, n_sizes, F, argmax,
                                          feat_raw, dfeat, ld, keep_prob, seed, g_emb, g_conv_w, g_conv_b);
}

}  // extern "C"

// alpha_t = lr * sqrt(1 - beta2^t) / (1 - beta1^t) with the beta powers kept as fp32 running products, as TF's
// beta1_power / beta2_power variables are.  The products are cached on the context (training calls this with
// t, t, t, t, t+1, ...): restarting the O(t) loop on every call costs milliseconds per step after 10^5 steps.
static float adam_alpha(dae_ctx* ctx, float lr, float beta1, float beta2, int t)
{
    if (ctx->adam_b1 != beta1 || ctx->adam_b2 != beta2 || t < ctx->adam_t) {
        ctx->adam_b1 = beta1; ctx->adam_b2 = beta2; ctx->adam_t = 0; ctx->adam_b1p = 1.0f; ctx->adam_b2p = 1.0f;
    }
    while (ctx->adam_t < t) { ctx->adam_b1p *= beta1; ctx->adam_b2p *= beta2; ++ctx->adam_t; }
    return lr * sqrtf(1.0f - ctx->adam_b2p) / (1.0f - ctx->adam_b1p);
}

extern "C" {

int dae_adam_step(dae_ctx* ctx, float* param, float* m, float* v, const float* grad, int64_t n,
                  float lr, float beta1, float beta2, float eps, int t)
{
    if (!ctx) return DAE_ERR_ARG;
    if (!param || !m || !v || !grad) return dae_fail(ctx, DAE_ERR_ARG, "null pointer");
    if (t < 1) return dae_fail(ctx, DAE_ERR_ARG, "t is the 1-based step count");
    if ((reinterpret_cast<uintptr_t>(param) | reinterpret_cast<uintptr_t>(m) |
         reinterpret_cast<uintptr_t>(v) | reinterpret_cast<uintptr_t>(grad)) % 16)
        return dae_fail(ctx, DAE_ERR_ARG, "param, m, v, grad must be 16-byte aligned");
    const float alpha = adam_alpha(ctx, lr, beta1, beta2, t);
    return dae_launch_adam(ctx, param, m, v, grad, n, alpha, beta1, beta2, eps);
}

int dae_arm_decoder_adam(dae_ctx* ctx, float* m, float* v, float lr, float beta1, float beta2, float eps, int t)
{
    if (!ctx) return DAE_ERR_ARG;
    if (!m || !v) { ctx->arm_m = nullptr; ctx->arm_v = nullptr; return DAE_OK; }          // disarm
    if (t < 1) return dae_fail(ctx, DAE_ERR_ARG, "t is the 1-based step count");
    if ((reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v)) % 16)
        return dae_fail(ctx, DAE_ERR_ARG, "m, v must be 16-byte aligned");
    ctx->arm_m = m; ctx->arm_v = v; ctx->arm_alpha = adam_alpha(ctx, lr, beta1, beta2, t);
    ctx->arm_b1 = beta1; ctx->arm_b2 = beta2; ctx->arm_eps = eps;
    return DAE_OK;
}

int dae_set_overlap_hint(dae_ctx* ctx, int batches_in_flight)
{
    if (!ctx) return DAE_ERR_ARG;
    ctx->overlap_hint = batches_in_flight > 1 ? 1 : 0;
    return DAE_OK;
}

int dae_set_decode_gate(dae_ctx* ctx, void* wait_event, void* record_event)
{
    if (!ctx) return DAE_ERR_ARG;
    ctx->gate_wait = static_cast<hipEvent_t>(wait_event);
    ctx->gate_record = static_cast<hipEvent_t>(record_event);
    return DAE_OK;
}

int dae_set_enc_grad_prezeroed(dae_ctx* ctx, int on)
{
    if (!ctx) return DAE_ERR_ARG;
    ctx->enc_grad_prezeroed = on ? 1 : 0;
    return DAE_OK;
}

static int adam_rows_check(dae_ctx* ctx, const void* param, const void* m, const void* v, const void* state,
                           const void* lr_tab, int n_rows, int row_len, int tab_cap, int t)
{
    if (!ctx) return DAE_ERR_ARG;
    if (!param || !m || !v || !state || !lr_tab) return dae_fail(ctx, DAE_ERR_ARG, "null pointer");
    if (n_rows < 1 || row_len < 1) return dae_fail(ctx, DAE_ERR_ARG, "empty tensor");
    if (t < 1) return dae_fail(ctx, DAE_ERR_ARG, "t is the 1-based step count");
    if (t >= tab_cap) return dae_fail(ctx, DAE_ERR_ARG, "step %d does not fit the alpha table (%d entries)", t, tab_cap);
    return DAE_OK;
}

int dae_adam_rows_begin(dae_ctx* ctx, float* param, float* m, float* v, int32_t* state, float* lr_tab, int tab_cap,
                        int n_rows, int row_len, const int32_t* rows, const int32_t* n_listed_dev, int n_listed_max,
                        float beta1, float beta2, float eps, int t)
{
    int rc = adam_rows_check(ctx, param, m, v, state, lr_tab, n_rows, row_len, tab_cap, t);
    if (rc) return rc;
    if (n_listed_max <= 0) return DAE_OK;
    if (!rows) return dae_fail(ctx, DAE_ERR_ARG, "null row list");
    return dae_launch_adam_rows(ctx, 0, param, m, v, nullptr, state, state + n_rows, lr_tab, n_rows, row_len, rows,
                                n_listed_dev, n_listed_max, 0.0f, beta1, beta2, eps, t);
}

int dae_adam_rows_apply(dae_ctx* ctx, float* param, float* m, float* v, float* grad, int32_t* state, float* lr_tab,
                        int tab_cap, int n_rows, int row_len, const int32_t* rows, const int32_t* n_listed_dev,
                        int n_listed_max, float lr, float beta1, float beta2, float eps, int t)
{
    int rc = adam_rows_check(ctx, param, m, v, state, lr_tab, n_rows, row_len, tab_cap, t);
    if (rc) return rc;
    if (!grad || (n_listed_max > 0 && !rows)) return dae_fail(ctx, DAE_ERR_ARG, "null pointer");
    const float alpha = adam_alpha(ctx, lr, beta1, beta2, t);
    return dae_launch_adam_rows(ctx, 1, param, m, v, grad, state, state + n_rows, lr_tab, n_rows, row_len, rows,
                                n_listed_dev, n_listed_max < 0 ? 0 : n_listed_max, alpha, beta1, beta2, eps, t);
}

int dae_adam_rows_flush(dae_ctx* ctx, float* param, float* m, float* v, int32_t* state, const float* lr_tab,
                        int tab_cap, int n_rows, int row_len, float beta1, float beta2, float eps, int t)
{
    if (t == 0) return DAE_OK;
    int rc = adam_rows_check(ctx, param, m, v, state, lr_tab, n_rows, row_len, tab_cap, t);
    if (rc) return rc;
    return dae_launch_adam_rows(ctx, 2, param, m, v, nullptr, state, nullptr, const_cast<float*>(lr_tab), n_rows, row_len,
                                nullptr, nullptr, 0, 0.0f, beta1, beta2, eps, t);
}

}  // extern "C"
