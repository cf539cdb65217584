// audit.hip -- DAE_DTYPE_BF16_EXACT: the part of the bound the refine launch's guard cannot see.
//
// The exact mode's lists are the fp32 ranking (main_runner/main_challenge.py:28-36 ranks fp32 y_pred) PROVIDED every column the
// bf16 filter launch dropped really lies below the row's threshold in fp32, i.e. provided z32 <= u holds for the upper bound u
// the filter launch computed for it.  The refine launch tests u - 2 eps_c <= z32 <= u for every SURVIVOR it recomputes
// (refine.hip, "BOUND GUARD") -- but a dropped column is never recomputed, so a violation there, the only kind that can change
// a list, stayed invisible (VERDICT r5 Weak #2).  This file samples it: every N-th scoring launch (dae_set_exact_audit) takes a
// pseudo-random set of rankable 32-column tiles -- almost all of their (row, column) elements are dropped ones: ~534 of 140 000
// columns survive per row on the bench model -- and for EVERY row of the launch
//   * recomputes the filter launch's upper bound u for those tiles with the dense bf16 kernel on the b + eps bias image (the
//     same MFMA sequence on the same operands as the filter kernel: the same bits, tests/test_gpu_bf16.py fused == unfused),
//   * recomputes z32 with the canonical chain acc = fmaf(h[k], W[c][k], acc), + b[c] (oracle/dae_oracle.c orc_decode),
//   * counts every element outside [u - 2 eps_c, u] in the context's GUARD WORDS -- the words the refine launch's guard counts
//     in, so DAE.recommend / dae_pipeline_poll re-score such a launch with the fp32 kernels exactly as for a survivor.
// What this does and does not establish: the bound is PROVEN given an error model of v_mfma_f32_32x32x16_bf16's accumulation
// (decode_f32.hip exact_bounds_kernel); survivors are CHECKED always; dropped columns are checked on a SAMPLE (all rows x
// n_tiles x 32 columns every N-th launch; over a loop every tile comes up); nothing else is assumed.
#include "dae_internal.h"

#include <limits.h>
#include <string.h>

namespace {

constexpr int AU_ROWS = 32;        // playlists per workgroup (8 per wave)
constexpr int AU_KC = 128;         // k values staged per pass (49 KB of LDS per workgroup with the hidden rows: three fit a CU)
constexpr int AU_LD = 65;          // dwords per staged k (64 columns + 1: conflict-free both ways)

// n tiles of [0, n_rank_tiles), different ones every launch (a 32-bit mixer on (seed, i))
__global__ __launch_bounds__(64) void audit_pick_kernel(unsigned seed, int n_rank_tiles, int n, int* __restrict__ tiles)
{
    const int i = threadIdx.x;
    if (i >= n) return;
    unsigned x = seed * 0x9E3779B1u + (unsigned)i * 0x85EBCA77u + 0x165667B1u;
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    tiles[i] = (int)(x % (unsigned)n_rank_tiles);
}

struct AuditP {
    const float* h; int64_t ld_h; int H;            // fp32 hidden rows of the launch
    const float* W32; const float* bias; const float* eps; int col_lo, ncols;     // the image: row-major decoder rows, b, eps_c
    const float* u; int64_t ld_u;                    // [B][n_tiles * 32] upper bounds of the sampled tiles (dense bf16 launch, b + eps)
    const int* tiles; int n_tiles;                   // image-local tile ids
    int B, col_bound;                                // rows; global end of the ranked columns
    const int* row_bad;                              // nullable: rows outside the bound's precondition
    int* guard; unsigned long long* stat;            // guard words {violations, a column}; {elements checked, violations}
    float* zout;                                     // non-null: no test here -- z32 goes out ([B][n_tiles * 32], like u) for the
                                                     // caller's own (the exact title mix: mixexact.hip mix_audit_kernel)
};

// workgroup (pair of sampled tiles = 64 columns, block of AU_ROWS playlists); lane = column, a wave walks 8 playlists.
// The 64 decoder rows go through LDS transposed ([k][column]) in passes of AU_KC k; the accumulators of a wave's 8 playlists
// live across the passes, so every chain runs k = 0 .. H-1 in order from +0: the canonical chain, bit for bit.
__global__ __launch_bounds__(256) void exact_audit_kernel(const AuditP p)
{
    extern __shared__ float wt[];                   // [AU_KC][AU_LD] decoder rows, transposed | [AU_ROWS][AU_KC] hidden rows
    float* hs = wt + AU_KC * AU_LD;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int item = blockIdx.x * 2 + (lane >> 5);
    const int tile = item < p.n_tiles ? p.tiles[item] : -1;
    const int cl = tile >= 0 ? tile * 32 + (lane & 31) : -1;                   // image-local column of this lane
    const bool col_ok = cl >= 0 && cl < p.ncols && p.col_lo + cl < p.col_bound;
    const int row0 = blockIdx.y * AU_ROWS;
    float acc[AU_ROWS / 4];
#pragma unroll
    for (int i = 0; i < AU_ROWS / 4; ++i) acc[i] = 0.0f;
    for (int k0 = 0; k0 < p.H; k0 += AU_KC) {
        const int kn = p.H - k0 < AU_KC ? p.H - k0 : AU_KC;
        __syncthreads();                                                       // (the pass before is consumed)
        // stage: wave w takes columns w, w + 4, ...; its lanes read consecutive k of that decoder row
        for (int c = wave; c < 64; c += 4) {
            const int it_c = blockIdx.x * 2 + (c >> 5);
            const int t_c = it_c < p.n_tiles ? p.tiles[it_c] : -1;
            const int cl_c = t_c >= 0 ? t_c * 32 + (c & 31) : -1;
            const bool ok_c = cl_c >= 0 && cl_c < p.ncols;
            const float* wr = p.W32 + (size_t)(ok_c ? cl_c : 0) * p.H + k0;
            for (int k = lane; k < kn; k += 64) wt[k * AU_LD + c] = ok_c ? wr[k] : 0.0f;
        }
        // ... and the block's hidden rows (a broadcast LDS read per k instead of a dependent trip to memory)
        for (int r = wave; r < AU_ROWS; r += 4) {
            const int row = row0 + r;
            const float* hr = p.h + (size_t)(row < p.B ? row : 0) * p.ld_h + k0;
            for (int k = lane; k < kn; k += 64) hs[r * AU_KC + k] = hr[k];
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < AU_ROWS / 4; ++i) {
            const int r = wave + 4 * i;                                        // wave-uniform
            if (row0 + r >= p.B) break;
            const float* hr = hs + r * AU_KC;
            float a = acc[i];
#pragma unroll 8
            for (int k = 0; k < kn; ++k) a = fmaf(hr[k], wt[k * AU_LD + lane], a);
            acc[i] = a;
        }
    }
    unsigned checked = 0, bad_n = 0;
#pragma unroll
    for (int i = 0; i < AU_ROWS / 4; ++i) {
        const int row = row0 + wave + 4 * i;
        if (row >= p.B) break;
        if (p.zout) {
            if (item < p.n_tiles)
                p.zout[(size_t)row * p.ld_u + (size_t)item * 32 + (lane & 31)] = col_ok ? acc[i] + p.bias[cl] : -__builtin_inff();
            continue;
        }
        if (p.row_bad && p.row_bad[row]) continue;                             // (no bound is claimed for such a row)
        bool bad = false;
        if (col_ok) {
            const float z = acc[i] + p.bias[cl];
            const float u = p.u[(size_t)row * p.ld_u + (size_t)item * 32 + (lane & 31)];
            // the filter launch's promise, as refine.hip's guard tests it (the lower end two floats down: the subtraction rounds)
            const float e2 = 2.0f * p.eps[cl] * 1.000001f;
            const float lo = dae_okey_inv(dae_okey(u - e2) - 2u);
            bad = !(z <= u && z >= lo);
            if (bad) { atomicAdd(p.guard, 1); p.guard[1] = p.col_lo + cl; }
        }
        checked += (unsigned)__popcll(__ballot(col_ok));
        bad_n += (unsigned)__popcll(__ballot(bad));
    }
    if (lane == 0 && p.stat && checked) {
        atomicAdd(p.stat, (unsigned long long)checked);
        if (bad_n) atomicAdd(p.stat + 1, (unsigned long long)bad_n);
    }
}

}  // namespace

static int launch_audit_kernel(dae_ctx* ctx, const AuditP& p)
{
    const size_t lds = ((size_t)AU_KC * AU_LD + (size_t)AU_ROWS * AU_KC) * sizeof(float);
    static const char key = 0;
    if (dae_first_use(ctx, &key))
        DAE_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&exact_audit_kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(exact_audit_kernel, dim3((p.n_tiles + 1) / 2, (p.B + AU_ROWS - 1) / AU_ROWS), dim3(256), lds, ctx->stream, p);
    DAE_CHECK_LAUNCH(ctx, "exact_audit_kernel");
    return DAE_OK;
}

// the context's audit block {elements checked, violations} | 64 tile ids (a block of its own, allocated once: the totals
// outlive every launch shape) and this audit's tiles: n of [0, n_rank_tiles), other ones every time
int dae_audit_pick_tiles(dae_ctx* ctx, int n_rank_tiles, int n_tiles, unsigned long long** stat_out, int** tiles_out)
{
    const bool fresh = ctx->audit_stat.p == nullptr;
    int rc = dae_reserve(ctx, ctx->audit_stat, 2 * sizeof(unsigned long long) + (size_t)64 * sizeof(int));
    if (rc) return rc;
    unsigned long long* stat = static_cast<unsigned long long*>(ctx->audit_stat.p);
    int* tiles = reinterpret_cast<int*>(stat + 2);
    if (fresh) DAE_HIP_CHECK(ctx, hipMemsetAsync(stat, 0, 2 * sizeof(unsigned long long), ctx->stream));
    hipLaunchKernelGGL(audit_pick_kernel, dim3(1), dim3(64), 0, ctx->stream, (unsigned)ctx->audit_seq, n_rank_tiles, n_tiles, tiles);
    DAE_CHECK_LAUNCH(ctx, "audit_pick_kernel");
    *stat_out = stat; *tiles_out = tiles;
    return DAE_OK;
}

// z32 = the canonical chain of rows [0, B) of h against the decoder rows of the sampled tiles, + bias, into zout[B][n_tiles * 32]
// (-inf where a sampled tile runs past the image or past col_bound)
int dae_launch_audit_chains(dae_ctx* ctx, const float* h, int64_t ld_h, int H, const float* W32, const float* bias, int ncols,
                            int col_bound, int B, const int* tiles, int n_tiles, float* zout)
{
    AuditP p;
    memset(&p, 0, sizeof(p));
    p.h = h; p.ld_h = ld_h; p.H = H; p.W32 = W32; p.bias = bias; p.ncols = ncols; p.col_bound = col_bound;
    p.tiles = tiles; p.n_tiles = n_tiles; p.B = B; p.ld_u = (int64_t)n_tiles * 32; p.zout = zout;
    return launch_audit_kernel(ctx, p);
}

// one audit of the scoring launch in progress on ctx (its packed bf16 hidden tile and fp32 rows are still in place)
int dae_launch_exact_audit(dae_ctx* ctx, const dae_rowgeom& g, int B, const dae_exact_src& x, int nrank, int n_tiles)
{
    const dae_packed& pk = ctx->pk_bf16;
    if (B <= 0 || nrank <= 0 || n_tiles <= 0 || !x.guard) return DAE_OK;
    if (n_tiles > 64) n_tiles = 64;
    const int n_rank_tiles = (nrank + 31) / 32;
    // the upper bounds [Bpad][n_tiles * 32] grow with the launch
    unsigned long long* stat; int* tiles;
    int rc = dae_audit_pick_tiles(ctx, n_rank_tiles, n_tiles, &stat, &tiles);
    if (rc) return rc;
    rc = dae_reserve(ctx, ctx->audit, (size_t)g.Bpad * n_tiles * 32 * sizeof(float));
    if (rc) return rc;
    float* u = static_cast<float*>(ctx->audit.p);
    // the filter launch's upper bounds of those tiles: dense bf16 decode on the b + eps image (bias_sel 2), nothing masked
    const int64_t ld_u = (int64_t)n_tiles * 32;
    dae_tileset ts{n_tiles, 1, 3, tiles};
    rc = dae_launch_decode_dense_f32(ctx, g, B, ts, 0, INT_MAX, u, ld_u, 1, DAE_DTYPE_BF16, nullptr, 0, 0, 2);
    if (rc) return rc;
    AuditP p;
    memset(&p, 0, sizeof(p));
    p.h = x.h; p.ld_h = x.ld_h; p.H = x.H; p.W32 = x.W32; p.bias = x.bias; p.eps = x.eps; p.col_lo = x.col_lo;
    p.ncols = pk.col_hi - pk.col_lo; p.u = u; p.ld_u = ld_u; p.tiles = tiles; p.n_tiles = n_tiles; p.B = B;
    p.col_bound = pk.col_lo + nrank; p.row_bad = x.row_bad; p.guard = x.guard; p.stat = stat;
    rc = launch_audit_kernel(ctx, p);
    if (rc) return rc;
    ++ctx->audits_run;
    return DAE_OK;
}

extern "C" {

int dae_set_exact_audit(dae_ctx* ctx, int every_n, int n_tiles)
{
    if (!ctx) return DAE_ERR_ARG;
    if (every_n < 0 || n_tiles < 0 || n_tiles > 64)
        return dae_fail(ctx, DAE_ERR_ARG, "dae_set_exact_audit: every_n >= 0 (0 = off), 0 <= n_tiles <= 64");
    ctx->audit_every = every_n; ctx->audit_tiles = n_tiles;
    return DAE_OK;
}

int dae_exact_audit_read(dae_ctx* ctx, uint64_t out3[3])
{
    if (!ctx) return DAE_ERR_ARG;
    if (!out3) return dae_fail(ctx, DAE_ERR_ARG, "null pointer");
    out3[0] = ctx->audits_run; out3[1] = out3[2] = 0;
    if (!ctx->audit_stat.p) return DAE_OK;
    unsigned long long h[2] = {0, 0};
    DAE_HIP_CHECK(ctx, hipMemcpyAsync(h, ctx->audit_stat.p, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    DAE_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    out3[1] = h[0]; out3[2] = h[1];
    return DAE_OK;
}

int dae_set_exact_margin_range(dae_ctx* ctx, int col_from, int col_to, float scale)
{
    if (!ctx) return DAE_ERR_ARG;
    // scale > 0: the factor on those columns' bounds; scale < 0: their UPPER bound is put |scale| logits too low outright -- a
    // forged filter that drops columns it must keep (the only way to make a dropped column CHANGE a list: the audits' tests)
    if (!(scale >= -1024.0f) || !(scale <= 1024.0f) || scale == 0.0f || col_to < col_from)
        return dae_fail(ctx, DAE_ERR_ARG, "dae_set_exact_margin_range: scale in [-1024, 1024] \\ {0}, col_from <= col_to");
    ctx->margin_lo = col_from; ctx->margin_hi = col_to; ctx->margin_scale = scale;
    return DAE_OK;
}

}  // extern "C"
