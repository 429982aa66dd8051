// LayerNorm forward / backward and column sums (bias gradients) for gfx950.  HBM-bound kernels:
// 16-byte vector loads, one wave per row, fp32 statistics, two-pass variance from registers.
//
// Replaces BertLayerNorm = apex FusedLayerNorm / the python fallback (modeling.py:174-192; used at
// :214,239 embeddings, :310,316 attention output, :350,356 FFN output, :429,434 head transform) and
// the dropout that follows the embedding LayerNorm (:240); backward replaces their autograd plus the
// SumBackward of every broadcast bias add.
#include "common.h"
#include <cstdlib>

#define LN_THREADS 256
#define LN_WAVES 4

template <int NP>   // per-lane pieces of 4 halfs (piece k = columns 256k + 4*lane .. +3): H <= 256*NP; at H = 768 every lane holds 12 columns
__global__ __launch_bounds__(LN_THREADS) void layernorm_fwd_kernel(
    const f16* __restrict__ x, int64_t ldx, const f16* __restrict__ gamma, const f16* __restrict__ beta,
    f16* __restrict__ y, int64_t ldy, float* __restrict__ mean, float* __restrict__ rstd, int M, int H, float eps, DropCtx drop, const int32_t* __restrict__ row_map) {
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * LN_WAVES + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * LN_WAVES;
    const float invH = 1.f / (float)H;
    for (int row = wave; row < M; row += nwaves) {
        const f16* xr = x + (int64_t)row * ldx;
        float v[NP][4];
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int c = 256 * k + 4 * lane;
            if (c < H) {
                f16x4 t = ld4(xr + c);
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[k][e] = (float)t[e]; s += v[k][e]; }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[k][e] = 0.f;
            }
        }
        const float mu = wave_sum(s) * invH;
        float q = 0.f;
#pragma unroll
        for (int k = 0; k < NP; ++k)
            if (256 * k + 4 * lane < H) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float d = v[k][e] - mu; q += d * d; }
            }
        const float var = wave_sum(q) * invH;
        const float rs = 1.f / sqrtf(var + eps);
        if (lane == 0) {
            if (mean) mean[row] = mu;
            if (rstd) rstd[row] = rs;
        }
        f16* yr = y + (int64_t)row * ldy;
        const uint32_t rkey = drop.thresh ? drop_rowkey(drop, row_map ? (uint64_t)(uint32_t)row_map[row] : (uint64_t)row) : 0u;   // dropout element = (row, col)
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int c = 256 * k + 4 * lane;
            if (c < H) {
                f16x4 gv = ld4(gamma + c), bv = ld4(beta + c), o;
                float m4[4] = {1.f, 1.f, 1.f, 1.f};
                if (drop.thresh) drop_mult4(drop, rkey, (uint32_t)c, m4);
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (f16)(((float)gv[e] * ((v[k][e] - mu) * rs) + (float)bv[e]) * m4[e]);
                st4_out<VLP_SS_LN>(yr + c, o);
            }
        }
    }
}

// Half-wave form for H % 256 == 0 (H = 768, 1024, 2048: every LayerNorm of the model): lanes 0-31 own one row, lanes 32-63 the next, a
// lane holds NC 16-byte chunks (chunk k = columns 256 k + 8 (lane & 31) .. +7).  16-byte loads, twice the bytes in flight per wave, and
// M / 2 waves (5 344 at M = 10 688) fit the chip in ONE round -- the one-row-per-wave grid needs 1.3 rounds of 8 192 resident waves.
template <int NC>
__global__ __launch_bounds__(LN_THREADS, NC <= 3 ? 6 : 4) void layernorm_fwd_hw_kernel(      // NC = 3: <= 80 VGPRs -> 6 144 resident waves >= the 5 344 of M = 10 688
    const f16* __restrict__ x, int64_t ldx, const f16* __restrict__ gamma, const f16* __restrict__ beta,
    f16* __restrict__ y, int64_t ldy, float* __restrict__ mean, float* __restrict__ rstd, int M, int H, float eps, DropCtx drop, const int32_t* __restrict__ row_map) {
    const int lane = threadIdx.x & 63, hl = lane & 31;
    const int wave = blockIdx.x * LN_WAVES + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * LN_WAVES;
    const float invH = 1.f / (float)H;
    for (int pair = wave; 2 * pair < M; pair += nwaves) {
        const int row = 2 * pair + (lane >> 5);
        const bool live = row < M;
        const f16* xr = x + (int64_t)(live ? row : M - 1) * ldx;
        f16x8 t[NC];
#pragma unroll
        for (int k = 0; k < NC; ++k) t[k] = ld8(xr + 256 * k + 8 * hl);
        float v[NC][8];
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < NC; ++k)
#pragma unroll
            for (int e = 0; e < 8; ++e) { v[k][e] = (float)t[k][e]; s += v[k][e]; }
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);          // within the 32-lane half
        const float mu = s * invH;
        float q = 0.f;
#pragma unroll
        for (int k = 0; k < NC; ++k)
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = v[k][e] - mu; q += d * d; }
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
        const float rs = 1.f / sqrtf(q * invH + eps);
        if (hl == 0 && live) {
            if (mean) mean[row] = mu;
            if (rstd) rstd[row] = rs;
        }
        if (!live) continue;
        f16* yr = y + (int64_t)row * ldy;
        const uint32_t rkey = drop.thresh ? drop_rowkey(drop, row_map ? (uint64_t)(uint32_t)row_map[row] : (uint64_t)row) : 0u;   // dropout element = (row, col)
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            const int c = 256 * k + 8 * hl;
            const f16x8 gv = ld8(gamma + c), bv = ld8(beta + c);
            float m8[8] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};
            if (drop.thresh) drop_mult8(drop, rkey, (uint32_t)c, m8);
            f16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (f16)(((float)gv[e] * ((v[k][e] - mu) * rs) + (float)bv[e]) * m8[e]);
            st8_out<VLP_SS_LN>(yr + c, o);
        }
    }
}

// (Round 4, measured and not kept: a software-pipelined form -- several pairs per wave, the next pair's row requested before the current one
// is reduced and stored, so that reads and writes of different iterations overlap -- is SLOWER at every grid size: 12.4 us at one pair per
// wave (7 spilled registers at the 80-register budget of 6 waves per SIMD), 13.0 / 13.5 / 15.4 / 17.1 us at 2 / 3 / 4 / 5.2 pairs per wave,
// profiles/r04_layernorm_lab.txt.  One pair per wave with every wave resident at once stays the forward.)
// one row per wave up to this many blocks (VLP_LN_BLOCKS overrides for A/B runs)
static int ln_fwd_blocks(int M) {
    static int cap = 0;
    if (!cap) {
        const char* e = getenv("VLP_LN_BLOCKS");
        cap = e ? atoi(e) : 4096;
        if (cap < 1) cap = 4096;
    }
    const int blocks = cdiv(M, LN_WAVES);
    return blocks > cap ? cap : blocks;
}

extern "C" int vlp_layernorm_fwd(const vlp_layernorm_fwd_args* a, void* stream) {
    VLP_CHECK_ARG(a && a->x && a->gamma && a->beta && a->y, "vlp_layernorm_fwd: null operand");
    VLP_ENTER(a->x, "vlp_layernorm_fwd");
    VLP_CHECK_ARG(a->M > 0 && a->H > 0 && a->H % 8 == 0 && a->H <= 4096, "vlp_layernorm_fwd: H=%d must be a multiple of 8 and <= 4096", a->H);
    VLP_CHECK_ARG(a->ldx % 8 == 0 && a->ldy % 8 == 0 && a->ldx >= a->H && a->ldy >= a->H, "vlp_layernorm_fwd: leading dims");
    VLP_CHECK_ARG(((uintptr_t)a->x | (uintptr_t)a->y | (uintptr_t)a->gamma | (uintptr_t)a->beta) % 16 == 0, "vlp_layernorm_fwd: alignment");
    DropCtx d = make_drop(a->dropout_p, a->seed, a->rng_stream);
    const int blocks = ln_fwd_blocks(a->M);
    hipStream_t s = (hipStream_t)stream;
#define LAUNCH_LN_FWD(NP_)                                                                                                              \
    hipLaunchKernelGGL(layernorm_fwd_kernel<NP_>, dim3(blocks), dim3(LN_THREADS), 0, s, (const f16*)a->x, a->ldx, (const f16*)a->gamma, \
                       (const f16*)a->beta, (f16*)a->y, a->ldy, a->mean, a->rstd, a->M, a->H, a->eps, d, a->row_map)
    // VLP_LN_HALFWAVE=0: the one-row-per-wave kernel everywhere (A/B runs).  Read once; the per-call decision is a local (the entry
    // points may be driven from several host threads)
    static const int use_hw = [] { const char* e = getenv("VLP_LN_HALFWAVE"); return e ? atoi(e) : 1; }();
    if (use_hw && a->H % 256 == 0 && a->H <= 2048) {
        bool launched = true;
        const int hblocks = ln_fwd_blocks((a->M + 1) / 2);
#define LAUNCH_LN_HW(NC_)                                                                                                                  \
    hipLaunchKernelGGL(layernorm_fwd_hw_kernel<NC_>, dim3(hblocks), dim3(LN_THREADS), 0, s, (const f16*)a->x, a->ldx, (const f16*)a->gamma, \
                       (const f16*)a->beta, (f16*)a->y, a->ldy, a->mean, a->rstd, a->M, a->H, a->eps, d, a->row_map)
        if (a->H == 768) LAUNCH_LN_HW(3);
        else if (a->H == 256) LAUNCH_LN_HW(1);
        else if (a->H == 512) LAUNCH_LN_HW(2);
        else if (a->H == 1024) LAUNCH_LN_HW(4);
        else if (a->H == 1536) LAUNCH_LN_HW(6);
        else if (a->H == 2048) LAUNCH_LN_HW(8);
        else launched = false;         // no half-wave instantiation for this H: the row-per-wave kernel below
#undef LAUNCH_LN_HW
        if (launched) { VLP_CHECK_LAUNCH("vlp_layernorm_fwd"); return VLP_OK; }
    }
    if (a->H <= 768) LAUNCH_LN_FWD(3);
    else if (a->H <= 1024) LAUNCH_LN_FWD(4);
    else if (a->H <= 2048) LAUNCH_LN_FWD(8);
    else LAUNCH_LN_FWD(16);
#undef LAUNCH_LN_FWD
    VLP_CHECK_LAUNCH("vlp_layernorm_fwd");
    return VLP_OK;
}

// ---------------------------------------------------------------------------------------------
// backward.  Each wave walks rows (grid-stride) keeping per-column partial sums of dgamma / dbeta in
// registers; partials [nblocks][2][H] go to the workspace and a second kernel reduces them.
// A lane owns NP pieces of 4 columns (piece k = columns 256k + 4*lane .. +3, 8-byte loads): at H = 768 every lane is busy with 12
// columns and the kernel needs ~100 VGPRs, so two 8-wave blocks fit a CU.  (The earlier 8-column chunks left half the lanes idle
// in the second chunk at H = 768 and took 142 VGPRs -- one block per CU, 2 waves per SIMD, one row of prefetch each: 3.4 TB/s.)
// Round 3 tried the half-wave form of the forward here (two rows per wave, 16-byte loads, 48 column sums per lane, values recomputed in
// the second pass): 168 VGPRs + spills, 3 waves per SIMD, no next-row prefetch -- 20.9 us against 14.5 us for this kernel (tools/ln_lab.py).
// A third row in flight (prefetch two rows ahead, 128 VGPRs): 17.4 us -- a wave only sees 2.6 rows, the extra requests just queue up front.
// ---------------------------------------------------------------------------------------------
#ifndef LNB_WAVES      // block geometry of the backward kernel; overridable for A/B builds (tools/build_variant_lib.sh)
#define LNB_WAVES 8
#endif
#define LNB_THREADS (64 * LNB_WAVES)
#define LNB_BLOCKS (4096 / LNB_WAVES)
#ifndef LNB_MINW
#define LNB_MINW 4
#endif

template <int NP>
__global__ __launch_bounds__(LNB_THREADS, NP <= 3 ? LNB_MINW : (NP == 4 ? 3 : 2)) void layernorm_bwd_kernel(
    const f16* __restrict__ dy, int64_t lddy, const f16* __restrict__ x, int64_t ldx, const f16* __restrict__ gamma,
    const float* __restrict__ mean, const float* __restrict__ rstd, f16* __restrict__ dx, int64_t lddx,
    f16* __restrict__ dxd, int64_t lddxd, float* __restrict__ part, int M, int H, DropCtx dyd, DropCtx outd, const int32_t* __restrict__ row_map) {
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * LNB_WAVES + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * LNB_WAVES;
    const float invH = 1.f / (float)H;
    float g[NP][4], dg[NP][4], db[NP][4];
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const int c = 256 * k + 4 * lane;
        f16x4 gv = (f16x4){0, 0, 0, 0};
        if (c < H) gv = ld4(gamma + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) { g[k][e] = (float)gv[e]; dg[k][e] = 0.f; db[k][e] = 0.f; }
    }
    // software pipeline: the raw vectors (and statistics) of the NEXT row are requested before the current row is
    // reduced, so two rows of HBM traffic are in flight per wave
    f16x4 xc[NP], dc[NP], xn[NP], dn[NP];
    float mu_c = 0.f, rs_c = 0.f, mu_n = 0.f, rs_n = 0.f;
    auto fetch = [&](int row, f16x4 (&X)[NP], f16x4 (&D)[NP], float& m_, float& r_) {
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int c = 256 * k + 4 * lane;
            if (c < H) {
                X[k] = ld4(x + (int64_t)row * ldx + c);
                D[k] = ld4(dy + (int64_t)row * lddy + c);
            }
        }
        m_ = mean[row];
        r_ = rstd[row];
    };
    if (wave < M) fetch(wave, xc, dc, mu_c, rs_c);
    for (int row = wave; row < M; row += nwaves) {
        if (row + nwaves < M) fetch(row + nwaves, xn, dn, mu_n, rs_n);
        const float mu = mu_c, rs = rs_c;
        const uint64_t drow = ((dyd.thresh | outd.thresh) && row_map) ? (uint64_t)(uint32_t)row_map[row] : (uint64_t)row;     // packed rows: logical index
        const uint32_t rk_dy = dyd.thresh ? drop_rowkey(dyd, drow) : 0u;
        const uint32_t rk_out = outd.thresh ? drop_rowkey(outd, drow) : 0u;
        float xh[NP][4], d[NP][4];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int c = 256 * k + 4 * lane;
            if (c < H) {
                const f16x4 xv = xc[k], dv = dc[k];
                float m4[4] = {1.f, 1.f, 1.f, 1.f};
                if (dyd.thresh) drop_mult4(dyd, rk_dy, (uint32_t)c, m4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float dd = (float)dv[e] * m4[e];
                    xh[k][e] = ((float)xv[e] - mu) * rs;
                    dg[k][e] += dd * xh[k][e];
                    db[k][e] += dd;
                    d[k][e] = dd * g[k][e];
                    s1 += d[k][e];
                    s2 += d[k][e] * xh[k][e];
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) { xh[k][e] = 0.f; d[k][e] = 0.f; }
            }
        }
        s1 = wave_sum(s1) * invH;
        s2 = wave_sum(s2) * invH;
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int c = 256 * k + 4 * lane;
            if (c < H) {
                f16x4 o, od;
                float m4[4] = {1.f, 1.f, 1.f, 1.f};
                if (dxd) drop_mult4(outd, rk_out, (uint32_t)c, m4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float t = rs * (d[k][e] - s1 - xh[k][e] * s2);
                    o[e] = (f16)t;
                    od[e] = (f16)(t * m4[e]);
                }
                st4_out<VLP_SS_LN>(dx + (int64_t)row * lddx + c, o);
                if (dxd) st4_out<VLP_SS_LN>(dxd + (int64_t)row * lddxd + c, od);
            }
        }
#pragma unroll
        for (int k = 0; k < NP; ++k) { xc[k] = xn[k]; dc[k] = dn[k]; }
        mu_c = mu_n;
        rs_c = rs_n;
    }
    // block-level reduction of the 8 waves' column partials through LDS, one partial row per block
    extern __shared__ float lnb_sh[];      // [LNB_WAVES][2H]
    {
        float* sg = lnb_sh + (threadIdx.x >> 6) * 2 * H;
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int c = 256 * k + 4 * lane;
            if (c < H) {
                *reinterpret_cast<f32x4*>(sg + c) = (f32x4){dg[k][0], dg[k][1], dg[k][2], dg[k][3]};
                *reinterpret_cast<f32x4*>(sg + H + c) = (f32x4){db[k][0], db[k][1], db[k][2], db[k][3]};
            }
        }
    }
    __syncthreads();
    float* dst = part + (int64_t)blockIdx.x * 2 * H;
    for (int i = threadIdx.x; i < 2 * H; i += LNB_THREADS) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < LNB_WAVES; ++w) t += lnb_sh[w * 2 * H + i];
        dst[i] = t;
    }
}

// Round 6: the SAME arithmetic for H == 256 NP (every LayerNorm of the model: 768) with a loop body the compiler can count.  In the kernel above
// the next row's loads sit under lane-dependent (c < H) and row-dependent (row + nwaves < M, row_map != 0, dxd != 0) branches, so hipcc's waitcnt pass
// cannot know how many requests are in flight at the join and puts `s_waitcnt vmcnt(0)` in front of the current row's arithmetic (seen in the ISA:
// three of them per iteration): the "prefetch" was waited for in the iteration that issued it -- one row in flight per wave, not two.  Here every
// iteration issues exactly the same loads (the next row is clamped to M - 1 instead of skipped; the optional operands are template switches), so the
// waits are counted and a wave really keeps two rows of HBM traffic in flight.
template <int NP, bool HAS_DXD, bool HAS_MAP>
__global__ __launch_bounds__(LNB_THREADS, LNB_MINW) void layernorm_bwd_full_kernel(
    const f16* __restrict__ dy, int64_t lddy, const f16* __restrict__ x, int64_t ldx, const f16* __restrict__ gamma,
    const float* __restrict__ mean, const float* __restrict__ rstd, f16* __restrict__ dx, int64_t lddx,
    f16* __restrict__ dxd, int64_t lddxd, float* __restrict__ part, int M, DropCtx dyd, DropCtx outd, const int32_t* __restrict__ row_map) {
    constexpr int H = 256 * NP;
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * LNB_WAVES + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * LNB_WAVES;
    constexpr float invH = 1.f / (float)H;
    float g[NP][4], dg[NP][4], db[NP][4];
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const f16x4 gv = ld4(gamma + 256 * k + 4 * lane);
#pragma unroll
        for (int e = 0; e < 4; ++e) { g[k][e] = (float)gv[e]; dg[k][e] = 0.f; db[k][e] = 0.f; }
    }
    f16x4 xc[NP], dc[NP], xn[NP], dn[NP];
    float mu_c, rs_c, mu_n, rs_n;
    int map_c = 0, map_n = 0;
    auto fetch = [&](int row, f16x4 (&X)[NP], f16x4 (&D)[NP], float& m_, float& r_, int& mp_) {
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            X[k] = ld4(x + (int64_t)row * ldx + 256 * k + 4 * lane);
            D[k] = ld4(dy + (int64_t)row * lddy + 256 * k + 4 * lane);
        }
        m_ = mean[row];
        r_ = rstd[row];
        if constexpr (HAS_MAP) mp_ = row_map[row];
    };
    if (wave >= M) {                       // (no rows: only the zero partials below)
        mu_c = rs_c = 0.f;
    } else {
        fetch(wave, xc, dc, mu_c, rs_c, map_c);
    }
    for (int row = wave; row < M; row += nwaves) {
        fetch(min(row + nwaves, M - 1), xn, dn, mu_n, rs_n, map_n);        // ALWAYS (the tail re-reads row M - 1): a countable loop body
        const float mu = mu_c, rs = rs_c;
        const uint64_t drow = HAS_MAP ? (uint64_t)(uint32_t)map_c : (uint64_t)row;
        const uint32_t rk_dy = dyd.thresh ? drop_rowkey(dyd, drow) : 0u;
        const uint32_t rk_out = outd.thresh ? drop_rowkey(outd, drow) : 0u;
        float xh[NP][4], d[NP][4];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int c = 256 * k + 4 * lane;
            float m4[4] = {1.f, 1.f, 1.f, 1.f};
            if (dyd.thresh) drop_mult4(dyd, rk_dy, (uint32_t)c, m4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float dd = (float)dc[k][e] * m4[e];
                xh[k][e] = ((float)xc[k][e] - mu) * rs;
                dg[k][e] += dd * xh[k][e];
                db[k][e] += dd;
                d[k][e] = dd * g[k][e];
                s1 += d[k][e];
                s2 += d[k][e] * xh[k][e];
            }
        }
        s1 = wave_sum(s1) * invH;
        s2 = wave_sum(s2) * invH;
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int c = 256 * k + 4 * lane;
            f16x4 o, od;
            float m4[4] = {1.f, 1.f, 1.f, 1.f};
            if (HAS_DXD) drop_mult4(outd, rk_out, (uint32_t)c, m4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float t = rs * (d[k][e] - s1 - xh[k][e] * s2);
                o[e] = (f16)t;
                od[e] = (f16)(t * m4[e]);
            }
            st4_out<VLP_SS_LN>(dx + (int64_t)row * lddx + c, o);
            if (HAS_DXD) st4_out<VLP_SS_LN>(dxd + (int64_t)row * lddxd + c, od);
        }
#pragma unroll
        for (int k = 0; k < NP; ++k) { xc[k] = xn[k]; dc[k] = dn[k]; }
        mu_c = mu_n;
        rs_c = rs_n;
        map_c = map_n;
    }
    extern __shared__ float lnb_sh[];      // [LNB_WAVES][2H]
    {
        float* sg = lnb_sh + (threadIdx.x >> 6) * 2 * H;
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int c = 256 * k + 4 * lane;
            *reinterpret_cast<f32x4*>(sg + c) = (f32x4){dg[k][0], dg[k][1], dg[k][2], dg[k][3]};
            *reinterpret_cast<f32x4*>(sg + H + c) = (f32x4){db[k][0], db[k][1], db[k][2], db[k][3]};
        }
    }
    __syncthreads();
    float* dst = part + (int64_t)blockIdx.x * 2 * H;
    for (int i = threadIdx.x; i < 2 * H; i += LNB_THREADS) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < LNB_WAVES; ++w) t += lnb_sh[w * 2 * H + i];
        dst[i] = t;
    }
}

// out[0..H) = dgamma, out[H..2H) = dbeta from part[nparts][2H].  Block = 64 columns x 16 partial-groups.
__global__ __launch_bounds__(1024) void ln_bwd_reduce_kernel(const float* __restrict__ part, int nparts, int H, f16* dgamma, f16* dbeta, int beta) {
    __shared__ float sh[16][64];
    const int cl = threadIdx.x & 63, pg = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + cl;
    float s = 0.f;
    if (i < 2 * H)
        for (int p = pg; p < nparts; p += 16) s += part[(int64_t)p * 2 * H + i];
    sh[pg][cl] = s;
    __syncthreads();
    if (pg == 0 && i < 2 * H) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += sh[k][cl];
        f16* dst = i < H ? dgamma + i : dbeta + (i - H);
        *dst = (f16)(beta ? (float)*dst + t : t);
    }
}

// row-walking blocks of a backward launch (= partial rows in the workspace); VLP_LNB_BLOCKS lowers it for A/B runs
static int lnb_blocks(int M) {
    static int cap = 0;
    if (!cap) {
        const char* e = getenv("VLP_LNB_BLOCKS");
        cap = e ? atoi(e) : LNB_BLOCKS;
        if (cap < 1 || cap > LNB_BLOCKS) cap = LNB_BLOCKS;
    }
    const int blocks = cdiv(M, LNB_WAVES);
    if (blocks <= cap) return blocks;
    // every wave the SAME number of rows: with the cap alone a wave walks M / (cap * 8) = 2.6 rows at M = 10 688 -- 61 % of the waves do
    // three, the rest two, and the kernel ends when the three-row waves do.  R = ceil(rows per wave) rows for (almost) all of them:
    // 446 blocks instead of 512 at M = 10 688: 14.8 -> 14.1 us plain, 18.4 -> 17.3 us with the dropped twin (profiles/r04_layernorm_lab.txt)
    const int R = cdiv(M, cap * LNB_WAVES);
    return cdiv(M, R * LNB_WAVES);
}

extern "C" int64_t vlp_layernorm_bwd_workspace_bytes(int32_t H) {
    return (int64_t)LNB_BLOCKS * 2 * H * (int64_t)sizeof(float);
}

extern "C" int vlp_layernorm_bwd(const vlp_layernorm_bwd_args* a, void* stream) {
    VLP_CHECK_ARG(a && a->dy && a->x && a->gamma && a->mean && a->rstd && a->dx && a->dgamma && a->dbeta, "vlp_layernorm_bwd: null operand");
    VLP_ENTER(a->dy, "vlp_layernorm_bwd");
    VLP_CHECK_ARG(a->M > 0 && a->H > 0 && a->H % 8 == 0 && a->H <= 2048, "vlp_layernorm_bwd: H=%d must be a multiple of 8 and <= 2048", a->H);
    VLP_CHECK_ARG(a->lddy % 8 == 0 && a->ldx % 8 == 0 && a->lddx % 8 == 0, "vlp_layernorm_bwd: leading dims");
    VLP_CHECK_ARG(((uintptr_t)a->dy | (uintptr_t)a->x | (uintptr_t)a->dx | (uintptr_t)a->gamma) % 16 == 0, "vlp_layernorm_bwd: alignment");
    if (a->dx_drop) VLP_CHECK_ARG(a->lddxd % 8 == 0 && (uintptr_t)a->dx_drop % 16 == 0 && a->out_drop_p > 0.f, "vlp_layernorm_bwd: dx_drop needs out_drop_p > 0");
    const int64_t need = vlp_layernorm_bwd_workspace_bytes(a->H);
    if (!a->workspace || a->workspace_bytes < need) return vlp_set_error(VLP_ERR_WORKSPACE, "vlp_layernorm_bwd: workspace %lld < %lld", (long long)a->workspace_bytes, (long long)need);
    DropCtx dyd = make_drop(a->dy_drop_p, a->dy_seed, a->dy_stream);
    DropCtx outd = make_drop(a->out_drop_p, a->out_seed, a->out_stream);
    const int blocks = lnb_blocks(a->M);
    hipStream_t s = (hipStream_t)stream;
    float* part = (float*)a->workspace;
    const size_t lnb_smem = (size_t)LNB_WAVES * 2 * a->H * sizeof(float);   // <= 128 KiB at H = 2048
    VLP_ONCE_PER_DEVICE({
        (void)hipFuncSetAttribute((const void*)layernorm_bwd_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, LNB_WAVES * 2 * 768 * 4);
        (void)hipFuncSetAttribute((const void*)layernorm_bwd_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, LNB_WAVES * 2 * 1024 * 4);
        (void)hipFuncSetAttribute((const void*)layernorm_bwd_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, LNB_WAVES * 2 * 2048 * 4);
    });
    // H = 768: the countable-loop form (VLP_LNB_FULL=0: the general kernel, A/B runs)
    static const int full_on = [] { const char* e = getenv("VLP_LNB_FULL"); return (e && e[0] == '0') ? 0 : 1; }();
    if (full_on && a->H == 768) {
        const bool hd = a->dx_drop != nullptr, hm = a->row_map != nullptr && (dyd.thresh || outd.thresh);
#define LAUNCH_LNB_FULL(HD_, HM_)                                                                                                                       \
    do {                                                                                                                                                \
        VLP_ONCE_PER_DEVICE((void)hipFuncSetAttribute((const void*)layernorm_bwd_full_kernel<3, HD_, HM_>, hipFuncAttributeMaxDynamicSharedMemorySize, LNB_WAVES * 2 * 768 * 4)); \
        hipLaunchKernelGGL((layernorm_bwd_full_kernel<3, HD_, HM_>), dim3(blocks), dim3(LNB_THREADS), lnb_smem, s, (const f16*)a->dy, a->lddy, (const f16*)a->x,       \
                           a->ldx, (const f16*)a->gamma, a->mean, a->rstd, (f16*)a->dx, a->lddx, (f16*)a->dx_drop, a->lddxd, part, a->M, dyd, outd, a->row_map);          \
    } while (0)
        if (hd && hm) LAUNCH_LNB_FULL(true, true); else if (hd) LAUNCH_LNB_FULL(true, false); else if (hm) LAUNCH_LNB_FULL(false, true); else LAUNCH_LNB_FULL(false, false);
#undef LAUNCH_LNB_FULL
    } else if (a->H <= 768)
        hipLaunchKernelGGL(layernorm_bwd_kernel<3>, dim3(blocks), dim3(LNB_THREADS), lnb_smem, s, (const f16*)a->dy, a->lddy, (const f16*)a->x, a->ldx,
                           (const f16*)a->gamma, a->mean, a->rstd, (f16*)a->dx, a->lddx, (f16*)a->dx_drop, a->lddxd, part, a->M, a->H, dyd, outd, a->row_map);
    else if (a->H <= 1024)
        hipLaunchKernelGGL(layernorm_bwd_kernel<4>, dim3(blocks), dim3(LNB_THREADS), lnb_smem, s, (const f16*)a->dy, a->lddy, (const f16*)a->x, a->ldx,
                           (const f16*)a->gamma, a->mean, a->rstd, (f16*)a->dx, a->lddx, (f16*)a->dx_drop, a->lddxd, part, a->M, a->H, dyd, outd, a->row_map);
    else
        hipLaunchKernelGGL(layernorm_bwd_kernel<8>, dim3(blocks), dim3(LNB_THREADS), lnb_smem, s, (const f16*)a->dy, a->lddy, (const f16*)a->x, a->ldx,
                           (const f16*)a->gamma, a->mean, a->rstd, (f16*)a->dx, a->lddx, (f16*)a->dx_drop, a->lddxd, part, a->M, a->H, dyd, outd, a->row_map);
    VLP_CHECK_LAUNCH("vlp_layernorm_bwd");
    if (a->defer_reduce) return VLP_OK;
    hipLaunchKernelGGL(ln_bwd_reduce_kernel, dim3(cdiv(2 * a->H, 64)), dim3(1024), 0, s, part, blocks, a->H, (f16*)a->dgamma,
                       (f16*)a->dbeta, a->beta);
    VLP_CHECK_LAUNCH("vlp_layernorm_bwd_reduce");
    return VLP_OK;
}

// blockIdx.y = LayerNorm index: same arithmetic (and summation order) as ln_bwd_reduce_kernel on slot y
__global__ __launch_bounds__(1024) void ln_bwd_reduce_batched_kernel(const float* __restrict__ parts, int64_t slot_stride, f16* const* __restrict__ dst,
                                                                     int nparts, int H, int beta) {
    __shared__ float sh[16][64];
    const float* part = parts + (int64_t)blockIdx.y * slot_stride;
    const int cl = threadIdx.x & 63, pg = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + cl;
    float s = 0.f;
    if (i < 2 * H)
        for (int p = pg; p < nparts; p += 16) s += part[(int64_t)p * 2 * H + i];
    sh[pg][cl] = s;
    __syncthreads();
    if (pg == 0 && i < 2 * H) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += sh[k][cl];
        f16* d = i < H ? dst[2 * blockIdx.y] + i : dst[2 * blockIdx.y + 1] + (i - H);
        *d = (f16)(beta ? (float)*d + t : t);
    }
}

extern "C" int vlp_layernorm_bwd_reduce_batched(const float* parts, const void* const* dst, int32_t count, int32_t M, int32_t H, int32_t beta, void* stream) {
    VLP_CHECK_ARG(parts && dst && count > 0 && count <= 65535 && M > 0 && H > 0 && H % 8 == 0 && H <= 2048, "vlp_layernorm_bwd_reduce_batched: bad args");
    VLP_ENTER(parts, "vlp_layernorm_bwd_reduce_batched");
    VLP_CHECK_ARG(beta == 0 || beta == 1, "vlp_layernorm_bwd_reduce_batched: beta must be 0 or 1");
    const int blocks = lnb_blocks(M);
    hipLaunchKernelGGL(ln_bwd_reduce_batched_kernel, dim3(cdiv(2 * H, 64), count), dim3(1024), 0, (hipStream_t)stream, parts,
                       (int64_t)LNB_BLOCKS * 2 * H, (f16* const*)dst, blocks, H, beta);
    VLP_CHECK_LAUNCH("vlp_layernorm_bwd_reduce_batched");
    return VLP_OK;
}

// ---------------------------------------------------------------------------------------------
// column sums: out[n] (+)= sum_m A[m,n].  grid (col blocks of 256, row splits); partials -> reduce.
// ---------------------------------------------------------------------------------------------
#define CS_SPLITS 64

__global__ __launch_bounds__(256) void colsum_kernel(const f16* __restrict__ A, int64_t lda, int M, int N, float* __restrict__ part) {
    __shared__ float red[8][256];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 column-chunks x 8 row lanes
    const int c0 = blockIdx.x * 256 + tx * 8;
    const int rows_per = (M + gridDim.y - 1) / gridDim.y;
    const int r0 = blockIdx.y * rows_per, r1 = min(M, r0 + rows_per);
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    if (c0 < N) {
        for (int r = r0 + ty; r < r1; r += 8) {
            f16x8 v = ld8(A + (int64_t)r * lda + c0);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += (float)v[e];
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[ty][tx * 8 + e] = acc[e];
    __syncthreads();
    const int c = threadIdx.x;
    float s = 0.f;
#pragma unroll
    for (int y = 0; y < 8; ++y) s += red[y][c];
    const int col = blockIdx.x * 256 + c;
    if (col < N) part[(int64_t)blockIdx.y * N + col] = s;
}
__global__ __launch_bounds__(1024) void colsum_reduce_kernel(const float* __restrict__ part, int nparts, int N, f16* out, int beta) {
    __shared__ float sh[16][64];
    const int cl = threadIdx.x & 63, pg = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + cl;
    float s = 0.f;
    if (i < N)
        for (int p = pg; p < nparts; p += 16) s += part[(int64_t)p * N + i];
    sh[pg][cl] = s;
    __syncthreads();
    if (pg == 0 && i < N) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += sh[k][cl];
        out[i] = (f16)(beta ? (float)out[i] + t : t);
    }
}

extern "C" int64_t vlp_colsum_workspace_bytes(int32_t M, int32_t N) {
    (void)M;
    return (int64_t)CS_SPLITS * N * (int64_t)sizeof(float);
}
extern "C" int vlp_colsum(const vlp_colsum_args* a, void* stream) {
    VLP_CHECK_ARG(a && a->A && a->out && a->M > 0 && a->N > 0, "vlp_colsum: bad args");
    VLP_ENTER(a->A, "vlp_colsum");
    VLP_CHECK_ARG(a->lda % 8 == 0 && a->lda >= (a->N + 7) / 8 * 8 && (uintptr_t)a->A % 16 == 0, "vlp_colsum: layout (lda must cover roundup8(N))");
    int splits = a->M >= CS_SPLITS * 8 ? CS_SPLITS : (a->M + 7) / 8;
    if (splits < 1) splits = 1;
    const int64_t need = (int64_t)splits * a->N * (int64_t)sizeof(float);
    if (!a->workspace || a->workspace_bytes < need) return vlp_set_error(VLP_ERR_WORKSPACE, "vlp_colsum: workspace %lld < %lld", (long long)a->workspace_bytes, (long long)need);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(a->N, 256), splits), dim3(256), 0, s, (const f16*)a->A, a->lda, a->M, a->N, (float*)a->workspace);
    VLP_CHECK_LAUNCH("vlp_colsum");
    hipLaunchKernelGGL(colsum_reduce_kernel, dim3(cdiv(a->N, 64)), dim3(1024), 0, s, (const float*)a->workspace, splits, a->N, (f16*)a->out, a->beta);
    VLP_CHECK_LAUNCH("vlp_colsum_reduce");
    return VLP_OK;
}
