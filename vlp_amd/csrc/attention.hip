// Fused masked-softmax self-attention for gfx950 (forward + backward).
//
// Replaces modeling.py:279-302 of the reference (BertSelfAttention.forward after the Q/K/V Linears):
// transpose_for_scores (:262-266), QK^T / sqrt(d) (:284-287), + extended mask (:289, built :807-833),
// Softmax(-1) (:292), dropout on the probabilities (:296), P.V (:298), permute+contiguous (:299-302),
// and the autograd backward of all of it.  The [B,heads,L,L] score/probability tensors the reference
// materialises four times per layer never leave registers here.
//
// L <= 256 (the reference runs L = 123 or 167), head_dim = 64: one workgroup owns one (batch, head),
// the whole K (and V / K^T / Q^T / dO^T as needed) of that head sits in LDS, each wave walks 16-row
// tiles.  All contractions run on v_mfma_f32_16x16x32_f16.  The kernels compute TRANSPOSED products
// (S^T = K.Q^T, O^T = V^T.P^T, dQ^T = K^T.dS^T, ...) so that, in the MFMA C/D layout
// (col = lane&15, row = 4*(lane>>4)+reg), a lane owns one query (or key) column: softmax row
// statistics are lane-local, a C/D tile can be re-used *in registers* as the B operand of the next MFMA
// (P^T / dS^T tiles feed PV / dQ / dK / dV directly, no LDS round trip), and outputs are 4 consecutive
// head-dim values per lane (8-byte stores).  The k-slot order of such a re-used tile pair (t, t+1) is
// keys {16t+4g+e, 16(t+1)+4g+e}; the other operand is gathered in the same slot order from an LDS image
// stored [head_dim][key] (8-byte ds_read_b64 x2), which is legal because a contraction is invariant
// under a consistent permutation of its index.
#include "common.h"

#define HD 64            // head dim
#define ATT_THREADS 256
#define ATT_WAVES 4

DEVFN int swzk(int r) { return r & 7; }

struct AttnParams {
    const f16* qkv; int64_t ld_qkv;
    const uint8_t* mask;
    f16* ctx; int64_t ld_ctx;              // fwd out / bwd in
    const f16* dctx; int64_t ld_dctx;
    float* lse;
    f16* dqkv; int64_t ld_dqkv;
    float* delta;
    int B, L, Lp, heads, H;
    float scale;
    DropCtx drop;
};

// ---- LDS staging helpers ---------------------------------------------------------------------
// row-major, 128-B rows, chunk-swizzled: dst[key][64]; rows >= L are zero.
DEVFN void stage_rowmajor(f16* dst, const f16* src, int64_t ld, int L, int Lp, int tid) {
    for (int idx = tid; idx < Lp * 8; idx += ATT_THREADS) {
        const int r = idx >> 3, c = idx & 7;
        u32x4 v = (u32x4){0, 0, 0, 0};
        if (r < L) v = *reinterpret_cast<const u32x4*>(src + (int64_t)r * ld + c * 8);
        *reinterpret_cast<u32x4*>(dst + r * HD + ((c ^ swzk(r)) << 3)) = v;
    }
}
// transposed: dst[dd][pitch] = src[key][dd]; keys >= L are zero.  lane <-> key so the 16-bit LDS writes
// of one instruction fall on consecutive addresses.
DEVFN void stage_transposed(f16* dst, int pitch, const f16* src, int64_t ld, int L, int Lp, int tid) {
    const int nkb = Lp / 64 + ((Lp % 64) ? 1 : 0);
    for (int idx = tid; idx < nkb * 64 * 8; idx += ATT_THREADS) {
        const int key = (idx & 63) + 64 * (idx / 512);
        const int c = (idx >> 6) & 7;
        if (key >= Lp) continue;
        f16x8 v = (f16x8){0, 0, 0, 0, 0, 0, 0, 0};
        if (key < L) v = ld8(src + (int64_t)key * ld + c * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) dst[(c * 8 + e) * pitch + key] = v[e];
    }
}

// additive mask term for 4 consecutive keys of one query row.  Mask bytes (vlp_mask_pack): 1 = attend (+0),
// 0 = masked (-10000, modeling.py:832), 2 = padding column past L (excluded: -inf).  Rows are Lp bytes.
DEVFN void mask4(const uint8_t* mrow, int key0, int Lp, float out[4]) {
    const uint32_t w = (key0 < Lp) ? *reinterpret_cast<const uint32_t*>(mrow + key0) : 0x02020202u;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const uint32_t v = (w >> (8 * e)) & 0xffu;
        out[e] = v == 1u ? 0.f : (v == 0u ? -10000.f : -INFINITY);
    }
}

// =================================================================================================
// forward
// =================================================================================================
template <int NT>   // NT = LP / 16 key tiles (4, 8, 12 or 16)
__global__ __launch_bounds__(ATT_THREADS, 3) void attn_fwd_kernel(AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int LP = NT * 16;
    constexpr int VP = LP + 8;                 // V^T row pitch (halfs); (2*VP/16) is odd -> conflict-free b64 reads
    f16* Ks = reinterpret_cast<f16*>(smem_raw);            // [LP][64] swizzled
    f16* Vt = Ks + LP * HD;                                // [64][VP]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, li = lane & 15;
    const int b = blockIdx.x / p.heads, h = blockIdx.x % p.heads;
    const int L = p.L;
    const f16* qbase = p.qkv + (int64_t)b * L * p.ld_qkv + h * HD;
    const f16* kbase = qbase + p.H;
    const f16* vbase = qbase + 2 * p.H;

    stage_rowmajor(Ks, kbase, p.ld_qkv, L, LP, tid);
    stage_transposed(Vt, VP, vbase, p.ld_qkv, L, LP, tid);
    __syncthreads();

    const int nqt = (L + 15) / 16;
    for (int qt = wid; qt < nqt; qt += ATT_WAVES) {
        const int q = qt * 16 + li;             // this lane's query (column of every transposed tile)
        const int qc = min(q, L - 1);
        int gq = g;                             // opaque copy: keeps per-key index math inside the loop (no LICM + spills)
        asm volatile("" : "+v"(gq));
        const f16* qrow = qbase + (int64_t)qc * p.ld_qkv;
        f16x8 qf[2];
        qf[0] = ld8(qrow + g * 8);
        qf[1] = ld8(qrow + 32 + g * 8);

        // S^T tiles: rows = keys 16t + 4g + reg, col = query
        f32x4 s[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            s[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const int kr = t * 16 + li;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                f16x8 kf = ld8(Ks + kr * HD + (((ks * 4 + g) ^ swzk(kr)) << 3));
                s[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[ks], s[t], 0, 0, 0);
            }
        }
        const uint8_t* mrow = p.mask + ((int64_t)b * L + qc) * p.Lp;
        float mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            float ma[4];
            mask4(mrow, t * 16 + 4 * gq, p.Lp, ma);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                s[t][r] = s[t][r] * p.scale + ma[r];
                mx = fmaxf(mx, s[t][r]);
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                s[t][r] = __expf(s[t][r] - mx);
                sum += s[t][r];
            }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.f / sum;
        if (g == 0 && q < L) p.lse[((int64_t)b * p.heads + h) * L + q] = mx + __logf(sum);

        // P^T (normalised, dropout applied) as fp16 B-operand fragments: pair u = tiles (2u, 2u+1)
        f16x8 pf[NT / 2];
        // dropout element = (row (b, h, q), col key)
        const uint32_t rk = p.drop.thresh ? drop_rowkey(p.drop, ((uint64_t)b * p.heads + h) * (uint64_t)L + (uint64_t)qc) : 0u;
#pragma unroll
        for (int u = 0; u < NT / 2; ++u)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int t = 2 * u + (e >> 2), r = e & 3;
                float pv = s[t][r] * inv;
                if (p.drop.thresh) pv *= drop_mult(p.drop, rk, (uint32_t)(t * 16 + 4 * gq + r));
                pf[u][e] = (f16)pv;
            }

        // O^T tiles: rows = head-dim 16n + 4g + reg, col = query
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            f32x4 o = (f32x4){0.f, 0.f, 0.f, 0.f};
            const f16* vrow = Vt + (n * 16 + li) * VP + 4 * g;
#pragma unroll
            for (int u = 0; u < NT / 2; ++u) {
                f16x4 v0 = ld4(vrow + 32 * u), v1 = ld4(vrow + 32 * u + 16);
                f16x8 vf = (f16x8){v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                o = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf[u], o, 0, 0, 0);
            }
            if (q < L) {
                f16x4 ov = (f16x4){(f16)o[0], (f16)o[1], (f16)o[2], (f16)o[3]};
                st4(p.ctx + ((int64_t)b * L + q) * p.ld_ctx + h * HD + n * 16 + 4 * g, ov);
            }
        }
    }
}

// =================================================================================================
// backward, part 1: dQ (and delta = rowsum(dO * O)).  Transposed orientation, query column per lane.
// =================================================================================================
template <int NT>
__global__ __launch_bounds__(ATT_THREADS, 2) void attn_bwd_dq_kernel(AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int LP = NT * 16;
    constexpr int TP = LP + 8;
    f16* Ks = reinterpret_cast<f16*>(smem_raw);     // [LP][64] swizzled
    f16* Vs = Ks + LP * HD;                         // [LP][64] swizzled
    f16* Kt = Vs + LP * HD;                         // [64][TP]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, li = lane & 15;
    const int b = blockIdx.x / p.heads, h = blockIdx.x % p.heads;
    const int L = p.L;
    const f16* qbase = p.qkv + (int64_t)b * L * p.ld_qkv + h * HD;
    const f16* kbase = qbase + p.H;
    const f16* vbase = qbase + 2 * p.H;

    stage_rowmajor(Ks, kbase, p.ld_qkv, L, LP, tid);
    stage_rowmajor(Vs, vbase, p.ld_qkv, L, LP, tid);
    stage_transposed(Kt, TP, kbase, p.ld_qkv, L, LP, tid);
    __syncthreads();

    const int nqt = (L + 15) / 16;
    for (int qt = wid; qt < nqt; qt += ATT_WAVES) {
        const int q = qt * 16 + li;
        const int qc = min(q, L - 1);
        int gq = g;
        asm volatile("" : "+v"(gq));
        const f16* qrow = qbase + (int64_t)qc * p.ld_qkv;
        const f16* dorow = p.dctx + ((int64_t)b * L + qc) * p.ld_dctx + h * HD;
        const f16* orow = p.ctx + ((int64_t)b * L + qc) * p.ld_ctx + h * HD;
        f16x8 qf[2], dof[2];
        float dl = 0.f;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            qf[ks] = ld8(qrow + ks * 32 + g * 8);
            dof[ks] = ld8(dorow + ks * 32 + g * 8);
            f16x8 of = ld8(orow + ks * 32 + g * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) dl += (float)dof[ks][e] * (float)of[e];
        }
        dl += __shfl_xor(dl, 16, 64);
        dl += __shfl_xor(dl, 32, 64);
        const int64_t stat = ((int64_t)b * p.heads + h) * L + qc;
        const float lse = p.lse[stat];
        if (g == 0 && q < L) p.delta[stat] = dl;

        const uint8_t* mrow = p.mask + ((int64_t)b * L + qc) * p.Lp;
        const uint32_t rk = p.drop.thresh ? drop_rowkey(p.drop, ((uint64_t)b * p.heads + h) * (uint64_t)L + (uint64_t)qc) : 0u;
        f16x8 dsf[NT / 2];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f}, dp = (f32x4){0.f, 0.f, 0.f, 0.f};
            const int kr = t * 16 + li;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int off = kr * HD + (((ks * 4 + g) ^ swzk(kr)) << 3);
                s = __builtin_amdgcn_mfma_f32_16x16x32_f16(ld8(Ks + off), qf[ks], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_16x16x32_f16(ld8(Vs + off), dof[ks], dp, 0, 0, 0);
            }
            float ma[4];
            mask4(mrow, t * 16 + 4 * gq, p.Lp, ma);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pr = __expf(s[r] * p.scale + ma[r] - lse);     // 0 for keys >= L (-inf)
                float dpr = dp[r];
                if (p.drop.thresh) dpr *= drop_mult(p.drop, rk, (uint32_t)(t * 16 + 4 * gq + r));
                const float ds = pr * (dpr - dl) * p.scale;
                dsf[t >> 1][(t & 1) * 4 + r] = (f16)ds;
            }
        }
        // dQ^T tiles: rows = head-dim, col = query;  dQ^T = K^T . dS^T
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            f32x4 o = (f32x4){0.f, 0.f, 0.f, 0.f};
            const f16* krow = Kt + (n * 16 + li) * TP + 4 * g;
#pragma unroll
            for (int u = 0; u < NT / 2; ++u) {
                f16x4 k0 = ld4(krow + 32 * u), k1 = ld4(krow + 32 * u + 16);
                f16x8 kf = (f16x8){k0[0], k0[1], k0[2], k0[3], k1[0], k1[1], k1[2], k1[3]};
                o = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, dsf[u], o, 0, 0, 0);
            }
            if (q < L) {
                f16x4 ov = (f16x4){(f16)o[0], (f16)o[1], (f16)o[2], (f16)o[3]};
                st4(p.dqkv + ((int64_t)b * L + q) * p.ld_dqkv + h * HD + n * 16 + 4 * g, ov);
            }
        }
    }
}

// =================================================================================================
// backward, part 2: dK, dV.  Each wave owns a 16-key tile (column per lane) and walks all queries.
// =================================================================================================
template <int NT>
__global__ __launch_bounds__(ATT_THREADS, 3) void attn_bwd_dkv_kernel(AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int LP = NT * 16;
    constexpr int TP = LP + 8;
    f16* Qt = reinterpret_cast<f16*>(smem_raw);     // [64][TP]  Q^T
    f16* dOt = Qt + HD * TP;                        // [64][TP]  dO^T
    float* lse_s = reinterpret_cast<float*>(dOt + HD * TP);   // [LP]
    float* dl_s = lse_s + LP;                                  // [LP]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, li = lane & 15;
    const int b = blockIdx.x / p.heads, h = blockIdx.x % p.heads;
    const int L = p.L;
    const f16* qbase = p.qkv + (int64_t)b * L * p.ld_qkv + h * HD;
    const f16* kbase = qbase + p.H;
    const f16* vbase = qbase + 2 * p.H;
    const f16* dobase = p.dctx + (int64_t)b * L * p.ld_dctx + h * HD;

    stage_transposed(Qt, TP, qbase, p.ld_qkv, L, LP, tid);
    stage_transposed(dOt, TP, dobase, p.ld_dctx, L, LP, tid);
    for (int i = tid; i < LP; i += ATT_THREADS) {
        const int64_t stat = ((int64_t)b * p.heads + h) * L + min(i, L - 1);
        lse_s[i] = p.lse[stat];
        dl_s[i] = p.delta[stat];
    }
    __syncthreads();

    const int nkt = (L + 15) / 16;
    for (int kt = wid; kt < nkt; kt += ATT_WAVES) {
        const int key = kt * 16 + li;            // this lane's key (column)
        const int kc = min(key, L - 1);
        f16x8 kf[2], vf[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            kf[ks] = ld8(kbase + (int64_t)kc * p.ld_qkv + ks * 32 + g * 8);
            vf[ks] = ld8(vbase + (int64_t)kc * p.ld_qkv + ks * 32 + g * 8);
        }
        f32x4 dk[4], dv[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) { dk[n] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[n] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

#pragma unroll 1
        for (int u = 0; u < NT / 2; ++u) {       // query tile pair (2u, 2u+1)
            f16x8 pdf, dsf;                      // B operands: rows = queries (pair slots), col = key
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int qt = 2 * u + half;
                // S tile: rows = queries 16qt + 4g + reg, col = key.  A = Q rows, B = K rows (NT form)
                const int qa = min(qt * 16 + li, L - 1);      // A-operand row of this lane
                const f16* qrow = qbase + (int64_t)qa * p.ld_qkv;
                const f16* dorow = dobase + (int64_t)qa * p.ld_dctx;
                f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f}, dp = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    s = __builtin_amdgcn_mfma_f32_16x16x32_f16(ld8(qrow + ks * 32 + g * 8), kf[ks], s, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_16x16x32_f16(ld8(dorow + ks * 32 + g * 8), vf[ks], dp, 0, 0, 0);
                }
                const int q0 = qt * 16 + 4 * g;
                const f32x4 lse4 = *reinterpret_cast<const f32x4*>(lse_s + q0);
                const f32x4 dl4 = *reinterpret_cast<const f32x4*>(dl_s + q0);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int q = q0 + r;
                    float pr = 0.f, dsv = 0.f, pd = 0.f;
                    if (q < L && key < L) {
                        const bool on = p.mask[((int64_t)b * L + q) * p.Lp + key] == 1;
                        pr = __expf(s[r] * p.scale + (on ? 0.f : -10000.f) - lse4[r]);
                        float mult = 1.f;
                        if (p.drop.thresh)
                            mult = drop_mult(p.drop, drop_rowkey(p.drop, ((uint64_t)b * p.heads + h) * (uint64_t)L + (uint64_t)q), (uint32_t)key);
                        pd = pr * mult;
                        dsv = pr * (dp[r] * mult - dl4[r]) * p.scale;
                    }
                    pdf[half * 4 + r] = (f16)pd;
                    dsf[half * 4 + r] = (f16)dsv;
                }
            }
            // dV^T += dO^T . Pd ; dK^T += Q^T . dS   (rows = head-dim, col = key)
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const f16* dorow = dOt + (n * 16 + li) * TP + 4 * g + 32 * u;
                const f16* qrow = Qt + (n * 16 + li) * TP + 4 * g + 32 * u;
                f16x4 a0 = ld4(dorow), a1 = ld4(dorow + 16);
                f16x4 c0 = ld4(qrow), c1 = ld4(qrow + 16);
                f16x8 dof = (f16x8){a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
                f16x8 qf = (f16x8){c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
                dv[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(dof, pdf, dv[n], 0, 0, 0);
                dk[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(qf, dsf, dk[n], 0, 0, 0);
            }
        }
        if (key < L) {
            f16* drow = p.dqkv + ((int64_t)b * L + key) * p.ld_dqkv + h * HD;
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                f16x4 kv = (f16x4){(f16)dk[n][0], (f16)dk[n][1], (f16)dk[n][2], (f16)dk[n][3]};
                f16x4 vv = (f16x4){(f16)dv[n][0], (f16)dv[n][1], (f16)dv[n][2], (f16)dv[n][3]};
                st4(drow + p.H + n * 16 + 4 * g, kv);
                st4(drow + 2 * p.H + n * 16 + 4 * g, vv);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
static int attn_common_check(const char* who, const void* qkv, int64_t ld_qkv, const uint8_t* mask, int B, int L, int heads) {
    VLP_CHECK_ARG(qkv && mask, "%s: null operand", who);
    VLP_CHECK_ARG(B > 0 && L > 0 && heads > 0, "%s: bad shape", who);
    VLP_CHECK_ARG(L <= 256, "%s: L=%d > 256 is not supported by the single-pass kernel", who, L);
    VLP_CHECK_ARG(ld_qkv % 8 == 0 && ld_qkv >= 3 * heads * HD, "%s: ld_qkv", who);
    VLP_CHECK_ARG((uintptr_t)qkv % 16 == 0 && (uintptr_t)mask % 4 == 0, "%s: alignment", who);
    return VLP_OK;
}

static inline int lp_of(int L) { return L <= 64 ? 64 : (L <= 128 ? 128 : (L <= 192 ? 192 : 256)); }

extern "C" int vlp_attn_fwd(const vlp_attn_fwd_args* a, void* stream) {
    VLP_CHECK_ARG(a != nullptr, "vlp_attn_fwd: null args");
    int rc = attn_common_check("vlp_attn_fwd", a->qkv, a->ld_qkv, a->mask, a->B, a->L, a->heads);
    if (rc) return rc;
    VLP_CHECK_ARG(a->ctx && a->lse && a->ld_ctx % 4 == 0 && (uintptr_t)a->ctx % 8 == 0, "vlp_attn_fwd: ctx/lse");
    AttnParams p = {};
    p.qkv = (const f16*)a->qkv; p.ld_qkv = a->ld_qkv; p.mask = a->mask;
    p.ctx = (f16*)a->ctx; p.ld_ctx = a->ld_ctx; p.lse = a->lse;
    p.B = a->B; p.L = a->L; p.heads = a->heads; p.H = a->heads * HD;
    p.Lp = (a->L + 31) / 32 * 32;
    p.scale = a->scale;
    p.drop = make_drop(a->dropout_p, a->seed, a->rng_stream);
    const int LP = lp_of(a->L);
    const size_t smem = (size_t)LP * HD * 2 + (size_t)HD * (LP + 8) * 2;
    dim3 grid(a->B * a->heads), block(ATT_THREADS);
    hipStream_t s = (hipStream_t)stream;
#define LAUNCH_FWD(NT_)                                                                                              \
    do {                                                                                                             \
        static bool attr = false;                                                                                    \
        if (!attr) { hipFuncSetAttribute((const void*)attn_fwd_kernel<NT_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); attr = true; } \
        hipLaunchKernelGGL(attn_fwd_kernel<NT_>, grid, block, smem, s, p);                                           \
    } while (0)
    if (LP == 64) LAUNCH_FWD(4); else if (LP == 128) LAUNCH_FWD(8); else if (LP == 192) LAUNCH_FWD(12); else LAUNCH_FWD(16);
#undef LAUNCH_FWD
    VLP_CHECK_LAUNCH("vlp_attn_fwd");
    return VLP_OK;
}

extern "C" int vlp_attn_bwd(const vlp_attn_bwd_args* a, void* stream) {
    VLP_CHECK_ARG(a != nullptr, "vlp_attn_bwd: null args");
    int rc = attn_common_check("vlp_attn_bwd", a->qkv, a->ld_qkv, a->mask, a->B, a->L, a->heads);
    if (rc) return rc;
    VLP_CHECK_ARG(a->ctx && a->dctx && a->lse && a->dqkv && a->delta, "vlp_attn_bwd: null operand");
    VLP_CHECK_ARG(a->ld_ctx % 8 == 0 && a->ld_dctx % 8 == 0 && a->ld_dqkv % 4 == 0, "vlp_attn_bwd: leading dims");
    VLP_CHECK_ARG(((uintptr_t)a->ctx | (uintptr_t)a->dctx) % 16 == 0 && (uintptr_t)a->dqkv % 8 == 0, "vlp_attn_bwd: alignment");
    AttnParams p = {};
    p.qkv = (const f16*)a->qkv; p.ld_qkv = a->ld_qkv; p.mask = a->mask;
    p.ctx = (f16*)a->ctx; p.ld_ctx = a->ld_ctx;
    p.dctx = (const f16*)a->dctx; p.ld_dctx = a->ld_dctx;
    p.lse = (float*)a->lse; p.dqkv = (f16*)a->dqkv; p.ld_dqkv = a->ld_dqkv; p.delta = a->delta;
    p.B = a->B; p.L = a->L; p.heads = a->heads; p.H = a->heads * HD;
    p.Lp = (a->L + 31) / 32 * 32;
    p.scale = a->scale;
    p.drop = make_drop(a->dropout_p, a->seed, a->rng_stream);
    const int LP = lp_of(a->L);
    const size_t smem_dq = (size_t)2 * LP * HD * 2 + (size_t)HD * (LP + 8) * 2;
    const size_t smem_dkv = (size_t)2 * HD * (LP + 8) * 2 + (size_t)2 * LP * 4;
    dim3 grid(a->B * a->heads), block(ATT_THREADS);
    hipStream_t s = (hipStream_t)stream;
#define LAUNCH_BWD(NT_)                                                                                              \
    do {                                                                                                             \
        static bool attr = false;                                                                                    \
        if (!attr) {                                                                                                 \
            hipFuncSetAttribute((const void*)attn_bwd_dq_kernel<NT_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_dq);   \
            hipFuncSetAttribute((const void*)attn_bwd_dkv_kernel<NT_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_dkv); \
            attr = true;                                                                                             \
        }                                                                                                            \
        hipLaunchKernelGGL(attn_bwd_dq_kernel<NT_>, grid, block, smem_dq, s, p);                                     \
        hipLaunchKernelGGL(attn_bwd_dkv_kernel<NT_>, grid, block, smem_dkv, s, p);                                   \
    } while (0)
    if (LP == 64) LAUNCH_BWD(4); else if (LP == 128) LAUNCH_BWD(8); else if (LP == 192) LAUNCH_BWD(12); else LAUNCH_BWD(16);
#undef LAUNCH_BWD
    VLP_CHECK_LAUNCH("vlp_attn_bwd");
    return VLP_OK;
}

// int64 [B,L,L] -> uint8 [B,L,Lp]  (1 attend, 0 masked, 2 = padding column)
__global__ void mask_pack_kernel(const int64_t* mask, uint8_t* out, int L, int Lp, int64_t rows) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < rows * Lp; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / Lp;
        const int c = (int)(i % Lp);
        out[i] = c < L ? (mask[r * L + c] != 0 ? 1 : 0) : 2;
    }
}
extern "C" int vlp_mask_pack(const int64_t* mask, uint8_t* out, int32_t B, int32_t L, int32_t Lp, void* stream) {
    VLP_CHECK_ARG(mask && out && B > 0 && L > 0, "vlp_mask_pack: bad args");
    VLP_CHECK_ARG(Lp == (L + 31) / 32 * 32, "vlp_mask_pack: Lp must be roundup32(L)");
    const int64_t rows = (int64_t)B * L;
    int blocks = (int)((rows * Lp + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(mask_pack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, mask, out, L, Lp, rows);
    VLP_CHECK_LAUNCH("vlp_mask_pack");
    return VLP_OK;
}
