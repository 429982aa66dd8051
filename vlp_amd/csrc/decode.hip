// Token-step kernels of the incremental decoder (replaces the per-token part of BertForSeq2SeqDecoder.forward,
// modeling.py:1189-1253, i.e. BertLayer x 12 on the 1-2 new positions of every sequence: :268-303 history path, :313-317, :340-343, :353-357).
//
// Why these exist (round 6).  A token step runs every Linear of the encoder on M = sequences x 2 rows (128 at B = 64): each GEMM moves
// 1 - 5 MB of weights -- under a microsecond of HBM time -- so the step is bound by launches and dependent round trips, not by bytes or flops.
// Until round 5 a layer was 12 launches (four split-K GEMMs + four slab reduces, kv_append, attention, two LayerNorms) of 6 - 9 us each:
// 1.13 ms per token step against ~0.1 ms of weight + K/V-cache traffic.  Here a layer is 7 launches whose bodies are ONE load burst each:
//
//   vlp_dec_gemm       Y[M, N] = epi(X[M, K] . W[N, K]^T): a workgroup owns 64 rows x 32 columns and its WHOLE k slice (<= 768): the X rows
//                      (96 KB) and W rows (48 KB) are requested by LDS-DMA in one burst at kernel start -- no k loop of dependent round
//                      trips -- and consumed in three groups behind counted vmcnt waits.  grid = (N / 32, M / 64, splits): 144 - 192
//                      workgroups for every Linear of BERT-base.  Epilogues: bias (+ erf-GeLU) -> fp16; the QKV form writes the K | V
//                      columns straight into the layer's K/V cache at the rows' absolute positions (no kv_append launch); splits > 1
//                      (FFN-down) writes fp32 partial tiles to a slab.  `residual`: + residual before the one rounding (BertSelfOutput).  LayerNorm
//                      PROLOGUE (`ln_gamma`): X holds pre-LayerNorm rows; every workgroup normalises its 64 rows in LDS right after the burst lands
//                      (a wave needs only its own vmcnt for the rows it staged) -- the LayerNorm launch between out-projection and FFN-up is gone.
//   vlp_dec_reduce_ln  sums the slabs in a fixed order (deterministic: no atomics), adds bias + residual, rounds to fp16 exactly where the
//                      unfused path rounds (its GEMM epilogue's fp16 output), and applies LayerNorm (TF style, fp32 statistics,
//                      modeling.py:188-192): one wave per row, the slab reduce the split-K needed anyway IS the LayerNorm launch.
//
// MFMA orientation / fragment layout as gemm_nt_ph.hip: v_mfma_f32_16x16x32_f16 computes Y^T tiles (A = W rows, B = X rows); W rows of a
// tile are visited in the order n = 8 (i >> 2) + 4 tn + (i & 3), so a lane ends with ONE row m and 8 consecutive n: one 16-byte store.
// LDS: k tiles of [rows][64 halfs] (128-byte rows), 16-byte-chunk XOR swizzle applied on the DMA source address (X: row & 7; W: bits
// 1, 3, 4 of the row = lane >> 1 of its fragment read).
#include "common.h"

#define DG_BM 64
#define DG_BN 32
#define DG_BK 64
#define DG_MAXKT 12            // k tiles of one workgroup's slice: 12 x (8 KB X + 4 KB W) = 144 KB of LDS
#define DG_GROUP 4             // k tiles consumed per wait

struct DecGemmParams {
    const f16* X; int64_t ldx;
    const f16* W; int64_t ldw;
    const f16* bias;
    f16* Y; int64_t ldy;
    float* slab; int64_t ldslab;
    f16* kv; int64_t kv_ld; int kv_col0, kv_Lcap, kv_T, kv_start;
    const f16* residual; int64_t ldr;                       // EPI 0: + residual[m][n] before the one rounding (BertSelfOutput's dense + residual)
    const f16* ln_gamma; const f16* ln_beta; float ln_eps;  // LNP: X holds PRE-LayerNorm rows; the workgroup normalises its 64 rows in LDS first
    f16* ln_out; int64_t ld_ln_out;                         //      and the workgroups of column tile 0 write the normalised rows here
    int M, N, K, nkt, act;
};

template <int EPI, bool LNP = false>      // EPI 0: bias (+ residual) (+ GeLU) -> fp16 Y / K|V cache    1: fp32 partial tile -> slab[blockIdx.z];  LNP: LayerNorm prologue
__global__ __launch_bounds__(256, 1) void dec_gemm_kernel(DecGemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    f16* smem = reinterpret_cast<f16*>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, li = lane & 15;
    const int n0 = blockIdx.x * DG_BN, m0 = blockIdx.y * DG_BM;
    const int nkt = p.nkt;                                   // k tiles of this workgroup's slice
    const int64_t k0 = (int64_t)blockIdx.z * nkt * DG_BK;
    constexpr int XT = DG_BM * DG_BK, WT = DG_BN * DG_BK;    // halfs per k tile
    f16* const xs = smem;                                    // [nkt][64][64]
    f16* const ws = smem + DG_MAXKT * XT;                    // [nkt][32][64]

    // ---- one burst of LDS-DMA: this wave's 16 X rows (two 8-row pieces) and W rows 8w .. 8w+7 of every k tile ------------------------
    {
        const int rb = lane >> 3, pc = lane & 7;             // row inside an 8-row piece, physical 16-byte chunk
        const int xr0 = 16 * w + rb, xr1 = xr0 + 8;          // tile rows
        const int wr = 8 * w + rb;
        const int wf = ((wr >> 1) & 1) | (((wr >> 3) & 1) << 1) | (((wr >> 4) & 1) << 2);
        const f16* x0 = p.X + (int64_t)min(m0 + xr0, p.M - 1) * p.ldx + k0 + ((pc ^ (xr0 & 7)) << 3);
        const f16* x1 = p.X + (int64_t)min(m0 + xr1, p.M - 1) * p.ldx + k0 + ((pc ^ (xr1 & 7)) << 3);
        const f16* w0 = p.W + (int64_t)min(n0 + wr, p.N - 1) * p.ldw + k0 + ((pc ^ wf) << 3);
        const uint32_t xs_l = lds_addr_of(xs) + (uint32_t)(16 * w) * (DG_BK * 2);
        const uint32_t ws_l = lds_addr_of(ws) + (uint32_t)(8 * w) * (DG_BK * 2);
        for (int kt = 0; kt < nkt; ++kt) {
            glds16(x0 + kt * DG_BK, xs_l + (uint32_t)kt * (XT * 2));
            glds16(x1 + kt * DG_BK, xs_l + (uint32_t)kt * (XT * 2) + 8 * (DG_BK * 2));
            glds16(w0 + kt * DG_BK, ws_l + (uint32_t)kt * (WT * 2));
        }
    }
    f32x4 acc[2];
    acc[0] = (f32x4){0.f, 0.f, 0.f, 0.f};
    acc[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int xrow = 16 * w + li;                            // this lane's X row inside the tile (B operand)
    const int wrow = 8 * (li >> 2) + (li & 3);               // + 4 tn: permuted W row (A operand)
    auto compute = [&](int kt) {
        const f16* xk = xs + kt * XT;
        const f16* wk = ws + kt * WT;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int c = ks * 4 + g;
            const f16x8 xf = ld8(xk + xrow * DG_BK + ((c ^ (xrow & 7)) << 3));
            const f16x8 wf0 = ld8(wk + wrow * DG_BK + ((c ^ (li >> 1)) << 3));
            const f16x8 wf1 = ld8(wk + (wrow + 4) * DG_BK + ((c ^ (li >> 1)) << 3));
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf0, xf, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf1, xf, acc[1], 0, 0, 0);
        }
    };
    if constexpr (LNP) {
        // ---- LayerNorm prologue (BertLayerNorm, modeling.py:188-192; fp32 statistics, one rounding -- the arithmetic of layernorm_fwd / dec_reduce_ln):
        // X holds the fp16 pre-LayerNorm rows (dense + bias + residual of the previous Linear).  A wave normalises the 16 rows IT staged (its own
        // vmcnt covers them: no barrier), in place in LDS: 4 lanes per row (row 16 w + li, lane quarter g takes k tiles g, g + 4, g + 8), so the
        // LayerNorm launch between the two Linears -- and the round trip of the normalised rows through HBM -- disappears.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // lane l <-> (row 8 j + (l >> 3) of the wave's 16, physical 16-byte chunk l & 7): one ds_read_b128 covers 8 whole 128-byte rows of a k tile =
        // 1 KB contiguous in LDS (conflict-free: the DMA's own image); a row's sum is the lane's 12 chunks + a 3-step DPP reduction over its 8 lanes.
        // (First form: 4 lanes per row, each reading 3 k tiles x 8 chunks -- 8-way bank conflicts and 48 parameter loads per lane: +9 us per launch.)
        const int rb = lane >> 3, pc = lane & 7;
        f16x8 raw[2][DG_MAXKT];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int kt = 0; kt < DG_MAXKT; ++kt)
                raw[j][kt] = (kt < nkt) ? ld8(xs + kt * XT + (16 * w + 8 * j + rb) * DG_BK + (pc << 3)) : (f16x8){0, 0, 0, 0, 0, 0, 0, 0};
        const float invK = 1.f / (float)p.K;
        float mu[2], rs[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float sum = 0.f;
#pragma unroll
            for (int kt = 0; kt < DG_MAXKT; ++kt)
#pragma unroll
                for (int e = 0; e < 8; ++e) sum += (float)raw[j][kt][e];
            mu[j] = sum8(sum) * invK;
            float sq = 0.f;
#pragma unroll
            for (int kt = 0; kt < DG_MAXKT; ++kt)
                if (kt < nkt) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) { const float d = (float)raw[j][kt][e] - mu[j]; sq += d * d; }
                }
            rs[j] = 1.f / sqrtf(sum8(sq) * invK + p.ln_eps);
        }
        // the row's LOGICAL chunk at physical slot pc is pc ^ (row & 7) (the DMA swizzle): that is the column block gamma / beta are read at
#pragma unroll
        for (int kt = 0; kt < DG_MAXKT; ++kt) {
            if (kt < nkt) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int r = 16 * w + 8 * j + rb;
                    const int c = (int)k0 + kt * DG_BK + ((pc ^ (r & 7)) << 3);
                    const f16x8 gv = ld8(p.ln_gamma + c), bv = ld8(p.ln_beta + c);
                    f16x8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = (f16)((float)gv[e] * (((float)raw[j][kt][e] - mu[j]) * rs[j]) + (float)bv[e]);
                    st8(xs + kt * XT + r * DG_BK + (pc << 3), o);
                    const int m_ln = m0 + r;
                    if (p.ln_out && blockIdx.x == 0 && m_ln < p.M) st8(p.ln_out + (int64_t)m_ln * p.ld_ln_out + c, o);
                }
            }
        }
        __syncthreads();                                     // every wave's normalised rows + everybody's W pieces (their vmcnt(0) above) are in LDS
        for (int kt = 0; kt < nkt; ++kt) compute(kt);
    }
    // groups of DG_GROUP k tiles: the in-order vmcnt of this wave says its own pieces of the group have landed, the barrier says
    // everybody's have (every wave issued the same number of DMAs per k tile: 3)
    const int ngroups = LNP ? 0 : (nkt + DG_GROUP - 1) / DG_GROUP;
    for (int gi = 0; gi < ngroups; ++gi) {
        const int left = max(nkt - (gi + 1) * DG_GROUP, 0) * 3;       // DMAs of later groups that may stay in flight
        if (left >= 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
        else if (left >= 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const int ke = min(nkt, (gi + 1) * DG_GROUP);
        for (int kt = gi * DG_GROUP; kt < ke; ++kt) compute(kt);
    }

    // ---- epilogue: lane owns row m and the 8 columns nc .. nc+7 -----------------------------------------------------------------------
    const int m = m0 + 16 * w + li;
    const int nc = n0 + 8 * g;
    if (m >= p.M || nc >= p.N) return;
    float v[8];
#pragma unroll
    for (int r = 0; r < 4; ++r) { v[r] = acc[0][r]; v[4 + r] = acc[1][r]; }
    if constexpr (EPI == 1) {
        float* dst = p.slab + ((int64_t)blockIdx.z * p.M + m) * p.ldslab + nc;
        *reinterpret_cast<f32x4*>(dst) = (f32x4){v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4*>(dst + 4) = (f32x4){v[4], v[5], v[6], v[7]};
        return;
    } else {
        const bool full = nc + 8 <= p.N;
        if (p.bias) {
            if (full) {
                const f16x8 b = ld8(p.bias + nc);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] += (float)b[j];
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) if (nc + j < p.N) v[j] += (float)p.bias[nc + j];
            }
        }
        if (p.residual) {
            if (full) {
                const f16x8 rr = ld8(p.residual + (int64_t)m * p.ldr + nc);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] += (float)rr[j];
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) if (nc + j < p.N) v[j] += (float)p.residual[(int64_t)m * p.ldr + nc + j];
            }
        }
        if (p.act == VLP_ACT_GELU) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = gelu_f(v[j]);
        }
        f16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (full || nc + j < p.N) ? (f16)v[j] : (f16)0.f;
        if (p.kv && nc >= p.kv_col0) {
            // K | V columns of the QKV projection go to the cache row of (sequence m / T, absolute position start + m % T): vlp_kv_append fused
            const int seq = m / p.kv_T, t = m - seq * p.kv_T;
            st8(p.kv + ((int64_t)seq * p.kv_Lcap + p.kv_start + t) * p.kv_ld + (nc - p.kv_col0), o);
        } else {
            st8(p.Y + (int64_t)m * p.ldy + nc, o);
        }
    }
}

extern "C" int vlp_dec_gemm(const vlp_dec_gemm_args* a, void* stream) {
    VLP_CHECK_ARG(a != nullptr && a->X && a->W, "vlp_dec_gemm: null operand");
    VLP_ENTER(a->X, "vlp_dec_gemm");
    VLP_CHECK_ARG(a->M > 0 && a->N > 0 && a->K > 0 && a->splits >= 1, "vlp_dec_gemm: bad shape M=%d N=%d K=%d splits=%d", a->M, a->N, a->K, a->splits);
    VLP_CHECK_ARG(a->K % (DG_BK * a->splits) == 0 && a->K / (DG_BK * a->splits) <= DG_MAXKT,
                  "vlp_dec_gemm: K=%d must split into %d slices of at most %d k tiles of %d", a->K, a->splits, DG_MAXKT, DG_BK);
    VLP_CHECK_ARG(a->ldx % 8 == 0 && a->ldw % 8 == 0 && a->ldx >= a->K && a->ldw >= a->K, "vlp_dec_gemm: leading dims");
    VLP_CHECK_ARG(((uintptr_t)a->X | (uintptr_t)a->W) % 16 == 0, "vlp_dec_gemm: operands must be 16-byte aligned");
    VLP_CHECK_ARG(a->act == VLP_ACT_NONE || a->act == VLP_ACT_GELU, "vlp_dec_gemm: act must be NONE or GELU");
    DecGemmParams p;
    p.X = (const f16*)a->X; p.ldx = a->ldx;
    p.W = (const f16*)a->W; p.ldw = a->ldw;
    p.bias = (const f16*)a->bias;
    p.Y = (f16*)a->Y; p.ldy = a->ldy;
    p.slab = (float*)a->slab; p.ldslab = a->ldslab;
    p.kv = (f16*)a->kv_cache; p.kv_ld = a->kv_ld; p.kv_col0 = a->kv_col0; p.kv_Lcap = a->kv_Lcap; p.kv_T = a->kv_T; p.kv_start = a->kv_start;
    p.residual = (const f16*)a->residual; p.ldr = a->ldr;
    p.ln_gamma = (const f16*)a->ln_gamma; p.ln_beta = (const f16*)a->ln_beta; p.ln_eps = a->ln_eps;
    p.ln_out = (f16*)a->ln_out; p.ld_ln_out = a->ld_ln_out;
    p.M = a->M; p.N = a->N; p.K = a->K; p.nkt = a->K / (DG_BK * a->splits); p.act = a->act;
    const dim3 grid(cdiv(a->N, DG_BN), cdiv(a->M, DG_BM), a->splits);
    const size_t smem = (size_t)DG_MAXKT * (DG_BM + DG_BN) * DG_BK * sizeof(f16);
    hipStream_t s = (hipStream_t)stream;
    if (a->splits > 1 || a->slab) {
        VLP_CHECK_ARG(a->slab && a->ldslab % 8 == 0 && a->ldslab >= (a->N + 7) / 8 * 8 && (uintptr_t)a->slab % 16 == 0,
                      "vlp_dec_gemm: the split form needs an fp32 slab [splits][M][ldslab], ldslab %% 8 == 0, 16-byte aligned");
        VLP_CHECK_ARG(a->act == VLP_ACT_NONE && !a->kv_cache && !a->residual && !a->ln_gamma,
                      "vlp_dec_gemm: the split form stores raw partial sums (bias / residual / activation / LayerNorm belong to vlp_dec_reduce_ln)");
        VLP_ONCE_PER_DEVICE((void)hipFuncSetAttribute((const void*)dec_gemm_kernel<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        hipLaunchKernelGGL((dec_gemm_kernel<1, false>), grid, dim3(256), smem, s, p);
    } else {
        VLP_CHECK_ARG(a->Y && a->ldy % 8 == 0 && (uintptr_t)a->Y % 16 == 0, "vlp_dec_gemm: bad Y layout");
        VLP_CHECK_ARG(!a->bias || (uintptr_t)a->bias % 16 == 0, "vlp_dec_gemm: bias must be 16-byte aligned");
        if (a->kv_cache)
            VLP_CHECK_ARG(a->kv_col0 % DG_BN == 0 && a->kv_col0 <= a->N && a->kv_ld % 8 == 0 && a->kv_T > 0 && a->kv_start >= 0 &&
                          a->kv_start + a->kv_T <= a->kv_Lcap && (uintptr_t)a->kv_cache % 16 == 0 && a->N % 8 == 0,
                          "vlp_dec_gemm: bad K/V cache arguments");
        if (a->residual) VLP_CHECK_ARG(a->ldr % 8 == 0 && (uintptr_t)a->residual % 16 == 0, "vlp_dec_gemm: bad residual layout");
        if (a->ln_gamma) {
            VLP_CHECK_ARG(a->ln_beta && ((uintptr_t)a->ln_gamma | (uintptr_t)a->ln_beta) % 16 == 0 && (!a->ln_out || (a->ld_ln_out % 8 == 0 && (uintptr_t)a->ln_out % 16 == 0)),
                          "vlp_dec_gemm: bad LayerNorm-prologue arguments");
            VLP_ONCE_PER_DEVICE((void)hipFuncSetAttribute((const void*)dec_gemm_kernel<0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
            hipLaunchKernelGGL((dec_gemm_kernel<0, true>), grid, dim3(256), smem, s, p);
        } else {
            VLP_ONCE_PER_DEVICE((void)hipFuncSetAttribute((const void*)dec_gemm_kernel<0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
            hipLaunchKernelGGL((dec_gemm_kernel<0, false>), grid, dim3(256), smem, s, p);
        }
    }
    VLP_CHECK_LAUNCH("vlp_dec_gemm");
    return VLP_OK;
}

// ---- slab reduce + bias + residual + LayerNorm: one wave per row, H = 64 * 4 * NV ---------------------------------------------------------
struct DecReduceLnParams {
    const float* slab; int64_t ldslab; int splits;
    const f16* bias; const f16* residual; int64_t ldr;
    const f16* gamma; const f16* beta; float eps;
    f16* Y; int64_t ldy;
    int M, H;
};

template <int NV, int S>       // NV: 4-column pieces per lane (H = 256 NV); S: slabs (compile time: every load of the row is requested before the first is used)
__global__ __launch_bounds__(256) void dec_reduce_ln_kernel(DecReduceLnParams p) {
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= p.M) return;
    f32x4 part[S][NV];
    f16x4 bv[NV], rv[NV], ga[NV], be[NV];
    const f16x4 z4 = (f16x4){(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f};
#pragma unroll
    for (int sp = 0; sp < S; ++sp)
#pragma unroll
        for (int i = 0; i < NV; ++i) part[sp][i] = *reinterpret_cast<const f32x4*>(p.slab + ((int64_t)sp * p.M + m) * p.ldslab + (i * 64 + lane) * 4);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        bv[i] = p.bias ? ld4(p.bias + c) : z4;
        rv[i] = p.residual ? ld4(p.residual + (int64_t)m * p.ldr + c) : z4;
        ga[i] = ld4(p.gamma + c);
        be[i] = ld4(p.beta + c);
    }
    float v[NV][4];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        f32x4 s = part[0][i];
#pragma unroll
        for (int sp = 1; sp < S; ++sp)               // fixed order: deterministic
#pragma unroll
            for (int j = 0; j < 4; ++j) s[j] += part[sp][i][j];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v[i][j] = (float)(f16)(s[j] + (float)bv[i][j] + (float)rv[i][j]);      // the unfused path's fp16 pre-LayerNorm tensor
            sum += v[i][j];
        }
    }
    const float mean = wave_sum(sum) / (float)p.H;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float d = v[i][j] - mean; sq += d * d; }
    const float rstd = rsqrtf(wave_sum(sq) / (float)p.H + p.eps);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        f16x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (f16)((float)ga[i][j] * ((v[i][j] - mean) * rstd) + (float)be[i][j]);
        st4(p.Y + (int64_t)m * p.ldy + (i * 64 + lane) * 4, o);
    }
}

extern "C" int vlp_dec_reduce_ln(const vlp_dec_reduce_ln_args* a, void* stream) {
    VLP_CHECK_ARG(a != nullptr && a->slab && a->gamma && a->beta && a->Y, "vlp_dec_reduce_ln: null operand");
    VLP_ENTER(a->slab, "vlp_dec_reduce_ln");
    VLP_CHECK_ARG(a->M > 0 && a->H > 0 && a->H % 256 == 0 && a->H <= 1024, "vlp_dec_reduce_ln: H must be 256, 512, 768 or 1024 (got %d)", a->H);
    VLP_CHECK_ARG(a->splits == 1 || a->splits == 2 || a->splits == 4 || a->splits == 8, "vlp_dec_reduce_ln: splits must be 1, 2, 4 or 8 (got %d)", a->splits);
    VLP_CHECK_ARG(a->ldslab % 4 == 0 && a->ldslab >= a->H && a->ldy % 4 == 0 && (!a->residual || a->ldr % 4 == 0), "vlp_dec_reduce_ln: leading dims");
    VLP_CHECK_ARG(((uintptr_t)a->slab % 16) == 0 && ((uintptr_t)a->Y | (uintptr_t)a->gamma | (uintptr_t)a->beta | (uintptr_t)a->bias | (uintptr_t)a->residual) % 8 == 0,
                  "vlp_dec_reduce_ln: alignment");
    DecReduceLnParams p;
    p.slab = (const float*)a->slab; p.ldslab = a->ldslab; p.splits = a->splits;
    p.bias = (const f16*)a->bias; p.residual = (const f16*)a->residual; p.ldr = a->ldr;
    p.gamma = (const f16*)a->gamma; p.beta = (const f16*)a->beta; p.eps = a->eps;
    p.Y = (f16*)a->Y; p.ldy = a->ldy; p.M = a->M; p.H = a->H;
    const dim3 grid(cdiv(a->M, 4));
    hipStream_t s = (hipStream_t)stream;
#define LAUNCH_RL(NV_, S_) hipLaunchKernelGGL((dec_reduce_ln_kernel<NV_, S_>), grid, dim3(256), 0, s, p)
#define LAUNCH_RL_S(NV_) do { if (a->splits == 1) LAUNCH_RL(NV_, 1); else if (a->splits == 2) LAUNCH_RL(NV_, 2); else if (a->splits == 4) LAUNCH_RL(NV_, 4); else LAUNCH_RL(NV_, 8); } while (0)
    switch (a->H / 256) {
        case 1: LAUNCH_RL_S(1); break;
        case 2: LAUNCH_RL_S(2); break;
        case 3: LAUNCH_RL_S(3); break;
        default: LAUNCH_RL_S(4); break;
    }
#undef LAUNCH_RL_S
#undef LAUNCH_RL
    VLP_CHECK_LAUNCH("vlp_dec_reduce_ln");
    return VLP_OK;
}

// ---- greedy token choice (modeling.py:1228): first maximum of every logits row, written to the output column AND to the next step's input --
__global__ __launch_bounds__(1024) void argmax_rows2_kernel(const f16* logits, int64_t ld, int V, int64_t* ids_a, int64_t sa, int64_t* ids_b, int64_t sb,
                                                            float* vals, int64_t sv_) {
    __shared__ float sv[16];
    __shared__ int si[16];
    const f16* x = logits + (int64_t)blockIdx.x * ld;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    const int nv = V >> 3;
    for (int i = threadIdx.x; i < nv; i += 1024) {            // ascending v within a thread: `>` keeps the first maximum
        const f16x8 q = ld8(x + i * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float f = (float)q[j];
            if (f > best) { best = f; bi = i * 8 + j; }
        }
    }
    for (int v = nv * 8 + threadIdx.x; v < V; v += 1024) {
        const float f = (float)x[v];
        if (f > best) { best = f; bi = v; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float f = __shfl_xor(best, o, 64);
        const int j = __shfl_xor(bi, o, 64);
        if (f > best || (f == best && j < bi)) { best = f; bi = j; }
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sv[w] = best; si[w] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 16; ++k)
            if (sv[k] > best || (sv[k] == best && si[k] < bi)) { best = sv[k]; bi = si[k]; }
        ids_a[blockIdx.x * sa] = bi;
        if (ids_b) ids_b[blockIdx.x * sb] = bi;
        vals[blockIdx.x * sv_] = best;
    }
}
extern "C" int vlp_argmax_rows2(const void* logits, int64_t ld, int32_t rows, int32_t V, int64_t* ids_a, int64_t ids_a_stride, int64_t* ids_b, int64_t ids_b_stride,
                                float* vals, int64_t vals_stride, void* stream) {
    VLP_CHECK_ARG(logits && ids_a && vals && rows > 0 && V > 0 && ld >= V, "vlp_argmax_rows2: bad args");
    VLP_ENTER(logits, "vlp_argmax_rows2");
    VLP_CHECK_ARG(ld % 8 == 0 && (uintptr_t)logits % 16 == 0, "vlp_argmax_rows2: rows must be 16-byte aligned (ld %% 8 == 0)");
    hipLaunchKernelGGL(argmax_rows2_kernel, dim3(rows), dim3(1024), 0, (hipStream_t)stream, (const f16*)logits, ld, V, ids_a, ids_a_stride, ids_b, ids_b_stride,
                       vals, vals_stride);
    VLP_CHECK_LAUNCH("vlp_argmax_rows2");
    return VLP_OK;
}
