// TN GEMM (weight gradients) for gfx950:   dW[N,K] (+)= dY[M,N]^T . X[M,K]
//
// Replaces autograd's LinearBackward weight-gradient GEMMs of every nn.Linear on the reference's hot
// path (modeling.py:270-272, 314, 341, 354, 432, 481, 1003-1005, 1016, 1027-1029).
//
// Both operands are stored with the contraction index (token row m) as the SLOW dimension, so MFMA
// fragments (8 consecutive contraction values per lane) are column gathers.  Tiles are staged row-major
// [64 m][128 cols] in LDS exactly as they lie in HBM (16-byte coalesced loads), and fragments are read
//   VARIANT 1: with ds_read_b64_tr_b16 (CDNA4 LDS transpose read: a 16-lane group reads a [4 m][16 col]
//              block, lane s supplying the 8-byte piece (row s>>2, cols 4*(s&3)..+3) and receiving
//              column s) -- two reads per 16x16x32 fragment;
//   VARIANT 0: with eight scalar 16-bit LDS reads (slow but layout-obvious; kept as the cross-check).
// The MFMA "A" operand is fed from X columns (k) and "B" from dY columns (n): in the C/D layout a lane
// then owns one output row n and, with the permuted column order  k = 16*(i>>2) + 4*tk + (i&3), sixteen
// consecutive k -> 16-byte stores.  M is split across blockIdx.z; partial products go to fp32 slabs
// that a second kernel reduces (deterministic, no atomics) and converts to fp16 (+= when beta = 1).
#include "common.h"
#include <cstdlib>

#define TN_BN 128     // output rows (n) per block
#define TN_BK 128     // output cols (k) per block
#define TN_BM 64      // contraction rows per stage
#define TN_THREADS 256
#define TN_PITCH (128 + 8)   // halfs; +16 B pad per LDS row

struct GemmTnParams {
    const f16* A; int64_t lda;   // dY [M,N]
    const f16* B; int64_t ldb;   // X  [M,K]
    f16* C; int64_t ldc;         // [N,K]
    float* slab;                 // [splits][N][K] fp32 (when splits > 1)
    float* bias_slab;            // [splits][N] fp32 (when splits > 1 and bias_out)
    f16* bias_out;               // [N] or NULL: column sums of A (bias gradient), fused
    int M, N, K, beta, splits, rows_per_split;
    int tiles_k, tiles_n, xcd_remap, split_major;
};

typedef __fp16 fp16x4_t __attribute__((__vector_size__(4 * sizeof(__fp16))));

DEVFN f16x4 lds_tr_read(const f16* p) {
    fp16x4_t t = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)(p));
    return __builtin_bit_cast(f16x4, t);
}

template <int VARIANT>
__global__ __launch_bounds__(TN_THREADS, 2) void gemm_tn_kernel(GemmTnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    f16* smem = reinterpret_cast<f16*>(smem_raw);
    const int TILE = TN_BM * TN_PITCH;   // halfs
    // layout: [buf][ A tile | B tile ]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wid >> 1, wk = wid & 1;
    const int g = lane >> 4, li = lane & 15;

    // 1-D grid, tile order (n-tile, k-tile, split); with xcd_remap the workgroups of one XCD (blockIdx % 8) take a contiguous
    // range of that order, i.e. whole n-tiles: the dY panel of an n-tile is fetched into ONE XCD's L2 and shared by its k-tiles.
    int bid = blockIdx.x;
    if (p.xcd_remap) {
        const int nb = gridDim.x, q = nb >> 3, r = nb & 7, xcd = bid & 7, loc = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    int split, ktile, ntile;
    if (p.split_major) {      // order (split, n-tile, k-tile): with xcd_remap one XCD works on ONE row range -> its co-resident
        ktile = bid % p.tiles_k;                       // workgroups share the few dY / X panels of that range in the XCD's L2
        ntile = (bid / p.tiles_k) % p.tiles_n;
        split = bid / (p.tiles_k * p.tiles_n);
    } else {
        split = bid % p.splits;
        ktile = (bid / p.splits) % p.tiles_k;
        ntile = bid / (p.splits * p.tiles_k);
    }
    const int n0 = ntile * TN_BN;
    const int k0 = ktile * TN_BK;
    const int m_begin = split * p.rows_per_split;
    const int m_end = min(p.M, m_begin + p.rows_per_split);
    const int nstages = (m_end - m_begin + TN_BM - 1) / TN_BM;

    // staging: thread -> (row = tid/16 + 16*i, chunk = tid%16)
    const int srow = tid >> 4, sch = tid & 15;
    const bool a_ok = (n0 + sch * 8) < p.N;   // chunk start inside the logical width (lda >= roundup8(N))
    const bool b_ok = (k0 + sch * 8) < p.K;

    f32x4 acc[4][4];   // [tn][tk]
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // fused bias gradient: in k-tile 0, the wk == 0 waves also multiply the dY fragments by an all-ones operand, which
    // yields the column sums of dY (every row of the 16x16 result is the same) for 4 extra MFMAs per 32-row step.
    const bool do_bias = (p.bias_out != nullptr) && ktile == 0 && wk == 0;
    f32x4 bacc[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) bacc[a] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const f16x8 ones = (f16x8){(f16)1.f, (f16)1.f, (f16)1.f, (f16)1.f, (f16)1.f, (f16)1.f, (f16)1.f, (f16)1.f};

    u32x4 areg[4], breg[4];
    auto gload = [&](int st) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m_begin + st * TN_BM + srow + 16 * i;
            const bool ok = m < m_end;
            areg[i] = (ok && a_ok) ? *reinterpret_cast<const u32x4*>(p.A + (int64_t)m * p.lda + n0 + sch * 8) : (u32x4){0, 0, 0, 0};
            breg[i] = (ok && b_ok) ? *reinterpret_cast<const u32x4*>(p.B + (int64_t)m * p.ldb + k0 + sch * 8) : (u32x4){0, 0, 0, 0};
        }
    };
    auto lstore = [&](int buf) {
        f16* as = smem + buf * 2 * TILE;
        f16* bs = as + TILE;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = srow + 16 * i;
            *reinterpret_cast<u32x4*>(as + r * TN_PITCH + sch * 8) = areg[i];
            *reinterpret_cast<u32x4*>(bs + r * TN_PITCH + sch * 8) = breg[i];
        }
    };
    auto compute = [&](int buf) {
        const f16* as = smem + buf * 2 * TILE;   // dY tile [m][n]
        const f16* bs = as + TILE;               // X tile  [m][k]
#pragma unroll
        for (int ms = 0; ms < 2; ++ms) {
            const int mrow = ms * 32 + 8 * g;     // this lane group's 8 contraction rows
            f16x8 xf[4], yf[4];
            if (VARIANT == 1) {
                // lane s = li supplies piece (row s>>2, 4 cols at 4*(s&3) [+ permutation]) and receives column s
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const f16* px = bs + (mrow + (li >> 2)) * TN_PITCH + wk * 64 + 16 * (li & 3) + 4 * t;
                    const f16* py = as + (mrow + (li >> 2)) * TN_PITCH + wn * 64 + 16 * t + 4 * (li & 3);
                    f16x4 x0 = lds_tr_read(px), x1 = lds_tr_read(px + 4 * TN_PITCH);
                    f16x4 y0 = lds_tr_read(py), y1 = lds_tr_read(py + 4 * TN_PITCH);
                    xf[t] = (f16x8){x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
                    yf[t] = (f16x8){y0[0], y0[1], y0[2], y0[3], y1[0], y1[1], y1[2], y1[3]};
                }
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int xc = wk * 64 + 16 * (li >> 2) + 4 * t + (li & 3);
                    const int yc = wn * 64 + 16 * t + li;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        xf[t][e] = bs[(mrow + e) * TN_PITCH + xc];
                        yf[t][e] = as[(mrow + e) * TN_PITCH + yc];
                    }
                }
            }
#pragma unroll
            for (int tn = 0; tn < 4; ++tn)
#pragma unroll
                for (int tk = 0; tk < 4; ++tk)
                    acc[tn][tk] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xf[tk], yf[tn], acc[tn][tk], 0, 0, 0);
            if (do_bias) {
#pragma unroll
                for (int tn = 0; tn < 4; ++tn) bacc[tn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ones, yf[tn], bacc[tn], 0, 0, 0);
            }
        }
    };

    if (nstages > 0) {
        gload(0);
        lstore(0);
        __syncthreads();
        for (int st = 0; st < nstages; ++st) {
            const int buf = st & 1;
            if (st + 1 < nstages) gload(st + 1);
            compute(buf);
            if (st + 1 < nstages) lstore(buf ^ 1);
            __syncthreads();
        }
    }

    // epilogue: lane owns row n (per tn) and 16 consecutive k starting at kc0
    const int kc0 = k0 + wk * 64 + 16 * g;
#pragma unroll
    for (int tn = 0; tn < 4; ++tn) {
        const int n = n0 + wn * 64 + 16 * tn + li;
        if (n >= p.N) continue;
        if (do_bias && g == 0) {      // all 16 rows of bacc are equal; lanes 0..15 (g == 0) publish reg 0
            if (p.splits > 1) p.bias_slab[(int64_t)split * p.N + n] = bacc[tn][0];
            else p.bias_out[n] = (f16)(p.beta ? (float)p.bias_out[n] + bacc[tn][0] : bacc[tn][0]);
        }
        if (p.splits > 1) {
            float* dst = p.slab + ((int64_t)split * p.N + n) * p.K + kc0;
#pragma unroll
            for (int tk = 0; tk < 4; ++tk)
                if (kc0 + 4 * tk < p.K) *reinterpret_cast<f32x4*>(dst + 4 * tk) = acc[tn][tk];
        } else {
            f16* dst = p.C + (int64_t)n * p.ldc + kc0;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (kc0 + 8 * h >= p.K) continue;
                f16x8 o;
                if (p.beta) o = ld8(dst + 8 * h);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float v = acc[tn][2 * h + (j >> 2)][j & 3];
                    o[j] = (f16)(p.beta ? (float)o[j] + v : v);
                }
                st8_out<VLP_SS_TN>(dst + 8 * h, o);
            }
        }
    }
}

// =================================================================================================
// VARIANT 2: LDS-DMA staging (global_load_lds_dwordx4) for the transpose-read kernel.
// LDS tiles are [64 m][128 cols], 256-B rows, lane-linear as the DMA writes them; the 16-B chunk c of row r sits at
// physical chunk c ^ swz(r), swz(r) = 2*((r&3) | ((r>>3)&1)<<2) (applied to the per-lane SOURCE address).  A
// ds_read_b64_tr_b16 half-wave touches 8 rows (r&3 x two 8-row groups) x 32 contiguous bytes: with this swizzle the eight
// 32-B segments fall on eight different 32-B bank groups -> conflict-free, for BOTH operands because both use the natural
// column order here (lane i of a 16-group receives column 16*t + i).  The price is an epilogue of 4-element (not
// 16-element) runs per lane: 16-B fp32 slab stores, 8-B fp16 stores.  Partial stages (M % 64, only the last stage of the
// last split) are zero-filled through registers into the same image.
// =================================================================================================
DEVFN int tn_swz(int r) { return ((r & 3) | (((r >> 3) & 1) << 2)) << 1; }

// BN_T x BK_T output tile (each 128 or 256); one wave per 64x64 sub-tile -> 4 or 8 waves.  `bid` = linear tile index of this workgroup
// inside the problem p (after any XCD remap): shared by the single-problem kernel and the grouped launch below.
// NS = LDS stages of the ring (NS - 2 stages stay in flight across a stage's barrier), BM_T = contraction rows per stage (64 or 32: with 32 rows a
// 128x128 tile fits FOUR stages in the 64 KiB that let two workgroups share a CU)
// SK (segments of the balanced grouped launch, gemm_tn_grouped_sk_kernel): the workgroup computes the contraction rows [sk_m_begin, sk_m_end) of
// the tile only and either WRITES its fp32 accumulators to `sk_part` (the kernel publishes `sk_flag` later), or -- the tile's finisher -- waits
// for the flag, ADDS the partner's partial (own chain + partner's chain, in that order: deterministic) and runs the ordinary epilogue.
#define TN_SK_WRITE 1
#define TN_SK_FINAL 2
#define TN_SK_SLOTS 20       // f32x4 per thread in a partial: 16 accumulator tiles + 4 bias tiles
template <int BN_T, int BK_T, int NS = 2, int BM_T = TN_BM, bool SK = false>
DEVFN void tn_glds_tile(const GemmTnParams& p, int bid, f16* smem, int sk_m_begin = 0, int sk_m_end = 0, int sk_role = 0, float* sk_part = nullptr,
                        int* sk_flag = nullptr) {
    constexpr int WK_ = BK_T / 64;
    constexpr int T = (BN_T / 64) * WK_ * 64;            // threads
    constexpr int ATILE = BM_T * BN_T, BTILE = BM_T * BK_T;   // halfs
    constexpr int ACH = BN_T / 8, BCH = BK_T / 8;        // 16-B chunks per tile row (16 or 32)
    constexpr int AP = BM_T * ACH / T, BP = BM_T * BCH / T;   // staging passes
    constexpr int ARP = T / ACH, BRP = T / BCH;          // tile rows covered per pass
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wid / WK_, wk = wid % WK_;
    const int g = lane >> 4, li = lane & 15;

    int split, ktile, ntile;
    if (p.split_major) {
        ktile = bid % p.tiles_k;
        ntile = (bid / p.tiles_k) % p.tiles_n;
        split = bid / (p.tiles_k * p.tiles_n);
    } else {
        split = bid % p.splits;
        ktile = (bid / p.splits) % p.tiles_k;
        ntile = bid / (p.splits * p.tiles_k);
    }
    const int n0 = ntile * BN_T;
    const int k0 = ktile * BK_T;
    const int m_begin = SK ? sk_m_begin : split * p.rows_per_split;
    const int m_end = SK ? sk_m_end : min(p.M, m_begin + p.rows_per_split);
    const int nstages = (m_end - m_begin + BM_T - 1) / BM_T;

    // staging: pass i covers tile rows ARP*i .. ; thread -> (row = ARP*i + tid/ACH, physical chunk = tid%ACH).  The swizzle acts on
    // the low 4 bits of the chunk index (one 256-byte bank row), so 512-byte rows behave like two independent 256-byte halves.
    const int arow = tid / ACH, apc = tid % ACH, brow = tid / BCH, bpc = tid % BCH;
    int acol[AP], bcol[BP];
#pragma unroll
    for (int i = 0; i < AP; ++i) acol[i] = min(n0 + (apc ^ tn_swz(arow + ARP * i)) * 8, (int)p.lda - 8);
#pragma unroll
    for (int i = 0; i < BP; ++i) bcol[i] = min(k0 + (bpc ^ tn_swz(brow + BRP * i)) * 8, (int)p.ldb - 8);

    f32x4 acc[4][4];   // [tn][tk]
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bool do_bias = (p.bias_out != nullptr) && ktile == 0 && wk == 0;
    f32x4 bacc[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) bacc[a] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const f16x8 ones = (f16x8){(f16)1.f, (f16)1.f, (f16)1.f, (f16)1.f, (f16)1.f, (f16)1.f, (f16)1.f, (f16)1.f};

    const uint32_t smem_lds = lds_addr_of(smem);
    // per-thread DMA source of stage 0 (pass i); a stage advances every pointer by BM_T rows -- one 64-bit add per DMA instead of a
    // row * pitch multiply
    const f16* asrc[AP];
    const f16* bsrc[BP];
#pragma unroll
    for (int i = 0; i < AP; ++i) asrc[i] = p.A + (int64_t)(m_begin + arow + ARP * i) * p.lda + acol[i];
#pragma unroll
    for (int i = 0; i < BP; ++i) bsrc[i] = p.B + (int64_t)(m_begin + brow + BRP * i) * p.ldb + bcol[i];
    auto stage = [&](int st, int buf) {
        f16* as = smem + buf * (ATILE + BTILE);
        f16* bs = as + ATILE;
        const int mbase = m_begin + st * BM_T;
        if (mbase + BM_T <= m_end) {            // full stage (wave-uniform): LDS-DMA, 64 lanes = 1 KiB contiguous in LDS
            const uint32_t as_l = smem_lds + (uint32_t)(buf * (ATILE + BTILE)) * 2u, bs_l = as_l + (uint32_t)ATILE * 2u;
            const int64_t aoff = (int64_t)st * BM_T * p.lda, boff = (int64_t)st * BM_T * p.ldb;
#pragma unroll
            for (int i = 0; i < AP; ++i) glds16(asrc[i] + aoff, as_l + (uint32_t)((ARP * i) * BN_T + wid * 512) * 2u);
#pragma unroll
            for (int i = 0; i < BP; ++i) glds16(bsrc[i] + boff, bs_l + (uint32_t)((BRP * i) * BK_T + wid * 512) * 2u);
        } else {                                  // ragged tail: through registers with zero fill, same LDS image
#pragma unroll
            for (int i = 0; i < AP; ++i) {
                const int r = arow + ARP * i, m = mbase + r;
                u32x4 v = (u32x4){0, 0, 0, 0};
                if (m < m_end) v = *reinterpret_cast<const u32x4*>(p.A + (int64_t)m * p.lda + acol[i]);
                *reinterpret_cast<u32x4*>(as + r * BN_T + apc * 8) = v;
            }
#pragma unroll
            for (int i = 0; i < BP; ++i) {
                const int r = brow + BRP * i, m = mbase + r;
                u32x4 v = (u32x4){0, 0, 0, 0};
                if (m < m_end) v = *reinterpret_cast<const u32x4*>(p.B + (int64_t)m * p.ldb + bcol[i]);
                *reinterpret_cast<u32x4*>(bs + r * BK_T + bpc * 8) = v;
            }
        }
    };
    // fragments of the m sub-step ms (32 rows) of a stage: 16 transpose reads
    auto read_frags = [&](int buf, int ms, f16x8 (&xf)[4], f16x8 (&yf)[4]) {
        const f16* as = smem + buf * (ATILE + BTILE);   // dY tile [m][n]
        const f16* bs = as + ATILE;                     // X tile  [m][k]
        const int r0 = ms * 32 + 8 * g + (li >> 2);     // this lane's piece row (first read); second read: +4 (same swizzle)
        const int sw = tn_swz(r0);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int cx = wk * 64 + 16 * t + 4 * (li & 3), cy = wn * 64 + 16 * t + 4 * (li & 3);
            const f16* px = bs + r0 * BK_T + (((cx >> 3) ^ sw) << 3) + (cx & 4);
            const f16* py = as + r0 * BN_T + (((cy >> 3) ^ sw) << 3) + (cy & 4);
            const u32x2 x0 = __builtin_bit_cast(u32x2, lds_tr_read(px)), x1 = __builtin_bit_cast(u32x2, lds_tr_read(px + 4 * BK_T));
            const u32x2 y0 = __builtin_bit_cast(u32x2, lds_tr_read(py)), y1 = __builtin_bit_cast(u32x2, lds_tr_read(py + 4 * BN_T));
            xf[t] = __builtin_bit_cast(f16x8, (u32x4){x0[0], x0[1], x1[0], x1[1]});
            yf[t] = __builtin_bit_cast(f16x8, (u32x4){y0[0], y0[1], y1[0], y1[1]});
        }
    };
    auto mfma_frags = [&](const f16x8 (&xf)[4], const f16x8 (&yf)[4]) {
#pragma unroll
        for (int tn = 0; tn < 4; ++tn)
#pragma unroll
            for (int tk = 0; tk < 4; ++tk)
                acc[tn][tk] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xf[tk], yf[tn], acc[tn][tk], 0, 0, 0);
        if (do_bias) {
#pragma unroll
            for (int tn = 0; tn < 4; ++tn) bacc[tn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ones, yf[tn], bacc[tn], 0, 0, 0);
        }
    };

    if (nstages > 0) {
        // two LDS stages as a ring: wait for stage st (DMA: vmcnt; the ragged tail's ds_writes: lgkmcnt), ONE raw barrier -- every wave's
        // pieces have landed and every wave is done with the fragment reads of stage st-1 -- then the first half's fragment reads, the refill
        // of the freed buffer with stage st+1 (it streams in under this stage's MFMAs), the second half's reads, and the 32 MFMAs
        constexpr int LPS = AP + BP;                  // DMA instructions per thread per full stage
#pragma unroll
        for (int i = 0; i < NS - 1; ++i)
            if (i < nstages) stage(i, i);
        int buf = 0, nbuf = NS - 1;
        const bool tail_ragged = ((m_end - m_begin) % BM_T) != 0;
        for (int st = 0; st < nstages; ++st) {
            // stage st must have landed; up to NS - 2 younger FULL stages may stay in flight.  A ragged last stage goes through registers +
            // ds_write (compiler-tracked loads, waited for -- with everything older -- inside stage()): once it is among the younger
            // ones the queue has been drained already, and vmcnt(0) costs nothing.
            const int left = nstages - 1 - st;
            int younger = left < NS - 2 ? left : NS - 2;
            if (tail_ragged && st + younger >= nstages - 1) younger = 0;
            if (NS >= 4 && younger == 2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * LPS) : "memory");
            else if (NS >= 3 && younger >= 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(LPS) : "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            f16x8 x0[4], y0[4], x1[4], y1[4];
            read_frags(buf, 0, x0, y0);
            __builtin_amdgcn_sched_barrier(0);
            if (st + NS - 1 < nstages) stage(st + NS - 1, nbuf);
            if (BM_T == 64) read_frags(buf, 1, x1, y1);
            __builtin_amdgcn_sched_barrier(0);
            mfma_frags(x0, y0);
            if (BM_T == 64) mfma_frags(x1, y1);
            buf = (buf + 1 == NS) ? 0 : buf + 1;
            nbuf = (nbuf + 1 == NS) ? 0 : nbuf + 1;
        }
    }

    if (SK) {
        // lane-linear partial image: slot i of thread tid at f32x4 index i * T + tid (16-byte stores / loads, fully coalesced)
        f32x4* part = reinterpret_cast<f32x4*>(sk_part) + tid;
        if (sk_role == TN_SK_WRITE) {
#pragma unroll
            for (int tn = 0; tn < 4; ++tn)
#pragma unroll
                for (int tk = 0; tk < 4; ++tk) part[(tn * 4 + tk) * T] = acc[tn][tk];
            if (do_bias) {
#pragma unroll
                for (int tn = 0; tn < 4; ++tn) part[(16 + tn) * T] = bacc[tn];
            }
            __syncthreads();       // every wave is done with the LDS stages before the next segment's prologue refills them; the kernel publishes once
            return;
        }
        // finisher: the group's tail workgroup publishes all six partials at its end, about when the main workgroups end; poll relaxed, acquire once
        if (tid == 0) {
            while (__hip_atomic_load(sk_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) __builtin_amdgcn_s_sleep(8);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
#pragma unroll
        for (int tn = 0; tn < 4; ++tn)
#pragma unroll
            for (int tk = 0; tk < 4; ++tk) acc[tn][tk] += part[(tn * 4 + tk) * T];
        if (do_bias) {
#pragma unroll
            for (int tn = 0; tn < 4; ++tn) bacc[tn] += part[(16 + tn) * T];
        }
    }
    // epilogue: lane owns row n (per tn) and, per tk, 4 consecutive k at k0 + wk*64 + 16*tk + 4*g
#pragma unroll
    for (int tn = 0; tn < 4; ++tn) {
        const int n = n0 + wn * 64 + 16 * tn + li;
        if (n >= p.N) continue;
        if (do_bias && g == 0) {
            if (p.splits > 1) p.bias_slab[(int64_t)split * p.N + n] = bacc[tn][0];
            else p.bias_out[n] = (f16)(p.beta ? (float)p.bias_out[n] + bacc[tn][0] : bacc[tn][0]);
        }
#pragma unroll
        for (int tk = 0; tk < 4; ++tk) {
            const int kc = k0 + wk * 64 + 16 * tk + 4 * g;
            if (kc >= p.K) continue;
            if (p.splits > 1) {
                *reinterpret_cast<f32x4*>(p.slab + ((int64_t)split * p.N + n) * p.K + kc) = acc[tn][tk];
            } else {
                f16* dst = p.C + (int64_t)n * p.ldc + kc;
                f16x4 o;
                if (p.beta) o = ld4(dst);
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = (f16)(p.beta ? (float)o[j] + acc[tn][tk][j] : acc[tn][tk][j]);
                st4_out<VLP_SS_TN>(dst, o);
            }
        }
    }
}

DEVFN int tn_xcd_remap(int bid, int nb) {      // bijective for any grid size: XCD x owns (q+1) tiles if x < r else q
    const int q = nb >> 3, r = nb & 7, xcd = bid & 7, loc = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
}

template <int BN_T, int BK_T>
__global__ __launch_bounds__((BN_T / 64) * (BK_T / 64) * 64, 2) void gemm_tn_glds_kernel(GemmTnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    int bid = blockIdx.x;
    if (p.xcd_remap) bid = tn_xcd_remap(bid, gridDim.x);
    tn_glds_tile<BN_T, BK_T>(p, bid, reinterpret_cast<f16*>(smem_raw));
}

// -------------------------------------------------------------------------------------------------
// Grouped launch: the weight gradients of several Linears in ONE grid (the four of a BertLayer: 36 + 108 + 144 + 144 = 432
// tiles of 128x128).  A single wgrad has too few output tiles for 256 CUs, which is what forced the split-M form above (fp32 slabs:
// 28 MB written + read per FFN wgrad, plus a reduce launch each); together they fill the chip with every workgroup walking the
// WHOLE contraction (M = 10 688 rows): no slabs, no reduce kernels, one fp32 accumulation chain per output element.
// -------------------------------------------------------------------------------------------------
#define TN_GROUP_MAX 8
struct TnGroupEntry {
    const f16* A; int64_t lda;
    const f16* B; int64_t ldb;
    f16* C; int64_t ldc;
    f16* bias_out;
    int M, N, K, beta, tiles_k, tile_begin;
};
struct TnGroupParams {
    TnGroupEntry e[TN_GROUP_MAX];
    int count, total_tiles, xcd_remap;
    int nst, tail;           // balanced form: contraction stages (of 64 rows) per tile, the same for every problem of the group; stages of the tail cohort
};

template <int BN_T, int BK_T, int NS, int BM_T>
__global__ __launch_bounds__((BN_T / 64) * (BK_T / 64) * 64, 2) void gemm_tn_grouped_kernel(TnGroupParams gp) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    int bid = blockIdx.x;
    if (gp.xcd_remap) bid = tn_xcd_remap(bid, gridDim.x);
    int j = 0;
#pragma unroll
    for (int i = 1; i < TN_GROUP_MAX; ++i)
        if (i < gp.count && bid >= gp.e[i].tile_begin) j = i;
    const TnGroupEntry& e = gp.e[j];
    GemmTnParams p;
    p.A = e.A; p.lda = e.lda; p.B = e.B; p.ldb = e.ldb; p.C = e.C; p.ldc = e.ldc;
    p.slab = nullptr; p.bias_slab = nullptr; p.bias_out = e.bias_out;
    p.M = e.M; p.N = e.N; p.K = e.K; p.beta = e.beta; p.splits = 1; p.rows_per_split = (e.M + TN_BM - 1) / TN_BM * TN_BM;
    p.tiles_k = e.tiles_k; p.tiles_n = 0; p.xcd_remap = 0; p.split_major = 0;
    tn_glds_tile<BN_T, BK_T, NS, BM_T>(p, bid - e.tile_begin, reinterpret_cast<f16*>(smem_raw));
}

#ifdef VLP_LAB_BUILD      // measured in round 4 and not kept (profiles/r04_grouped_wgrad_balanced.txt): investigation builds only
// Balanced form of the grouped launch ("tail cohort").  432 equal tiles on 2 x 256 workgroup slots leave 80 CUs with one workgroup instead
// of two.  Here the tiles are taken six at a time by SEVEN workgroups: six "main" workgroups walk the first nst - tail contraction stages of
// one tile each -- all main workgroups of the launch sweep the contraction rows in step, which is what keeps every operand block a one-time
// HBM read (a plain stream-K deal of equal contiguous runs was built first: correct, and 18 % SLOWER than the unbalanced launch, because runs
// that start at seven different row offsets read every operand block at seven different times: profiles/r04_grouped_wgrad_balanced.txt) --
// and the seventh walks the LAST `tail` stages of all six tiles, one after the other, and leaves six fp32 partials in `part` (80 KB per tile).
// It publishes them with ONE agent-scope release at its end; a main workgroup then adds its tile's partial (own chain + tail chain: deterministic)
// and runs the ordinary epilogue.  `tail` is chosen so that 6 * (tail + restart cost) ~ nst - tail.
template <int BN_T, int BK_T, int NS, int BM_T>
__global__ __launch_bounds__((BN_T / 64) * (BK_T / 64) * 64, 2) void gemm_tn_grouped_sk_kernel(TnGroupParams gp, float* part, int* flags) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int T = (BN_T / 64) * (BK_T / 64) * 64;
    int u = blockIdx.x;
    if (gp.xcd_remap) u = tn_xcd_remap(u, gridDim.x);     // a group's seven workgroups stay on one XCD when the grid is a multiple of 56
    const int grp = u / 7, j = u % 7;
    const int nst = gp.nst, cut = gp.nst - gp.tail;
    const int nseg = (j < 6) ? 1 : 6;
    for (int seg = 0; seg < nseg; ++seg) {
        const int tile = 6 * grp + ((j < 6) ? j : seg);
        int ei = 0;
#pragma unroll
        for (int i = 1; i < TN_GROUP_MAX; ++i)
            if (i < gp.count && tile >= gp.e[i].tile_begin) ei = i;
        const TnGroupEntry& e = gp.e[ei];
        GemmTnParams p;
        p.A = e.A; p.lda = e.lda; p.B = e.B; p.ldb = e.ldb; p.C = e.C; p.ldc = e.ldc;
        p.slab = nullptr; p.bias_slab = nullptr; p.bias_out = e.bias_out;
        p.M = e.M; p.N = e.N; p.K = e.K; p.beta = e.beta; p.splits = 1; p.rows_per_split = 0;
        p.tiles_k = e.tiles_k; p.tiles_n = 0; p.xcd_remap = 0; p.split_major = 0;
        if (j < 6)
            tn_glds_tile<BN_T, BK_T, NS, BM_T, true>(p, tile - e.tile_begin, reinterpret_cast<f16*>(smem_raw), 0, cut * BM_T, TN_SK_FINAL,
                                                       part + (int64_t)tile * (TN_SK_SLOTS * T * 4), flags + grp);
        else
            tn_glds_tile<BN_T, BK_T, NS, BM_T, true>(p, tile - e.tile_begin, reinterpret_cast<f16*>(smem_raw), cut * BM_T, min(e.M, nst * BM_T), TN_SK_WRITE,
                                                       part + (int64_t)tile * (TN_SK_SLOTS * T * 4), flags + grp);
    }
    if (j == 6) {
        // publish (cdna_hip_programming.md, in-launch hand-off): every wave's stores waited for, workgroup barrier, ONE agent-scope release by
        // one lane, then the flag
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(flags + grp, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

#endif      // VLP_LAB_BUILD

// out[n,k] = (beta ? out : 0) + sum_s slab[s][n][k]; the tail of the grid reduces the fused bias partials [s][N]
__global__ void gemm_tn_reduce_kernel(const float* slab, f16* C, int64_t ldc, int N, int K, int splits, int beta,
                                      const float* bias_slab, f16* bias_out, int main_blocks) {
    if ((int)blockIdx.x >= main_blocks) {
        const int n = ((int)blockIdx.x - main_blocks) * blockDim.x + threadIdx.x;
        if (n < N) {
            float s = 0.f;
            for (int sp = 0; sp < splits; ++sp) s += bias_slab[(int64_t)sp * N + n];
            bias_out[n] = (f16)(beta ? (float)bias_out[n] + s : s);
        }
        return;
    }
    const int64_t total8 = (int64_t)N * (K / 8);
    const int64_t stride = (int64_t)N * K;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total8; i += (int64_t)main_blocks * blockDim.x) {
        const int n = (int)(i / (K / 8));
        const int k = (int)(i % (K / 8)) * 8;
        const float* s = slab + (int64_t)n * K + k;
        float v[8];
        f32x4 a0 = *reinterpret_cast<const f32x4*>(s), a1 = *reinterpret_cast<const f32x4*>(s + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[j] = a0[j]; v[4 + j] = a1[j]; }
        for (int sp = 1; sp < splits; ++sp) {
            a0 = *reinterpret_cast<const f32x4*>(s + sp * stride);
            a1 = *reinterpret_cast<const f32x4*>(s + sp * stride + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[j] += a0[j]; v[4 + j] += a1[j]; }
        }
        f16* dst = C + (int64_t)n * ldc + k;
        f16x8 o;
        if (beta) o = ld8(dst);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (f16)(beta ? (float)o[j] + v[j] : v[j]);
        st8_out<VLP_SS_TN>(dst, o);
    }
}

static int choose_splits(int M, int N, int K) {
    const int tiles = cdiv(N, TN_BN) * cdiv(K, TN_BK);
    int s = cdiv(512, tiles);                // aim for >= 2 workgroups per CU
    const int max_by_rows = M / 256 > 0 ? M / 256 : 1;
    if (s > max_by_rows) s = max_by_rows;
    if (s > 16) s = 16;
    if (s < 1) s = 1;
    return s;
}

extern "C" int64_t vlp_gemm_tn_workspace_bytes(int32_t M, int32_t N, int32_t K) {
    const int s = 16;   // upper bound of choose_splits / explicit splits
    (void)M;
    return (int64_t)s * N * K * (int64_t)sizeof(float) + (int64_t)s * ((N + 63) / 64 * 64) * (int64_t)sizeof(float);
}

extern "C" int vlp_gemm_tn(const vlp_gemm_tn_args* a, void* stream) {
    VLP_CHECK_ARG(a != nullptr, "vlp_gemm_tn: null args");
    VLP_ENTER(a->A, "vlp_gemm_tn");
    VLP_CHECK_ARG(a->A && a->B && a->C, "vlp_gemm_tn: null operand");
    VLP_CHECK_ARG(a->M > 0 && a->N > 0 && a->K > 0, "vlp_gemm_tn: bad shape");
    VLP_CHECK_ARG(a->K % 8 == 0, "vlp_gemm_tn: K=%d must be a multiple of 8", a->K);
    VLP_CHECK_ARG(a->lda % 8 == 0 && a->ldb % 8 == 0 && a->ldc % 8 == 0, "vlp_gemm_tn: leading dims must be multiples of 8");
    VLP_CHECK_ARG(a->lda >= (a->N + 7) / 8 * 8 && a->ldb >= a->K && a->ldc >= a->K, "vlp_gemm_tn: leading dim too small");
    VLP_CHECK_ARG(((uintptr_t)a->A | (uintptr_t)a->B | (uintptr_t)a->C) % 16 == 0, "vlp_gemm_tn: operands must be 16-byte aligned");
    VLP_CHECK_ARG(a->beta == 0 || a->beta == 1, "vlp_gemm_tn: beta must be 0 or 1");
    int splits = a->splits > 0 ? a->splits : choose_splits(a->M, a->N, a->K);
    if (splits > 16) splits = 16;
    GemmTnParams p;
    p.A = (const f16*)a->A; p.lda = a->lda;
    p.B = (const f16*)a->B; p.ldb = a->ldb;
    p.C = (f16*)a->C; p.ldc = a->ldc;
    p.M = a->M; p.N = a->N; p.K = a->K; p.beta = a->beta;
    int rps = cdiv(a->M, splits);
    rps = (rps + TN_BM - 1) / TN_BM * TN_BM;
    splits = cdiv(a->M, rps);
    p.splits = splits; p.rows_per_split = rps;
    p.slab = (float*)a->workspace;
    p.bias_out = (f16*)a->bias_out;
    p.bias_slab = p.slab ? p.slab + (int64_t)splits * a->N * a->K : nullptr;
    p.tiles_k = cdiv(a->K, TN_BK);
    p.tiles_n = cdiv(a->N, TN_BN);
    p.xcd_remap = (a->variant & 8) ? 1 : 0;
    p.split_major = (a->variant & 16) ? 1 : 0;
    if (splits > 1) {
        const int64_t need = (int64_t)splits * a->N * a->K * (int64_t)sizeof(float) + (a->bias_out ? (int64_t)splits * a->N * (int64_t)sizeof(float) : 0);
        if (!a->workspace || a->workspace_bytes < need)
            return vlp_set_error(VLP_ERR_WORKSPACE, "vlp_gemm_tn: workspace %lld < %lld bytes", (long long)a->workspace_bytes, (long long)need);
        VLP_CHECK_ARG((uintptr_t)a->workspace % 16 == 0, "vlp_gemm_tn: workspace must be 16-byte aligned");
    }
    hipStream_t s = (hipStream_t)stream;
    const int base = a->variant & 7;
#define LAUNCH_TN_GLDS(BNT, BKT)                                                                                        \
    do {                                                                                                                \
        const size_t smem2 = (size_t)2 * TN_BM * ((BNT) + (BKT)) * sizeof(f16);                                         \
        VLP_ONCE_PER_DEVICE((void)hipFuncSetAttribute((const void*)gemm_tn_glds_kernel<BNT, BKT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem2));                                                            \
        p.tiles_k = cdiv(a->K, (BKT)); p.tiles_n = cdiv(a->N, (BNT));                                                   \
        hipLaunchKernelGGL((gemm_tn_glds_kernel<BNT, BKT>), dim3(p.tiles_k * p.tiles_n * splits), dim3(((BNT) / 64) * ((BKT) / 64) * 64), smem2, s, p); \
    } while (0)
    if (base == 2) LAUNCH_TN_GLDS(128, 128);
    else if (base == 3) LAUNCH_TN_GLDS(256, 128);
    else if (base == 4) LAUNCH_TN_GLDS(128, 256);
    else if (base == 5) return vlp_set_error(VLP_ERR_BAD_ARG, "vlp_gemm_tn: variant 5 (256x256 tiles, 16 waves) was removed: it needed 139 VGPRs under a "
                                             "128-register cap (spills) and lost to the 128x128 kernel on every training shape");
    else {
        dim3 grid(p.tiles_k * p.tiles_n * splits), block(TN_THREADS);
        const size_t smem = 2 * 2 * TN_BM * TN_PITCH * sizeof(f16);   // 68 KiB
        if (base == 1) {
            VLP_ONCE_PER_DEVICE((void)hipFuncSetAttribute((const void*)gemm_tn_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
            hipLaunchKernelGGL(gemm_tn_kernel<1>, grid, block, smem, s, p);
        } else {
            VLP_ONCE_PER_DEVICE((void)hipFuncSetAttribute((const void*)gemm_tn_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
            hipLaunchKernelGGL(gemm_tn_kernel<0>, grid, block, smem, s, p);
        }
    }
#undef LAUNCH_TN_GLDS
    VLP_CHECK_LAUNCH("vlp_gemm_tn");
    if (splits > 1) {
        const int64_t total8 = (int64_t)a->N * (a->K / 8);
        int blocks = (int)((total8 + 255) / 256);
        if (blocks > 2048) blocks = 2048;
        const int bias_blocks = a->bias_out ? cdiv(a->N, 256) : 0;
        hipLaunchKernelGGL(gemm_tn_reduce_kernel, dim3(blocks + bias_blocks), dim3(256), 0, s, p.slab, p.C, p.ldc, a->N, a->K, splits, a->beta,
                           p.bias_slab, p.bias_out, blocks);
        VLP_CHECK_LAUNCH("vlp_gemm_tn_reduce");
    }
    return VLP_OK;
}

#define TN_GROUP_DEFAULT_MODE 0
// workspace of the stream-K grouped launch (VLP_TN_GROUP_MODE=5) for `tiles` 128x128 output tiles over all problems: one fp32 partial
// (16 accumulator + 4 bias f32x4 per thread) and one flag per tile
extern "C" int64_t vlp_gemm_tn_grouped_workspace_bytes(int32_t tiles) {
    return (int64_t)tiles * TN_SK_SLOTS * 256 * 16 + (int64_t)((tiles + 63) / 64 * 64) * (int64_t)sizeof(int);
}

static int tn_check_one(const vlp_gemm_tn_args* a) {
    VLP_CHECK_ARG(a->A && a->B && a->C, "vlp_gemm_tn: null operand");
    VLP_CHECK_ARG(a->M > 0 && a->N > 0 && a->K > 0, "vlp_gemm_tn: bad shape");
    VLP_CHECK_ARG(a->K % 8 == 0, "vlp_gemm_tn: K=%d must be a multiple of 8", a->K);
    VLP_CHECK_ARG(a->lda % 8 == 0 && a->ldb % 8 == 0 && a->ldc % 8 == 0, "vlp_gemm_tn: leading dims must be multiples of 8");
    VLP_CHECK_ARG(a->lda >= (a->N + 7) / 8 * 8 && a->ldb >= a->K && a->ldc >= a->K, "vlp_gemm_tn: leading dim too small");
    VLP_CHECK_ARG(((uintptr_t)a->A | (uintptr_t)a->B | (uintptr_t)a->C) % 16 == 0, "vlp_gemm_tn: operands must be 16-byte aligned");
    VLP_CHECK_ARG(a->beta == 0 || a->beta == 1, "vlp_gemm_tn: beta must be 0 or 1");
    return VLP_OK;
}

extern "C" int vlp_gemm_tn_grouped(const vlp_gemm_tn_args* list, int32_t count, void* stream) {
    VLP_CHECK_ARG(list != nullptr && count >= 1 && count <= TN_GROUP_MAX, "vlp_gemm_tn_grouped: 1..%d problems", TN_GROUP_MAX);
    VLP_ENTER(list[0].A, "vlp_gemm_tn_grouped");
    // tile shape / ring depth of the grouped launch: 0 = 128x128 tiles, 2 stages (two 4-wave workgroups per CU); 1 = 256x128, 2 stages;
    // 2 = 256x128, 3 stages; 3 = 128x256, 3 stages (8-wave workgroups, one per CU); 4 = 128x128, FOUR stages of 32 contraction rows (same 64 KiB:
    // two workgroups per CU, three stages in flight).  VLP_TN_GROUP_MODE overrides (A/B runs).
    int mode = TN_GROUP_DEFAULT_MODE;      // the product library carries mode 0 only; 1 .. 5 were measured in round 4 and lose (-DVLP_LAB_BUILD)
#ifdef VLP_LAB_BUILD
    if (const char* e = getenv("VLP_TN_GROUP_MODE")) { mode = atoi(e); if (mode < 0 || mode > 5) mode = 0; }      // read per launch: A/B runs switch it inside one process
#endif
    // 5 = stream-K form of mode 0 (gemm_tn_grouped_sk_kernel): needs list[0].workspace (vlp_gemm_tn_grouped_workspace_bytes), a tile count that is a
    // multiple of 6, one M for all problems and >= 14 stages; otherwise the launch runs as mode 0
    const int bn = (mode == 0 || mode >= 3) ? 128 : 256, bk = mode == 3 ? 256 : 128;
    TnGroupParams gp;
    int tiles = 0;
    for (int i = 0; i < count; ++i) {
        const vlp_gemm_tn_args* a = list + i;
        const int rc = tn_check_one(a);
        if (rc != VLP_OK) return rc;
        TnGroupEntry& e = gp.e[i];
        e.A = (const f16*)a->A; e.lda = a->lda; e.B = (const f16*)a->B; e.ldb = a->ldb; e.C = (f16*)a->C; e.ldc = a->ldc;
        e.bias_out = (f16*)a->bias_out;
        e.M = a->M; e.N = a->N; e.K = a->K; e.beta = a->beta;
        e.tiles_k = cdiv(a->K, bk);
        e.tile_begin = tiles;
        tiles += e.tiles_k * cdiv(a->N, bn);
    }
    for (int i = count; i < TN_GROUP_MAX; ++i) { gp.e[i] = gp.e[0]; gp.e[i].tile_begin = 0x7fffffff; }
    gp.count = count; gp.total_tiles = tiles; gp.xcd_remap = 1;
#define LAUNCH_TN_GROUP(BNT, BKT, NSV, BMV)                                                                                             \
    do {                                                                                                                                \
        const size_t smem = (size_t)(NSV) * (BMV) * ((BNT) + (BKT)) * sizeof(f16);                                                      \
        VLP_ONCE_PER_DEVICE((void)hipFuncSetAttribute((const void*)gemm_tn_grouped_kernel<BNT, BKT, NSV, BMV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));                                                            \
        hipLaunchKernelGGL((gemm_tn_grouped_kernel<BNT, BKT, NSV, BMV>), dim3(tiles), dim3(((BNT) / 64) * ((BKT) / 64) * 64), smem, (hipStream_t)stream, gp); \
    } while (0)
#ifdef VLP_LAB_BUILD
    bool sk = (mode == 5);
    if (sk) {
        const int64_t need = vlp_gemm_tn_grouped_workspace_bytes(tiles);
        sk = tiles % 6 == 0 && list[0].workspace != nullptr && list[0].workspace_bytes >= need && (uintptr_t)list[0].workspace % 16 == 0 && cdiv(list[0].M, 64) >= 14;
        for (int i = 1; i < count && sk; ++i) sk = list[i].M == list[0].M;
    }
    if (sk) {
        float* part = (float*)list[0].workspace;
        int* flags = (int*)((char*)list[0].workspace + (int64_t)tiles * TN_SK_SLOTS * 256 * 16);
        gp.nst = cdiv(list[0].M, 64);
        // tail stages: 6 * (tail + r) = nst - tail with r ~ 3 stages of restart cost per segment (ring prologue + partial stores)
        gp.tail = (gp.nst - 18) / 7;
        if (const char* e = getenv("VLP_TN_SK_TAIL")) { const int v = atoi(e); if (v >= 1 && v < gp.nst) gp.tail = v; }
        if (gp.tail < 1) gp.tail = 1;
        if (hipMemsetAsync(flags, 0, (size_t)tiles * sizeof(int), (hipStream_t)stream) != hipSuccess)
            return vlp_set_error(VLP_ERR_HIP, "vlp_gemm_tn_grouped: hipMemsetAsync(flags): %s", hipGetErrorString(hipGetLastError()));
        const size_t smem = (size_t)2 * 64 * (128 + 128) * sizeof(f16);
        VLP_ONCE_PER_DEVICE((void)hipFuncSetAttribute((const void*)gemm_tn_grouped_sk_kernel<128, 128, 2, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        hipLaunchKernelGGL((gemm_tn_grouped_sk_kernel<128, 128, 2, 64>), dim3(tiles / 6 * 7), dim3(256), smem, (hipStream_t)stream, gp, part, flags);
    } else if (mode == 1) LAUNCH_TN_GROUP(256, 128, 2, 64);
    else if (mode == 2) LAUNCH_TN_GROUP(256, 128, 3, 64);
    else if (mode == 3) LAUNCH_TN_GROUP(128, 256, 3, 64);
    else if (mode == 4) LAUNCH_TN_GROUP(128, 128, 4, 32);
    else
#endif
    LAUNCH_TN_GROUP(128, 128, 2, 64);
#undef LAUNCH_TN_GROUP
    VLP_CHECK_LAUNCH("vlp_gemm_tn_grouped");
    return VLP_OK;
}
