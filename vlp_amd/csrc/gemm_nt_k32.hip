// NT GEMM, 256x256 tile, 16 waves, k tiles of 32 in a 4-stage LDS-DMA ring (variant 5 + 16 + 32 of vlp_gemm_nt; same contract and
// epilogue as gemm_nt.hip, same summation order per output element: k ascending in steps of 32 = one MFMA each).
//
// Why: the 256x256 tile with k tiles of 64 fits two stages in 160 KiB of LDS, i.e. ONE stage of prefetch.  Its loop was measured
// (profiles/r02_nt_loop_decomposition.txt) at 1.75 us per k tile = 1.0 (MFMAs + fragment reads) + 0.8 (the DMA round trip of the one
// stage in flight): the refill of stage kt+1 is issued at the top of iteration kt and must have landed at the top of kt+1, so every
// iteration pays what is left of the L2 / HBM latency after one iteration of compute.  Halving the k tile gives four 32 KiB stages in
// the same 128 KiB: three stages are in flight, a stage has three iterations (3 x 0.55 us) to land, and the same number of bytes moves.
// The price is one raw s_barrier per 16 MFMAs of a wave instead of per 32, and 64-byte rows in the DMA requests.
//
// LDS image of a stage: [X tile 256 rows x 32 halfs | W tile 256 rows x 32 halfs], 64-byte rows, four 16-byte chunks per row; chunk c of
// row r sits at physical chunk c ^ f(r) (applied on the per-lane DMA SOURCE address: the DMA destination is lane-linear).  A 16-lane
// group of a ds_read_b128 reads one chunk index (g = k / 8) of 16 rows; rows r and r' collide when r = r' (mod 4), so f must separate the
// four rows of a residue class: X fragments read consecutive rows (f = (r >> 2) & 3), W fragments read the permuted rows
// 16*(i >> 2) + 4*tn + (i & 3) (f = (r >> 4) & 3).
#include "common.h"
#include "gemm_nt.h"
#include "gemm_nt_epilogue.h"

#define K32_BK 32
#define K32_BM 256
#define K32_BN 256
#define K32_NS 4
#define K32_T 1024

DEVFN int k32_fx(int r) { return (r >> 2) & 3; }
DEVFN int k32_fw(int r) { return (r >> 4) & 3; }

template <bool SG>
__global__ __launch_bounds__(K32_T, 4) void gemm_nt_k32_kernel(GemmNtParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    f16* smem = reinterpret_cast<f16*>(smem_raw);
    constexpr int XT = K32_BM * K32_BK, WT = K32_BN * K32_BK;      // halfs per operand tile of a stage
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 2, wn = wid & 3;
    const int g = lane >> 4, li = lane & 15;

    int bid = blockIdx.x;
    if (p.xcd_remap) {      // bijective for any grid size: XCD x owns (q+1) tiles if x < r else q
        const int nb = gridDim.x, q = nb >> 3, r = nb & 7, xcd = bid & 7, loc = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int tile_m = bid / p.tiles_n, tile_n = bid % p.tiles_n;
    const int m0 = tile_m * K32_BM, n0 = tile_n * K32_BN;

    // staging: one DMA instruction per thread per operand per stage: thread -> (tile row tid >> 2, physical chunk tid & 3); a wave's 64 lanes
    // cover 16 rows x 64 bytes = 1 KiB contiguous in LDS
    const int srow = tid >> 2, sc = tid & 3;
    const f16* xsrc = p.X + (int64_t)min(m0 + srow, p.M - 1) * p.ldx + (sc ^ k32_fx(srow)) * 8;
    const f16* wsrc = p.W + (int64_t)min(n0 + srow, p.N - 1) * p.ldw + (sc ^ k32_fw(srow)) * 8;
    const uint32_t smem_lds = lds_addr_of(smem);
    auto glds = [&](int kt, int buf) {
        const uint32_t xs = smem_lds + (uint32_t)(buf * (XT + WT)) * 2u + (uint32_t)wid * 1024u;
        glds16(xsrc + (int64_t)kt * K32_BK, xs);
        glds16(wsrc + (int64_t)kt * K32_BK, xs + (uint32_t)XT * 2u);
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // fragment rows inside a tile (as gemm_nt.hip): X natural, W permuted so that a lane ends up with 16 consecutive n
    int xoff[4], woff[4];         // halfs inside a stage
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int xr = wm * 64 + 16 * t + li;
        const int wr = wn * 64 + 16 * (li >> 2) + 4 * t + (li & 3);
        xoff[t] = xr * K32_BK + ((g ^ k32_fx(xr)) << 3);
        woff[t] = XT + wr * K32_BK + ((g ^ k32_fw(wr)) << 3);
    }

    const int nk = p.K / K32_BK;
#pragma unroll
    for (int st = 0; st < K32_NS - 1; ++st) glds(min(st, nk - 1), st);
    int buf = 0, nbuf = K32_NS - 1;
    for (int kt = 0; kt < nk; ++kt) {
        // this wave's pieces of stage kt have landed (the two younger stages stay in flight); barrier: everybody's have, and everybody is
        // done with the fragments of stage kt-1, whose buffer takes stage kt+3 (k tiles past the end are clamped: constant count)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (K32_NS - 2)) : "memory");
        __builtin_amdgcn_s_barrier();
        const f16* st = smem + buf * (XT + WT);
        f16x8 xf[4], wf[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            xf[t] = ld8(st + xoff[t]);
            wf[t] = ld8(st + woff[t]);
        }
        __builtin_amdgcn_sched_barrier(0);
        glds(min(kt + K32_NS - 1, nk - 1), nbuf);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tm = 0; tm < 4; ++tm)
#pragma unroll
            for (int tn = 0; tn < 4; ++tn)
                acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[tn], xf[tm], acc[tm][tn], 0, 0, 0);
        buf = (buf + 1 == K32_NS) ? 0 : buf + 1;
        nbuf = (nbuf + 1 == K32_NS) ? 0 : nbuf + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the clamped tail reloads must land before the LDS is released

    // ---- epilogue (as gemm_nt.hip): lane owns row m (per tm) and 16 consecutive n --------------------
    const int ncol0 = n0 + wn * 64 + 16 * g;
    const bool full_n = (ncol0 + 16 <= p.N);
    float bias_v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) bias_v[j] = 0.f;
    if (p.bias) {
        if (full_n) {
            f16x8 b0 = ld8(p.bias + ncol0), b1 = ld8(p.bias + ncol0 + 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) { bias_v[j] = (float)b0[j]; bias_v[8 + j] = (float)b1[j]; }
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) if (ncol0 + j < p.N) bias_v[j] = (float)p.bias[ncol0 + j];
        }
    }
#pragma unroll
    for (int tm = 0; tm < 4; ++tm) {
        const int m = m0 + wm * 64 + 16 * tm + li;
        if (m >= p.M) continue;
        const uint32_t rkey = p.drop.thresh ? drop_rowkey(p.drop, nt_drop_row(p, m)) : 0u;   // dropout element = (row m, col n)
        float v[16];
#pragma unroll
        for (int tn = 0; tn < 4; ++tn)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[tn * 4 + r] = acc[tm][tn][r] * p.alpha + bias_v[tn * 4 + r];
#pragma unroll
        for (int h = 0; h < 2; ++h) nt_epilogue8<SG>(p, m, ncol0 + 8 * h, v + 8 * h, rkey, true);
    }
}

int vlp_gemm_nt_k32_launch(GemmNtParams& p, bool sg, hipStream_t s) {
    const size_t smem = (size_t)K32_NS * (K32_BM + K32_BN) * K32_BK * sizeof(f16);      // 128 KiB
    p.tiles_n = cdiv(p.N, K32_BN);
    const dim3 grid(cdiv(p.M, K32_BM) * p.tiles_n), block(K32_T);
    VLP_ONCE_PER_DEVICE((void)hipFuncSetAttribute((const void*)gemm_nt_k32_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    VLP_ONCE_PER_DEVICE((void)hipFuncSetAttribute((const void*)gemm_nt_k32_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    if (sg) hipLaunchKernelGGL(gemm_nt_k32_kernel<true>, grid, block, smem, s, p);
    else hipLaunchKernelGGL(gemm_nt_k32_kernel<false>, grid, block, smem, s, p);
    return VLP_OK;
}
