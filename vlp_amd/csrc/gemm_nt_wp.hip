// Wave-pipelined NT GEMM for gfx950:  Y[M,N] = epi( alpha * X[M,K] . W[N,K]^T ), same contract and epilogue as gemm_nt.hip
// (replaces the nn.Linear forwards / dgrads listed there: modeling.py:270-272, :314, :341, :354, :432, :481, :1003-1029).
//
// Why a third kernel family.  The rings of gemm_nt.hip give every wave a 64x64 output in arch VGPRs and run "barrier -> all fragment
// reads -> all MFMAs" per k tile: both waves of a SIMD are in the same phase, so the matrix pipe idles while the LDS serves 128 KB of
// fragment reads, and the refill DMA competes with those reads (profiles/r02_nt_loop_decomposition.txt: reads +0.17 us and DMA
// +0.2 us ADD to the 0.56 us of MFMAs per k tile).  Here the loop is software-pipelined INSIDE the wave:
//   * 256 x BN_T block tile (BN_T = 256 | 128), wave tiles of 128x128 / 128x64 (4 waves, one per SIMD, up to 256 fp32 accumulators
//     in the unified VGPR/AGPR file) or 128x64 / 64x64 (8 waves), built from v_mfma_f32_32x32x16_f16;
//   * a k tile (BK_T = 64 | 32) is walked in k16 steps; the fragments of step s+1 are read (ds_read_b128, double-buffered registers)
//     BETWEEN the MFMAs of step s, so the pipe never waits for the LDS and the reads are spread over the whole tile;
//   * one raw s_barrier per k tile, placed inside the last step: it publishes stage kt+1 (counted vmcnt: the younger stages stay in
//     flight) and frees slot kt, which is refilled with stage kt+NS by LDS-DMA (global_load_lds_dwordx4, SGPR base + 32-bit lane
//     offset, issued through inline asm so the compiler's waitcnt pass does not drain the queue -- common.h);
//   * DMA issue policy: SPREAD = 1 issues the whole stage right behind the barrier (needed with 2 stages), SPREAD = S spreads it over
//     the S k16 steps that follow; LOADER = true adds four DMA-only waves (one per SIMD) so the MFMA waves never wait on VMEM issue.
// The MFMA computes Y^T tiles (A operand = W rows, B operand = X rows): in the 32x32 C/D layout (col = lane & 31,
// row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)) a lane then owns ONE output row m per m-tile; with the W rows of the wave's slice
// visited in the order  n = 16 TN i2 + 16 tn + 4 (i >> 3) + (i & 3)  (i = A row, i2 = bit 2 of i) the TN tiles x 16 registers of a
// lane are 16 TN consecutive n: bias / residual / multiplier inputs are read and Y is written as 16-byte vectors from registers.
// LDS rows are BK_T halfs (128 | 64 bytes) with a 16-byte-chunk XOR swizzle on lane bits (4,3,1) | (4,3), applied on the DMA
// source address (the LDS destination of an LDS-DMA is lane-linear): every ds_read_b128 lane group touches 16 distinct slots.
#include <type_traits>

#include "common.h"
#include "gemm_nt.h"
#include "gemm_nt_epilogue.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int I, int N, class F>
DEVFN void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// LDS-DMA with a wave-uniform 64-bit base (SGPR pair) and a 32-bit per-lane byte offset
DEVFN void glds16_s(const char* sbase, uint32_t voff, uint32_t lds_dst) {
    const uint32_t m0v = __builtin_amdgcn_readfirstlane(lds_dst);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(m0v), "v"(voff), "s"(sbase) : "memory", "m0");
}
#define WP_SCHED() __builtin_amdgcn_sched_barrier(0)

template <int BK_T>
DEVFN int wp_swz(int r) {      // XOR applied to the 16-byte chunk index of LDS row r (both operands: bits 4,3,1 | 4,3 of the reading lane)
    if constexpr (BK_T == 64) return (((r >> 4) & 1) << 2) | (((r >> 3) & 1) << 1) | ((r >> 1) & 1);
    else return (r >> 3) & 3;
}
template <int BK_T>
DEVFN int wp_swz_w(int R) {    // the same lane bits seen from the W tile's row index (rows are visited in permuted order)
    if constexpr (BK_T == 64) return (R >> 1) & 7;
    else return (R >> 2) & 3;
}

// BN_T: columns of the block tile; WGM x WGN: compute waves; BK_T: k per stage; NS: ring slots; SPREAD: 1 | BK_T / 16; LOADER: 4 extra
// DMA-only waves; SG: save-grad GeLU epilogue; PB: MFMAs of the barrier step issued in front of the barrier
template <int BN_T, int WGM, int WGN, int BK_T, int NS, int SPREAD, bool LOADER, bool SG, int PB>
__global__ __launch_bounds__((WGM * WGN + (LOADER ? 4 : 0)) * 64, (WGM * WGN + (LOADER ? 4 : 0)) / 4) void gemm_nt_wp_kernel(GemmNtParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BM = 256;
    constexpr int NWC = WGM * WGN;                 // compute waves
    constexpr int NWL = LOADER ? 4 : NWC;          // waves that issue DMA
    constexpr int WROWS_M = BM / WGM, WROWS_N = BN_T / WGN;
    constexpr int TM = WROWS_M / 32, TN = WROWS_N / 32;
    constexpr int NM = TM * TN;                    // MFMAs per k16 step
    constexpr int NR = TM + TN;                    // fragment reads per k16 step
    constexpr int ROWB = BK_T * 2;                 // bytes per LDS row
    constexpr int CPR = BK_T / 8;                  // 16-byte chunks per row
    constexpr int RPB = 1024 / ROWB;               // rows per 1-KiB DMA piece
    constexpr int LPX = BM / RPB / NWL, LPW = BN_T / RPB / NWL, LPS = LPX + LPW;   // DMA pieces per issuing wave per stage
    constexpr int XBYTES = BM * ROWB, STAGE = (BM + BN_T) * ROWB;
    constexpr int S = BK_T / 16;                   // k16 steps per k tile
    static_assert(TN >= 2 && (BM / RPB) % NWL == 0 && (BN_T / RPB) % NWL == 0, "tile / wave geometry");
    static_assert(SPREAD == 1 || SPREAD == S, "SPREAD");
    static_assert((NS - 1) * LPS <= 63, "vmcnt range");
    static_assert(PB < NM, "PB");

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);

    int bid = blockIdx.x;
    if (p.xcd_remap) {      // bijective for any grid size: XCD x owns (q+1) tiles if x < r else q
        const int nb = gridDim.x, q = nb >> 3, r = nb & 7, xcd = bid & 7, loc = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int tile_m = bid / p.tiles_n, tile_n = bid % p.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN_T;
    const int nk = p.K / BK_T;
    const uint32_t lds0 = lds_addr_of(smem);
    const char* const xg = reinterpret_cast<const char*>(p.X);
    const char* const wg = reinterpret_cast<const char*>(p.W);

    // ---- DMA geometry: piece j of issuing wave lw = tile rows RPB*(lw + NWL*j) .. +RPB-1; lane -> (row lane / CPR, physical chunk lane % CPR)
    const bool issuer = LOADER ? (wid >= NWC) : true;
    const int lw = LOADER ? wid - NWC : wid;
    uint32_t voff[LPS];
    if (issuer) {
        const int rb = lane / CPR, pc = lane % CPR;
        static_for<0, LPS>([&](auto J) {
            constexpr int j = decltype(J)::value;
            if constexpr (j < LPX) {
                const int r = (lw + NWL * j) * RPB + rb;
                const int mr = min(m0 + r, p.M - 1);
                voff[j] = (uint32_t)mr * (uint32_t)p.ldx * 2u + (uint32_t)((pc ^ wp_swz<BK_T>(r)) << 4);
            } else {
                const int R = (lw + NWL * (j - LPX)) * RPB + rb;
                const int nr = min(n0 + R, p.N - 1);
                voff[j] = (uint32_t)nr * (uint32_t)p.ldw * 2u + (uint32_t)((pc ^ wp_swz_w<BK_T>(R)) << 4);
            }
        });
    }
    // pieces [J0, J1) of stage kt into ring slot `slot`
    auto issue = [&](auto J0_, auto J1_, int kt, int slot) {
        constexpr int J0 = decltype(J0_)::value, J1 = decltype(J1_)::value;
        const uint32_t dst = __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)slot * STAGE);
        const char* xs = xg + (int64_t)kt * ROWB;
        const char* ws = wg + (int64_t)kt * ROWB;
        static_for<J0, J1>([&](auto J) {
            constexpr int j = decltype(J)::value;
            if constexpr (j < LPX) glds16_s(xs, voff[j], dst + (uint32_t)(lw + NWL * j) * 1024u);
            else glds16_s(ws, voff[j], dst + (uint32_t)XBYTES + (uint32_t)(lw + NWL * (j - LPX)) * 1024u);
        });
    };
    using I0 = std::integral_constant<int, 0>;
    using ILPS = std::integral_constant<int, LPS>;

    // =====================================================================================================================
    // loader waves: issue, wait, barrier -- the same barrier sequence as the compute waves (1 + (nk - 1))
    // =====================================================================================================================
    if (LOADER && wid >= NWC) {
        const int pre = min(NS, nk);
        for (int st = 0; st < pre; ++st) issue(I0{}, ILPS{}, st, st);
        // stage 0 landed
        if (nk >= NS) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 1) * LPS) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        int slot = 0;
        for (int kt = 0; kt + 1 < nk; ++kt) {
            // barrier B_kt: stage kt+1 landed; outstanding stages kt+1 .. min(kt+NS-1, nk-1)
            const int outst = min(NS - 1, nk - 1 - kt);
            if (outst == NS - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * LPS) : "memory");
            else {
                bool done = false;
                static_for<1, NS - 1>([&](auto Jc) {
                    constexpr int jc = decltype(Jc)::value;
                    if (!done && outst == jc) { asm volatile("s_waitcnt vmcnt(%0)" ::"n"((jc - 1) * LPS) : "memory"); done = true; }
                });
            }
            __builtin_amdgcn_s_barrier();
            if (kt + NS < nk) issue(I0{}, ILPS{}, kt + NS, slot);
            slot = (slot + 1 == NS) ? 0 : slot + 1;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }

    // =====================================================================================================================
    // compute waves
    // =====================================================================================================================
    const int wm = wid / WGN, wn = wid % WGN;
    const int li = lane & 31, hi = lane >> 5;
    const int fsw = wp_swz<BK_T>(li);
    uint32_t xoff[S], woff[S];          // byte offsets inside a stage of this lane's fragment chunks, per k16 step
    {
        const int rowx = wm * WROWS_M + li;
        const int roww = wn * WROWS_N + 16 * TN * ((li >> 2) & 1) + 4 * (li >> 3) + (li & 3);
#pragma unroll
        for (int s = 0; s < S; ++s) {
            xoff[s] = (uint32_t)(rowx * ROWB + (((2 * s + hi) ^ fsw) << 4));
            woff[s] = (uint32_t)(XBYTES + roww * ROWB + (((2 * s + hi) ^ fsw) << 4));
        }
    }
    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    f16x8 xf[2][TM], wf[2][TN];

    // fragment read q of a step (order: x0, w0 .. w(TN-1), x1 .. x(TM-1): the first MFMAs' operands first) into register set BUF
    auto read_one = [&](auto BUF_, auto Q_, const char* stage, auto SS_) {
        constexpr int BUF = decltype(BUF_)::value, q = decltype(Q_)::value, ss = decltype(SS_)::value;
        if constexpr (q == 0) xf[BUF][0] = ld8(reinterpret_cast<const f16*>(stage + xoff[ss]));
        else if constexpr (q <= TN) wf[BUF][q - 1] = ld8(reinterpret_cast<const f16*>(stage + woff[ss] + (q - 1) * 16 * ROWB));
        else xf[BUF][q - TN] = ld8(reinterpret_cast<const f16*>(stage + xoff[ss] + (q - TN) * 32 * ROWB));
    };
    auto mfma_one = [&](auto BUF_, auto K_) {
        constexpr int BUF = decltype(BUF_)::value, k = decltype(K_)::value, tm = k / TN, tn = k % TN;
        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[BUF][tn], xf[BUF][tm], acc[tm][tn], 0, 0, 0);
    };
    // One k16 step.  MODE 0: regular (reads of the next step spread over all MFMAs); 1: barrier step (PB MFMAs, counted wait,
    // barrier, then the reads of the next tile's step 0 and -- for DMA-issuing compute waves -- pieces [G0, G1) of stage gkt);
    // 2: last step of the last tile (no reads).  WAITN: vmcnt immediate of the barrier step (-1: none, loader builds).
    auto step = [&](auto CUR_, auto MODE_, auto NEXTS_, auto WAITN_, auto G0_, auto G1_, const char* rstage, int gkt, int gslot, bool gdo) {
        constexpr int CUR = decltype(CUR_)::value, MODE = decltype(MODE_)::value, NEXTS = decltype(NEXTS_)::value, WAITN = decltype(WAITN_)::value;
        constexpr int G0 = decltype(G0_)::value, G1 = decltype(G1_)::value, NG = G1 - G0;
        constexpr int FIRST = (MODE == 1) ? PB : 0;          // first MFMA slot that may carry reads / DMA
        constexpr int SPAN = NM - FIRST;
        static_for<0, NM>([&](auto K_) {
            constexpr int k = decltype(K_)::value;
            if constexpr (MODE == 1 && k == PB) {
                WP_SCHED();
                if constexpr (WAITN >= 0) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(WAITN) : "memory");
                else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                WP_SCHED();
            }
            mfma_one(CUR_, K_);
            if constexpr (MODE != 2 && k >= FIRST) {
                static_for<0, NR>([&](auto Q_) {
                    constexpr int q = decltype(Q_)::value;
                    // regular step: spread over the step; barrier step: one per MFMA right behind the barrier (the loop back edge waits
                    // for all of them: lgkmcnt(0) at the loop head)
                    constexpr int rslot = (MODE == 1) ? ((FIRST + q < NM) ? FIRST + q : NM - 1) : (q * SPAN) / NR;
                    if constexpr (rslot == k) read_one(std::integral_constant<int, CUR ^ 1>{}, Q_, rstage, NEXTS_);
                });
            }
            if constexpr (NG > 0 && k >= FIRST) {
                static_for<0, NG>([&](auto G_) {
                    constexpr int g = decltype(G_)::value;
                    if constexpr (FIRST + (g * SPAN) / NG == k) {
                        if (gdo) issue(std::integral_constant<int, G0 + g>{}, std::integral_constant<int, G0 + g + 1>{}, gkt, gslot);
                    }
                });
            }
            WP_SCHED();
        });
    };
    using IM1 = std::integral_constant<int, -1>;
    constexpr bool CDMA = !LOADER;                 // compute waves issue the DMA
    // spread issue: group g of a stage = pieces [g LPS / SPREAD, (g + 1) LPS / SPREAD); group 0 goes behind the barrier

    // ---- prologue ---------------------------------------------------------------------------------------------------
    if (CDMA) {
        const int pre = min(NS, nk);
        for (int st = 0; st < pre; ++st) issue(I0{}, ILPS{}, st, st);
        if (nk >= NS) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 1) * LPS) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    static_for<0, NR>([&](auto Q_) { read_one(I0{}, Q_, smem, I0{}); });

    // One k tile (slot cs, next slot ns).  J = stages behind this tile that are (or will be) in the ring: min(NS-1, nk-1-kt);
    // ISSUE: stage kt+NS exists and is issued at this tile's barrier.
    auto tile = [&](auto J_, auto ISSUE_, int kt, int cs, int ns, bool prev_spread) {
        constexpr int J = decltype(J_)::value;
        constexpr bool ISSUE = decltype(ISSUE_)::value;
        const char* cstage = smem + cs * STAGE;
        const char* nstage = smem + ns * STAGE;
        const int pslot = __builtin_amdgcn_readfirstlane((cs == 0) ? NS - 1 : cs - 1);   // slot of tile kt-1: target of a spread issue still in progress
        static_for<0, S>([&](auto SS_) {
            constexpr int ss = decltype(SS_)::value;
            using CUR = std::integral_constant<int, ss & 1>;
            if constexpr (ss < S - 1) {
                // regular step; with SPREAD = S it carries group ss+1 of the stage issued behind the previous barrier
                if constexpr (CDMA && SPREAD > 1)
                    step(CUR{}, I0{}, std::integral_constant<int, ss + 1>{}, IM1{}, std::integral_constant<int, (ss + 1) * LPS / SPREAD>{},
                         std::integral_constant<int, (ss + 2) * LPS / SPREAD>{}, cstage, kt - 1 + NS, pslot, prev_spread);
                else
                    step(CUR{}, I0{}, std::integral_constant<int, ss + 1>{}, IM1{}, I0{}, I0{}, cstage, 0, 0, false);
            } else if constexpr (J == 0) {
                step(CUR{}, std::integral_constant<int, 2>{}, I0{}, IM1{}, I0{}, I0{}, cstage, 0, 0, false);
            } else {
                constexpr int WAITN = CDMA ? (J - 1) * LPS : -1;
                if constexpr (CDMA && ISSUE)
                    step(CUR{}, std::integral_constant<int, 1>{}, I0{}, std::integral_constant<int, WAITN>{}, I0{},
                         std::integral_constant<int, LPS / SPREAD>{}, nstage, kt + NS, cs, true);
                else
                    step(CUR{}, std::integral_constant<int, 1>{}, I0{}, std::integral_constant<int, WAITN>{}, I0{}, I0{}, nstage, 0, 0, false);
            }
        });
    };
    int kt = 0, cs = 0;
    bool prev_spread = false;
    for (; kt + NS < nk; ++kt) {
        const int ns = (cs + 1 == NS) ? 0 : cs + 1;
        tile(std::integral_constant<int, NS - 1>{}, std::true_type{}, kt, cs, ns, prev_spread);
        prev_spread = true;
        cs = ns;
    }
    static_for<0, NS>([&](auto JJ_) {
        constexpr int J = NS - 1 - decltype(JJ_)::value;
        if (nk - 1 - kt == J) {
            const int ns = (cs + 1 == NS) ? 0 : cs + 1;
            tile(std::integral_constant<int, J>{}, std::false_type{}, kt, cs, ns, prev_spread);
            prev_spread = false;
            cs = ns;
            ++kt;
        }
    });

    // ---- epilogue: lane owns row m per m-tile and 16 TN consecutive n ---------------------------------------------------
    const int nbase = n0 + wn * WROWS_N + 16 * TN * hi;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
        const int m = m0 + wm * WROWS_M + 32 * tm + li;
        if (m >= p.M) continue;
        const uint32_t rkey = p.drop.thresh ? drop_rowkey(p.drop, (uint64_t)m) : 0u;
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = acc[tm][tn][8 * h + j] * p.alpha;
                nt_epilogue8<SG, true>(p, m, nbase + 16 * tn + 8 * h, v, rkey);
            }
    }
}

// cfg (vlp_gemm_nt variant 64 + cfg, + 8 = XCD-aware tile order):
//   0: 256x256, 4 waves (128x128), BK 64, 2 slots, burst DMA          1: 256x256, 8 waves (128x64), BK 64, 2 slots, burst
//   2: 256x256, 4 waves, BK 32, 4 slots, spread DMA                    3: 256x256, 8 waves, BK 32, 4 slots, spread
//   4: 256x128, 4 waves (128x64), BK 64, 3 slots, spread               5: 256x128, 8 waves (64x64), BK 64, 3 slots, spread
//   6: 256x128, 4 compute waves (128x64) + 4 DMA waves, BK 64, 3 slots 7: 256x128, 4 waves, BK 64, 3 slots, burst
int vlp_gemm_nt_wp_launch(GemmNtParams& p, int cfg, bool sg, hipStream_t s) {
    VLP_CHECK_ARG(sg || nt_epilogue_is_light(p), "vlp_gemm_nt: the wave-pipelined variants carry bias / ReLU / multiplier / dropout / residual / save-grad GeLU epilogues only");
    VLP_CHECK_ARG((int64_t)p.M * p.ldx < (1ll << 31) && (int64_t)p.N * p.ldw < (1ll << 31), "vlp_gemm_nt: wave-pipelined variants need M*ldx, N*ldw < 2^31");
#define LAUNCH_WP_(BNT, WGM, WGN, BKT, NSV, SPR, LDR, SGV, PBV)                                                                            \
    do {                                                                                                                               \
        const size_t smem = (size_t)(NSV) * (256 + (BNT)) * (BKT) * sizeof(f16);                                                       \
        auto kfn = gemm_nt_wp_kernel<BNT, WGM, WGN, BKT, NSV, SPR, LDR, SGV, PBV>;                                                     \
        static bool attr = false;                                                                                                      \
        if (!attr) { (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); attr = true; } \
        p.tiles_n = cdiv(p.N, (BNT));                                                                                                  \
        hipLaunchKernelGGL(kfn, dim3(cdiv(p.M, 256) * p.tiles_n), dim3(((WGM) * (WGN) + ((LDR) ? 4 : 0)) * 64), smem, s, p);          \
    } while (0)
#define LAUNCH_WP(BNT, WGM, WGN, BKT, NSV, SPR, LDR, PBV) \
    do { if (sg) LAUNCH_WP_(BNT, WGM, WGN, BKT, NSV, SPR, LDR, true, PBV); else LAUNCH_WP_(BNT, WGM, WGN, BKT, NSV, SPR, LDR, false, PBV); } while (0)
    switch (cfg) {
        case 0: LAUNCH_WP(256, 2, 2, 64, 2, 1, false, 4); break;
        case 1: LAUNCH_WP(256, 2, 4, 64, 2, 1, false, 2); break;
        case 2: LAUNCH_WP(256, 2, 2, 32, 4, 2, false, 4); break;
        case 3: LAUNCH_WP(256, 2, 4, 32, 4, 2, false, 2); break;
        case 4: LAUNCH_WP(128, 2, 2, 64, 3, 4, false, 2); break;
        case 5: LAUNCH_WP(128, 4, 2, 64, 3, 4, false, 0); break;
        case 6: LAUNCH_WP(128, 2, 2, 64, 3, 1, true, 2); break;
        default: LAUNCH_WP(128, 2, 2, 64, 3, 1, false, 2); break;
    }
#undef LAUNCH_WP
#undef LAUNCH_WP_
    return VLP_OK;
}
