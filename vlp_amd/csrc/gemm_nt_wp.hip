// Wave-pipelined NT GEMM for gfx950:  Y[M,N] = epi( alpha * X[M,K] . W[N,K]^T ), same contract as gemm_nt.hip
// (replaces the nn.Linear forwards / dgrads listed there: modeling.py:270-272, :314, :341, :354, :432, :481, :1003-1029).
//
// Why a third kernel family.  The rings of gemm_nt.hip give every wave a 64x64 output in arch VGPRs and run "barrier -> all fragment
// reads -> all MFMAs" per k tile.  Here the loop is software-pipelined INSIDE the wave and the ring is shaped by what round 3 measured
// (profiles/r03_nt_wp_decomposition.txt, tools/dma_path_bench.hip):
//   * 256 x BN_T block tile (BN_T = 256 | 128), k tiles of 64, wave tiles of 128x128 / 128x64 (4 waves, one per SIMD, up to 256 fp32
//     accumulators in AGPRs) or 128x64 / 64x64 (8 waves), built from v_mfma_f32_32x32x16_f16;
//   * a k tile is walked in four k16 steps; the fragments of step s+1 are read (ds_read_b128, double-buffered registers) BETWEEN the
//     MFMAs of step s, one raw s_barrier per k tile sits inside the last step (counted vmcnt: the younger stages stay in flight);
//   * the staging side, not the matrix pipe, bounds these GEMMs: a CU streams cold operands at (bytes in flight) / (1.3 - 1.8 us), the
//     same through LDS-DMA or through VGPRs.  The X stream (activations: HBM / Infinity Cache) is the slow one, W hits L2.  ROLES = true
//     therefore gives X and W their OWN rings -- X deep (NSX slots), W shallow (NSW = 2) inside the same 160 KiB -- and lets half of the
//     waves issue only X pieces and the other half only W pieces: vmcnt is per wave and in order, so a wave that mixed the two streams
//     would have to drain its deep X stages every time it waits for the young W stage.
// The MFMA computes Y^T tiles (A operand = W rows, B operand = X rows): in the 32x32 C/D layout (col = lane & 31,
// row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)) a lane owns ONE output row per m-tile and, with the W rows of the wave's slice
// visited in the order  n = 16 TN i2 + 16 tn + 4 (i >> 3) + (i & 3)  (i = A row, i2 = bit 2 of i), 16 TN consecutive n.
// LDS rows are 128 bytes with a 16-byte-chunk XOR swizzle on lane bits (4,3,1), applied on the DMA source address (the LDS
// destination of an LDS-DMA is lane-linear): every ds_read_b128 lane group touches 16 distinct slots.
// Epilogue: each wave transposes its tile through LDS (fp32, one 32-row m-tile at a time) so that 4 TN lanes cover one row: residual /
// multiplier loads and the Y store are whole 128-byte lines (in the accumulator layout a store touched 64 rows x 16 bytes: 25-29 us).
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
template <int V> using IC = std::integral_constant<int, V>;

// LDS-DMA with a wave-uniform 64-bit base (SGPR pair) and a 32-bit per-lane byte offset
DEVFN void glds16_s(const char* sbase, uint32_t voff, uint32_t lds_dst) {
    const uint32_t m0v = __builtin_amdgcn_readfirstlane(lds_dst);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(m0v), "v"(voff), "s"(sbase) : "memory", "m0");
}
#define WP_SCHED() __builtin_amdgcn_sched_barrier(0)
// investigation builds (tools/build_wp_dbg.sh): WP_DBG bit 0 = no MFMAs, 1 = no DMA, 2 = no epilogue, 3 = no fragment reads,
// 4 = fragment reads issued (asm, never waited for except at the barrier) while the MFMAs run on constant fragments: issue cost without latency
#ifndef WP_DBG
#define WP_DBG 0
#endif

DEVFN int wp_swz(int r) { return (((r >> 4) & 1) << 2) | (((r >> 3) & 1) << 1) | ((r >> 1) & 1); }   // X tile row -> chunk XOR (lane bits 4,3,1)
DEVFN int wp_swz_w(int R) { return (R >> 1) & 7; }                                                    // the same lane bits seen from the permuted W row

// light epilogue of 8 consecutive columns of row m, same order of operations as nt_epilogue8: pre-activation store, ReLU, multiplier,
// dropout, residual; vv = alpha * acc + bias on entry
DEVFN void wp_light_store8(const GemmNtParams& p, int m, int nc, bool ragged, float (&vv)[8], const f16x8& mulv, const f16x8& resv) {
    if (p.preact) {
        f16x8 z;
#pragma unroll
        for (int j = 0; j < 8; ++j) z[j] = (nc + j < p.N) ? (f16)vv[j] : (f16)0.f;
        st8_out<VLP_SS_SAVED>(p.preact + (int64_t)m * p.ldp + nc, z);
#pragma unroll
        for (int j = 0; j < 8; ++j) vv[j] = (float)z[j];
    }
    if (p.act == VLP_ACT_RELU) {
#pragma unroll
        for (int j = 0; j < 8; ++j) vv[j] = fmaxf(vv[j], 0.f);
    }
    if (p.mulmode == VLP_MUL_PLAIN) {
#pragma unroll
        for (int j = 0; j < 8; ++j) vv[j] *= (float)mulv[j];
    } else if (p.mulmode != VLP_MUL_NONE) {           // VLP_MUL_RELU_MASK (GELU_GRAD is refused by the launcher)
#pragma unroll
        for (int j = 0; j < 8; ++j) vv[j] = ((float)mulv[j] > 0.f) ? vv[j] : 0.f;
    }
    if (p.drop.thresh) drop_mult8(p.drop, drop_rowkey(p.drop, nt_drop_row(p, m)), (uint32_t)nc, vv);
    if (p.residual) {
#pragma unroll
        for (int j = 0; j < 8; ++j) vv[j] += (float)resv[j];
    }
    f16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (!ragged || nc + j < p.N) ? (f16)vv[j] : (f16)0.f;
    st8_out<VLP_SS_NT>(p.Y + (int64_t)m * p.ldy + nc, o);
}

// save-grad GeLU epilogue of 8 columns: z = fp16-rounded pre-activation; y = gelu(z); preact <- gelu'(z)  (as nt_epilogue8<true>)
DEVFN void wp_sg_store8(const GemmNtParams& p, int m, int nc, bool ragged, const float (&vv)[8]) {
    f16x8 d, o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float gl, gp;
        gelu_and_grad_f((float)(f16)vv[j], gl, gp);
        o[j] = (f16)gl;
        d[j] = (f16)gp;
    }
    if (ragged) {
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (nc + j >= p.N) { o[j] = (f16)0.f; d[j] = (f16)0.f; }
    }
    st8_out<VLP_SS_SAVED>(p.preact + (int64_t)m * p.ldp + nc, d);
    st8_out<VLP_SS_NT>(p.Y + (int64_t)m * p.ldy + nc, o);
}

// what a step may issue: a group of X stage kx into X slot sx (if dox), a group of W stage kw into W slot sw (if dow)
struct WpDma { int kx, sx, kw, sw; bool dox, dow; };

// BN_T: columns of the block tile; WGM x WGN: waves; NSX / NSW: slots of the X / W ring; ROLES: waves 0 .. NW/2-1 issue the X pieces,
// the others the W pieces (needed when NSX != NSW); SPREAD: rings with >= 3 slots issue a stage over the four k16 steps behind the
// barrier instead of right behind it; SG: save-grad GeLU epilogue; PB: MFMAs of the barrier step issued in front of the barrier;
// LEAD: fragments are read LEAD k16 steps ahead of their MFMAs (2 LEAD register sets; the barrier sits in step S - LEAD)
template <int BN_T, int WGM, int WGN, int NSX, int NSW, bool ROLES, bool SPREAD, bool SG, int PB, int LEAD>
__global__ __launch_bounds__(WGM * WGN * 64, WGM * WGN / 4) void gemm_nt_wp_kernel(GemmNtParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BM = 256, BK_T = 64, S = 4;
    constexpr int NW = WGM * WGN;
    constexpr int WROWS_M = BM / WGM, WROWS_N = BN_T / WGN;
    constexpr int TM = WROWS_M / 32, TN = WROWS_N / 32;
    constexpr int NM = TM * TN;                    // MFMAs per k16 step
    constexpr int NR = TM + TN;                    // fragment reads per k16 step
    constexpr int ROWB = BK_T * 2;                 // bytes per LDS row
    constexpr int NIS = ROLES ? NW / 2 : NW;       // waves issuing each operand's pieces
    constexpr int LPX = BM / 8 / NIS, LPW = BN_T / 8 / NIS;   // 1-KiB pieces (8 rows) per issuing wave per stage
    constexpr int XBYTES = BM * ROWB, WBYTES = BN_T * ROWB;
    constexpr int WRING = NSX * XBYTES;            // LDS offset of the W ring
    constexpr int NSMAX = NSX > NSW ? NSX : NSW;
    constexpr bool SPX = SPREAD && NSX >= 3, SPW = SPREAD && NSW >= 3;
    static_assert(ROLES || NSX == NSW, "one wave mixing both streams needs equal ring depths");
    static_assert(TN >= 2 && (BM / 8) % NIS == 0 && (BN_T / 8) % NIS == 0, "tile / wave geometry");
    static_assert((NSX - 1) * LPX + (ROLES ? 0 : (NSW - 1) * LPW) <= 63 && (NSW - 1) * LPW <= 63, "vmcnt range");
    static_assert(PB < NM, "PB");
    constexpr int NB = 2 * LEAD;                   // fragment register sets
    static_assert(LEAD == 1 || LEAD == 2, "LEAD");

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);

    // tile t of the launch -> origin (XCD-aware order: bijective for any tile count, XCD x owns (q+1) tiles if x < r else q)
    auto tile_origin = [&](int t, int& mo, int& no) {
        if (p.xcd_remap) {
            const int nb = p.tiles_total, q = nb >> 3, r = nb & 7, xcd = t & 7, loc = t >> 3;
            t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
        }
        mo = (t / p.tiles_n) * BM;
        no = (t % p.tiles_n) * BN_T;
    };
    int m0, n0;
    tile_origin(blockIdx.x, m0, n0);
    const int nk = p.K / BK_T;
    const uint32_t lds0 = lds_addr_of(smem);
    const char* const xg = reinterpret_cast<const char*>(p.X);
    const char* const wg = reinterpret_cast<const char*>(p.W);

    // ---- DMA geometry: piece j of issuing wave lw = tile rows 8 (lw + NIS j) .. +7; lane -> (row lane >> 3, physical chunk lane & 7)
    const bool xrole = ROLES ? (wid < NW / 2) : true;
    const int lw = ROLES ? (xrole ? wid : wid - NW / 2) : wid;
    uint32_t voffx[LPX], voffw[LPW];
    auto set_dma_tile = [&](int mo, int no) {
        const int rb = lane >> 3, pc = lane & 7;
        static_for<0, LPX>([&](auto J) {
            constexpr int j = decltype(J)::value;
            const int r = (lw + NIS * j) * 8 + rb;
            voffx[j] = (uint32_t)min(mo + r, p.M - 1) * (uint32_t)p.ldx * 2u + (uint32_t)((pc ^ wp_swz(r)) << 4);
        });
        static_for<0, LPW>([&](auto J) {
            constexpr int j = decltype(J)::value;
            const int R = (lw + NIS * j) * 8 + rb;
            voffw[j] = (uint32_t)min(no + R, p.N - 1) * (uint32_t)p.ldw * 2u + (uint32_t)((pc ^ wp_swz_w(R)) << 4);
        });
    };
    set_dma_tile(m0, n0);
    // X pieces [J0, J1) of stage kt into X ring slot `slot`; W likewise
    auto issue_x = [&](auto J0_, auto J1_, int kt, int slot) {
        if constexpr (WP_DBG & 2) return;
        const uint32_t dst = __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)slot * XBYTES);
        const char* xs = xg + (int64_t)kt * ROWB;
        static_for<decltype(J0_)::value, decltype(J1_)::value>([&](auto J) {
            constexpr int j = decltype(J)::value;
            glds16_s(xs, voffx[j], dst + (uint32_t)(lw + NIS * j) * 1024u);
        });
    };
    auto issue_w = [&](auto J0_, auto J1_, int kt, int slot) {
        if constexpr (WP_DBG & 2) return;
        const uint32_t dst = __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)WRING + (uint32_t)slot * WBYTES);
        const char* ws = wg + (int64_t)kt * ROWB;
        static_for<decltype(J0_)::value, decltype(J1_)::value>([&](auto J) {
            constexpr int j = decltype(J)::value;
            glds16_s(ws, voffw[j], dst + (uint32_t)(lw + NIS * j) * 1024u);
        });
    };

    // ---- fragment geometry ----------------------------------------------------------------------------------------------
    const int wm = wid / WGN, wn = wid % WGN;
    const int li = lane & 31, hi = lane >> 5;
    const int fsw = wp_swz(li);
    uint32_t xoff[S], woff[S];          // byte offsets inside a ring slot of this lane's fragment chunks, per k16 step
    {
        const int rowx = wm * WROWS_M + li;
        const int roww = wn * WROWS_N + 16 * TN * ((li >> 2) & 1) + 4 * (li >> 3) + (li & 3);
#pragma unroll
        for (int s = 0; s < S; ++s) {
            xoff[s] = (uint32_t)(rowx * ROWB + (((2 * s + hi) ^ fsw) << 4));
            woff[s] = (uint32_t)(WRING + roww * ROWB + (((2 * s + hi) ^ fsw) << 4));
        }
    }
    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    constexpr int CW = 32 * TN;                   // columns of the wave tile
    f16x8 xf[NB][TM], wf[NB][TN];
    if constexpr (WP_DBG & 24) {
#pragma unroll
        for (int b = 0; b < NB; ++b) {
#pragma unroll
            for (int t = 0; t < TM; ++t) xf[b][t] = __builtin_bit_cast(f16x8, (u32x4){(uint32_t)lane, 1u, 2u, 3u});
#pragma unroll
            for (int t = 0; t < TN; ++t) wf[b][t] = __builtin_bit_cast(f16x8, (u32x4){(uint32_t)lane, 5u, 6u, 7u});
        }
    }

    // fragment read q of a step (order: x0, w0 .. w(TN-1), x1 .. x(TM-1): the first MFMAs' operands first) into register set BUF;
    // xs / ws = byte offsets of the X / W ring slot that holds the k tile
    auto read_one = [&](auto BUF_, auto Q_, int xs, int ws, auto SS_) {
        constexpr int BUF = decltype(BUF_)::value, q = decltype(Q_)::value, ss = decltype(SS_)::value;
        if constexpr (WP_DBG & 8) return;
        if constexpr (WP_DBG & 16) {
            const uint32_t a = lds0 + (q == 0 ? (uint32_t)xs + xoff[ss] : (q <= TN ? (uint32_t)ws + woff[ss] + (q - 1) * 16 * ROWB : (uint32_t)xs + xoff[ss] + (q - TN) * 32 * ROWB));
            u32x4 junk;
            asm volatile("ds_read_b128 %0, %1" : "=v"(junk) : "v"(a));
            return;
        }
        if constexpr (q == 0) xf[BUF][0] = ld8(reinterpret_cast<const f16*>(smem + xs + xoff[ss]));
        else if constexpr (q <= TN) wf[BUF][q - 1] = ld8(reinterpret_cast<const f16*>(smem + ws + woff[ss] + (q - 1) * 16 * ROWB));
        else xf[BUF][q - TN] = ld8(reinterpret_cast<const f16*>(smem + xs + xoff[ss] + (q - TN) * 32 * ROWB));
    };
    auto mfma_one = [&](auto BUF_, auto K_) {
        constexpr int BUF = decltype(BUF_)::value, k = decltype(K_)::value, tm = k / TN, tn = k % TN;
        if constexpr (WP_DBG & 1) { const f16x8 a_ = wf[BUF][tn], b_ = xf[BUF][tm]; asm volatile("" ::"v"(a_), "v"(b_)); }
        else acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[BUF][tn], xf[BUF][tm], acc[tm][tn], 0, 0, 0);
    };
    // One k16 step.  MODE 0: regular (reads of the next step spread over all MFMAs); 1: barrier step (PB MFMAs, counted wait, barrier,
    // then the reads of the next tile's step 0); 2: last step of the last tile (no reads).  WX / WW: vmcnt immediates of the barrier
    // step for X- / W-issuing waves (one wave issuing both: WX counts both).  GX / GW: DMA group (0 .. S-1) carried, -1 none.
    auto step = [&](auto CUR_, auto DST_, auto MODE_, auto NEXTS_, auto WX_, auto WW_, auto GX_, auto GW_, int rxs, int rws, const WpDma& d) {
        constexpr int CUR = decltype(CUR_)::value, MODE = decltype(MODE_)::value, WX = decltype(WX_)::value, WW = decltype(WW_)::value;
        constexpr int GX = decltype(GX_)::value, GW = decltype(GW_)::value;
        constexpr int FIRST = (MODE == 1) ? PB : 0;          // first MFMA slot that may carry reads / DMA
        constexpr int SPAN = NM - FIRST;
        // pieces of the groups: a burst ring issues everything in group 0
        constexpr int X0 = GX < 0 ? 0 : (SPX ? GX * LPX / S : 0), X1 = GX < 0 ? 0 : (SPX ? (GX + 1) * LPX / S : (GX == 0 ? LPX : 0));
        constexpr int W0 = GW < 0 ? 0 : (SPW ? GW * LPW / S : 0), W1 = GW < 0 ? 0 : (SPW ? (GW + 1) * LPW / S : (GW == 0 ? LPW : 0));
        constexpr int NGX = X1 - X0, NGW = W1 - W0, NGB = NGX + NGW;
        static_for<0, NM>([&](auto K_) {
            constexpr int k = decltype(K_)::value;
            if constexpr (MODE == 1 && k == PB) {
                WP_SCHED();
                if (ROLES && !xrole) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(WW) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(WX) : "memory");
                __builtin_amdgcn_s_barrier();
                WP_SCHED();
            }
            mfma_one(CUR_, K_);
            if constexpr (MODE != 2 && k >= FIRST) {
                static_for<0, NR>([&](auto Q_) {
                    constexpr int q = decltype(Q_)::value;
                    // regular step: spread over the step; barrier step: one per MFMA right behind the barrier (the loop back edge
                    // waits for all of them: lgkmcnt(0) at the loop head)
                    constexpr int rslot = (MODE == 1) ? ((FIRST + q < NM) ? FIRST + q : NM - 1) : (q * SPAN) / NR;
                    if constexpr (rslot == k) read_one(DST_, Q_, rxs, rws, NEXTS_);
                });
            }
            if constexpr (k >= FIRST) {
                if constexpr (ROLES) {
                    if (xrole) {
                        static_for<0, NGX>([&](auto G_) {
                            constexpr int g = decltype(G_)::value;
                            if constexpr (FIRST + (g * SPAN) / (NGX > 0 ? NGX : 1) == k) { if (d.dox) issue_x(IC<X0 + g>{}, IC<X0 + g + 1>{}, d.kx, d.sx); }
                        });
                    } else {
                        static_for<0, NGW>([&](auto G_) {
                            constexpr int g = decltype(G_)::value;
                            if constexpr (FIRST + (g * SPAN) / (NGW > 0 ? NGW : 1) == k) { if (d.dow) issue_w(IC<W0 + g>{}, IC<W0 + g + 1>{}, d.kw, d.sw); }
                        });
                    }
                } else {
                    static_for<0, NGB>([&](auto G_) {          // X pieces first, then W pieces, spread over the slots together
                        constexpr int g = decltype(G_)::value;
                        if constexpr (FIRST + (g * SPAN) / (NGB > 0 ? NGB : 1) == k) {
                            if constexpr (g < NGX) { if (d.dox) issue_x(IC<X0 + g>{}, IC<X0 + g + 1>{}, d.kx, d.sx); }
                            else { if (d.dow) issue_w(IC<W0 + g - NGX>{}, IC<W0 + g - NGX + 1>{}, d.kw, d.sw); }
                        }
                    });
                }
            }
            WP_SCHED();
        });
    };

    // ---- prologue: fill both rings, wait for stage 0 ------------------------------------------------------------------------
    if constexpr (ROLES) {
        if (xrole) { for (int st = 0; st < min(NSX, nk); ++st) issue_x(IC<0>{}, IC<LPX>{}, st, st); }
        else { for (int st = 0; st < min(NSW, nk); ++st) issue_w(IC<0>{}, IC<LPW>{}, st, st); }
        if (nk >= NSMAX) {
            if (xrole) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSX - 1) * LPX) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSW - 1) * LPW) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    } else {
        for (int st = 0; st < min(NSX, nk); ++st) { issue_x(IC<0>{}, IC<LPX>{}, st, st); issue_w(IC<0>{}, IC<LPW>{}, st, st); }   // stage by stage: the in-order count
        if (nk >= NSMAX) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSX - 1) * (LPX + LPW)) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    static_for<0, LEAD>([&](auto L_) { static_for<0, NR>([&](auto Q_) { read_one(L_, Q_, 0, 0, L_); }); });

    // One k tile.  J = min(NSMAX, nk - 1 - kt): k tiles behind this one (NSMAX = "at least NSMAX": the main loop).
    // csx / csw: ring slots of this tile; the spread groups 1 .. S-1 of the stages issued behind the PREVIOUS barrier ride on steps 0 .. S-2.
    auto tile = [&](auto J_, int kt, int csx, int csw, bool prevx, bool prevw) {
        constexpr int J = decltype(J_)::value;
        constexpr int OX = (J < NSX - 1) ? J : NSX - 1, OW = (J < NSW - 1) ? J : NSW - 1;       // stages of each ring behind this tile
        constexpr bool ISX = J >= NSX, ISW = J >= NSW;                                          // stage kt + NSX / kt + NSW exists
        // vmcnt at the barrier: everything but the (O - 1) youngest stages of the wave's stream(s) has landed
        constexpr int WX = ROLES ? (OX > 0 ? (OX - 1) * LPX : 0) : (OX > 0 ? (OX - 1) * (LPX + LPW) : 0);
        constexpr int WW = OW > 0 ? (OW - 1) * LPW : 0;
        const int nsx = (csx + 1 == NSX) ? 0 : csx + 1, nsw = (csw + 1 == NSW) ? 0 : csw + 1;
        const int psx = __builtin_amdgcn_readfirstlane((csx == 0) ? NSX - 1 : csx - 1), psw = __builtin_amdgcn_readfirstlane((csw == 0) ? NSW - 1 : csw - 1);
        const WpDma dprev{kt - 1 + NSX, psx, kt - 1 + NSW, psw, prevx, prevw};
        const WpDma dnew{kt + NSX, csx, kt + NSW, csw, ISX, ISW};
        const WpDma dnone{0, 0, 0, 0, false, false};
        const int cxo = csx * XBYTES, cwo = csw * WBYTES, nxo = nsx * XBYTES, nwo = nsw * WBYTES;
        static_for<0, S>([&](auto SS_) {
            constexpr int ss = decltype(SS_)::value;
            using CUR = IC<(ss % NB)>;
            using DST = IC<((ss + LEAD) % NB)>;
            constexpr int BS = S - LEAD;                       // the barrier step
            if constexpr (ss < BS) {
                // regular step reading step ss + LEAD of this tile; with spread rings it carries group ss + LEAD of the stage issued
                // behind the previous tile's barrier
                step(CUR{}, DST{}, IC<0>{}, IC<ss + LEAD>{}, IC<0>{}, IC<0>{}, IC<(SPX ? ss + LEAD : -1)>{}, IC<(SPW ? ss + LEAD : -1)>{}, cxo, cwo, dprev);
            } else if constexpr (J == 0) {
                step(CUR{}, DST{}, IC<2>{}, IC<0>{}, IC<0>{}, IC<0>{}, IC<-1>{}, IC<-1>{}, cxo, cwo, dnone);
            } else if constexpr (ss == BS) {
                step(CUR{}, DST{}, IC<1>{}, IC<0>{}, IC<WX>{}, IC<WW>{}, IC<(ISX ? 0 : -1)>{}, IC<(ISW ? 0 : -1)>{}, nxo, nwo, dnew);
            } else {
                // behind the barrier: reads of the next tile's step ss - BS, groups ss - BS of the stage issued at this tile's barrier
                step(CUR{}, DST{}, IC<0>{}, IC<ss - BS>{}, IC<0>{}, IC<0>{}, IC<((ISX && SPX) ? ss - BS : -1)>{}, IC<((ISW && SPW) ? ss - BS : -1)>{}, nxo, nwo, dnew);
            }
        });
    };
    int kt = 0, csx = 0, csw = 0;
    bool prevx = false, prevw = false;
    for (; kt + NSMAX < nk; ++kt) {
        tile(IC<NSMAX>{}, kt, csx, csw, prevx, prevw);
        prevx = prevw = true;
        csx = (csx + 1 == NSX) ? 0 : csx + 1;
        csw = (csw + 1 == NSW) ? 0 : csw + 1;
    }
    static_for<0, NSMAX>([&](auto JJ_) {
        constexpr int J = NSMAX - 1 - decltype(JJ_)::value;
        if (nk - 1 - kt == J) {
            tile(IC<J>{}, kt, csx, csw, prevx, prevw);
            prevx = J >= NSX;                 // a stage issued behind this tile's barrier continues on the next tile's steps
            prevw = J >= NSW;
            csx = (csx + 1 == NSX) ? 0 : csx + 1;
            csw = (csw + 1 == NSW) ? 0 : csw + 1;
            ++kt;
        }
    });

    // ---- epilogue --------------------------------------------------------------------------------------------------------
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();             // every wave is done with the stage buffers
    if constexpr (WP_DBG & 4) { if (acc[0][0][0] != 12345.678f) return; }
    constexpr int RS = CW * 4 + 16;           // padded row stride of the transpose buffer (bytes): conflict-free ds_write_b128 of 8 rows
    constexpr int LPR = CW / 8, RPP = 64 / LPR, NPASS = (32 + RPP - 1) / RPP;
    constexpr bool EVEN = (64 % LPR == 0) && (32 % RPP == 0);     // CW = 96: 12 lanes per row, 5 rows per pass, 4 idle lanes
    char* const ebuf = smem + wid * (32 * RS);
    const int er = lane / LPR, ec = (lane % LPR) * 8;
    const int nc = n0 + wn * WROWS_N + ec;    // this lane's 8 columns (the same for every row it handles)
    const bool ncol_ok = nc < p.N && (EVEN || er < RPP);
    float bias_v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) bias_v[j] = 0.f;
    if (p.bias && ncol_ok) {
        if (nc + 8 <= p.N) {
            const f16x8 b = ld8(p.bias + nc);
#pragma unroll
            for (int j = 0; j < 8; ++j) bias_v[j] = (float)b[j];
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) if (nc + j < p.N) bias_v[j] = (float)p.bias[nc + j];
        }
    }
    const bool ragged = (p.N & 7) && nc + 8 > p.N;      // last, partial vector of a row: pad columns are written as zero
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
        // accumulators -> transpose buffer (row li, columns 16 TN hi + 16 tn + r)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 t;
#pragma unroll
                for (int j = 0; j < 4; ++j) t[j] = acc[tm][tn][4 * q + j];
                *reinterpret_cast<f32x4*>(ebuf + li * RS + (16 * TN * hi + 16 * tn + 4 * q) * 4) = t;
            }
        const int mb = m0 + wm * WROWS_M + 32 * tm + er;       // row of pass 0
        // batched operand loads (whole lines per row): the latency of all passes overlaps
        f16x8 mulv[NPASS], resv[NPASS];
        if constexpr (!SG) {
            if (p.mulmode != VLP_MUL_NONE) {
#pragma unroll
                for (int ps = 0; ps < NPASS; ++ps) {
                    const int m = mb + ps * RPP;
                    if (m < p.M && ncol_ok && (EVEN || er + ps * RPP < 32)) mulv[ps] = ld8(p.mulsrc + (int64_t)m * p.ldm + nc);
                }
            }
            if (p.residual) {
#pragma unroll
                for (int ps = 0; ps < NPASS; ++ps) {
                    const int m = mb + ps * RPP;
                    if (m < p.M && ncol_ok && (EVEN || er + ps * RPP < 32)) resv[ps] = ld8(p.residual + (int64_t)m * p.ldr + nc);
                }
            }
        }
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            const int m = mb + ps * RPP;
            const int erow = EVEN ? er + ps * RPP : min(er + ps * RPP, 31);
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(ebuf + erow * RS + ec * 4);
            const f32x4 a1 = *reinterpret_cast<const f32x4*>(ebuf + erow * RS + ec * 4 + 16);
            if (m >= p.M || !ncol_ok) continue;
            if (!EVEN && er + ps * RPP >= 32) continue;
            float vv[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) { vv[j] = a0[j] * p.alpha + bias_v[j]; vv[4 + j] = a1[j] * p.alpha + bias_v[4 + j]; }
            if constexpr (SG) {
                wp_sg_store8(p, m, nc, ragged, vv);
            } else {
                wp_light_store8(p, m, nc, ragged, vv, mulv[ps], resv[ps]);
            }
        }
    }
}

// cfg (vlp_gemm_nt variant 64 + cfg, + 8 = XCD-aware tile order; cfg 8.. = variant 192 + cfg - 8):
//   0: 256x256, 4 waves (128x128), rings 2/2                1: 256x256, 8 waves (128x64), rings 2/2
//   2: as 0, fragments read 2 steps ahead                   3: 256x256, 4 waves, X/W roles, rings 3/2 (160 KiB), 2 steps ahead
//   4: 256x128, 4 waves (128x64), rings 3/3, spread         5: 256x128, 8 waves (64x64), rings 3/3, spread
//   6: as 4, 2 steps ahead                                  7: as 5, 2 steps ahead
//   8: 256x128, 8 waves, roles, rings 3/3 (control)         9: 256x128, 8 waves, roles, rings 4/2 (160 KiB)
//  10: 256x192, 8 waves (64x96), rings 2/2
int vlp_gemm_nt_wp_launch(GemmNtParams& p, int cfg, bool sg, hipStream_t s) {
    VLP_CHECK_ARG(sg || nt_epilogue_is_light(p), "vlp_gemm_nt: the wave-pipelined variants carry bias / ReLU / multiplier / dropout / residual / save-grad GeLU epilogues only");
    VLP_CHECK_ARG((int64_t)p.M * p.ldx < (1ll << 31) && (int64_t)p.N * p.ldw < (1ll << 31), "vlp_gemm_nt: wave-pipelined variants need M*ldx, N*ldw < 2^31");
#define LAUNCH_WP_(BNT, WGM, WGN, NSXV, NSWV, RLS, SPR, SGV, PBV, LDV)                                                                       \
    do {                                                                                                                               \
        const size_t smem = ((size_t)(NSXV) * 256 + (size_t)(NSWV) * (BNT)) * 64 * sizeof(f16);                                        \
        auto kfn = gemm_nt_wp_kernel<BNT, WGM, WGN, NSXV, NSWV, RLS, SPR, SGV, PBV, LDV>;                                                   \
        VLP_ONCE_PER_DEVICE((void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));                                                            \
        p.tiles_n = cdiv(p.N, (BNT));                                                                                                  \
        p.tiles_total = cdiv(p.M, 256) * p.tiles_n;                                                                                    \
        hipLaunchKernelGGL(kfn, dim3(p.tiles_total), dim3((WGM) * (WGN) * 64), smem, s, p);                                            \
    } while (0)
#define LAUNCH_WP(BNT, WGM, WGN, NSXV, NSWV, RLS, SPR, PBV, LDV) \
    do { if (sg) LAUNCH_WP_(BNT, WGM, WGN, NSXV, NSWV, RLS, SPR, true, PBV, LDV); else LAUNCH_WP_(BNT, WGM, WGN, NSXV, NSWV, RLS, SPR, false, PBV, LDV); } while (0)
    switch (cfg) {
        // product library: cfg 5 (variant 77: every N <= 1024 shape of the step) and cfg 1 (variant 73: the 256x256 wave-pipelined tile, kept
        // as a tuned-table candidate for wide outputs); the other configurations are investigation variants (-DVLP_LAB_BUILD)
        case 1: LAUNCH_WP(256, 2, 4, 2, 2, false, false, 2, 1); break;
        case 5: LAUNCH_WP(128, 4, 2, 3, 3, false, true, 0, 1); break;
#ifdef VLP_LAB_BUILD
        case 0: LAUNCH_WP(256, 2, 2, 2, 2, false, false, 4, 1); break;
        case 2: LAUNCH_WP(256, 2, 2, 2, 2, false, false, 4, 2); break;
        case 3: LAUNCH_WP(256, 2, 2, 3, 2, true, true, 4, 2); break;
        case 4: LAUNCH_WP(128, 2, 2, 3, 3, false, true, 2, 1); break;
        case 6: LAUNCH_WP(128, 2, 2, 3, 3, false, true, 2, 2); break;
        case 7: LAUNCH_WP(128, 4, 2, 3, 3, false, true, 0, 2); break;
        case 8: LAUNCH_WP(128, 4, 2, 3, 3, true, true, 0, 1); break;
        case 10: LAUNCH_WP(192, 4, 2, 2, 2, false, false, 2, 1); break;
        default: LAUNCH_WP(128, 4, 2, 4, 2, true, true, 0, 1); break;
#else
        default: return vlp_set_error(VLP_ERR_BAD_ARG, "vlp_gemm_nt: wave-pipelined configuration %d is an investigation variant (library built without -DVLP_LAB_BUILD)", cfg);
#endif
    }
#undef LAUNCH_WP
#undef LAUNCH_WP_
    return VLP_OK;
}
