// Phased NT GEMM for gfx950:  Y[M,N] = epi( alpha * X[M,K] . W[N,K]^T ), same contract and epilogue as gemm_nt.hip
// (replaces the nn.Linear forwards / dgrads listed there), built for the large encoder GEMMs (M = B*L = 10 688).
//
// Why a second kernel: the 2-barrier kernels of gemm_nt.hip drain the LDS-DMA queue (vmcnt(0)) in front of every barrier and give
// each wave a 64x64 output, i.e. one ds_read_b128 per 2 MFMAs; rocprofv3 shows their waves parked ~40 % of the time.  Here
//   * the workgroup tile is 256 rows x BN_T (256 | 128) columns x 64 k, 8 waves, 1 workgroup per CU (LDS 128 / 96 KiB);
//   * each operand tile is staged as two HALF tiles (128 X rows; BN_T/2 W rows) by LDS-DMA (global_load_lds_dwordx4), and a wave
//     owns rows from BOTH halves of X and of W, so its output splits into 2x2 quadrants (X half hm) x (W half hn);
//   * one K tile = 4 phases walking the quadrants (0,0) (0,1) (1,1) (1,0): a phase reads only the operand half that changes
//     (12 / 4 / 8 / 0 ds_read_b128 for BN_T = 256), issues the DMA of one half tile that is needed 4-5 phases later, then runs
//     one quadrant of MFMAs (16 x v_mfma_f32_16x16x32_f16 for BN_T = 256): 28 LDS reads per 64 MFMAs;
//   * DMA is never drained inside the loop: "s_waitcnt vmcnt(N)" with N = the loads of the 4 youngest half tiles; a half tile is
//     read one phase after the wait that retires it and overwritten >= 2 phases after its last read (rules below);
//   * waves 0-3 and 4-7 (the two waves of each SIMD) run half a phase apart (one extra s_barrier for the second group), so one
//     wave's LDS reads / DMA issue overlap with the MFMA section of the other wave on the same SIMD.
//
// Synchronisation rules (slot k = time between workgroup barriers k and k+1; group 0 runs its read section R_j in slot 2j and its
// MFMA section M_j in slot 2j+1, group 1 one slot later):
//   RAW  every wave executes the counted vmcnt for half tile Z in R_j (slots 2j / 2j+1); any wave may ds_read Z from slot 2j+2 on,
//        i.e. in R_{j+1} or later -- never in the phase of the wait itself.
//   WAR  reads issued in R_j are retired by the lgkmcnt(0) that opens M_j (slots 2j+1 / 2j+2); a DMA into that buffer may be issued
//        from slot 2j+3 on, i.e. in R_{j+2} or later for both groups.
// Schedule per K tile u (parity b = u & 1), "issue Z(t)" = DMA of half tile Z of K tile t into parity t & 1:
//   phase 4u+0  read X0,W0[b]   issue W1(u+1)   vmcnt(N)  -> retires W1(u)      MFMA (X0,W0)
//   phase 4u+1  read W1[b]      issue X1(u+1)   vmcnt(N)  -> retires X1(u)      MFMA (X0,W1)
//   phase 4u+2  read X1[b]      issue X0(u+2)                                   MFMA (X1,W1)
//   phase 4u+3  -               issue W0(u+2)   vmcnt(N)  -> retires X0,W0(u+1) MFMA (X1,W0)   (W0 fragments stay in registers)
// K tiles past the end are clamped to the last one (dummy reloads into buffers that are never read again) so N stays constant.
//
// Fragment / epilogue layout as in gemm_nt.hip: the MFMA computes Y^T tiles (A operand = W rows, B operand = X rows); W rows of a
// wave's 32-row slice of a half are visited in the order n = 8*(i>>2) + 4*tn + (i&3), so a lane ends up with ONE output row m per
// m-tile and 8 consecutive n per W half -> 16-byte stores straight from registers.  LDS rows are 128 B with a 16-B-chunk XOR
// swizzle (X: row & 7; W: row bits 1,3,4 = lane>>1 for the fragment reads), applied on the DMA source address.
#include "common.h"
#include "gemm_nt.h"
#include "gemm_nt_epilogue.h"

#define PH_BK 64

#define PH_GLDS(src, dst) glds16((src), lds_addr_of(dst))
#define PH_SCHED() __builtin_amdgcn_sched_barrier(0)
#define PH_BAR()        \
    do {                \
        PH_SCHED();     \
        __builtin_amdgcn_s_barrier(); \
        PH_SCHED();     \
    } while (0)

template <int BN_T, int MODE>
__global__ __launch_bounds__(512, 1) void gemm_nt_ph_kernel(GemmNtParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    f16* smem = reinterpret_cast<f16*>(smem_raw);
    constexpr int WAVES_N = BN_T / 64;            // 4 | 2
    constexpr int WAVES_M = 8 / WAVES_N;          // 2 | 4
    constexpr int QM = 128 / (WAVES_M * 16);      // m-tiles of a wave inside one X half: 4 | 2
    constexpr int QN = 2;                         // n-tiles of a wave inside one W half
    constexpr int WROWS = BN_T / 2;               // rows of a W half: 128 | 64
    constexpr int NXP = 2, NWP = WROWS / 64;      // DMA instructions per thread per half tile
    constexpr int XH = 128 * PH_BK, WH = WROWS * PH_BK;   // halfs per half tile
    constexpr int PAR = 2 * XH + 2 * WH;          // halfs per parity
    constexpr int VMN = 2 * (NXP + NWP);          // loads of the 4 youngest half tiles

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid / WAVES_N, wn = wid % WAVES_N;
    const int grp = (MODE == 3 || MODE == 1) ? 0 : (wid >> 2);   // waves w and w+4 share a SIMD: one of each group per SIMD
    const int g = lane >> 4, li = lane & 15;

    int bid = blockIdx.x;
    if (p.xcd_remap) {
        const int nb = gridDim.x, q = nb >> 3, r = nb & 7, xcd = bid & 7, loc = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int tile_m = bid / p.tiles_n, tile_n = bid % p.tiles_n;
    const int m0 = tile_m * 256, n0 = tile_n * BN_T;
    const int nk = p.K / PH_BK;

    // ---- DMA geometry: pass j of a half tile covers its rows 64j..64j+63; thread -> (row 64j + tid>>3, physical chunk tid&7)
    const int srow = tid >> 3, sx = tid & 7;
    uint32_t xoff[2 * NXP], woff[2 * NWP];        // element offsets of this thread's source chunks (k tile 0)
#pragma unroll
    for (int j = 0; j < 2 * NXP; ++j) {
        const int r = 64 * j + srow;               // tile row (half = j / NXP)
        const int mr = min(m0 + r, p.M - 1);
        xoff[j] = (uint32_t)((int64_t)mr * p.ldx + ((sx ^ (r & 7)) << 3));
    }
#pragma unroll
    for (int j = 0; j < 2 * NWP; ++j) {
        const int r = 64 * j + srow;               // tile row; row inside its half = r % WROWS
        const int nr = min(n0 + r, p.N - 1);
        const int f = ((r >> 1) & 1) | (((r >> 3) & 1) << 1) | (((r >> 4) & 1) << 2);
        woff[j] = (uint32_t)((int64_t)nr * p.ldw + ((sx ^ f) << 3));
    }
    auto issueX = [&](int h, int t, int b) {
        const int kt = min(t, nk - 1) * PH_BK;
#pragma unroll
        for (int i = 0; i < NXP; ++i) PH_GLDS(p.X + xoff[h * NXP + i] + kt, smem + b * PAR + h * XH + (64 * i + 8 * wid) * PH_BK);
    };
    auto issueW = [&](int h, int t, int b) {
        const int kt = min(t, nk - 1) * PH_BK;
#pragma unroll
        for (int i = 0; i < NWP; ++i) PH_GLDS(p.W + woff[h * NWP + i] + kt, smem + b * PAR + 2 * XH + h * WH + (64 * i + 8 * wid) * PH_BK);
    };

    // ---- fragment reads ----------------------------------------------------------------------------
    int xrd[2], wrd[2];                            // halfs inside a half tile, per 32-wide k sub-step
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        xrd[ks] = (wm * QM * 16 + li) * PH_BK + (((ks * 4 + g) ^ (li & 7)) << 3);
        wrd[ks] = (wn * 32 + 8 * (li >> 2) + (li & 3)) * PH_BK + (((ks * 4 + g) ^ (li >> 1)) << 3);
    }
    f16x8 xf[QM][2], w0f[QN][2], w1f[QN][2];
    f32x4 acc[2 * QM][2 * QN];
#pragma unroll
    for (int a = 0; a < 2 * QM; ++a)
#pragma unroll
        for (int b = 0; b < 2 * QN; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    auto readX = [&](int h, int b) {
        const f16* base = smem + b * PAR + h * XH;
#pragma unroll
        for (int t = 0; t < QM; ++t)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) xf[t][ks] = ld8(base + xrd[ks] + t * 16 * PH_BK);
    };
    auto readW = [&](f16x8 (&wf)[QN][2], int h, int b) {
        const f16* base = smem + b * PAR + 2 * XH + h * WH;
#pragma unroll
        for (int t = 0; t < QN; ++t)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) wf[t][ks] = ld8(base + wrd[ks] + t * 4 * PH_BK);
    };
    auto mma = [&](int hm, int hn, f16x8 (&wf)[QN][2]) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int tm = 0; tm < QM; ++tm)
#pragma unroll
                for (int tn = 0; tn < QN; ++tn)
                    acc[hm * QM + tm][hn * QN + tn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[tn][ks], xf[tm][ks], acc[hm * QM + tm][hn * QN + tn], 0, 0, 0);
    };
#define PH_VMWAIT() asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VMN) : "memory")
#define PH_R2M()                                              \
    do {                                                      \
        PH_BAR();                                             \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    \
        PH_SCHED();                                           \
        __builtin_amdgcn_s_setprio(1);                        \
    } while (0)
#define PH_MEND()                      \
    do {                               \
        __builtin_amdgcn_s_setprio(0);                \
        if (MODE != 3) PH_BAR();       \
    } while (0)

    // ---- prologue: X0,W0,W1,X1 of tile 0 and X0,W0 of tile 1 --------------------------------------
    issueX(0, 0, 0); issueW(0, 0, 0); issueW(1, 0, 0); issueX(1, 0, 0); issueX(0, 1, 1); issueW(0, 1, 1);
    PH_VMWAIT();
    PH_BAR();
    if (grp == 1) PH_BAR();

    auto ktile = [&](int u, int b) {
        // phase 0: quadrant (X0, W0)
        readW(w0f, 0, b); readX(0, b);
        issueW(1, u + 1, b ^ 1);
        PH_VMWAIT();
        PH_R2M(); mma(0, 0, w0f); PH_MEND();
        // phase 1: quadrant (X0, W1)
        readW(w1f, 1, b);
        issueX(1, u + 1, b ^ 1);
        PH_VMWAIT();
        PH_R2M(); mma(0, 1, w1f); PH_MEND();
        // phase 2: quadrant (X1, W1)
        readX(1, b);
        issueX(0, u + 2, b);
        PH_R2M(); mma(1, 1, w1f); PH_MEND();
        // phase 3: quadrant (X1, W0)
        issueW(0, u + 2, b);
        PH_VMWAIT();
        PH_R2M(); mma(1, 0, w0f); PH_MEND();
    };
    // MODE 1: one barrier per phase (in front of the reads), no wave-group stagger, instruction order inside a phase left to the compiler
    auto ktile1 = [&](int u, int b) {
        PH_VMWAIT(); PH_BAR();
        issueW(1, u + 1, b ^ 1);
        readW(w0f, 0, b); readX(0, b);
        mma(0, 0, w0f);
        PH_VMWAIT(); PH_BAR();
        issueX(1, u + 1, b ^ 1);
        readW(w1f, 1, b);
        mma(0, 1, w1f);
        PH_VMWAIT(); PH_BAR();
        issueX(0, u + 2, b);
        readX(1, b);
        mma(1, 1, w1f);
        issueW(0, u + 2, b);
        mma(1, 0, w0f);
    };
    int u = 0;
    if (MODE == 1) {
        for (; u + 1 < nk; u += 2) {
            ktile1(u, 0);
            ktile1(u + 1, 1);
        }
        if (u < nk) ktile1(u, 0);
    } else {
        for (; u + 1 < nk; u += 2) {
            ktile(u, 0);
            ktile(u + 1, 1);
        }
        if (u < nk) ktile(u, 0);
    }
    if (grp == 0) PH_BAR();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the clamped tail reloads must land before this workgroup's LDS is released

    // ---- epilogue: lane owns row m per m-tile and 8 consecutive n per W half -----------------------
#pragma unroll
    for (int tm = 0; tm < 2 * QM; ++tm) {
        const int m = m0 + (tm / QM) * 128 + wm * QM * 16 + 16 * (tm % QM) + li;
        if (m >= p.M) continue;
        const uint32_t rkey = p.drop.thresh ? drop_rowkey(p.drop, nt_drop_row(p, m)) : 0u;
#pragma unroll
        for (int hn = 0; hn < 2; ++hn) {
            float v[8];
#pragma unroll
            for (int tn = 0; tn < QN; ++tn)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[tn * 4 + r] = acc[tm][hn * QN + tn][r] * p.alpha;
            nt_epilogue8(p, m, n0 + hn * WROWS + wn * 32 + 8 * g, v, rkey);
        }
    }
}

int vlp_gemm_nt_ph_launch(GemmNtParams& p, int bn, int mode, hipStream_t s) {
    VLP_CHECK_ARG((int64_t)p.M * p.ldx < (1ll << 31) && (int64_t)p.N * p.ldw < (1ll << 31), "vlp_gemm_nt: phased variants need M*ldx, N*ldw < 2^31");
    VLP_CHECK_ARG(p.K >= 2 * PH_BK, "vlp_gemm_nt: phased variants need K >= 128");
#define LAUNCH_PH(BNT, MD)                                                                                                 \
    do {                                                                                                                 \
        const size_t smem = (size_t)2 * (2 * 128 + (BNT)) * PH_BK * sizeof(f16);                                         \
        VLP_ONCE_PER_DEVICE((void)hipFuncSetAttribute((const void*)gemm_nt_ph_kernel<BNT, MD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));                                                            \
        p.tiles_n = cdiv(p.N, (BNT));                                                                                    \
        hipLaunchKernelGGL((gemm_nt_ph_kernel<BNT, MD>), dim3(cdiv(p.M, 256) * p.tiles_n), dim3(512), smem, s, p);           \
    } while (0)
    if (bn == 256) {
        if (mode == 0) LAUNCH_PH(256, 0); else if (mode == 1) LAUNCH_PH(256, 1); else if (mode == 2) LAUNCH_PH(256, 2); else LAUNCH_PH(256, 3);
    } else {
        if (mode == 0) LAUNCH_PH(128, 0); else if (mode == 1) LAUNCH_PH(128, 1); else if (mode == 2) LAUNCH_PH(128, 2); else LAUNCH_PH(128, 3);
    }
#undef LAUNCH_PH
    return VLP_OK;
}
