// Persistent k-stream NT GEMM for gfx950 (vlp_gemm_nt variant 256, + 8 = XCD-aware workgroup order):   Y[M,N] = epi( alpha * X[M,K] . W[N,K]^T )
//
// Replaces (reference): the wide nn.Linear launches of a layer -- modeling.py:270-272 (packed QKV), :341 + :62-67 (FFN up + erf-GeLU) and
// the dgrad of :354 (FFN down) -- i.e. the N >= 2304 shapes, which the one-tile-per-workgroup kernels run as 1.5 - 2 synchronized tile
// rounds: every round pays a cold ring prologue, and all 256 CUs store their output tiles in the same few microseconds.
//
// Structure: one 8-wave workgroup per CU owns a CONTIGUOUS run of 256x128 output tiles (n fastest: the tiles of a run share their X
// panel) and walks them as ONE stream of k tiles through the 3-stage LDS-DMA ring of gemm_nt.hip (variant 3): the ring never drains at
// a tile boundary -- the first stages of tile i+1 are requested while the last k tiles of tile i are computed.  Wave tiles are 64x64 on
// v_mfma_f32_16x16x32_f16 with the W rows permuted so that a lane owns ONE output row and 16 consecutive columns per 16-row block
// (gemm_nt.hip): the epilogue needs no LDS and no barrier.  At a tile boundary a wave turns its accumulators into packed fp16 results
// (bias / save-grad GeLU / stored-derivative multiply) held in registers, and STORES them in eight slices behind the refill DMA of the
// first eight k tiles of the next tile: the output bytes leave the CU while its matrix cores work, and different CUs reach their
// boundaries at different times.  Only the last tile of a run stores at once.
//
// vmcnt: stores and the ordinary loads of bias / multiplier share the counter with the DMA stages.  Loads return in order among loads,
// so "at most LPS * (NS - 2) operations outstanding" still implies that every DMA piece older than the newest LPS has landed, whatever
// stores are in flight (a store still in flight only makes the wait stricter); the wait is unchanged from the ring.  (Counting the
// slices of the last two k tiles as allowed-in-flight as well -- gfx9-family parts return loads and stores in issue order -- was measured
// and changes nothing: profiles/r04_nt_persistent_stream_lab.txt, mode 2.)
// Results are bit-identical to the other variants (same MFMA, same k order per output element, same epilogue arithmetic).
#include "common.h"
#include "gemm_nt.h"
#include "gemm_nt_epilogue.h"

#define PS_BM 256
#define PS_BN 128
#define PS_BK 64
#define PS_NS 3
#define PS_MAXN 8192      // the bias vector is staged in the 16 KB of LDS behind the ring
#define PS_SLOTS 8        // store slices of a deferred tile = k tiles of the next tile that carry one (K >= PS_SLOTS * PS_BK)

#define PS_EPI_BIAS 0     // y = f16(alpha * acc + bias)
#define PS_EPI_SG 1       // z = f16(alpha * acc + bias); y = gelu(z); preact <- gelu'(z)          (VLP_ACT_GELU_SAVE_GRAD)
#define PS_EPI_MUL 2      // y = f16((alpha * acc + bias) * mul_src)                               (VLP_MUL_PLAIN)

DEVFN int ps_swz_x(int r) { return r & 7; }
DEVFN int ps_swz_w(int r) { return (((r >> 4) & 3) << 1) | ((r >> 1) & 1); }

template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm_nt_ps_kernel(GemmNtParams p, int tiles_per_wg) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    f16* smem = reinterpret_cast<f16*>(smem_raw);
    constexpr int RPP = 64, XP = PS_BM / RPP, WP = PS_BN / RPP;      // 512 threads stage 64 rows per pass
    constexpr int XT = PS_BM * PS_BK, WT = PS_BN * PS_BK, STG = XT + WT;   // halfs
    constexpr int LPS = XP + WP;                                       // DMA instructions per thread per stage
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 1, wn = wid & 1;
    const int g = lane >> 4, li = lane & 15;

    int wg = blockIdx.x;
    if (p.xcd_remap) {      // bijective: XCD x (= blockIdx % 8) owns a contiguous range of runs, i.e. of X panels
        const int nb = gridDim.x, q = nb >> 3, r = nb & 7, xcd = wg & 7, loc = wg >> 3;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int t_begin = wg * tiles_per_wg;
    const int t_end = min(t_begin + tiles_per_wg, p.tiles_total);
    const int ntl = t_end - t_begin;
    const int nk = p.K / PS_BK;
    const int F = ntl * nk;                  // k tiles of the run

    // ---- load stream: thread -> (row, physical 16-byte chunk) of each staging pass -----------------------------------------
    const int srow = tid >> 3, sx = tid & 7;
    const f16* xsrc[XP];
    const f16* wsrc[WP];
    auto set_load_tile = [&](int t) {
        const int m0 = (t / p.tiles_n) * PS_BM, n0 = (t % p.tiles_n) * PS_BN;
#pragma unroll
        for (int i = 0; i < XP; ++i) {
            const int r = srow + RPP * i;
            xsrc[i] = p.X + (int64_t)min(m0 + r, p.M - 1) * p.ldx + (sx ^ ps_swz_x(r)) * 8;
        }
#pragma unroll
        for (int i = 0; i < WP; ++i) {
            const int r = srow + RPP * i;
            wsrc[i] = p.W + (int64_t)(n0 + r) * p.ldw + (sx ^ ps_swz_w(r)) * 8;      // N % 128 == 0 (launcher)
        }
    };
    const uint32_t smem_lds = lds_addr_of(smem);
    auto glds = [&](int kt, int buf) {
        const uint32_t xs = smem_lds + (uint32_t)(buf * STG) * 2u;
        const uint32_t ws = xs + (uint32_t)XT * 2u;
#pragma unroll
        for (int i = 0; i < XP; ++i) glds16(xsrc[i] + (int64_t)kt * PS_BK, xs + (uint32_t)((RPP * i + 8 * wid) * PS_BK) * 2u);
#pragma unroll
        for (int i = 0; i < WP; ++i) glds16(wsrc[i] + (int64_t)kt * PS_BK, ws + (uint32_t)((RPP * i + 8 * wid) * PS_BK) * 2u);
    };
    int ld_f = 0, ld_kt = 0, ld_t = t_begin;          // next k tile of the run to request
    auto issue_next = [&](int buf) {
        if (ld_f < F) {
            glds(ld_kt, buf);
            ++ld_f;
            if (++ld_kt == nk) {
                ld_kt = 0;
                if (++ld_t < t_end) set_load_tile(ld_t);
            }
        } else {
            glds(nk - 1, buf);                        // past the end: a dummy reload (nobody reads it) keeps the vmcnt arithmetic constant
        }
    };

    // ---- fragment rows: X natural (row = wm*64 + 16*t + li), W permuted (row = wn*64 + 16*(li>>2) + 4*t + (li&3)) -------------
    int xrow[4], wrow[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        xrow[t] = wm * 64 + 16 * t + li;
        wrow[t] = wn * 64 + 16 * (li >> 2) + 4 * t + (li & 3);
    }

    f32x4 acc[4][4];
    f16x8 pend_y[4][2];                      // the previous tile's results: row block tm, column half h (8 columns)
    f16x8 pend_d[4][2];                      // save-grad GeLU: the derivative twin
    f16x8 mulv[4][2];                        // multiplier rows of the current tile (requested during its last k tile)
    bool have_pending = false;
    int pm0 = 0, pn0 = 0;                    // origin of the pending tile

    auto store_slot = [&](int j) {           // j = 2 * tm + h
        const int tm = j >> 1, h = j & 1;
        const int m = pm0 + wm * 64 + 16 * tm + li;
        const int nc = pn0 + wn * 64 + 16 * g + 8 * h;
        if (m < p.M) {
            if (EPI == PS_EPI_SG) st8_out<VLP_SS_SAVED>(p.preact + (int64_t)m * p.ldp + nc, pend_d[tm][h]);
            st8_out<VLP_SS_NT>(p.Y + (int64_t)m * p.ldy + nc, pend_y[tm][h]);
        }
    };

    set_load_tile(t_begin);
#pragma unroll
    for (int st = 0; st < PS_NS - 1; ++st) issue_next(st);
    int buf = 0, nbuf = PS_NS - 1;
    // The bias vector goes into LDS behind the ring, once: an ordinary global load beside LDS-DMA makes the compiler wait vmcnt(0) at
    // its first use, which at every tile boundary would drain the two stages in flight (measured: 52.3 vs 47.9 us on N = 2304); here
    // the wait only covers the two prologue stages, which the first k tiles need anyway.  The first read is twelve barriers away.
    f16* bias_s = smem + PS_NS * STG;
    if (p.bias) {
        for (int c = tid * 8; c < p.N; c += 512 * 8) st8(bias_s + c, ld8(p.bias + c));        // N % 128 == 0
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }

    for (int tl = 0; tl < ntl; ++tl) {
        const int t = t_begin + tl;
        const int m0 = (t / p.tiles_n) * PS_BM, n0 = (t % p.tiles_n) * PS_BN;
        const int ncol0 = n0 + wn * 64 + 16 * g;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

        // one k tile: wait for this wave's pieces of the stage, barrier (all pieces landed; the stage consumed one iteration ago is free),
        // first fragment reads, refill DMA, [deferred stores | multiplier request], second fragment reads, 32 MFMAs
        auto ktile = [&](int slot, bool last) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPS * (PS_NS - 2)) : "memory");
            __builtin_amdgcn_s_barrier();
            const f16* xs = smem + buf * STG;
            const f16* ws = xs + XT;
            f16x8 xf0[4], wf0[4], xf1[4], wf1[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                xf0[q] = ld8(xs + xrow[q] * PS_BK + ((g ^ ps_swz_x(xrow[q])) << 3));
                wf0[q] = ld8(ws + wrow[q] * PS_BK + ((g ^ ps_swz_w(wrow[q])) << 3));
            }
            __builtin_amdgcn_sched_barrier(0);
            issue_next(nbuf);
            if (slot >= 0 && have_pending) store_slot(slot);
            if (EPI == PS_EPI_MUL && last) {
#pragma unroll
                for (int tm = 0; tm < 4; ++tm) {
                    const int m = min(m0 + wm * 64 + 16 * tm + li, p.M - 1);
                    mulv[tm][0] = ld8(p.mulsrc + (int64_t)m * p.ldm + ncol0);
                    mulv[tm][1] = ld8(p.mulsrc + (int64_t)m * p.ldm + ncol0 + 8);
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                xf1[q] = ld8(xs + xrow[q] * PS_BK + (((4 + g) ^ ps_swz_x(xrow[q])) << 3));
                wf1[q] = ld8(ws + wrow[q] * PS_BK + (((4 + g) ^ ps_swz_w(wrow[q])) << 3));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int tm = 0; tm < 4; ++tm)
#pragma unroll
                for (int tn = 0; tn < 4; ++tn)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf0[tn], xf0[tm], acc[tm][tn], 0, 0, 0);
#pragma unroll
            for (int tm = 0; tm < 4; ++tm)
#pragma unroll
                for (int tn = 0; tn < 4; ++tn)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf1[tn], xf1[tm], acc[tm][tn], 0, 0, 0);
            buf = (buf + 1 == PS_NS) ? 0 : buf + 1;
            nbuf = (nbuf + 1 == PS_NS) ? 0 : nbuf + 1;
        };
#pragma unroll
        for (int j = 0; j < PS_SLOTS; ++j) ktile(j, false);
        for (int kt = PS_SLOTS; kt < nk - 1; ++kt) ktile(-1, false);
        ktile(-1, true);                     // nk > PS_SLOTS (launcher)

        // ---- tile boundary: accumulators -> packed results (bit-for-bit the arithmetic of nt_epilogue8) ---------------------------
        f16x8 b0, b1;
#pragma unroll
        for (int j = 0; j < 8; ++j) { b0[j] = (f16)0.f; b1[j] = (f16)0.f; }
        if (p.bias) {
            b0 = ld8(bias_s + ncol0);
            b1 = ld8(bias_s + ncol0 + 8);
        }
#pragma unroll
        for (int tm = 0; tm < 4; ++tm) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float vv[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int c = 8 * h + j;           // column inside the lane's 16: MFMA tile tn = c >> 2, register c & 3
                    vv[j] = acc[tm][c >> 2][c & 3] * p.alpha + (float)(h ? b1[j] : b0[j]);
                }
                f16x8 o;
                if (EPI == PS_EPI_SG) {
                    f16x8 d;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        float gl, gp;
                        gelu_and_grad_f((float)(f16)vv[j], gl, gp);
                        o[j] = (f16)gl;
                        d[j] = (f16)gp;
                    }
                    pend_d[tm][h] = d;
                } else if (EPI == PS_EPI_MUL) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) o[j] = (f16)(vv[j] * (float)mulv[tm][h][j]);
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) o[j] = (f16)vv[j];
                }
                pend_y[tm][h] = o;
            }
        }
        pm0 = m0;
        pn0 = n0;
        have_pending = true;
        if (tl + 1 == ntl) {                 // last tile of the run: nothing left to hide the stores behind
#pragma unroll
            for (int j = 0; j < PS_SLOTS; ++j) store_slot(j);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the dummy tail reloads must land before the LDS is released
}

static int ps_cu_count() {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        n = 256;
    }
    return n;
}

// which epilogues / shapes the persistent kernel carries; anything else runs on the rings (vlp_gemm_nt falls back)
bool vlp_gemm_nt_ps_eligible(const GemmNtParams& p, bool sg) {
    if (p.N % PS_BN || p.K % PS_BK || p.K / PS_BK <= PS_SLOTS || p.N > PS_MAXN) return false;
    if (p.residual || p.drop.thresh) return false;
    if (sg) return p.mulmode == VLP_MUL_NONE && p.preact != nullptr;
    if (p.preact || p.act != VLP_ACT_NONE) return false;
    return p.mulmode == VLP_MUL_NONE || p.mulmode == VLP_MUL_PLAIN;
}

int vlp_gemm_nt_ps_launch(GemmNtParams& p, bool sg, hipStream_t s) {
    VLP_CHECK_ARG(vlp_gemm_nt_ps_eligible(p, sg), "vlp_gemm_nt: the persistent variant carries bias / save-grad GeLU / plain-multiplier epilogues on N %% 128 == 0, N <= 8192, K > 512 only");
    p.tiles_n = p.N / PS_BN;
    p.tiles_total = cdiv(p.M, PS_BM) * p.tiles_n;
    static thread_local int ncu_dev = -1, ncu = 0;
    { int dev = 0; (void)hipGetDevice(&dev); if (dev != ncu_dev) { ncu = ps_cu_count(); ncu_dev = dev; } }
    int cap = ncu;
    if (const char* e = getenv("VLP_NT_PS_GRID")) { const int v = atoi(e); if (v > 0) cap = v; }      // investigation: workgroups of the launch
    const int per = cdiv(p.tiles_total, cap);
    const int grid = cdiv(p.tiles_total, per);
    const size_t smem = (size_t)PS_NS * (PS_BM + PS_BN) * PS_BK * sizeof(f16) + (size_t)PS_MAXN * sizeof(f16);      // 144 + 16 KB
#define PS_LAUNCH(E)                                                                                                                          \
    do {                                                                                                                                      \
        VLP_ONCE_PER_DEVICE((void)hipFuncSetAttribute((const void*)gemm_nt_ps_kernel<E>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)); \
        hipLaunchKernelGGL((gemm_nt_ps_kernel<E>), dim3(grid), dim3(512), smem, s, p, per);                                                 \
    } while (0)
    if (sg) PS_LAUNCH(PS_EPI_SG);
    else if (p.mulmode == VLP_MUL_PLAIN) PS_LAUNCH(PS_EPI_MUL);
    else PS_LAUNCH(PS_EPI_BIAS);
#undef PS_LAUNCH
    return VLP_OK;
}
