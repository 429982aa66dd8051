// NT GEMM with fused epilogue for gfx950:   Y[M,N] = epi( alpha * X[M,K] . W[N,K]^T )
//
// Replaces (reference): every nn.Linear forward on the hot path and, with a transposed weight
// shadow as W, every dgrad:  modeling.py:270-272 (QKV), :314 (attn out), :341 (FFN up + gelu :62-67),
// :354 (FFN down), :432 (head transform), :481 (tied decoder), :1003-1005 (fc7 + region proj),
// :1016 (box/class proj), :1027-1029 (VQA classifier), plus dropout/residual (:315-316, :355-356).
//
// Design (CDNA4): 128x128x64 block tile, 4 waves (2x2), each wave 64x64 = 4x4 v_mfma_f32_16x16x32_f16
// tiles, fp32 accumulate.  The MFMA "A" operand is fed from W rows and the "B" operand from X rows,
// i.e. the instruction computes Y^T tiles: in the C/D layout (col = lane&15, row = 4*(lane>>4)+reg) a
// lane then owns ONE output row m and FOUR consecutive n per tile; with the W rows of the wave's 64-row
// sub-tile visited in the permuted order  n = 16*(i>>2) + 4*tn + (i&3)  the 4 tiles x 4 regs of a lane
// are 16 *consecutive* n, so bias / residual / gelu'() inputs are read and Y is written as 16-byte
// vectors straight from registers -- no LDS round trip in the epilogue.
// LDS tiles are [128 rows][64 halfs] (128-B rows) with a 16-B-chunk XOR swizzle chosen per operand so
// that every ds_read_b128 lane group of the fragment reads touches 16 distinct 16-B slots.
// Staging: VARIANT 0 = global_load_dwordx4 -> VGPR -> ds_write_b128 (issued before / written after the
// MFMA block of the current tile), VARIANT 1 = global_load_lds_dwordx4 (LDS-DMA; the swizzle is applied
// to the per-lane SOURCE address because the LDS destination is lane-linear).
#include "common.h"
#include "gemm_nt.h"
#include "gemm_nt_epilogue.h"

#define BM 128
#define BK 64


#ifdef VLP_NT_DEBUG
// investigation build: wave 0 of every workgroup records the 100 MHz wall clock at phase boundaries (tools/nt_trace.py)
__device__ unsigned long long g_nt_trace[4096 * 4];
#define NT_TRACE(slot) do { if (wid == 0 && lane == 0 && blockIdx.x < 4096) g_nt_trace[blockIdx.x * 4 + (slot)] = wall_clock64(); } while (0)
extern "C" int vlp_debug_read_nt_trace(void* dst, int64_t bytes) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_nt_trace), bytes); }
#else
#define NT_TRACE(slot) do { } while (0)
#endif
DEVFN int swz_x(int r) { return r & 7; }
DEVFN int swz_w(int r) { return (((r >> 4) & 3) << 1) | ((r >> 1) & 1); }

// VARIANT 0: register-staged, double-buffered   1: LDS-DMA, double-buffered   2: LDS-DMA, single buffer (32 KiB -> up to
// 4 workgroups per CU; overlap comes from co-resident workgroups instead of an in-block pipeline)
// 3: LDS-DMA ring of NS stages (3 for the 256x128 tile = 144 KiB, 4 for 128x128 = 128 KiB; one workgroup per CU), counted vmcnt: the
//    DMA queue is never drained inside the loop -- NS-1 stages stay in flight across the single raw s_barrier of a k tile.
//    On the training shapes (thousands of workgroups) it equals variant 1 (793 vs 776 TFLOP/s at K = 3072): those loops are not
//    bound by the DMA round trip.  On the decoder's skinny GEMMs (M = 128..640: a few dozen workgroups, each alone on its CU and
//    paying one L2 round trip per k tile) the 128x128 ring (variant 17) is the latency-hiding kernel; the autotuner picks it there.
// BM_T: rows of the block tile (128 -> 4 waves 2x2, 256 -> 8 waves 4x2); the wave tile is always 64x64.
// NSR: stages of the VARIANT 3 ring (0 = default: 3 for 256-row tiles, 4 for 128x128).
// EPI: 0 = the shared epilogue (every mode, run-time switches); 1 = "multiply by a stored tensor" only (the FFN-down dgrad: alpha * acc (+ bias)
//      times mulsrc, nothing else; N % 16 == 0): the lane's eight multiplier vectors are requested TOGETHER before the first one is used.  In the
//      shared epilogue each 8-wide vector sits behind its own run-time branches -- load, full vmcnt wait, multiply, store, eight times in a row per
//      lane: the stored-derivative multiply cost 14 us on a 59 us GEMM (10 688 x 768 x 3072).  Same arithmetic, same bits.
template <int VARIANT, int BM_T, int BN_T, bool SG, int NSR = 0, int EPI = 0>
__global__ __launch_bounds__((BM_T / 64) * (BN_T / 64) * 64, (VARIANT == 3 ? (BN_T == 256 ? 4 : BM_T / 128) : ((VARIANT == 2 || BN_T == 256) ? 4 : 2))) void gemm_nt_kernel(GemmNtParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    f16* smem = reinterpret_cast<f16*>(smem_raw);
    constexpr int WN_ = BN_T / 64;        // waves along n
    constexpr int T = (BM_T / 64) * WN_ * 64;   // threads
    constexpr int RPP = T / 8;            // tile rows covered per staging pass
    constexpr int XP = BM_T / RPP;        // passes for the X tile
    constexpr int WP = BN_T / RPP;        // passes for the W tile
    constexpr int XT = BM_T * BK, WT = BN_T * BK;      // halfs
    // layout: [buf][ X tile BM_T*64 | W tile BN_T*64 ]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid / WN_, wn = wid % WN_;
    const int g = lane >> 4, li = lane & 15;

    int bid = blockIdx.x;
    if (p.xcd_remap) {      // bijective for any grid size: XCD x owns (q+1) tiles if x < r else q
        const int nb = gridDim.x, q = nb >> 3, r = nb & 7, xcd = bid & 7, loc = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int tile_m = bid / p.tiles_n;
    const int tile_n = bid % p.tiles_n;
    const int m0 = tile_m * BM_T, n0 = tile_n * BN_T;

    // ---- staging geometry: thread -> (row, physical chunk) per pass ------------------------------
    const int srow = tid >> 3;        // 0..RPP-1 (+RPP*i)
    const int sx = tid & 7;           // physical 16-B chunk inside the 128-B LDS row
    const f16* xsrc[XP];
    const f16* wsrc[WP];
    int lds_off[XP];                  // halfs, inside a tile (register-staged writes)
#pragma unroll
    for (int i = 0; i < XP; ++i) {
        const int r = srow + RPP * i;
        const int mr = min(m0 + r, p.M - 1);
        xsrc[i] = p.X + (int64_t)mr * p.ldx + (sx ^ swz_x(r)) * 8;     // logical chunk held at physical slot sx of row r
        lds_off[i] = r * BK + sx * 8;
    }
#pragma unroll
    for (int i = 0; i < WP; ++i) {
        const int r = srow + RPP * i;
        const int nr = min(n0 + r, p.N - 1);
        wsrc[i] = p.W + (int64_t)nr * p.ldw + (sx ^ swz_w(r)) * 8;
    }

    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

#ifdef VLP_NT_DEBUG
    // start-stagger experiment: every other workgroup of an XCD (first round only) starts (dbg >> 8) / 10 us late, so that the store-bound
    // epilogues of the two halves of the chip do not coincide
    if ((p.dbg >> 8) && ((blockIdx.x >> 3) & 1) && blockIdx.x < 256) {
        const uint64_t t0 = wall_clock64();
        while (wall_clock64() - t0 < (uint64_t)(p.dbg >> 8) * 10u) __builtin_amdgcn_s_sleep(8);
    }
#endif
    const int nk = p.K / BK;
    NT_TRACE(0);

    // fragment read rows inside a tile
    // X tile (natural rows): row = wm*64 + 16*tm + li ; W tile (permuted rows): row = wn*64 + 16*(li>>2) + 4*tn + (li&3)
    int xrow[4], wrow[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        xrow[t] = wm * 64 + 16 * t + li;
        wrow[t] = wn * 64 + 16 * (li >> 2) + 4 * t + (li & 3);
    }

    u32x4 xreg[XP], wreg[WP];

    auto gload = [&](int kt) {
#pragma unroll
        for (int i = 0; i < XP; ++i) xreg[i] = *reinterpret_cast<const u32x4*>(xsrc[i] + (int64_t)kt * BK);
#pragma unroll
        for (int i = 0; i < WP; ++i) wreg[i] = *reinterpret_cast<const u32x4*>(wsrc[i] + (int64_t)kt * BK);
    };
    auto lstore = [&](int buf) {
        f16* xs = smem + buf * (XT + WT);
        f16* ws = xs + XT;
#pragma unroll
        for (int i = 0; i < XP; ++i) *reinterpret_cast<u32x4*>(xs + lds_off[i]) = xreg[i];
#pragma unroll
        for (int i = 0; i < WP; ++i) *reinterpret_cast<u32x4*>(ws + lds_off[i]) = wreg[i];
    };
    const uint32_t smem_lds = lds_addr_of(smem);
    auto glds = [&](int kt, int buf) {
        // LDS-DMA: destination = wave-uniform base + lane*16 B.  Pass i covers tile rows RPP*i..RPP*i+RPP-1;
        // this wave's 64 lanes cover rows RPP*i + 8*wid .. +7 (8 lanes per 128-B row).
        const uint32_t xs = smem_lds + (uint32_t)(buf * (XT + WT)) * 2u;
        const uint32_t ws = xs + (uint32_t)XT * 2u;
#ifdef VLP_NT_DEBUG
        if (p.dbg & 24) {      // cache-policy experiments: 8 = X tile nt, 16 = X tile sc1 (W stays default)
#pragma unroll
            for (int i = 0; i < XP; ++i) {
                if (p.dbg & 8) glds16_nt(xsrc[i] + (int64_t)kt * BK, xs + (uint32_t)((RPP * i + 8 * wid) * BK) * 2u);
                else glds16_sc1(xsrc[i] + (int64_t)kt * BK, xs + (uint32_t)((RPP * i + 8 * wid) * BK) * 2u);
            }
#pragma unroll
            for (int i = 0; i < WP; ++i) glds16(wsrc[i] + (int64_t)kt * BK, ws + (uint32_t)((RPP * i + 8 * wid) * BK) * 2u);
            return;
        }
#endif
#pragma unroll
        for (int i = 0; i < XP; ++i) glds16(xsrc[i] + (int64_t)kt * BK, xs + (uint32_t)((RPP * i + 8 * wid) * BK) * 2u);
#pragma unroll
        for (int i = 0; i < WP; ++i) glds16(wsrc[i] + (int64_t)kt * BK, ws + (uint32_t)((RPP * i + 8 * wid) * BK) * 2u);
    };
    auto compute_ks = [&](int buf, int ks) {
        const f16* xs = smem + buf * (XT + WT);
        const f16* ws = xs + XT;
        const int c = ks * 4 + g;
        f16x8 xf[4], wf[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            xf[t] = ld8(xs + xrow[t] * BK + ((c ^ swz_x(xrow[t])) << 3));
            wf[t] = ld8(ws + wrow[t] * BK + ((c ^ swz_w(wrow[t])) << 3));
        }
#pragma unroll
        for (int tm = 0; tm < 4; ++tm)
#pragma unroll
            for (int tn = 0; tn < 4; ++tn)
                acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[tn], xf[tm], acc[tm][tn], 0, 0, 0);
    };
    auto compute = [&](int buf) {
        compute_ks(buf, 0);
        compute_ks(buf, 1);
    };

    if (VARIANT == 0) {
        gload(0);
        lstore(0);
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            const int buf = kt & 1;
            if (kt + 1 < nk) gload(kt + 1);
            compute(buf);
            if (kt + 1 < nk) lstore(buf ^ 1);
            __syncthreads();
        }
    } else if (VARIANT == 1) {
        glds(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            const int buf = kt & 1;
            if (kt + 1 < nk) glds(kt + 1, buf ^ 1);
            compute(buf);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    } else if (VARIANT == 2) {
        for (int kt = 0; kt < nk; ++kt) {
            glds(kt, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            compute(0);
            __syncthreads();
        }
    } else {
        // ring of NS stages.  Iteration kt: (1) wait until this wave's pieces of stage kt have landed -- the NS-2 younger stages may
        // stay in flight; (2) barrier: every wave's pieces have landed AND every wave has finished the MFMAs (hence the ds_reads) of
        // stage kt-1, whose buffer is therefore free; (3) refill that buffer with stage kt+NS-1; (4) compute stage kt.
        // k tiles past the end are clamped to the last one (dummy reloads into buffers nobody reads again) so the count stays constant.
        constexpr int NS = NSR ? NSR : ((BM_T == 256) ? 3 : 4);
        constexpr int LPS = XP + WP;                 // DMA instructions per thread per stage
        // (An L2 prefetch of later stages by dummy dword loads -- one 128-byte row per thread, counted into the vmcnt budget -- was measured
        // and removed: every byte then crosses the L2 -> CU path twice, and that path, not HBM latency alone, is what bounds these loops:
        // 78 vs 63 us on 10688x768x3072 with cold operands, tools/nt_lab.py --rotate=12.)
#pragma unroll
        for (int st = 0; st < NS - 1; ++st) glds(min(st, nk - 1), st);
        int buf = 0, nbuf = NS - 1;
        for (int kt = 0; kt < nk; ++kt) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPS * (NS - 2)) : "memory");
            __builtin_amdgcn_s_barrier();
            // (issuing the refill later in the iteration -- staggered between the waves that share a SIMD so that one streams DMA addresses
            // while the other runs MFMAs -- was measured and lost 0.2 ms per step: the refill has to start as early as possible)
            // the first fragment reads go out BEFORE the refill's DMA instructions: the MFMA pipe idles from the barrier until they
            // return, and six address computations + global_load_lds in front of them lengthen exactly that window
            const f16* xs = smem + buf * (XT + WT);
            const f16* ws = xs + XT;
            f16x8 xf0[4], wf0[4];
#ifdef VLP_NT_DEBUG
            if (p.dbg & 256) {          // no fragment reads at all (with 128: MFMAs + DMA only)
#pragma unroll
                for (int t = 0; t < 4; ++t) { xf0[t] = __builtin_bit_cast(f16x8, (u32x4){(uint32_t)kt, 1u, 2u, 3u}); wf0[t] = xf0[t]; }
            } else
#endif
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                xf0[t] = ld8(xs + xrow[t] * BK + ((g ^ swz_x(xrow[t])) << 3));
                wf0[t] = ld8(ws + wrow[t] * BK + ((g ^ swz_w(wrow[t])) << 3));
            }
            __builtin_amdgcn_sched_barrier(0);
#ifdef VLP_NT_DEBUG
            if (!(p.dbg & 2))
#endif
            glds(min(kt + NS - 1, nk - 1), nbuf);
#ifdef VLP_NT_DEBUG
            if (p.dbg & 1) { buf = (buf + 1 == NS) ? 0 : buf + 1; nbuf = (nbuf + 1 == NS) ? 0 : nbuf + 1; acc[0][0][0] += (float)xf0[0][0] + (float)wf0[0][0]; continue; }
#endif
#ifdef VLP_NT_DEBUG
            if (p.dbg & 128) {          // half the fragment reads: the second half's MFMAs reuse the first half's operands (wrong results, LDS-read cost probe)
#pragma unroll
                for (int rep = 0; rep < 2; ++rep)
#pragma unroll
                    for (int tm = 0; tm < 4; ++tm)
#pragma unroll
                        for (int tn = 0; tn < 4; ++tn)
                            acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf0[tn], xf0[tm], acc[tm][tn], 0, 0, 0);
            } else
#endif
            if (BN_T == 128) {          // 2 waves per SIMD: room for both fragment sets -- the second half's reads fly under the first half's MFMAs
                f16x8 xf1[4], wf1[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    xf1[t] = ld8(xs + xrow[t] * BK + (((4 + g) ^ swz_x(xrow[t])) << 3));
                    wf1[t] = ld8(ws + wrow[t] * BK + (((4 + g) ^ swz_w(wrow[t])) << 3));
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
            } else {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int tm = 0; tm < 4; ++tm)
#pragma unroll
                    for (int tn = 0; tn < 4; ++tn)
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf0[tn], xf0[tm], acc[tm][tn], 0, 0, 0);
                compute_ks(buf, 1);
            }
            buf = (buf + 1 == NS) ? 0 : buf + 1;
            nbuf = (nbuf + 1 == NS) ? 0 : nbuf + 1;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the clamped tail reloads must land before the LDS is released
    }

    NT_TRACE(1);
#ifdef VLP_NT_DEBUG
    if ((p.dbg & 4) && acc[0][0][0] != 12345.678f) return;
#endif
    // ---- epilogue: lane owns row m (per tm) and 16 consecutive n -------------------------------
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
    if constexpr (EPI == 1) {
        if (ncol0 >= p.N) return;
        f16x8 mv[4][2];
#pragma unroll
        for (int tm = 0; tm < 4; ++tm) {
            const int mc = min(m0 + wm * 64 + 16 * tm + li, p.M - 1);
#pragma unroll
            for (int h = 0; h < 2; ++h) mv[tm][h] = ld8(p.mulsrc + (int64_t)mc * p.ldm + ncol0 + 8 * h);
        }
#pragma unroll
        for (int tm = 0; tm < 4; ++tm) {
            const int m = m0 + wm * 64 + 16 * tm + li;
            if (m >= p.M) continue;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                f16x8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int c = 8 * h + j;
                    float vj = acc[tm][c >> 2][c & 3] * p.alpha + bias_v[c];
                    vj *= (float)mv[tm][h][j];
                    o[j] = (f16)vj;
                }
                ST8_OUT(p, p.Y + (int64_t)m * p.ldy + ncol0 + 8 * h, o);
            }
        }
        return;
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
        for (int h = 0; h < 2; ++h)              // two 8-wide vectors through the shared epilogue (bias already added)
            nt_epilogue8<SG>(p, m, ncol0 + 8 * h, v + 8 * h, rkey, true);
    }
#ifdef VLP_NT_DEBUG
    NT_TRACE(2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    NT_TRACE(3);
#endif
}

int vlp_gemm_nt_fill_params(const vlp_gemm_nt_args* a, GemmNtParams& p) {
    VLP_CHECK_ARG(a != nullptr, "vlp_gemm_nt: null args");
    VLP_CHECK_ARG(a->X && a->W && a->Y, "vlp_gemm_nt: null operand");
    VLP_CHECK_ARG(a->M > 0 && a->N > 0 && a->K > 0, "vlp_gemm_nt: bad shape M=%d N=%d K=%d", a->M, a->N, a->K);
    VLP_CHECK_ARG(a->K % BK == 0, "vlp_gemm_nt: K=%d must be a multiple of %d (pad with vlp_copy2d)", a->K, BK);
    VLP_CHECK_ARG(a->ldx % 8 == 0 && a->ldw % 8 == 0 && a->ldy % 8 == 0, "vlp_gemm_nt: leading dims must be multiples of 8 halfs");
    VLP_CHECK_ARG(a->ldx >= a->K && a->ldw >= a->K, "vlp_gemm_nt: ldx/ldw < K");
    const int n8 = (a->N + 7) / 8 * 8;
    VLP_CHECK_ARG(a->ldy >= n8, "vlp_gemm_nt: ldy=%lld < roundup8(N)=%d", (long long)a->ldy, n8);
    VLP_CHECK_ARG(((uintptr_t)a->X | (uintptr_t)a->W | (uintptr_t)a->Y) % 16 == 0, "vlp_gemm_nt: operands must be 16-byte aligned");
    if (a->residual) VLP_CHECK_ARG(a->ldr % 8 == 0 && a->ldr >= n8 && (uintptr_t)a->residual % 16 == 0, "vlp_gemm_nt: bad residual layout");
    if (a->preact) VLP_CHECK_ARG(a->ldp % 8 == 0 && a->ldp >= n8 && (uintptr_t)a->preact % 16 == 0, "vlp_gemm_nt: bad preact layout");
    if (a->mul_mode != VLP_MUL_NONE)
        VLP_CHECK_ARG(a->mul_src && a->ldm % 8 == 0 && a->ldm >= n8 && (uintptr_t)a->mul_src % 16 == 0, "vlp_gemm_nt: bad mul_src layout");
    if (a->bias) VLP_CHECK_ARG((uintptr_t)a->bias % 16 == 0, "vlp_gemm_nt: bias must be 16-byte aligned");
    VLP_CHECK_ARG(a->act >= VLP_ACT_NONE && a->act <= VLP_ACT_GELU_SAVE_GRAD, "vlp_gemm_nt: bad act %d", a->act);
    VLP_CHECK_ARG(a->act != VLP_ACT_GELU_SAVE_GRAD || a->preact, "vlp_gemm_nt: VLP_ACT_GELU_SAVE_GRAD needs `preact` (receives gelu'(z))");
    VLP_CHECK_ARG(a->mul_mode >= VLP_MUL_NONE && a->mul_mode <= VLP_MUL_PLAIN, "vlp_gemm_nt: bad mul_mode %d", a->mul_mode);
    VLP_CHECK_ARG(a->dropout_p >= 0.f && a->dropout_p < 1.f, "vlp_gemm_nt: bad dropout p");

    p.X = (const f16*)a->X; p.ldx = a->ldx;
    p.W = (const f16*)a->W; p.ldw = a->ldw;
    p.Y = (f16*)a->Y; p.ldy = a->ldy;
    p.bias = (const f16*)a->bias;
    p.residual = (const f16*)a->residual; p.ldr = a->ldr;
    p.preact = (f16*)a->preact; p.ldp = a->ldp;
    p.mulsrc = (const f16*)a->mul_src; p.ldm = a->ldm;
    p.M = a->M; p.N = a->N; p.K = a->K;
    p.act = a->act; p.mulmode = a->mul_mode;
    p.alpha = a->alpha;
    p.drop = make_drop(a->dropout_p, a->seed, a->rng_stream);
    p.row_map = a->row_map;
    p.tiles_n = 0;
    p.xcd_remap = 0;
#ifdef VLP_NT_DEBUG
    { const char* e = getenv("VLP_NT_DEBUG"); p.dbg = e ? atoi(e) : 0; }
#endif
    return VLP_OK;
}

// the variant the calling thread's last vlp_gemm_nt actually launched (after the fallbacks below): lets a test assert that a forced
// variant is the kernel that ran (tests/test_00_kernels_gpu.py::test_gemm_nt_variant_identity)
static thread_local int t_last_variant = -1;
extern "C" int vlp_gemm_nt_resolved_variant(void) { return t_last_variant; }

extern "C" int vlp_gemm_nt(const vlp_gemm_nt_args* a, void* stream) {
    VLP_ENTER(a ? a->X : nullptr, "vlp_gemm_nt");        // the guard lives until the launch below has been issued
    GemmNtParams p;
    const int frc = vlp_gemm_nt_fill_params(a, p);
    if (frc != VLP_OK) return frc;
    hipStream_t s = (hipStream_t)stream;
#define LAUNCH_NT_(V, BMT, BNT, NBUF, SGV, NSRV)                                                                  \
    do {                                                                                                                \
        const size_t smem = (size_t)(NBUF) * ((BMT) + (BNT)) * BK * sizeof(f16);                                        \
        VLP_ONCE_PER_DEVICE((void)hipFuncSetAttribute((const void*)gemm_nt_kernel<V, BMT, BNT, SGV, NSRV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));                                                            \
        p.tiles_n = cdiv(a->N, (BNT));                                                                                  \
        hipLaunchKernelGGL((gemm_nt_kernel<V, BMT, BNT, SGV, NSRV>), dim3(cdiv(a->M, (BMT)) * p.tiles_n), dim3(((BMT) / 64) * ((BNT) / 64) * 64), smem, s, p); \
    } while (0)
#define LAUNCH_NT(V, BMT, BNT, NBUF) \
    do { if (sg) LAUNCH_NT_(V, BMT, BNT, NBUF, true, 0); else LAUNCH_NT_(V, BMT, BNT, NBUF, false, 0); } while (0)
    /* ring kernels with an explicit stage count (NBUF = stages) */
#define LAUNCH_RING(BMT, BNT, NSV) \
    do { if (sg) LAUNCH_NT_(3, BMT, BNT, NSV, true, NSV); else LAUNCH_NT_(3, BMT, BNT, NSV, false, NSV); } while (0)
    const bool sg = a->act == VLP_ACT_GELU_SAVE_GRAD;
    int variant = a->variant;
    // the persistent k-stream kernel carries three epilogues on N % 128 == 0, K > 512: anything else runs on the rings
    if ((variant & 256) && !vlp_gemm_nt_ps_eligible(p, sg)) variant = (a->N > 1024) ? 29 : 27;
    // the wave-pipelined family carries the light epilogues + save-grad GeLU; anything else (erf / tanh in the epilogue) runs on the rings
    if ((variant & 64) && !sg && !nt_epilogue_is_light(p)) variant = (a->N > 1024) ? 29 : 27;
    // the phased kernels (6, 7) do not instantiate the save-grad epilogue: same fallback instead of an error for a table / override entry
    if (sg && !(variant & 64) && ((variant & 7) == 6 || (variant & 7) == 7)) variant = (a->N > 1024) ? 29 : 27;
    // the wave-pipelined kernels address their operands with 32-bit lane offsets: a problem beyond that falls back to the rings too
    if ((variant & 64) && ((int64_t)p.M * p.ldx >= (1ll << 31) || (int64_t)p.N * p.ldw >= (1ll << 31))) variant = (a->N > 1024) ? 29 : 27;
#ifndef VLP_LAB_BUILD
    // investigation variants are not in the product library (tools/build_variant_lib.sh <out.so> -DVLP_LAB_BUILD builds them): the phased
    // kernels (6 / 7), the k32 ring (53 / 61) and the wave-pipelined configurations other than 1 / 5 run on the rings instead
    if (!(variant & (64 | 256)) && ((variant & 7) == 6 || (variant & 7) == 7 || ((variant & 7) == 5 && (variant & 48) == 48))) variant = (a->N > 1024) ? 29 : 27;
    if ((variant & 64) && !(variant & 256)) {
        const int cfg = (variant & 7) + ((variant & 128) ? 8 : 0);
        if (cfg != 1 && cfg != 5) variant = (a->N > 1024) ? 29 : 27;
    }
#endif
    t_last_variant = variant;
    if (sg) {
        VLP_CHECK_ARG(!a->residual && a->mul_mode == VLP_MUL_NONE && a->dropout_p == 0.f,
                      "vlp_gemm_nt: VLP_ACT_GELU_SAVE_GRAD fuses bias + gelu + derivative only (no residual / multiplier / dropout)");
    }
    p.xcd_remap = (variant & 8) ? 1 : 0;
    if (variant & 256) {         // persistent k-stream kernel (gemm_nt_ps.hip): 256 (+ 8 = XCD-aware run order)
        const int rc = vlp_gemm_nt_ps_launch(p, sg, s);
        if (rc != VLP_OK) return rc;
        VLP_CHECK_LAUNCH("vlp_gemm_nt");
        return VLP_OK;
    }
    if (variant & 64) {          // wave-pipelined family (gemm_nt_wp.hip): 64 + cfg (+ 8 = XCD-aware tile order)
        const int rc = vlp_gemm_nt_wp_launch(p, (variant & 7) + ((variant & 128) ? 8 : 0), sg, s);
        if (rc != VLP_OK) return rc;
        VLP_CHECK_LAUNCH("vlp_gemm_nt");
        return VLP_OK;
    }
    switch (variant & 7) {
#ifdef VLP_LAB_BUILD
        case 6: case 7: {
            const int rc = vlp_gemm_nt_ph_launch(p, (variant & 7) == 6 ? 256 : 128, (variant >> 4) & 3, s);
            if (rc != VLP_OK) return rc;
            break;
        }
#endif
        case 0: LAUNCH_NT(0, 128, 128, 2); break;
        case 1: if (variant & 16) LAUNCH_NT(3, 128, 128, 4); else LAUNCH_NT(1, 128, 128, 2); break;
        case 3:
            if (variant & 16) LAUNCH_NT(3, 256, 128, 3);
            else LAUNCH_NT(1, 256, 128, 2);
            break;
        case 2: LAUNCH_NT(2, 128, 128, 1); break;
        case 4: LAUNCH_NT(2, 256, 128, 1); break;
        case 5:
#ifdef VLP_LAB_BUILD
            if ((variant & 48) == 48) { const int rc = vlp_gemm_nt_k32_launch(p, sg, s); if (rc != VLP_OK) return rc; }      /* 53 / 61: k tiles of 32, 4 stages */
            else
#endif
            if (variant & 16) {
                // the stored-tensor multiply alone (FFN-down dgrad): lean epilogue instantiation (VLP_NT_LEAN_EPI=0: the shared one, for A/B runs)
                static const int lean_on = [] { const char* e = getenv("VLP_NT_LEAN_EPI"); return (e && e[0] == '0') ? 0 : 1; }();
                const bool lean = lean_on && !sg && p.act == VLP_ACT_NONE && !p.preact && p.mulmode == VLP_MUL_PLAIN && !p.drop.thresh && !p.residual &&
                                  p.N % 16 == 0 && p.ldm % 8 == 0 && ((uintptr_t)p.mulsrc & 15) == 0;
                if (lean) {
                    const size_t smem = (size_t)2 * (256 + 256) * BK * sizeof(f16);
                    VLP_ONCE_PER_DEVICE((void)hipFuncSetAttribute((const void*)gemm_nt_kernel<3, 256, 256, false, 2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
                    p.tiles_n = cdiv(a->N, 256);
                    hipLaunchKernelGGL((gemm_nt_kernel<3, 256, 256, false, 2, 1>), dim3(cdiv(a->M, 256) * p.tiles_n), dim3(1024), smem, s, p);
                } else {
                    LAUNCH_RING(256, 256, 2);
                }
            }
            else LAUNCH_NT(1, 256, 256, 2);
            break;
        default: LAUNCH_NT(1, 128, 128, 2); break;
    }
#undef LAUNCH_RING
#undef LAUNCH_NT
#undef LAUNCH_NT_
    VLP_CHECK_LAUNCH("vlp_gemm_nt");
    return VLP_OK;
}

// split-K form for skinny M (gemm_nt_splitk.hip); same arguments, plus the slice count and an fp32 workspace
extern "C" int vlp_gemm_nt_splitk(const vlp_gemm_nt_args* a, int32_t splits, void* workspace, int64_t workspace_bytes, void* stream) {
    VLP_ENTER(a ? a->X : nullptr, "vlp_gemm_nt_splitk");
    GemmNtParams p;
    const int frc = vlp_gemm_nt_fill_params(a, p);
    if (frc != VLP_OK) return frc;
    VLP_CHECK_ARG(a->act != VLP_ACT_GELU_SAVE_GRAD, "vlp_gemm_nt_splitk: VLP_ACT_GELU_SAVE_GRAD is not provided (training-shape kernels only)");
    return vlp_gemm_nt_splitk_launch(p, splits, (float*)workspace, workspace_bytes, (hipStream_t)stream);
}
