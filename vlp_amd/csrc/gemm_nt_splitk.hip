// Split-K NT GEMM for skinny problems (incremental decoding: M = batch * 2 = 128..640 rows):
//     Y[M,N] = epi( alpha * X[M,K] . W[N,K]^T ),  same contract and epilogue as vlp_gemm_nt.
// A 128-row problem has only N/128 output tiles (6 for the 768-wide projections): the ordinary kernels would run 6 workgroups, each
// streaming its 128 x K weight panel through one CU at one L2/HBM round trip per k tile (measured 50 us for 128x768x3072, i.e. 1 us per k tile,
// on 2 % of the chip).  Here the k range is cut into `splits` slices: (tiles x splits) workgroups write fp32 partial tiles to a
// slab, a second kernel sums the slabs in a fixed order and applies the fused epilogue (bias / activation / residual / ...).
// No atomics: deterministic.  Tile 128x128x64, 4 waves, LDS-DMA double buffer -- the staging / fragment layout of gemm_nt.hip.
#include "common.h"
#include "gemm_nt.h"
#include "gemm_nt_epilogue.h"

#define SK_BM 128
#define SK_BK 64

DEVFN int sk_swz_x(int r) { return r & 7; }
DEVFN int sk_swz_w(int r) { return (((r >> 4) & 3) << 1) | ((r >> 1) & 1); }

struct SplitKParams {
    GemmNtParams g;
    float* slab;           // [splits][M][ldslab] fp32
    int64_t ldslab;        // roundup8(N)
    int splits, kt_per_split;
};

__global__ __launch_bounds__(256, 2) void gemm_nt_splitk_kernel(SplitKParams q) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    f16* smem = reinterpret_cast<f16*>(smem_raw);
    const GemmNtParams& p = q.g;
    constexpr int T = 256, RPP = T / 8, XP = SK_BM / RPP, WP = SK_BM / RPP, XT = SK_BM * SK_BK, WT = SK_BM * SK_BK;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 1, wn = wid & 1;
    const int g = lane >> 4, li = lane & 15;
    const int split = blockIdx.x % q.splits;
    const int tile = blockIdx.x / q.splits;
    const int tile_m = tile / p.tiles_n, tile_n = tile % p.tiles_n;
    const int m0 = tile_m * SK_BM, n0 = tile_n * SK_BM;
    const int nk_all = p.K / SK_BK;
    const int kt0 = split * q.kt_per_split, kt1 = min(nk_all, kt0 + q.kt_per_split);

    const int srow = tid >> 3, sx = tid & 7;
    const f16* xsrc[XP];
    const f16* wsrc[WP];
#pragma unroll
    for (int i = 0; i < XP; ++i) {
        const int r = srow + RPP * i;
        xsrc[i] = p.X + (int64_t)min(m0 + r, p.M - 1) * p.ldx + (sx ^ sk_swz_x(r)) * 8;
    }
#pragma unroll
    for (int i = 0; i < WP; ++i) {
        const int r = srow + RPP * i;
        wsrc[i] = p.W + (int64_t)min(n0 + r, p.N - 1) * p.ldw + (sx ^ sk_swz_w(r)) * 8;
    }
    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int xrow[4], wrow[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        xrow[t] = wm * 64 + 16 * t + li;
        wrow[t] = wn * 64 + 16 * (li >> 2) + 4 * t + (li & 3);      // permuted: a lane ends up with 16 consecutive n
    }
    auto glds = [&](int kt, int buf) {
        f16* xs = smem + buf * (XT + WT);
        f16* ws = xs + XT;
#pragma unroll
        for (int i = 0; i < XP; ++i)
            glds16(xsrc[i] + (int64_t)kt * SK_BK, lds_addr_of(xs + (RPP * i + 8 * wid) * SK_BK));
#pragma unroll
        for (int i = 0; i < WP; ++i)
            glds16(wsrc[i] + (int64_t)kt * SK_BK, lds_addr_of(ws + (RPP * i + 8 * wid) * SK_BK));
    };
    auto compute = [&](int buf) {
        const f16* xs = smem + buf * (XT + WT);
        const f16* ws = xs + XT;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int c = ks * 4 + g;
            f16x8 xf[4], wf[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                xf[t] = ld8(xs + xrow[t] * SK_BK + ((c ^ sk_swz_x(xrow[t])) << 3));
                wf[t] = ld8(ws + wrow[t] * SK_BK + ((c ^ sk_swz_w(wrow[t])) << 3));
            }
#pragma unroll
            for (int tm = 0; tm < 4; ++tm)
#pragma unroll
                for (int tn = 0; tn < 4; ++tn) acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[tn], xf[tm], acc[tm][tn], 0, 0, 0);
        }
    };
    if (kt0 < kt1) {
        glds(kt0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int kt = kt0; kt < kt1; ++kt) {
            const int buf = (kt - kt0) & 1;
            if (kt + 1 < kt1) glds(kt + 1, buf ^ 1);
            compute(buf);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    }
    // partial tile -> slab[split]: lane owns row m (per tm) and 16 consecutive n
    float* dst = q.slab + (int64_t)split * p.M * q.ldslab;
    const int ncol0 = n0 + wn * 64 + 16 * g;
#pragma unroll
    for (int tm = 0; tm < 4; ++tm) {
        const int m = m0 + wm * 64 + 16 * tm + li;
        if (m >= p.M) continue;
#pragma unroll
        for (int tn = 0; tn < 4; ++tn) {
            const int nc = ncol0 + 4 * tn;
            if (nc < q.ldslab) *reinterpret_cast<f32x4*>(dst + (int64_t)m * q.ldslab + nc) = acc[tm][tn];
        }
    }
}

// Y[m, nc..nc+7] = epilogue( alpha * sum_s slab[s][m][nc..] )
__global__ __launch_bounds__(256) void gemm_nt_splitk_reduce_kernel(SplitKParams q) {
    const GemmNtParams& p = q.g;
    const int nv = (int)(q.ldslab >> 3);
    const int64_t total = (int64_t)p.M * nv;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int m = (int)(i / nv), nc = (int)(i % nv) * 8;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = 0.f;
        for (int s = 0; s < q.splits; ++s) {
            const float* src = q.slab + ((int64_t)s * p.M + m) * q.ldslab + nc;
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(src), a1 = *reinterpret_cast<const f32x4*>(src + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] += a0[e]; v[4 + e] += a1[e]; }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= p.alpha;
        const uint32_t rkey = p.drop.thresh ? drop_rowkey(p.drop, nt_drop_row(p, m)) : 0u;
        nt_epilogue8(p, m, nc, v, rkey);
    }
}

extern "C" int64_t vlp_gemm_nt_splitk_workspace_bytes(int32_t M, int32_t N, int32_t splits) {
    return (int64_t)splits * M * ((N + 7) / 8 * 8) * (int64_t)sizeof(float);
}

int vlp_gemm_nt_splitk_launch(GemmNtParams& p, int splits, float* workspace, int64_t workspace_bytes, hipStream_t s) {
    const int nk = p.K / SK_BK;
    if (splits > nk) splits = nk;
    VLP_CHECK_ARG(splits >= 1 && splits <= 64, "vlp_gemm_nt_splitk: splits must be in [1, 64]");
    const int64_t need = vlp_gemm_nt_splitk_workspace_bytes(p.M, p.N, splits);
    if (!workspace || workspace_bytes < need || ((uintptr_t)workspace & 15))
        return vlp_set_error(VLP_ERR_WORKSPACE, "vlp_gemm_nt_splitk: workspace %lld < %lld (or not 16-byte aligned)", (long long)workspace_bytes, (long long)need);
    SplitKParams q;
    q.g = p;
    q.g.tiles_n = cdiv(p.N, SK_BM);
    q.slab = workspace;
    q.ldslab = (p.N + 7) / 8 * 8;
    q.splits = splits;
    q.kt_per_split = cdiv(nk, splits);
    q.splits = cdiv(nk, q.kt_per_split);                 // no empty slices
    const size_t smem = (size_t)2 * 2 * SK_BM * SK_BK * sizeof(f16);
    VLP_ONCE_PER_DEVICE((void)hipFuncSetAttribute((const void*)gemm_nt_splitk_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipLaunchKernelGGL(gemm_nt_splitk_kernel, dim3(cdiv(p.M, SK_BM) * q.g.tiles_n * q.splits), dim3(256), smem, s, q);
    VLP_CHECK_LAUNCH("vlp_gemm_nt_splitk");
    const int64_t total = (int64_t)p.M * (q.ldslab / 8);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(gemm_nt_splitk_reduce_kernel, dim3(blocks), dim3(256), 0, s, q);
    VLP_CHECK_LAUNCH("vlp_gemm_nt_splitk_reduce");
    return VLP_OK;
}
