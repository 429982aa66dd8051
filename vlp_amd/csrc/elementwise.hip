// Memory-bound glue kernels of the VLP hot path for gfx950 (16-byte vector accesses where layout permits).
#include <hip/hip_fp16.h>
#include "common.h"

// =================================================================================================
// Embedding splice, modeling.py:217-236
// =================================================================================================
__global__ __launch_bounds__(256) void embed_fwd_kernel(vlp_embed_fwd_args a) {
    const int nch = a.H >> 3;
    const int64_t total = (a.row_map ? (int64_t)a.rows : (int64_t)a.B * a.L) * nch;
    const f16* word = (const f16*)a.word_emb;
    const f16* pos = (const f16*)a.pos_emb;
    const f16* typ = (const f16*)a.type_emb;
    const f16* vis = (const f16*)a.vis_h;
    const f16* vpe = (const f16*)a.vispe_h;
    f16* pre = (f16*)a.pre;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % nch);
        const int64_t prow = i / nch;                                   // output row (packed layout: row_map gives its logical index)
        const int64_t row = a.row_map ? (int64_t)a.row_map[prow] : prow;
        const int l = (int)(row % a.L);
        const int b = (int)(row / a.L);
        f16x8 w, p;
        if (l >= 1 && l <= a.Nv) {
            const int64_t vr = (int64_t)b * a.Nv + (l - 1);
            if (a.region_mask && a.region_mask[vr]) {          // masked region: zeros in the word and the position stream (modeling.py:1054-1055)
#pragma unroll
                for (int e = 0; e < 8; ++e) { w[e] = (f16)0.f; p[e] = (f16)0.f; }
            } else {
                w = ld8(vis + vr * a.H + c * 8);
                p = ld8(vpe + vr * a.H + c * 8);
            }
        } else {
            int64_t id = a.input_ids[row];
            id = id < 0 ? 0 : (id >= a.vocab ? a.vocab - 1 : id);
            w = ld8(word + id * a.H + c * 8);
            int64_t pi = a.position_ids ? a.position_ids[row] : (int64_t)l;
            pi = pi < 0 ? 0 : (pi >= a.max_pos ? a.max_pos - 1 : pi);
            p = ld8(pos + pi * a.H + c * 8);
        }
        int64_t sg = a.segment_ids[row];
        sg = sg < 0 ? 0 : (sg >= a.type_vocab ? a.type_vocab - 1 : sg);
        f16x8 t = ld8(typ + sg * a.H + c * 8), o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (f16)((float)w[e] + (float)p[e] + (float)t[e]);
        st8(pre + prow * a.H + c * 8, o);
    }
}
extern "C" int vlp_embed_fwd(const vlp_embed_fwd_args* a, void* stream) {
    VLP_CHECK_ARG(a && a->input_ids && a->segment_ids && a->word_emb && a->pos_emb && a->type_emb && a->pre, "vlp_embed_fwd: null operand");
    VLP_ENTER(a->pre, "vlp_embed_fwd");
    VLP_CHECK_ARG(a->H % 8 == 0 && a->B > 0 && a->L > 0 && a->Nv >= 0 && a->Nv + 1 < a->L + 1, "vlp_embed_fwd: bad shape");
    VLP_CHECK_ARG(a->Nv == 0 || (a->vis_h && a->vispe_h), "vlp_embed_fwd: region rows need vis_h / vispe_h");
    VLP_CHECK_ARG(a->row_map == nullptr || (a->rows > 0 && (int64_t)a->rows <= (int64_t)a->B * a->L), "vlp_embed_fwd: packed output needs 0 < rows <= B*L");
    const int64_t total = (a->row_map ? (int64_t)a->rows : (int64_t)a->B * a->L) * (a->H / 8);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(embed_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, *a);
    VLP_CHECK_LAUNCH("vlp_embed_fwd");
    return VLP_OK;
}

// backward.  (1) one pass over dpre: region rows -> d_vis_h / d_vispe_h (through ReLU + dropout); (1b) token rows -> d_word_emb by
// embed_word_bwd_kernel (one owner per distinct id, row-ordered fp32 sums); (2) position table: deterministic sum over the batch per
// token position; (3) type table: masked column sums (one register accumulator per type) over row splits + a small reduce.
// No atomics anywhere: the whole backward is bitwise reproducible.
#define EMB_TSPLITS 64
#define EMB_MAXT 8
__global__ __launch_bounds__(256) void embed_bwd_kernel(vlp_embed_bwd_args a, DropCtx dvis, DropCtx dvpe) {
    const int nch = a.H >> 3;
    const int64_t total = (int64_t)a.B * a.L * nch;
    const f16* dpre = (const f16*)a.dpre;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % nch);
        const int64_t row = i / nch;
        const int l = (int)(row % a.L);
        const int b = (int)(row / a.L);
        const f16x8 d = ld8(dpre + row * a.H + c * 8);
        if (l >= 1 && l <= a.Nv) {
            const int64_t vr = (int64_t)b * a.Nv + (l - 1);
            if (a.region_mask && a.region_mask[vr]) continue;  // a masked region fed zeros into the encoder: its rows belong to vlp_pretext_bwd
            const f16x8 yv = ld8((const f16*)a.vis_h + vr * a.H + c * 8);
            const f16x8 yp = ld8((const f16*)a.vispe_h + vr * a.H + c * 8);
            f16x8 ov, op;
            const uint32_t kv = dvis.thresh ? drop_rowkey(dvis, (uint64_t)vr) : 0u;   // (row of the [B*Nv, H] projection, col)
            const uint32_t kp = dvpe.thresh ? drop_rowkey(dvpe, (uint64_t)vr) : 0u;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const uint32_t col = (uint32_t)(c * 8 + e);
                // y > 0 implies the ReLU was active AND the element survived dropout
                float gv = ((float)yv[e] > 0.f) ? (float)d[e] : 0.f;
                float gp = ((float)yp[e] > 0.f) ? (float)d[e] : 0.f;
                if (dvis.thresh) gv *= drop_mult(dvis, kv, col);
                if (dvpe.thresh) gp *= drop_mult(dvpe, kp, col);
                ov[e] = (f16)gv;
                op[e] = (f16)gp;
            }
            st8((f16*)a.d_vis_h + vr * a.H + c * 8, ov);
            st8((f16*)a.d_vispe_h + vr * a.H + c * 8, op);
        }
        // token rows: see embed_word_bwd_kernel (deterministic, no atomics)
    }
}
// d_word_emb[id] += sum of dpre over the token rows (l == 0 or l > Nv) whose input id is `id` -- without atomics, in a fixed order, so the
// result is bitwise reproducible (the reference's nn.Embedding backward on the GPU is not).  Token rows are enumerated as
// t = b*(L-Nv) + j (j = 0 -> l = 0, j >= 1 -> l = Nv + j); rows sharing an id form a chain ordered by t.
//   pass 1 (one workgroup per token row): rank = number of earlier rows with the same id; every EWB_G-th row of a chain is a group
//           leader and writes the fp32 sum of its group (itself + the next EWB_G-1 chain members, in order) to partial[t];
//   pass 2 (one workgroup per token row, only chain heads act): sums the leaders' partials of its chain in order and adds them to
//           d_word_emb[id] (single writer per id).
// Long chains ([PAD] rows: ~2000 at B = 64) are thereby summed by ~60 workgroups in parallel instead of one.
#define EWB_G 32
#define EWB_LIST 1024
struct EwbGeom {
    const int64_t* ids;
    int L, Nv, T, NT, vocab;
    DEVFN int64_t row_of(int u) const { const int b = u / T, j = u - b * T; return (int64_t)b * L + (j == 0 ? 0 : Nv + j); }
    DEVFN int id_of(int u) const { const int64_t id = ids[row_of(u)]; return (int)(id < 0 ? 0 : (id >= vocab ? vocab - 1 : id)); }
};
// ordered compaction of { u in [u0, min(u0+256, NT)) : pred(u) } behind list[n..]; returns the new n (uniform).  256 threads.
template <typename Pred>
DEVFN int ewb_compact(int* list, int* wcnt, int n, int u0, int NT, Pred pred) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int u = u0 + tid;
    const bool m = u < NT && pred(u);
    const uint64_t bal = __ballot(m);
    if (lane == 0) wcnt[wv] = __popcll(bal);
    __syncthreads();
    int off = n;
    for (int w = 0; w < wv; ++w) off += wcnt[w];
    if (m) list[off + __popcll(bal & ((1ull << lane) - 1ull))] = u;
    n += wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
    __syncthreads();
    return n;
}
__global__ __launch_bounds__(256) void embed_word_partial_kernel(EwbGeom g, const f16* dpre, int H, int* rank, float* partial) {
    __shared__ int list[256 + EWB_G];
    __shared__ int wcnt[4];
    __shared__ float wsum[4];
    const int tid = threadIdx.x, t = blockIdx.x;
    const int id = g.id_of(t);
    float c = 0.f;
    for (int u = tid; u < t; u += 256) c += (g.id_of(u) == id) ? 1.f : 0.f;
    c = wave_sum(c);
    if ((tid & 63) == 0) wsum[tid >> 6] = c;
    __syncthreads();
    const int r = (int)(wsum[0] + wsum[1] + wsum[2] + wsum[3]);
    if (tid == 0) rank[t] = r;
    if (r % EWB_G) return;
    int n = 0;
    for (int u0 = t; u0 < g.NT && n < EWB_G; u0 += 256) n = ewb_compact(list, wcnt, n, u0, g.NT, [&](int u) { return g.id_of(u) == id; });
    n = min(n, EWB_G);
    const int nch = H >> 3;
    if (tid < nch) {
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
        int i = 0;
        for (; i + 4 <= n; i += 4) {
            f16x8 d[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) d[k] = ld8(dpre + g.row_of(list[i + k]) * H + tid * 8);
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += (float)d[k][e];
        }
        for (; i < n; ++i) {
            const f16x8 d = ld8(dpre + g.row_of(list[i]) * H + tid * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += (float)d[e];
        }
        float* dst = partial + (int64_t)t * H + tid * 8;
        *reinterpret_cast<f32x4*>(dst) = (f32x4){acc[0], acc[1], acc[2], acc[3]};
        *reinterpret_cast<f32x4*>(dst + 4) = (f32x4){acc[4], acc[5], acc[6], acc[7]};
    }
}
__global__ __launch_bounds__(256) void embed_word_reduce_kernel(EwbGeom g, int H, const int* rank, const float* partial, f16* dword) {
    __shared__ int list[EWB_LIST + 256];
    __shared__ int wcnt[4];
    const int tid = threadIdx.x, t = blockIdx.x;
    if (rank[t] != 0) return;                            // only the first row of a chain owns the id
    const int id = g.id_of(t);
    const int nch = H >> 3;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    int start = t;
    while (start < g.NT) {
        int n = 0, u0 = start;
        for (; u0 < g.NT && n < EWB_LIST; u0 += 256)
            n = ewb_compact(list, wcnt, n, u0, g.NT, [&](int u) { return (rank[u] % EWB_G) == 0 && g.id_of(u) == id; });
        start = u0;
        if (tid < nch) {
            for (int i = 0; i < n; ++i) {
                const float* src = partial + (int64_t)list[i] * H + tid * 8;
                const f32x4 a0 = *reinterpret_cast<const f32x4*>(src), a1 = *reinterpret_cast<const f32x4*>(src + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { acc[e] += a0[e]; acc[4 + e] += a1[e]; }
            }
        }
        __syncthreads();
    }
    if (tid < nch) {
        f16* dst = dword + (int64_t)id * H + tid * 8;
        f16x8 o = ld8(dst);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (f16)((float)o[e] + acc[e]);
        st8(dst, o);
    }
}
// d_pos_emb[l] += sum_b dpre[b,l]  for token positions (l == 0 or l > Nv)
__global__ void embed_bwd_pos_kernel(const f16* dpre, f16* dpos, int B, int L, int Nv, int H) {
    const int nch = H >> 3;
    const int64_t total = (int64_t)L * nch;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % nch);
        const int l = (int)(i / nch);
        if (l >= 1 && l <= Nv) continue;
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
        for (int b = 0; b < B; ++b) {
            const f16x8 d = ld8(dpre + ((int64_t)b * L + l) * H + c * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += (float)d[e];
        }
        f16* dst = dpos + (int64_t)l * H + c * 8;
        f16x8 o = ld8(dst);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (f16)((float)o[e] + acc[e]);
        st8(dst, o);
    }
}
// part[split][t][H] = sum over the split's rows with segment t of dpre
__global__ __launch_bounds__(256) void embed_bwd_type_kernel(const f16* dpre, const int64_t* seg, float* part, int64_t rows, int H, int T) {
    __shared__ float red[8][256];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int c0 = blockIdx.x * 256 + tx * 8;
    const int64_t rows_per = (rows + gridDim.y - 1) / gridDim.y;
    const int64_t r0 = blockIdx.y * rows_per, r1 = min(rows, r0 + rows_per);
    float acc[EMB_MAXT][8];
#pragma unroll
    for (int t = 0; t < EMB_MAXT; ++t)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[t][e] = 0.f;
    if (c0 < H) {
        for (int64_t r = r0 + ty; r < r1; r += 8) {
            int64_t sg = seg[r];
            sg = sg < 0 ? 0 : (sg >= T ? T - 1 : sg);
            const f16x8 d = ld8(dpre + r * H + c0);
#pragma unroll
            for (int t = 0; t < EMB_MAXT; ++t) {
                const float m = (sg == t) ? 1.f : 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[t][e] += m * (float)d[e];
            }
        }
    }
#pragma unroll
    for (int t = 0; t < EMB_MAXT; ++t) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 8; ++e) red[ty][tx * 8 + e] = acc[t][e];
        __syncthreads();
        float s = 0.f;
#pragma unroll
        for (int y = 0; y < 8; ++y) s += red[y][threadIdx.x];
        const int col = blockIdx.x * 256 + threadIdx.x;
        if (col < H && t < T) part[((int64_t)blockIdx.y * EMB_MAXT + t) * H + col] = s;
    }
}
__global__ void embed_bwd_type_reduce_kernel(const float* part, int nsplits, f16* dtyp, int T, int H) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= T * H) return;
    const int t = i / H, c = i % H;
    float s = 0.f;
    for (int p = 0; p < nsplits; ++p) s += part[((int64_t)p * EMB_MAXT + t) * H + c];
    dtyp[i] = (f16)((float)dtyp[i] + s);
}
extern "C" int64_t vlp_embed_bwd_workspace_floats(int32_t B, int32_t L, int32_t Nv, int32_t H) {
    const int64_t nt = (int64_t)B * (L - Nv);
    return (int64_t)EMB_TSPLITS * EMB_MAXT * H + nt * H + nt;
}
extern "C" int vlp_embed_bwd(const vlp_embed_bwd_args* a, void* stream) {
    VLP_CHECK_ARG(a && a->dpre && a->input_ids && a->segment_ids && a->d_word_emb && a->d_pos_emb && a->d_type_emb && a->acc32, "vlp_embed_bwd: null operand");
    VLP_ENTER(a->dpre, "vlp_embed_bwd");
    VLP_CHECK_ARG(a->H % 8 == 0 && a->H <= 2048 && a->B > 0 && a->L > a->Nv && a->type_vocab >= 1 && a->type_vocab <= EMB_MAXT,
                  "vlp_embed_bwd: bad shape (H % 8 == 0, H <= 2048, L > Nv, type_vocab <= 8)");
    VLP_CHECK_ARG(a->Nv == 0 || (a->vis_h && a->vispe_h && a->d_vis_h && a->d_vispe_h), "vlp_embed_bwd: region buffers");
    hipStream_t s = (hipStream_t)stream;
    const int64_t total = (int64_t)a->B * a->L * (a->H / 8);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    VLP_CHECK_ARG(a->parts >= 0 && a->parts <= 2, "vlp_embed_bwd: parts must be 0 (all), 1 (region rows) or 2 (tables)");
    if (a->parts != 2 && a->Nv > 0) {
        hipLaunchKernelGGL(embed_bwd_kernel, dim3(blocks), dim3(256), 0, s, *a, make_drop(a->drop_p, a->seed, a->vis_stream),
                           make_drop(a->drop_p, a->seed, a->vispe_stream));
        VLP_CHECK_LAUNCH("vlp_embed_bwd");
    }
    if (a->parts == 1) return VLP_OK;
    {
        EwbGeom g;
        g.ids = a->input_ids; g.L = a->L; g.Nv = a->Nv; g.T = a->L - a->Nv; g.NT = a->B * g.T; g.vocab = a->vocab;
        float* partial = a->acc32 + (int64_t)EMB_TSPLITS * EMB_MAXT * a->H;                 // behind the type-table partials
        int* rank = reinterpret_cast<int*>(partial + (int64_t)g.NT * a->H);
        hipLaunchKernelGGL(embed_word_partial_kernel, dim3(g.NT), dim3(256), 0, s, g, (const f16*)a->dpre, a->H, rank, partial);
        VLP_CHECK_LAUNCH("vlp_embed_word_partial");
        hipLaunchKernelGGL(embed_word_reduce_kernel, dim3(g.NT), dim3(256), 0, s, g, a->H, (const int*)rank, (const float*)partial, (f16*)a->d_word_emb);
        VLP_CHECK_LAUNCH("vlp_embed_word_reduce");
    }
    hipLaunchKernelGGL(embed_bwd_pos_kernel, dim3(cdiv((int64_t)a->L * (a->H / 8), 256)), dim3(256), 0, s, (const f16*)a->dpre,
                       (f16*)a->d_pos_emb, a->B, a->L, a->Nv, a->H);
    VLP_CHECK_LAUNCH("vlp_embed_bwd_pos");
    const int64_t rows = (int64_t)a->B * a->L;
    int splits = rows >= EMB_TSPLITS * 8 ? EMB_TSPLITS : (int)((rows + 7) / 8);
    hipLaunchKernelGGL(embed_bwd_type_kernel, dim3(cdiv(a->H, 256), splits), dim3(256), 0, s, (const f16*)a->dpre, a->segment_ids, a->acc32,
                       rows, a->H, a->type_vocab);
    VLP_CHECK_LAUNCH("vlp_embed_bwd_type");
    hipLaunchKernelGGL(embed_bwd_type_reduce_kernel, dim3(cdiv((int64_t)a->type_vocab * a->H, 256)), dim3(256), 0, s, a->acc32, splits,
                       (f16*)a->d_type_emb, a->type_vocab, a->H);
    VLP_CHECK_LAUNCH("vlp_embed_bwd_type_reduce");
    return VLP_OK;
}

// =================================================================================================
// copy2d (pad / crop / cast / accumulate) and transpose
// =================================================================================================
__global__ void copy2d_kernel(const void* src, int64_t lds, int src_f32, f16* dst, int64_t ldd, int rows, int cols_src, int cols_dst, int beta) {
    const int64_t total = (int64_t)rows * cols_dst;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / cols_dst;
        const int c = (int)(i % cols_dst);
        float v = 0.f;
        if (c < cols_src) v = src_f32 ? ((const float*)src)[r * lds + c] : (float)((const f16*)src)[r * lds + c];
        f16* d = dst + r * ldd + c;
        *d = (f16)(beta ? (float)*d + v : v);
    }
}
extern "C" int vlp_copy2d(const void* src, int64_t lds, int32_t src_f32, void* dst, int64_t ldd, int32_t rows, int32_t cols_src,
                          int32_t cols_dst, int32_t beta, void* stream) {
    VLP_CHECK_ARG(src && dst && rows > 0 && cols_src > 0 && cols_dst > 0, "vlp_copy2d: bad args");
    VLP_ENTER(src, "vlp_copy2d");
    VLP_CHECK_ARG(lds >= (cols_src < cols_dst ? cols_src : cols_dst) && ldd >= cols_dst, "vlp_copy2d: leading dims");
    const int64_t total = (int64_t)rows * cols_dst;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(copy2d_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, lds, src_f32, (f16*)dst, ldd, rows,
                       cols_src < cols_dst ? cols_src : cols_dst, cols_dst, beta);
    VLP_CHECK_LAUNCH("vlp_copy2d");
    return VLP_OK;
}

// dst[c][r] = src[r][c]; 64x64 tiles through LDS; dst columns r in [rows, rows_pad) are zero-filled.
__global__ __launch_bounds__(256) void transpose_kernel(const f16* __restrict__ src, int64_t lds, f16* __restrict__ dst, int64_t ldd, int rows, int cols, int rows_pad) {
    __shared__ f16 tile[64][66];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < rows && c < cols) ? src[(int64_t)r * lds + c] : (f16)0.f;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int c = c0 + i, r = r0 + tx;
        if (c < cols && r < rows_pad) dst[(int64_t)c * ldd + r] = tile[tx][i];
    }
}
extern "C" int vlp_transpose(const void* src, int64_t lds, void* dst, int64_t ldd, int32_t rows, int32_t cols, int32_t rows_pad, void* stream) {
    VLP_CHECK_ARG(src && dst && rows > 0 && cols > 0 && rows_pad >= rows && ldd >= rows_pad && lds >= cols, "vlp_transpose: bad args");
    VLP_ENTER(src, "vlp_transpose");
    dim3 grid(cdiv(cols, 64), cdiv(rows_pad, 64));
    hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const f16*)src, lds, (f16*)dst, ldd, rows, cols, rows_pad);
    VLP_CHECK_LAUNCH("vlp_transpose");
    return VLP_OK;
}

// =================================================================================================
// gather / scatter of masked positions (modeling.py:1068-1069) and the VQA fusion (:1044, :1138)
// =================================================================================================
// first row and row count of sample b: dense [B, L] layout, or the packed layout given by row_off [B+1]
DEVFN int64_t sample_base(const int32_t* row_off, int64_t b, int L, int& n) {
    if (row_off) { const int lo = row_off[b]; n = row_off[b + 1] - lo; return lo; }
    n = L;
    return b * L;
}
__global__ void gather_rows_kernel(const f16* src, int64_t lds, const int64_t* pos, f16* out, int64_t ldo, int B, int P, int L, int H, const int32_t* row_off) {
    const int nch = H >> 3;
    const int64_t total = (int64_t)B * P * nch;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % nch);
        const int64_t r = i / nch;
        int n;
        const int64_t base = sample_base(row_off, r / P, L, n);
        int64_t ps = pos[r];
        ps = ps < 0 ? 0 : (ps >= n ? n - 1 : ps);
        st8(out + r * ldo + c * 8, ld8(src + (base + ps) * lds + c * 8));
    }
}
extern "C" int vlp_gather_rows(const void* src, int64_t lds, const int64_t* pos, void* out, int64_t ldo, int32_t B, int32_t P, int32_t L,
                               int32_t H, const int32_t* row_off, void* stream) {
    VLP_CHECK_ARG(src && pos && out && B > 0 && P > 0 && L > 0 && H % 8 == 0 && lds % 8 == 0 && ldo % 8 == 0, "vlp_gather_rows: bad args");
    VLP_ENTER(src, "vlp_gather_rows");
    const int64_t total = (int64_t)B * P * (H / 8);
    hipLaunchKernelGGL(gather_rows_kernel, dim3((int)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const f16*)src, lds, pos,
                       (f16*)out, ldo, B, P, L, H, row_off);
    VLP_CHECK_LAUNCH("vlp_gather_rows");
    return VLP_OK;
}
__global__ void scatter_add_rows_kernel(const f16* src, int64_t lds, const int64_t* pos, f16* dst, int64_t ldd, int B, int P, int L, int H, const int32_t* row_off) {
    const int nch = H >> 3;
    const int64_t total = (int64_t)B * P * nch;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % nch);
        const int64_t r = i / nch;
        int n;
        const int64_t base = sample_base(row_off, r / P, L, n);
        int64_t ps = pos[r];
        ps = ps < 0 ? 0 : (ps >= n ? n - 1 : ps);
        const f16x8 v = ld8(src + r * lds + c * 8);
        __half2* d = reinterpret_cast<__half2*>(dst + (base + ps) * ldd + c * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) unsafeAtomicAdd(d + e, __floats2half2_rn((float)v[2 * e], (float)v[2 * e + 1]));
    }
}
extern "C" int vlp_scatter_add_rows(const void* src, int64_t lds, const int64_t* pos, void* dst, int64_t ldd, int32_t B, int32_t P, int32_t L,
                                    int32_t H, const int32_t* row_off, void* stream) {
    VLP_CHECK_ARG(src && pos && dst && B > 0 && P > 0 && L > 0 && H % 8 == 0 && lds % 8 == 0 && ldd % 8 == 0, "vlp_scatter_add_rows: bad args");
    VLP_ENTER(src, "vlp_scatter_add_rows");
    const int64_t total = (int64_t)B * P * (H / 8);
    hipLaunchKernelGGL(scatter_add_rows_kernel, dim3((int)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const f16*)src, lds,
                       pos, (f16*)dst, ldd, B, P, L, H, row_off);
    VLP_CHECK_LAUNCH("vlp_scatter_add_rows");
    return VLP_OK;
}

__global__ void vqa_mul_fwd_kernel(const f16* h, f16* out, int B, int L, int Nv, int H, const int32_t* row_off) {
    const int64_t total = (int64_t)B * H;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / H;
        const int c = (int)(i % H);
        const int64_t r0 = row_off ? (int64_t)row_off[b] : b * L;
        const float a0 = (float)h[r0 * H + c], a1 = (float)h[(r0 + Nv + 1) * H + c];
        out[i] = (f16)(a0 * a1);
    }
}
extern "C" int vlp_vqa_mul_fwd(const void* h, void* out, int32_t B, int32_t L, int32_t Nv, int32_t H, const int32_t* row_off, void* stream) {
    VLP_CHECK_ARG(h && out && B > 0 && Nv + 1 < L, "vlp_vqa_mul_fwd: bad args");
    VLP_ENTER(h, "vlp_vqa_mul_fwd");
    hipLaunchKernelGGL(vqa_mul_fwd_kernel, dim3(cdiv((int64_t)B * H, 256)), dim3(256), 0, (hipStream_t)stream, (const f16*)h, (f16*)out, B, L, Nv, H, row_off);
    VLP_CHECK_LAUNCH("vlp_vqa_mul_fwd");
    return VLP_OK;
}
// dh[b,0] += dout * h[b,Nv+1];  dh[b,Nv+1] += dout * h[b,0]   (rows are private to this kernel -> plain RMW)
__global__ void vqa_mul_bwd_kernel(const f16* h, const f16* dout, f16* dh, int B, int L, int Nv, int H, const int32_t* row_off) {
    const int64_t total = (int64_t)B * H;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / H;
        const int c = (int)(i % H);
        const int64_t r0 = row_off ? (int64_t)row_off[b] : b * L;
        const int64_t i0 = r0 * H + c, i1 = (r0 + Nv + 1) * H + c;
        const float d = (float)dout[i], a0 = (float)h[i0], a1 = (float)h[i1];
        dh[i0] = (f16)((float)dh[i0] + d * a1);
        dh[i1] = (f16)((float)dh[i1] + d * a0);
    }
}
extern "C" int vlp_vqa_mul_bwd(const void* h, const void* dout, void* dh, int32_t B, int32_t L, int32_t Nv, int32_t H, const int32_t* row_off, void* stream) {
    VLP_CHECK_ARG(h && dout && dh && B > 0 && Nv + 1 < L, "vlp_vqa_mul_bwd: bad args");
    VLP_ENTER(h, "vlp_vqa_mul_bwd");
    hipLaunchKernelGGL(vqa_mul_bwd_kernel, dim3(cdiv((int64_t)B * H, 256)), dim3(256), 0, (hipStream_t)stream, (const f16*)h, (const f16*)dout,
                       (f16*)dh, B, L, Nv, H, row_off);
    VLP_CHECK_LAUNCH("vlp_vqa_mul_bwd");
    return VLP_OK;
}

// =================================================================================================
// padding-free (packed) row layout: sample b keeps its first n_b = row_off[b+1] - row_off[b] positions (include/vlp_hip.h)
// =================================================================================================
__global__ void rowmap_build_kernel(const int32_t* row_off, int L, int32_t* row_map) {
    const int b = blockIdx.x;
    const int lo = row_off[b], n = row_off[b + 1] - lo;
    for (int l = threadIdx.x; l < n; l += blockDim.x) row_map[lo + l] = b * L + l;
}
extern "C" int vlp_rowmap_build(const int32_t* row_off, int32_t B, int32_t L, int32_t* row_map, void* stream) {
    VLP_CHECK_ARG(row_off && row_map && B > 0 && L > 0, "vlp_rowmap_build: bad args");
    VLP_ENTER(row_off, "vlp_rowmap_build");
    hipLaunchKernelGGL(rowmap_build_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, row_off, L, row_map);
    VLP_CHECK_LAUNCH("vlp_rowmap_build");
    return VLP_OK;
}
template <bool UNPACK>
__global__ void rows_move_kernel(const f16* src, int64_t lds, const int32_t* row_map, int rows, f16* dst, int64_t ldd, int H) {
    const int nch = H >> 3;
    const int64_t total = (int64_t)rows * nch;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % nch);
        const int64_t p = i / nch, r = row_map[p];
        if (UNPACK) st8(dst + r * ldd + c * 8, ld8(src + p * lds + c * 8));
        else st8(dst + p * ldd + c * 8, ld8(src + r * lds + c * 8));
    }
}
static int rows_move(bool unpack, const void* src, int64_t lds, const int32_t* row_map, int32_t rows, void* dst, int64_t ldd, int32_t H, void* stream) {
    VLP_CHECK_ARG(src && dst && row_map && rows > 0 && H > 0 && H % 8 == 0 && lds % 8 == 0 && ldd % 8 == 0 && lds >= H && ldd >= H, "vlp_rows_pack/unpack: bad args");
    VLP_ENTER(src, "vlp_rows_pack/unpack");
    const int64_t total = (int64_t)rows * (H / 8);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    if (unpack) hipLaunchKernelGGL(rows_move_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const f16*)src, lds, row_map, rows, (f16*)dst, ldd, H);
    else hipLaunchKernelGGL(rows_move_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const f16*)src, lds, row_map, rows, (f16*)dst, ldd, H);
    VLP_CHECK_LAUNCH("vlp_rows_pack/unpack");
    return VLP_OK;
}
extern "C" int vlp_rows_unpack(const void* src, int64_t lds, const int32_t* row_map, int32_t rows, void* dst, int64_t ldd, int32_t H, void* stream) {
    return rows_move(true, src, lds, row_map, rows, dst, ldd, H, stream);
}
extern "C" int vlp_rows_pack(const void* src, int64_t lds, const int32_t* row_map, int32_t rows, void* dst, int64_t ldd, int32_t H, void* stream) {
    return rows_move(false, src, lds, row_map, rows, dst, ldd, H, stream);
}

// dz = dy * dropmask * (y > 0), idx = row*ncols + col (the forward GEMM epilogue's index)
__global__ void relu_dropout_bwd_kernel(const f16* dy, const f16* y, f16* dz, int64_t n8, int64_t ncols, DropCtx d) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        const f16x8 g = ld8(dy + i * 8), yv = ld8(y + i * 8);
        const int64_t row = (i * 8) / ncols;
        const uint32_t col0 = (uint32_t)((i * 8) % ncols);
        const uint32_t rk = d.thresh ? drop_rowkey(d, (uint64_t)row) : 0u;
        f16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float v = ((float)yv[e] > 0.f) ? (float)g[e] : 0.f;
            if (d.thresh) v *= drop_mult(d, rk, col0 + (uint32_t)e);
            o[e] = (f16)v;
        }
        st8(dz + i * 8, o);
    }
}
extern "C" int vlp_relu_dropout_bwd(const void* dy, const void* y, void* dz, int64_t n, int64_t ncols, float drop_p, uint64_t seed,
                                    uint32_t rng_stream, void* stream) {
    VLP_CHECK_ARG(dy && y && dz && n > 0 && ncols > 0 && ncols % 8 == 0 && n % ncols == 0, "vlp_relu_dropout_bwd: contiguous [rows, ncols], ncols % 8 == 0");
    VLP_ENTER(dy, "vlp_relu_dropout_bwd");
    int blocks = (int)((n / 8 + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(relu_dropout_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const f16*)dy, (const f16*)y, (f16*)dz, n / 8, ncols,
                       make_drop(drop_p, seed, rng_stream));
    VLP_CHECK_LAUNCH("vlp_relu_dropout_bwd");
    return VLP_OK;
}

__global__ void gelu_bwd_kernel(const f16* dy, const f16* z, f16* dz, int64_t n8) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        const f16x8 g = ld8(dy + i * 8), zv = ld8(z + i * 8);
        f16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (f16)((float)g[e] * gelu_grad_f((float)zv[e]));
        st8(dz + i * 8, o);
    }
}
extern "C" int vlp_gelu_bwd(const void* dy, const void* z, void* dz, int64_t n, void* stream) {
    VLP_CHECK_ARG(dy && z && dz && n > 0 && n % 8 == 0, "vlp_gelu_bwd: n must be a positive multiple of 8");
    VLP_ENTER(dy, "vlp_gelu_bwd");
    int blocks = (int)((n / 8 + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(gelu_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const f16*)dy, (const f16*)z, (f16*)dz, n / 8);
    VLP_CHECK_LAUNCH("vlp_gelu_bwd");
    return VLP_OK;
}

// ---------------------------------------------------------------------------------------------
// batched transposes (all weight shadows W^T of a step in ONE launch).  64x64 tiles, 16-byte global accesses on both
// sides; grid.x walks the concatenated tile list, `tile_start[i]` = first tile of matrix i.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void transpose_batched_kernel(const vlp_transpose_desc* __restrict__ descs, const int32_t* __restrict__ tile_start, int n) {
    __shared__ f16 tile[64][72];
    // locate the matrix of this tile (n is small: linear scan by one lane, broadcast through LDS-free readfirstlane)
    int mi = 0;
    const int t = blockIdx.x;
    while (mi + 1 < n && tile_start[mi + 1] <= t) ++mi;
    const vlp_transpose_desc d = descs[mi];
    const f16* src = (const f16*)d.src;
    f16* dst = (f16*)d.dst;
    const int tiles_c = (d.cols + 63) / 64;
    const int lt = t - tile_start[mi];
    const int r0 = (lt / tiles_c) * 64, c0 = (lt % tiles_c) * 64;
    const int tr = threadIdx.x >> 3, tc = (threadIdx.x & 7) * 8;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = r0 + tr + 32 * i, c = c0 + tc;
        f16x8 v = (f16x8){0, 0, 0, 0, 0, 0, 0, 0};
        if (r < d.rows) {
            if (c + 8 <= d.cols) v = ld8(src + (int64_t)r * d.lds + c);
            else
#pragma unroll
                for (int e = 0; e < 8; ++e) if (c + e < d.cols) v[e] = src[(int64_t)r * d.lds + c + e];
        }
        *reinterpret_cast<f16x8*>(&tile[tr + 32 * i][tc]) = v;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c = c0 + tr + 32 * i;         // output row
        const int r = r0 + tc;                  // output column start
        if (c >= d.cols || r >= d.rows_pad) continue;
        f16x8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = tile[tc + e][tr + 32 * i];
        if (r + 8 <= d.rows_pad) st8(dst + (int64_t)c * d.ldd + r, v);
        else
#pragma unroll
            for (int e = 0; e < 8; ++e) if (r + e < d.rows_pad) dst[(int64_t)c * d.ldd + r + e] = v[e];
    }
}
extern "C" int vlp_transpose_batched(const vlp_transpose_desc* descs_dev, const int32_t* tile_start_dev, int32_t n, int32_t total_tiles, void* stream) {
    VLP_CHECK_ARG(descs_dev && tile_start_dev && n > 0 && total_tiles > 0, "vlp_transpose_batched: bad args");
    VLP_ENTER(descs_dev, "vlp_transpose_batched");
    hipLaunchKernelGGL(transpose_batched_kernel, dim3(total_tiles), dim3(256), 0, (hipStream_t)stream, descs_dev, tile_start_dev, n);
    VLP_CHECK_LAUNCH("vlp_transpose_batched");
    return VLP_OK;
}

// ---------------------------------------------------------------------------------------------
// incremental decoding helpers
// ---------------------------------------------------------------------------------------------
// cache[b, start + i, 0:2H] = qkv_new[b*T + i, H:3H]   (k | v of the new tokens go to their positions in the K/V cache)
__global__ void kv_append_kernel(const f16* qkv, int64_t ld, f16* cache, int Lcap, int B, int T, int start, int H) {
    const int nch = (2 * H) >> 3;
    const int64_t total = (int64_t)B * T * nch;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % nch);
        const int64_t r = i / nch;
        const int t = (int)(r % T);
        const int64_t b = r / T;
        st8(cache + ((b * Lcap + start + t) * 2 * (int64_t)H) + c * 8, ld8(qkv + r * ld + H + c * 8));
    }
}
extern "C" int vlp_kv_append(const void* qkv_new, int64_t ld, void* cache, int32_t Lcap, int32_t B, int32_t T, int32_t start, int32_t H, void* stream) {
    VLP_CHECK_ARG(qkv_new && cache && B > 0 && T > 0 && start >= 0 && start + T <= Lcap && H % 8 == 0 && ld % 8 == 0, "vlp_kv_append: bad args");
    VLP_ENTER(qkv_new, "vlp_kv_append");
    const int64_t total = (int64_t)B * T * (2 * H / 8);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(kv_append_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const f16*)qkv_new, ld, (f16*)cache, Lcap, B, T, start, H);
    VLP_CHECK_LAUNCH("vlp_kv_append");
    return VLP_OK;
}

// ids[r] = argmax_v logits[r, v] (first maximum), vals[r] = that logit   (torch.max(prediction_scores, -1), modeling.py:1228)
__global__ __launch_bounds__(256) void argmax_rows_kernel(const f16* logits, int64_t ld, int V, int64_t* ids, int64_t ids_stride, float* vals,
                                                          int64_t vals_stride) {
    __shared__ float sv[256];
    __shared__ int si[256];
    const f16* x = logits + (int64_t)blockIdx.x * ld;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int v = threadIdx.x; v < V; v += 256) {
        const float f = (float)x[v];
        if (f > best) { best = f; bi = v; }
    }
    sv[threadIdx.x] = best;
    si[threadIdx.x] = bi;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            const float f = sv[threadIdx.x + o];
            const int j = si[threadIdx.x + o];
            if (f > sv[threadIdx.x] || (f == sv[threadIdx.x] && j < si[threadIdx.x])) { sv[threadIdx.x] = f; si[threadIdx.x] = j; }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { ids[blockIdx.x * ids_stride] = si[0]; vals[blockIdx.x * vals_stride] = sv[0]; }
}
extern "C" int vlp_argmax_rows(const void* logits, int64_t ld, int32_t rows, int32_t V, int64_t* ids, int64_t ids_stride, float* vals,
                               int64_t vals_stride, void* stream) {
    VLP_CHECK_ARG(logits && ids && vals && rows > 0 && V > 0 && ld >= V, "vlp_argmax_rows: bad args");
    VLP_ENTER(logits, "vlp_argmax_rows");
    hipLaunchKernelGGL(argmax_rows_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, (const f16*)logits, ld, V, ids, ids_stride, vals, vals_stride);
    VLP_CHECK_LAUNCH("vlp_argmax_rows");
    return VLP_OK;
}

// ---------------------------------------------------------------------------------------------
// beam search helpers (modeling.py:1255-1494)
// ---------------------------------------------------------------------------------------------
// per row: log_softmax over V, + (-10000) on forbidden words, eos column forced to -10000 when blocked, then the K best
// (value descending, index ascending on ties)  -- :1297-1303
__global__ __launch_bounds__(256) void logsoftmax_topk_kernel(const f16* logits, int64_t ld, int V, int K, const uint8_t* forbid, int eos_id,
                                                              int block_eos, float* out_scores, int64_t* out_ids) {
    __shared__ float sv[256];
    __shared__ int si[256];
    const int row = blockIdx.x, tid = threadIdx.x;
    const f16* x = logits + (int64_t)row * ld;
    const uint8_t* fb = forbid ? forbid + (int64_t)row * V : nullptr;
    float mx = -INFINITY;
    for (int v = tid; v < V; v += 256) mx = fmaxf(mx, (float)x[v]);
    sv[tid] = mx;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) sv[tid] = fmaxf(sv[tid], sv[tid + o]);
        __syncthreads();
    }
    mx = sv[0];
    __syncthreads();
    float sum = 0.f;
    for (int v = tid; v < V; v += 256) sum += __expf((float)x[v] - mx);
    sv[tid] = sum;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) sv[tid] += sv[tid + o];
        __syncthreads();
    }
    const float lse = mx + __logf(sv[0]);
    __syncthreads();
    float pv = INFINITY;
    int pi = -1;
    for (int k = 0; k < K; ++k) {
        float best = -INFINITY;
        int bi = 0x7fffffff;
        for (int v = tid; v < V; v += 256) {
            float val = (float)x[v] - lse;
            if (fb && fb[v]) val += -10000.0f;
            if (block_eos && v == eos_id) val = -10000.0f;
            const bool after = (val < pv) || (val == pv && v > pi);          // not yet taken
            if (after && (val > best || (val == best && v < bi))) { best = val; bi = v; }
        }
        sv[tid] = best;
        si[tid] = bi;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (tid < o) {
                const float f = sv[tid + o];
                const int j = si[tid + o];
                if (f > sv[tid] || (f == sv[tid] && j < si[tid])) { sv[tid] = f; si[tid] = j; }
            }
            __syncthreads();
        }
        pv = sv[0];
        pi = si[0];
        if (tid == 0) { out_scores[(int64_t)row * K + k] = pv; out_ids[(int64_t)row * K + k] = pi; }
        __syncthreads();
    }
}
// Two-pass variant for K <= KMAX (<= 16): pass 1 = max / sum-exp, pass 2 = every thread keeps the KMAX best of its strided elements in
// registers (sorted; an element is first tested against the thread's worst entry, so almost all cost one compare), then K rounds of a
// 256-candidate block arg-max over the threads' heads.  Same results as the K+2-pass kernel above (value descending, index ascending).
// Round 6: 1024 threads per row and 16-byte loads (the 256-thread form walked the row three times with 2-byte loads at a 512-byte stride: 113
// dependent iterations per pass, 73 us per launch at 192 rows -- 5 % of a beam-3 token step); block reductions by wave shuffles + 16 LDS words.
DEVFN void topk_better(float& bv, int& bi, float f, int j) {
    if (f > bv || (f == bv && j < bi)) { bv = f; bi = j; }
}
template <int KMAX>
__global__ __launch_bounds__(1024) void logsoftmax_topk_small_kernel(const f16* logits, int64_t ld, int V, int K, const uint8_t* forbid, int eos_id,
                                                                     int block_eos, float* out_scores, int64_t* out_ids) {
    __shared__ float sv[16];
    __shared__ int si[16];
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv_ = tid >> 6;
    const f16* x = logits + (int64_t)row * ld;
    const uint8_t* fb = forbid ? forbid + (int64_t)row * V : nullptr;
    const int nv = V >> 3;                               // whole 8-element vectors (rows are 16-byte aligned: ld % 8 == 0 is checked by the launcher)
    float mx = -INFINITY;
    for (int i = tid; i < nv; i += 1024) {
        const f16x8 q = ld8(x + i * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) mx = fmaxf(mx, (float)q[e]);
    }
    for (int v = nv * 8 + tid; v < V; v += 1024) mx = fmaxf(mx, (float)x[v]);
    mx = wave_max(mx);
    if (lane == 0) sv[wv_] = mx;
    __syncthreads();
    mx = sv[0];
#pragma unroll
    for (int k = 1; k < 16; ++k) mx = fmaxf(mx, sv[k]);
    __syncthreads();
    float sum = 0.f;
    for (int i = tid; i < nv; i += 1024) {
        const f16x8 q = ld8(x + i * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) sum += __expf((float)q[e] - mx);
    }
    for (int v = nv * 8 + tid; v < V; v += 1024) sum += __expf((float)x[v] - mx);
    sum = wave_sum(sum);
    if (lane == 0) sv[wv_] = sum;
    __syncthreads();
    sum = sv[0];
#pragma unroll
    for (int k = 1; k < 16; ++k) sum += sv[k];
    const float lse = mx + __logf(sum);
    __syncthreads();
    float lv[KMAX];
    int lidx[KMAX];
#pragma unroll
    for (int j = 0; j < KMAX; ++j) { lv[j] = -INFINITY; lidx[j] = 0x7fffffff; }
    auto offer = [&](float raw, int v) {
        float val = raw - lse;
        if (fb && fb[v]) val += -10000.0f;
        if (block_eos && v == eos_id) val = -10000.0f;
        if (val > lv[KMAX - 1] || (val == lv[KMAX - 1] && v < lidx[KMAX - 1])) {
            int vi = v;
#pragma unroll
            for (int j = 0; j < KMAX; ++j) {         // insertion into the sorted list (compare-and-swap down the chain)
                const bool better = val > lv[j] || (val == lv[j] && vi < lidx[j]);
                const float tv = better ? lv[j] : val;
                const int ti = better ? lidx[j] : vi;
                lv[j] = better ? val : lv[j];
                lidx[j] = better ? vi : lidx[j];
                val = tv;
                vi = ti;
            }
        }
    };
    for (int i = tid; i < nv; i += 1024) {
        const f16x8 q = ld8(x + i * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) offer((float)q[e], i * 8 + e);
    }
    for (int v = nv * 8 + tid; v < V; v += 1024) offer((float)x[v], v);
    for (int k = 0; k < K; ++k) {
        float bv = lv[0];
        int bi = lidx[0];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) topk_better(bv, bi, __shfl_xor(bv, o, 64), __shfl_xor(bi, o, 64));
        if (lane == 0) { sv[wv_] = bv; si[wv_] = bi; }
        __syncthreads();
        bv = sv[0];
        bi = si[0];
#pragma unroll
        for (int j = 1; j < 16; ++j) topk_better(bv, bi, sv[j], si[j]);
        if (tid == 0) { out_scores[(int64_t)row * K + k] = bv; out_ids[(int64_t)row * K + k] = bi; }
        if (lidx[0] == bi) {                         // the winner pops its head
#pragma unroll
            for (int j = 0; j + 1 < KMAX; ++j) { lv[j] = lv[j + 1]; lidx[j] = lidx[j + 1]; }
            lv[KMAX - 1] = -INFINITY;
            lidx[KMAX - 1] = 0x7fffffff;
        }
        __syncthreads();
    }
}
extern "C" int vlp_logsoftmax_topk(const void* logits, int64_t ld, int32_t rows, int32_t V, int32_t K, const uint8_t* forbid, int32_t eos_id,
                                   int32_t block_eos, float* out_scores, int64_t* out_ids, void* stream) {
    VLP_CHECK_ARG(logits && out_scores && out_ids && rows > 0 && V > 0 && K > 0 && K <= V && ld >= V, "vlp_logsoftmax_topk: bad args");
    VLP_ENTER(logits, "vlp_logsoftmax_topk");
    hipStream_t s = (hipStream_t)stream;
#define LAUNCH_TOPK(KM) hipLaunchKernelGGL(logsoftmax_topk_small_kernel<KM>, dim3(rows), dim3(1024), 0, s, (const f16*)logits, ld, V, K, forbid, eos_id, \
                                           block_eos, out_scores, out_ids)
    const bool vec_ok = ld % 8 == 0 && (uintptr_t)logits % 16 == 0;      // 16-byte row loads; anything else takes the scalar K+2-pass kernel
    if (vec_ok && K <= 4) LAUNCH_TOPK(4);
    else if (vec_ok && K <= 8) LAUNCH_TOPK(8);
    else if (vec_ok && K <= 16) LAUNCH_TOPK(16);
    else
        hipLaunchKernelGGL(logsoftmax_topk_kernel, dim3(rows), dim3(256), 0, s, (const f16*)logits, ld, V, K, forbid, eos_id, block_eos,
                           out_scores, out_ids);
#undef LAUNCH_TOPK
    VLP_CHECK_LAUNCH("vlp_logsoftmax_topk");
    return VLP_OK;
}

// one workgroup per sample: the K best of the K*K continuations (:1304-1320); first step: the K candidates of the single row
__global__ void beam_select_kernel(vlp_beam_select_args a) {
    const int b = blockIdx.x, K = a.K;
    if (threadIdx.x != 0) return;
    float pv = INFINITY;
    int pi = -1;
    for (int k = 0; k < K; ++k) {
        float best = -INFINITY;
        int bi = 0x7fffffff;
        const int ncand = a.first ? K : K * K;
        for (int i = 0; i < ncand; ++i) {
            float val;
            if (a.first) {
                val = a.kk_scores[(int64_t)b * K + i];
            } else {
                const int src = i / K;
                val = a.kk_scores[((int64_t)b * K + src) * K + (i % K)] + a.last_eos[(int64_t)b * K + src] * -10000.0f + a.last_total[(int64_t)b * K + src];
            }
            const bool after = (val < pv) || (val == pv && i > pi);
            if (after && (val > best || (val == best && i < bi))) { best = val; bi = i; }
        }
        pv = best;
        pi = bi;
        const int ptr = a.first ? 0 : bi / K;
        const int64_t id = a.first ? a.kk_ids[(int64_t)b * K + bi] : a.kk_ids[((int64_t)b * K + ptr) * K + (bi % K)];
        const int64_t o = (int64_t)b * K + k;
        a.out_scores[o] = best;
        a.out_ids[o] = id;
        a.out_ptrs[o] = ptr;
        a.out_eos[o] = id == a.eos_id ? 1.0f : 0.0f;
        a.src_rows[o] = a.first ? (int64_t)b : (int64_t)b * K + ptr;
        a.next_ids[o * a.next_ids_stride] = id;
    }
}
extern "C" int vlp_beam_select(const vlp_beam_select_args* a, void* stream) {
    VLP_CHECK_ARG(a && a->kk_scores && a->kk_ids && a->out_scores && a->out_ids && a->out_ptrs && a->out_eos && a->src_rows && a->next_ids,
                  "vlp_beam_select: null operand");
    VLP_ENTER(a->kk_scores, "vlp_beam_select");
    VLP_CHECK_ARG(a->B > 0 && a->K > 0 && a->K <= 64 && (a->first || (a->last_total && a->last_eos)), "vlp_beam_select: bad args");
    hipLaunchKernelGGL(beam_select_kernel, dim3(a->B), dim3(64), 0, (hipStream_t)stream, *a);
    VLP_CHECK_LAUNCH("vlp_beam_select");
    return VLP_OK;
}

// dst[r, pos, :] = src[idx[r], pos, :] for pos in [lo, hi): first_expand / select_beam_items (:1325-1349) on the K/V caches
__global__ void kv_gather_kernel(const f16* src, int64_t src_rows, f16* dst, int64_t dst_rows, const int64_t* idx, int R, int lo, int hi, int E) {
    const int nch = E >> 3;
    const int64_t total = (int64_t)R * (hi - lo) * nch;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % nch);
        const int64_t t = i / nch;
        const int pos = lo + (int)(t % (hi - lo));
        const int64_t r = t / (hi - lo);
        st8(dst + ((r * dst_rows + pos) * E) + c * 8, ld8(src + ((idx[r] * src_rows + pos) * E) + c * 8));
    }
}
extern "C" int vlp_kv_gather(const void* src, int64_t src_rows_per_batch, void* dst, int64_t dst_rows_per_batch, const int64_t* idx, int32_t R,
                             int32_t lo, int32_t hi, int32_t row_elems, void* stream) {
    VLP_CHECK_ARG(src && dst && idx && R > 0 && lo >= 0 && hi >= lo && hi <= src_rows_per_batch && hi <= dst_rows_per_batch && row_elems % 8 == 0,
                  "vlp_kv_gather: bad args");
    VLP_ENTER(src, "vlp_kv_gather");
    if (hi == lo) return VLP_OK;
    const int64_t total = (int64_t)R * (hi - lo) * (row_elems / 8);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(kv_gather_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const f16*)src, src_rows_per_batch, (f16*)dst,
                       dst_rows_per_batch, idx, R, lo, hi, row_elems);
    VLP_CHECK_LAUNCH("vlp_kv_gather");
    return VLP_OK;
}

// ids[r] ~ Categorical(softmax(logits[r, :V])) by the Gumbel-max trick on the counter hash (element = (row, v) of stream `stream`), and
// logp[r] = log_softmax(logits[r])[ids[r]]   (sample_mode == 'sample', modeling.py:1229-1235)
__global__ __launch_bounds__(256) void sample_rows_kernel(const f16* logits, int64_t ld, int V, DropCtx rng, int64_t* ids, int64_t ids_stride, float* logp,
                                                          int64_t logp_stride) {
    __shared__ float sv[256];
    __shared__ int si[256];
    const int row = blockIdx.x, tid = threadIdx.x;
    const f16* x = logits + (int64_t)row * ld;
    float mx = -INFINITY;
    for (int v = tid; v < V; v += 256) mx = fmaxf(mx, (float)x[v]);
    sv[tid] = mx;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) sv[tid] = fmaxf(sv[tid], sv[tid + o]);
        __syncthreads();
    }
    mx = sv[0];
    __syncthreads();
    const uint32_t rkey = drop_rowkey(rng, (uint64_t)row);
    float sum = 0.f, best = -INFINITY;
    int bi = 0x7fffffff;
    for (int v = tid; v < V; v += 256) {
        const float l = (float)x[v];
        sum += __expf(l - mx);
        const uint32_t h = mix32(rkey + (uint32_t)v * 0x9E3779B9u);
        const float u = ((float)(h >> 8) + 0.5f) * (1.0f / 16777216.0f);      // (0, 1)
        const float gum = -__logf(-__logf(u));
        const float val = l + gum;
        if (val > best) { best = val; bi = v; }
    }
    sv[tid] = sum;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) sv[tid] += sv[tid + o];
        __syncthreads();
    }
    const float lse = mx + __logf(sv[0]);
    __syncthreads();
    sv[tid] = best;
    si[tid] = bi;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) {
            const float f = sv[tid + o];
            const int j = si[tid + o];
            if (f > sv[tid] || (f == sv[tid] && j < si[tid])) { sv[tid] = f; si[tid] = j; }
        }
        __syncthreads();
    }
    if (tid == 0) {
        ids[row * ids_stride] = si[0];
        logp[row * logp_stride] = (float)x[si[0]] - lse;
    }
}
extern "C" int vlp_sample_rows(const void* logits, int64_t ld, int32_t rows, int32_t V, uint64_t seed, uint32_t rng_stream, int64_t* ids,
                               int64_t ids_stride, float* logp, int64_t logp_stride, void* stream) {
    VLP_CHECK_ARG(logits && ids && logp && rows > 0 && V > 0 && ld >= V, "vlp_sample_rows: bad args");
    VLP_ENTER(logits, "vlp_sample_rows");
    DropCtx rng = make_drop(0.5f, seed, rng_stream);
    hipLaunchKernelGGL(sample_rows_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, (const f16*)logits, ld, V, rng, ids, ids_stride, logp, logp_stride);
    VLP_CHECK_LAUNCH("vlp_sample_rows");
    return VLP_OK;
}

// ---------------------------------------------------------------------------------------------
// box / class encoding of the regions (seq2seq_loader.py:338-351), one wave per region row
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void vis_pe_prep_kernel(vlp_vis_pe_prep_args a) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (int64_t)a.B * a.Nv) return;
    const int64_t b = row / a.Nv;
    // "lazy normalisation": largest x / y corner of the image (:339-340)
    const float* bb = a.bbox + b * a.Nv * 6;
    float wmax = -INFINITY, hmax = -INFINITY;
    for (int j = lane; j < a.Nv; j += 64) {
        wmax = fmaxf(wmax, fmaxf(bb[j * 6 + 0], bb[j * 6 + 2]));
        hmax = fmaxf(hmax, fmaxf(bb[j * 6 + 1], bb[j * 6 + 3]));
    }
    const float w_est = wave_max(wmax) + 1e-5f, h_est = wave_max(hmax) + 1e-5f;
    const float* r = a.bbox + row * 6;
    float six[6];
    six[0] = r[0] / w_est; six[1] = r[1] / h_est; six[2] = r[2] / w_est; six[3] = r[3] / h_est;
    six[4] = fmaxf((six[3] - six[1]) * (six[2] - six[0]), 0.f);          // relative area, clamped (:345-346)
    six[5] = r[5];                                                        // confidence (:348)
    float mu = 0.f;
#pragma unroll
    for (int i = 0; i < 6; ++i) mu += six[i];
    mu *= (1.f / 6.f);
    float var = 0.f;
#pragma unroll
    for (int i = 0; i < 6; ++i) var += (six[i] - mu) * (six[i] - mu);
    const float rs6 = rsqrtf(var * (1.f / 6.f) + a.eps);
    f16* o = (f16*)a.out + row * a.ld_out;
    if (lane < 6) {
        float v = six[0];
#pragma unroll
        for (int i = 1; i < 6; ++i) v = lane == i ? six[i] : v;
        o[lane] = (f16)((v - mu) * rs6);
    }
    // class probabilities: layer norm over n_cls (:350-351)
    constexpr int MAXE = 32;                                              // up to 2048 classes per row
    float x[MAXE];
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < MAXE; ++e) {
        const int c = lane + 64 * e;
        float v = 0.f;
        if (c < a.n_cls) v = a.cls_is_f32 ? ((const float*)a.cls)[row * a.ld_cls + c] : (float)((const f16*)a.cls)[row * a.ld_cls + c];
        x[e] = v;
        s += v;
    }
    const float cm = wave_sum(s) / (float)a.n_cls;
    float q = 0.f;
#pragma unroll
    for (int e = 0; e < MAXE; ++e) {
        const int c = lane + 64 * e;
        if (c < a.n_cls) q += (x[e] - cm) * (x[e] - cm);
    }
    const float crs = rsqrtf(wave_sum(q) / (float)a.n_cls + a.eps);
#pragma unroll
    for (int e = 0; e < MAXE; ++e) {
        const int c = lane + 64 * e;
        if (c < a.n_cls) o[6 + c] = (f16)((x[e] - cm) * crs);
    }
    for (int c = 6 + a.n_cls + lane; c < a.pad_to; c += 64) o[c] = (f16)0.f;
}
extern "C" int vlp_vis_pe_prep(const vlp_vis_pe_prep_args* a, void* stream) {
    VLP_CHECK_ARG(a && a->bbox && a->cls && a->out && a->B > 0 && a->Nv > 0, "vlp_vis_pe_prep: null operand / bad shape");
    VLP_ENTER(a->out, "vlp_vis_pe_prep");
    VLP_CHECK_ARG(a->n_cls > 0 && a->n_cls <= 2048 && a->ld_cls >= a->n_cls && a->pad_to >= 6 + a->n_cls && a->ld_out >= a->pad_to,
                  "vlp_vis_pe_prep: n_cls <= 2048, ld_cls >= n_cls, ld_out >= pad_to >= 6 + n_cls");
    const int64_t rows = (int64_t)a->B * a->Nv;
    hipLaunchKernelGGL(vis_pe_prep_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, *a);
    VLP_CHECK_LAUNCH("vlp_vis_pe_prep");
    return VLP_OK;
}
