// mask_image_regions / vis_pretext_loss of BertForPreTrainingLossMask for gfx950 (modeling.py:1049-1056, 1113-1131).
//
// The branch is small (25 masked regions per sample at --vis_mask_prob 0.25: a [25, 25] similarity matrix over H = 768 per sample), so
// these are latency-shaped kernels: one workgroup per sample, everything read through L2, fp32 arithmetic, fixed summation orders
// (no atomics -> bitwise reproducible like the rest of the backward pass).
#include "common.h"

#define PT_MAXP 64      // masked regions per sample (one softmax column per lane)

__global__ __launch_bounds__(256) void region_mask_kernel(const int64_t* pos, int B, int Pm, int Nv, uint8_t* out) {
    const int b = blockIdx.x;
    for (int i = threadIdx.x; i < Nv; i += blockDim.x) out[(int64_t)b * Nv + i] = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < Pm; i += blockDim.x) {
        const int64_t r = pos[(int64_t)b * Pm + i] - 1;
        if (r >= 0 && r < Nv) out[(int64_t)b * Nv + r] = 1;
    }
}
extern "C" int vlp_region_mask_build(const int64_t* vis_masked_pos, int32_t B, int32_t Pm, int32_t Nv, uint8_t* out, void* stream) {
    VLP_CHECK_ARG(vis_masked_pos && out, "vlp_region_mask_build: null operand");
    VLP_ENTER(out, "vlp_region_mask_build");
    VLP_CHECK_ARG(B > 0 && Pm >= 0 && Nv > 0, "vlp_region_mask_build: bad shape");
    hipLaunchKernelGGL(region_mask_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, vis_masked_pos, B, Pm, Nv, out);
    VLP_CHECK_LAUNCH("vlp_region_mask_build");
    return VLP_OK;
}

DEVFN int64_t pt_row(const int64_t* pos, int b, int i, int Pm, int Nv) {       // region row of the i-th masked position of sample b
    int64_t r = pos[(int64_t)b * Pm + i] - 1;
    r = r < 0 ? 0 : (r >= Nv ? Nv - 1 : r);
    return (int64_t)b * Nv + r;
}

// One workgroup (4 waves) per sample; wave w owns the similarity rows i = w, w + 4, ...: lane l accumulates chunk columns l, l + 64, ...
// of the H-long dot product, a wave reduction finishes sim[i][j]; lane j then holds sim[i][j] for the row softmax.
__global__ __launch_bounds__(256) void pretext_fwd_kernel(vlp_pretext_fwd_args a) {
    __shared__ float row_loss[PT_MAXP];
    const int b = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int nch = a.H >> 3;
    const f16* vis = (const f16*)a.vis_h;
    const f16* vpe = (const f16*)a.vispe_h;
    const f16* pooled = (const f16*)a.pooled + (int64_t)b * a.H;
    for (int i = wv; i < a.Pm; i += 4) {
        const f16* arow = vpe + pt_row(a.vis_masked_pos, b, i, a.Pm, a.Nv) * a.H;
        float mine = 0.f;                       // sim[i][lane]
        for (int j = 0; j < a.Pm; ++j) {
            const f16* vrow = vis + pt_row(a.vis_masked_pos, b, j, a.Pm, a.Nv) * a.H;
            float s = 0.f;
            for (int c = lane; c < nch; c += 64) {
                const f16x8 e = ld8(arow + c * 8), q = ld8(pooled + c * 8), v = ld8(vrow + c * 8);
#pragma unroll
                for (int t = 0; t < 8; ++t) s = fmaf((float)(f16)((float)e[t] + (float)q[t]), (float)v[t], s);    // A rounded to fp16 (:1124)
            }
            s = (float)(f16)wave_sum(s);        // the half matmul's output rounding (:1126)
            if (lane == j) mine = s;
        }
        const float x = lane < a.Pm ? mine : -INFINITY;
        const float mx = wave_max(x);
        const float ex = lane < a.Pm ? __expf(x - mx) : 0.f;
        const float sum = wave_sum(ex);
        if (lane < a.Pm) a.probs[((int64_t)b * a.Pm + i) * a.Pm + lane] = ex / sum;
        if (lane == i) row_loss[i] = -(x - mx - __logf(sum));          // -log_softmax(sim)[i][i]  (:1127-1130)
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int i = 0; i < a.Pm; ++i) s += row_loss[i];
        a.sample_loss[b] = s / (float)a.Pm;
    }
}
__global__ __launch_bounds__(64) void pretext_finish_kernel(const float* sample_loss, int B, float* loss) {
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += sample_loss[b];              // fixed order (:1131)
        loss[0] = s / (float)B;
    }
}
extern "C" int vlp_pretext_fwd(const vlp_pretext_fwd_args* a, void* stream) {
    VLP_CHECK_ARG(a && a->vis_h && a->vispe_h && a->pooled && a->vis_masked_pos && a->probs && a->sample_loss && a->loss, "vlp_pretext_fwd: null operand");
    VLP_ENTER(a->vis_h, "vlp_pretext_fwd");
    VLP_CHECK_ARG(a->B > 0 && a->Nv > 0 && a->Pm > 0 && a->Pm <= PT_MAXP && a->H > 0 && a->H % 8 == 0, "vlp_pretext_fwd: bad shape (1 <= Pm <= %d, H %% 8 == 0)", PT_MAXP);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(pretext_fwd_kernel, dim3(a->B), dim3(256), 0, s, *a);
    hipLaunchKernelGGL(pretext_finish_kernel, dim3(1), dim3(64), 0, s, (const float*)a->sample_loss, a->B, a->loss);
    VLP_CHECK_LAUNCH("vlp_pretext_fwd");
    return VLP_OK;
}

// backward: one workgroup per sample.  dsim lives in LDS; an (output row r, 8-column chunk c) item computes dA_r[c] and dV_r[c] as Pm-long
// fp32 sums in index order; the pooled gradient is sum_j (sum_i dsim[i][j]) V_j.
__global__ __launch_bounds__(256) void pretext_bwd_kernel(vlp_pretext_bwd_args a, DropCtx dvis, DropCtx dvpe) {
    __shared__ float dsim[PT_MAXP * PT_MAXP];
    __shared__ float colsum[PT_MAXP];
    __shared__ int64_t rows[PT_MAXP];
    const int b = blockIdx.x, Pm = a.Pm, nch = a.H >> 3;
    const f16* vis = (const f16*)a.vis_h;
    const f16* vpe = (const f16*)a.vispe_h;
    const f16* pooled = (const f16*)a.pooled + (int64_t)b * a.H;
    const float g = a.gscale[0] / ((float)a.B * (float)Pm);
    for (int i = threadIdx.x; i < Pm; i += blockDim.x) rows[i] = pt_row(a.vis_masked_pos, b, i, Pm, a.Nv);
    for (int t = threadIdx.x; t < Pm * Pm; t += blockDim.x) {
        const int i = t / Pm, j = t - i * Pm;
        dsim[t] = g * (a.probs[(int64_t)b * Pm * Pm + t] - (i == j ? 1.f : 0.f));
    }
    __syncthreads();
    for (int j = threadIdx.x; j < Pm; j += blockDim.x) {
        float s = 0.f;
        for (int i = 0; i < Pm; ++i) s += dsim[i * Pm + j];
        colsum[j] = s;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < Pm * nch; t += blockDim.x) {
        const int r = t / nch, c = t - r * nch;
        float dA[8], dV[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { dA[e] = 0.f; dV[e] = 0.f; }
        const f16x8 q = ld8(pooled + c * 8);
        for (int k = 0; k < Pm; ++k) {
            const f16x8 v = ld8(vis + rows[k] * a.H + c * 8);           // V_k
            const f16x8 e8 = ld8(vpe + rows[k] * a.H + c * 8);          // A_k = vispe_k + pooled (fp16, as the forward)
            const float w_rk = dsim[r * Pm + k], w_kr = dsim[k * Pm + r];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                dA[e] = fmaf(w_rk, (float)v[e], dA[e]);
                dV[e] = fmaf(w_kr, (float)(f16)((float)e8[e] + (float)q[e]), dV[e]);
            }
        }
        // through ReLU + dropout of the projections, exactly as embed_bwd_kernel does for the unmasked rows
        const int64_t vr = rows[r];
        const f16x8 yv = ld8(vis + vr * a.H + c * 8), yp = ld8(vpe + vr * a.H + c * 8);
        const uint32_t kv = dvis.thresh ? drop_rowkey(dvis, (uint64_t)vr) : 0u;
        const uint32_t kp = dvpe.thresh ? drop_rowkey(dvpe, (uint64_t)vr) : 0u;
        f16x8 ov, op;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const uint32_t col = (uint32_t)(c * 8 + e);
            float gv = ((float)yv[e] > 0.f) ? dV[e] : 0.f;
            float gp = ((float)yp[e] > 0.f) ? dA[e] : 0.f;
            if (dvis.thresh) gv *= drop_mult(dvis, kv, col);
            if (dvpe.thresh) gp *= drop_mult(dvpe, kp, col);
            ov[e] = (f16)gv;
            op[e] = (f16)gp;
        }
        st8((f16*)a.d_vis_h + vr * a.H + c * 8, ov);
        st8((f16*)a.d_vispe_h + vr * a.H + c * 8, op);
    }
    for (int c = threadIdx.x; c < nch; c += blockDim.x) {
        float dp[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) dp[e] = 0.f;
        for (int k = 0; k < Pm; ++k) {
            const f16x8 v = ld8(vis + rows[k] * a.H + c * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) dp[e] = fmaf(colsum[k], (float)v[e], dp[e]);
        }
        const f16x8 q = ld8(pooled + c * 8);
        f16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (f16)(dp[e] * (1.f - (float)q[e] * (float)q[e]));      // tanh' = 1 - tanh^2
        st8((f16*)a.d_pooled_pre + (int64_t)b * a.H + c * 8, o);
    }
}
extern "C" int vlp_pretext_bwd(const vlp_pretext_bwd_args* a, void* stream) {
    VLP_CHECK_ARG(a && a->vis_h && a->vispe_h && a->pooled && a->vis_masked_pos && a->probs && a->gscale && a->d_vis_h && a->d_vispe_h && a->d_pooled_pre,
                  "vlp_pretext_bwd: null operand");
    VLP_ENTER(a->vis_h, "vlp_pretext_bwd");
    VLP_CHECK_ARG(a->B > 0 && a->Nv > 0 && a->Pm > 0 && a->Pm <= PT_MAXP && a->H > 0 && a->H % 8 == 0, "vlp_pretext_bwd: bad shape (1 <= Pm <= %d, H %% 8 == 0)", PT_MAXP);
    VLP_CHECK_ARG(a->drop_p >= 0.f && a->drop_p < 1.f, "vlp_pretext_bwd: bad dropout p");
    hipLaunchKernelGGL(pretext_bwd_kernel, dim3(a->B), dim3(256), 0, (hipStream_t)stream, *a, make_drop(a->drop_p, a->seed, a->vis_stream),
                       make_drop(a->drop_p, a->seed, a->vispe_stream));
    VLP_CHECK_LAUNCH("vlp_pretext_bwd");
    return VLP_OK;
}
