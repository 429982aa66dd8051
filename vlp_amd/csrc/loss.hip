// Loss kernels of the VLP hot path for gfx950.
//   masked-LM: CrossEntropyLoss(reduction='none') in fp32 over the tied-decoder logits, masked_weights,
//              per-sample sum, drop-worst top-k, normalisation (modeling.py:1083-1111);
//   VQA:       BCEWithLogitsLoss(mean) * num_answers (modeling.py:1030, 1140).
#include "common.h"

#define CE_THREADS 256

DEVFN float block_reduce_sum(float v, float* sh) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    __syncthreads();
    if (l == 0) sh[w] = v;
    __syncthreads();
    float r = 0.f;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) r += sh[i];
    return r;
}
DEVFN float block_reduce_max(float v, float* sh) {
    v = wave_max(v);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    __syncthreads();
    if (l == 0) sh[w] = v;
    __syncthreads();
    float r = -INFINITY;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) r = fmaxf(r, sh[i]);
    return r;
}

// one block per row: lse[row] and row_loss[row] = lse - logit[label]
__global__ __launch_bounds__(CE_THREADS) void ce_row_kernel(const f16* __restrict__ logits, int64_t ld, const int64_t* __restrict__ labels,
                                                            float* __restrict__ lse, float* __restrict__ row_loss, int V) {
    __shared__ float sh[8];
    const int row = blockIdx.x;
    const f16* x = logits + (int64_t)row * ld;
    const int v8 = V >> 3;
    float mx = -INFINITY;
    for (int c = threadIdx.x; c < v8; c += CE_THREADS) {
        f16x8 t = ld8(x + c * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) mx = fmaxf(mx, (float)t[e]);
    }
    for (int i = v8 * 8 + threadIdx.x; i < V; i += CE_THREADS) mx = fmaxf(mx, (float)x[i]);
    mx = block_reduce_max(mx, sh);
    float s = 0.f;
    for (int c = threadIdx.x; c < v8; c += CE_THREADS) {
        f16x8 t = ld8(x + c * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) s += __expf((float)t[e] - mx);
    }
    for (int i = v8 * 8 + threadIdx.x; i < V; i += CE_THREADS) s += __expf((float)x[i] - mx);
    s = block_reduce_sum(s, sh);
    if (threadIdx.x == 0) {
        const float l = mx + __logf(s);
        int64_t lab = labels[row];
        lab = lab < 0 ? 0 : (lab >= V ? V - 1 : lab);
        lse[row] = l;
        row_loss[row] = l - (float)x[lab];
    }
}

// single block: masking, per-sample sums, drop-worst selection by rank, normalisation (modeling.py:1083-1093)
__global__ __launch_bounds__(1024) void mlm_finish_kernel(const float* __restrict__ row_loss, const int64_t* __restrict__ weights,
                                                          float* __restrict__ loss, float* __restrict__ coef, int B, int P, int keep_n) {
    extern __shared__ float shm[];
    float* ssum = shm;          // [B] per-sample masked loss
    float* wsum = shm + B;      // [B] per-sample weight sum
    float* keep = shm + 2 * B;  // [B]
    __shared__ float red[16];
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        float s = 0.f, w = 0.f;
        for (int p = 0; p < P; ++p) {
            const float wt = (float)weights[b * P + p];
            s += row_loss[b * P + p] * wt;
            w += wt;
        }
        ssum[b] = s;
        wsum[b] = w;
    }
    __syncthreads();
    float den_part = 0.f, loss_part = 0.f;
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        // rank among samples (smallest first, ties broken by index) == torch.topk(largest=False) membership
        int rank = 0;
        const float sb = ssum[b];
        for (int j = 0; j < B; ++j) rank += (ssum[j] < sb || (ssum[j] == sb && j < b)) ? 1 : 0;
        const float k = rank < keep_n ? 1.f : 0.f;
        keep[b] = k;
        den_part += k * wsum[b];
        loss_part += k * sb;
    }
    const float den = block_reduce_sum(den_part, red) + 1e-5f;
    const float tot = block_reduce_sum(loss_part, red);
    if (threadIdx.x == 0) loss[0] = tot / den;
    __syncthreads();
    for (int i = threadIdx.x; i < B * P; i += blockDim.x) coef[i] = keep[i / P] * (float)weights[i] / den;
}

extern "C" int vlp_mlm_loss_fwd(const vlp_mlm_loss_fwd_args* a, void* stream) {
    VLP_CHECK_ARG(a && a->logits && a->labels && a->weights && a->loss && a->lse && a->coef && a->row_loss, "vlp_mlm_loss_fwd: null operand");
    VLP_ENTER(a->logits, "vlp_mlm_loss_fwd");
    VLP_CHECK_ARG(a->B > 0 && a->P > 0 && a->V > 0 && a->B <= 4096, "vlp_mlm_loss_fwd: bad shape (B <= 4096)");
    VLP_CHECK_ARG(a->ld_logits % 8 == 0 && a->ld_logits >= a->V && (uintptr_t)a->logits % 16 == 0, "vlp_mlm_loss_fwd: logits layout");
    VLP_CHECK_ARG(a->drop_worst_ratio >= 0.f && a->drop_worst_ratio < 1.f, "vlp_mlm_loss_fwd: drop_worst_ratio");
    hipStream_t s = (hipStream_t)stream;
    const int rows = a->B * a->P;
    hipLaunchKernelGGL(ce_row_kernel, dim3(rows), dim3(CE_THREADS), 0, s, (const f16*)a->logits, a->ld_logits, a->labels, a->lse, a->row_loss, a->V);
    VLP_CHECK_LAUNCH("vlp_mlm_loss_fwd(ce)");
    // int(loss.size(0) * (1 - ratio)) computed like python: double arithmetic, truncation
    const int keep_n = (int)((double)a->B * (1.0 - (double)a->drop_worst_ratio));
    hipLaunchKernelGGL(mlm_finish_kernel, dim3(1), dim3(1024), 3 * a->B * sizeof(float), s, a->row_loss, a->weights, a->loss, a->coef, a->B, a->P, keep_n);
    VLP_CHECK_LAUNCH("vlp_mlm_loss_fwd(finish)");
    return VLP_OK;
}

__global__ __launch_bounds__(CE_THREADS) void ce_bwd_kernel(const f16* __restrict__ logits, int64_t ld, const int64_t* __restrict__ labels,
                                                            const float* __restrict__ lse, const float* __restrict__ coef,
                                                            const float* __restrict__ gscale, f16* __restrict__ dl, int64_t ldd, int V) {
    const int row = blockIdx.y;
    const f16* x = logits + (int64_t)row * ld;
    f16* d = dl + (int64_t)row * ldd;
    const float c = coef[row] * gscale[0];
    const float l = lse[row];
    int64_t lab = labels[row];
    lab = lab < 0 ? 0 : (lab >= V ? V - 1 : lab);
    const int n8 = (int)(ldd >> 3);
    for (int ch = blockIdx.x * CE_THREADS + threadIdx.x; ch < n8; ch += gridDim.x * CE_THREADS) {
        const int v0 = ch * 8;
        f16x8 o;
        if (v0 + 8 <= V) {
            f16x8 t = ld8(x + v0);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (f16)(c * (__expf((float)t[e] - l) - ((v0 + e) == lab ? 1.f : 0.f)));
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int v = v0 + e;
                o[e] = v < V ? (f16)(c * (__expf((float)x[v] - l) - (v == lab ? 1.f : 0.f))) : (f16)0.f;
            }
        }
        st8(d + v0, o);
    }
}
extern "C" int vlp_mlm_loss_bwd(const vlp_mlm_loss_bwd_args* a, void* stream) {
    VLP_CHECK_ARG(a && a->logits && a->labels && a->lse && a->coef && a->grad_scale && a->dlogits, "vlp_mlm_loss_bwd: null operand");
    VLP_ENTER(a->logits, "vlp_mlm_loss_bwd");
    VLP_CHECK_ARG(a->rows > 0 && a->V > 0 && a->ld_logits % 8 == 0 && a->ld_dlogits % 8 == 0 && a->ld_dlogits >= a->V, "vlp_mlm_loss_bwd: layout");
    VLP_CHECK_ARG(((uintptr_t)a->logits | (uintptr_t)a->dlogits) % 16 == 0, "vlp_mlm_loss_bwd: alignment");
    int bx = cdiv(a->ld_dlogits / 8, CE_THREADS);
    if (bx > 16) bx = 16;
    hipLaunchKernelGGL(ce_bwd_kernel, dim3(bx, a->rows), dim3(CE_THREADS), 0, (hipStream_t)stream, (const f16*)a->logits, a->ld_logits, a->labels,
                       a->lse, a->coef, a->grad_scale, (f16*)a->dlogits, a->ld_dlogits, a->V);
    VLP_CHECK_LAUNCH("vlp_mlm_loss_bwd");
    return VLP_OK;
}

// ---------------------------------------------------------------------------------------------
// BCE with logits.  loss = sum_{b,n} [max(x,0) - x*y + log(1 + exp(-|x|))] / (B*N) * N
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bce_fwd_kernel(const f16* x, int64_t ld, const float* y, int64_t ldl, int B, int N, float* part) {
    __shared__ float sh[8];
    float s = 0.f;
    const int64_t total = (int64_t)B * N;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / N;
        const int n = (int)(i % N);
        const float xv = (float)x[b * ld + n], yv = y[b * ldl + n];
        s += fmaxf(xv, 0.f) - xv * yv + log1pf(__expf(-fabsf(xv)));
    }
    s = block_reduce_sum(s, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}
__global__ void bce_finish_kernel(const float* part, int n, float inv, float* loss) {
    __shared__ float sh[8];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s += part[i];
    s = block_reduce_sum(s, sh);
    if (threadIdx.x == 0) loss[0] = s * inv;
}
extern "C" int vlp_bce_loss_fwd(const void* logits, int64_t ld, const void* labels, int64_t ldl, int32_t B, int32_t N, float* loss, void* stream) {
    VLP_CHECK_ARG(logits && labels && loss && B > 0 && N > 0 && ld >= N && ldl >= N, "vlp_bce_loss_fwd: bad args");
    VLP_ENTER(logits, "vlp_bce_loss_fwd");
    // loss[1..257) is used as scratch: the caller passes a buffer of >= 257 floats
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(bce_fwd_kernel, dim3(256), dim3(256), 0, s, (const f16*)logits, ld, (const float*)labels, ldl, B, N, loss + 1);
    VLP_CHECK_LAUNCH("vlp_bce_loss_fwd");
    hipLaunchKernelGGL(bce_finish_kernel, dim3(1), dim3(256), 0, s, loss + 1, 256, 1.f / (float)B, loss);
    VLP_CHECK_LAUNCH("vlp_bce_loss_fwd(finish)");
    return VLP_OK;
}
__global__ void bce_bwd_kernel(const f16* x, int64_t ld, const float* y, int64_t ldl, int B, int N, const float* gscale, f16* d, int64_t ldd) {
    const float c = gscale[0] / (float)B;
    const int64_t total = (int64_t)B * ldd;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / ldd;
        const int n = (int)(i % ldd);
        float v = 0.f;
        if (n < N) {
            const float xv = (float)x[b * ld + n];
            v = c * (1.f / (1.f + __expf(-xv)) - y[b * ldl + n]);
        }
        d[i] = (f16)v;
    }
}
extern "C" int vlp_bce_loss_bwd(const void* logits, int64_t ld, const void* labels, int64_t ldl, int32_t B, int32_t N, const float* grad_scale,
                                void* dlogits, int64_t ldd, void* stream) {
    VLP_CHECK_ARG(logits && labels && grad_scale && dlogits && B > 0 && N > 0 && ld >= N && ldl >= N && ldd >= N, "vlp_bce_loss_bwd: bad args");
    VLP_ENTER(logits, "vlp_bce_loss_bwd");
    const int64_t total = (int64_t)B * ldd;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(bce_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const f16*)logits, ld, (const float*)labels, ldl, B, N,
                       grad_scale, (f16*)dlogits, ldd);
    VLP_CHECK_LAUNCH("vlp_bce_loss_bwd");
    return VLP_OK;
}
