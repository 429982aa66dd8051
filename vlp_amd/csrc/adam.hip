// Optimizer kernels for gfx950 (HBM-bound, 16/32-byte vector accesses, flat parameter buffers).
//
//  * vlp_sumsq / vlp_adam_hyper / vlp_fused_adam: apex FP16_Optimizer + FusedAdam as configured by the
//    reference (run_img2txt_dist.py:411-420, step at :584; optimization_fp16.py:7-80): global grad norm and
//    overflow check over the flat fp16 gradient of a param group, clip folded into the unscale factor, one
//    pass that updates fp32 master weights + both moments and writes the fp16 model copy.  All scalar
//    decisions (clip coefficient, skip-on-overflow) stay on the device, so a step needs no host sync.
//  * vlp_bert_adam: BertAdam.step (optimization.py:112-182): per-TENSOR clip, eps outside the sqrt added to
//    sqrt(v), decoupled weight decay, no bias correction.
#include "common.h"
#include <stdlib.h>

#define SQ_BLOCKS 1024

__global__ __launch_bounds__(256) void sumsq_kernel(const f16* __restrict__ g, int64_t n, float* __restrict__ partial) {
    __shared__ float sh[4], shb[4];
    float s = 0.f, bad = 0.f;
    const int64_t n8 = n >> 3;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        f16x8 v = ld8(g + i * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float f = (float)v[e];
            s += f * f;
            if (!(fabsf(f) <= 65504.f)) bad = 1.f;   // inf or nan
        }
    }
    if (blockIdx.x == 0)
        for (int64_t i = n8 * 8 + threadIdx.x; i < n; i += blockDim.x) {
            const float f = (float)g[i];
            s += f * f;
            if (!(fabsf(f) <= 65504.f)) bad = 1.f;
        }
    s = wave_sum(s);
    bad = wave_max(bad);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    if (l == 0) { sh[w] = s; shb[w] = bad; }
    __syncthreads();
    if (threadIdx.x == 0) {
        partial[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
        partial[SQ_BLOCKS + blockIdx.x] = fmaxf(fmaxf(shb[0], shb[1]), fmaxf(shb[2], shb[3]));
    }
}
__global__ __launch_bounds__(256) void sumsq_finish_kernel(const float* __restrict__ partial, int nblocks, float* __restrict__ out2, int accumulate) {
    __shared__ float sh[4], shb[4];
    float s = 0.f, bad = 0.f;
    for (int i = threadIdx.x; i < nblocks; i += blockDim.x) { s += partial[i]; bad = fmaxf(bad, partial[SQ_BLOCKS + i]); }
    s = wave_sum(s);
    bad = wave_max(bad);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    if (l == 0) { sh[w] = s; shb[w] = bad; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float tot = sh[0] + sh[1] + sh[2] + sh[3];
        float b = fmaxf(fmaxf(shb[0], shb[1]), fmaxf(shb[2], shb[3]));
        if (!(tot <= 3.0e38f)) b = 1.f;   // the fp32 norm itself overflowed
        out2[0] = accumulate ? out2[0] + tot : tot;
        out2[1] = accumulate ? fmaxf(out2[1], b) : b;
    }
}
static int sumsq_launch(const void* g, int64_t n, float* out2, float* partial, int accumulate, void* stream);
extern "C" int vlp_sumsq(const void* g, int64_t n, float* out2, float* partial, void* stream) { return sumsq_launch(g, n, out2, partial, 0, stream); }
// out2 += (sum of squares, flag) of another range: the sharded optimizer step sums over the chunks a rank owns (launches of one stream run
// in order, so the sum has a fixed order: reproducible)
extern "C" int vlp_sumsq_acc(const void* g, int64_t n, float* out2, float* partial, void* stream) { return sumsq_launch(g, n, out2, partial, 1, stream); }
static int sumsq_launch(const void* g, int64_t n, float* out2, float* partial, int accumulate, void* stream) {
    VLP_CHECK_ARG(g && out2 && partial && n > 0 && (uintptr_t)g % 16 == 0, "vlp_sumsq: bad args (partial must hold 2048 floats)");
    VLP_ENTER(g, "vlp_sumsq");
    int blocks = (int)((n / 8 + 255) / 256);
    if (blocks > SQ_BLOCKS) blocks = SQ_BLOCKS;
    if (blocks < 1) blocks = 1;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(sumsq_kernel, dim3(blocks), dim3(256), 0, s, (const f16*)g, n, partial);
    VLP_CHECK_LAUNCH("vlp_sumsq");
    hipLaunchKernelGGL(sumsq_finish_kernel, dim3(1), dim3(256), 0, s, partial, blocks, out2, accumulate);
    VLP_CHECK_LAUNCH("vlp_sumsq(finish)");
    return VLP_OK;
}

__global__ void adam_hyper_kernel(const float* sumsq2, const float* any_overflow, const float* scale_state, float max_grad_norm, float step_size,
                                  float* hyper) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const float scale = scale_state[0];
    float overflow = sumsq2[1];
    if (any_overflow) overflow = fmaxf(overflow, any_overflow[0]);
    const float norm = sqrtf(sumsq2[0]);          // = true norm * scale
    float combined = scale;
    if (max_grad_norm > 0.f) {
        const float clip = (norm / scale + 1e-6f) / max_grad_norm;
        if (clip > 1.f) combined = clip * scale;
    }
    hyper[0] = combined;
    hyper[1] = step_size;
    hyper[2] = overflow;
}
extern "C" int vlp_adam_hyper(const float* sumsq2, const float* any_overflow, const float* scale_state, float max_grad_norm, float step_size,
                              float* hyper3, void* stream) {
    VLP_CHECK_ARG(sumsq2 && hyper3 && scale_state, "vlp_adam_hyper: bad args");
    VLP_ENTER(sumsq2, "vlp_adam_hyper");
    hipLaunchKernelGGL(adam_hyper_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, sumsq2, any_overflow, scale_state, max_grad_norm, step_size, hyper3);
    VLP_CHECK_LAUNCH("vlp_adam_hyper");
    return VLP_OK;
}

// apex FP16_Optimizer._update_scale on the device.  state = {cur_scale, cur_iter, last_overflow_iter, scale_factor,
// scale_window, dynamic, n_skipped, reserved}
__global__ void loss_scale_update_kernel(float* st, const float* overflow) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const bool skip = overflow[0] != 0.f;
    if (st[5] != 0.f) {
        if (skip) {
            st[0] = fmaxf(st[0] / st[3], 1.f);
            st[2] = st[1];
        } else if (fmodf(st[1] - st[2], st[4]) == 0.f) {
            st[0] = st[0] * st[3];
        }
    }
    if (skip) st[6] += 1.f;
    st[1] += 1.f;
}
extern "C" int vlp_loss_scale_update(float* scale_state, const float* overflow, void* stream) {
    VLP_CHECK_ARG(scale_state && overflow, "vlp_loss_scale_update: bad args");
    VLP_ENTER(scale_state, "vlp_loss_scale_update");
    hipLaunchKernelGGL(loss_scale_update_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, scale_state, overflow);
    VLP_CHECK_LAUNCH("vlp_loss_scale_update");
    return VLP_OK;
}

// Non-temporal loads / stores for the three fp32 state arrays and the gradient: each is touched exactly once per step (2.8 GB of the
// 3.25 GB), so their lines are kept out of L2 / the Infinity Cache, which are left to the fp16 parameters the next forward reads.
// Round 5 lab (tools/adam_lab.py, profiles/r05_adam_lab.txt, 115.9 M elements, cold caches): 4096 blocks, temporal accesses (rounds 2-4)
// 654 - 666 us = 4.9 TB/s; non-temporal 634 - 650; and FEWER resident waves stream better -- 512 blocks (two per CU, 2 waves per SIMD, 8
// loads of 1 KB in flight per wave) 560 - 588 us = 5.5 - 5.8 TB/s, 768 / 1024 blocks 610 - 628, 256 blocks 740; four runs per wave or
// one contiguous range per block change nothing.  A torch in-place multiply of one fp32 array reaches 5.7 TB/s on the same box.
DEVFN f32x4 adam_ld4(const float* p) { return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p)); }
DEVFN void adam_st4(float* p, f32x4 v) { __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p)); }
__global__ __launch_bounds__(256) void fused_adam_kernel(vlp_fused_adam_args a) {
    const float combined = a.hyper[0], step_size = a.hyper[1], skip = a.hyper[2];
    if (skip != 0.f) return;
    const float inv = 1.f / combined;
    const f16* g = (const f16*)a.g16;
    f16* p16 = (f16*)a.p16;
    // a wave covers 512 consecutive elements per iteration as two runs of 256: lane l owns elements 4l..4l+3 of each run, so every
    // load / store instruction of the wave touches one contiguous 1 KiB (fp32 arrays) or 512 B (fp16 arrays) span -- with 8 consecutive
    // elements per lane the two 16-byte halves of a lane's fp32 data sat 32 bytes apart in every instruction.  Same arithmetic per element.
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t base = wave * 512; base < a.n; base += nwaves * 512) {
        f16x4 gv[2];
        f32x4 p0[2], m0[2], v0[2];
        bool ok[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {            // all eight loads of the wave's 512 elements are in flight before the first use
            const int64_t i = base + 256 * h + 4 * lane;
            ok[h] = i < a.n;                      // n is a multiple of 8, i of 4: a run is whole or absent per lane
            if (ok[h]) {
                gv[h] = __builtin_nontemporal_load(reinterpret_cast<const f16x4*>(g + i));
                p0[h] = adam_ld4(a.p32 + i);
                m0[h] = adam_ld4(a.m + i);
                v0[h] = adam_ld4(a.v + i);
            }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (!ok[h]) continue;
            const int64_t i = base + 256 * h + 4 * lane;
            float pp[4], mm[4], vv[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) { pp[e] = p0[h][e]; mm[e] = m0[h][e]; vv[e] = v0[h][e]; }
            f16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float sg = (float)gv[h][e] * inv;
                mm[e] = a.b1 * mm[e] + (1.f - a.b1) * sg;
                vv[e] = a.b2 * vv[e] + (1.f - a.b2) * sg * sg;
                const float denom = a.eps_inside_sqrt ? sqrtf(vv[e] + a.eps) : sqrtf(vv[e]) + a.eps;
                pp[e] = pp[e] - step_size * (mm[e] / denom + a.decay * pp[e]);
                // the fp16 model copy is the rounding of the STORED fp32 master (apex: p_copy = (half) p).  Without the opaque copy the
                // compiler folds the last fma and the conversion into v_fma_mixlo_f16 (one rounding of the exact fma), which differs from
                // half(master) in double-rounding cases -- a resumed run, which can only rebuild the copy from the master, would diverge.
                asm volatile("" : "+v"(pp[e]));
                o[e] = (f16)pp[e];
            }
            adam_st4(a.p32 + i, (f32x4){pp[0], pp[1], pp[2], pp[3]});
            adam_st4(a.m + i, (f32x4){mm[0], mm[1], mm[2], mm[3]});
            adam_st4(a.v + i, (f32x4){vv[0], vv[1], vv[2], vv[3]});
            st4(p16 + i, o);          // (temporal: the next forward reads the fp16 parameters)
        }
    }
}
extern "C" int vlp_fused_adam(const vlp_fused_adam_args* a, void* stream) {
    VLP_CHECK_ARG(a && a->p32 && a->m && a->v && a->g16 && a->p16 && a->hyper, "vlp_fused_adam: null operand");
    VLP_ENTER(a->p32, "vlp_fused_adam");
    VLP_CHECK_ARG(a->n > 0 && a->n % 8 == 0, "vlp_fused_adam: n must be a positive multiple of 8 (pad the flat buffer)");
    VLP_CHECK_ARG(((uintptr_t)a->p32 | (uintptr_t)a->m | (uintptr_t)a->v | (uintptr_t)a->g16 | (uintptr_t)a->p16) % 16 == 0, "vlp_fused_adam: alignment");
    static const int ncu = [] { int dev = 0, n = 256; (void)hipGetDevice(&dev); if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 256; return n; }();
    int blocks = (int)((a->n / 8 + 255) / 256);
    if (blocks > 2 * ncu) blocks = 2 * ncu;          // two blocks per CU: see the lab note above the kernel
    hipLaunchKernelGGL(fused_adam_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, *a);
    VLP_CHECK_LAUNCH("vlp_fused_adam");
    return VLP_OK;
}

// ---------------------------------------------------------------------------------------------
// BertAdam: pass 1a = partial sums of squares per (4096-element chunk, tensor) pair into norms[ntensors + chunk + tensor] (the slots of
// successive pairs are strictly increasing: a tensor's chunks are consecutive and the next tensor starts in the chunk the previous one
// ended in or later), pass 1b = one wave per tensor adds its partials in chunk order into norms[tensor] -- deterministic, no atomics;
// pass 2 = update with the per-tensor clip coefficient.
// ---------------------------------------------------------------------------------------------
#define BA_CHUNK 4096   // elements per block in both passes

DEVFN float ba_load(const void* g, int g_is_f32, int64_t i) { return g_is_f32 ? ((const float*)g)[i] : (float)((const f16*)g)[i]; }

// binary search: tensor t with seg_off[t] <= i < seg_off[t+1]
DEVFN int ba_find(const int64_t* seg_off, int nt, int64_t i) {
    int lo = 0, hi = nt - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (seg_off[mid] <= i) lo = mid; else hi = mid - 1;
    }
    return lo;
}

__global__ __launch_bounds__(256) void bert_adam_norm_kernel(vlp_bert_adam_args a) {
    __shared__ float sh[4];
    const int64_t start = (int64_t)blockIdx.x * BA_CHUNK;
    const int64_t end = min(a.n, start + BA_CHUNK);
    // a chunk may straddle tensor boundaries: accumulate per element into its tensor (rarely > 2 tensors)
    int t = ba_find(a.seg_off, a.ntensors, start);
    int64_t pos = start;
    while (pos < end) {
        if (t >= a.ntensors) break;                  // n beyond seg_off[ntensors] (padded flat buffer): the tail belongs to no tensor
        const int64_t tend = min(end, a.seg_off[t + 1]);
        if (a.active && !a.active[t]) { pos = tend; ++t; continue; }
        float s = 0.f;
        for (int64_t i = pos + threadIdx.x; i < tend; i += blockDim.x) {
            const float f = ba_load(a.g, a.g_is_f32, i) / a.grad_scale;
            s += f * f;
        }
        s = wave_sum(s);
        const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
        __syncthreads();
        if (l == 0) sh[w] = s;
        __syncthreads();
        if (threadIdx.x == 0) a.norms[a.ntensors + (int64_t)blockIdx.x + t] = sh[0] + sh[1] + sh[2] + sh[3];
        pos = tend;
        ++t;
    }
}
__global__ __launch_bounds__(64) void bert_adam_norm_finish_kernel(vlp_bert_adam_args a) {
    const int t = blockIdx.x;
    if (a.active && !a.active[t]) { if (threadIdx.x == 0) a.norms[t] = 0.f; return; }
    const int64_t lo = a.seg_off[t], hi = a.seg_off[t + 1];
    if (hi <= lo) { if (threadIdx.x == 0) a.norms[t] = 0.f; return; }
    const int64_t c0 = lo / BA_CHUNK, c1 = (hi - 1) / BA_CHUNK;
    const float* part = a.norms + a.ntensors + t;
    float s = 0.f;
    for (int64_t c = c0 + threadIdx.x; c <= c1; c += 64) s += part[c];       // lane l: chunks c0 + l, c0 + l + 64, ... (fixed order)
    s = wave_sum(s);
    if (threadIdx.x == 0) a.norms[t] = s;
}
extern "C" int64_t vlp_bert_adam_norms_floats(int64_t n, int32_t ntensors) {
    return (int64_t)ntensors + (n + BA_CHUNK - 1) / BA_CHUNK + (int64_t)ntensors;
}
__global__ __launch_bounds__(256) void bert_adam_update_kernel(vlp_bert_adam_args a) {
    const int64_t start = (int64_t)blockIdx.x * BA_CHUNK;
    const int64_t end = min(a.n, start + BA_CHUNK);
    int t = ba_find(a.seg_off, a.ntensors, start);
    int64_t pos = start;
    f16* p16 = (f16*)a.p16;
    while (pos < end) {
        if (t >= a.ntensors) break;
        const int64_t tend = min(end, a.seg_off[t + 1]);
        if (a.active && !a.active[t]) { pos = tend; ++t; continue; }
        float coef = 1.f;
        if (a.max_grad_norm > 0.f) {
            // torch clip_grad_norm_: clip_coef = max_norm / (total_norm + 1e-6), applied when < 1
            const float c = a.max_grad_norm / (sqrtf(a.norms[t]) + 1e-6f);
            if (c < 1.f) coef = c;
        }
        coef /= a.grad_scale;
        for (int64_t i = pos + threadIdx.x; i < tend; i += blockDim.x) {
            const float g = ba_load(a.g, a.g_is_f32, i) * coef;
            const float m = a.b1 * a.m[i] + (1.f - a.b1) * g;
            const float v = a.b2 * a.v[i] + (1.f - a.b2) * g * g;
            float p = a.p32[i];
            float upd = m / (sqrtf(v) + a.eps);
            if (a.decay > 0.f) upd += a.decay * p;
            p -= a.lr * upd;
            a.m[i] = m; a.v[i] = v; a.p32[i] = p;
            if (p16) p16[i] = (f16)p;
        }
        pos = tend;
        ++t;
    }
}
extern "C" int vlp_bert_adam(const vlp_bert_adam_args* a, void* stream) {
    VLP_CHECK_ARG(a && a->p32 && a->m && a->v && a->g && a->seg_off && a->norms, "vlp_bert_adam: null operand");
    VLP_ENTER(a->p32, "vlp_bert_adam");
    VLP_CHECK_ARG(a->n > 0 && a->ntensors > 0 && a->grad_scale > 0.f, "vlp_bert_adam: bad sizes");
    // ABI 3: the scratch size travels with the call (an ABI-1 consumer that sized `norms` as [ntensors] would otherwise get silent
    // out-of-bounds device writes from the two-stage norm reduction)
    VLP_CHECK_ARG(a->norms_floats >= vlp_bert_adam_norms_floats(a->n, a->ntensors),
                  "vlp_bert_adam: norms_floats=%lld < vlp_bert_adam_norms_floats(n, ntensors)=%lld", (long long)a->norms_floats,
                  (long long)vlp_bert_adam_norms_floats(a->n, a->ntensors));
    hipStream_t s = (hipStream_t)stream;
    const int blocks = (int)((a->n + BA_CHUNK - 1) / BA_CHUNK);
    if (a->max_grad_norm > 0.f) {
        hipLaunchKernelGGL(bert_adam_norm_kernel, dim3(blocks), dim3(256), 0, s, *a);
        hipLaunchKernelGGL(bert_adam_norm_finish_kernel, dim3(a->ntensors), dim3(64), 0, s, *a);
        VLP_CHECK_LAUNCH("vlp_bert_adam(norm)");
    }
    hipLaunchKernelGGL(bert_adam_update_kernel, dim3(blocks), dim3(256), 0, s, *a);
    VLP_CHECK_LAUNCH("vlp_bert_adam(update)");
    return VLP_OK;
}
