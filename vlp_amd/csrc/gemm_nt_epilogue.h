// Shared epilogue of the NT GEMM kernels (gemm_nt_ph.hip, gemm_nt_splitk.hip): one 8-wide output vector of row m.
#pragma once
#include "gemm_nt.h"

// one 8-wide output vector of row m: bias, pre-activation store, activation, gelu'/relu-mask multiply, dropout, residual, store
// SG (compile time) = the VLP_ACT_GELU_SAVE_GRAD form; it lives in its own kernel instantiations so that the generic epilogue keeps
// its register footprint (the erf/exp pair + derivative of 8 elements in flight costs ~25 VGPRs: the 128-VGPR variants spilled).
#ifdef VLP_NT_DEBUG
DEVFN void st8_pol(const GemmNtParams& p, f16* dst, f16x8 v) {      // store cache-policy experiment: dbg & 32 = nontemporal, & 64 = sc1
    if (p.dbg & 32) __builtin_nontemporal_store(v, reinterpret_cast<f16x8*>(dst));
    else if (p.dbg & 64) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst), "v"(v) : "memory");
    else st8(dst, v);
}
#define ST8_OUT(p, dst, v) st8_pol(p, dst, v)
#else
#define ST8_OUT(p, dst, v) st8_out<VLP_SS_NT>(dst, v)
#endif
// LIGHT (compile time): the caller guarantees act is NONE / RELU and mul_mode is not GELU_GRAD (nt_epilogue_is_light): the erf / exp /
// tanh expansions are compiled out -- the wave-pipelined kernels inline this function 16-32 times per lane.
static inline bool nt_epilogue_is_light(const GemmNtParams& p) {
    return (p.act == VLP_ACT_NONE || p.act == VLP_ACT_RELU) && p.mulmode != VLP_MUL_GELU_GRAD;
}
template <bool SG = false, bool LIGHT = false>
DEVFN void nt_epilogue8(const GemmNtParams& p, int m, int nc, float* vv, uint32_t rkey, bool bias_done = false) {
    if (nc >= p.N) return;
    if (p.bias && !bias_done) {
        if (nc + 8 <= p.N) {
            const f16x8 b = ld8(p.bias + nc);
#pragma unroll
            for (int j = 0; j < 8; ++j) vv[j] += (float)b[j];
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) if (nc + j < p.N) vv[j] += (float)p.bias[nc + j];
        }
    }
    if (SG) {
        // z = fp16-rounded pre-activation (what a stored z would hold); y = gelu(z); preact <- gelu'(z); nothing else is fused
        f16x8 d, o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float gl, gp;
            gelu_and_grad_f((float)(f16)vv[j], gl, gp);
            o[j] = (f16)gl;
            d[j] = (f16)gp;
        }
        if ((p.N & 7) && nc + 8 > p.N) {        // ragged last vector of a row (scalar test first: N % 8 == 0 skips it): zero the pad columns
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (nc + j >= p.N) { o[j] = (f16)0.f; d[j] = (f16)0.f; }
        }
        st8_out<VLP_SS_SAVED>(p.preact + (int64_t)m * p.ldp + nc, d);
        ST8_OUT(p, p.Y + (int64_t)m * p.ldy + nc, o);
        return;
    } else {
        if (p.preact) {
            f16x8 z;
#pragma unroll
            for (int j = 0; j < 8; ++j) z[j] = (nc + j < p.N) ? (f16)vv[j] : (f16)0.f;
            st8_out<VLP_SS_SAVED>(p.preact + (int64_t)m * p.ldp + nc, z);
#pragma unroll
            for (int j = 0; j < 8; ++j) vv[j] = (float)z[j];      // the activation sees the fp16-rounded pre-activation (as backward will)
        }
        if (!LIGHT && p.act == VLP_ACT_GELU) {
#pragma unroll
            for (int j = 0; j < 8; ++j) vv[j] = gelu_f(vv[j]);
        } else if (p.act == VLP_ACT_RELU) {
#pragma unroll
            for (int j = 0; j < 8; ++j) vv[j] = fmaxf(vv[j], 0.f);
        } else if (!LIGHT && p.act == VLP_ACT_TANH) {
#pragma unroll
            for (int j = 0; j < 8; ++j) vv[j] = tanhf(vv[j]);
        }
    }
    if (p.mulmode != VLP_MUL_NONE) {
        const f16x8 s = ld8(p.mulsrc + (int64_t)m * p.ldm + nc);
        if (!LIGHT && p.mulmode == VLP_MUL_GELU_GRAD) {
#pragma unroll
            for (int j = 0; j < 8; ++j) vv[j] *= gelu_grad_f((float)s[j]);
        } else if (p.mulmode == VLP_MUL_PLAIN) {
#pragma unroll
            for (int j = 0; j < 8; ++j) vv[j] *= (float)s[j];
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) vv[j] = ((float)s[j] > 0.f) ? vv[j] : 0.f;
        }
    }
    if (p.drop.thresh) {
        drop_mult8(p.drop, rkey, (uint32_t)nc, vv);
    }
    if (p.residual) {
        const f16x8 r = ld8(p.residual + (int64_t)m * p.ldr + nc);
#pragma unroll
        for (int j = 0; j < 8; ++j) vv[j] += (float)r[j];
    }
    f16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (nc + j < p.N) ? (f16)vv[j] : (f16)0.f;
    ST8_OUT(p, p.Y + (int64_t)m * p.ldy + nc, o);
}
