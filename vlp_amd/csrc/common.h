// Shared device/host helpers for libvlp_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "vlp_hip.h"

typedef _Float16 f16;
typedef f16 f16x2 __attribute__((ext_vector_type(2)));
typedef f16 f16x4 __attribute__((ext_vector_type(4)));
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

#define WAVE 64
#define DEVFN __device__ __forceinline__

// ---------------------------------------------------------------------------------------------
// host-side error plumbing (thread-local last error string; never throws, never aborts)
// ---------------------------------------------------------------------------------------------
int vlp_set_error(int code, const char* fmt, ...);
#define VLP_CHECK_ARG(cond, ...)                                         \
    do {                                                                 \
        if (!(cond)) return vlp_set_error(VLP_ERR_BAD_ARG, __VA_ARGS__); \
    } while (0)
#define VLP_CHECK_LAUNCH(name)                                                                        \
    do {                                                                                              \
        hipError_t e_ = hipGetLastError();                                                            \
        if (e_ != hipSuccess) return vlp_set_error(VLP_ERR_HIP, "%s: %s", name, hipGetErrorString(e_)); \
    } while (0)

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// ---------------------------------------------------------------------------------------------
// device selection on entry (SURVEY.md 8b threading contract): every entry point makes the device that owns its first operand current
// for the calling thread for the duration of the call and puts the caller's device back on return (the owner of a pointer is asked from
// the driver once per allocation and cached per thread), so the library can be driven from any host thread and for several devices of
// one process.  A host pointer is refused here with a clear message instead of faulting in a kernel.
// ---------------------------------------------------------------------------------------------
struct VlpDeviceGuard {      // api.cpp: selects the operand's device for the scope of one entry point, restores the caller's on return
    int prev, rc;
    VlpDeviceGuard(const void* device_ptr, const char* who);
    ~VlpDeviceGuard();
    VlpDeviceGuard(const VlpDeviceGuard&) = delete;
    VlpDeviceGuard& operator=(const VlpDeviceGuard&) = delete;
};
int vlp_current_device(void);                                       // hipGetDevice() of the calling thread
#define VLP_ENTER(ptr, who)                                 \
    VlpDeviceGuard vlp_guard_((ptr), (who));                \
    if (vlp_guard_.rc != VLP_OK) return vlp_guard_.rc
// launcher state that HIP keeps per device (hipFuncSetAttribute: dynamic LDS limit): set once per (kernel instantiation, device), race-free --
// two threads may both run `stmt` (idempotent), nobody launches before it has run on his device
#define VLP_ONCE_PER_DEVICE(stmt)                                                      \
    do {                                                                               \
        static std::atomic<uint64_t> once_mask_{0};                                    \
        const uint64_t once_bit_ = 1ull << (vlp_current_device() & 63);                \
        if (!(once_mask_.load(std::memory_order_acquire) & once_bit_)) {               \
            stmt;                                                                      \
            once_mask_.fetch_or(once_bit_, std::memory_order_release);                 \
        }                                                                              \
    } while (0)

// ---------------------------------------------------------------------------------------------
// counter-based dropout RNG.  keep(seed, stream, idx) is a pure function, so backward recomputes the
// mask instead of storing it.  (The reference uses torch's Philox stream, which cannot be matched
// bit-for-bit anyway -- parity tests run with p = 0; see DESIGN.md.)
// ---------------------------------------------------------------------------------------------
// Mixer: two rounds of xorshift + 24-bit multiply-add.  Every instruction is full rate on CDNA4 (v_mad_u32_u24), whereas the
// 32-bit multiplies of the usual murmur / lowbias32 finalizers run at quarter rate -- with one hash per element those multiplies were
// the single largest VALU cost of the attention kernels.  Avalanche measured over 2e5 random inputs: every output bit flips with
// probability 0.5 +- 0.008 for every input bit (lowbias32: +- 0.004); keep-rate, adjacent-element and 2-D autocorrelation of the
// resulting masks are at the sampling-noise level (tools/mixer_eval.py).
DEVFN uint32_t mix32(uint32_t x) {
    x ^= x >> 15; x = __umul24(x, 0xd3833fu) + (x >> 7);
    x ^= x >> 13; x = __umul24(x, 0x7a6b35u) + (x >> 9);
    x ^= x >> 16;
    return x;
}
#define VLP_PHI 0x9E3779B9u
struct DropCtx {
    uint32_t k0, k1, thresh;   // 16-bit threshold: an element is dropped when its 16-bit hash half < thresh (0 = dropout off)
    float scale;               // 1/(1-p)
};
static inline DropCtx make_drop(float p, uint64_t seed, uint32_t stream) {
    DropCtx d;
    uint64_t s = seed * 0x9E3779B97F4A7C15ull + (uint64_t)stream * 0xD1B54A32D192ED03ull + 0x632BE59BD9B4E019ull;
    d.k0 = (uint32_t)s;
    d.k1 = (uint32_t)(s >> 32) | 1u;
    double t = (double)p * 65536.0 + 0.5;
    d.thresh = p <= 0.f ? 0u : (t >= 65535.0 ? 65535u : (t < 1.0 ? 1u : (uint32_t)t));
    d.scale = p <= 0.f ? 1.f : 1.f / (1.f - p);
    return d;
}
// The mask of element (row, col) of a logical 2-D tensor: a 32-bit row key (computed once per row) mixed with the column PAIR
// col >> 1; the even column takes the low 16 bits of the hash, the odd one the high 16 bits (drop probability resolution 2^-16).
// Forward and backward of an op must agree on what (row, col) mean -- each kernel documents it.
DEVFN uint32_t drop_rowkey(const DropCtx& d, uint64_t row) {
    return mix32((uint32_t)row ^ d.k0) + mix32((uint32_t)(row >> 32) + d.k1);
}
// pair key of columns (2j, 2j+1): kernels that walk consecutive columns compute it once and add VLP_PHI per pair
DEVFN uint32_t drop_pairkey(uint32_t rowkey, uint32_t col) { return rowkey + (col >> 1) * VLP_PHI; }
// multiplier (0 or 1/(1-p)) of the even (odd = 0) / odd (odd = 1) column of a hashed pair
DEVFN float drop_mult_h(const DropCtx& d, uint32_t h, uint32_t odd) {
    const uint32_t v = odd ? (h >> 16) : (h & 0xffffu);
    return v < d.thresh ? 0.f : d.scale;
}
DEVFN float drop_mult(const DropCtx& d, uint32_t rowkey, uint32_t col) {
    return drop_mult_h(d, mix32(drop_pairkey(rowkey, col)), col & 1u);
}
// 8 consecutive columns starting at the (even) column col0: 4 hashes
DEVFN void drop_mult8(const DropCtx& d, uint32_t rowkey, uint32_t col0, float* vv) {
    const uint32_t base = drop_pairkey(rowkey, col0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t h = mix32(base + (uint32_t)j * VLP_PHI);
        vv[2 * j] *= drop_mult_h(d, h, 0u);
        vv[2 * j + 1] *= drop_mult_h(d, h, 1u);
    }
}
// 4 consecutive columns starting at col0 (a multiple of 4): 2 hashes
DEVFN void drop_mult4(const DropCtx& d, uint32_t rowkey, uint32_t col0, float* vv) {
    const uint32_t base = drop_pairkey(rowkey, col0);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const uint32_t h = mix32(base + (uint32_t)j * VLP_PHI);
        vv[2 * j] *= drop_mult_h(d, h, 0u);
        vv[2 * j + 1] *= drop_mult_h(d, h, 1u);
    }
}

// ---------------------------------------------------------------------------------------------
// math
// ---------------------------------------------------------------------------------------------
// erf via Abramowitz-Stegun 7.1.26 (|abs error| <= 1.5e-7, far below the fp16 rounding of every consumer):
// one v_rcp + one v_exp instead of the ~40-instruction libm erff in the GEMM epilogues.
DEVFN float fast_erf(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));     // v_rcp_f32 (1 ulp); __frcp_rn expands to the 10-instruction IEEE division
    float poly = fmaf(1.061405429f, t, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    const float r = 1.0f - poly * t * __builtin_amdgcn_exp2f(-ax * ax * 1.4426950408889634f);      // one v_exp_f32 (argument <= 0: no range fix-ups)
    return copysignf(r, x);
}
DEVFN float gelu_f(float x) { return x * 0.5f * (1.0f + fast_erf(x * 0.70710678118654752440f)); }   // modeling.py:62-67
DEVFN float gelu_grad_f(float x) {
    // d/dx [x * Phi(x)] = Phi(x) + x * phi(x)
    float cdf = 0.5f * (1.0f + fast_erf(x * 0.70710678118654752440f));
    float pdf = 0.39894228040143267794f * __builtin_amdgcn_exp2f(-0.5f * 1.4426950408889634f * x * x);
    return cdf + x * pdf;
}

// gelu(x) and gelu'(x) together: the derivative's pdf term is the exponential fast_erf already evaluates (exp(-x^2/2)), so the pair
// costs two FMAs more than gelu alone.  The FFN-up forward stores gelu'(z) in place of z; the FFN-down dgrad then multiplies by a
// stored number instead of re-evaluating erf + exp per element (the backward epilogue was VALU-bound: +22 us per launch).
DEVFN void gelu_and_grad_f(float x, float& gl, float& gp) {
    const float ax = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    float poly = fmaf(1.061405429f, t, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(-ax * ax * 1.4426950408889634f);      // exp(-x^2 / 2): one v_exp_f32, no range fix-ups (argument <= 0)
    const float cdf = 0.5f * (1.0f + copysignf(1.0f - poly * t * e, x));
    gl = x * cdf;
    gp = fmaf(x * 0.39894228040143267794f, e, cdf);
}

// Cross-lane exchanges that stay in the VALU (no LDS round trip: a ds_bpermute-based __shfl_xor costs ~100 cycles of dependent latency each):
//   lanes ^1, ^2 inside a quad and the mirror inside 8 lanes by DPP, ^8 by a rotate inside the row of 16, ^16 / ^32 by the gfx950
//   v_permlane16_swap / v_permlane32_swap (with both operands the same value the two results are "mine" and "the partner's").
#ifdef VLP_SHFL_BPERMUTE      // A/B builds (tools/build_variant_lib.sh <out.so> -DVLP_SHFL_BPERMUTE): every exchange as the ds_bpermute __shfl_xor of rounds 1-5
DEVFN float lane_xor1(float v) { return __shfl_xor(v, 1, 64); }
DEVFN float lane_xor2(float v) { return __shfl_xor(v, 2, 64); }
DEVFN float lane_mirror8(float v) { return __shfl_xor(v, 4, 64); }   // (equal to the mirror wherever the quads are uniform: every use below)
DEVFN float lane_xor8(float v) { return __shfl_xor(v, 8, 64); }
#else
#define VLP_DPP(v, ctrl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (ctrl), 0xF, 0xF, true))
DEVFN float lane_xor1(float v) { return VLP_DPP(v, 0xB1); }          // quad_perm [1,0,3,2]
DEVFN float lane_xor2(float v) { return VLP_DPP(v, 0x4E); }          // quad_perm [2,3,0,1]
DEVFN float lane_mirror8(float v) { return VLP_DPP(v, 0x141); }      // row_half_mirror: lane i <-> 7 - i inside every 8 lanes
DEVFN float lane_xor8(float v) { return VLP_DPP(v, 0x128); }         // row_ror:8 inside every 16 lanes
#endif
// (inline asm, not __builtin_amdgcn_permlane{16,32}_swap: with the builtin hipcc / ROCm 7.2 folded op(r[0], r[1]) to r[0] -- the v_max / v_add
// behind the swap was missing from the ISA of attn_decode_small_kernel, also with the second operand made opaque.  The two v_nop are the
// wait states the "VALU write -> v_permlane read" hazard needs, cdna_hip_programming.md T21.)
#if defined(VLP_SHFL_BPERMUTE) || defined(VLP_PAIR_BPERMUTE)
DEVFN void lane_pair16_u(unsigned v, unsigned& a, unsigned& b) { a = v; b = (unsigned)__shfl_xor((int)v, 16, 64); }
DEVFN void lane_pair32_u(unsigned v, unsigned& a, unsigned& b) { a = v; b = (unsigned)__shfl_xor((int)v, 32, 64); }
#else
#ifdef VLP_PAIR_PAD
#define VLP_PL_PRE "s_nop 7\n\t"
#define VLP_PL_POST "\n\ts_nop 7"
#else
#define VLP_PL_PRE "v_nop\n\tv_nop\n\t"
#define VLP_PL_POST ""
#endif
DEVFN void lane_pair16_u(unsigned v, unsigned& a, unsigned& b) {     // op(a, b) = the ^16 combination of v, whichever of the two is "mine"
    a = v; b = v;
    asm volatile(VLP_PL_PRE "v_permlane16_swap_b32 %0, %1" VLP_PL_POST : "+v"(a), "+v"(b));
}
DEVFN void lane_pair32_u(unsigned v, unsigned& a, unsigned& b) {
    a = v; b = v;
    asm volatile(VLP_PL_PRE "v_permlane32_swap_b32 %0, %1" VLP_PL_POST : "+v"(a), "+v"(b));
}
#endif
DEVFN void lane_pair16(float v, float& a, float& b) {
    unsigned x, y;
    lane_pair16_u(__builtin_bit_cast(unsigned, v), x, y);
    a = __builtin_bit_cast(float, x); b = __builtin_bit_cast(float, y);
}
DEVFN void lane_pair32(float v, float& a, float& b) {
    unsigned x, y;
    lane_pair32_u(__builtin_bit_cast(unsigned, v), x, y);
    a = __builtin_bit_cast(float, x); b = __builtin_bit_cast(float, y);
}
DEVFN float add_xor16(float v) { float a, b; lane_pair16(v, a, b); return a + b; }       // v + (v of lane ^ 16): bitwise the __shfl_xor form (a + b commutes)
DEVFN float add_xor32(float v) { float a, b; lane_pair32(v, a, b); return a + b; }
DEVFN float max_xor16(float v) { float a, b; lane_pair16(v, a, b); return fmaxf(a, b); }
DEVFN float max_xor32(float v) { float a, b; lane_pair32(v, a, b); return fmaxf(a, b); }
DEVFN unsigned or_xor16(unsigned v) { unsigned a, b; lane_pair16_u(v, a, b); return a | b; }
DEVFN unsigned or_xor32(unsigned v) { unsigned a, b; lane_pair32_u(v, a, b); return a | b; }
DEVFN float sum8(float v) { v += lane_xor1(v); v += lane_xor2(v); return v + lane_mirror8(v); }     // all 8 lanes of a group end with the group's sum
DEVFN float sum_over_groups8(float v) {                              // lanes with equal (lane & 7): sum over the 8 groups of a wave
    v += lane_xor8(v);
    float a, b;
    lane_pair16(v, a, b); v = a + b;
    lane_pair32(v, a, b); return a + b;
}
DEVFN float max_over_groups8(float v) {
    v = fmaxf(v, lane_xor8(v));
    float a, b;
    lane_pair16(v, a, b); v = fmaxf(a, b);
    lane_pair32(v, a, b); return fmaxf(a, b);
}

// Whole-wave reductions: the ds_bpermute butterflies of rounds 1-5 (hipcc lowers __shfl_xor to ds_bpermute_b32).  Round 6 measured the VALU
// forms above in their place (every reduction of LayerNorm / attention / the loss kernels; profiles/r06_instep_ab_lane_exchanges.txt):
// LayerNorm forward 10.0 -> 9.7 us, backward 15.1 -> 14.7 us in the lab, 9.105 / 9.140 vs 9.163 / 9.123 ms per step in the same-box A/B -- inside
// the noise (the chains hide behind the other resident waves) -- AND the asm v_permlane16/32_swap form made attn_fwd_kernel<16, 8> with dropout
// draw wrong keep decisions for one hash word of key tile 8 (tests/test_00_kernels_gpu.py::test_attention_forward_dropout_decisions[223-100];
// deterministic, not cured by wait states, not root-caused).  The training kernels therefore keep the proven form; the VALU exchanges are
// used by attn_decode_small_kernel only (its own tests cover them), where they are worth 2.5 us of a 12.8 us launch.
DEVFN float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
DEVFN float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// LDS-DMA of 16 bytes per lane: global -> LDS at (wave-uniform base + lane * 16).  Issued through inline asm, not
// __builtin_amdgcn_global_load_lds: the compiler's waitcnt pass books the builtin as a FLAT access that may touch LDS, and from then
// until the VM queue is fully drained it turns EVERY lgkmcnt wait into lgkmcnt(0) -- a ring that keeps DMA stages in flight never
// drains, so each MFMA group waited for all outstanding fragment reads instead of its own (16 ds_read_b128 in flight, first MFMA
// after the last of them).  The asm form is invisible to that pass: fragment waits are counted again (lgkmcnt(13), (12), (9) ...).
// The caller owns the vmcnt accounting of these loads (asm s_waitcnt), as the rings already did.
DEVFN uint32_t lds_addr_of(const void* p) {     // LDS byte address of a __shared__ pointer (take it once: the generic -> LDS cast carries a null check)
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}
DEVFN void glds16(const f16* gsrc, uint32_t lds_wave_base) {      // lds_wave_base: wave-uniform LDS byte address
    const uint32_t m0v = __builtin_amdgcn_readfirstlane(lds_wave_base);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(m0v), "v"(gsrc) : "memory", "m0");
}
#ifdef VLP_NT_DEBUG
DEVFN void glds16_nt(const f16* gsrc, uint32_t lds_wave_base) {
    const uint32_t m0v = __builtin_amdgcn_readfirstlane(lds_wave_base);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt" ::"s"(m0v), "v"(gsrc) : "memory", "m0");
}
DEVFN void glds16_sc1(const f16* gsrc, uint32_t lds_wave_base) {
    const uint32_t m0v = __builtin_amdgcn_readfirstlane(lds_wave_base);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off sc1" ::"s"(m0v), "v"(gsrc) : "memory", "m0");
}
#endif
DEVFN f16x8 ld8(const f16* p) { return *reinterpret_cast<const f16x8*>(p); }
DEVFN void st8(f16* p, f16x8 v) { *reinterpret_cast<f16x8*>(p) = v; }
DEVFN f16x4 ld4(const f16* p) { return *reinterpret_cast<const f16x4*>(p); }
DEVFN void st4(f16* p, f16x4 v) { *reinterpret_cast<f16x4*>(p) = v; }
// Streamed (non-temporal) output stores, per kernel family -- OFF in the product (VLP_STREAM_STORES = 0).  Measured in round 3
// (profiles/r03_nt_wp_decomposition.txt): on a lone NT GEMM with cold operands they gain 3 - 12 % (the written lines no longer evict
// operand lines from the 4 MB L2 of an XCD), but inside the step the NEXT kernel reads the tensor from the Infinity Cache when it was
// stored normally and from HBM when it was streamed: 10.14 -> 10.99 ms / step with the NT GEMM outputs streamed, 9.64 -> 9.73 ms with only
// the FFN side outputs the backward pass reads much later.  Kept as an investigation switch (tools/build_variant_lib.sh -DVLP_STREAM_STORES=mask).
#define VLP_SS_NT 1        // NT GEMM outputs
#define VLP_SS_TN 2        // weight gradients
#define VLP_SS_ATTN 4      // attention context / dQ dK dV
#define VLP_SS_LN 8        // LayerNorm outputs
#define VLP_SS_EW 16       // element-wise kernels
#define VLP_SS_SAVED 32    // NT GEMM side outputs that only the backward pass reads (pre-activation / gelu' of the FFN)
#ifndef VLP_STREAM_STORES
#define VLP_STREAM_STORES 0
#endif
template <int FAM>
DEVFN void st8_out(f16* p, f16x8 v) {
    if constexpr ((VLP_STREAM_STORES & FAM) != 0) __builtin_nontemporal_store(v, reinterpret_cast<f16x8*>(p));
    else st8(p, v);
}
template <int FAM>
DEVFN void st4_out(f16* p, f16x4 v) {
    if constexpr ((VLP_STREAM_STORES & FAM) != 0) __builtin_nontemporal_store(v, reinterpret_cast<f16x4*>(p));
    else st4(p, v);
}
