// Fused masked-softmax self-attention for gfx950 (forward + backward).
//
// Replaces modeling.py:279-302 of the reference (BertSelfAttention.forward after the Q/K/V Linears):
// transpose_for_scores (:262-266), QK^T / sqrt(d) (:284-287), + extended mask (:289, built :807-833),
// Softmax(-1) (:292), dropout on the probabilities (:296), P.V (:298), permute+contiguous (:299-302),
// and the autograd backward of all of it.  The [B,heads,L,L] score/probability tensors the reference
// materialises four times per layer never leave registers here.
//
// L <= 256 (the reference runs L = 123 or 167), head_dim = 64: one workgroup owns one (batch, head), the
// operands of that head sit in LDS as ROW-MAJOR [row][64] tiles (128-B rows, 16-B chunk c of row r stored
// at chunk c ^ (r & 7)), each wave walks 16-row tiles.  All contractions run on v_mfma_f32_16x16x32_f16.
// The kernels compute TRANSPOSED products (S^T = K.Q^T, O^T = V^T.P^T, dQ^T = K^T.dS^T, ...) so that, in the
// MFMA C/D layout (col = lane&15, row = 4*(lane>>4)+reg), a lane owns one query (or key) column: softmax row
// statistics are lane-local, a C/D tile is re-used *in registers* as the B operand of the next MFMA (P^T / dS^T
// tiles feed PV / dQ / dK / dV directly, no LDS round trip), and outputs are 4 consecutive head-dim values per
// lane (8-byte stores).  The k-slot order of such a re-used tile pair (t, t+1) is rows {16t+4g+e, 16(t+1)+4g+e};
// the other operand (V^T, K^T, Q^T, dO^T) is gathered in that same slot order straight from the row-major tile
// with the CDNA4 LDS transpose read ds_read_b64_tr_b16 (two per fragment) -- no transposed copy is ever built.
// With the (r & 7) chunk swizzle both access patterns are bank-conflict free: ds_read_b128 row fragments
// (16 rows x one chunk) and transpose reads (8 rows x 32 contiguous bytes per half-wave).
#include "common.h"
#include <stdlib.h>

#define HD 64            // head dim
#define ATT_THREADS 256         // dQ kernel (240 VGPRs: 2 workgroups of 4 waves per CU)
#define ATT_WAVES 4
// forward and dK/dV kernels fit 128 VGPRs: 8 waves per workgroup, 2 workgroups per CU -> 4 waves per SIMD to hide the global / LDS
// latency chain of a tile, and the 11 tiles of L = 167 take 2 rounds over the waves instead of 3
#define ATT_THREADS8 512
#define ATT_WAVES8 8

DEVFN int swzk(int r) { return r & 7; }

struct AttnParams {
    const f16* qkv; int64_t ld_qkv;
    const uint8_t* mask;
    const uint8_t* mask_t;                 // [B][Lp][Lp] transposed byte mask (key-major), backward only
    f16* ctx; int64_t ld_ctx;              // fwd out / bwd in
    const f16* dctx; int64_t ld_dctx;
    float* lse;
    f16* dqkv; int64_t ld_dqkv;
    float* delta;
    int B, L, Lp, heads, H;
    float scale;
    DropCtx drop;
    // forward only: queries and keys/values may come from different buffers / lengths (incremental decoding).  Training is
    // the special case Lq = Lk = L with q | k | v interleaved in one packed buffer.
    const f16* q; int64_t ld_q; int64_t bs_q;      // row (b, i) at q + (b*bs_q + i)*ld_q   (+ h*64)
    const f16* k; const f16* v; int64_t ld_kv; int64_t bs_kv;
    int Lq, Lk;
    // decode with beams: rows [0, n_prefix) of sequence b come from the per-SAMPLE cache k2 / v2 (row b / beams), the rest from k / v
    const f16* k2; const f16* v2; int64_t bs_kv2; int n_prefix, beams;
    // 1: (query tile, key tile) blocks whose probabilities are EXACTLY zero (every key masked or padding for every query of the tile,
    // and every query of the tile has at least one attended key, so exp2(-10000 log2 e - max) underflows to 0) are skipped: bit-identical
    // results, ~1/3 fewer tiles under seq2seq masks (regions never attend caption tokens; tokens attend causally)
    int skip;
    // padding-free (packed) rows, training kernels: sample b owns rows [row_off[b], row_off[b+1]) of q / k / v / ctx / dctx / dqkv (its first
    // n_b positions); mask, lse, delta and the dropout element stay logical [B, L, ..].  nullptr: row b*L + l, n_b = L.
    const int32_t* row_off;
};

#ifdef VLP_ATTN_TRACE
// investigation build only (tools/attn_trace.sh): wave 0 of every workgroup of attn_fwd records s_memtime at phase boundaries
__device__ unsigned long long g_attn_trace[4096 * 8];
#define TRACE(slot) do { if (wid == 0 && lane == 0 && blockIdx.x < 4096) g_attn_trace[blockIdx.x * 8 + (slot)] = __builtin_readcyclecounter(); } while (0)
extern "C" int vlp_debug_read_attn_trace(void* dst, int64_t bytes) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_attn_trace), bytes); }
#else
#define TRACE(slot) do { } while (0)
#endif
#define ANY_ATTEND(w) (((w) & 0x01010101u) != 0u)        /* some byte of the mask word is 1 (bytes are 0, 1 or 2) */

// ---- LDS staging helpers ---------------------------------------------------------------------
// row-major, 128-B rows, chunk-swizzled: dst[key][64]; rows >= L are zero.
// rows < n_first come from src_first (shared prefix of a beam group), the others from src
DEVFN void stage_rowmajor(f16* dst, const f16* src, int64_t ld, int L, int Lp, int tid, int nthreads = ATT_THREADS, const f16* src_first = nullptr,
                          int n_first = 0) {
    for (int idx = tid; idx < Lp * 8; idx += nthreads) {
        const int r = idx >> 3, c = idx & 7;
        u32x4 v = (u32x4){0, 0, 0, 0};
        if (r < L) v = *reinterpret_cast<const u32x4*>((r < n_first ? src_first : src) + (int64_t)r * ld + c * 8);
        *reinterpret_cast<u32x4*>(dst + r * HD + ((c ^ swzk(r)) << 3)) = v;
    }
}
// Two tiles at once with ALL global loads in flight before the first LDS store: the loop above compiles to one HBM round trip per
// 16-byte piece (load, wait, store; 12 in a row at L = 167 with 256 threads) -- measured 27 % of the forward kernel's time per workgroup
// (tools/attn_trace.py) -- this form pays one.  The loads go through buffer descriptors whose extent ends with row L - 1, so rows
// >= L read as zero in hardware: a guarded global load per piece made the compiler carry both tiles as one select-merged register
// block (~400 v_mov of the 3 000 instructions of a dq workgroup).
DEVFN __amdgpu_buffer_rsrc_t rows_rsrc(const f16* src, int64_t ld, int L) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (int)(((int64_t)(L - 1) * ld + HD) * 2), 0x00020000);
}
template <int LP_, int NTHR>
DEVFN void stage_two_rowmajor(f16* dst0, const f16* src0, int64_t ld0, f16* dst1, const f16* src1, int64_t ld1, int L, int tid) {
    constexpr int IT = (LP_ * 8 + NTHR - 1) / NTHR;
    const __amdgpu_buffer_rsrc_t r0 = rows_rsrc(src0, ld0, L), r1 = rows_rsrc(src1, ld1, L);
    u32x4 v0[IT], v1[IT];
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        const int idx = tid + i * NTHR, r = idx >> 3, c = idx & 7;
        v0[i] = __builtin_amdgcn_raw_buffer_load_b128(r0, (r * (int)ld0 + c * 8) * 2, 0, 0);
        v1[i] = __builtin_amdgcn_raw_buffer_load_b128(r1, (r * (int)ld1 + c * 8) * 2, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        const int idx = tid + i * NTHR, r = idx >> 3, c = idx & 7;
        if (idx < LP_ * 8) {
            *reinterpret_cast<u32x4*>(dst0 + r * HD + ((c ^ swzk(r)) << 3)) = v0[i];
            *reinterpret_cast<u32x4*>(dst1 + r * HD + ((c ^ swzk(r)) << 3)) = v1[i];
        }
    }
}
typedef __fp16 fp16x4_t __attribute__((__vector_size__(4 * sizeof(__fp16))));
DEVFN f16x4 lds_tr_read(const f16* p) {
    fp16x4_t t = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)(p));
    return __builtin_bit_cast(f16x4, t);
}
// MFMA A-operand fragment of the TRANSPOSE of a row-major swizzled tile: the lane receives column col0 + (lane&15) for the
// 8 k-slots {r_first + 4g + e, r_second + 4g + e} (e = 0..3): two ds_read_b64_tr_b16; lane s of a 16-lane group supplies
// the 8-byte piece (row + (s>>2), cols col0 + 4*(s&3) .. +3).
// MFMA fragments are assembled as four 32-bit words (two fp16 each) and bit-cast: writing fp16 ELEMENTS of an f16x8 from different
// branches makes the compiler extract / re-insert 16-bit halves (v_lshrrev + v_perm + v_mov: ~1 000 of the 3 500 instructions of a
// forward query tile), and a wave issues one instruction per 4-5 cycles whatever else is resident (tools/attn_trace.py).
DEVFN uint32_t pack_f16x2(float a, float b) {
    const f16x2 v = (f16x2){(f16)a, (f16)b};
    return __builtin_bit_cast(uint32_t, v);
}
DEVFN f16x8 words_f16x8(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
    const u32x4 w = (u32x4){a, b, c, d};
    return __builtin_bit_cast(f16x8, w);
}
DEVFN f16x8 tr_frag(const f16* tile, int r_first, int r_second, int col0, int g, int li) {
    const int ra = r_first + 4 * g + (li >> 2), rb = r_second + 4 * g + (li >> 2);
    const int c = col0 + 4 * (li & 3);
    const u32x2 a = __builtin_bit_cast(u32x2, lds_tr_read(tile + ra * HD + (((c >> 3) ^ (ra & 7)) << 3) + (c & 4)));
    const u32x2 b = __builtin_bit_cast(u32x2, lds_tr_read(tile + rb * HD + (((c >> 3) ^ (rb & 7)) << 3) + (c & 4)));
    return words_f16x8(a[0], a[1], b[0], b[1]);
}

// Mask bytes (vlp_mask_pack): 1 = attend (+0), 0 = masked (-10000, modeling.py:832), 2 = padding column past L (excluded: -inf).
// Rows are Lp bytes.
// Additive mask term of 4 consecutive keys, in the log2 domain: out[e] = c0 + c1 * [byte == 1] (byte in {0, 1}), -inf for byte 2.  c1 = 10000*log2(e), c0 =
// -10000*log2(e) (+ a per-row offset such as -lse).  `full` (uniform) says that no byte of the word can be 2: one v_cvt_f32_ubyteN and
// one fma per element instead of shift / compare / select chains.
#define LOG2E_F 1.4426950408889634f
#define LN2_F 0.6931471805599453f
#define MASK_C1 (10000.0f * LOG2E_F)
// Key-major mask copy (out_t of vlp_mask_pack / vlp_mask_build, backward only): row `key` holds Lp bytes, one per query, in LANE
// ORDER: the byte of query q = 16 t + 4 g + e sits at g * (Lp / 4) + 4 t + e.  A lane of the key-owner backward kernels (key column
// li, row group g) wants the words (4 queries 16 t + 4g .. +3) of ALL query tiles t: they are CONSECUTIVE in this order (48 bytes at
// L = 167: three 16-byte loads, and a wave covers its 16 key rows exactly once) -- in plain query order they were 12 scattered
// 4-byte gathers per lane, the slowest loads of the one-kernel backward's prologue.
DEVFN int maskt_pos(int q, int Lp) { return ((q >> 2) & 3) * (Lp >> 2) + ((q >> 4) << 2) + (q & 3); }
DEVFN int maskt_query(int pos, int Lp) { const int gsz = Lp >> 2, g = pos / gsz, rem = pos - g * gsz; return ((rem >> 2) << 4) + 4 * g + (rem & 3); }
// 4 mask bytes of one row at columns key0..key0+3, branch-free (the address is clamped into the row; columns past the row read as 2)
DEVFN uint32_t mask_word(const uint8_t* mrow, int key0, int Lp) {
    const uint32_t w = *reinterpret_cast<const uint32_t*>(mrow + min(key0, Lp - 4));
    return key0 < Lp ? w : 0x02020202u;
}
DEVFN void mask4w(uint32_t w, bool full, float c0, float out[4]) {
    if (full) {
#pragma unroll
        for (int e = 0; e < 4; ++e) out[e] = fmaf((float)((w >> (8 * e)) & 0xffu), MASK_C1, c0);
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const uint32_t v = (w >> (8 * e)) & 0xffu;
            out[e] = v == 1u ? c0 + MASK_C1 : (v == 0u ? c0 : -INFINITY);
        }
    }
}

// =================================================================================================
// forward
// =================================================================================================
// NW = waves per workgroup.  8 for most shapes; 4 at NT = 12 (129 <= L <= 192, the training shape L = 167): B x heads = 768 workgroups
// on 256 CUs is 1.5 rounds of the 2 x 8-wave workgroups a CU holds -- with 4-wave workgroups three fit (3 x 48 KB LDS, 12 waves), all 768
// are resident at once, and the 11 query tiles of L = 167 spread over 4 waves (3 passes, 92 % busy) instead of 8 (2 passes, 69 %).
// One 16-query tile of the forward: S^T = K Q^T over the live key tiles, log2-domain softmax with the byte mask, dropout, O^T = V^T P^T,
// context rows + lse stored.  Shared by the one-workgroup-per-(batch, head) kernel below and the persistent streaming kernel: the tile code
// is the same, so both produce the same bits.  Ks / Vs: the head's K and V rows in LDS (row-major, chunk-swizzled); rows >= the staged
// count are zero.  `first`: the wave's first tile of the launch (phase trace only).
// packed 16-bit integer ops on a 32-bit word (written as asm: from vector C the compiler falls back to per-half compare / select / v_perm)
DEVFN uint32_t pk_sub_sat_u16(uint32_t a, uint32_t b) { uint32_t d; asm("v_pk_sub_u16 %0, %1, %2 clamp" : "=v"(d) : "v"(a), "v"(b)); return d; }
DEVFN uint32_t pk_min_u16(uint32_t a, uint32_t b) { uint32_t d; asm("v_pk_min_u16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
DEVFN uint32_t pk_mul_lo_u16(uint32_t a, uint32_t b) { uint32_t d; asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
// Round 5 (late): the tile is straight-line code per key-tile PAIR, like the backward's pair loop -- dropout is a compile-time switch, the
// padding-column test is folded into the additive mask base (and exists only for the last four key tiles: 16 (NT - 4) < L by the
// launcher's choice of NT), exp2 / row sum / dropout / fp16 packing / P.V of a pair form ONE block, and the dropout zeroing works on the packed
// fp16 pair with three packed 16-bit instructions (saturating subtract, min, multiply) instead of shift / compare / select per element.
// The first form had a scalar branch per key tile in each of its four loops plus run-time dropout / interior-tile switches: ~60 basic blocks
// per query tile with 168 hazard s_nops between them.  Same bits: a dead tile inside a live pair is now computed, and its probabilities are the
// exact zeros (exp2 of -14 000) the skipped form assumed.
// the global loads of a query tile: this lane's Q fragments and (NT <= 12) its mask words for all key tiles -- independent loads, one L2
// round trip.  The kernel issues the loads of a wave's FIRST tile before the K / V staging, so that they travel with it: in the phase
// trace every wave of a workgroup sat ~8 000 cycles (15 % of the workgroup's life) behind them right after the staging barrier.
template <int NT>
DEVFN void attn_fwd_tile_load(const AttnParams& p, const f16* qbase, int b, int Lq, int nq, int qt, int g, int li, f16x8 (&qf)[2], uint32_t (&mw)[NT <= 12 ? NT : 1]) {
    const int qc = min(qt * 16 + li, nq - 1);
    int gq = g;                             // opaque copy: keeps per-key index math inside the loop (no LICM + spills)
    asm volatile("" : "+v"(gq));
    const f16* qrow = qbase + (int64_t)qc * p.ld_q;
    qf[0] = ld8(qrow + g * 8);
    qf[1] = ld8(qrow + 32 + g * 8);
    if (NT <= 12) {
        const uint8_t* mrow = p.mask + ((int64_t)b * Lq + qc) * p.Lp;
#pragma unroll
        for (int t = 0; t < NT; ++t) mw[NT <= 12 ? t : 0] = mask_word(mrow, t * 16 + 4 * gq, p.Lp);
    }
}
template <int NT, bool DROP>
DEVFN void attn_fwd_tile_compute(const AttnParams& p, const f16* Ks, const f16* Vs, int b, int h, int L, int Lq, int nq, int64_t rbo,
                                 int qt, int g, int li, int wid, int lane, bool first, const f16x8 (&qf)[2], const uint32_t (&mw)[NT <= 12 ? NT : 1]) {
    (void)wid; (void)lane;
#ifdef VLP_ISA_MARKERS
    asm volatile("; TILE_BEGIN drop=%0" :: "n"((int)DROP));
#endif
    const int q = qt * 16 + li;             // this lane's query (column of every transposed tile)
    const int qc = min(q, nq - 1);
    int gq = g;
    asm volatile("" : "+v"(gq));
    constexpr bool PRELOAD = NT <= 12;           // L > 192: the words would push the kernel into spills -- fetch them per tile there
    const uint8_t* mrow = p.mask + ((int64_t)b * Lq + qc) * p.Lp;
    if (first) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); TRACE(3); }      // first tile: Q + mask words have arrived
    // key tiles that are dead for ALL 16 queries of this wave's tile (bit t clear); only when every query row has an attended key
    uint32_t live = 0xffffffffu;
    if (PRELOAD && p.skip) {
        uint32_t any1 = 0u;
#pragma unroll
        for (int t = 0; t < NT; ++t) any1 |= mw[PRELOAD ? t : 0] & 0x01010101u;
        int rowlive = any1 != 0u;
        rowlive |= __shfl_xor(rowlive, 16, 64);
        rowlive |= __shfl_xor(rowlive, 32, 64);
        if (__all(rowlive)) {
            live = 0u;
#pragma unroll
            for (int t = 0; t < NT; ++t) live |= (__any(ANY_ATTEND(mw[PRELOAD ? t : 0])) ? 1u : 0u) << t;
        }
        live = __builtin_amdgcn_readfirstlane(live);
    }
    // S^T tiles: rows = keys 16t + 4g + reg, col = query.  Groups of 4 key tiles: one scalar branch per group (a branch per tile
    // makes every tile its own basic block: two LDS reads, a full lgkmcnt wait, two dependent MFMAs -- ~500 cycles per tile in the
    // trace); inside a group the 8 fragment reads are issued together and the 8 MFMAs follow.  A dead group's tiles are never read below
    // (its pairs are dead), so they are not initialised either.
    f32x4 s[NT];
#pragma unroll
    for (int t0 = 0; t0 < NT; t0 += 4) {
        if (!((live >> t0) & 15u)) continue;
        f16x8 kf[4][2];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int kr = (t0 + j) * 16 + li;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) kf[j][ks] = ld8(Ks + kr * HD + (((ks * 4 + g) ^ swzk(kr)) << 3));
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) s[t0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[j][0], qf[0], (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) s[t0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[j][1], qf[1], s[t0 + j], 0, 0, 0);
    }
    if (first) TRACE(4);       // S MFMAs issued
    // scores in the log2 domain: s2 = s * scale * log2(e) + mask term; softmax = exp2(s2 - max) / sum.  Mask term of a key: byte * C1 + base,
    // base = -C1 for a real key (byte 0 / 1) and -inf for a padding column (byte 2; 2 C1 - inf = -inf)
    const float sc2 = p.scale * LOG2E_F;
    float mx = -INFINITY;
#pragma unroll
    for (int u = 0; u < NT / 2; ++u) {
        if (!((live >> (2 * u)) & 3u)) continue;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int t = 2 * u + hh;
            const uint32_t w = PRELOAD ? mw[PRELOAD ? t : 0] : mask_word(mrow, t * 16 + 4 * gq, p.Lp);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float base = (t < NT - 4 || t * 16 + 4 * gq + r < L) ? -MASK_C1 : -INFINITY;      // (t < NT - 4: compile time)
                s[t][r] = fmaf(s[t][r], sc2, fmaf((float)((w >> (8 * r)) & 0xffu), MASK_C1, base));
                mx = fmaxf(mx, s[t][r]);
            }
        }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    if (first) TRACE(5);       // scores scaled + masked, row maximum known

    // per pair: P^T (UNnormalised exp2 values in (0, 1], dropped entries zeroed) as the fp16 B-operand fragment of the pair, then
    // O^T += V^T P^T (rows = head-dim 16n + 4g + reg, col = query).  The normalisation and the dropout scale 1/(1-p) are applied to the 16
    // outputs of the lane instead of its 4*NT probabilities.  dropout element = (row (b, h, q), col key)
    const uint32_t rk = DROP ? drop_rowkey(p.drop, ((uint64_t)b * p.heads + h) * (uint64_t)Lq + (uint64_t)qc) : 0u;
    // columns (keys) of tile t held by this lane: 16t + 4g + {0..3} = two hash pairs; pair key advances by 8*PHI per tile
    const uint32_t pk0 = drop_pairkey(rk, (uint32_t)(4 * gq));
    const uint32_t tm1 = (p.drop.thresh - 1u) * 0x10001u;          // (threshold - 1) in both halves: hash half <= thresh - 1 = dropped
    const uint32_t ones = 0x00010001u;
    float sum = 0.f;
    f32x4 o[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) o[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < NT / 2; ++u) {
        if (!((live >> (2 * u)) & 3u)) continue;       // both key tiles of the pair dead: P = 0 exactly
        uint32_t pw[4];                                // words 2hh, 2hh+1 = the four probabilities of tile 2u+hh
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int t = 2 * u + hh;
            float pr[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                pr[r] = __builtin_amdgcn_exp2f(s[t][r] - mx);
                sum += pr[r];
            }
            pw[2 * hh] = pack_f16x2(pr[0], pr[1]);
            pw[2 * hh + 1] = pack_f16x2(pr[2], pr[3]);
            if (DROP) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const uint32_t hj = mix32(pk0 + (uint32_t)(8 * t + j) * VLP_PHI);      // low half: key 4g + 2j, high half: key 4g + 2j + 1
                    pw[2 * hh + j] = pk_mul_lo_u16(pw[2 * hh + j], pk_min_u16(pk_sub_sat_u16(hj, tm1), ones));      // keep word: 0 = dropped (hash half < threshold), 1 = kept
                }
            }
        }
        f16x8 vfr[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) vfr[n] = tr_frag(Vs, 32 * u, 32 * u + 16, 16 * n, g, li);
        const f16x8 pfu = words_f16x8(pw[0], pw[1], pw[2], pw[3]);
#pragma unroll
        for (int n = 0; n < 4; ++n) o[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vfr[n], pfu, o[n], 0, 0, 0);
    }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = p.drop.scale / sum;
    if (p.lse && g == 0 && q < nq) p.lse[((int64_t)b * p.heads + h) * Lq + q] = (mx + __builtin_amdgcn_logf(sum)) * LN2_F;
    if (q < nq) {
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            f16x4 ov = (f16x4){(f16)(o[n][0] * inv), (f16)(o[n][1] * inv), (f16)(o[n][2] * inv), (f16)(o[n][3] * inv)};
            st4_out<VLP_SS_ATTN>(p.ctx + (rbo + q) * p.ld_ctx + h * HD + n * 16 + 4 * g, ov);
        }
    }
    if (first) TRACE(6);       // first tile stored
#ifdef VLP_ISA_MARKERS
    asm volatile("; TILE_END drop=%0" :: "n"((int)DROP));
#endif
}
// load + compute of one tile (the streaming kernel's form)
template <int NT, bool DROP>
DEVFN void attn_fwd_tile(const AttnParams& p, const f16* Ks, const f16* Vs, const f16* qbase, int b, int h, int L, int Lq, int nq, int64_t rbo,
                         int qt, int g, int li, int wid, int lane, bool first) {
    f16x8 qf[2];
    uint32_t mw[NT <= 12 ? NT : 1];
    attn_fwd_tile_load<NT>(p, qbase, b, Lq, nq, qt, g, li, qf, mw);
    attn_fwd_tile_compute<NT, DROP>(p, Ks, Vs, b, h, L, Lq, nq, rbo, qt, g, li, wid, lane, first, qf, mw);
}

template <int NT, int NW>   // NT = LP / 16 key tiles (4, 8, 12 or 16)
__global__ __launch_bounds__(NW * 64, NW == 4 ? 3 : (NT <= 12 ? 4 : 2)) void attn_fwd_kernel(AttnParams p) {   // (threads, min waves per SIMD)
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int LP = NT * 16;
    f16* Ks = reinterpret_cast<f16*>(smem_raw);            // [LP][64] swizzled
    f16* Vs = Ks + LP * HD;                                // [LP][64] swizzled
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, li = lane & 15;
    const int b = blockIdx.x / p.heads, h = blockIdx.x % p.heads;
    const int L = p.Lk;                        // keys (rows >= Lk of the LDS tiles are zero, mask bytes there are 2)
    const int Lq = p.Lq;
    // packed rows (training, Lq == Lk): the sample's nb kept positions start at row rb; positions >= nb are attended by no kept query
    // (mask byte 0 -> probability exactly 0 in the dense run as well), so their K / V rows are staged as zeros and their query tiles skipped
    const int rb_pk = p.row_off ? p.row_off[b] : 0;
    const int nb = p.row_off ? p.row_off[b + 1] - rb_pk : L;       // key rows to stage
    const int nq = p.row_off ? nb : Lq;                            // query rows to compute
    const int64_t rbq = p.row_off ? (int64_t)rb_pk : (int64_t)b * p.bs_q, rbk = p.row_off ? (int64_t)rb_pk : (int64_t)b * p.bs_kv;
    const int64_t rbo = p.row_off ? (int64_t)rb_pk : (int64_t)b * Lq;      // ctx rows
    const f16* qbase = p.q + rbq * p.ld_q + h * HD;
    const f16* kbase = p.k + rbk * p.ld_kv + h * HD;
    const f16* vbase = p.v + rbk * p.ld_kv + h * HD;

    const f16* kpre = p.n_prefix ? p.k2 + (int64_t)(b / p.beams) * p.bs_kv2 * p.ld_kv + h * HD : nullptr;
    const f16* vpre = p.n_prefix ? p.v2 + (int64_t)(b / p.beams) * p.bs_kv2 * p.ld_kv + h * HD : nullptr;
    TRACE(0);
    // the wave's first query tile: its Q fragments and mask words are requested BEFORE the K / V rows and arrive with them
    const int nqt = (nq + 15) / 16;
    f16x8 qf[2];
    uint32_t mw[NT <= 12 ? NT : 1];
    if (wid < nqt) attn_fwd_tile_load<NT>(p, qbase, b, Lq, nq, wid, g, li, qf, mw);
    if (p.n_prefix) {       // beam decode: rows < n_prefix come from the per-sample prefix cache
        stage_rowmajor(Ks, kbase, p.ld_kv, L, LP, tid, NW * 64, kpre, p.n_prefix);
        stage_rowmajor(Vs, vbase, p.ld_kv, L, LP, tid, NW * 64, vpre, p.n_prefix);
    } else {
        stage_two_rowmajor<LP, NW * 64>(Ks, kbase, p.ld_kv, Vs, vbase, p.ld_kv, nb, tid);
    }
    TRACE(1);
    __syncthreads();
    TRACE(2);

    // (Round 4, measured and not kept: requesting the Q fragments + mask words of the wave's NEXT query tile while the current one is
    // computed -- 149 instead of 108 VGPRs, same 3 workgroups per CU -- 36.5 us against 34.3 us at B = 64, tools/attn_lab.py.)
    for (int qt = wid; qt < nqt; qt += NW) {
        if (qt != wid) attn_fwd_tile_load<NT>(p, qbase, b, Lq, nq, qt, g, li, qf, mw);
        if (p.drop.thresh) attn_fwd_tile_compute<NT, true>(p, Ks, Vs, b, h, L, Lq, nq, rbo, qt, g, li, wid, lane, qt == wid, qf, mw);
        else attn_fwd_tile_compute<NT, false>(p, Ks, Vs, b, h, L, Lq, nq, rbo, qt, g, li, wid, lane, qt == wid, qf, mw);
    }
    TRACE(7);
}

#ifdef VLP_LAB_BUILD
// =================================================================================================
// forward, persistent streaming form (round 5; the training shape 129 <= L <= 192, NT = 12)
// =================================================================================================
// INVESTIGATION BUILDS ONLY (-DVLP_LAB_BUILD, VLP_ATTN_FWD_STREAM=1): built in round 5 as asked by two verdicts, bit-identical to the kernel
// above, 35.0 -> 32.4 us per layer in the cold-operand lab with the first tile code (B = 64; 15.8 us for one item per workgroup, +8.2 us per
// further item) and NO gain in the step (9.511 / 9.516 vs 9.524 / 9.512 ms/step, profiles/r05_attention_forward_streaming_lab.txt): inside
// the step the packed QKV rows were written by the previous kernel and come from L2 / the Infinity Cache, so the load phase this design
// hides is already short.  With the rewritten tile (889 instead of ~1 400 VALU instructions) and the first tile's loads ahead of the staging
// the plain kernel caught up: 30.4 us both (profiles/r05_attention_forward_tile_rewrite.txt).  A second form -- static rotating tile
// assignment, so that a wave's next tile is known and its Q / mask loads can be requested one item ahead -- was bit-identical and SLOWER
// (35.0 us, +0.045 ms/step): the dynamic hand-out below is worth more than the hidden round trip.  It lives in the history only.
// The kernel above starts all B x heads workgroups at once: every one of them first waits for its K / V rows, then computes, three 4-wave
// workgroups per CU in the same phase -- 32 us per layer at B = 64 where the VALU work of the tiles (the forward is VALU-issue-bound:
// exp2, mask term, dropout hash, fp16 packing: ~27 instructions per score) is ~14 us per CU.  Here ONE 12-wave workgroup per CU walks its
// (batch, head) items through THREE K / V buffer pairs in LDS (144 KB):
//   * producer side, every wave: the K / V rows of item k+1 (its 1/12 share: 4 x 16 bytes per lane, requested one item earlier into
//     registers through buffer descriptors that zero-fill past the sample's rows) are written into buffer (k+1) % 3 at the START of item k
//     -- a buffer whose previous occupant (item k-2) is known to be consumed from an LDS counter -- then item k+2 is requested;
//   * consumer side: the 16-query tiles of item k are TASKS handed out by an LDS atomic counter, so a wave that drew a cheap tile (region
//     rows attend 7 of 12 key tiles under the seq2seq mask) simply draws the next one, also across the item boundary: there is NO
//     workgroup barrier per item -- a wave waits only for "all 12 shares of item k are in LDS" (counter), which the slowest wave wrote
//     a whole item earlier.
// The tile code is attn_fwd_tile, shared with the kernel above: identical bits.  Counters are monotonic over the launch:
//   ready[b] counts shares written, task[b] hands out indices (every wave overshoots exactly once per item: the next occupant's base
//   moves by tiles + waves), done[b] counts finished tiles.
#define AFS_WAVES 12
#define AFS_NBUF 3
template <int NT>
__global__ __launch_bounds__(AFS_WAVES * 64, 1) void attn_fwd_stream_kernel(AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int LP = NT * 16, TILE = LP * HD, NTHR = AFS_WAVES * 64;
    constexpr int IT = (LP * 8 + NTHR - 1) / NTHR;                 // 16-byte pieces per thread and tile (2 at LP = 192)
    f16* bufs = reinterpret_cast<f16*>(smem_raw);                  // [AFS_NBUF][K | V][LP][64] swizzled
    int* ctr = reinterpret_cast<int*>(bufs + AFS_NBUF * 2 * TILE); // task[3] | done[3] | ready[3]
    int* task = ctr; int* done = ctr + 4; int* ready = ctr + 8;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, li = lane & 15;
    const int L = p.Lk, Lq = p.Lq;
    const int nitems = p.B * p.heads;
    const int n_my = ((int)blockIdx.x < nitems) ? (nitems - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    if (tid < 12) ctr[tid] = 0;
    __syncthreads();
    if (n_my == 0) return;

    u32x4 rk[IT], rv[IT];
    auto rows_of = [&](int item, int& b_, int& h_, int& rb_, int& nb_) {
        b_ = item / p.heads; h_ = item % p.heads;
        rb_ = p.row_off ? p.row_off[b_] : b_ * (int)p.bs_kv;
        nb_ = p.row_off ? p.row_off[b_ + 1] - rb_ : L;
    };
    auto request = [&](int item) {          // this thread's pieces of the item's K and V rows -> registers (rows >= nb read as zero)
        int b_, h_, rb_, nb_;
        rows_of(item, b_, h_, rb_, nb_);
        const __amdgpu_buffer_rsrc_t r0 = rows_rsrc(p.k + (int64_t)rb_ * p.ld_kv + h_ * HD, p.ld_kv, nb_),
                                     r1 = rows_rsrc(p.v + (int64_t)rb_ * p.ld_kv + h_ * HD, p.ld_kv, nb_);
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            const int idx = tid + i * NTHR, r = idx >> 3, c = idx & 7;
            rk[i] = __builtin_amdgcn_raw_buffer_load_b128(r0, (r * (int)p.ld_kv + c * 8) * 2, 0, 0);
            rv[i] = __builtin_amdgcn_raw_buffer_load_b128(r1, (r * (int)p.ld_kv + c * 8) * 2, 0, 0);
        }
    };
    auto publish = [&](int buf) {           // registers -> LDS buffer `buf`, then this wave's share is announced
        f16* Kd = bufs + buf * 2 * TILE;
        f16* Vd = Kd + TILE;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            const int idx = tid + i * NTHR, r = idx >> 3, c = idx & 7;
            if (idx < LP * 8) {
                *reinterpret_cast<u32x4*>(Kd + r * HD + ((c ^ swzk(r)) << 3)) = rk[i];
                *reinterpret_cast<u32x4*>(Vd + r * HD + ((c ^ swzk(r)) << 3)) = rv[i];
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_fetch_add(ready + buf, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    auto wait_ge = [&](int* c, int target) {
        while (__hip_atomic_load(c, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < target) __builtin_amdgcn_s_sleep(1);
    };

    int task_base[AFS_NBUF] = {0, 0, 0}, done_base[AFS_NBUF] = {0, 0, 0};      // per buffer: first task index / finished tiles of earlier occupants
    request(blockIdx.x);
    publish(0);
    if (n_my > 1) request(blockIdx.x + gridDim.x);
#pragma unroll 1
    for (int k = 0; k < n_my; ++k) {
        const int bk = k % AFS_NBUF;
        const int item = blockIdx.x + k * gridDim.x;
        int b, h, rb, nb;
        rows_of(item, b, h, rb, nb);
        const int nq = p.row_off ? nb : Lq;
        const int nqt = (nq + 15) / 16;
        // ---- producer: item k+1 into its buffer (free once item k-2 is consumed), then the request for item k+2
        if (k + 1 < n_my) {
            const int bn = (k + 1) % AFS_NBUF;
            wait_ge(done + bn, done_base[bn]);
            publish(bn);
            if (k + 2 < n_my) request(blockIdx.x + (k + 2) * gridDim.x);
        }
        // ---- consumer: tiles of item k, drawn from the buffer's task counter
        const f16* Ks = bufs + bk * 2 * TILE;
        const f16* Vs = Ks + TILE;
        const int64_t rbq = p.row_off ? (int64_t)rb : (int64_t)b * p.bs_q, rbo = p.row_off ? (int64_t)rb : (int64_t)b * Lq;
        const f16* qbase = p.q + rbq * p.ld_q + h * HD;
        bool waited = false;
        for (;;) {
            int t = 0;
            if (lane == 0) t = __hip_atomic_fetch_add(task + bk, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            t = __builtin_amdgcn_readfirstlane(t) - task_base[bk];
            if (t >= nqt) break;
            if (!waited) { wait_ge(ready + bk, AFS_WAVES * (k / AFS_NBUF + 1)); waited = true; }
            if (p.drop.thresh) attn_fwd_tile<NT, true>(p, Ks, Vs, qbase, b, h, L, Lq, nq, rbo, t, g, li, wid, lane, false);
            else attn_fwd_tile<NT, false>(p, Ks, Vs, qbase, b, h, L, Lq, nq, rbo, t, g, li, wid, lane, false);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                  // this tile's K / V fragment reads have returned
            if (lane == 0) __hip_atomic_fetch_add(done + bk, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        task_base[bk] += nqt + AFS_WAVES;          // every wave drew exactly one index past the end
        done_base[bk] += nqt;
    }
}


#endif      // VLP_LAB_BUILD (streaming forward)

#ifdef VLP_LAB_BUILD      // the two-kernel backward (dQ, then dK / dV): the form the one-kernel backward was validated against; investigation builds only
// =================================================================================================
// backward, part 1: dQ (and delta = rowsum(dO * O)).  Transposed orientation, query column per lane.
// =================================================================================================
// NT = 16 (192 < L <= 256) needs ~300 registers per lane (16 score tiles + 8 dS fragments live): one workgroup per CU there (512-entry
// unified VGPR/AGPR file, no spills) instead of two with 43 spilled VGPRs
template <int NT>
__global__ __launch_bounds__(ATT_THREADS, NT <= 12 ? 2 : 1) void attn_bwd_dq_kernel(AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int LP = NT * 16;
    f16* Ks = reinterpret_cast<f16*>(smem_raw);     // [LP][64] swizzled
    f16* Vs = Ks + LP * HD;                         // [LP][64] swizzled
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, li = lane & 15;
    const int b = blockIdx.x / p.heads, h = blockIdx.x % p.heads;
    const int L = p.L;
    const f16* qbase = p.qkv + (int64_t)b * L * p.ld_qkv + h * HD;
    const f16* kbase = qbase + p.H;
    const f16* vbase = qbase + 2 * p.H;

    stage_two_rowmajor<LP, ATT_THREADS>(Ks, kbase, p.ld_qkv, Vs, vbase, p.ld_qkv, L, tid);
    __syncthreads();

    const int nqt = (L + 15) / 16;
    for (int qt = wid; qt < nqt; qt += ATT_WAVES) {
        const int q = qt * 16 + li;
        const int qc = min(q, L - 1);
        int gq = g;
        asm volatile("" : "+v"(gq));
        const f16* qrow = qbase + (int64_t)qc * p.ld_qkv;
        const f16* dorow = p.dctx + ((int64_t)b * L + qc) * p.ld_dctx + h * HD;
        const f16* orow = p.ctx + ((int64_t)b * L + qc) * p.ld_ctx + h * HD;
        f16x8 qf[2], dof[2];
        float dl = 0.f;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            qf[ks] = ld8(qrow + ks * 32 + g * 8);
            dof[ks] = ld8(dorow + ks * 32 + g * 8);
            f16x8 of = ld8(orow + ks * 32 + g * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) dl += (float)dof[ks][e] * (float)of[e];
        }
        dl += __shfl_xor(dl, 16, 64);
        dl += __shfl_xor(dl, 32, 64);
        const int64_t stat = ((int64_t)b * p.heads + h) * L + qc;
        const float lse = p.lse[stat];
        if (g == 0 && q < L) p.delta[stat] = dl;

        const uint8_t* mrow = p.mask + ((int64_t)b * L + qc) * p.Lp;
        constexpr bool PRELOAD = NT <= 12;           // L > 192: the words would push the kernel into spills -- fetch them per tile there
        uint32_t mw[PRELOAD ? NT : 1];
        if (PRELOAD) {
#pragma unroll
            for (int t = 0; t < NT; ++t) mw[t] = mask_word(mrow, t * 16 + 4 * gq, p.Lp);
        }
        const uint32_t rk = p.drop.thresh ? drop_rowkey(p.drop, ((uint64_t)b * p.heads + h) * (uint64_t)L + (uint64_t)qc) : 0u;
        const uint32_t pk0 = drop_pairkey(rk, (uint32_t)(4 * gq));
        const float sc2 = p.scale * LOG2E_F, c0 = -MASK_C1 - lse * LOG2E_F;      // P = exp2(s * sc2 + mask term - lse * log2(e))
        uint32_t dsw[NT / 2][4];                 // dS^T fragments as packed words (see pack_f16x2)
        // dead (query tile, key tile) blocks: P = 0 exactly (see AttnParams::skip); a query row without any attended key has
        // lse ~ -10000 (every score carries the -10000 mask term) -- such rows need every key tile
        const bool rows_live = PRELOAD && p.skip && __all(lse > -5000.f);
        uint32_t live = 0u;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (rows_live && !__any(ANY_ATTEND(mw[PRELOAD ? t : 0]))) {
                dsw[t >> 1][2 * (t & 1)] = 0u;
                dsw[t >> 1][2 * (t & 1) + 1] = 0u;
                continue;
            }
            live |= 1u << t;
            f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f}, dp = (f32x4){0.f, 0.f, 0.f, 0.f};
            const int kr = t * 16 + li;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int off = kr * HD + (((ks * 4 + g) ^ swzk(kr)) << 3);
                s = __builtin_amdgcn_mfma_f32_16x16x32_f16(ld8(Ks + off), qf[ks], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_16x16x32_f16(ld8(Vs + off), dof[ks], dp, 0, 0, 0);
            }
            float ma[4];
            mask4w(PRELOAD ? mw[PRELOAD ? t : 0] : mask_word(mrow, t * 16 + 4 * gq, p.Lp), t * 16 + 16 <= L, c0, ma);
            float m4[4] = {1.f, 1.f, 1.f, 1.f};
            if (p.drop.thresh) {
                const uint32_t h0 = mix32(pk0 + (uint32_t)(8 * t) * VLP_PHI), h1 = mix32(pk0 + (uint32_t)(8 * t + 1) * VLP_PHI);
                m4[0] = drop_mult_h(p.drop, h0, 0u); m4[1] = drop_mult_h(p.drop, h0, 1u);
                m4[2] = drop_mult_h(p.drop, h1, 0u); m4[3] = drop_mult_h(p.drop, h1, 1u);
            }
            float ds4[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pr = __builtin_amdgcn_exp2f(fmaf(s[r], sc2, ma[r]));     // 0 for keys >= L (-inf)
                const float dpr = dp[r] * m4[r];
                ds4[r] = pr * (dpr - dl) * p.scale;
            }
            dsw[t >> 1][2 * (t & 1)] = pack_f16x2(ds4[0], ds4[1]);
            dsw[t >> 1][2 * (t & 1) + 1] = pack_f16x2(ds4[2], ds4[3]);
        }
        // dQ^T tiles: rows = head-dim, col = query;  dQ^T = K^T . dS^T
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            f32x4 o = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < NT / 2; ++u) {
                if (!((live >> (2 * u)) & 3u)) continue;
                o = __builtin_amdgcn_mfma_f32_16x16x32_f16(tr_frag(Ks, 32 * u, 32 * u + 16, 16 * n, g, li),
                                                           words_f16x8(dsw[u][0], dsw[u][1], dsw[u][2], dsw[u][3]), o, 0, 0, 0);
            }
            if (q < L) {
                f16x4 ov = (f16x4){(f16)o[0], (f16)o[1], (f16)o[2], (f16)o[3]};
                st4_out<VLP_SS_ATTN>(p.dqkv + ((int64_t)b * L + q) * p.ld_dqkv + h * HD + n * 16 + 4 * g, ov);
            }
        }
    }
}

// =================================================================================================
// backward, part 2: dK, dV.  Each wave owns a 16-key tile (column per lane) and walks all queries.
// Q and dO of the head are LDS-resident (row-major, swizzled): they serve both as MFMA A operands of the score / dP
// recomputation (row fragments, ds_read_b128) and, through transpose reads, as the Q^T / dO^T operands of dK^T / dV^T.
// The byte mask is read from its key-major copy so that a lane fetches 4 queries of its key with one dword load; all
// loads of a key tile are issued before its query loop.
// =================================================================================================
template <int NT, int NW>     // NW: as in the forward (4 waves at NT = 12: three workgroups per CU, 11 key tiles over 4 waves)
__global__ __launch_bounds__(NW * 64, NW == 4 ? 3 : (NT >= 8 ? 4 : 2)) void attn_bwd_dkv_kernel(AttnParams p) {   // NT = 4: the fully unrolled pair loop needs 130 VGPRs
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int LP = NT * 16;
    f16* Qs = reinterpret_cast<f16*>(smem_raw);     // [LP][64] swizzled
    f16* dOs = Qs + LP * HD;                        // [LP][64] swizzled
    float* lse_s = reinterpret_cast<float*>(dOs + LP * HD);   // [LP]
    float* dl_s = lse_s + LP;                                  // [LP]
    uint32_t* rk_s = reinterpret_cast<uint32_t*>(dl_s + LP);   // [LP] dropout row keys of the queries
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, li = lane & 15;
    const int b = blockIdx.x / p.heads, h = blockIdx.x % p.heads;
    const int L = p.L;
    const f16* qbase = p.qkv + (int64_t)b * L * p.ld_qkv + h * HD;
    const f16* kbase = qbase + p.H;
    const f16* vbase = qbase + 2 * p.H;
    const f16* dobase = p.dctx + (int64_t)b * L * p.ld_dctx + h * HD;

    stage_two_rowmajor<LP, NW * 64>(Qs, qbase, p.ld_qkv, dOs, dobase, p.ld_dctx, L, tid);
    for (int i = tid; i < LP; i += NW * 64) {
        const int64_t stat = ((int64_t)b * p.heads + h) * L + min(i, L - 1);
        lse_s[i] = p.lse[stat] * LOG2E_F;
        dl_s[i] = p.delta[stat];
        rk_s[i] = p.drop.thresh ? drop_rowkey(p.drop, (uint64_t)stat) : 0u;      // dropout element = (row (b, h, q), col key)
    }
    __syncthreads();

    const int nkt = (L + 15) / 16;
    for (int kt = wid; kt < nkt; kt += NW) {
        const int key = kt * 16 + li;            // this lane's key (column)
        const int kc = min(key, L - 1);
        f16x8 kf[2], vf[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            kf[ks] = ld8(kbase + (int64_t)kc * p.ld_qkv + ks * 32 + g * 8);
            vf[ks] = ld8(vbase + (int64_t)kc * p.ld_qkv + ks * 32 + g * 8);
        }
        // mask bytes of this key for queries 16*qt + 4g .. +3 (rows >= L / keys >= L hold 2 = excluded); the words of the
        // next query-tile pair are fetched one iteration ahead
        const uint8_t* mtr = p.mask_t + ((int64_t)b * p.Lp + min(key, p.Lp - 1)) * p.Lp + g * (p.Lp >> 2);      // lane order, see maskt_pos
        auto mload = [&](int qt) -> uint32_t {       // branch-free: clamp the address, select afterwards
            const uint32_t w = *reinterpret_cast<const uint32_t*>(mtr + min(4 * qt, (p.Lp >> 2) - 4));
            return (qt * 16 < p.Lp) ? w : 0x02020202u;
        };
        uint32_t mcur[2] = {mload(0), mload(1)};
        const uint32_t keyphi = ((uint32_t)key >> 1) * VLP_PHI, kodd = (uint32_t)key & 1u;
        const float sc2 = p.scale * LOG2E_F;
        f32x4 dk[4], dv[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) { dk[n] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[n] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

#pragma unroll 1
        for (int u = 0; u < NT / 2; ++u) {       // query tile pair (2u, 2u+1)
            int gq = g;                          // opaque copy: keeps address math inside the loop (no LICM + spills)
            asm volatile("" : "+v"(gq));
            uint32_t mnext[2] = {0x02020202u, 0x02020202u};
            if (u + 1 < NT / 2) { mnext[0] = mload(2 * u + 2); mnext[1] = mload(2 * u + 3); }
            uint32_t pdw[4], dsw[4];             // B operands (rows = queries (pair slots), col = key) as packed words
            bool pair_live = false;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int qt = 2 * u + half;
                // dead block: no (query, key) pair of it is attended AND all its queries have an attended key somewhere (lse_s holds
                // lse * log2 e; rows whose every key is masked sit at ~ -14 400) -> P = 0 exactly for the whole block
                if (p.skip) {
                    const f32x4 l4 = *reinterpret_cast<const f32x4*>(lse_s + qt * 16 + 4 * gq);
                    const bool need = ANY_ATTEND(mcur[half]) || fminf(fminf(l4[0], l4[1]), fminf(l4[2], l4[3])) < -7000.f;
                    if (!__any(need)) {
                        pdw[2 * half] = pdw[2 * half + 1] = 0u;
                        dsw[2 * half] = dsw[2 * half + 1] = 0u;
                        continue;
                    }
                }
                pair_live = true;
                // S tile: rows = queries 16qt + 4g + reg, col = key.  A = Q rows (LDS), B = K rows (registers)
                const int qa = qt * 16 + li;
                f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f}, dp = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const int off = qa * HD + (((ks * 4 + gq) ^ swzk(qa)) << 3);
                    s = __builtin_amdgcn_mfma_f32_16x16x32_f16(ld8(Qs + off), kf[ks], s, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_16x16x32_f16(ld8(dOs + off), vf[ks], dp, 0, 0, 0);
                }
                const int q0 = qt * 16 + 4 * gq;
                const f32x4 lse4 = *reinterpret_cast<const f32x4*>(lse_s + q0);
                const f32x4 dl4 = *reinterpret_cast<const f32x4*>(dl_s + q0);
                const u32x4 rk4 = *reinterpret_cast<const u32x4*>(rk_s + q0);
                float ma[4];
                mask4w(mcur[half], qt * 16 + 16 <= L && kt * 16 + 16 <= L, -MASK_C1, ma);
                float pd4[4], ds4[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pr = __builtin_amdgcn_exp2f(fmaf(s[r], sc2, ma[r] - lse4[r]));     // lse_s holds lse * log2(e); excluded (padding) -> exp2(-inf) = 0
                    float mult = 1.f;
                    if (p.drop.thresh) mult = drop_mult_h(p.drop, mix32(rk4[r] + keyphi), kodd);
                    pd4[r] = pr * mult;
                    ds4[r] = pr * (dp[r] * mult - dl4[r]) * p.scale;
                }
                pdw[2 * half] = pack_f16x2(pd4[0], pd4[1]);
                pdw[2 * half + 1] = pack_f16x2(pd4[2], pd4[3]);
                dsw[2 * half] = pack_f16x2(ds4[0], ds4[1]);
                dsw[2 * half + 1] = pack_f16x2(ds4[2], ds4[3]);
            }
            // dV^T += dO^T . Pd ; dK^T += Q^T . dS   (rows = head-dim, col = key); transposed operands by ds_read_b64_tr_b16
            if (pair_live)
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const f16x8 pdf = words_f16x8(pdw[0], pdw[1], pdw[2], pdw[3]), dsf = words_f16x8(dsw[0], dsw[1], dsw[2], dsw[3]);
                dv[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(tr_frag(dOs, 32 * u, 32 * u + 16, 16 * n, gq, li), pdf, dv[n], 0, 0, 0);
                dk[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(tr_frag(Qs, 32 * u, 32 * u + 16, 16 * n, gq, li), dsf, dk[n], 0, 0, 0);
            }
            mcur[0] = mnext[0];
            mcur[1] = mnext[1];
        }
        if (key < L) {
            f16* drow = p.dqkv + ((int64_t)b * L + key) * p.ld_dqkv + h * HD;
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                f16x4 kv = (f16x4){(f16)dk[n][0], (f16)dk[n][1], (f16)dk[n][2], (f16)dk[n][3]};
                f16x4 vv = (f16x4){(f16)dv[n][0], (f16)dv[n][1], (f16)dv[n][2], (f16)dv[n][3]};
                st4_out<VLP_SS_ATTN>(drow + p.H + n * 16 + 4 * g, kv);
                st4_out<VLP_SS_ATTN>(drow + 2 * p.H + n * 16 + 4 * g, vv);
            }
        }
    }
}

#endif      // VLP_LAB_BUILD
// =================================================================================================
// backward, ONE kernel (round 4): dQ, dK, dV and delta in a single pass over the head.
// The two kernels above each recompute S, P (one v_exp per element), the dropout hash and dP, in opposite orientations: the VALU work
// that bounds both of them is done twice and Q|K|V + dO are read twice from HBM (233 MB per layer where one pass needs ~135 MB).
// Here the score recomputation runs ONCE, in the key-owner orientation of the dK/dV kernel: wave w owns key tiles 2w, 2w+1 (column per
// lane), walks the query-tile pairs u and keeps dK, dV of its keys in registers.  dQ needs the contraction over KEYS, i.e. across the
// waves: every wave drops its dS^T block (fp16, 16 keys x 32 queries) into an LDS exchange tile [LP keys][32 queries]; after ONE
// workgroup barrier per query pair the dQ^T tiles of that pair (2 query tiles x 4 head-dim tiles, each K^T . dS^T over all keys: NT/2
// MFMAs) are computed by the LAST waves of the workgroup -- under the seq2seq mask those own the caption keys, whose blocks are mostly
// dead, so the dQ duty fills their idle time -- from K^T fragments they hold in registers (K passes through the exchange tile once, in
// the prologue).  The exchange tile is double buffered: step u + 1 writes the other half while the duty waves still read step u, so the
// single barrier per step also frees the half that step u + 2 will write.  dQ is summed in a fixed key order by one wave per tile:
// deterministic, no atomics.  delta = rowsum(dO * O) is computed while Q / dO are staged (8 lanes per row).
// LDS at L = 167: Q 24 KB + dO 24 KB + exchange 24 KB + row statistics 2.3 KB = 74.3 KB -> two workgroups of 6 waves per CU.
// =================================================================================================
DEVFN int swz_ds(int r) { return (((r >> 2) & 1) << 2) | (((r >> 1) & 1) << 1) | ((r >> 3) & 1); }
// exchange tile: [keys][32 queries] fp16, 64-byte rows = 8 pieces of 4 queries; piece pc of row r sits at piece pc ^ swz_ds(r):
// the 8-byte writes of a 16-key tile and the transpose reads of the duty waves (8 rows x one aligned group of 4 pieces) are conflict free
DEVFN f16x8 tr_frag_ds(const f16* buf, int r_first, int r_second, int col0, int g, int li) {
    const int ra = r_first + 4 * g + (li >> 2), rb = r_second + 4 * g + (li >> 2);
    const int pc = (col0 >> 2) + (li & 3);
    const u32x2 a = __builtin_bit_cast(u32x2, lds_tr_read(buf + ra * 32 + ((pc ^ swz_ds(ra)) << 2)));
    const u32x2 b = __builtin_bit_cast(u32x2, lds_tr_read(buf + rb * 32 + ((pc ^ swz_ds(rb)) << 2)));
    return words_f16x8(a[0], a[1], b[0], b[1]);
}
// KPW = key tiles per wave.  1 (NT <= 12): NT waves, ONE workgroup per CU (12 waves at L = 167 = 3 per SIMD, <= 168 VGPRs, no spills):
// with two key tiles per wave the dK / dV accumulators (64 registers) + both tiles' K / V rows + the duty's K^T fragments need ~200
// registers, and the 6-wave form spilled the K / V row fragments (scratch reloads inside the loop: 128 us per layer against 83 us for
// the two-kernel form at B = 64).  2 at NT = 16 (L > 192: 8 waves, 256-register budget).
template <int NT, int KPW>
__global__ __launch_bounds__((NT / KPW) * 64, KPW == 2 ? 2 : 3) void attn_bwd_one_kernel(AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int LP = NT * 16, NP = NT / 2, NW = NT / KPW, NTHR = NW * 64;
    constexpr int DW = NW < 4 ? NW : 4;             // duty waves (the last DW of the workgroup), 8 / DW dQ tiles each per step
    constexpr int DT = 8 / DW;
    f16* Qs = reinterpret_cast<f16*>(smem_raw);     // [LP][64] swizzled
    f16* dOs = Qs + LP * HD;                        // [LP][64] swizzled
    f16* xch = dOs + LP * HD;                       // exchange: 2 x [LP][32]  (= one [LP][64] tile: K passes through it in the prologue)
    float* lse_s = reinterpret_cast<float*>(xch + LP * HD);    // [LP] lse * log2(e)
    float* dl_s = lse_s + LP;                                  // [LP] delta
    uint32_t* rk_s = reinterpret_cast<uint32_t*>(dl_s + LP);   // [LP] dropout row keys of the queries
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, li = lane & 15;
    const int b = blockIdx.x / p.heads, h = blockIdx.x % p.heads;
    const int L = p.L;
    const f16* qbase = p.qkv + (int64_t)b * L * p.ld_qkv + h * HD;
    const f16* kbase = qbase + p.H;
    const f16* vbase = qbase + 2 * p.H;
    const f16* dobase = p.dctx + (int64_t)b * L * p.ld_dctx + h * HD;
    const f16* obase = p.ctx + (int64_t)b * L * p.ld_ctx + h * HD;

    // ---- prologue: Q, dO -> LDS (all loads in flight first), delta from the dO / O pieces on the way, K -> exchange tile ---------------
    {
        constexpr int IT = (LP * 8 + NTHR - 1) / NTHR;
        const __amdgpu_buffer_rsrc_t rq = rows_rsrc(qbase, p.ld_qkv, L), rdo = rows_rsrc(dobase, p.ld_dctx, L), ro = rows_rsrc(obase, p.ld_ctx, L),
                                     rkk = rows_rsrc(kbase, p.ld_qkv, L);
        u32x4 vq[IT], vd[IT], vo[IT], vk[IT];
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            const int idx = tid + i * NTHR, r = idx >> 3, c = idx & 7;
            vq[i] = __builtin_amdgcn_raw_buffer_load_b128(rq, (r * (int)p.ld_qkv + c * 8) * 2, 0, 0);
            vd[i] = __builtin_amdgcn_raw_buffer_load_b128(rdo, (r * (int)p.ld_dctx + c * 8) * 2, 0, 0);
            vo[i] = __builtin_amdgcn_raw_buffer_load_b128(ro, (r * (int)p.ld_ctx + c * 8) * 2, 0, 0);
            vk[i] = __builtin_amdgcn_raw_buffer_load_b128(rkk, (r * (int)p.ld_qkv + c * 8) * 2, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            const int idx = tid + i * NTHR, r = idx >> 3, c = idx & 7;
            const f16x8 d8 = __builtin_bit_cast(f16x8, vd[i]), o8 = __builtin_bit_cast(f16x8, vo[i]);
            float dl = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) dl = fmaf((float)d8[e], (float)o8[e], dl);
            dl += __shfl_xor(dl, 1, 64);                  // the 8 lanes of a row are consecutive (c = idx & 7)
            dl += __shfl_xor(dl, 2, 64);
            dl += __shfl_xor(dl, 4, 64);
            if (idx < LP * 8) {
                const int off = r * HD + ((c ^ swzk(r)) << 3);
                *reinterpret_cast<u32x4*>(Qs + off) = vq[i];
                *reinterpret_cast<u32x4*>(dOs + off) = vd[i];
                *reinterpret_cast<u32x4*>(xch + off) = vk[i];
                if (c == 0) dl_s[r] = dl;
            }
        }
    }
    for (int i = tid; i < LP; i += NTHR) {
        const int64_t stat = ((int64_t)b * p.heads + h) * L + min(i, L - 1);
        lse_s[i] = p.lse[stat] * LOG2E_F;
        rk_s[i] = p.drop.thresh ? drop_rowkey(p.drop, (uint64_t)stat) : 0u;      // dropout element = (row (b, h, q), col key)
    }
    // this wave's keys (B operands of the score / dP recomputation), straight from HBM / L2: key = (KPW wid + j) * 16 + li
    f16x8 kf[KPW][2], vf[KPW][2];
#pragma unroll
    for (int j = 0; j < KPW; ++j) {
        const int kc = min((KPW * wid + j) * 16 + li, L - 1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            kf[j][ks] = ld8(kbase + (int64_t)kc * p.ld_qkv + ks * 32 + g * 8);
            vf[j][ks] = ld8(vbase + (int64_t)kc * p.ld_qkv + ks * 32 + g * 8);
        }
    }
    __syncthreads();
    // duty waves: K^T fragments of their head-dim tiles for every key pair, kept in registers for the whole kernel
    const int dwi = wid - (NW - DW);                // >= 0 on duty waves
    f16x8 ktf[DT > 2 ? 2 : 1][NP];                  // d-tiles handled by this wave: n = dwi (+ DW if DW == 2)
    constexpr int NDT = DW == 4 ? 1 : 2;            // distinct d-tiles per duty wave (DW = 4: one d-tile, both query tiles; DW = 2: two d-tiles)
    if (dwi >= 0) {
#pragma unroll
        for (int dn = 0; dn < NDT; ++dn)
#pragma unroll
            for (int kp = 0; kp < NP; ++kp) ktf[dn][kp] = tr_frag(xch, 32 * kp, 32 * kp + 16, 16 * (dwi + dn * DW), g, li);
    }
    __syncthreads();                                // K has been read out of the exchange tile: it now carries dS^T blocks

    const int nkt = (L + 15) / 16;
    const uint8_t* mtr[KPW];
    uint32_t keyphi[KPW], kodd[KPW];
#pragma unroll
    for (int j = 0; j < KPW; ++j) {
        const int key = (KPW * wid + j) * 16 + li;
        mtr[j] = p.mask_t + ((int64_t)b * p.Lp + min(key, p.Lp - 1)) * p.Lp + g * (p.Lp >> 2);      // lane order, see maskt_pos
        keyphi[j] = ((uint32_t)key >> 1) * VLP_PHI;
        kodd[j] = (uint32_t)key & 1u;
    }
    auto mload = [&](int j, int qt) -> uint32_t {       // 4 mask bytes of this lane's key j for queries 16 qt + 4g .. +3; branch-free
        const uint32_t w = *reinterpret_cast<const uint32_t*>(mtr[j] + min(4 * qt, (p.Lp >> 2) - 4));
        return ((KPW * wid + j) * 16 < p.Lp && qt * 16 < p.Lp) ? w : 0x02020202u;
    };
    uint32_t mcur[KPW][2];
#pragma unroll
    for (int j = 0; j < KPW; ++j) { mcur[j][0] = mload(j, 0); mcur[j][1] = mload(j, 1); }
    const float sc2 = p.scale * LOG2E_F;
    f32x4 dk[KPW][4], dv[KPW][4];
#pragma unroll
    for (int j = 0; j < KPW; ++j)
#pragma unroll
        for (int n = 0; n < 4; ++n) { dk[j][n] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[j][n] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

#pragma unroll 1
    for (int u = 0; u < NP; ++u) {               // query tile pair (2u, 2u+1)
        int gq = g;                              // opaque copy: keeps address math inside the loop (no LICM + spills)
        asm volatile("" : "+v"(gq));
        f16* xb = xch + (u & 1) * (LP * 32);
        uint32_t mnext[KPW][2];
#pragma unroll
        for (int j = 0; j < KPW; ++j) {
            mnext[j][0] = mnext[j][1] = 0x02020202u;
            if (u + 1 < NP) { mnext[j][0] = mload(j, 2 * u + 2); mnext[j][1] = mload(j, 2 * u + 3); }
        }
#pragma unroll
        for (int j = 0; j < KPW; ++j) {          // this wave's key tile(s)
            const int kt = KPW * wid + j;
            uint32_t pdw[4], dsw[4];             // B operands (rows = queries (pair slots), col = key) as packed words
            bool pair_live = false;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int qt = 2 * u + half;
                bool dead = kt >= nkt;
                if (!dead && p.skip) {
                    // dead block: no (query, key) pair of it is attended AND all its queries have an attended key somewhere -> P = 0 exactly
                    const f32x4 l4 = *reinterpret_cast<const f32x4*>(lse_s + qt * 16 + 4 * gq);
                    const bool need = ANY_ATTEND(mcur[j][half]) || fminf(fminf(l4[0], l4[1]), fminf(l4[2], l4[3])) < -7000.f;
                    dead = !__any(need);
                }
                // the block's row of the exchange tile: key row kt * 16 + li, piece = queries 16 half + 4g .. +3
                const int xr = kt * 16 + li;
                uint32_t* xdst = reinterpret_cast<uint32_t*>(xb + xr * 32 + (((half * 4 + gq) ^ swz_ds(xr)) << 2));
                if (dead) {
                    pdw[2 * half] = pdw[2 * half + 1] = 0u;
                    dsw[2 * half] = dsw[2 * half + 1] = 0u;
                    *reinterpret_cast<u32x2*>(xdst) = (u32x2){0u, 0u};
                    continue;
                }
                pair_live = true;
                // S tile: rows = queries 16qt + 4g + reg, col = key.  A = Q rows (LDS), B = K rows (registers)
                const int qa = qt * 16 + li;
                f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f}, dp = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const int off = qa * HD + (((ks * 4 + gq) ^ swzk(qa)) << 3);
                    s = __builtin_amdgcn_mfma_f32_16x16x32_f16(ld8(Qs + off), kf[j][ks], s, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_16x16x32_f16(ld8(dOs + off), vf[j][ks], dp, 0, 0, 0);
                }
                const int q0 = qt * 16 + 4 * gq;
                const f32x4 lse4 = *reinterpret_cast<const f32x4*>(lse_s + q0);
                const f32x4 dl4 = *reinterpret_cast<const f32x4*>(dl_s + q0);
                const u32x4 rk4 = *reinterpret_cast<const u32x4*>(rk_s + q0);
                float ma[4];
                mask4w(mcur[j][half], qt * 16 + 16 <= L && kt * 16 + 16 <= L, -MASK_C1, ma);
                float pd4[4], ds4[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pr = __builtin_amdgcn_exp2f(fmaf(s[r], sc2, ma[r] - lse4[r]));     // excluded (padding) -> exp2(-inf) = 0
                    float mult = 1.f;
                    if (p.drop.thresh) mult = drop_mult_h(p.drop, mix32(rk4[r] + keyphi[j]), kodd[j]);
                    pd4[r] = pr * mult;
                    ds4[r] = pr * (dp[r] * mult - dl4[r]) * p.scale;
                }
                pdw[2 * half] = pack_f16x2(pd4[0], pd4[1]);
                pdw[2 * half + 1] = pack_f16x2(pd4[2], pd4[3]);
                dsw[2 * half] = pack_f16x2(ds4[0], ds4[1]);
                dsw[2 * half + 1] = pack_f16x2(ds4[2], ds4[3]);
                *reinterpret_cast<u32x2*>(xdst) = (u32x2){dsw[2 * half], dsw[2 * half + 1]};      // dS^T block for the dQ duty
            }
            // dV^T += dO^T . Pd ; dK^T += Q^T . dS   (rows = head-dim, col = key); transposed operands by ds_read_b64_tr_b16
            if (pair_live) {
                const f16x8 pdf = words_f16x8(pdw[0], pdw[1], pdw[2], pdw[3]), dsf = words_f16x8(dsw[0], dsw[1], dsw[2], dsw[3]);
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    dv[j][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(tr_frag(dOs, 32 * u, 32 * u + 16, 16 * n, gq, li), pdf, dv[j][n], 0, 0, 0);
                    dk[j][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(tr_frag(Qs, 32 * u, 32 * u + 16, 16 * n, gq, li), dsf, dk[j][n], 0, 0, 0);
                }
            }
            mcur[j][0] = mnext[j][0];
            mcur[j][1] = mnext[j][1];
        }
        // every wave's dS^T blocks of pair u are in xb; the other half is free for step u + 1.  LDS writes only: a __syncthreads() would
        // also wait (vmcnt(0)) for the duty waves' dQ stores of the previous step and for the mask words just requested
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (dwi >= 0) {                          // dQ duty: tiles (qh, n) of this pair
#pragma unroll
            for (int t = 0; t < DT; ++t) {
                const int qh = DW == 4 ? t : (t & 1);
                const int dn = DW == 4 ? 0 : (t >> 1);
                const int n = dwi + dn * DW;
                f32x4 o = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kp = 0; kp < NP; ++kp)
                    o = __builtin_amdgcn_mfma_f32_16x16x32_f16(ktf[dn][kp], tr_frag_ds(xb, 32 * kp, 32 * kp + 16, 16 * qh, gq, li), o, 0, 0, 0);
                const int q = (2 * u + qh) * 16 + li;
                if (q < L) {
                    const f16x4 ov = (f16x4){(f16)o[0], (f16)o[1], (f16)o[2], (f16)o[3]};
                    st4_out<VLP_SS_ATTN>(p.dqkv + ((int64_t)b * L + q) * p.ld_dqkv + h * HD + n * 16 + 4 * gq, ov);
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < KPW; ++j) {
        const int key = (KPW * wid + j) * 16 + li;
        if (key < L) {
            f16* drow = p.dqkv + ((int64_t)b * L + key) * p.ld_dqkv + h * HD;
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const f16x4 kv = (f16x4){(f16)dk[j][n][0], (f16)dk[j][n][1], (f16)dk[j][n][2], (f16)dk[j][n][3]};
                const f16x4 vv = (f16x4){(f16)dv[j][n][0], (f16)dv[j][n][1], (f16)dv[j][n][2], (f16)dv[j][n][3]};
                st4_out<VLP_SS_ATTN>(drow + p.H + n * 16 + 4 * g, kv);
                st4_out<VLP_SS_ATTN>(drow + 2 * p.H + n * 16 + 4 * g, vv);
            }
        }
    }
}

// =================================================================================================
// backward, one kernel, NO per-step barriers (L <= 192): the whole dS^T of the head lives in LDS.
// Measured with the exchange-tile form above at L = 167, B = 64 (tools/attn_lab.py, round 4): 73 us per layer against 80 us for the two
// kernels -- one workgroup per CU runs 25 us of which the six step barriers expose every wave's two-block latency chain six times.  With one
// workgroup per CU the 160 KB of LDS are free anyway: Q, dO, K (24 KB each) and dS^T [LP keys][LP queries] fp16 (72 KB) = 146 KB.
// Phase 1: wave w owns key tile w (column per lane) and walks ALL query tiles without synchronisation (dK, dV in registers, its dS^T
// rows into LDS); ONE barrier; phase 2: the NT waves share the dQ^T tiles (query tile x head-dim tile), each K^T . dS^T over all keys with
// both operands read through ds_read_b64_tr_b16 -- no K^T fragments parked in registers.  Deterministic (one wave per output tile, fixed
// key order).
// =================================================================================================
DEVFN int swz_dsf(int r) { return (((r >> 1) & 3) << 2) | (((r >> 3) & 1) << 1) | (r & 1); }
// dS^T tile: [LP keys][LP queries] fp16, rows of LP / 4 pieces (4 queries = 8 bytes); piece pc of row r at pc ^ swz_dsf(r) (the XOR stays
// inside an aligned group of 16 pieces): the 8-byte block writes of a key tile and the transpose reads of phase 2 are conflict free
template <int LP>
DEVFN f16x8 tr_frag_dsf(const f16* buf, int r_first, int r_second, int col0, int g, int li) {
    const int ra = r_first + 4 * g + (li >> 2), rb = r_second + 4 * g + (li >> 2);
    const int pc = (col0 >> 2) + (li & 3);
    const u32x2 a = __builtin_bit_cast(u32x2, lds_tr_read(buf + ra * LP + ((pc ^ swz_dsf(ra)) << 2)));
    const u32x2 b = __builtin_bit_cast(u32x2, lds_tr_read(buf + rb * LP + ((pc ^ swz_dsf(rb)) << 2)));
    return words_f16x8(a[0], a[1], b[0], b[1]);
}
// DROP (compile time): dropout on the probabilities.  The phase-1 loop is straight-line code per query pair: a phase trace of the first
// version (tools/attn_bwd_trace.py: 4 250 cycles per pair for ~500 instructions, 31 s_waitcnt) showed it latency-bound on (a) the next
// pair's mask words, fetched inside the loop behind scalar branches and waited for with vmcnt(0) -- all NT words of the lane's key are
// loaded before the loop now --, and (b) a dozen small basic blocks from the run-time dropout / interior-tile switches.
template <int NT, bool DROP>     // NT = LP / 16 key tiles = waves of the workgroup (4, 8, 12)
__global__ __launch_bounds__(NT * 64, NT == 12 ? 3 : 2) void attn_bwd_full_kernel(AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int LP = NT * 16, NP = NT / 2, NTHR = NT * 64;
    f16* Qs = reinterpret_cast<f16*>(smem_raw);     // [LP][64] swizzled
    f16* dOs = Qs + LP * HD;                        // [LP][64] swizzled
    f16* Ks = dOs + LP * HD;                        // [LP][64] swizzled
    f16* dSs = Ks + LP * HD;                        // [LP][LP] piece-swizzled
    float* lse_s = reinterpret_cast<float*>(dSs + LP * LP);    // [LP] lse * log2(e)
    float* dl_s = lse_s + LP;                                  // [LP] delta
    uint32_t* rk_s = reinterpret_cast<uint32_t*>(dl_s + LP);   // [LP] dropout row keys of the queries
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g0 = lane >> 4, li0 = lane & 15;
    const int L = p.L;
    const int nitems = p.B * p.heads;
    const int kt = wid;
    const int key = kt * 16 + li0;
    constexpr int IT = (LP * 8 + NTHR - 1) / NTHR;
    constexpr int ST = (LP + NTHR - 1) / NTHR;           // row statistics per thread (1 unless the workgroup has fewer threads than rows)
    // PERSISTENT over (batch, head) items: item, item + gridDim.x, ... (the launcher starts one workgroup per CU).  With one workgroup per CU
    // nothing else hides an item's load prologue (Q, dO, K, O: 96 KB per item, ~7 us per round when every CU loads at once; 45 % of a
    // workgroup's time in the phase trace of the one-item form).  The NEXT item's Q and dO tiles are requested into registers (vq / vd) right
    // after the current item's have been written to LDS, travel under its two compute phases, and are stored when phase 2 has released the LDS.
    // (Q and dO only: with all four tiles -- 32 more registers per lane across both compute phases -- the NT = 12 instantiation spilled 120
    // registers at the 168-register budget of 3 waves per SIMD; K and O are requested at the head of their item, with the V rows, lse and mask words)
    u32x4 vq[IT], vd[IT];
    auto request_tiles = [&](int it) {
        const int b_ = it / p.heads, h_ = it % p.heads;
        const int rb_ = p.row_off ? p.row_off[b_] : b_ * L, nb_ = p.row_off ? p.row_off[b_ + 1] - rb_ : L;      // packed rows: the sample's kept rows
        const __amdgpu_buffer_rsrc_t rq = rows_rsrc(p.qkv + (int64_t)rb_ * p.ld_qkv + h_ * HD, p.ld_qkv, nb_),
                                     rdo = rows_rsrc(p.dctx + (int64_t)rb_ * p.ld_dctx + h_ * HD, p.ld_dctx, nb_);
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            const int idx = tid + i * NTHR, r = idx >> 3, c = idx & 7;
            vq[i] = __builtin_amdgcn_raw_buffer_load_b128(rq, (r * (int)p.ld_qkv + c * 8) * 2, 0, 0);
            vd[i] = __builtin_amdgcn_raw_buffer_load_b128(rdo, (r * (int)p.ld_dctx + c * 8) * 2, 0, 0);
        }
    };
    if ((int)blockIdx.x < nitems) request_tiles(blockIdx.x);
#pragma unroll 1
    for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
    const int b = item / p.heads, h = item % p.heads;
    int g = g0, li = li0;                    // opaque per-item copies: the LDS address math of the unrolled pair loop is item-invariant, and hoisted
    asm volatile("" : "+v"(g), "+v"(li));    // out of the item loop it would occupy ~60 registers across both phases (spills at the 168-register budget)
    // packed rows: the sample's nb kept positions start at row rb.  Positions >= nb are treated like the padding rows >= L: query rows get
    // lse = +inf, key columns a -inf mask base -- their probabilities were exactly 0 in the dense run too (mask byte 0 / dO = 0), so every
    // sum below receives the same zeros
    const int rb = p.row_off ? p.row_off[b] : b * L, nb = p.row_off ? p.row_off[b + 1] - rb : L;
    const f16* vbase = p.qkv + (int64_t)rb * p.ld_qkv + h * HD + 2 * p.H;

    if (item == (int)blockIdx.x) TRACE(0);
    // ---- prologue: EVERY global load of the item is issued before the first one is consumed (one memory round trip, not four: the
    // first build staged, then fetched lse, then the V rows, then the mask words behind scalar branches -- 18 000 cycles of the 39 000 a
    // workgroup takes).  Q, dO, K -> LDS; delta from the dO / O pieces on the way.
    float lse_v[ST];
    f16x8 kf[2], vf[2];
    uint32_t mw[NT];                                      // mask bytes of this lane's key for ALL queries (word t = queries 16 t + 4g .. +3)
    u32x4 vo[IT], vk[IT];
    {
        const __amdgpu_buffer_rsrc_t ro = rows_rsrc(p.ctx + (int64_t)rb * p.ld_ctx + h * HD, p.ld_ctx, nb),
                                     rkk = rows_rsrc(p.qkv + (int64_t)rb * p.ld_qkv + h * HD + p.H, p.ld_qkv, nb);
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            const int idx = tid + i * NTHR, r = idx >> 3, c = idx & 7;
            vk[i] = __builtin_amdgcn_raw_buffer_load_b128(rkk, (r * (int)p.ld_qkv + c * 8) * 2, 0, 0);
            vo[i] = __builtin_amdgcn_raw_buffer_load_b128(ro, (r * (int)p.ld_ctx + c * 8) * 2, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < ST; ++i) lse_v[i] = p.lse[((int64_t)b * p.heads + h) * L + min(tid + i * NTHR, L - 1)];
        const int kc = min(key, nb - 1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) vf[ks] = ld8(vbase + (int64_t)kc * p.ld_qkv + ks * 32 + g * 8);
        // unconditional loads from a clamped address + a bitwise select: a `cond ? *ptr : pad` lets the compiler sink every load into its
        // own scalar branch
        __builtin_amdgcn_sched_barrier(0);        // the mask words are the youngest loads: the staging writes above must not queue behind them
        const uint8_t* mtr = p.mask_t + ((int64_t)b * p.Lp + min(key, p.Lp - 1)) * p.Lp + g * (p.Lp >> 2);      // lane order (maskt_pos): words t = 0 .. Lp/16 - 1 are consecutive
        if (NT % 4 == 0 && p.Lp == LP && ((uintptr_t)p.mask_t & 15) == 0) {      // (L = 167: three 16-byte loads per lane)
#pragma unroll
            for (int t4 = 0; t4 < NT / 4; ++t4) {
                const u32x4 w4 = *reinterpret_cast<const u32x4*>(mtr + 16 * t4);
#pragma unroll
                for (int e = 0; e < 4; ++e) mw[4 * t4 + e] = w4[e];
            }
        } else {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const uint32_t w = *reinterpret_cast<const uint32_t*>(mtr + min(4 * t, (p.Lp >> 2) - 4));
                const uint32_t keep = (kt * 16 < p.Lp && t * 16 < p.Lp) ? 0xffffffffu : 0u;       // rows >= Lp / keys >= Lp: 2 = excluded
                mw[t] = (w & keep) | (0x02020202u & ~keep);
            }
        }
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            const int idx = tid + i * NTHR, r = idx >> 3, c = idx & 7;
            const f16x8 d8 = __builtin_bit_cast(f16x8, vd[i]), o8 = __builtin_bit_cast(f16x8, vo[i]);
            float dl = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) dl = fmaf((float)d8[e], (float)o8[e], dl);
            dl += __shfl_xor(dl, 1, 64);                  // the 8 lanes of a row are consecutive (c = idx & 7)
            dl += __shfl_xor(dl, 2, 64);
            dl += __shfl_xor(dl, 4, 64);
            if (idx < LP * 8) {
                const int off = r * HD + ((c ^ swzk(r)) << 3);
                *reinterpret_cast<u32x4*>(Qs + off) = vq[i];
                *reinterpret_cast<u32x4*>(dOs + off) = vd[i];
                *reinterpret_cast<u32x4*>(Ks + off) = vk[i];
                if (c == 0) dl_s[r] = dl;
            }
        }
#pragma unroll
        for (int i = 0; i < ST; ++i) {
            const int r = tid + i * NTHR;
            if (r < LP) {
                // padding rows (r >= L): +inf, so that every probability of the row comes out as exp2(-inf) = 0 without a per-element test
                lse_s[r] = r < nb ? lse_v[i] * LOG2E_F : INFINITY;
                rk_s[r] = p.drop.thresh ? drop_rowkey(p.drop, (uint64_t)(((int64_t)b * p.heads + h) * L + min(r, L - 1))) : 0u;   // dropout element = (row (b, h, q), col key)
            }
        }
    }
    if (item == (int)blockIdx.x) TRACE(1);
    // staging barrier on the LDS writes only: the mask words (12 small gathers per lane, the youngest loads in the queue) and the V rows stay
    // in flight across it and are waited for at their first use -- a __syncthreads() drains them too (vmcnt(0): +8 000 cycles per workgroup
    // in the phase trace)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (item == (int)blockIdx.x) TRACE(2);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) kf[ks] = ld8(Ks + key * HD + (((ks * 4 + g) ^ swzk(key)) << 3));
    if (item + (int)gridDim.x < nitems) request_tiles(item + gridDim.x);        // the staging registers are free again: next item's tiles

    // ---- phase 1: this wave's key tile against every query tile ------------------------------------------------------------------------
    const int nkt = (nb + 15) / 16;
    const uint32_t keyphi = ((uint32_t)key >> 1) * VLP_PHI, kodd = (uint32_t)key & 1u;
    const float sc2 = p.scale * LOG2E_F;
    // additive mask term in the log2 domain WITHOUT a per-element padding test: byte b in {0, 1} -> b * C1 + kmbase with kmbase = -C1 for a
    // real key; a padding key (>= L) has kmbase = -inf, a padding query row has lse = +inf (above): their byte is 2, and 2 C1 - inf = -inf /
    // C1 - (+inf) = -inf -- every excluded probability is exp2(-inf) = 0 and multiplies finite numbers only (dO, Q rows >= L are zero)
    const float kmbase = key < nb ? -MASK_C1 : -INFINITY;
    // (Packed fp32 pair arithmetic -- v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 on element pairs -- was measured for this block and is a wash:
    // 54.2 - 54.5 us against 54.0 - 55.4 us per layer; in the FFN-up GEMM epilogue the packed erf / gelu' pair was 17 % SLOWER than the scalar
    // form (89 vs 76 us per launch).  The guide prices v_pk_* above two scalar ops on this chip; kept scalar.)
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x4 dk[4], dv[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) { dk[n] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[n] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    f16* xrow = dSs + key * LP;                  // this lane's dS^T row
    const int xsw = swz_dsf(key);

#pragma unroll
    for (int u = 0; u < NP; ++u) {               // query tile pair (2u, 2u+1); unrolled: mw[] stays in registers, addresses fold into offsets
        uint32_t pdw[4], dsw[4];                 // B operands (rows = queries (pair slots), col = key) as packed words
        // dead blocks (P = 0 exactly: no (query, key) pair attended AND every query has an attended key somewhere): the PAIR is skipped when
        // both of its blocks are dead; a single dead block of a live pair is simply computed -- its probabilities come out as exact zeros
        // (exp2 of -14 000)
        bool pair_live = kt < nkt && 32 * u < nb;        // (a query pair past the sample's kept rows: lse = +inf, every probability 0)
        if (pair_live && p.skip) {
            bool need = false;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const f32x4 l4 = *reinterpret_cast<const f32x4*>(lse_s + (2 * u + half) * 16 + 4 * g);
                need = need || ANY_ATTEND(mw[2 * u + half]) || fminf(fminf(l4[0], l4[1]), fminf(l4[2], l4[3])) < -7000.f;
            }
            pair_live = __any(need);
        }
        uint32_t* xdst[2];
#pragma unroll
        for (int half = 0; half < 2; ++half) xdst[half] = reinterpret_cast<uint32_t*>(xrow + ((((2 * u + half) * 4 + g) ^ xsw) << 2));   // piece = queries 16 qt + 4g .. +3
        if (!pair_live) {
            *reinterpret_cast<u32x2*>(xdst[0]) = (u32x2){0u, 0u};
            *reinterpret_cast<u32x2*>(xdst[1]) = (u32x2){0u, 0u};
            continue;
        }
        // S / dP tiles of both blocks: rows = queries 16qt + 4g + reg, col = key.  A = Q / dO rows (LDS), B = K / V rows (registers)
        f32x4 s[2], dp[2];
        f32x4 lse4[2], dl4[2];
        u32x2 rk2[2];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int qa = (2 * u + half) * 16 + li;
            s[half] = (f32x4){0.f, 0.f, 0.f, 0.f};
            dp[half] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int off = qa * HD + (((ks * 4 + g) ^ swzk(qa)) << 3);
                s[half] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ld8(Qs + off), kf[ks], s[half], 0, 0, 0);
                dp[half] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ld8(dOs + off), vf[ks], dp[half], 0, 0, 0);
            }
        }
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int q0 = (2 * u + half) * 16 + 4 * g;
            lse4[half] = *reinterpret_cast<const f32x4*>(lse_s + q0);
            dl4[half] = *reinterpret_cast<const f32x4*>(dl_s + q0);
            // dropout: the hash of (query row, key PAIR) serves both keys of the pair (low / high 16 bits), and the two keys of a pair sit in
            // ADJACENT lanes here: the even-key lane hashes rows q0, q0 + 1, the odd-key lane rows q0 + 2, q0 + 3, and they swap (DPP quad_perm)
            // the 16-bit halves the other one needs -- 2 hashes per lane and block instead of 4 identical pairs of them
            if (DROP) rk2[half] = *reinterpret_cast<const u32x2*>(rk_s + q0 + 2 * (int)kodd);
            const uint32_t w = mw[2 * u + half];
            float pd4[4], ds4[4];
            f32x2 multp[2] = {(f32x2){1.f, 1.f}, (f32x2){1.f, 1.f}};      // dropout multipliers of the element pairs (r = 0, 1) and (r = 2, 3)
            if (DROP) {
                const uint32_t h0 = mix32(rk2[half][0] + keyphi), h1 = mix32(rk2[half][1] + keyphi);
                const uint32_t sh_mine = 16u * kodd, sh_other = 16u - sh_mine;         // my key's half of a hash / my neighbour's
                const uint32_t m0 = __builtin_amdgcn_ubfe(h0, sh_mine, 16u), m1 = __builtin_amdgcn_ubfe(h1, sh_mine, 16u);
                const uint32_t o0 = (uint32_t)__builtin_amdgcn_mov_dpp((int)__builtin_amdgcn_ubfe(h0, sh_other, 16u), 0xB1, 0xF, 0xF, true);      // quad_perm [1, 0, 3, 2]
                const uint32_t o1 = (uint32_t)__builtin_amdgcn_mov_dpp((int)__builtin_amdgcn_ubfe(h1, sh_other, 16u), 0xB1, 0xF, 0xF, true);
                const f32x2 mine = (f32x2){m0 < p.drop.thresh ? 0.f : p.drop.scale, m1 < p.drop.thresh ? 0.f : p.drop.scale};
                const f32x2 other = (f32x2){o0 < p.drop.thresh ? 0.f : p.drop.scale, o1 < p.drop.thresh ? 0.f : p.drop.scale};
                multp[0] = kodd ? other : mine;           // rows q0, q0 + 1 were hashed by the even-key lane
                multp[1] = kodd ? mine : other;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float ma = fmaf((float)((w >> (8 * r)) & 0xffu), MASK_C1, kmbase);          // v_cvt_f32_ubyteN + one FMA
                const float pr = __builtin_amdgcn_exp2f(fmaf(s[half][r], sc2, ma - lse4[half][r]));
                const float mult = multp[r >> 1][r & 1];
                pd4[r] = pr * mult;
                ds4[r] = pr * (dp[half][r] * mult - dl4[half][r]) * p.scale;
            }
            pdw[2 * half] = pack_f16x2(pd4[0], pd4[1]);
            pdw[2 * half + 1] = pack_f16x2(pd4[2], pd4[3]);
            dsw[2 * half] = pack_f16x2(ds4[0], ds4[1]);
            dsw[2 * half + 1] = pack_f16x2(ds4[2], ds4[3]);
            *reinterpret_cast<u32x2*>(xdst[half]) = (u32x2){dsw[2 * half], dsw[2 * half + 1]};      // dS^T block for phase 2
        }
        // dV^T += dO^T . Pd ; dK^T += Q^T . dS   (rows = head-dim, col = key); transposed operands by ds_read_b64_tr_b16
        {
            const f16x8 pdf = words_f16x8(pdw[0], pdw[1], pdw[2], pdw[3]), dsf = words_f16x8(dsw[0], dsw[1], dsw[2], dsw[3]);
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                dv[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(tr_frag(dOs, 32 * u, 32 * u + 16, 16 * n, g, li), pdf, dv[n], 0, 0, 0);
                dk[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(tr_frag(Qs, 32 * u, 32 * u + 16, 16 * n, g, li), dsf, dk[n], 0, 0, 0);
            }
        }
        if (u == 0 && item == (int)blockIdx.x) TRACE(3);
    }
    if (item == (int)blockIdx.x) TRACE(4);
#ifdef VLP_ATTN_TRACE
    if (item == (int)blockIdx.x && wid == (int)(blockIdx.x % NT) && lane == 0 && blockIdx.x < 4096) g_attn_trace[blockIdx.x * 8 + 7] = __builtin_readcyclecounter();     // end of phase 1 of wave blockIdx % NT
#endif
    if (key < nb) {
        f16* drow = p.dqkv + ((int64_t)rb + key) * p.ld_dqkv + h * HD;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const f16x4 kv = (f16x4){(f16)dk[n][0], (f16)dk[n][1], (f16)dk[n][2], (f16)dk[n][3]};
            const f16x4 vv = (f16x4){(f16)dv[n][0], (f16)dv[n][1], (f16)dv[n][2], (f16)dv[n][3]};
            st4_out<VLP_SS_ATTN>(drow + p.H + n * 16 + 4 * g, kv);
            st4_out<VLP_SS_ATTN>(drow + 2 * p.H + n * 16 + 4 * g, vv);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (LDS writes only: the dK / dV stores above stay in flight across the barrier)
    __builtin_amdgcn_s_barrier();
    if (item == (int)blockIdx.x) TRACE(5);

    // ---- phase 2: dQ^T tiles (rows = head-dim 16 n + 4g + reg, col = query), K^T . dS^T over all keys ---------------------------------
    // (two tiles of a wave in flight at once: the NT/2 MFMAs of a tile form one dependent chain)
    const int nqt = (nb + 15) / 16;
    for (int t = wid; t < nqt * 4; t += 2 * NT) {
        const int t1 = t + NT;
        const bool two = t1 < nqt * 4;
        const int qt0 = t >> 2, n0 = t & 3, qt1 = two ? (t1 >> 2) : qt0, n1 = two ? (t1 & 3) : n0;
        f32x4 o0 = (f32x4){0.f, 0.f, 0.f, 0.f}, o1 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kp = 0; kp < NP; ++kp) {
            o0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(tr_frag(Ks, 32 * kp, 32 * kp + 16, 16 * n0, g, li),
                                                        tr_frag_dsf<LP>(dSs, 32 * kp, 32 * kp + 16, 16 * qt0, g, li), o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(tr_frag(Ks, 32 * kp, 32 * kp + 16, 16 * n1, g, li),
                                                        tr_frag_dsf<LP>(dSs, 32 * kp, 32 * kp + 16, 16 * qt1, g, li), o1, 0, 0, 0);
        }
        const int q0 = qt0 * 16 + li, q1 = qt1 * 16 + li;
        if (q0 < nb) {
            const f16x4 ov = (f16x4){(f16)o0[0], (f16)o0[1], (f16)o0[2], (f16)o0[3]};
            st4_out<VLP_SS_ATTN>(p.dqkv + ((int64_t)rb + q0) * p.ld_dqkv + h * HD + n0 * 16 + 4 * g, ov);
        }
        if (two && q1 < nb) {
            const f16x4 ov = (f16x4){(f16)o1[0], (f16)o1[1], (f16)o1[2], (f16)o1[3]};
            st4_out<VLP_SS_ATTN>(p.dqkv + ((int64_t)rb + q1) * p.ld_dqkv + h * HD + n1 * 16 + 4 * g, ov);
        }
    }
    if (item == (int)blockIdx.x) TRACE(6);
    // the next item's staging writes may only start when every wave has finished reading this item's tiles (phase 2: K and dS^T)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    }      // item loop
}

// ---------------------------------------------------------------------------------------------
static int attn_common_check(const char* who, const void* qkv, int64_t ld_qkv, const uint8_t* mask, int B, int L, int heads) {
    VLP_CHECK_ARG(qkv && mask, "%s: null operand", who);
    VLP_CHECK_ARG(B > 0 && L > 0 && heads > 0, "%s: bad shape", who);
    VLP_CHECK_ARG(L <= 256, "%s: L=%d > 256 is not supported by the single-pass kernel", who, L);
    VLP_CHECK_ARG(ld_qkv % 8 == 0 && ld_qkv >= 3 * heads * HD, "%s: ld_qkv", who);
    VLP_CHECK_ARG((uintptr_t)qkv % 16 == 0 && (uintptr_t)mask % 4 == 0, "%s: alignment", who);
    return VLP_OK;
}

static inline int lp_of(int L) { return L <= 64 ? 64 : (L <= 128 ? 128 : (L <= 192 ? 192 : 256)); }

// waves per workgroup of the forward / dK-dV kernels at NT = 12 (VLP_ATTN_WAVES=8 restores the round-1 geometry for A/B runs)
static int attn_waves_nt12() {
    const char* e = getenv("VLP_ATTN_WAVES");
    return (e && atoi(e) == 8) ? 8 : 4;
}

// VLP_ATTN_SKIP=0 disables the dead-block skipping (bit-identical either way; read at every launch so tests can toggle it)
static int attn_skip_enabled() {
    const char* e = getenv("VLP_ATTN_SKIP");
    return (e && e[0] == '0') ? 0 : 1;
}

static int launch_attn_fwd(AttnParams& p, hipStream_t s) {
    p.skip = attn_skip_enabled();
    const int LP = lp_of(p.Lk);
#ifdef VLP_LAB_BUILD
    // investigation builds: VLP_ATTN_FWD_STREAM=1 runs the training shape (129 <= L <= 192, queries = keys from one packed buffer) on the
    // persistent streaming kernel (identical bits; no gain in the step, see the note above it)
    if (LP == 192 && p.n_prefix == 0 && p.Lq == p.Lk && p.lse != nullptr) {
        const char* e = getenv("VLP_ATTN_FWD_STREAM");
        if (e && e[0] == '1') {
            static const int ncu = [] { int dev = 0, n = 256; (void)hipGetDevice(&dev); if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 256; return n; }();
            const int items = p.B * p.heads;
            const size_t smem = (size_t)AFS_NBUF * 2 * LP * HD * 2 + 64;
            VLP_ONCE_PER_DEVICE((void)hipFuncSetAttribute((const void*)attn_fwd_stream_kernel<12>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
            hipLaunchKernelGGL((attn_fwd_stream_kernel<12>), dim3(items < ncu ? items : ncu), dim3(AFS_WAVES * 64), smem, s, p);
            VLP_CHECK_LAUNCH("vlp_attn_fwd");
            return VLP_OK;
        }
    }
#endif
    const size_t smem = (size_t)2 * LP * HD * 2;
    dim3 grid(p.B * p.heads);
    static const int nw12 = attn_waves_nt12();
#define LAUNCH_FWD(NT_, NW_)                                                                                         \
    do {                                                                                                             \
        VLP_ONCE_PER_DEVICE((void)hipFuncSetAttribute((const void*)attn_fwd_kernel<NT_, NW_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));                                                            \
        hipLaunchKernelGGL((attn_fwd_kernel<NT_, NW_>), grid, dim3((NW_) * 64), smem, s, p);                           \
    } while (0)
    if (LP == 64) LAUNCH_FWD(4, 8); else if (LP == 128) LAUNCH_FWD(8, 8);
    else if (LP == 192) { if (nw12 == 4) LAUNCH_FWD(12, 4); else LAUNCH_FWD(12, 8); }
    else LAUNCH_FWD(16, 8);
#undef LAUNCH_FWD
    VLP_CHECK_LAUNCH("vlp_attn_fwd");
    return VLP_OK;
}

extern "C" int vlp_attn_fwd(const vlp_attn_fwd_args* a, void* stream) {
    VLP_CHECK_ARG(a != nullptr, "vlp_attn_fwd: null args");
    VLP_ENTER(a->qkv, "vlp_attn_fwd");
    int rc = attn_common_check("vlp_attn_fwd", a->qkv, a->ld_qkv, a->mask, a->B, a->L, a->heads);
    if (rc) return rc;
    VLP_CHECK_ARG(a->ctx && a->lse && a->ld_ctx % 4 == 0 && (uintptr_t)a->ctx % 8 == 0, "vlp_attn_fwd: ctx/lse");
    AttnParams p = {};
    p.mask = a->mask;
    p.ctx = (f16*)a->ctx; p.ld_ctx = a->ld_ctx; p.lse = a->lse;
    p.B = a->B; p.L = a->L; p.heads = a->heads; p.H = a->heads * HD;
    p.Lp = (a->L + 31) / 32 * 32;
    p.scale = a->scale;
    p.drop = make_drop(a->dropout_p, a->seed, a->rng_stream);
    p.q = (const f16*)a->qkv; p.k = p.q + p.H; p.v = p.q + 2 * p.H;
    p.ld_q = p.ld_kv = a->ld_qkv; p.bs_q = p.bs_kv = a->L; p.Lq = p.Lk = a->L;
    p.row_off = a->row_off;
    return launch_attn_fwd(p, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------
// Token-step attention of the incremental decoder (round 6): Lq <= 2 new queries per sequence against Lk <= 128 cached keys.  The general
// forward kernel above stages K and V of the head in LDS behind two workgroup barriers and walks 16-query tiles -- 10.2 us per layer for
// 2 x 110 scores per (sequence, head).  Here ONE WAVE owns a (sequence, head) and nothing goes through LDS: the K and V rows are requested
// up front as whole 128-byte lines (8 lanes per row, 16 rows-of-8 groups: lane l holds 16-byte chunk l & 7 of rows 8 i + (l >> 3)), a score
// is 4 v_dot2_f32_f16 per lane + a 3-step DPP reduction over the row's 8 lanes, the softmax statistics and P.V reduce over the 8 row groups of
// a wave (row rotate + v_permlane16/32_swap; the ds_bpermute form of __shfl_xor cost ~100 dependent cycles per exchange: 9 us per launch).  (A first form with one WHOLE row per lane -- 16 loads of 16 bytes at a 3 KB stride per lane -- measured 12.1 us:
// every load instruction touched 64 cache lines.)  Same arithmetic as the forward tile: scores in the log2 domain with the additive -10000
// mask term, P~ = exp2(s - max) rounded to fp16 for P.V, the fp32 sum of the UNROUNDED P~ as the normaliser, one rounding of the context.
// ---------------------------------------------------------------------------------------------
template <int LQ>
__global__ __launch_bounds__(256) void attn_decode_small_kernel(AttnParams p) {
    const int lane = threadIdx.x & 63;
    const int item = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= p.B * p.heads) return;
    const int b = item / p.heads, h = item - b * p.heads;
    const int Lk = p.Lk;
    const int rg = lane >> 3, c = lane & 7;                 // row inside a group of 8, 16-byte chunk of the 128-byte head row
    constexpr int NG = 16;                                  // 16 groups x 8 rows = 128 keys
    const int ng = (Lk + 7) >> 3;                           // groups that hold a real key (wave-uniform)
    f16x8 kr[NG], vr[NG];
    const int64_t own_base = (int64_t)b * p.bs_kv, pre_base = (int64_t)(b / p.beams) * p.bs_kv2;
#pragma unroll
    for (int i = 0; i < NG; ++i) {
        if (i < ng) {
            // branch-free source select (a divergent branch per group made the compiler wait for every group's loads before the next
            // group's branch: 16 serialized round trips, 13.5 us)
            const int kc = min(8 * i + rg, Lk - 1);
            const bool pre = kc < p.n_prefix;
            const int64_t off = ((pre ? pre_base : own_base) + kc) * p.ld_kv + h * HD + 8 * c;
            kr[i] = ld8((pre ? p.k2 : p.k) + off);
            vr[i] = ld8((pre ? p.v2 : p.v) + off);
        }
    }
    f16x8 qr[LQ];
    uint8_t mb[LQ][NG];                                     // mask bytes of this lane's rows: requested with everything else (a load inside the score loop is a round trip per group)
#pragma unroll
    for (int q = 0; q < LQ; ++q) {
        const int qq = min(q, p.Lq - 1);
        qr[q] = ld8(p.q + ((int64_t)b * p.bs_q + qq) * p.ld_q + h * HD + 8 * c);
        const uint8_t* mrow = p.mask + ((int64_t)b * p.Lq + qq) * p.Lp;
#pragma unroll
        for (int i = 0; i < NG; ++i) mb[q][i] = (i < ng) ? mrow[min(8 * i + rg, p.Lp - 1)] : (uint8_t)2;
    }
    const float sc2 = p.scale * LOG2E_F;
#pragma unroll
    for (int q = 0; q < LQ; ++q) {
        if (q >= p.Lq) break;
        float s2[NG];
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < NG; ++i) {
            s2[i] = -INFINITY;
            if (i < ng) {
                const int key = 8 * i + rg;
                const uint32_t mbyte = mb[q][i];
                float acc = 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    acc = __builtin_amdgcn_fdot2((f16x2){kr[i][2 * e], kr[i][2 * e + 1]}, (f16x2){qr[q][2 * e], qr[q][2 * e + 1]}, acc, false);
                acc = sum8(acc);                            // all 8 lanes of the row hold its score (DPP: no LDS round trip)
                const float term = (key >= Lk || mbyte >= 2u) ? -INFINITY : (mbyte == 1u ? 0.f : -MASK_C1);
                s2[i] = fmaf(acc, sc2, term);
                mx = fmaxf(mx, s2[i]);
            }
        }
        mx = max_over_groups8(mx);
        float sum = 0.f;
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = 0.f;
#pragma unroll
        for (int i = 0; i < NG; ++i) {
            if (i < ng) {
                const float pr = __builtin_amdgcn_exp2f(s2[i] - mx);
                sum += pr;
                const float ph = (float)(f16)pr;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = fmaf(ph, (float)vr[i][e], o[e]);
            }
        }
        sum = sum_over_groups8(sum);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = sum_over_groups8(o[e]);
        if (rg == 0) {
            const float inv = 1.f / sum;
            f16x8 ov;
#pragma unroll
            for (int e = 0; e < 8; ++e) ov[e] = (f16)(o[e] * inv);
            st8(p.ctx + ((int64_t)b * p.Lq + q) * p.ld_ctx + h * HD + 8 * c, ov);
        }
    }
}

static int launch_attn_decode_small(AttnParams& p, hipStream_t s) {
    const dim3 grid(cdiv(p.B * p.heads, 4));
    if (p.Lq == 1) hipLaunchKernelGGL(attn_decode_small_kernel<1>, grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL(attn_decode_small_kernel<2>, grid, dim3(256), 0, s, p);
    VLP_CHECK_LAUNCH("vlp_attn_decode");
    return VLP_OK;
}

extern "C" int vlp_attn_decode(const vlp_attn_decode_args* a, void* stream) {
    VLP_CHECK_ARG(a != nullptr && a->q && a->k && a->v && a->mask && a->ctx, "vlp_attn_decode: null operand");
    VLP_ENTER(a->q, "vlp_attn_decode");
    VLP_CHECK_ARG(a->B > 0 && a->heads > 0 && a->Lq > 0 && a->Lk > 0 && a->Lk <= 256, "vlp_attn_decode: bad shape (Lk <= 256)");
    VLP_CHECK_ARG(a->ld_q % 8 == 0 && a->ld_kv % 8 == 0 && a->ld_ctx % 4 == 0, "vlp_attn_decode: leading dims");
    const bool ctx16 = a->ld_ctx % 8 == 0 && (uintptr_t)a->ctx % 16 == 0;
    VLP_CHECK_ARG(((uintptr_t)a->q | (uintptr_t)a->k | (uintptr_t)a->v) % 16 == 0 && (uintptr_t)a->ctx % 8 == 0 && (uintptr_t)a->mask % 4 == 0,
                  "vlp_attn_decode: alignment");
    VLP_CHECK_ARG(a->kv_rows_per_batch >= a->Lk && a->q_rows_per_batch >= a->Lq, "vlp_attn_decode: batch strides");
    AttnParams p = {};
    p.mask = a->mask;
    p.ctx = (f16*)a->ctx; p.ld_ctx = a->ld_ctx; p.lse = nullptr;
    p.B = a->B; p.L = a->Lk; p.heads = a->heads; p.H = a->heads * HD;
    p.Lp = (a->Lk + 31) / 32 * 32;
    p.scale = a->scale;
    p.drop = make_drop(0.f, 0, 0);
    p.q = (const f16*)a->q; p.ld_q = a->ld_q; p.bs_q = a->q_rows_per_batch;
    p.k = (const f16*)a->k; p.v = (const f16*)a->v; p.ld_kv = a->ld_kv; p.bs_kv = a->kv_rows_per_batch;
    p.Lq = a->Lq; p.Lk = a->Lk;
    if (a->n_prefix > 0) {
        VLP_CHECK_ARG(a->k_prefix && a->v_prefix && a->beams > 0 && a->n_prefix <= a->Lk && a->prefix_rows_per_batch >= a->n_prefix &&
                          a->B % a->beams == 0 && ((uintptr_t)a->k_prefix | (uintptr_t)a->v_prefix) % 16 == 0,
                      "vlp_attn_decode: bad shared-prefix arguments");
        p.k2 = (const f16*)a->k_prefix; p.v2 = (const f16*)a->v_prefix; p.bs_kv2 = a->prefix_rows_per_batch;
        p.n_prefix = a->n_prefix; p.beams = a->beams;
    }
    if (p.beams < 1) p.beams = 1;
    // token steps (1-2 new rows per sequence, history <= 128): the wave-per-(sequence, head) kernel; VLP_ATTN_DECODE_SMALL=0: the general kernel
    static const int small_on = [] { const char* e = getenv("VLP_ATTN_DECODE_SMALL"); return (e && e[0] == '0') ? 0 : 1; }();
    if (small_on && ctx16 && a->Lq <= 2 && a->Lk <= 128) return launch_attn_decode_small(p, (hipStream_t)stream);
    return launch_attn_fwd(p, (hipStream_t)stream);
}

extern "C" int vlp_attn_bwd(const vlp_attn_bwd_args* a, void* stream) {
    VLP_CHECK_ARG(a != nullptr, "vlp_attn_bwd: null args");
    VLP_ENTER(a->qkv, "vlp_attn_bwd");
    int rc = attn_common_check("vlp_attn_bwd", a->qkv, a->ld_qkv, a->mask, a->B, a->L, a->heads);
    if (rc) return rc;
    VLP_CHECK_ARG(a->ctx && a->dctx && a->lse && a->dqkv && a->delta && a->mask_t, "vlp_attn_bwd: null operand");
    VLP_CHECK_ARG((uintptr_t)a->mask_t % 4 == 0, "vlp_attn_bwd: mask_t alignment");
    VLP_CHECK_ARG(a->ld_ctx % 8 == 0 && a->ld_dctx % 8 == 0 && a->ld_dqkv % 4 == 0, "vlp_attn_bwd: leading dims");
    VLP_CHECK_ARG(((uintptr_t)a->ctx | (uintptr_t)a->dctx) % 16 == 0 && (uintptr_t)a->dqkv % 8 == 0, "vlp_attn_bwd: alignment");
    AttnParams p = {};
    p.qkv = (const f16*)a->qkv; p.ld_qkv = a->ld_qkv; p.mask = a->mask; p.mask_t = a->mask_t;
    p.ctx = (f16*)a->ctx; p.ld_ctx = a->ld_ctx;
    p.dctx = (const f16*)a->dctx; p.ld_dctx = a->ld_dctx;
    p.lse = (float*)a->lse; p.dqkv = (f16*)a->dqkv; p.ld_dqkv = a->ld_dqkv; p.delta = a->delta;
    p.B = a->B; p.L = a->L; p.heads = a->heads; p.H = a->heads * HD;
    p.Lp = (a->L + 31) / 32 * 32;
    p.scale = a->scale;
    p.drop = make_drop(a->dropout_p, a->seed, a->rng_stream);
    p.skip = attn_skip_enabled();
    p.row_off = a->row_off;
    const int LP = lp_of(a->L);
    dim3 grid(a->B * a->heads);
    hipStream_t s = (hipStream_t)stream;
    // one kernel for dQ, dK, dV (default since round 4).  Investigation builds (-DVLP_LAB_BUILD) also carry the two-kernel form
    // (VLP_ATTN_BWD=split: the reference the merged kernel was validated against) and the exchange-tile form at every L (VLP_ATTN_BWD=xch)
    const char* bwd_env = getenv("VLP_ATTN_BWD");         // read at every launch so that tests can toggle it
#ifdef VLP_LAB_BUILD
    const bool split = bwd_env && bwd_env[0] == 's', xch = bwd_env && bwd_env[0] == 'x';
#else
    const bool split = false, xch = false;
    VLP_CHECK_ARG(!(bwd_env && (bwd_env[0] == 's' || bwd_env[0] == 'x')), "vlp_attn_bwd: VLP_ATTN_BWD=%s needs a library built with -DVLP_LAB_BUILD", bwd_env);
#endif
    VLP_CHECK_ARG(a->row_off == nullptr || (LP <= 192 && !split && !xch),
                  "vlp_attn_bwd: packed rows (row_off) are supported by the one-kernel backward at L <= 192");
    if (!split) {
        const size_t smem_one = (size_t)3 * LP * HD * 2 + (size_t)3 * LP * 4;
#define LAUNCH_ONE(NT_, KPW_)                                                                                           \
    do {                                                                                                             \
        VLP_ONCE_PER_DEVICE((void)hipFuncSetAttribute((const void*)attn_bwd_one_kernel<NT_, KPW_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_one)); \
        hipLaunchKernelGGL((attn_bwd_one_kernel<NT_, KPW_>), grid, dim3((NT_) / (KPW_) * 64), smem_one, s, p);       \
    } while (0)
        const size_t smem_full = (size_t)3 * LP * HD * 2 + (size_t)LP * LP * 2 + (size_t)3 * LP * 4;
        // persistent grid of the whole-dS^T kernel: one workgroup per CU walks items b*heads + h = blockIdx, + grid, ... (the next item's
        // tiles prefetched into registers); VLP_ATTN_BWD_GRID=0 launches one workgroup per item instead (A/B runs)
        static const int ncu = [] { int dev = 0, n = 256; (void)hipGetDevice(&dev); if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 256; return n; }();
        const char* ge = getenv("VLP_ATTN_BWD_GRID");
        const int gcap = ge ? atoi(ge) : ncu;
        const dim3 pgrid((gcap > 0 && gcap < a->B * a->heads) ? gcap : a->B * a->heads);
#define LAUNCH_FULL(NT_)                                                                                             \
    do {                                                                                                             \
        if (p.drop.thresh) {                                                                                         \
            VLP_ONCE_PER_DEVICE((void)hipFuncSetAttribute((const void*)attn_bwd_full_kernel<NT_, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_full)); \
            hipLaunchKernelGGL((attn_bwd_full_kernel<NT_, true>), pgrid, dim3((NT_) * 64), smem_full, s, p);         \
        } else {                                                                                                     \
            VLP_ONCE_PER_DEVICE((void)hipFuncSetAttribute((const void*)attn_bwd_full_kernel<NT_, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_full)); \
            hipLaunchKernelGGL((attn_bwd_full_kernel<NT_, false>), pgrid, dim3((NT_) * 64), smem_full, s, p);        \
        }                                                                                                            \
    } while (0)
#ifdef VLP_LAB_BUILD
        if (xch && LP == 64) LAUNCH_ONE(4, 1);
        else if (xch && LP == 128) LAUNCH_ONE(8, 1);
        else if (xch && LP == 192) LAUNCH_ONE(12, 1);
        else
#endif
        if (LP == 64) LAUNCH_FULL(4);
        else if (LP == 128) LAUNCH_FULL(8);
        else if (LP == 192) LAUNCH_FULL(12);
        else LAUNCH_ONE(16, 2);                               // L > 192: dS^T (128 KB) does not fit beside Q, dO, K: exchange-tile form
#undef LAUNCH_FULL
#undef LAUNCH_ONE
        VLP_CHECK_LAUNCH("vlp_attn_bwd");
        return VLP_OK;
    }
#ifdef VLP_LAB_BUILD
    {
        const size_t smem_dq = (size_t)2 * LP * HD * 2;
        const size_t smem_dkv = (size_t)2 * LP * HD * 2 + (size_t)3 * LP * 4;
        dim3 block(ATT_THREADS);
        static const int nw12 = attn_waves_nt12();
#define LAUNCH_BWD(NT_, NW_)                                                                                         \
    do {                                                                                                             \
        VLP_ONCE_PER_DEVICE((void)hipFuncSetAttribute((const void*)attn_bwd_dq_kernel<NT_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_dq)); \
        VLP_ONCE_PER_DEVICE((void)hipFuncSetAttribute((const void*)attn_bwd_dkv_kernel<NT_, NW_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_dkv)); \
        hipLaunchKernelGGL(attn_bwd_dq_kernel<NT_>, grid, block, smem_dq, s, p);                                     \
        hipLaunchKernelGGL((attn_bwd_dkv_kernel<NT_, NW_>), grid, dim3((NW_) * 64), smem_dkv, s, p);                   \
    } while (0)
        if (LP == 64) LAUNCH_BWD(4, 8); else if (LP == 128) LAUNCH_BWD(8, 8);
        else if (LP == 192) { if (nw12 == 4) LAUNCH_BWD(12, 4); else LAUNCH_BWD(12, 8); }
        else LAUNCH_BWD(16, 8);
#undef LAUNCH_BWD
        VLP_CHECK_LAUNCH("vlp_attn_bwd");
    }
#endif
    return VLP_OK;
}

// int64 [B,L,L] -> uint8 [B,L,Lp]  (1 attend, 0 masked, 2 = padding column) and, optionally, the key-major copy
// [B,Lp,Lp] (row `key`, one byte per query in the lane order of maskt_pos; 2 wherever key >= L or q >= L)
__global__ void mask_pack_kernel(const int64_t* mask, uint8_t* out, uint8_t* out_t, int L, int Lp, int B) {
    const int64_t n1 = (int64_t)B * L * Lp, n2 = out_t ? (int64_t)B * Lp * Lp : 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n1 + n2; i += (int64_t)gridDim.x * blockDim.x) {
        if (i < n1) {
            const int64_t r = i / Lp;
            const int c = (int)(i % Lp);
            out[i] = c < L ? (mask[r * L + c] != 0 ? 1 : 0) : 2;
        } else {
            const int64_t j = i - n1;
            const int q = maskt_query((int)(j % Lp), Lp);           // key-major rows in lane order (maskt_pos)
            const int key = (int)((j / Lp) % Lp);
            const int64_t b = j / ((int64_t)Lp * Lp);
            out_t[j] = (q < L && key < L) ? (mask[(b * L + q) * L + key] != 0 ? 1 : 0) : 2;
        }
    }
}
extern "C" int vlp_mask_pack(const int64_t* mask, uint8_t* out, uint8_t* out_t, int32_t B, int32_t L, int32_t Lp, void* stream) {
    VLP_CHECK_ARG(mask && out && B > 0 && L > 0, "vlp_mask_pack: bad args");
    VLP_ENTER(mask, "vlp_mask_pack");
    VLP_CHECK_ARG(Lp == (L + 31) / 32 * 32, "vlp_mask_pack: Lp must be roundup32(L)");
    const int64_t total = (int64_t)B * L * Lp + (out_t ? (int64_t)B * Lp * Lp : 0);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(mask_pack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, mask, out, out_t, L, Lp, B);
    VLP_CHECK_LAUNCH("vlp_mask_pack");
    return VLP_OK;
}

// rectangular slice [B, Lq, Lk] of an int64 mask given by strides (elements) -> uint8 [B, Lq, Lkp], columns >= Lk hold 2
__global__ void mask_pack_rect_kernel(const int64_t* mask, int64_t bs, int64_t rs, uint8_t* out, int B, int Lq, int Lk, int Lkp) {
    const int64_t total = (int64_t)B * Lq * Lkp;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % Lkp);
        const int64_t r = i / Lkp;
        const int q = (int)(r % Lq);
        const int64_t b = r / Lq;
        out[i] = c < Lk ? (mask[b * bs + q * rs + c] != 0 ? 1 : 0) : 2;
    }
}
extern "C" int vlp_mask_pack_rect(const int64_t* mask, int64_t batch_stride, int64_t row_stride, uint8_t* out, int32_t B, int32_t Lq, int32_t Lk,
                                  int32_t Lkp, void* stream) {
    VLP_CHECK_ARG(mask && out && B > 0 && Lq > 0 && Lk > 0 && Lkp == (Lk + 31) / 32 * 32, "vlp_mask_pack_rect: bad args");
    VLP_ENTER(mask, "vlp_mask_pack_rect");
    const int64_t total = (int64_t)B * Lq * Lkp;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(mask_pack_rect_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, mask, batch_stride, row_stride, out, B, Lq, Lk, Lkp);
    VLP_CHECK_LAUNCH("vlp_mask_pack_rect");
    return VLP_OK;
}

// packed masks straight from the per-sample lengths (seq2seq_loader.py:292-301) -- no int64 [B,L,L] tensor is ever built:
//   s2s[b] != 0:  attend(q, k) = k < st[b]  ||  (st[b] <= q < en[b]  &&  st[b] <= k <= q)
//   else (bi):    attend(q, k) = k < en[b]
// same byte format as mask_pack (1 attend, 0 masked, 2 = padding column / row of the key-major copy)
// region_mask (optional, [B*Nv] bytes): key columns 1..Nv of masked regions are blocked for every query (seq2seq_loader.py:303-304)
__global__ void mask_build_kernel(const int32_t* st, const int32_t* en, const int32_t* s2s, uint8_t* out, uint8_t* out_t, int L, int Lp, int B,
                                  const uint8_t* region_mask, int Nv) {
    const int64_t n1 = (int64_t)B * L * Lp, n2 = out_t ? (int64_t)B * Lp * Lp : 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n1 + n2; i += (int64_t)gridDim.x * blockDim.x) {
        int q, k;
        int64_t b;
        uint8_t* dst;
        if (i < n1) {
            k = (int)(i % Lp);
            q = (int)((i / Lp) % L);
            b = i / ((int64_t)Lp * L);
            dst = out + i;
        } else {
            const int64_t j = i - n1;
            q = maskt_query((int)(j % Lp), Lp);                     // key-major rows in lane order (maskt_pos)
            k = (int)((j / Lp) % Lp);
            b = j / ((int64_t)Lp * Lp);
            dst = out_t + j;
        }
        uint8_t v = 2;
        if (q < L && k < L) {
            const int s = st[b], e = en[b];
            v = s2s[b] ? ((k < s || (q >= s && q < e && k >= s && k <= q)) ? 1 : 0) : (k < e ? 1 : 0);
            if (region_mask && k >= 1 && k <= Nv && region_mask[b * Nv + (k - 1)]) v = 0;
        }
        *dst = v;
    }
}
extern "C" int vlp_mask_build(const int32_t* second_st, const int32_t* second_end, const int32_t* is_s2s, uint8_t* out, uint8_t* out_t, int32_t B,
                              int32_t L, int32_t Lp, const uint8_t* region_mask, int32_t Nv, void* stream) {
    VLP_CHECK_ARG(second_st && second_end && is_s2s && out && B > 0 && L > 0, "vlp_mask_build: bad args");
    VLP_ENTER(second_st, "vlp_mask_build");
    VLP_CHECK_ARG(Lp == (L + 31) / 32 * 32, "vlp_mask_build: Lp must be roundup32(L)");
    VLP_CHECK_ARG(region_mask == nullptr || (Nv > 0 && Nv < L), "vlp_mask_build: region_mask needs 0 < Nv < L");
    const int64_t total = (int64_t)B * L * Lp + (out_t ? (int64_t)B * Lp * Lp : 0);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(mask_build_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, second_st, second_end, is_s2s, out, out_t, L, Lp, B, region_mask, Nv);
    VLP_CHECK_LAUNCH("vlp_mask_build");
    return VLP_OK;
}
