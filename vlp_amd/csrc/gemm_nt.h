// Parameter block shared by the NT GEMM kernels (gemm_nt.hip: 2-barrier kernels; gemm_nt_ph.hip: phased 256-row kernels).
#pragma once
#include "common.h"

struct GemmNtParams {
    const f16* X; int64_t ldx;
    const f16* W; int64_t ldw;
    f16* Y; int64_t ldy;
    const f16* bias;
    const f16* residual; int64_t ldr;
    f16* preact; int64_t ldp;
    const f16* mulsrc; int64_t ldm;
    int M, N, K;
    int act;       // VLP_ACT_*
    int mulmode;   // VLP_MUL_*
    float alpha;
    DropCtx drop;
    const int32_t* row_map;   // packed (padding-free) runs: logical row of output row m for the dropout hash, or nullptr
    int tiles_n;
    int tiles_total; // wave-pipelined kernels: tiles of the launch (persistent variants stride them over the grid)
    int xcd_remap;   // 1: workgroups of one XCD (blockIdx % 8) take a contiguous range of tiles (X-panel reuse in that XCD's L2)
#ifdef VLP_NT_DEBUG
    int dbg;         // investigation build (tools/build_variant_lib.sh): 1 = ring loop without MFMAs, 2 = without refill DMA, 4 = without the epilogue
#endif
};

// dropout row of output row m: its logical index in a packed run (masks bit-identical to the dense run), m itself otherwise
DEVFN uint64_t nt_drop_row(const GemmNtParams& p, int m) { return p.row_map ? (uint64_t)(uint32_t)p.row_map[m] : (uint64_t)m; }

// phased kernels (gemm_nt_ph.hip): bn = 256 or 128 columns per workgroup tile (256 rows); p.xcd_remap honoured
int vlp_gemm_nt_ph_launch(GemmNtParams& p, int bn, int mode, hipStream_t s);

// 256x256 tile, k tiles of 32, 4-stage ring (gemm_nt_k32.hip); p.xcd_remap honoured
int vlp_gemm_nt_k32_launch(GemmNtParams& p, bool sg, hipStream_t s);

// wave-pipelined kernels (gemm_nt_wp.hip): cfg 0..7, see the table there; p.xcd_remap honoured
int vlp_gemm_nt_wp_launch(GemmNtParams& p, int cfg, bool sg, hipStream_t s);

// persistent k-stream kernel (gemm_nt_ps.hip): 256x128 tiles, one workgroup per CU walks a run of tiles, deferred output stores; p.xcd_remap honoured
bool vlp_gemm_nt_ps_eligible(const GemmNtParams& p, bool sg);
int vlp_gemm_nt_ps_launch(GemmNtParams& p, bool sg, hipStream_t s);

// argument validation + parameter block of vlp_gemm_nt (shared by the split-K entry point)
int vlp_gemm_nt_fill_params(const vlp_gemm_nt_args* a, GemmNtParams& p);
// split-K kernels (gemm_nt_splitk.hip)
int vlp_gemm_nt_splitk_launch(GemmNtParams& p, int splits, float* workspace, int64_t workspace_bytes, hipStream_t s);
