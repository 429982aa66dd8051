// Staging-path microbenchmark (round 3): how fast can ONE workgroup per CU stream a 256x128x64 fp16 GEMM stage ring (48 KiB per k tile,
// 3 slots, 4 waves, barrier per tile) from global memory into LDS, by path:
//   mode 0: every 1-KiB piece by LDS-DMA (global_load_lds_dwordx4)               -- what gemm_nt / gemm_nt_wp do
//   mode 1: X pieces by LDS-DMA, W pieces global_load_dwordx4 -> VGPR -> ds_write_b128 (one tile later)
//   mode 2: every piece through VGPRs
//   mode 3: X by LDS-DMA only (W not loaded at all: what the X stream alone costs)
//   mode 4: mode 0 + a FIFTH wave that touches the X lines of stage kt + PD with one dword load per 128-byte line (64 lines per instruction,
//           never waited for): does pulling the cold operand into L2 a few stages early lift the stream from the cold to the warm rate?
//   mode 5: mode 0 + a fifth wave that touches this workgroup's SHARE (rows with (row / 8) % tiles_n == tile_n) of those lines with scalar
//           loads (s_load_dword: the scalar cache's path into L2, nothing crosses the vector L1)
// No MFMAs, no fragment reads: the loop is the memory side of the GEMM main loop only.
// build: hipcc --offload-arch=gfx950 -O3 -o gpurun_out/dma_path_bench tools/dma_path_bench.hip ; run: gpurun_out/dma_path_bench [M N K]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ void glds16_s(const char* sbase, uint32_t voff, uint32_t lds_dst) {
    const uint32_t m0v = __builtin_amdgcn_readfirstlane(lds_dst);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(m0v), "v"(voff), "s"(sbase) : "memory", "m0");
}
__device__ __forceinline__ u32x4 gload16_s(const char* sbase, uint32_t voff) {
    u32x4 r;
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(r) : "v"(voff), "s"(sbase) : "memory");
    return r;
}

template <int MODE, int PD = 4>
__global__ __launch_bounds__((MODE >= 4 ? 320 : 256), 1) void stream_kernel(const char* X, const char* W, int M, int N, int K, int tiles_n, uint32_t* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BM = 256, BN = 128, ROWB = 128, NS = 3, STAGE = (BM + BN) * ROWB, XBYTES = BM * ROWB;
    constexpr int LPX = 8, LPW = 4;
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int bid = blockIdx.x;
    { const int nb = gridDim.x, q = nb >> 3, r = nb & 7, xcd = bid & 7, loc = bid >> 3; bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc; }
    const int m0 = (bid / tiles_n) * BM, n0 = (bid % tiles_n) * BN;
    const int nk = K / 64;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)smem;
    if (MODE >= 4 && wid == 4) {
        // ---- prefetch wave: same barrier sequence as the DMA waves (2 per k tile), touches stage kt + PD ------------------------------
        const int tile_n = bid % tiles_n;
        // the touched words land in registers nobody reads; they are DEDICATED for the whole loop (an asynchronous return into a register
        // the compiler had meanwhile reused for an address is a memory fault)
        uint32_t vj[4] = {0u, 0u, 0u, 0u}, sj = 0u;
        for (int kt = 0; kt < nk; ++kt) {
            __builtin_amdgcn_s_barrier();
            const int pk = kt + PD;
            if (pk < nk) {
                if (MODE == 4) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int r = j * 64 + lane;
                        const char* a = X + (int64_t)min(m0 + r, M - 1) * K * 2 + (int64_t)pk * ROWB;
                        asm volatile("global_load_dword %0, %1, off" : "+v"(vj[j]) : "v"(a) : "memory");
                    }
                } else {
                    for (int r8 = tile_n; r8 < 32; r8 += tiles_n) {          // 8-row groups of the X tile shared out over the column tiles
#pragma unroll
                        for (int rr = 0; rr < 8; ++rr) {
                            const char* a = X + (int64_t)min(m0 + r8 * 8 + rr, M - 1) * K * 2 + (int64_t)pk * ROWB;
                            asm volatile("s_load_dword %0, %1, 0x0" : "+s"(sj) : "s"(a) : "memory");
                        }
                    }
                }
            }
            __builtin_amdgcn_s_barrier();
        }
        if (MODE == 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        asm volatile("" ::"v"(vj[0]), "v"(vj[1]), "v"(vj[2]), "v"(vj[3]), "s"(sj));
        return;
    }
    const int rb = lane >> 3, pc = lane & 7;
    uint32_t xo[LPX], wo[LPW];
#pragma unroll
    for (int j = 0; j < LPX; ++j) { const int r = (wid + 4 * j) * 8 + rb; const int mr = min(m0 + r, M - 1); xo[j] = (uint32_t)mr * (uint32_t)K * 2u + (uint32_t)((pc ^ (r & 7)) << 4); }
#pragma unroll
    for (int j = 0; j < LPW; ++j) { const int r = (wid + 4 * j) * 8 + rb; const int nr = min(n0 + r, N - 1); wo[j] = (uint32_t)nr * (uint32_t)K * 2u + (uint32_t)((pc ^ (r & 7)) << 4); }
    u32x4 xr[NS][LPX], wr[NS][LPW];            // register stages, one set per ring slot (the k loop is unrolled by NS: static indices)
    constexpr int PER = (MODE == 3) ? LPX : LPX + LPW;      // vm ops per stage per wave
    uint32_t acc = 0;
#define ISSUE(R, kt_)                                                                                                         \
    do {                                                                                                                      \
        const char* xs = X + (int64_t)(kt_) * ROWB;                                                                           \
        const char* ws = W + (int64_t)(kt_) * ROWB;                                                                           \
        const uint32_t dst = __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)(R) * STAGE);                                    \
        if (MODE == 1 || MODE == 2) { _Pragma("unroll") for (int j = 0; j < LPW; ++j) wr[R][j] = gload16_s(ws, wo[j]); }      \
        if (MODE == 2) { _Pragma("unroll") for (int j = 0; j < LPX; ++j) xr[R][j] = gload16_s(xs, xo[j]); }                   \
        else { _Pragma("unroll") for (int j = 0; j < LPX; ++j) glds16_s(xs, xo[j], dst + (uint32_t)(wid + 4 * j) * 1024u); }  \
        if (MODE == 0 || MODE >= 4) { _Pragma("unroll") for (int j = 0; j < LPW; ++j) glds16_s(ws, wo[j], dst + XBYTES + (uint32_t)(wid + 4 * j) * 1024u); } \
    } while (0)
    // one k tile: stage kt (slot R) complete -- the two younger stages stay in flight --, its register pieces go to LDS, barrier,
    // "consume" a word, barrier, refill slot R with stage kt + NS
#define BODY(R, kt_)                                                                                                          \
    do {                                                                                                                      \
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER) : "memory");                                                        \
        if (MODE == 1 || MODE == 2) {                                                                                         \
            char* base = smem + (R) * STAGE;                                                                                  \
            _Pragma("unroll") for (int j = 0; j < LPW; ++j) { asm volatile("" : "+v"(wr[R][j])); *reinterpret_cast<u32x4*>(base + XBYTES + (wid + 4 * j) * 1024 + lane * 16) = wr[R][j]; } \
            if (MODE == 2) { _Pragma("unroll") for (int j = 0; j < LPX; ++j) { asm volatile("" : "+v"(xr[R][j])); *reinterpret_cast<u32x4*>(base + (wid + 4 * j) * 1024 + lane * 16) = xr[R][j]; } } \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                \
        }                                                                                                                     \
        __builtin_amdgcn_s_barrier();                                                                                         \
        acc += *reinterpret_cast<const uint32_t*>(smem + (R) * STAGE + (((lane * 772 + (kt_) * 52) % STAGE) & ~3));        \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                    \
        __builtin_amdgcn_s_barrier();                                                                                         \
        ISSUE(R, min((kt_) + NS, nk - 1));                                                                                    \
    } while (0)
    ISSUE(0, 0); ISSUE(1, min(1, nk - 1)); ISSUE(2, min(2, nk - 1));
    for (int kt = 0; kt < nk; kt += 3) {          // nk % 3 == 0 (K = 768, 2304, 3072)
        BODY(0, kt); BODY(1, kt + 1); BODY(2, kt + 2);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (acc == 0x12345678u) sink[0] = acc;
}

#define MODE_BYTES(name) ((name)[0] == '3' ? 32768.0 : 49152.0)
int main(int argc, char** argv) {
    int M = argc > 1 ? atoi(argv[1]) : 10688, N = argc > 2 ? atoi(argv[2]) : 768, K = argc > 3 ? atoi(argv[3]) : 3072;
    const int ROT = 12;
    std::vector<char*> xs(ROT), ws(ROT);
    for (int i = 0; i < ROT; ++i) {
        CHECK(hipMalloc(&xs[i], (size_t)M * K * 2)); CHECK(hipMalloc(&ws[i], (size_t)N * K * 2));
        CHECK(hipMemset(xs[i], 0x3c + i, (size_t)M * K * 2)); CHECK(hipMemset(ws[i], 0x2e + i, (size_t)N * K * 2));
    }
    uint32_t* sink; CHECK(hipMalloc(&sink, 64));
    const int tiles_n = (N + 127) / 128, grid = ((M + 255) / 256) * tiles_n;
    const size_t smem = 3 * (256 + 128) * 128;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    auto run = [&](auto kern, const char* name, int rot, int threads = 256) {
        CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        for (int i = 0; i < 6; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), smem, 0, xs[i % rot], ws[i % rot], M, N, K, tiles_n, sink);
        CHECK(hipDeviceSynchronize());
        const int iters = 36;
        CHECK(hipEventRecord(e0));
        for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), smem, 0, xs[i % rot], ws[i % rot], M, N, K, tiles_n, sink);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / iters, bytes = (double)grid * (K / 64) * (MODE_BYTES(name));
        fflush(stdout); printf("%-40s rot=%-2d M=%d N=%d K=%d WGs=%d  %7.1f us  %6.1f GB/s per CU  %5.2f TB/s\n", name, rot, M, N, K, grid, us, bytes / grid / us / 1e3, bytes / us / 1e6);
    };
    for (int rot : {12, 1}) {
        run(stream_kernel<0>, "0 all LDS-DMA", rot);
        run(stream_kernel<1>, "1 X LDS-DMA + W via VGPR", rot);
        run(stream_kernel<2>, "2 all via VGPR", rot);
        run(stream_kernel<3>, "3 X LDS-DMA only (32 KiB/tile)", rot);
        run(stream_kernel<4, 3>, "4 + vector line-touch, 3 ahead", rot, 320);
        run(stream_kernel<4, 6>, "4 + vector line-touch, 6 ahead", rot, 320);
        run(stream_kernel<5, 3>, "5 + scalar line-touch (share), 3 ahead", rot, 320);
        run(stream_kernel<5, 6>, "5 + scalar line-touch (share), 6 ahead", rot, 320);
    }
    return 0;
}
