// Probe (round 6): what does ONE dependency cost inside a kernel, against a kernel boundary?  A decode token step is ~91 dependent launches of
// 32 - 192 workgroups at ~6 us each.  Alternative: ONE launch whose workgroups are the concatenation of all problems; problem j's workgroups wait on
// a device-scope counter that problem j-1's workgroups bump when their stores are released (workgroups are dispatched in block-id order, so a waiter's
// producers are always resident or finished: no deadlock; every spin is bounded anyway).  Each workgroup here reads 1 KB per thread-row of the previous
// problem's output (dependent data), optionally streams `wbytes` of independent "weights" BEFORE it waits (what the real kernels could overlap), writes
// its output and signals.   hipcc --offload-arch=gfx950 -O3 -o dataflow_chain_probe dataflow_chain_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ void wait_counter(const int* c, int target, int* err) {
    if (threadIdx.x == 0) {
        int spins = 0;
        while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (++spins > (1 << 22)) { *err = 1; break; }          // bounded: a logic error must not hang the GPU
            __builtin_amdgcn_s_sleep(8);
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);                   // ONE acquire (L2 invalidate) after the poll, not one per poll
    }
    __syncthreads();
}

extern __shared__ char dyn_lds[];
__global__ __launch_bounds__(256) void chain_kernel(int* counters, int wg_per_prob, int epoch, float* buf, int stride, const float4* weights, int wvec,
                                                    float* sink, int* err) {
    const int prob = blockIdx.x / wg_per_prob, wg = blockIdx.x % wg_per_prob;
    float4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int i = threadIdx.x; i < wvec; i += 256) {               // independent operand (the weights): requested before the wait
        const float4 w = weights[((size_t)prob * wg_per_prob + wg) * wvec + i];
        acc.x += w.x; acc.y += w.y; acc.z += w.z; acc.w += w.w;
    }
    if (prob > 0) wait_counter(&counters[prob - 1], epoch * wg_per_prob, err);
    const int idx = wg * 256 + threadIdx.x;
    // dependent operand: EVERY workgroup of the previous problem contributed (row sum over a strided sample of its output)
    float v = 0.f;
    if (prob > 0) {
        for (int k = 0; k < 8; ++k) v += buf[(size_t)(prob - 1) * stride + ((idx * 8 + k * 977) % (wg_per_prob * 256))];
        v *= 0.125f;
    }
    buf[(size_t)prob * stride + idx] = v + 1.0f + 0.f * (acc.x + acc.y + acc.z + acc.w);
    if (acc.x == 12345.678f) { sink[0] = acc.y; dyn_lds[threadIdx.x] = 1; }
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(&counters[prob], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(256) void single_kernel(float* buf, int stride, int prob, int wg_per_prob, const float4* weights, int wvec, float* sink) {
    const int wg = blockIdx.x;
    float4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int i = threadIdx.x; i < wvec; i += 256) {
        const float4 w = weights[((size_t)prob * wg_per_prob + wg) * wvec + i];
        acc.x += w.x; acc.y += w.y; acc.z += w.z; acc.w += w.w;
    }
    const int idx = wg * 256 + threadIdx.x;
    float v = 0.f;
    if (prob > 0) {
        for (int k = 0; k < 8; ++k) v += buf[(size_t)(prob - 1) * stride + ((idx * 8 + k * 977) % (wg_per_prob * 256))];
        v *= 0.125f;
    }
    buf[(size_t)prob * stride + idx] = v + 1.0f + 0.f * (acc.x + acc.y + acc.z + acc.w);
    if (acc.x == 12345.678f) sink[0] = acc.y;
}

int main(int argc, char** argv) {
    const int nprob = 91;
    const int lds_bytes = argc > 1 ? atoi(argv[1]) * 1024 : 100 * 1024;      // dynamic LDS per workgroup: limits residency as the real kernels' 144 KB do
    CK(hipFuncSetAttribute((const void*)chain_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    printf("dynamic LDS per workgroup: %d KB\n", lds_bytes / 1024);
    for (int wg_per_prob : {32, 96, 192}) {
        for (int wkb : {0, 24, 48}) {                              // KB of independent operand per workgroup
            const int wvec = wkb * 1024 / 16, stride = wg_per_prob * 256;
            int *counters, *err; float *buf, *sink; float4* weights;
            CK(hipMalloc(&counters, nprob * sizeof(int))); CK(hipMemset(counters, 0, nprob * sizeof(int)));
            CK(hipMalloc(&err, sizeof(int))); CK(hipMemset(err, 0, sizeof(int)));
            CK(hipMalloc(&buf, (size_t)nprob * stride * sizeof(float))); CK(hipMalloc(&sink, 16));
            const size_t wbytes = (size_t)nprob * wg_per_prob * (wvec ? wvec : 1) * 16;
            CK(hipMalloc(&weights, wbytes)); CK(hipMemset(weights, 0, wbytes));
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            hipStream_t s; CK(hipStreamCreate(&s));
            int epoch = 0;
            const int reps = 20;
            // (a) one launch, in-kernel dependencies
            for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(chain_kernel, dim3(nprob * wg_per_prob), dim3(256), lds_bytes, s, counters, wg_per_prob, ++epoch, buf, stride, weights, wvec, sink, err);
            CK(hipStreamSynchronize(s));
            CK(hipEventRecord(e0, s));
            for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(chain_kernel, dim3(nprob * wg_per_prob), dim3(256), lds_bytes, s, counters, wg_per_prob, ++epoch, buf, stride, weights, wvec, sink, err);
            CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
            float ms_a; CK(hipEventElapsedTime(&ms_a, e0, e1));
            std::vector<float> h(stride); int herr;
            CK(hipMemcpy(h.data(), buf + (size_t)(nprob - 1) * stride, stride * sizeof(float), hipMemcpyDeviceToHost));
            CK(hipMemcpy(&herr, err, sizeof(int), hipMemcpyDeviceToHost));
            int bad = 0; for (int i = 0; i < stride; ++i) bad += (h[i] != (float)nprob);
            // (b) 91 launches captured in a graph
            hipGraph_t g; hipGraphExec_t ge;
            CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
            for (int p = 0; p < nprob; ++p) hipLaunchKernelGGL(single_kernel, dim3(wg_per_prob), dim3(256), 0, s, buf, stride, p, wg_per_prob, weights, wvec, sink);
            CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            for (int w = 0; w < 3; ++w) CK(hipGraphLaunch(ge, s));
            CK(hipStreamSynchronize(s));
            CK(hipEventRecord(e0, s));
            for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, s));
            CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
            float ms_b; CK(hipEventElapsedTime(&ms_b, e0, e1));
            printf("wg/problem %3d, %2d KB independent operand per wg: one launch %7.1f us per %d-problem chain = %5.2f us per dependency (wrong %d, spin-timeout %d) | graph of %d launches %7.1f us = %5.2f us per launch\n",
                   wg_per_prob, wkb, ms_a * 1e3 / reps, nprob, ms_a * 1e3 / reps / nprob, bad, herr, nprob, ms_b * 1e3 / reps, ms_b * 1e3 / reps / nprob);
            fflush(stdout);
            CK(hipFree(counters)); CK(hipFree(err)); CK(hipFree(buf)); CK(hipFree(sink)); CK(hipFree(weights));
        }
    }
    return 0;
}
