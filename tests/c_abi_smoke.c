/* C-only consumer of the C ABI (no Python, no torch, no C++): proves that include/vlp_hip.h stands alone and that the entry points
 * may be driven from several host threads (SURVEY.md 8b threading contract).  Built and run by tests/test_00_kernels_gpu.py:
 *   gcc -O2 -D__HIP_PLATFORM_AMD__ -I include -I /opt/rocm/include tests/c_abi_smoke.c -L vlp_amd -lvlp_hip -L /opt/rocm/lib -lamdhip64 -lpthread -lm
 * Each of two threads owns a stream and its buffers, and runs vlp_gemm_nt (Y = X W^T + bias) followed by vlp_layernorm_fwd 50 times;
 * results are checked against a plain C computation.  Exit code 0 = pass. */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "vlp_hip.h"

#define MM 200
#define NN 256
#define KK 128

static uint16_t f2h(float f) {           /* round-to-nearest-even float -> IEEE half (normal range is enough here) */
    uint32_t x; memcpy(&x, &f, 4);
    uint32_t sign = (x >> 16) & 0x8000u;
    int32_t e = (int32_t)((x >> 23) & 0xff) - 127 + 15;
    uint32_t m = x & 0x7fffffu;
    if (e <= 0) return (uint16_t)sign;
    if (e >= 31) return (uint16_t)(sign | 0x7c00u);
    uint32_t h = sign | ((uint32_t)e << 10) | (m >> 13);
    uint32_t rem = m & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) h++;
    return (uint16_t)h;
}
static float h2f(uint16_t h) {
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 0x1f, m = h & 0x3ffu, x;
    if (e == 0) { float f = (float)m * 5.9604644775390625e-8f; return (h & 0x8000u) ? -f : f; }
    x = sign | ((e - 15 + 127) << 23) | (m << 13);
    float f; memcpy(&f, &x, 4); return f;
}

typedef struct { int id; int failed; } job_t;

static void* worker(void* arg) {
    job_t* job = (job_t*)arg;
    unsigned seed = 1234u + 77u * (unsigned)job->id;
    uint16_t *hx = malloc(MM * KK * 2), *hw = malloc(NN * KK * 2), *hb = malloc(NN * 2), *hg = malloc(NN * 2), *hbe = malloc(NN * 2), *hy = malloc(MM * NN * 2);
    float *ref = malloc(MM * NN * sizeof(float));
    for (int i = 0; i < MM * KK; ++i) hx[i] = f2h(((float)rand_r(&seed) / RAND_MAX - 0.5f));
    for (int i = 0; i < NN * KK; ++i) hw[i] = f2h(((float)rand_r(&seed) / RAND_MAX - 0.5f) * 0.2f);
    for (int i = 0; i < NN; ++i) { hb[i] = f2h(((float)rand_r(&seed) / RAND_MAX - 0.5f)); hg[i] = f2h(1.0f + 0.1f * i / NN); hbe[i] = f2h(0.01f * (i % 7)); }
    /* reference: pre = fp16(X W^T + b); y = LayerNorm(pre) * gamma + beta  (eps 1e-12, biased variance) */
    for (int m = 0; m < MM; ++m) {
        float mean = 0.f, var = 0.f;
        for (int n = 0; n < NN; ++n) {
            float s = 0.f;
            for (int k = 0; k < KK; ++k) s += h2f(hx[m * KK + k]) * h2f(hw[n * KK + k]);
            ref[m * NN + n] = h2f(f2h(s + h2f(hb[n])));
            mean += ref[m * NN + n];
        }
        mean /= NN;
        for (int n = 0; n < NN; ++n) var += (ref[m * NN + n] - mean) * (ref[m * NN + n] - mean);
        var /= NN;
        for (int n = 0; n < NN; ++n) ref[m * NN + n] = (ref[m * NN + n] - mean) / sqrtf(var + 1e-12f) * h2f(hg[n]) + h2f(hbe[n]);
    }
    void *dx, *dw, *db, *dg, *dbe, *dpre, *dy, *dmean, *drstd;
    hipStream_t st;
    int bad = 0;
    bad |= hipStreamCreate(&st) != hipSuccess;
    bad |= hipMalloc(&dx, MM * KK * 2) | hipMalloc(&dw, NN * KK * 2) | hipMalloc(&db, NN * 2) | hipMalloc(&dg, NN * 2) | hipMalloc(&dbe, NN * 2);
    bad |= hipMalloc(&dpre, MM * NN * 2) | hipMalloc(&dy, MM * NN * 2) | hipMalloc(&dmean, MM * 4) | hipMalloc(&drstd, MM * 4);
    bad |= hipMemcpy(dx, hx, MM * KK * 2, hipMemcpyHostToDevice) | hipMemcpy(dw, hw, NN * KK * 2, hipMemcpyHostToDevice) | hipMemcpy(db, hb, NN * 2, hipMemcpyHostToDevice);
    bad |= hipMemcpy(dg, hg, NN * 2, hipMemcpyHostToDevice) | hipMemcpy(dbe, hbe, NN * 2, hipMemcpyHostToDevice);
    if (bad) { fprintf(stderr, "thread %d: HIP setup failed\n", job->id); job->failed = 1; return NULL; }
    for (int it = 0; it < 50 && !job->failed; ++it) {
        vlp_gemm_nt_args g;
        memset(&g, 0, sizeof(g));
        g.X = dx; g.ldx = KK; g.W = dw; g.ldw = KK; g.Y = dpre; g.ldy = NN; g.bias = db;
        g.M = MM; g.N = NN; g.K = KK; g.act = VLP_ACT_NONE; g.mul_mode = VLP_MUL_NONE; g.alpha = 1.0f; g.variant = (it & 1) ? 1 : 0;
        if (vlp_gemm_nt(&g, st) != VLP_OK) { fprintf(stderr, "thread %d: vlp_gemm_nt: %s\n", job->id, vlp_last_error_string()); job->failed = 1; break; }
        vlp_layernorm_fwd_args l;
        memset(&l, 0, sizeof(l));
        l.x = dpre; l.ldx = NN; l.gamma = dg; l.beta = dbe; l.y = dy; l.ldy = NN; l.mean = dmean; l.rstd = drstd; l.M = MM; l.H = NN; l.eps = 1e-12f;
        if (vlp_layernorm_fwd(&l, st) != VLP_OK) { fprintf(stderr, "thread %d: vlp_layernorm_fwd: %s\n", job->id, vlp_last_error_string()); job->failed = 1; break; }
    }
    if (!job->failed) {
        if (hipStreamSynchronize(st) != hipSuccess || hipMemcpy(hy, dy, MM * NN * 2, hipMemcpyDeviceToHost) != hipSuccess) job->failed = 1;
        float worst = 0.f, scale = 0.f;
        for (int i = 0; i < MM * NN; ++i) { float d = fabsf(h2f(hy[i]) - ref[i]); if (d > worst) worst = d; if (fabsf(ref[i]) > scale) scale = fabsf(ref[i]); }
        printf("thread %d: max |y - ref| = %.3e (scale %.2f)\n", job->id, worst, scale);
        if (!(worst <= 4e-3f * scale)) job->failed = 1;
    }
    /* a null operand is an error code + message, never a crash */
    vlp_gemm_nt_args bad_args;
    memset(&bad_args, 0, sizeof(bad_args));
    if (vlp_gemm_nt(&bad_args, st) != VLP_ERR_BAD_ARG || strlen(vlp_last_error_string()) == 0) job->failed = 1;
    return NULL;
}

int main(void) {
    if (vlp_version() != VLP_ABI_VERSION) { fprintf(stderr, "ABI version %d != header %d\n", vlp_version(), VLP_ABI_VERSION); return 2; }
    pthread_t th[2];
    job_t jobs[2] = {{0, 0}, {1, 0}};
    for (int i = 0; i < 2; ++i) pthread_create(&th[i], NULL, worker, &jobs[i]);
    for (int i = 0; i < 2; ++i) pthread_join(th[i], NULL);
    if (jobs[0].failed || jobs[1].failed) { fprintf(stderr, "FAILED\n"); return 1; }
    printf("c_abi_smoke ok\n");
    return 0;
}
