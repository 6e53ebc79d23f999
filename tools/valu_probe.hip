// tools/valu_probe.hip -- measured issue rate of plain fp32 VALU ops on gfx950 (wave64), per occupancy.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize valu_probe.hip -o valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(64) void k(float *out, int iters, float a, float b)
{
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.001f + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) v[i] = v[i] * a;                 // v_mul_f32, 8 independent chains
                else if (MODE == 1) v[i] = v[i] * a + b;        // mul + add (two ops, contract off)
                else v[i] = __builtin_fmaf(v[i], a, b);         // v_fma_f32
            }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
using v2f = float __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(64) void kp(float *out, int iters, float a, float b)
{
    v2f v[8];
    const v2f A = {a, a * 1.00001f}, B = {b, b * 0.5f};
    for (int i = 0; i < 8; ++i) v[i] = (v2f){threadIdx.x * 0.001f + i, threadIdx.x * 0.002f - i};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) v[i] = v[i] * A;                                   // v_pk_mul_f32
                else if (MODE == 1) v[i] = v[i] + B;                              // v_pk_add_f32
                else v[i] = __builtin_elementwise_fma(v[i], A, B);                // v_pk_fma_f32
            }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += v[i].x + v[i].y;
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <int MODE>
void runp(const char *name, int waves_per_simd)
{
    float *d; hipMalloc(&d, 256 * 4 * 8 * 64 * sizeof(float));
    const int iters = 4000, grid = 256 * 4 * waves_per_simd;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(kp<MODE>, dim3(grid), dim3(64), 0, 0, d, 10, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(kp<MODE>, dim3(grid), dim3(64), 0, 0, d, iters, 1.0001f, 0.5f);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double insts = (double)grid * iters * 64;
    double per_simd_per_s = insts / (ms * 1e-3) / (256.0 * 4);
    printf("%-10s waves/SIMD=%d  %.3f ms  %.2f G wave-inst/s/SIMD  => %.2f cycles/inst @2.4GHz  (%.1f T lane-flop-ops/s chip, 2 per lane per inst)\n", name,
           waves_per_simd, ms, per_simd_per_s / 1e9, 2.4e9 / per_simd_per_s, insts * 128 / (ms * 1e-3) / 1e12);
    hipFree(d);
}
template <int MODE>
void run(const char *name, int waves_per_simd, int ops_per_iter)
{
    float *d; hipMalloc(&d, 256 * 4 * 8 * 64 * sizeof(float));
    const int iters = 4000, grid = 256 * 4 * waves_per_simd;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 0, 0, d, 10, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 0, 0, d, iters, 1.0001f, 0.5f);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double insts = (double)grid * iters * ops_per_iter;            // wave-instructions
    double per_simd_per_s = insts / (ms * 1e-3) / (256.0 * 4);
    printf("%-10s waves/SIMD=%d  %.3f ms  %.2f G wave-inst/s/SIMD  => %.2f cycles/inst @2.4GHz  (%.1f T lane-ops/s chip)\n", name,
           waves_per_simd, ms, per_simd_per_s / 1e9, 2.4e9 / per_simd_per_s, insts * 64 / (ms * 1e-3) / 1e12);
    hipFree(d);
}
int main()
{
    for (int w : {1, 2, 4, 5, 8}) {
        run<0>("v_mul", w, 64);
        run<1>("mul+add", w, 128);
        run<2>("v_fma", w, 64);
    }
    for (int w : {1, 4, 8}) {
        runp<0>("pk_mul", w);
        runp<1>("pk_add", w);
        runp<2>("pk_fma", w);
    }
    return 0;
}
