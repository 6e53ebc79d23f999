// tools/pk_opsel_probe.hip -- issue rate of v_pk_mul_f32 with and without op_sel broadcast modifiers,
// and of dependent vs independent packed chains, on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
using v2f = float __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(64) void k(float *out, int iters, float a)
{
    v2f v[8], t = {a, a * 1.00001f};
    for (int i = 0; i < 8; ++i) v[i] = (v2f){threadIdx.x * 0.001f + i, threadIdx.x * 0.002f - i};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(v[i]) : "v"(t));
                if (MODE == 1) asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel:[0,0] op_sel_hi:[1,0]" : "+v"(v[i]) : "v"(t));
                if (MODE == 2) asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,1]" : "+v"(v[i]) : "v"(t));
                if (MODE == 3) asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel:[1,0] op_sel_hi:[0,1]" : "+v"(v[i]) : "v"(t));
                if (MODE == 4) asm volatile("v_pk_add_f32 %0, %0, %1 neg_lo:[0,1]" : "+v"(v[i]) : "v"(t));
                if (MODE == 5) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[i].x) : "v"(t.x));
                if (MODE == 6) asm volatile("v_pk_mul_f32 %0, %0, %1\n\ts_nop 0" : "+v"(v[i]) : "v"(t));
                if (MODE == 7) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(v[0]) : "v"(t)); // one dependent chain
            }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += v[i].x + v[i].y;
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <int MODE>
void run(const char *name, int w)
{
    float *d; hipMalloc(&d, 256 * 4 * 8 * 64 * sizeof(float));
    const int iters = 2000, grid = 256 * 4 * w;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 0, 0, d, 10, 1.0001f);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 0, 0, d, iters, 1.0001f);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double per_simd = (double)grid * iters * 64 / (ms * 1e-3) / 1024.0;
    printf("%-34s waves/SIMD=%d  %.2f cycles/inst @2.4GHz\n", name, w, 2.4e9 / per_simd);
    hipFree(d);
}
int main()
{
    for (int w : {1, 4}) {
        run<0>("pk_mul plain", w);
        run<1>("pk_mul op_sel_hi:[1,0] (bcast lo)", w);
        run<2>("pk_mul op_sel bcast hi", w);
        run<3>("pk_mul op_sel swap", w);
        run<4>("pk_add neg_lo", w);
        run<5>("v_mul_f32", w);
        run<6>("pk_mul + s_nop 0", w);
        run<7>("pk_mul dependent chain", w);
    }
}
