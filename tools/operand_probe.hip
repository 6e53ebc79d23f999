// tools/operand_probe.hip -- does an SGPR / literal operand change the issue rate of fp32 VALU on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(64) void k(float *out, int iters, float a)
{
    float v[8], tv = a + threadIdx.x * 1e-9f;
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.001f + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[i]) : "v"(tv));
                if (MODE == 1) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(v[i]) : "s"(a));
                if (MODE == 2) asm volatile("v_mul_f32 %0, 0x3f800347, %0" : "+v"(v[i]));
                if (MODE == 3) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[i]) : "v"(tv));
                if (MODE == 4) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(tv));
                if (MODE == 5) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "s"(a), "v"(tv));
                if (MODE == 6) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(v[i]) : "v"(tv), "v"(tv));
                if (MODE == 7) asm volatile("v_mul_f32 %0, 0.5, %0" : "+v"(v[i]));
            }
        // packed forms on register pairs (two results per instruction)
        if (MODE >= 8) {
            typedef float v2 __attribute__((ext_vector_type(2)));
            v2 p[4], t2 = {tv, tv * 1.5f}, acc[4];
            for (int i = 0; i < 4; ++i) p[i] = v2{v[2 * i], v[2 * i + 1]}, acc[i] = v2{0.f, 0.f};
            const unsigned long long spair = ((unsigned long long)__float_as_uint(a) << 32) | __float_as_uint(a * 0.5f);
#pragma unroll
            for (int r = 0; r < 16; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (MODE == 8) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(t2));
                    if (MODE == 9) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(p[i]) : "s"(spair));
                    if (MODE == 10) asm volatile("v_pk_mul_f32 %0, %1, %0 op_sel:[1,0] op_sel_hi:[0,0]" : "+v"(p[i]) : "s"(spair));
                    if (MODE == 11) { // exact MAC pair: pk_mul (sgpr pair x broadcast vgpr) + pk_add
                        v2 m;
                        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,0]" : "=v"(m) : "s"(spair), "v"(p[i]));
                        asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(acc[i]) : "v"(m));
                    }
                    if (MODE == 12) { // the same two MACs with scalar ops: 2 x (v_mul sgpr + v_add)
                        float m0, m1;
                        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(m0) : "s"(a), "v"(p[i].x));
                        asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc[i].x) : "v"(m0));
                        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(m1) : "s"(a), "v"(p[i].x));
                        asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc[i].y) : "v"(m1));
                    }
                }
            for (int i = 0; i < 4; ++i) v[2 * i] = p[i].x + acc[i].x, v[2 * i + 1] = p[i].y + acc[i].y;
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <int MODE>
void run(const char *name, int w)
{
    float *d; hipMalloc(&d, 256 * 4 * 8 * 64 * sizeof(float));
    const int iters = 4000, grid = 256 * 4 * w;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 0, 0, d, 10, 1.0001f);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 0, 0, d, iters, 1.0001f);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double per_simd = (double)grid * iters * 64 / (ms * 1e-3) / 1024.0;
    printf("%-28s waves/SIMD=%d  %.2f cycles/inst @2.4GHz\n", name, w, 2.4e9 / per_simd);
    hipFree(d);
}
int main()
{
    for (int w : {2, 5, 8}) {
        run<0>("v_mul vgpr,vgpr", w);
        run<1>("v_mul sgpr,vgpr", w);
        run<2>("v_mul literal,vgpr", w);
        run<7>("v_mul inline-const,vgpr", w);
        run<3>("v_add vgpr,vgpr", w);
        run<4>("v_fma vgpr x3", w);
        run<5>("v_fma vgpr,sgpr,vgpr", w);
        run<6>("v_fmac vgpr", w);
        run<8>("v_pk_mul vgpr pairs", w);
        run<9>("v_pk_mul sgpr pair,vgpr", w);
        run<10>("v_pk_mul sgpr pair op_sel", w);
        run<11>("MAC pair: pk_mul s + pk_add (per 64 ops)", w);
        run<12>("MAC pair: 2x(v_mul s + v_add) (per 64 ops)", w);
    }
}
