// tools/dc_chain_probe.hip -- what does one step of the DC-bias recurrence (sdrj.cpp:277-283)
//     avept = fl(fl(avept * keep) + p)          two DEPENDENT fp32 operations per sample and component
// cost on gfx950 when nothing else is in the way?  One wave (or two, on different SIMDs) runs a long chain; cycles
// are read with s_memtime around it.  Variants:
//   0  v_mul_f32 -> v_add_f32, VGPR operands, one chain                                   (the dependent-issue floor)
//   1  two independent such chains interleaved in one wave (I and Q)                      (does one hide the other?)
//   2  one chain whose multiply reads the accumulator through DPP wave_shr:1              (the systolic form: lane t
//      computes sample t from lane t-1, the products sit one per lane, no LDS on the chain)
//   3  two interleaved DPP chains in one wave
//   4  variant 2 with row_shr:1 (stays inside a row of 16 lanes)
//   5  four interleaved plain chains (how many chains fill the pipeline?)
//   6  variant 0 with s_setprio 3
//   7  v_mul_f32 -> v_add_f32 where the add takes its product operand from an SGPR
// Output: cycles per chain step (= per sample and component) for each variant at 1 wave per workgroup.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

constexpr int kUnroll = 64;

template <int V>
__global__ __launch_bounds__(64) void k_probe(float *out, long long *cycles, int iters, float keep_in, float p_in)
{
    float a0 = 0.25f + threadIdx.x * 1e-3f, a1 = 0.5f, a2 = 0.75f, a3 = 1.0f;
    float p = p_in + threadIdx.x * 1e-7f, keep = keep_in, t0, t1, t2, t3;
    float sp = p_in;
    if (V == 6)
        __builtin_amdgcn_s_setprio(3);
    const long long c0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            if (V == 0 || V == 6)
                asm volatile("v_mul_f32 %1, %0, %2\n\tv_add_f32 %0, %1, %3" : "+v"(a0), "=&v"(t0) : "v"(keep), "v"(p));
            if (V == 1)
                asm volatile("v_mul_f32 %2, %0, %4\n\tv_mul_f32 %3, %1, %4\n\tv_add_f32 %0, %2, %5\n\tv_add_f32 %1, %3, %5"
                             : "+v"(a0), "+v"(a1), "=&v"(t0), "=&v"(t1)
                             : "v"(keep), "v"(p));
            if (V == 2)
                asm volatile("s_nop 1\n\tv_mul_f32_dpp %1, %0, %2 wave_shr:1 row_mask:0xf bank_mask:0xf\n\tv_add_f32 %0, %1, %3"
                             : "+v"(a0), "=&v"(t0)
                             : "v"(keep), "v"(p));
            if (V == 3)
                asm volatile("s_nop 0\n\tv_mul_f32_dpp %2, %0, %4 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                             "v_mul_f32_dpp %3, %1, %4 wave_shr:1 row_mask:0xf bank_mask:0xf\n\tv_add_f32 %0, %2, %5\n\tv_add_f32 %1, %3, %5"
                             : "+v"(a0), "+v"(a1), "=&v"(t0), "=&v"(t1)
                             : "v"(keep), "v"(p));
            if (V == 4)
                asm volatile("s_nop 1\n\tv_mul_f32_dpp %1, %0, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_add_f32 %0, %1, %3"
                             : "+v"(a0), "=&v"(t0)
                             : "v"(keep), "v"(p));
            if (V == 5)
                asm volatile("v_mul_f32 %4, %0, %8\n\tv_mul_f32 %5, %1, %8\n\tv_mul_f32 %6, %2, %8\n\tv_mul_f32 %7, %3, %8\n\t"
                             "v_add_f32 %0, %4, %9\n\tv_add_f32 %1, %5, %9\n\tv_add_f32 %2, %6, %9\n\tv_add_f32 %3, %7, %9"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
                             : "v"(keep), "v"(p));
            if (V == 7)
                asm volatile("v_mul_f32 %1, %0, %2\n\tv_add_f32 %0, %3, %1" : "+v"(a0), "=&v"(t0) : "v"(keep), "s"(sp));
        }
    }
    const long long c1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3;
    if (threadIdx.x == 0)
        cycles[blockIdx.x] = c1 - c0;
}

template <int V>
void run(const char *name, int chains)
{
    float *out;
    long long *cyc;
    hipMalloc(&out, 64 * 64 * sizeof(float));
    hipMalloc(&cyc, 64 * sizeof(long long));
    const int iters = 4000;
    for (int grid : {1, 2}) { // 2 workgroups: does a neighbour on the chip change anything?
        hipLaunchKernelGGL(k_probe<V>, dim3(grid), dim3(64), 0, 0, out, cyc, 10, 0.999999f, 1e-6f);
        hipDeviceSynchronize();
        hipEvent_t a, b;
        hipEventCreate(&a);
        hipEventCreate(&b);
        hipEventRecord(a);
        hipLaunchKernelGGL(k_probe<V>, dim3(grid), dim3(64), 0, 0, out, cyc, iters, 0.999999f, 1e-6f);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms = 0;
        hipEventElapsedTime(&ms, a, b);
        std::vector<long long> h(grid);
        hipMemcpy(h.data(), cyc, grid * sizeof(long long), hipMemcpyDeviceToHost);
        const double steps = (double)iters * kUnroll;
        // (s_memtime ticks at the shader clock here: 12.25 ticks per step = 5.13 ns at 2.39 GHz)
        printf("%-46s grid=%d  %.2f ns per step of the wave = of all its %d chain(s) (event)  cycles per step %.3f\n", name, grid,
               ms * 1e6 / steps, chains, (double)h[0] / steps);
        hipEventDestroy(a);
        hipEventDestroy(b);
    }
    hipFree(out);
    hipFree(cyc);
}

int main()
{
    int clk = 0;
    hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    printf("device clock rate attribute: %d kHz (a lone wave leaves the chip at its top clock)\n", clk);
    run<0>("0 mul->add, one chain", 1);
    run<1>("1 mul->add, two chains interleaved", 2);
    run<2>("2 DPP wave_shr:1 mul -> add (+s_nop 1)", 1);
    run<3>("3 two DPP chains interleaved (+s_nop 0)", 2);
    run<4>("4 DPP row_shr:1 mul -> add (+s_nop 1)", 1);
    run<5>("5 four plain chains interleaved", 4);
    run<6>("6 one chain, s_setprio 3", 1);
    run<7>("7 one chain, product operand in an SGPR", 1);
    return 0;
}
