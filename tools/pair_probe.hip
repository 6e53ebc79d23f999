// tools/pair_probe.hip -- when does gfx950 issue two fp32 VALU instructions in one quad-cycle?  MAC streams shaped like the
// Hilbert loop of k_usb_demod (v_mul_f32 by a tap + v_add_f32 into an accumulator) with the number of independent
// accumulator chains and the kind of tap operand (SGPR / VGPR) varied; run under
//   rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU2 SQ_BUSY_CYCLES -- ./pair_probe
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize pair_probe.hip -o pair_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int NACC, bool VTAPS, int W>
__global__ __launch_bounds__(64 * W) void k_pair(float *out, int iters, float b, const float *__restrict__ taps)
{
    float acc[NACC];
    float x[8];
    for (int i = 0; i < NACC; ++i)
        acc[i] = 0.f;
    for (int i = 0; i < 8; ++i)
        x[i] = threadIdx.x * 0.001f + i;
    float s[16];
    for (int i = 0; i < 16; ++i)
        s[i] = VTAPS ? taps[i] + threadIdx.x * 1e-9f : taps[i]; // per-lane values stay in VGPRs, uniform ones in SGPRs
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 56 / NACC; ++t)
#pragma unroll
            for (int c = 0; c < NACC; ++c)
                acc[c] = acc[c] + s[(2 * t + c) & 15] * x[(t + c) & 7];
#pragma unroll
        for (int i = 0; i < 8; ++i)
            x[i] = x[i] + b;
    }
    float r = 0.f;
    for (int i = 0; i < NACC; ++i)
        r += acc[i];
    out[blockIdx.x * 64 * W + threadIdx.x] = r;
}

template <int NACC, bool VTAPS, int W>
void run(const char *name, int waves_per_simd, int iters, float *d, const float *taps, int n_simd)
{
    const int grid = n_simd * waves_per_simd / W;
    for (int w = 0; w < 3; ++w)
        hipLaunchKernelGGL((k_pair<NACC, VTAPS, W>), dim3(grid), dim3(64 * W), 0, 0, d, iters, 0.5f, taps);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k_pair<NACC, VTAPS, W>), dim3(grid), dim3(64 * W), 0, 0, d, iters, 0.5f, taps);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double insts = (double)grid * W * iters * (2.0 * (56 / NACC) * NACC + 8);
    printf("{\"kernel\": \"%s\", \"waves_per_simd\": %d, \"valu_wave_insts\": %.0f, \"ms\": %.4f, \"cycles_per_inst_at_2p4GHz\": %.3f}\n", name,
           waves_per_simd, insts, ms, ms * 1e-3 * 2.4e9 * n_simd / insts);
}

int main(int argc, char **argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 3000;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess)
        return 1;
    const int n_simd = prop.multiProcessorCount * 4;
    float *d = nullptr, *taps = nullptr;
    (void)hipMalloc(&d, (size_t)n_simd * 16 * 64 * sizeof(float));
    (void)hipMalloc(&taps, 64 * sizeof(float));
    float h[64];
    for (int i = 0; i < 64; ++i)
        h[i] = 0.99f + 0.0001f * i;
    (void)hipMemcpy(taps, h, sizeof h, hipMemcpyHostToDevice);
    run<2, false, 1>("acc2_sgpr_w5", 5, iters, d, taps, n_simd);
    run<4, false, 1>("acc4_sgpr_w5", 5, iters, d, taps, n_simd);
    run<7, false, 1>("acc7_sgpr_w5", 5, iters, d, taps, n_simd);
    run<8, false, 1>("acc8_sgpr_w5", 5, iters, d, taps, n_simd);
    run<4, true, 1>("acc4_vgpr_w5", 5, iters, d, taps, n_simd);
    run<8, true, 1>("acc8_vgpr_w5", 5, iters, d, taps, n_simd);
    run<4, false, 1>("acc4_sgpr_w1", 1, iters, d, taps, n_simd);
    run<4, false, 1>("acc4_sgpr_w2", 2, iters, d, taps, n_simd);
    run<4, false, 1>("acc4_sgpr_w7", 7, iters, d, taps, n_simd);
    run<4, false, 1>("acc4_sgpr_w8", 8, iters, d, taps, n_simd);
    run<4, false, 4>("acc4_sgpr_w7_block256", 7, iters, d, taps, n_simd); // 256-thread blocks like k_usb_demod (grid rounds down)
    return 0;
}
