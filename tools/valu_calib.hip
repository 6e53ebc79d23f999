// tools/valu_calib.hip -- VALU streams of KNOWN length and class, one kernel per class, meant to be run
// under the SAME rocprofv3 --pmc set as bench.py (tools/calib.sh).  It is the ground truth for the
// "VALU busy" reading of the bench line: every kernel below keeps all four SIMDs of every CU issuing
// VALU instructions of one class back to back (W waves per SIMD, no memory or LDS traffic inside the
// loop), so by construction its VALU is 100 % busy, and
//     cycles per wave-instruction of class c  =  elapsed cycles * N_SIMD / wave-instructions
// is that class's issue cost.  tools/pmc_summary.py turns the per-class costs into the calibrated
// busy fraction  sum_c n_c * cost_c / (N_SIMD * elapsed cycles)  of the product kernels (n_c from the
// SQ_INSTS_VALU* counters and the static ISA mix), and checks that THESE kernels read 1.00 +- 0.03
// under the same formula -- including `mixlike`, whose instruction mix is that of the mix/decimate item.
//
// What the first run of this probe showed (MI355X, profiles/README.md "Round 3"): SQ_ACTIVE_INST_VALU is simply
// SQ_INSTS_VALU (one quad-cycle per instruction, whatever the class), and the SIMD issues TWO plain two-VGPR-operand
// fp32 instructions per quad-cycle when it can (SQ_ACTIVE_INST_VALU2 counts those quad-cycles; not with an SGPR
// operand, not FMA, not packed).  So the VALU's busy quad-cycles are SQ_INSTS_VALU - SQ_ACTIVE_INST_VALU2, and the
// elapsed cycles that belong next to them are SQ_BUSY_CYCLES / 32 SEs of the SAME pass (GRBM_GUI_ACTIVE / 8 also counts
// ~8 us of dispatch overhead outside the kernel's own timestamps: +25 % on a 34 us kernel).
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize valu_calib.hip -o valu_calib
// Run:   ./valu_calib [waves_per_simd=5] [iters=3000]   (prints one JSON line per kernel: name, wave-instructions, ms)
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

using v2f = float __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float dpp_shr1(float old, float src)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src), 0x138, 0xf, 0xf, false));
}

enum { C_FMA32 = 0, C_MUL32, C_ADD32, C_MUL32_SGPR, C_PK_MUL, C_PK_ADD, C_PK_FMA, C_FMA64, C_CVT64, C_DPP, C_MIXLIKE, C_DEMODLIKE, C_ADD32_VV, N_CLASSES };
static const char *kNames[N_CLASSES] = {"fma32", "mul32", "add32", "mul32_sgpr", "pk_mul", "pk_add", "pk_fma", "fma64", "cvt64", "dpp",
                                        "mixlike", "demodlike", "add32_vv"};
// VALU wave-instructions per loop iteration of each kernel (checked against the ISA: tools/calib.sh greps the .s)
static const int kPerIter[N_CLASSES] = {64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64};

template <int CLASS>
__global__ __launch_bounds__(64) void k_calib(float *out, int iters, float a, float b, const float *__restrict__ taps)
{
    float r = 0.f;
    if constexpr (CLASS == C_FMA32 || CLASS == C_MUL32 || CLASS == C_ADD32 || CLASS == C_MUL32_SGPR || CLASS == C_DPP || CLASS == C_ADD32_VV) {
        float v[8];
        for (int i = 0; i < 8; ++i)
            v[i] = threadIdx.x * 0.001f + i;
        float s[8]; // wave-uniform operands (SGPRs), a different one per instruction like the Hilbert taps
        for (int i = 0; i < 8; ++i)
            s[i] = taps[i];
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int rr = 0; rr < 8; ++rr)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if (CLASS == C_FMA32)
                        v[i] = __builtin_fmaf(v[i], a, b);
                    else if (CLASS == C_ADD32_VV) // both operands VGPRs: the form the SIMD issues two of per quad-cycle
                        v[i] = v[i] + v[(i + 3) & 7];
                    else if (CLASS == C_MUL32)
                        v[i] = v[i] * a;
                    else if (CLASS == C_ADD32)
                        v[i] = v[i] + a;
                    else if (CLASS == C_MUL32_SGPR)
                        v[i] = v[i] * s[(i + rr) & 7];
                    else
                        v[i] = dpp_shr1(v[i], v[(i + 1) & 7]);
                }
        }
        for (int i = 0; i < 8; ++i)
            r += v[i];
    } else if constexpr (CLASS == C_PK_MUL || CLASS == C_PK_ADD || CLASS == C_PK_FMA) {
        v2f v[8];
        const v2f A = {a, a * 1.00001f}, B = {b, b * 0.5f};
        for (int i = 0; i < 8; ++i)
            v[i] = (v2f){threadIdx.x * 0.001f + i, threadIdx.x * 0.002f - i};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int rr = 0; rr < 8; ++rr)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if (CLASS == C_PK_MUL)
                        v[i] = v[i] * A;
                    else if (CLASS == C_PK_ADD)
                        v[i] = v[i] + B;
                    else
                        v[i] = __builtin_elementwise_fma(v[i], A, B);
                }
        for (int i = 0; i < 8; ++i)
            r += v[i].x + v[i].y;
    } else if constexpr (CLASS == C_FMA64) {
        double v[8];
        const double A = a, B = b;
        for (int i = 0; i < 8; ++i)
            v[i] = threadIdx.x * 0.001 + i;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int rr = 0; rr < 8; ++rr)
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    v[i] = __builtin_fma(v[i], A, B);
        for (int i = 0; i < 8; ++i)
            r += (float)v[i];
    } else if constexpr (CLASS == C_CVT64) {
        // a double excursion the compiler cannot fold back to fp32 (the demodulation's own -- vfo.cpp:317-328 -- IS
        // folded: k_usb_demod contains no f64 instruction): cvt_f64_f32, mul_f64, add_f64, cvt_f32_f64 -- 4 per value
        float v[8];
        const double A = (double)a + 1e-9, B = (double)b + 1e-9;
        for (int i = 0; i < 8; ++i)
            v[i] = threadIdx.x * 0.001f + i;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int rr = 0; rr < 2; ++rr)
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    v[i] = (float)((double)v[i] * A + B);
        for (int i = 0; i < 8; ++i)
            r += v[i];
    } else if constexpr (CLASS == C_MIXLIKE) {
        // the static VALU mix of the mix/decimate item per 1024-sample chunk (tools/inst_mix.py: 148 pk_mul,
        // 91 pk_add, 32 pk_fma, 32 DPP moves, ~110 plain fp32 / integer of ~414), scaled to 64 per iteration:
        // 23 pk_mul, 14 pk_add, 5 pk_fma, 5 dpp, 9 plain fp32 (add / mul), 8 integer (address arithmetic)
        v2f v[8];
        float w[4];
        int q[4];
        const v2f A = {a, a * 1.00001f}, B = {b, b * 0.5f};
        for (int i = 0; i < 8; ++i)
            v[i] = (v2f){threadIdx.x * 0.001f + i, threadIdx.x * 0.002f - i};
        for (int i = 0; i < 4; ++i)
            w[i] = threadIdx.x * 0.003f + i, q[i] = threadIdx.x + i;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 23; ++i)
                v[i & 7] = v[i & 7] * A;
#pragma unroll
            for (int i = 0; i < 14; ++i)
                v[i & 7] = v[i & 7] + B;
#pragma unroll
            for (int i = 0; i < 5; ++i)
                v[i] = __builtin_elementwise_fma(v[i], A, B);
#pragma unroll
            for (int i = 0; i < 5; ++i)
                w[i & 3] = dpp_shr1(w[i & 3], w[(i + 1) & 3]);
#pragma unroll
            for (int i = 0; i < 9; ++i)
                w[i & 3] = (i & 1) ? w[i & 3] * a : w[i & 3] + b;
#pragma unroll
            for (int i = 0; i < 8; ++i)
                q[i & 3] = q[i & 3] + q[(i + 1) & 3];
        }
        for (int i = 0; i < 8; ++i)
            r += v[i].x + v[i].y;
        for (int i = 0; i < 4; ++i)
            r += w[i] + (float)q[i];
    } else { // C_DEMODLIKE: the Hilbert MAC of k_usb_demod -- v_mul_f32 by an SGPR tap + v_add_f32, 4 chains
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        float x[8];
        for (int i = 0; i < 8; ++i)
            x[i] = threadIdx.x * 0.001f + i;
        float s[16];
        for (int i = 0; i < 16; ++i)
            s[i] = taps[i];
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int t = 0; t < 7; ++t)
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    acc[c] = acc[c] + s[(2 * t + c) & 15] * x[(t + c) & 7];
#pragma unroll
            for (int i = 0; i < 8; ++i) // the window moves on (in the kernel: the next ds_read_b128)
                x[i] = x[i] + b;
        }
        r = acc[0] + acc[1] + acc[2] + acc[3];
    }
    out[blockIdx.x * 64 + threadIdx.x] = r;
}

template <int CLASS>
static void run(int waves_per_simd, int iters, float *d, const float *taps, int n_simd)
{
    const int grid = n_simd * waves_per_simd;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    // a long warm-up of the same class: the clock settles (DVFS) before the measured launch
    for (int w = 0; w < 3; ++w)
        hipLaunchKernelGGL(k_calib<CLASS>, dim3(grid), dim3(64), 0, 0, d, iters, 1.0001f, 0.5f, taps);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_calib<CLASS>, dim3(grid), dim3(64), 0, 0, d, iters, 1.0001f, 0.5f, taps);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double insts = (double)grid * iters * kPerIter[CLASS];
    printf("{\"kernel\": \"%s\", \"class\": %d, \"waves_per_simd\": %d, \"iters\": %d, \"valu_wave_insts\": %.0f, \"ms\": %.4f, "
           "\"cycles_per_inst_at_2p4GHz\": %.3f}\n",
           kNames[CLASS], CLASS, waves_per_simd, iters, insts, ms, ms * 1e-3 * 2.4e9 * n_simd / insts);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
}

int main(int argc, char **argv)
{
    const int W = argc > 1 ? atoi(argv[1]) : 5, iters = argc > 2 ? atoi(argv[2]) : 3000;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) {
        fprintf(stderr, "no HIP device\n");
        return 1;
    }
    const int n_simd = prop.multiProcessorCount * 4;
    float *d = nullptr, *taps = nullptr;
    hipMalloc(&d, (size_t)n_simd * 16 * 64 * sizeof(float));
    hipMalloc(&taps, 64 * sizeof(float));
    float h[64];
    for (int i = 0; i < 64; ++i)
        h[i] = 0.99f + 0.0001f * i;
    hipMemcpy(taps, h, sizeof h, hipMemcpyHostToDevice);
    run<C_FMA32>(W, iters, d, taps, n_simd);
    run<C_MUL32>(W, iters, d, taps, n_simd);
    run<C_ADD32>(W, iters, d, taps, n_simd);
    run<C_MUL32_SGPR>(W, iters, d, taps, n_simd);
    run<C_PK_MUL>(W, iters, d, taps, n_simd);
    run<C_PK_ADD>(W, iters, d, taps, n_simd);
    run<C_PK_FMA>(W, iters, d, taps, n_simd);
    run<C_FMA64>(W, iters, d, taps, n_simd);
    run<C_CVT64>(W, iters, d, taps, n_simd);
    run<C_DPP>(W, iters, d, taps, n_simd);
    run<C_MIXLIKE>(W, iters, d, taps, n_simd);
    run<C_DEMODLIKE>(W, iters, d, taps, n_simd);
    run<C_ADD32_VV>(W, iters, d, taps, n_simd);
    hipFree(d);
    hipFree(taps);
    return 0;
}
