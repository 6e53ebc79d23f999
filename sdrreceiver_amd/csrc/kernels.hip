// kernels.hip -- hand-written HIP kernels of the per-VFO IQ chain for gfx950 (MI355X, CDNA4).
//
// Compiled with -ffp-contract=off: every a*b+c below is two roundings unless fmaf() is
// spelled out.  That is what makes the EXACT variants reproduce the reference's -O2 x86-64
// arithmetic bit for bit; the FAST variants use explicit fmaf() (option "exact" = 0: and the NCO table as
// rotations of its exact checkpoints, ROT; = 2, the robust arithmetic: FMAs, but the table replayed exactly).
//
// Kernel map (reference function -> kernel):
//   Oscillator::Oscillator            oscillator.cpp:4-32      -> k_nco_init
//   vfo::process mix loop             vfo.cpp:237-245          \
//   HalfBandDecimator::decimate       halfbanddecimator.cpp:43-72 } k_mix_decimate
//     FIR::...HalfBandQueue           dsp.cpp:96-173           /
//   vfo::usb_decimdemod (FIR part)    vfo.cpp:334-387          -> k_late_decimate
//   vfo::usb_demod / demod tail       vfo.cpp:300-332,350-364  -> k_usb_demod  (option fuse_demod: demod_chunk inside the d = 2 leaf's mix wave)
//   vfo::compress                     vfo.cpp:389-424          -> k_compress
//
// Execution model: 64-wide wavefronts.  k_mix_decimate runs ONE wave per workgroup and one
// workgroup per (VFO, time segment); a wave owns all the LDS it touches, so its
// __syncthreads() are wave-local (no cross-wave barrier traffic) and the segment walks the
// frame chunk by chunk with the filter state resident in LDS.
#include "sdrx_dev.h"

#include <type_traits>
#include <utility>

namespace sdrx {

// Pointers that come out of a descriptor in memory are generic ("flat") to the compiler.  They
// all point into HBM: these helpers say so, so that accesses become global_load/global_store
// (or s_load for wave-uniform read-only data) instead of flat_* instructions.
#define SDRX_AS1 __attribute__((address_space(1)))
// Descriptors, work lists and filter taps are written once at sdrx_finalize and never by a kernel:
// reading them through the CONSTANT address space tells the compiler so (no store of the kernel can
// alias them), and a wave-uniform address then becomes a scalar load into SGPRs whatever else the
// kernel does.  (Without it the 62 Hilbert taps of the demodulation turn into 62 VGPRs -- and spill --
// as soon as the same kernel also contains the mix/decimate code: k_frame.)
#define SDRX_AS4 __attribute__((address_space(4)))
template <typename T>
__device__ __forceinline__ T ldc(const T *p)
{
    return *(const SDRX_AS4 T *)p;
}
using v2f = float __attribute__((ext_vector_type(2)));
using v4f = float __attribute__((ext_vector_type(4)));
using v4s = short __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float gld(const float *p) { return *(const SDRX_AS1 float *)p; }
__device__ __forceinline__ float2 gld2(const float2 *p)
{
    const v2f v = *(const SDRX_AS1 v2f *)p;
    return make_float2(v.x, v.y);
}
__device__ __forceinline__ float4 gld4(const float4 *p)
{
    const v4f v = *(const SDRX_AS1 v4f *)p;
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void gst2(float2 *p, float2 a)
{
    v2f v = {a.x, a.y};
    *(SDRX_AS1 v2f *)p = v;
}
__device__ __forceinline__ void gst4(float4 *p, float4 a)
{
    v4f v = {a.x, a.y, a.z, a.w};
    *(SDRX_AS1 v4f *)p = v;
}

// ---- packed complex arithmetic ------------------------------------------------------------------
// Measured on MI355X (tools/valu_probe.hip): a plain fp32 VALU instruction issues every ~4.1 cycles
// per wave, a packed v_pk_{mul,add,fma}_f32 every ~4.4 -- so complex (re, im) pairs are kept as
// native 2-vectors and every operation is ONE packed instruction on both components.  Each packed
// lane is an ordinary IEEE fp32 operation, so EXACT results are unchanged.
__device__ __forceinline__ v2f xx(v2f a) { return __builtin_shufflevector(a, a, 0, 0); }
__device__ __forceinline__ v2f yy(v2f a) { return __builtin_shufflevector(a, a, 1, 1); }
__device__ __forceinline__ v2f yx(v2f a) { return __builtin_shufflevector(a, a, 1, 0); }
__device__ __forceinline__ v2f lo2(v4f a) { return __builtin_shufflevector(a, a, 0, 1); }
__device__ __forceinline__ v2f hi2(v4f a) { return __builtin_shufflevector(a, a, 2, 3); }
__device__ __forceinline__ v4f cat2(v2f a, v2f b) { return __builtin_shufflevector(a, b, 0, 1, 2, 3); }
// (a.x b.x - a.y b.y, a.x b.y + a.y b.x) with every product and the final sum/difference rounded
// separately, like libstdc++'s complex<float> operator* (vfo.cpp:241, oscillator.cpp:22): the
// combine is fma(t2, (-1,+1), t1) -- multiplying by +-1 is exact, so it is one rounding, i.e. a
// subtraction in the low lane and an addition in the high lane.
__device__ __forceinline__ v2f cmul(v2f a, v2f b)
{
    const v2f t1 = xx(a) * b;
    const v2f t2 = yy(a) * yx(b);
    const v2f pm = {-1.0f, 1.0f};
    return __builtin_elementwise_fma(t2, pm, t1);
}
// One window sample w of a FIR feeding up to three accumulator chains: a_r += (h_r, h_r) * w with h_r = the low (H_r = 0) or
// high (H_r = 1) half of a PAIR of taps -- op_sel / op_sel_hi pick that half for both lanes of the packed instruction.
// (Written as `v2f{h, h} * w` the compiler materialises every pair (h, h) with two v_mov and the registers to hold them.)
// EXACT: product and sum rounded separately, the products first and the sums behind them so that no instruction waits for the
// one before it; all of it in ONE asm statement: a scheduler cannot pull the products of later samples up front (they do
// not depend on the accumulators: it did, and spilled them), and the hazard recogniser, which puts a wait state behind every
// asm statement whose result the next instruction reads, sees one statement per sample instead of one per product.
// Each lane of each instruction is an ordinary IEEE fp32 operation.
template <bool EXACT, int H0>
__device__ __forceinline__ void fir_mac1(v2f &a0, v2f h0, v2f w)
{
    v2f t0;
    if (EXACT)
        // (gfx950 wants one wait state between a packed fp32 instruction and a reader of its result -- the compiler puts an
        // s_nop there in its own code; inside an asm statement nobody does.  With two or three chains the other products are
        // that wait state.)
        asm("v_pk_mul_f32 %1, %2, %3 op_sel:[%4,0] op_sel_hi:[%4,1]\n\ts_nop 0\n\tv_pk_add_f32 %0, %0, %1" : "+v"(a0), "=&v"(t0) : "v"(h0), "v"(w), "n"(H0));
    else
        asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[%3,0,0] op_sel_hi:[%3,1,1]" : "+v"(a0) : "v"(h0), "v"(w), "n"(H0));
}
template <bool EXACT, int H0, int H1>
__device__ __forceinline__ void fir_mac2(v2f &a0, v2f &a1, v2f h0, v2f h1, v2f w)
{
    v2f t0, t1;
    if (EXACT)
        asm("v_pk_mul_f32 %2, %4, %6 op_sel:[%7,0] op_sel_hi:[%7,1]\n\tv_pk_mul_f32 %3, %5, %6 op_sel:[%8,0] op_sel_hi:[%8,1]\n\t"
            "v_pk_add_f32 %0, %0, %2\n\tv_pk_add_f32 %1, %1, %3"
            : "+v"(a0), "+v"(a1), "=&v"(t0), "=&v"(t1)
            : "v"(h0), "v"(h1), "v"(w), "n"(H0), "n"(H1));
    else
        asm("v_pk_fma_f32 %0, %2, %4, %0 op_sel:[%5,0,0] op_sel_hi:[%5,1,1]\n\tv_pk_fma_f32 %1, %3, %4, %1 op_sel:[%6,0,0] op_sel_hi:[%6,1,1]"
            : "+v"(a0), "+v"(a1)
            : "v"(h0), "v"(h1), "v"(w), "n"(H0), "n"(H1));
}
template <bool EXACT, int H0, int H1, int H2>
__device__ __forceinline__ void fir_mac3(v2f &a0, v2f &a1, v2f &a2, v2f h0, v2f h1, v2f h2, v2f w)
{
    v2f t0, t1, t2;
    if (EXACT)
        asm("v_pk_mul_f32 %3, %6, %9 op_sel:[%10,0] op_sel_hi:[%10,1]\n\tv_pk_mul_f32 %4, %7, %9 op_sel:[%11,0] op_sel_hi:[%11,1]\n\t"
            "v_pk_mul_f32 %5, %8, %9 op_sel:[%12,0] op_sel_hi:[%12,1]\n\t"
            "v_pk_add_f32 %0, %0, %3\n\tv_pk_add_f32 %1, %1, %4\n\tv_pk_add_f32 %2, %2, %5"
            : "+v"(a0), "+v"(a1), "+v"(a2), "=&v"(t0), "=&v"(t1), "=&v"(t2)
            : "v"(h0), "v"(h1), "v"(h2), "v"(w), "n"(H0), "n"(H1), "n"(H2));
    else
        asm("v_pk_fma_f32 %0, %3, %6, %0 op_sel:[%7,0,0] op_sel_hi:[%7,1,1]\n\tv_pk_fma_f32 %1, %4, %6, %1 op_sel:[%8,0,0] op_sel_hi:[%8,1,1]\n\t"
            "v_pk_fma_f32 %2, %5, %6, %2 op_sel:[%9,0,0] op_sel_hi:[%9,1,1]"
            : "+v"(a0), "+v"(a1), "+v"(a2)
            : "v"(h0), "v"(h1), "v"(h2), "v"(w), "n"(H0), "n"(H1), "n"(H2));
}
// f(std::integral_constant<int, 0>{}), ..., f(integral_constant<int, N - 1>{}): a loop whose index is a constant EXPRESSION
// inside the body (the op_sel digits above are template arguments)
template <typename F, int... I>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, I...>)
{
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F &&f)
{
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}
__device__ __forceinline__ v2f gldv2(const float2 *p) { return *(const SDRX_AS1 v2f *)p; }
__device__ __forceinline__ v4f gldv4(const float4 *p) { return *(const SDRX_AS1 v4f *)p; }
typedef unsigned v4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v4u gldv4u(const v4u *p) { return *(const SDRX_AS1 v4u *)p; }
__device__ __forceinline__ void gstv2(float2 *p, v2f v) { *(SDRX_AS1 v2f *)p = v; }
// Experiment switch -DSDRX_NT=1 (profiles/README.md, round 3): the leaf streams -- written by the mix/decimate launch,
// read exactly once by the demodulation ~40 us later -- with the non-temporal hint on both sides.
#ifndef SDRX_NT
#define SDRX_NT 0
#endif
__device__ __forceinline__ void gstv2_leaf(float2 *p, v2f v)
{
#if SDRX_NT
    __builtin_nontemporal_store(v, (SDRX_AS1 v2f *)p);
#else
    *(SDRX_AS1 v2f *)p = v;
#endif
}
__device__ __forceinline__ void gstv4_leaf(float4 *p, v4f v)
{
#if SDRX_NT
    __builtin_nontemporal_store(v, (SDRX_AS1 v4f *)p);
#else
    *(SDRX_AS1 v4f *)p = v;
#endif
}
__device__ __forceinline__ float2 gld2_once(const float2 *p)
{
#if SDRX_NT
    const v2f v = __builtin_nontemporal_load((const SDRX_AS1 v2f *)p);
#else
    const v2f v = *(const SDRX_AS1 v2f *)p;
#endif
    return make_float2(v.x, v.y);
}
__device__ __forceinline__ void gstv4(float4 *p, v4f v) { *(SDRX_AS1 v4f *)p = v; }
// Stores that are ALWAYS issued, lanes that have nothing to store included: a buffer store whose offset lies beyond the
// resource's num_records is dropped by the hardware, so "this lane does not store" is an offset, not a branch -- and a
// store the compiler can see on every path is one it can COUNT: the s_waitcnt for loads issued in front of it then leaves
// it outstanding (vmcnt(n) with the stores among the n) instead of waiting for its acknowledgement.
typedef int v2i __attribute__((ext_vector_type(2)));
typedef int v4i __attribute__((ext_vector_type(4)));
constexpr unsigned kNoStore = 0x7fffffffu; // a byte offset beyond every stream: the lane's store is dropped
__device__ __forceinline__ __amdgpu_buffer_rsrc_t stream_rsrc(float2 *base)
{
    return __builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7ffffffe, 0x00020000); // raw buffer, no stride / swizzle, 32-bit data format
}
__device__ __forceinline__ void bst2(__amdgpu_buffer_rsrc_t r, unsigned byte_off, v2f v)
{
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2i, v), r, (int)byte_off, 0, 0);
}
__device__ __forceinline__ void bst4(__amdgpu_buffer_rsrc_t r, unsigned byte_off, v4f v)
{
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i, v), r, (int)byte_off, 0, 0);
}

// ------------------------------------------------------------------------------------ NCO
// One step of the reference's table recurrence (oscillator.cpp:20-28): v *= rot (complex
// product re = ac - bd, im = ad + bc), then v *= 1.95f - |v|^2.  Strict fp32, no FMA.
__device__ __forceinline__ float2 nco_step(float2 v, float rc, float rs)
{
    float nr = v.x * rc - v.y * rs;
    float ni = v.x * rs + v.y * rc;
    float norm = 1.95f - (nr * nr + ni * ni);
    return make_float2(nr * norm, ni * norm);
}

// The same step on a packed (re, im) pair: 7 instructions instead of 12.
__device__ __forceinline__ v2f nco_step_pk(v2f v, v2f rot)
{
    const v2f n = cmul(v, rot);          // nr = v.x rc - v.y rs, ni = v.x rs + v.y rc
    const v2f sq = n * n;
    const float norm = 1.95f - (sq.x + sq.y);
    return n * norm;
}

// NCO in the tolerance arithmetic (option exact = 0; north_star: "within 1e-5 relative float tolerance").  Once the table's
// amplitude has settled (the stabiliser 1.95f - |v|^2 pulls |v|^2 to 0.95 with the factor -0.9 per step: ~150 entries,
// oscillator.cpp:20-28) a step of the recurrence is a rotation by arg(rot) at constant modulus, plus its own rounding noise.
// So from the EXACT checkpoint c = table[16 j - 1] the 16 entries behind it are rotations of it:  with rk[q] = u^(q+1),
// q = 0..3, computed in double at finalize,  table[b + q] = base * rk[q]  for base = c, then table[b + 3] of the block
// before (four blocks of four; an entry is at most four products away from the checkpoint) -- two-instruction products
// with four wave-uniform constants instead of a 16-deep chain of seven instructions.  Measured against the reference's
// tables (the (Fs, f) pairs of the shipped profiles): max 1.0e-6, rms 2e-7 relative on entries >= 512; a checkpoint is
// never more than 16 entries away, so nothing accumulates over a frame or a run.  Chunks that touch the first kNcoSettle
// entries of the table (start-up ringing; the very first sample's table[L-1]) replay it exactly.
//
// Two samples per statement:  m_k = base * r_k  (the table entries; r_k wave-uniform, an SGPR pair),  x_k = m_k * x_k  (the
// mixer, vfo.cpp:241).  A complex product (a.x b.x - a.y b.y, a.x b.y + a.y b.x) is ONE packed multiply (a.x, a.x) * b and
// ONE packed FMA (a.y, a.y) * (-b.y, b.x) + that: the broadcast of a's halves, the swap of b's and the sign are source
// modifiers (op_sel, op_sel_hi, neg_lo) -- written out because the compiler finds them for one product in eight and
// builds (-b.y, b.x) with a v_xor and a v_mov for the others.  The two samples' instructions alternate, so none reads the
// result of the one before it (gfx950 wants one wait state behind a packed fp32 instruction; nobody inserts it inside an
// asm statement).  Returns m_1 (the later entry: the next block's base).
__device__ __forceinline__ v2f nco_mix_fast2(v2f base, v2f r0, v2f r1, v2f &x0, v2f &x1)
{
    v2f m0, m1, y0, y1; // (results in registers of their own: written in place, the compiler copies the inputs first)
    asm("v_pk_mul_f32 %2, %6, %7 op_sel_hi:[0,1]\n\t"
        "v_pk_mul_f32 %3, %6, %8 op_sel_hi:[0,1]\n\t"
        "v_pk_fma_f32 %2, %6, %7, %2 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]\n\t"
        "v_pk_fma_f32 %3, %6, %8, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]\n\t"
        "v_pk_mul_f32 %0, %2, %4 op_sel_hi:[0,1]\n\t"
        "v_pk_mul_f32 %1, %3, %5 op_sel_hi:[0,1]\n\t"
        "v_pk_fma_f32 %0, %2, %4, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]\n\t"
        "v_pk_fma_f32 %1, %3, %5, %1 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]"
        : "=&v"(y0), "=&v"(y1), "=&v"(m0), "=&v"(m1)
        : "v"(x0), "v"(x1), "v"(base), "s"(r0), "s"(r1));
    x0 = y0, x1 = y1;
    return m1;
}
// The mixer alone in that form, two samples per statement:  x_k = m_k * x_k  (vfo.cpp:241) as one packed multiply and one packed
// FMA each, alternating (the ROBUST arithmetic: table entries m_k from the exact recurrence, everything after them as FMAs).
__device__ __forceinline__ void cmul_fast2(v2f m0, v2f m1, v2f &x0, v2f &x1)
{
    v2f y0, y1;
    asm("v_pk_mul_f32 %0, %2, %4 op_sel_hi:[0,1]\n\t"
        "v_pk_mul_f32 %1, %3, %5 op_sel_hi:[0,1]\n\t"
        "v_pk_fma_f32 %0, %2, %4, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]\n\t"
        "v_pk_fma_f32 %1, %3, %5, %1 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]"
        : "=&v"(y0), "=&v"(y1)
        : "v"(m0), "v"(m1), "v"(x0), "v"(x1));
    x0 = y0, x1 = y1;
}
// the 16 entries behind checkpoint `c` and the mixer on the lane's 16 samples
__device__ __forceinline__ void nco_mix_fast16(v2f c, const float2 *__restrict__ rk, v2f *x)
{
    const v2f r0 = {ldc(&rk[0].x), ldc(&rk[0].y)}, r1 = {ldc(&rk[1].x), ldc(&rk[1].y)};
    const v2f r2 = {ldc(&rk[2].x), ldc(&rk[2].y)}, r3 = {ldc(&rk[3].x), ldc(&rk[3].y)};
#pragma unroll
    for (int b = 0; b < kRun; b += 4) {
        nco_mix_fast2(c, r0, r1, x[b], x[b + 1]);
        c = nco_mix_fast2(c, r2, r3, x[b + 2], x[b + 3]);
    }
}

// Replays the whole table once per VFO and keeps every 16th entry: cp[j] = table[16j-1]
// (cp[0] = the initial (1,0)), so any aligned run of 16 entries can be regenerated in
// registers, bit-exact by construction.  One thread per VFO; init-time only.
__global__ void k_nco_init(const NcoInit *__restrict__ jobs, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    NcoInit J = jobs[i];
    float2 v = make_float2(1.0f, 0.0f);
    J.cp[0] = v;
    for (int k = 0; k < J.L; k += kRun) {
#pragma unroll
        for (int t = 0; t < kRun; ++t)
            v = nco_step(v, J.rot_re, J.rot_im);
        J.cp[(k >> 4) + 1] = v;
    }
}

// Debug/parity helper: regenerate table[first .. first+count) from the checkpoints.
__global__ void k_nco_dump(const float2 *__restrict__ cp, float rc, float rs, long first, long count,
                           float2 *__restrict__ out)
{
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count)
        return;
    long idx = first + i;
    long j = idx >> 4;
    float2 v = cp[j];
    for (long t = j << 4; t <= idx; ++t)
        v = nco_step(v, rc, rs);
    out[i] = v;
}

// ------------------------------------------------------------------------------------ half-band
// hbcoeff11, halfbanddecimator.h:66-79 (only the 11-tap case is ever instantiated, vfo.cpp:130).
#define HB0 0.0060431029837374152f
#define HB2 (-0.049372515458761493f)
#define HB4 0.29332944952052842f
#define HB5 0.5f

// FIR::FIRUpdateAndProcessHalfBandQueue case 11 (dsp.cpp:137-143): symmetric-pair form,
// evaluated left to right, then `0 + ...`.
template <bool EXACT>
__device__ __forceinline__ float hb_dot(float w0, float w2, float w4, float w5, float w6, float w8, float w10)
{
    if (EXACT) {
        float s = HB0 * (w0 + w10) + HB2 * (w2 + w8) + HB4 * (w4 + w6) + HB5 * w5;
        return 0.0f + s;
    } else {
        return fmaf(HB0, w0 + w10, fmaf(HB2, w2 + w8, fmaf(HB4, w4 + w6, HB5 * w5)));
    }
}

// the same on a packed complex sample (both components in one instruction each)
template <bool EXACT>
__device__ __forceinline__ v2f hb_dot2(v2f w0, v2f w2, v2f w4, v2f w5, v2f w6, v2f w8, v2f w10)
{
    if (EXACT) {
        const v2f s = HB0 * (w0 + w10) + HB2 * (w2 + w8) + HB4 * (w4 + w6) + HB5 * w5;
        return 0.0f + s;
    } else {
        const v2f h0 = {HB0, HB0}, h2 = {HB2, HB2}, h4 = {HB4, HB4};
        return __builtin_elementwise_fma(h0, w0 + w10,
                                         __builtin_elementwise_fma(h2, w2 + w8, __builtin_elementwise_fma(h4, w4 + w6, HB5 * w5)));
    }
}

// ------------------------------------------------------------------------------------ streams
// TILE LAYOUT.  Every cf32 stream that feeds k_mix_decimate is stored in tiles of 1024 samples:
// tile c is 512 float4 units [i2 = 0..7][lane = 0..63]; unit (i2, lane) holds samples
// c*1024 + lane*16 + 2*i2 and +1.  A wave then loads its chunk as 8 fully coalesced 1 KiB
// instructions and every lane ends up with 16 CONSECUTIVE samples in registers -- the shape both
// the NCO replay and the register-resident half-band stages want -- with no LDS transpose.
__device__ __forceinline__ size_t tile_unit(int chunk, int i2, int lane) { return (size_t)chunk * 512 + i2 * 64 + lane; }
// float2 index of natural sample g inside a tile-layout stream
__device__ __forceinline__ size_t tile_pos(int g)
{
    const int c = g >> 10, r = g & 1023, ln = r >> 4, i = r & 15;
    return (size_t)c * 1024 + (i >> 1) * 128 + ln * 2 + (i & 1);
}

// 16-byte units from one buffer to another, either of which may be pinned HOST memory reached over PCIe: the payloads' way
// out (sdrx.hip, enqueue_frame).  Grid-stride with four units per thread in flight; the launch decides how much of the chip it
// may occupy -- a PCIe-bound copy needs bytes in flight, not CUs.
__global__ __launch_bounds__(256) void k_copy16(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n16)
{
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        const uint4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
        dst[i] = a, dst[i + stride] = b, dst[i + 2 * stride] = c, dst[i + 3 * stride] = d;
    }
    for (; i < n16; i += stride)
        dst[i] = src[i];
}

// natural cf32 frame -> tile layout (host-fed / broadcast raw frames enter the pipeline here)
__global__ __launch_bounds__(256) void k_ingest_f32(const float4 *__restrict__ nat, float4 *__restrict__ tiled, int n_pairs)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n_pairs)
        return;
    const int c = p >> 9, r = p & 511;
    tiled[tile_unit(c, r & 7, r >> 3)] = nat[p];
}

// Dongle bytes -> cf32 in tile layout: floats[b] = b - 127 (jonti/sdr.cpp:43-49,122-129;
// sdrj.cpp:155-160).  One thread = one complex pair (4 bytes).
__global__ __launch_bounds__(256) void k_ingest_u8(const unsigned *__restrict__ bytes4, float4 *__restrict__ tiled, int n_pairs)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n_pairs)
        return;
    const unsigned w = bytes4[p];
    const float4 v = make_float4((float)((int)(w & 255u) - 127), (float)((int)((w >> 8) & 255u) - 127),
                                 (float)((int)((w >> 16) & 255u) - 127), (float)((int)(w >> 24) - 127));
    const int c = p >> 9, r = p & 511;
    tiled[tile_unit(c, r & 7, r >> 3)] = v;
}

// The same with the DC-bias removal of sdrj::demodData (sdrj.cpp:271-286):
//   avept = avept*(1.0f-0.000001f) + 0.000001f*curr;  curr -= avept        (per component, fp32)
// with an accumulator that lives for the whole process (function-static there; `state` here).
// It is a true first-order recurrence in ROUNDED fp32 arithmetic -- fl(fl(avept*keep) + fl(k*curr)) -- so the
// bit-exact form is sequential in the frame.  What IS sequential is two dependent VALU operations per sample and
// component, and on this machine a dependent v_mul_f32 -> v_add_f32 pair of a lone wave takes 12.25 cycles whatever
// else is or is not going on (tools/dc_chain_probe.hip: one chain per wave 12.25 cycles per step; I and Q interleaved
// in one wave 20.25 for the two; a DPP systolic form 20.25; a scalar operand costs nothing) -- while every OTHER
// instruction the chain wave issues costs its ~4.5 cycles on top (the round-3 kernel's LDS traffic and register copies:
// 24.6 cycles per sample).  So the chain gets a kernel in which it is nearly alone:
//   k_dc_products   (parallel)  P[c][t] = fl(k * curr)                       all samples, both components
//   k_dc_chain      (2 waves)   A[c][j] = avept before sample 16 j           wave c = component c, on a CU of its own; per 32
//                                                                            samples two s_load_dwordx16 (the products arrive as
//                                                                            scalar operands, a group ahead), the 64 chain
//                                                                            operations, ONE 8-byte store
//   k_dc_apply      (parallel)  replays the 16 steps behind each A[c][j], curr - avept, tile layout
// Measured: 2.03 ms per 384 000-sample frame (4.5 ms for the one-workgroup pipeline of round 3, 9.3 ms for the one-wave
// version of rounds 1-2); the floor of the recurrence itself is 384 000 x 12.25 cycles = 1.96 ms at 2.4 GHz.
typedef float v16f __attribute__((ext_vector_type(16)));
constexpr int kDcPad = 32 * 16 + 128; // floats behind the last sample of P / A: the chain's scalar prefetch runs two groups ahead, its line prefetch kDcAhead
__global__ __launch_bounds__(256) void k_dc_products(const unsigned *__restrict__ bytes4, float *__restrict__ P, int n_complex, int stride)
{
    const int w = blockIdx.x * 256 + threadIdx.x; // one word = 2 complex samples
    if (2 * w >= n_complex)
        return;
    const float k = 0.000001f;
    const unsigned u = bytes4[w];
    const v2f pi = {k * (float)((int)(u & 255u) - 127), k * (float)((int)((u >> 16) & 255u) - 127)};
    const v2f pq = {k * (float)((int)((u >> 8) & 255u) - 127), k * (float)((int)(u >> 24) - 127)};
    *(SDRX_AS1 v2f *)(P + 2 * w) = pi;
    *(SDRX_AS1 v2f *)(P + stride + 2 * w) = pq;
}

// 32 products as scalar operands: requested here, ARRIVED only behind dc_products_wait() (scalar loads return out of order:
// nothing short of lgkmcnt(0) orders them).  The values the chain uses are the OUTPUTS of the wait, so no use of them can
// be scheduled above it.  (`before`: a value whose further uses must come AFTER the request -- the accumulator: the chain
// then cannot be scheduled above the loads it is meant to cover.)
// The products were written by other CUs: a scalar load of them misses all the way to the Infinity Cache (~500 cycles, more
// than the 392 cycles of chain a group of 32 samples covers).  So one VECTOR load per group touches the line kDcAhead groups
// further on -- its result is never used and never waited for (another counter: vmcnt) -- and the scalar loads find their
// lines in this XCD's L2.
constexpr int kDcAhead = 16; // groups of 32 samples = 2 KiB
// (`junk`: the register the line prefetches land in, some time after their issue and unknown to the compiler: one register,
// handed through every request, so that it is never anything else's.)
__device__ __forceinline__ void dc_products_request(v16f &lo, v16f &hi, const float *p, float &before, float &junk)
{
    asm volatile("s_load_dwordx16 %0, %4, 0x0\n\ts_load_dwordx16 %1, %4, 0x40\n\tglobal_load_dword %2, %5, off"
                 : "=s"(lo), "=s"(hi), "+v"(junk), "+v"(before)
                 : "s"(p), "v"(p + 32 * kDcAhead));
}
__device__ __forceinline__ void dc_products_wait(v16f &lo, v16f &hi) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(lo), "+s"(hi)); }

__device__ __forceinline__ float dc_block16(float acc, const v16f p)
{
    const float keep = 1.0f - 0.000001f;
#pragma unroll
    for (int i = 0; i < 16; ++i)
        acc = acc * keep + p[i]; // -ffp-contract=off: v_mul_f32, v_add_f32 -- two roundings, like the reference's -O2 build
    return acc;
}

// A[c][j] = avept BEFORE sample 16 j of component c: one value per 16 samples is all the chain wave stores (every store is
// an instruction it issues on top of its chain); k_dc_apply replays the 16 steps behind each of them -- the same operations
// on the same values, so the same bits -- in parallel.
__global__ __launch_bounds__(64) void k_dc_chain(const float *__restrict__ P, float *__restrict__ A, int n_complex, int stride,
                                                 float *__restrict__ state)
{
    if (threadIdx.x != 0) // the chain is one value: one lane computes and stores it (exec is set once, here)
        return;
    const int c = blockIdx.x; // component
    const float *p = P + (size_t)c * stride;
    float2 *a = reinterpret_cast<float2 *>(A + (size_t)c * (stride >> 4)); // (start of block 2 g, start of block 2 g + 1)
    float acc = state[c];
    const int ngrp = (n_complex + 31) >> 5; // groups of 32 samples (the last one may reach 16 samples into the padding behind the frame)
    v16f c0, c1, n0, n1;
    float junk = 0.f;
    dc_products_request(c0, c1, p, acc, junk);
    int g = 0;
    for (; g + 1 < ngrp; g += 2) {
        dc_products_wait(c0, c1);
        dc_products_request(n0, n1, p + 32 * (g + 1), acc, junk);
        float s0 = acc;
        acc = dc_block16(acc, c0);
        gst2(a + g, make_float2(s0, acc));
        acc = dc_block16(acc, c1);
        dc_products_wait(n0, n1);
        dc_products_request(c0, c1, p + 32 * (g + 2), acc, junk); // (past the frame's end: the padding behind it)
        s0 = acc;
        acc = dc_block16(acc, n0);
        gst2(a + g + 1, make_float2(s0, acc));
        if (32 * (g + 1) + 16 < n_complex)
            acc = dc_block16(acc, n1);
    }
    dc_products_wait(c0, c1);
    if (g < ngrp) {
        const float s0 = acc;
        acc = dc_block16(acc, c0);
        gst2(a + g, make_float2(s0, acc));
        if (32 * g + 16 < n_complex)
            acc = dc_block16(acc, c1);
    }
    state[c] = acc;
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(junk)); // the last line prefetches have landed: the register is free again
}

// The SAME recurrence, bit for bit, a block of 1024 samples at a time in parallel: k_dc_chain_spec.
//
// What makes fl(fl(a keep) + p) sequential is its two roundings.  Write a = s m 2^e with the 24-bit mantissa 2^23 <= m < 2^24
// (ulp = 2^e) and keep = 1 - 17 2^-24:
//   fl(a keep) = s (m - r(m)) 2^e,   r(m) = floor(17 m / 2^24 + 1/2)     the product m (2^24 - 17) 2^(e-24) rounded to 24 bits; it
//                                                                        stays in the binade for m >= 2^23 + 9 and is a tie only
//                                                                        for m = 2^23.  r is one of 9 .. 17 and constant over
//                                                                        runs of ~987 000 consecutive m; T = where it steps
//   fl(u + p)  = s (m_u + q) 2^e,    q = s RN(p / 2^e)                   u and the result multiples of the same ulp as long as the
//                                                                        sum stays in the binade; which way an exact tie of
//                                                                        p / 2^e goes depends on the parity of m_u
// So WHILE the binade and the sign stay what they are at the block's first sample and no p is an exact tie, the recurrence is
// one in INTEGERS:  m' = m + q - r_lo - [m >= T]  around the threshold T of r nearest to the block's first m (r = r_lo below
// it, r_lo + 1 from it on).  q comes out of ONE float addition that does the rounding for us -- v = fl(C + p) with C =
// s 1.5 2^23 2^e has a's ulp, so the mantissa bits of v are those of C plus q (C's mantissa is even: its ties go where an even
// m_u's would, and ties are excluded anyway) --, the tie test out of the exact residual p - (v - C).
// (The reference's estimate LIVES at such a threshold: below T the decrement r_lo is smaller than the mean of q, above it
// r_lo + 1 is larger, so m climbs to T and then hovers around it -- 1.2353 for an offset of 1.3 LSB, not 1.3.)
// One workgroup per component walks the frame, NW blocks per step, one wave each; lane l owns samples 16 l .. 16 l + 15 of its
// wave's block.  Every lane runs the integer recurrence for its 16 samples from a SPECULATED start value; a step stands once
// every lane -- of every wave -- started where its predecessor ended (dc_spec_blocks below: how the start values are found).
// Lane 0 of wave 0 always starts from the true value, lane l does if all before it did: then the 16 steps of every lane are the
// recurrence itself.  Such a step is then VERIFIED: every m it visited, INCLUDING the last, inside [start of r_lo's run, end of
// (r_lo + 1)'s run) and >= 2^23 + 32 (nothing may touch the binade's lower end, where fl(a keep) falls into the finer grid
// below), no exact tie.  A step that does not stand within kDcMaxIter rounds or fails the verification -- the binade or the
// sign changes inside it, a tie, |a| < 2^-9 (start-up: p is no longer small against a) -- is taken again block by block, each
// with its own context, and a block that fails on its own is redone with the rounded float operations themselves, one dependent
// mul + add pair per sample: k_dc_chain's arithmetic, 1024 steps.
// Verified blocks are bit-exact BY CONSTRUCTION, the others by definition (sdrx_stats.dc_blocks / dc_retried_blocks /
// dc_fallback_blocks say how many there were of each).  Output as k_dc_chain's: A[c][j] = avept before sample 16 j, for k_dc_apply.
constexpr int kDcBlock = 64 * kRun; // samples per block: a lane's run is one stored estimate's 16 samples
constexpr int kDcMaxIter = 24;      // rounds of "every lane from its speculated start value" before a step is given up (a quiet front end takes up to ~16)
__device__ __forceinline__ int wave_inclusive_scan(int x)
{
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true); // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true); // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, true); // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, true); // row_shr:8
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false); // row_bcast:15 -> rows 1, 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false); // row_bcast:31 -> rows 2, 3
    return x;
}
// NW waves, NW consecutive blocks side by side: what one wave does for its block, one 16-byte record per wave and round exchanged
// through LDS (two sets alternating: one barrier per round).  Every wave derives the SAME context (binade, sign, T, r_lo) from the
// accumulator before the first of the NW blocks.
// Finding the start values.  A lane's total is a function of its start value: tot(x + d) - tot(x) is between -d and 0 (every step
// merges the states -1 and 0: two trajectories only ever come closer).  Each round EVALUATES every lane's run at its current start
// value (the integer recurrence itself) and looks at the gaps  g_l = start_l + tot_l - start_(l+1)  between a run's end and the
// start the next lane assumed: all gaps zero = every lane started where its predecessor ended = the recurrence itself.
// Otherwise the starts move by the solution u of  u_(l+1) = (1 + s_l) u_l + g_l,  u_0 = 0  (a scan of affine maps: DPP inside a
// wave, the waves' composed maps through LDS), s_l = the lane's secant slope out of its last two evaluations; in the first
// round, when there is only one, -min(1, 16 / range) for a run that saw both sides of T (its 16 merge points are spread over
// about the range it covered) and 0 otherwise.  With all slopes 0 the step is "every start := the sum of the totals before
// it", and that iteration needs as many rounds as there are lanes whose runs come across T one after the other; an estimate
// hovering at T with small steps (a quiet front end) or pinned to it (an offset many times the noise) has slopes near -1 --
// every run forgets where it started -- and the plain sums overshoot for dozens of rounds, where the affine step lands next to
// the answer (tools/dc_iteration_model.py: the iteration as a numpy model, rounds by noise level and waves per step).
// Nothing here has to be exact except the evaluation: a block is only accepted in a round that found every gap zero, and then
// VERIFIED (that round's runs all stayed inside the context, no tie).  NW = 1: no LDS, no barrier.
// In: `acc` (uniform over the workgroup), the lane's 16 products `p`, nv = lanes of this wave that hold samples, last_wave = the
// last wave that holds any.  Out: `start` = avept before the lane's first sample, `acc_out` = avept after the last block;
// false = nothing to be used.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f32(float old, float src)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src), CTRL, ROW_MASK, 0xf, false));
}
// Inclusive scan over the wave's lanes of the maps x -> A x + B (lane l: its own map after those of the lanes before it).  One
// step = two instructions with the DPP operand in place: B += A * B[l - n] (lanes without a source read 0), A *= A[l - n] (lanes
// without a source are disabled: A stays) -- and the two wait states a DPP read of a just-written VGPR needs.
#define SDRX_AFFINE_STEP(ctrl, fill) "v_fmac_f32_dpp %1, %1, %0 " ctrl " bank_mask:0xf bound_ctrl:1\n\tv_mul_f32_dpp %0, %0, %0 " ctrl " bank_mask:0xf\n\t" fill
__device__ __forceinline__ void affine_scan_wave(float &A, float &B)
{
    asm volatile("s_nop 1\n\t"
                 SDRX_AFFINE_STEP("row_shr:1 row_mask:0xf", "s_nop 0\n\t")
                 SDRX_AFFINE_STEP("row_shr:2 row_mask:0xf", "s_nop 0\n\t")
                 SDRX_AFFINE_STEP("row_shr:4 row_mask:0xf", "s_nop 0\n\t")
                 SDRX_AFFINE_STEP("row_shr:8 row_mask:0xf", "s_nop 0\n\t")
                 SDRX_AFFINE_STEP("row_bcast:15 row_mask:0xa", "s_nop 0\n\t")
                 SDRX_AFFINE_STEP("row_bcast:31 row_mask:0xc", "s_nop 1")
                 : "+v"(A), "+v"(B));
}
__device__ __forceinline__ void affine_scan_lanes8(float &A, float &B, int n) // the same over the first n <= 8 lanes
{
    if (n > 4)
        asm volatile("s_nop 1\n\t" SDRX_AFFINE_STEP("row_shr:1 row_mask:0xf", "s_nop 0\n\t") SDRX_AFFINE_STEP("row_shr:2 row_mask:0xf", "s_nop 0\n\t")
                         SDRX_AFFINE_STEP("row_shr:4 row_mask:0xf", "s_nop 1")
                     : "+v"(A), "+v"(B));
    else if (n > 2)
        asm volatile("s_nop 1\n\t" SDRX_AFFINE_STEP("row_shr:1 row_mask:0xf", "s_nop 0\n\t") SDRX_AFFINE_STEP("row_shr:2 row_mask:0xf", "s_nop 1") : "+v"(A), "+v"(B));
    else
        asm volatile("s_nop 1\n\t" SDRX_AFFINE_STEP("row_shr:1 row_mask:0xf", "s_nop 1") : "+v"(A), "+v"(B));
}
template <int NW, typename AfterInputs>
__device__ __forceinline__ bool dc_spec_blocks(float acc, const float *p, int nv, int lane, int wave, int last_wave, int4 *xch, int &par, int rounds,
                                               float &start, float &acc_out, AfterInputs &&after_inputs)
{
    const unsigned bits = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, acc));
    const unsigned ex = (bits >> 23) & 255u, sign = bits & 0x80000000u;
    const int m0 = (int)((bits & 0x7fffffu) | 0x800000u);
    // r(m) = r for  r 2^24 <= 17 m + 2^23 < (r + 1) 2^24, i.e. from run_start(r) up to run_start(r + 1)
    auto run_start = [](int r) { return (int)((((unsigned)r << 24) - (1u << 23) + 16u) / 17u); };
    const int r0 = (int)((17u * (unsigned)m0 + (1u << 23)) >> 24);
    const int lo0 = run_start(r0), hi0 = run_start(r0 + 1);
    const bool near_lo = m0 - lo0 < hi0 - m0;
    const int T = near_lo ? lo0 : hi0, rlo = near_lo ? r0 - 1 : r0;
    const int lo = max(run_start(rlo), (1 << 23) + 32), hi = min(run_start(rlo + 2), 1 << 24);
    const bool ctx_ok = ex >= 118u && ex < 255u && m0 >= lo && m0 < hi; // |a| >= 2^-9: |p| / ulp < 2^21
    if (!ctx_ok) // (the same decision in every wave: they hold the same accumulator)
        return false;
    const unsigned cb = sign | (ex << 23) | 0x400000u;
    const float C = __builtin_bit_cast(float, cb);                                   // s 1.5 2^23 ulp
    const float half_ulp = __builtin_bit_cast(float, (ex >= 25u ? ex - 24u : 1u) << 23);
    // bits(v) - bits(C) = the change of the MAGNITUDE's mantissa (for a negative accumulator -RN(p / ulp), as it must be);
    // e[i] = that - r_lo - 1, so that a step is  z' = z + e[i] + [z < 0]  for z = m - T
    const int ebase = (int)cb + rlo + 1;
    float worst = 0.f; // max |residual|: never above half an ulp, equal to it = an exact tie
    int e[kRun], tot = 0;
#pragma unroll
    for (int i = 0; i < kRun; i += 2) {
        const v2f pp = {p[i], p[i + 1]}, CC = {C, C};
        const v2f v = CC + pp;           // (v_pk_add_f32: two samples per instruction)
        const v2f resid = pp - (v - CC); // exact
        const float v0 = v.x, v1 = v.y;  // (not __builtin_bit_cast(int, v.y): this compiler takes element 0 for it)
        worst = fmaxf(worst, fmaxf(fabsf(resid.x), fabsf(resid.y)));
        e[i] = __builtin_bit_cast(int, v0) - ebase;
        e[i + 1] = __builtin_bit_cast(int, v1) - ebase;
        tot += e[i] + e[i + 1];
    }
    asm volatile("" : "+v"(tot) : : "memory"); // (the products have arrived and are consumed)
    after_inputs();
    const bool tie = worst == half_ulp;
    const bool mine = lane < nv;
    const int S = m0 - T;
    // a lane behind this one takes over where this one ends (in this wave, or lane 0 of the next)
    const bool linked = lane + 1 < nv || (NW > 1 && lane == 63 && wave < last_wave);
    // the waves' records of a round -> what the waves before this one make of u = 0, the same for the next wave, the flags of all
    // (1: a gap somewhere, 2: a run left the context) and the last block's end
    auto exchange = [&](float A, float B, int flags, int end, float &u_in, float &u_next, int &end_last) {
        if constexpr (NW == 1) {
            u_in = 0.f, u_next = B, end_last = end;
            return flags;
        } else {
            if (lane == 0)
                xch[par * NW + wave] = make_int4(__builtin_bit_cast(int, A), __builtin_bit_cast(int, B), flags, end);
            __syncthreads();
            // lane w < NW takes wave w's record; the maps composed over those lanes (three DPP steps), the flags by ballot
            const int4 r = xch[par * NW + (lane < NW ? lane : NW - 1)];
            const bool rec = lane < NW;
            float Ar = rec ? __builtin_bit_cast(float, r.x) : 1.0f, Br = rec ? __builtin_bit_cast(float, r.y) : 0.0f;
            affine_scan_lanes8(Ar, Br, NW);
            const int Bi = __builtin_bit_cast(int, Br);
            u_in = wave > 0 ? __builtin_bit_cast(float, __builtin_amdgcn_readlane(Bi, wave > 0 ? wave - 1 : 0)) : 0.f;
            u_next = __builtin_bit_cast(float, __builtin_amdgcn_readlane(Bi, wave));
            end_last = __builtin_amdgcn_readlane(r.w, last_wave);
            par ^= 1;
            return (__builtin_amdgcn_ballot_w64(rec && (r.z & 1)) != 0ull ? 1 : 0) | (__builtin_amdgcn_ballot_w64(rec && (r.z & 2)) != 0ull ? 2 : 0);
        }
    };
    // first guess.  With t steps before it a lane's start value lies between  a = S + (the sum of their e)  -- no step saw the
    // estimate below T -- and  b = a + t  -- every step did --: take the one of a, 0, b in the middle ("at T, unless not even the
    // extreme count gets it there").  An estimate that stays on its side of T through the block makes that exact; one that hovers
    // at T is guessed to within its excursion.  In integers (these sums go up to 2^31; the gaps of the rounds are counts of steps).
    int zs, next0; // this lane's start value minus T; the same of the lane behind lane 63
    {
        tot = mine ? tot : 0;
        const int incl = wave_inclusive_scan(tot);
        const int total = __builtin_amdgcn_readlane(incl, 63);
        int before = 0;
        if constexpr (NW > 1) {
            if (lane == 0)
                xch[par * NW + wave] = make_int4(total, 0, 0, 0);
            __syncthreads();
            int t = lane < NW ? xch[par * NW + (lane < NW ? lane : NW - 1)].x : 0; // lane w: wave w's total
            t += __builtin_amdgcn_update_dpp(0, t, 0x111, 0xf, 0xf, true);        // row_shr:1, 2, 4: the sums up to wave w
            if constexpr (NW > 2)
                t += __builtin_amdgcn_update_dpp(0, t, 0x112, 0xf, 0xf, true);
            if constexpr (NW > 4)
                t += __builtin_amdgcn_update_dpp(0, t, 0x114, 0xf, 0xf, true);
            before = wave > 0 ? __builtin_amdgcn_readlane(t, wave > 0 ? wave - 1 : 0) : 0;
            par ^= 1;
        }
        const int a = S + before + (incl - tot), a_next = S + before + total;
        zs = a >= 0 ? a : min(0, a + kRun * (64 * wave + lane));
        next0 = a_next >= 0 ? a_next : min(0, a_next + kRun * 64 * (wave + 1));
    }
    float slope = 0.f;
    int zs_prev = 0, tot_prev = 0;
    for (int it = 0; it < rounds; ++it) {
        int z = zs, zmin = zs, zmax = zs;
#pragma unroll
        for (int i = 0; i < kRun; ++i) {
            z = z + e[i] + (int)((unsigned)z >> 31);
            zmin = min(zmin, z);
            zmax = max(zmax, z);
        }
        tot = mine ? z - zs : 0;
        const int end = zs + tot;
        const int zs_behind = __builtin_amdgcn_update_dpp(next0, zs, 0x130, 0xf, 0xf, false); // wave_shl:1 (lane 63: next0)
        const int gap = linked ? end - zs_behind : 0;
        const bool ok = !mine || (!tie && zmin + T >= lo && zmax + T < hi);
        const int flags = (__builtin_amdgcn_ballot_w64(gap != 0) != 0ull ? 1 : 0) | (__builtin_amdgcn_ballot_w64(!ok) != 0ull ? 2 : 0);
        // the slope of this lane's total in its start value: the secant of its last two evaluations; before there are two, out
        // of the run itself -- one that saw both sides of T has its 16 merge points spread over about the range it covered
        if (it == 0)
            slope = zmin < 0 && zmax >= 0 ? -fminf(1.f, (float)kRun * __builtin_amdgcn_rcpf((float)(zmax - zmin + 1))) : 0.f;
        else if (zs != zs_prev)
            slope = fminf(0.f, fmaxf(-1.f, (float)(tot - tot_prev) * __builtin_amdgcn_rcpf((float)(zs - zs_prev))));
        zs_prev = zs, tot_prev = tot;
        float A = linked ? 1.0f + slope : 1.0f, B = (float)gap;
        affine_scan_wave(A, B);
        const float Aw = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, A), 63));
        const float Bw = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, B), 63));
        const int end_mine = __builtin_amdgcn_readlane(end, nv > 0 ? nv - 1 : 0);
        float u_in = 0.f, u_next = 0.f;
        int end_last = end_mine;
        const int f = exchange(Aw, Bw, flags, end_mine, u_in, u_next, end_last);
        if (!(f & 1)) { // every lane of every wave started this round where its predecessor ended
            if (f & 2)
                return false;
            const unsigned hi_bits = sign | ((ex - 1u) << 23); // + a mantissa with its leading one = the float
            start = __builtin_bit_cast(float, hi_bits + (unsigned)(zs + T));
            acc_out = __builtin_bit_cast(float, hi_bits + (unsigned)(end_last + T));
            return true;
        }
        const float Ae = dpp_f32<0x138, 0xf>(1.0f, A), Be = dpp_f32<0x138, 0xf>(0.0f, B); // wave_shr:1: the maps of the lanes BEFORE this one
        zs += (int)__builtin_rintf(__builtin_fmaf(Ae, u_in, Be));
        next0 += (int)__builtin_rintf(u_next);
    }
    return false;
}
// The rounded operations themselves on one wave's block, lane after lane: every lane runs the 16 steps on its own products from
// the uniform accumulator, lane l's result is the accumulator of the next round.  Returns the accumulator after the block.
__device__ __forceinline__ float dc_sequential_block(float acc, const float *p, int nv, int lane, float &start)
{
    const float keep = 1.0f - 0.000001f;
    start = acc;
    for (int l = 0; l < nv; ++l) {
        float t = acc;
        if (lane == l)
            start = acc;
#pragma unroll
        for (int i = 0; i < kRun; ++i)
            t = t * keep + p[i]; // -ffp-contract=off: v_mul_f32, v_add_f32
        acc = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, t), l));
    }
    return acc;
}
// One workgroup of NW waves per component walks the frame NW blocks at a time.  A step that does not verify as a whole is
// taken again wave after wave -- each block on its own, from the accumulator its predecessor leaves in LDS: with its own
// context, and with the rounded operations if that does not verify either -- so that one bad spot costs its own 1024 samples
// the sequential time and its NW - 1 neighbours the one-wave time.
// counters: [0] blocks walked, [1] blocks redone with the sequential operations, [2] blocks taken again on their own
template <int NW>
__global__ __launch_bounds__(64 * NW) void k_dc_chain_spec(const float *__restrict__ P, float *__restrict__ A, int n_complex, int stride,
                                                           float *__restrict__ state, unsigned long long *__restrict__ counters, int rounds)
{
    __shared__ int4 xch[2 * NW];
    __shared__ float handover[2];
    const int c = blockIdx.x, lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); // component; lane; wave
    const float *p_c = P + (size_t)c * stride;
    float *a_c = A + (size_t)c * (stride >> 4);
    float acc = state[c]; // (uniform over the workgroup throughout)
    const int nblk = (n_complex + kDcBlock - 1) / kDcBlock;
    unsigned fallbacks = 0, retried = 0;
    int par = 0;
    float p[kRun], pn[kRun];
    auto load = [&](int b, float *dst) { // lane's 16 products of block b; behind the frame the last block's again (never used) --
        // ALWAYS four loads: behind a branch the compiler could not count them and would wait for the prefetch where it only
        // needs the block before it
        const float4 *src = reinterpret_cast<const float4 *>(p_c + (size_t)min(b, nblk - 1) * kDcBlock + lane * kRun);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const v4f v = gldv4(src + i);
            dst[4 * i] = v.x, dst[4 * i + 1] = v.y, dst[4 * i + 2] = v.z, dst[4 * i + 3] = v.w;
        }
    };
    // A step's start values leave in the NEXT step, right behind the point where that one has waited for its products: loads and
    // stores share one counter here, and a store issued at the end of a step would be waited for along with the products (its
    // acknowledgement takes longer than the step's arithmetic up to there).
    float held = 0.f;
    int held_at = -1;
    auto store_held = [&]() {
        if (held_at >= 0)
            *(SDRX_AS1 float *)(a_c + held_at) = held;
        held_at = -1;
    };
    // one step: `p` holds the lane's products of block b, `pn` receives those of block b + NW meanwhile
    auto step = [&](int b, float *p, float *pn) {
        load(b + NW, pn);
        const int nv = max(0, min(64, (n_complex - b * kDcBlock) >> 4)); // lanes that hold samples (frames are multiples of 16)
        const int last_wave = min(NW, nblk - (b - wave)) - 1;            // the last wave that holds any
        float start = acc, acc_out = acc;
        if (dc_spec_blocks<NW>(acc, p, nv, lane, wave, last_wave, xch, par, rounds, start, acc_out, store_held)) {
            acc = acc_out;
        } else if constexpr (NW == 1) {
            ++fallbacks;
            acc = dc_sequential_block(acc, p, nv, lane, start);
        } else {
            for (int w = 0; w < NW; ++w) {
                if (wave == w) {
                    int none = 0;
                    if (nv > 0 && dc_spec_blocks<1>(acc, p, nv, lane, 0, 0, nullptr, none, rounds, start, acc_out, [] {})) {
                        ++retried;
                        acc = acc_out;
                    } else if (nv > 0) {
                        ++fallbacks;
                        acc = dc_sequential_block(acc, p, nv, lane, start);
                    }
                    if (lane == 0)
                        handover[w & 1] = acc;
                }
                __syncthreads();
                acc = handover[w & 1];
            }
        }
        store_held(); // (only if this step never came to its inputs' end: no context)
        held = start, held_at = lane < nv ? b * 64 + lane : -1;
    };
    load(wave, p);
    for (int b = wave; b - wave < nblk; b += 2 * NW) { // (two steps per round: the product registers swap roles instead of being copied)
        step(b, p, pn);
        if (b - wave + NW < nblk)
            step(b + NW, pn, p);
    }
    store_held();
    if (lane == 0) {
        if (wave == 0)
            state[c] = acc;
        if (counters) {
            if (wave == 0)
                atomicAdd(counters + 0, (unsigned long long)nblk);
            if (fallbacks)
                atomicAdd(counters + 1, (unsigned long long)fallbacks);
            if (retried)
                atomicAdd(counters + 2, (unsigned long long)retried);
        }
    }
}

// curr - avept in tile layout: one thread = the 16 samples behind one stored estimate = one lane's run of a tile
__global__ __launch_bounds__(256) void k_dc_apply(const unsigned *__restrict__ bytes4, const float *__restrict__ A, float4 *__restrict__ tiled,
                                                  int n_complex, int stride)
{
    const int j = blockIdx.x * 256 + threadIdx.x; // block of 16 samples
    if (16 * j >= n_complex)
        return;
    const float keep = 1.0f - 0.000001f, k = 0.000001f;
    float ai = A[j], aq = A[(stride >> 4) + j];
    const v4u *src = reinterpret_cast<const v4u *>(bytes4) + 2 * j; // 32 bytes
    const int ch = j >> 6, lane = j & 63;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const v4u q = gldv4u(src + h);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const unsigned u = q[e];
            const float x0 = (float)((int)(u & 255u) - 127), y0 = (float)((int)((u >> 8) & 255u) - 127);
            const float x1 = (float)((int)((u >> 16) & 255u) - 127), y1 = (float)((int)(u >> 24) - 127);
            ai = ai * keep + k * x0;
            aq = aq * keep + k * y0;
            float4 v;
            v.x = x0 - ai, v.y = y0 - aq;
            ai = ai * keep + k * x1;
            aq = aq * keep + k * y1;
            v.z = x1 - ai, v.w = y1 - aq;
            tiled[tile_unit(ch, 4 * h + e, lane)] = v;
        }
    }
}

// The fast-arithmetic form of the same DC-bias removal (option "exact" = 0): the recurrence is
// linear, a_t = A a_{t-1} + B x_t, so it is evaluated as a blocked scan -- per 1024-sample chunk
// one wave runs 16 sequential steps per lane from zero, combines the 64 lane totals with a
// log-step scan of affine maps, and adds the carry-in  A^(16 l + i + 1) * (state before the chunk).
// k_dc_block_sums leaves each chunk's zero-state response; k_ingest_u8_dc_fast folds the responses
// of all earlier chunks into its own carry-in (<= 375 terms, powers of A^1024 from a host table)
// and writes the corrected chunk in tile layout.  Chunk-level sums are kept in double; against the
// reference's sequentially ROUNDED fp32 recurrence the estimate differs by ~1e-5 of the DC offset
// itself (a random walk of half-ulp roundings over the filter's 1e6-sample memory), far below 1e-5
// of the signal.  dctab: [0..16] = A^k, [32..95] = A^(16 l), [96 + k] = A^(1024 k).
struct DcLocal {
    float ai[kRun], aq[kRun]; // zero-state response at each of the lane's 16 samples
    float xi[kRun], xq[kRun];
};
__device__ __forceinline__ void dc_local(const unsigned *__restrict__ bytes4, int base, int valid, int lane, DcLocal &L)
{
    const float keep = 1.0f - 0.000001f, k = 0.000001f;
    float ai = 0.f, aq = 0.f;
#pragma unroll
    for (int i2 = 0; i2 < 8; ++i2) {
        const int s0 = lane * kRun + 2 * i2;
        const unsigned w = s0 < valid ? bytes4[((base + s0) >> 1)] : 0x7f7f7f7fu;
        const float x[4] = {(float)((int)(w & 255u) - 127), (float)((int)((w >> 8) & 255u) - 127), (float)((int)((w >> 16) & 255u) - 127),
                            (float)((int)(w >> 24) - 127)};
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            ai = fmaf(ai, keep, k * x[2 * h]);
            aq = fmaf(aq, keep, k * x[2 * h + 1]);
            L.ai[2 * i2 + h] = ai, L.aq[2 * i2 + h] = aq;
            L.xi[2 * i2 + h] = x[2 * h], L.xq[2 * i2 + h] = x[2 * h + 1];
        }
    }
}
// inclusive scan over the lanes of S_l = T_l + A16 * S_{l-1}
__device__ __forceinline__ void dc_wave_scan(double &si, double &sq, const double *__restrict__ dctab, int lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const double m = dctab[32 + d]; // A^(16 d)
        const double ti = __shfl_up(si, d), tq = __shfl_up(sq, d);
        if (lane >= d) {
            si = fma(m, ti, si);
            sq = fma(m, tq, sq);
        }
    }
}
__global__ __launch_bounds__(64) void k_dc_block_sums(const unsigned *__restrict__ bytes4, int n_complex, const double *__restrict__ dctab,
                                                      double2 *__restrict__ sums)
{
    const int c = blockIdx.x, lane = threadIdx.x, base = c * kChunk;
    DcLocal L;
    dc_local(bytes4, base, min(kChunk, n_complex - base), lane, L);
    double si = L.ai[kRun - 1], sq = L.aq[kRun - 1];
    dc_wave_scan(si, sq, dctab, lane);
    if (lane == 63)
        sums[c] = make_double2(si, sq);
}
__global__ __launch_bounds__(64) void k_ingest_u8_dc_fast(const unsigned *__restrict__ bytes4, float4 *__restrict__ tiled, int n_complex,
                                                          const float *__restrict__ state_in, float *__restrict__ state_out,
                                                          const double *__restrict__ dctab, const double2 *__restrict__ sums)
{
    const int c = blockIdx.x, lane = threadIdx.x, base = c * kChunk;
    const int valid = min(kChunk, n_complex - base);
    // carry-in of the chunk: the state before the frame and every earlier chunk's response
    double ci = 0.0, cq = 0.0;
    for (int j = lane; j < c; j += 64) {
        const double m = dctab[96 + (c - 1 - j)];
        const double2 b = sums[j];
        ci = fma(m, b.x, ci);
        cq = fma(m, b.y, cq);
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        ci += __shfl_xor(ci, d);
        cq += __shfl_xor(cq, d);
    }
    ci = fma(dctab[96 + c], (double)state_in[0], ci);
    cq = fma(dctab[96 + c], (double)state_in[1], cq);
    DcLocal L;
    dc_local(bytes4, base, valid, lane, L);
    double si = L.ai[kRun - 1], sq = L.aq[kRun - 1];
    dc_wave_scan(si, sq, dctab, lane);
    double pi = __shfl_up(si, 1), pq = __shfl_up(sq, 1); // S_{l-1}
    if (lane == 0)
        pi = 0.0, pq = 0.0;
    const double inc_i = fma(dctab[32 + lane], ci, pi), inc_q = fma(dctab[32 + lane], cq, pq); // state before this lane's run
    float ai_last = 0.f, aq_last = 0.f;
#pragma unroll
    for (int i2 = 0; i2 < 8; ++i2) {
        float y[4];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int i = 2 * i2 + h;
            const float ai = (float)fma(dctab[i + 1], inc_i, (double)L.ai[i]);
            const float aq = (float)fma(dctab[i + 1], inc_q, (double)L.aq[i]);
            y[2 * h] = L.xi[i] - ai;
            y[2 * h + 1] = L.xq[i] - aq;
            ai_last = ai, aq_last = aq;
        }
        if (lane * kRun + 2 * i2 < valid)
            tiled[tile_unit(c, i2, lane)] = make_float4(y[0], y[1], y[2], y[3]);
    }
    if (base + valid == n_complex && lane == (valid >> 4) - 1) { // the frame's last sample (frames are multiples of 16)
        state_out[0] = ai_last;
        state_out[1] = aq_last;
    }
}

// ------------------------------------------------------------------------------------ k_mix_decimate
// Whole-wave DPP shift by one lane (GFX9 `wave_shr:1`): lane l receives src of lane l-1,
// lane 0 keeps `old`.  Lane semantics verified on gfx950 by tools/dpp_probe.hip.
__device__ __forceinline__ float shr1(float old, float src)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src),
                                                                 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ v2f shr1(v2f old, v2f src)
{
    v2f r = {shr1(old.x, src.x), shr1(old.y, src.y)};
    return r;
}

// LDS of k_mix_decimate (per wave): two 8-entry carry rows for the register stages, then the
// linear stage arrays A_s = [16 carry | 1024 >> s data] for the stages s >= 2 that run in LDS,
// and (only when a d == 0 VFO must emit natural order) a padded transpose tile.
constexpr int kRegStages = 2;                       // stages 0 and 1 live in registers
constexpr int kCarryBytes = 2 * 8 * 8;              // car0[8], car1[8] float2
__host__ __device__ constexpr int stage_elems(int s) { return kCarry + (s >= 3 ? 2 : 1) * (kChunk >> s); } // (stages >= 3: two chunks' worth, hb_stage_fixed)
__host__ __device__ constexpr int stage_offset(int s) // float2 index of A_s, s >= kRegStages
{
    int o = 0;
    for (int t = kRegStages; t < s; ++t)
        o += stage_elems(t);
    return o;
}
__host__ __device__ constexpr int pad0(int p) { return p + 2 * (p >> 4); }
constexpr int kTransposeElems = kChunk + 2 * (kChunk >> 4); // 1152 float2
__host__ __device__ constexpr int k1_lds_bytes(int d, bool need_transpose)
{
    int stages = 8 * stage_offset(d < kRegStages ? kRegStages : d);
    if (d == kRegStages && stages < 64 * 32)
        stages = 64 * 32; // a d = 2 leaf parks its four outputs per lane here until the next chunk's loads are issued (HeldStores)
    int tr = need_transpose ? 8 * kTransposeElems : 0;
    return kCarryBytes + (stages > tr ? stages : tr);
}

// k_mix_decimate is ONE wave per workgroup and every LDS byte it touches is private to that wave.
// A wave's LDS instructions execute in program order, so a write by one lane is visible to a later
// read by another lane without any wait; all that is needed between phases is that the COMPILER
// keeps the order.  __syncthreads() would do that too, but it also emits s_waitcnt vmcnt(0): every
// phase boundary would drain the global loads and stores in flight (ten times per chunk).
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// One half-band stage of one chunk in LDS (stages >= 2): A = [16 carry | cnt data] -> B data, or
// the output stream when this is the last stage.
template <bool EXACT>
__device__ __forceinline__ void hb_stage_lds(v2f *__restrict__ A, v2f *__restrict__ B, float2 *__restrict__ gout, int gbase,
                                             bool tiled, bool last, int jmin, int cnt, int lane, bool save,
                                             float2 *__restrict__ hbsave)
{
    wave_sync(); // stage input (written by the previous phase) is visible
    const int nout = cnt >> 1;
    for (int j = lane; j < nout; j += 64) {
        const v2f *w = A + kCarry + 2 * j - 10; // w[0..10], newest = input sample 2j of this chunk
        const v2f y = hb_dot2<EXACT>(w[0], w[2], w[4], w[5], w[6], w[8], w[10]);
        if (!last)
            B[kCarry + j] = y;
        else if (j >= jmin) { // outputs below jmin belong to the warm-up of a segment that starts inside the frame
            if (tiled)
                gstv2(gout + tile_pos(gbase + j), y);
            else
                gstv2_leaf(gout + (size_t)(gbase + j), y);
        }
    }
    wave_sync(); // all window reads done before the carry is overwritten
    // FIRQueueBackToFront (dsp.cpp:163-173) at the end of the FRAME: x[-k] := x[size-1-k]
    if (save && lane < kHbHist)
        gstv2(hbsave + lane, A[kCarry + cnt - 2 - lane]);
    v2f t;
    if (lane < kCarry)
        t = A[cnt + lane]; // the last 16 of [carry | data]
    wave_sync();
    if (lane < kCarry)
        A[lane] = t;
}

// The same LDS stages 2 .. D-1 for the common case -- a FULL chunk (1024 inputs) of a leaf VFO (natural-order output)
// that does not hold the frame's last sample -- with the depth as a compile-time constant: every count, LDS offset
// and trip count folds, the per-stage loop, the tiled / natural and last / not-last selects and the save branch
// disappear (the generic routine spends about as many VALU instructions on them as on the 11-operation dot product).
// Same arithmetic, same order, same LDS contents afterwards.  -DSDRX_FIXED_STAGES=0 switches it off (A/B).
#ifndef SDRX_FIXED_STAGES
#define SDRX_FIXED_STAGES 1
#endif
constexpr int kFixedDepth = 5; // 1.536 MS/s / 384 kS/s -> 48 / 12 kS/s: the depth of the reference's sub VFOs below a 384 k main
// Stage S of a full chunk: A_S = [16 carry | M * (1024 >> S) inputs] -> outputs into B's data from entry `boff` on, or -- the last stage --
// to the leaf's stream.  M = 2: the stage runs every SECOND chunk on two chunks' worth of input (SDRX_PAIR_STAGES: stages >= 3
// of a d = 5 leaf, whose last stage otherwise fills 32 of the 64 lanes; one carry hand-over and one set of phase fences per
// two chunks).  Same arithmetic on the same values in the same order: an output does not know how many of its neighbours
// were computed in the same pass.
// A chunk's output stores, held back until the NEXT chunk's loads have been issued (mix_item's chunk loop): vmcnt counts a
// wave's loads and stores in ONE in-order queue, so a wave that waits for loads issued behind its stores also waits for the
// stores' acknowledgements -- measured on the tolerance-arithmetic kernel, the stores in front of the next chunk's loads cost
// the launch 13 of its 63 us (profiles/README.md, round 5).  Stores issued behind the loads are never waited for.
struct HeldStores {
    v2f one;       // units == 1: the cf32 itself (last LDS stage of a d = 5 leaf)
    int units = 0; // (everything below is wave-uniform: SGPRs) 0 nothing held; 1: `one` = output gbase + lane, held by the lanes
                   //   jmin <= lane < nout; 2: z[0..3] of a d = 2 leaf, parked in the lane's 32 bytes of LDS, of the chunk at `base`
    int gbase = 0, jmin = 0, nout = 0;
    int base = 0, lv = 0;
};
template <bool EXACT, int S, int D, int M, int boff = 0>
__device__ __forceinline__ void hb_stage_fixed1(v2f *__restrict__ lds, HeldStores &held, int gbase, int jmin, int lane)
{
    constexpr int cnt = M * (kChunk >> S), nout = cnt >> 1;
    v2f *A = lds + stage_offset(S);
    v2f *B = lds + stage_offset(S + 1); // (not touched by the last stage)
    asm volatile("" : "+v"(lane)); // (the pass's LDS addresses are computed here, from this: hoisted out of the chunk loop they spill)
    wave_sync(); // stage input (written by the previous phase) is visible
    if constexpr (nout >= 64) {
#pragma unroll
        for (int it = 0; it < nout / 64; ++it) {
            const int j = lane + 64 * it;
            const v2f *w = A + kCarry + 2 * j - 10;
            const v2f y = hb_dot2<EXACT>(w[0], w[2], w[4], w[5], w[6], w[8], w[10]);
            if constexpr (S + 1 < D) {
                B[kCarry + boff + j] = y;
            } else { // the leaf's stream: one output per lane, stored behind the next chunk's loads
                static_assert(nout <= 64, "the last stage of a fixed-depth leaf has at most one output per lane");
                held.one = y;
                held.units = 1, held.gbase = gbase, held.jmin = jmin, held.nout = nout;
            }
        }
    } else {
        if constexpr (S + 1 == D)
            held.units = 1, held.gbase = gbase, held.jmin = jmin, held.nout = nout;
        if (lane < nout) {
            const v2f *w = A + kCarry + 2 * lane - 10;
            const v2f y = hb_dot2<EXACT>(w[0], w[2], w[4], w[5], w[6], w[8], w[10]);
            if constexpr (S + 1 < D) {
                B[kCarry + boff + lane] = y;
            } else {
                held.one = y;
            }
        }
    }
    wave_sync(); // all window reads done before the carry is overwritten
    v2f t;
    if (lane < kCarry)
        t = A[cnt + lane]; // the last 16 of [carry | data]
    wave_sync();
    if (lane < kCarry)
        A[lane] = t;
}
// stages S .. D-1, each on M chunks' worth of input
template <bool EXACT, int S, int D, int M>
__device__ __forceinline__ void hb_stage_fixed(v2f *__restrict__ lds, HeldStores &held, int gbase, int jmin, int lane)
{
    hb_stage_fixed1<EXACT, S, D, M>(lds, held, gbase, jmin, lane);
    if constexpr (S + 1 < D)
        hb_stage_fixed<EXACT, S + 1, D, M>(lds, held, gbase, jmin, lane);
}
#ifndef SDRX_PAIR_STAGES
#define SDRX_PAIR_STAGES 1
#endif

// NOUT outputs of a register-resident stage.  ext[k] holds input sample k-10 of this lane's run
// (k = 0..9: the halo, only the needed ones set).
template <bool EXACT, int NOUT>
__device__ __forceinline__ void hb_regs(const v2f *ext, v2f *y)
{
#pragma unroll
    for (int j = 0; j < NOUT; ++j) {
        const v2f *w = ext + 2 * j; // w[t] = input sample 2j - 10 + t
        y[j] = hb_dot2<EXACT>(w[0], w[2], w[4], w[5], w[6], w[8], w[10]);
    }
}

// halo slot q (0..7) <-> distance k back from the lane's first sample: -10,-8,-6,-5,-4,-3,-2,-1
__device__ __forceinline__ constexpr int halo_k(int q) { return q == 0 ? 10 : q == 1 ? 8 : q == 2 ? 6 : 8 - q; }

// ---- the pieces of a chunk that mix_item and late_item share ---------------------------------------------------------------
// The lane's 16 consecutive samples of a raw (natural-order) frame: cf32 as the caller handed it -- each lane reads its own
// 128 contiguous bytes: uncoalesced across the wave, but a level of 2-3 main VFOs is latency bound and this saves the layout
// pass over the raw frame -- or dongle bytes, floats[b] = b - 127 (jonti/sdr.cpp:43-49), 32 bytes per lane.
__device__ __forceinline__ void load_run_raw(const void *__restrict__ raw, int raw_mode, int p16, bool active, v2f *x)
{
    if (raw_mode == kRawF32) {
        const float4 *nat = reinterpret_cast<const float4 *>(raw) + (size_t)p16 * 8;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const v4f z4 = {0.f, 0.f, 0.f, 0.f};
            const v4f v = active ? gldv4(nat + i) : z4;
            x[2 * i] = lo2(v);
            x[2 * i + 1] = hi2(v);
        }
    } else {
        const v4u *nat = reinterpret_cast<const v4u *>(raw) + (size_t)p16 * 2;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const v4u z4 = {0x7f7f7f7fu, 0x7f7f7f7fu, 0x7f7f7f7fu, 0x7f7f7f7fu}; // 127 -> 0.0f
            const v4u q = active ? gldv4u(nat + h) : z4;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const unsigned w = q[k];
                const v2f a = {(float)((int)(w & 255u) - 127), (float)((int)((w >> 8) & 255u) - 127)};
                const v2f b = {(float)((int)((w >> 16) & 255u) - 127), (float)((int)(w >> 24) - 127)};
                x[8 * h + 2 * k] = a;
                x[8 * h + 2 * k + 1] = b;
            }
        }
    }
}
// The same out of a tile-layout stream: 8 coalesced 16-byte loads, unit (i2, lane) of the tile at `src`.
__device__ __forceinline__ void load_run_tile(const float4 *__restrict__ src, v2f *x)
{
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const v4f v = gldv4(src + 64 * i); // tile_unit(c, i, l) = tile_unit(c, 0, l) + 64 i
        x[2 * i] = lo2(v);
        x[2 * i + 1] = hi2(v);
    }
}
// NCO + mixer on the lane's run (oscillator.cpp:20-28,39-50; vfo.cpp:241): regenerate table[idx .. idx + 16) from the
// checkpoint `o` before it -- the exact recurrence, or in the tolerance arithmetic rotations of the checkpoint where the
// chunk's `span` table entries from position i0 on are settled ones and do not wrap -- and multiply.  The very first sample
// after start-up is multiplied by the LAST table entry (`last`: oscillator.cpp:30,39-50).
// Three arithmetics (option "exact"): EXACT -- the reference's operations in its order, every product and sum rounded.
// !EXACT && ROT ("tolerance") -- table entries as rotations of the exact checkpoint, FMA mixer and filters: cheapest, but the
// table's error (~1e-6 of |v|) multiplies the TOTAL input power, so a strong out-of-band carrier eats the 1e-5 bar.
// !EXACT && !ROT ("robust") -- the table entries from the exact recurrence (error 0: bit-identical to the reference's table),
// FMA mixer and filters: what remains is FMA-versus-two-roundings noise, ~6e-8 per operation, whatever is out of band.
template <bool EXACT, bool ROT>
__device__ __forceinline__ void nco_mix(v2f o, v2f rot, const float2 *__restrict__ rk, const float2 *__restrict__ last, bool first_ever, int i0,
                                        int span, int L, v2f *x)
{
    static_assert(!(EXACT && ROT), "the exact arithmetic replays the table");
    if constexpr (!EXACT && !ROT) {
#pragma unroll
        for (int i = 0; i < kRun; i += 2) {
            o = nco_step_pk(o, rot);
            v2f m0 = o;
            if (i == 0 && first_ever)
                m0 = gldv2(last);
            o = nco_step_pk(o, rot);
            cmul_fast2(m0, o, x[i], x[i + 1]);
        }
        return;
    }
    bool replay = true;
    if constexpr (!EXACT) // (wave-uniform)
        replay = i0 < kNcoSettle || i0 + span > L;
    if (replay) {
#pragma unroll
        for (int i = 0; i < kRun; ++i) {
            o = nco_step_pk(o, rot);
            v2f m = o;
            if (i == 0 && first_ever)
                m = gldv2(last);
            x[i] = cmul(m, x[i]);
        }
    } else {
        nco_mix_fast16(o, rk, x);
    }
}
// Stage 0's halo: ext0[0..9] = x[-10..-1] = the previous lane's x[6, 8, 10 .. 15] (one whole-wave DPP shift each); lane 0 takes
// the previous chunk's lane 63 from the 8-entry LDS row car0, which lane 63 then refills for the next chunk.
__device__ __forceinline__ void halo_stage0(v2f *car0, v2f *ext0, int lane)
{
    const v2f zero2 = {0.f, 0.f};
    const v2f *x = ext0 + 10;
    wave_sync(); // car0 of the previous chunk (or the initial state) is visible
    {
        const v4f *c4 = reinterpret_cast<const v4f *>(car0); // broadcast reads
        const v4f q0 = c4[0], q1 = c4[1], q2 = c4[2], q3 = c4[3];
        ext0[0] = shr1(lo2(q0), x[6]);   // x[-10]
        ext0[2] = shr1(hi2(q0), x[8]);   // x[-8]
        ext0[4] = shr1(lo2(q1), x[10]);  // x[-6]
        ext0[5] = shr1(hi2(q1), x[11]);  // x[-5]
        ext0[6] = shr1(lo2(q2), x[12]);  // x[-4]
        ext0[7] = shr1(hi2(q2), x[13]);  // x[-3]
        ext0[8] = shr1(lo2(q3), x[14]);  // x[-2]
        ext0[9] = shr1(hi2(q3), x[15]);  // x[-1]
        ext0[1] = ext0[3] = zero2;       // x[-9], x[-7]: never read
    }
    wave_sync(); // every lane holds its halo: lane 63 may now leave ITS tail for the next chunk
    if (lane == 63) {
        v4f *c4 = reinterpret_cast<v4f *>(car0);
        c4[0] = cat2(x[6], x[8]);
        c4[1] = cat2(x[10], x[11]);
        c4[2] = cat2(x[12], x[13]);
        c4[3] = cat2(x[14], x[15]);
    }
}
// Stage 1's: y[-8, -6, -5, -4, -3, -2, -1] = the previous lane's y[0, 2 .. 7], y[-10] = the lane before that one's y[6] (a second
// shift of the shifted y[6]); lanes 63 and 62 refill car1.
__device__ __forceinline__ void halo_stage1(v2f *car1, v2f *ext1, int lane)
{
    const v2f zero2 = {0.f, 0.f};
    const v2f *y = ext1 + 10;
    {
        const v4f *c4 = reinterpret_cast<const v4f *>(car1);
        const v4f q0 = c4[0], q1 = c4[1], q2 = c4[2], q3 = c4[3];
        ext1[2] = shr1(hi2(q0), y[0]);  // y[-8]
        ext1[4] = shr1(lo2(q1), y[2]);  // y[-6]
        ext1[5] = shr1(hi2(q1), y[3]);  // y[-5]
        ext1[6] = shr1(lo2(q2), y[4]);  // y[-4]
        ext1[7] = shr1(hi2(q2), y[5]);  // y[-3]
        ext1[8] = shr1(lo2(q3), y[6]);  // y[-2]
        ext1[9] = shr1(hi2(q3), y[7]);  // y[-1]
        ext1[0] = shr1(lo2(q0), ext1[8]); // y[-10]
        ext1[1] = ext1[3] = zero2;
    }
    wave_sync(); // every lane has read car1
    if (lane == 63) {
        car1[1] = y[0]; // slot 1 (y[-8]); slot 0 comes from lane 62
        v4f *d4 = reinterpret_cast<v4f *>(car1);
        d4[1] = cat2(y[2], y[3]);
        d4[2] = cat2(y[4], y[5]);
        d4[3] = cat2(y[6], y[7]);
    }
    if (lane == 62)
        car1[0] = y[6];
}

// ------------------------------------------------------------------------------------ demod tail
// `short = double` of the reference's x86-64 build (vfo.cpp:328,364): cvttsd2si truncates toward zero to
// int32 and yields INT32_MIN ("integer indefinite") for anything outside int32 or NaN; the low 16 bits are
// kept.  (The C standard calls the out-of-range case undefined; this is what the reference's binary does,
// and what the oracle restates.)  v_cvt_i32_f64 truncates the same way but SATURATES, hence the select.
__device__ __forceinline__ short to_short(double d)
{
    const int t = (d >= -2147483648.0 && d < 2147483648.0) ? (int)d : (int)0x80000000;
    return (short)(unsigned short)(unsigned)t;
}
// The reference's two excursions into double on this path (vfo.cpp:317-328) are kept literally.  Both have
// an fp32 form with identical results -- the exact difference of two floats rounded to 53 and then to 24
// bits is the fp32 subtraction (double rounding is innocuous for + - * / when the wide format has
// >= 2*24 + 2 significand bits), and `float * 2^15` is exact in either format with v_cvt_i32_f32
// saturating like v_cvt_i32_f64 -- which -DSDRX_DEMOD_F32 selects: all 115 GPU parity tests pass with it,
// and it is not faster (k_usb_demod 35.8-37.0 vs 35.7-36.6 us on config 3), so the literal form stays.
__device__ __forceinline__ float usb_difference(float delayed_i, float hilbert)
{
#ifndef SDRX_DEMOD_F32
    return (float)((double)delayed_i - (double)hilbert);
#else
    return delayed_i - hilbert;
#endif
}
__device__ __forceinline__ float quantise(float scaled, short &out)
{
#ifndef SDRX_DEMOD_F32
    const double pre = (double)scaled * 32768.0;
    // |pre| < 2^31 <=> |scaled| < 2^16 (the product is exact): the range test of to_short() as ONE fp32 compare
    out = fabsf(scaled) < 65536.0f ? (short)(unsigned short)(unsigned)(int)pre : (short)0;
    return (float)pre; // exact: float * 2^15
#else
    const float pre = scaled * 32768.0f;
    out = (short)(unsigned short)(unsigned)((pre >= -2147483648.0f && pre < 2147483648.0f) ? (int)pre : (int)0x80000000);
    return pre;
#endif
}


// ---- the USB demodulation INSIDE the leaf's wave (vfo::usb_demod, vfo.cpp:300-332) -----------------------------------------
// A d = 2 leaf below a parent (the reference's 48 kS/s sub VFOs: 83 % of BASELINE config 3's demodulation work) turns every
// 1024-sample chunk into 256 stream samples, four per lane -- and demodulates them on the spot: the leaf's cf32 stream never
// reaches HBM (round 5: 63 MB written by the mix launch and 75 MB read back by k_usb_demod per frame of config 3), only
// the int16 payload does.  Same arithmetic, same order as demod_block (one accumulator per output, taps ascending, every
// product and sum rounded in the exact arithmetic):
//   usb[m]  = I[m-62] - sum_{s<62} hnz[s] Q[m-123+2s]       (the 62 odd Hilbert taps; the even ones are exactly 0)
//   usb'[m] = sum_{i<N} hu[i] usb[m-N+i]                    (FIR::FIRUpdateAndProcess: newest sample excluded, dsp.cpp:59-71)
//   out[m]  = short(usb' * gain * 32768.0)
// LDS of the wave (floats; all histories in FRONT of the chunk's new values, so a window is one contiguous run):
//   QO [62 | 128]   odd-indexed Q of the stream (local index 2j+1 -> entry 62 + j): what the EVEN outputs read
//   QE [3 + 62 | 128]  even-indexed Q, stored 3 floats up so that the b128 window reads of the odd outputs are aligned
//   I  [2 + 62 | 256]  I, stored 2 floats up so that a lane's four new values are one aligned b128 write
//   U  [Nh | 256 | 8]  usb: Nh = N rounded up to a multiple of 4 values of history, then the chunk's, then zeros the low-pass
//                      reads under its zero padding taps
//   H  [N + 15]        the low-pass taps with 3 zeros in front (K2Vfo::lpf_pad)
// Output t = p + 8 q + 2 rr of lane (p = lane / 32, q = lane % 32) -- four outputs of equal parity share one contiguous run of
// 65 plane entries (17 b128 reads for 248 MACs), the Hilbert taps are wave-uniform scalars (re-requested every chunk: 62 SGPRs
// that are free again for the mix phase).  The low-pass then takes four CONSECUTIVE outputs per lane from U.
// Between chunks the last 62 / 62 / 62 / Nh entries move to the front; between frames they live in K2Vfo::state (256 floats per
// frame parity: QO | QE | I | U at 64-float strides), zero at start-up like every filter state of the reference (dsp.cpp:40-49).
constexpr int kDmHist = 62;
constexpr int kDmQO = 0, kDmQE = 192, kDmI = 388, kDmU = 708, kDmH = 1036, kDmFloats = 1116;
constexpr int kDmMaxLpf = 64; // longest audio low-pass the wave applies itself (the reference's 10 kHz filter at 48 kS/s has 47 taps)
static_assert(kDmQE - kDmQO >= kDmHist + 128 + 2 && kDmI - kDmQE >= 3 + kDmHist + 128 + 3 && kDmU - kDmI >= 2 + kDmHist + 256, "plane sizes");
static_assert(kDmH - kDmU >= kDmMaxLpf + 256 + 8 && kDmFloats - kDmH >= kDmMaxLpf + 15, "usb / tap sizes");
static_assert(kDmQE % 4 == 0 && kDmI % 4 == 0 && kDmU % 4 == 0 && kDmH % 4 == 0, "b128 alignment of the arrays");
__host__ __device__ constexpr int demod_lds_bytes() { return kCarryBytes + 4 * kDmFloats; }

struct DemodCtx { // wave-uniform (SGPRs): as little as possible lives across the mix phase -- the descriptor's fields are
    const K2Vfo *Kp; //   re-read (scalar loads from the constant address space) where a chunk needs them
    int par, nlpf, Nh;
};
__device__ __forceinline__ void demod_prologue(float *dm, const K2Vfo *Kp, int par, bool from_state, int lane, DemodCtx &C)
{
    C.Kp = Kp;
    C.par = par;
    C.nlpf = ldc(&Kp->nlpf);
    C.Nh = (C.nlpf + 3) & ~3;
    const float *st = ldc(&Kp->state[par]);
    const float *lpf = ldc(&Kp->lpf_pad);
    // all loads first, then the LDS writes
    float h0 = 0.f, h1 = 0.f, h2 = 0.f, h3 = 0.f, t0 = 0.f, t1 = 0.f;
    if (from_state && lane < kDmHist)
        h0 = gld(st + lane), h1 = gld(st + 64 + lane), h2 = gld(st + 128 + lane);
    if (from_state && lane < C.Nh)
        h3 = gld(st + 192 + lane);
    if (lane < C.nlpf + 15 && C.nlpf > 0)
        t0 = gld(lpf + lane);
    if (lane + 64 < C.nlpf + 15 && C.nlpf > 0)
        t1 = gld(lpf + lane + 64);
    const v4f z4 = {0.f, 0.f, 0.f, 0.f};
    for (int t = lane; t < kDmFloats / 4; t += 64)
        reinterpret_cast<v4f *>(dm)[t] = z4;
    wave_sync();
    if (lane < kDmHist)
        dm[kDmQO + lane] = h0, dm[kDmQE + 3 + lane] = h1, dm[kDmI + 2 + lane] = h2;
    if (lane < C.Nh)
        dm[kDmU + lane] = h3;
    dm[kDmH + lane] = t0;
    if (lane + 64 < kDmMaxLpf + 15)
        dm[kDmH + 64 + lane] = t1;
}
// One chunk: z[0..3] = the lane's stream samples 4 lane .. 4 lane + 3 of the chunk (stream index g0 + ...), nv = how many of
// the chunk's 256 are real (a multiple of 4), fo = first stream index this item emits, `last` = the chunk holds the frame's end.
template <bool EXACT>
__device__ __forceinline__ void demod_chunk(float *dm, const DemodCtx &C, const v2f *z, int lane, int nv, int g0, int fo, bool last)
{
    const K2Vfo *Kp = C.Kp;
    asm volatile("" : "+s"(Kp)); // (what is read through it below is read in THIS chunk, not once in front of the chunk loop)
    wave_sync(); // the previous chunk's reads and its hand-over are done
    {
        const v2f qo = {z[1].y, z[3].y};
        *reinterpret_cast<v2f *>(dm + kDmQO + kDmHist + 2 * lane) = qo;
        dm[kDmQE + 3 + kDmHist + 2 * lane] = z[0].y;
        dm[kDmQE + 3 + kDmHist + 2 * lane + 1] = z[2].y;
        const v4f iv = {z[0].x, z[1].x, z[2].x, z[3].x};
        *reinterpret_cast<v4f *>(dm + kDmI + 2 + kDmHist + 4 * lane) = iv;
    }
    wave_sync();
    // ---- Hilbert: four outputs of one parity per lane
    {
        const int p = lane >> 5, q = lane & 31;
        const float *plane = p == 0 ? dm + kDmQO + 4 * q : dm + kDmQE + 4 + 4 * q;
        // The taps sixteen at a time, each batch requested right in front of its pass (every chunk: hoisted out of the chunk
        // loop, or all at once, they would hold 62 SGPRs -- the mix phase needs its own -- and spill into a VGPR that every
        // body of the kernel then loses).  An accumulator still takes its taps in ascending order: pass k has s = 16 k .. 16 k + 15,
        // i.e. plane entries rr + s in 16 k .. 16 k + 18: groups 4 k .. 4 k + 4 (20 b128 reads per chunk instead of 17).
        const float *hnz_all = ldc(&Kp->hnz);
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        static_for<4>([&](auto pass_ic) {
            constexpr int pass = decltype(pass_ic)::value, s_lo = 16 * pass, s_hi = s_lo + 16 < kHilbertNz ? s_lo + 16 : kHilbertNz, g_lo = 4 * pass;
            const float *hp = hnz_all + s_lo;
            asm volatile("" : "+s"(hp));
            float hnz[s_hi - s_lo];
#pragma unroll
            for (int s = 0; s < s_hi - s_lo; ++s)
                hnz[s] = ldc(hp + s);
#pragma unroll
            for (int g = g_lo; g < g_lo + 5 && g < 17; ++g) {
                const v4f v4 = *reinterpret_cast<const v4f *>(plane + 4 * g);
                const float v[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        const int s = 4 * g + e - rr;
                        if (s >= s_lo && s < s_hi) {
                            if (EXACT)
                                acc[rr] = acc[rr] + hnz[s - s_lo] * v[e];
                            else
                                acc[rr] = fmaf(hnz[s - s_lo], v[e], acc[rr]);
                        }
                    }
            }
            __builtin_amdgcn_sched_barrier(0); // (the next batch of taps is not requested while this one is live)
        });
        const int t0 = p + 8 * q;
        const float *iv = dm + kDmI + 2 + t0; // I[t - 62]
        float *u = dm + kDmU + C.Nh + t0;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
            u[2 * rr] = usb_difference(iv[2 * rr], acc[rr]);
    }
    wave_sync();
    // ---- audio low-pass (newest sample excluded) on four consecutive outputs, int16
    {
        const int N = C.nlpf;
        float u4[4];
        if (N > 0) {
            // output 4 lane + rr, tap i reads U[Nh + 4 lane + rr - N + i]; H[i + 3] = hu[i], zeros around
            const float *w = dm + kDmU + (C.Nh - N) + 4 * lane;
            const float *H = dm + kDmH;
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
            const int groups = (N + 6) / 4;
            for (int g = 0; g < groups; ++g) {
                const v4f v4 = *reinterpret_cast<const v4f *>(w + 4 * g);
                const float v[4] = {v4.x, v4.y, v4.z, v4.w};
                const v4f ha = *reinterpret_cast<const v4f *>(H + 4 * g), hb4 = *reinterpret_cast<const v4f *>(H + 4 * g + 4);
                const float h[8] = {ha.x, ha.y, ha.z, ha.w, hb4.x, hb4.y, hb4.z, hb4.w}; // h[k] = hu[4g + k - 3]
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        if (EXACT)
                            acc[rr] = acc[rr] + h[e - rr + 3] * v[e];
                        else
                            acc[rr] = fmaf(h[e - rr + 3], v[e], acc[rr]);
                    }
            }
#pragma unroll
            for (int rr = 0; rr < 4; ++rr)
                u4[rr] = acc[rr];
        } else {
            const v4f v4 = *reinterpret_cast<const v4f *>(dm + kDmU + 4 * lane); // (Nh = 0)
            u4[0] = v4.x, u4[1] = v4.y, u4[2] = v4.z, u4[3] = v4.w;
        }
        short o4[4];
        float pq[4];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
            pq[rr] = quantise(u4[rr] * ldc(&Kp->gain), o4[rr]);
        const int g = g0 + 4 * lane;
        if (4 * lane < nv && g >= fo) { // (nv and fo are multiples of 4: a lane's four outputs are emitted together or not at all)
            const v4s o = {o4[0], o4[1], o4[2], o4[3]};
            *(SDRX_AS1 v4s *)(ldc(&Kp->pay[C.par]) + g) = o;
            float *prequant = ldc(&Kp->prequant);
            if (prequant)
                gst4(reinterpret_cast<float4 *>(prequant + g), make_float4(pq[0], pq[1], pq[2], pq[3]));
        }
    }
    wave_sync();
    // ---- the last 62 / 62 / 62 / Nh values: the next chunk's history, or -- at the frame's end -- the next frame's
    if (last) {
        const int h = nv >> 1;
        float *state_save = ldc(&Kp->state[C.par ^ 1]);
        if (lane < kDmHist) {
            *(SDRX_AS1 float *)(state_save + lane) = dm[kDmQO + h + lane];
            *(SDRX_AS1 float *)(state_save + 64 + lane) = dm[kDmQE + 3 + h + lane];
            *(SDRX_AS1 float *)(state_save + 128 + lane) = dm[kDmI + 2 + nv + lane];
        }
        if (lane < C.Nh)
            *(SDRX_AS1 float *)(state_save + 192 + lane) = dm[kDmU + nv + lane];
    } else {
        float a = 0.f, b = 0.f, c = 0.f, d = 0.f;
        if (lane < kDmHist)
            a = dm[kDmQO + 128 + lane], b = dm[kDmQE + 3 + 128 + lane], c = dm[kDmI + 2 + 256 + lane];
        if (lane < C.Nh)
            d = dm[kDmU + 256 + lane];
        wave_sync();
        if (lane < kDmHist)
            dm[kDmQO + lane] = a, dm[kDmQE + 3 + lane] = b, dm[kDmI + 2 + lane] = c;
        if (lane < C.Nh)
            dm[kDmU + lane] = d;
    }
}

// Fused NCO + mixer + half-band cascade.
#ifndef SDRX_K1_MIN_WAVES
#define SDRX_K1_MIN_WAVES 5 // waves per SIMD the register allocator must leave room for
#endif
// The body of one work item, run by ONE wave on LDS of its own (`smem`: k1_lds_bytes()); `level0`: the
// item's VFO is fed by the raw frame (`raw`, `raw_mode`), otherwise by its parent's tile-layout stream.
// DM: the leaf demodulates its stream in this very wave (demod_chunk; DEPTH == 2 only) -- decimate[2] itself is then written only
// where it is wanted (K1Vfo::tap: the spectrum tap, option keep_streams).
template <bool EXACT, int DEPTH, bool ROT, bool DM = false>
__device__ __forceinline__ void mix_item(const K1Vfo *__restrict__ vfos, const K1Work W, unsigned long long frame_no,
                                         const void *__restrict__ raw, int raw_mode, bool level0_arg, unsigned char *smem, int lane)
{
    v2f *car0 = reinterpret_cast<v2f *>(smem);             // [8]
    v2f *car1 = car0 + 8;                                  // [8]
    v2f *lds = reinterpret_cast<v2f *>(smem + kCarryBytes);

    const int par = (int)(frame_no & 1ull);
    const K1Vfo *Dp = vfos + W.vfo;
    struct {
        const float2 *cp;
        int n_in, d, L, out_tiled;
    } D = {ldc(&Dp->cp), ldc(&Dp->n_in), ldc(&Dp->d), ldc(&Dp->L), ldc(&Dp->out_tiled)};
    // DEPTH >= 0: the item is a LEAF with that many half-band stages below a parent (the shapes the time goes into: the
    // reference's sub VFOs, 5 and 2 stages) -- depth, input form and output form are compile-time constants, every branch on
    // them folds, and the chunk loop keeps only what that shape needs in registers.  DEPTH < 0: any VFO (run-time fields).
    constexpr bool kShape = DEPTH >= 0;
    const int dd = kShape ? DEPTH : D.d;
    const bool level0 = kShape ? false : level0_arg;
    const int otiled = kShape ? 0 : D.out_tiled;
    const v2f rot = {ldc(&Dp->rot_re), ldc(&Dp->rot_im)};
    const float4 *in = reinterpret_cast<const float4 *>(ldc(&Dp->in[par]));
    float2 *out = ldc(&Dp->out[par]);
    const float2 *hb_load = ldc(&Dp->hb[par]);
    float2 *hb_save = ldc(&Dp->hb[par ^ 1]);
    const bool from_state = W.s_begin == 0;
    const v2f zero2 = {0.f, 0.f};

    // Filter state at the start of this segment: segment 0 continues from the previous frame's
    // history (zero at start-up, dsp.cpp:40-49); a later segment starts from zeros and runs
    // warm-up chunks until every stage's window holds real samples again.
    if (lane < 8) {
        const int k = halo_k(lane);
        car0[lane] = (from_state && dd > 0) ? gldv2(hb_load + 0 * kHbHist + k - 1) : zero2;
        car1[lane] = (from_state && dd > 1) ? gldv2(hb_load + 1 * kHbHist + k - 1) : zero2;
    }
    for (int s = kRegStages; s < dd; ++s)
        if (lane < kCarry) {
            const int k = kCarry - lane; // carry position `lane` is x[-k]
            lds[stage_offset(s) + lane] = (from_state && k <= kHbHist) ? gldv2(hb_load + s * kHbHist + k - 1) : zero2;
        }
    static_assert(!DM || DEPTH == 2, "the in-wave demodulation is laid out for 256 stream samples per chunk");
    float *dm = reinterpret_cast<float *>(lds); // (DM: the demodulation's arrays live where a d = 2 leaf would park its held stores)
    DemodCtx dmc;
    if constexpr (DM)
        demod_prologue(dm, ldc(&Dp->dm), par, from_state, lane, dmc);
    const int phase_frame = (int)((frame_no * (unsigned long long)D.n_in) % (unsigned long long)D.L);

    // The item walks 1024-sample chunks from sample s_begin (any multiple of 16: a chunk need not
    // coincide with a tile of the input) and emits the outputs whose input position is >= s_first_out.
    const int first_out = W.s_first_out;
    const int lane16 = (W.s_begin >> 4) + lane; // this lane's run in the item's first chunk, in units of 16 samples
    // the lane's first 16-byte unit of the item as a 32-bit index (a frame has < 2^20 units): the loads below then take the
    // uniform stream pointer from SGPRs and one 32-bit VGPR offset instead of keeping a 64-bit pointer alive per lane
    const unsigned unit_item = (unsigned)(lane16 >> 6) * 512u + (unsigned)(lane16 & 63);
    // this lane's NCO checkpoint for the chunk at `at`: cp[idx >> 4], idx = table position of the lane's first sample
    auto cp_of = [&](int at) {
        int ix = phase_frame + at; // both < L
        ix -= ix >= D.L ? D.L : 0;
        ix += lane * kRun;         // L >= kChunk (checked by sdrx_finalize)
        ix -= ix >= D.L ? D.L : 0;
        return D.cp + (ix >> 4);
    };
    // (the shaped bodies only: the any-VFO body has no register to spare -- it requests its checkpoint and stores its outputs
    // where they arise)
    v2f o_next = zero2;                   // the next chunk's checkpoint, requested a chunk ahead: the whole NCO hangs on these 8 bytes
    if constexpr (kShape)
        o_next = gldv2(cp_of(W.s_begin));
    HeldStores held;                      // the previous chunk's output stores (issued behind this chunk's loads)
    held.one = zero2;
    auto flush_held = [&]() {
        if constexpr (DM) { // (nothing is ever held: the payload leaves where it arises, 8 bytes per lane and chunk)
        } else if constexpr (!kShape) { // (the any-VFO body stores at once: an ordinary conditional store)
            if (held.units == 1 && lane >= held.jmin && lane < held.nout)
                gstv2_leaf(out + (size_t)(held.gbase + lane), held.one);
        } else if constexpr (DEPTH == 2) {
            const __amdgpu_buffer_rsrc_t out_rsrc = stream_rsrc(out);
            // (a lane reads back what it wrote itself: no fence needed; before the first chunk the LDS holds anything -- and
            // units is 0, so nothing is stored)
            const bool mine = held.units == 2 && (((held.base >> 4) + lane) << 4) >= first_out && lane <= held.lv;
            const unsigned off = mine ? 8u * (unsigned)((held.base >> 2) + lane * 4) : kNoStore;
            const v4f *park = reinterpret_cast<const v4f *>(lds) + 2 * lane;
            bst4(out_rsrc, off, park[0]);
            bst4(out_rsrc, off + 16u, park[1]);
        } else {
            const __amdgpu_buffer_rsrc_t out_rsrc = stream_rsrc(out);
            // outputs below jmin belong to the warm-up of a segment that starts inside the frame
            const bool mine = held.units == 1 && lane >= held.jmin && lane < held.nout;
            bst2(out_rsrc, mine ? 8u * (unsigned)(held.gbase + lane) : kNoStore, held.one);
        }
        held.units = 0;
    };
    int pair_base = -1; // >= 0: the stage-2 outputs of the chunk at this position wait in A_3 for the next chunk's (SDRX_PAIR_STAGES)
    for (int base = W.s_begin; base < W.s_end; base += kChunk) {
        const int valid = min(kChunk, D.n_in - base);
        const int p16 = (base >> 4) + lane;               // this lane's run, in units of 16 samples
        const bool emit = base + kChunk > first_out;      // the chunk reaches into the emitted range
        const bool emit_l = (p16 << 4) >= first_out;      // ... and this lane's run lies inside it
        const bool save = base + valid == D.n_in;         // the chunk that holds the frame's last sample
        const int lv = (valid >> 4) - 1; // last lane holding real samples
        const bool active = lane <= lv;

        // 1. this lane's run of 16 consecutive samples: 8 coalesced 16-byte loads
        v2f ext0[10 + kRun]; // ext0[10 + t] = x[t]; ext0[0..9] = halo x[-10..-1]
        v2f *x = ext0 + 10;
        if (level0 && raw_mode != kRawTiled) {
            load_run_raw(raw, raw_mode, p16, active, x);
        } else {
            // The lane's position inside its tile is the same in every chunk of the item (the walk advances by exactly one tile
            // per chunk): unit_item is computed once, a chunk adds 512 units.  A shifted walk straddles two tiles; its idle lanes
            // in the frame's last chunk may read the (zero) tile behind the last one, which every tile-layout buffer has.
            load_run_tile(in + (unit_item + (unsigned)((base - W.s_begin) >> 10) * 512u), x);
        }

        // 2. NCO and mixer (nco_mix).  The shaped bodies have this chunk's checkpoint already and request the next one.
        v2f o;
        if constexpr (kShape) {
            o = o_next; // (its load was issued a chunk ago)
            o_next = gldv2(cp_of(base + kChunk < W.s_end ? base + kChunk : base)); // (always issued -- and therefore counted, like the stores)
            flush_held(); // the previous chunk's outputs leave BEHIND this chunk's loads: nothing ever waits for them
        } else {
            o = gldv2(cp_of(base));
        }
        {
            int i0 = phase_frame + base;
            i0 -= i0 >= D.L ? D.L : 0;
            nco_mix<EXACT, ROT>(o, rot, Dp->rk, D.cp + (D.L >> 4), frame_no == 0 && base == 0 && lane == 0, i0, kChunk, D.L, x);
        }

        if (dd == 0) {
            // no decimation: decimate[0] is the mixed stream itself
            if (otiled) {
                if (emit_l && active) {
                    float4 *o4 = reinterpret_cast<float4 *>(out);
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        gstv4(o4 + tile_unit(p16 >> 6, i, p16 & 63), cat2(x[2 * i], x[2 * i + 1]));
                }
            } else {
                // natural order wanted: transpose through LDS so the stores are coalesced
                wave_sync();
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    *reinterpret_cast<v4f *>(lds + pad0(lane * kRun + 2 * i)) = cat2(x[2 * i], x[2 * i + 1]);
                wave_sync();
                if (emit) {
                    // pad0(128 i + 2 lane) = 144 i + pad0(2 lane): one base address, constant offsets
                    const v2f *src = lds + pad0(2 * lane);
                    // (a 32-bit sample index per lane on top of the uniform stream pointer: a 64-bit per-lane pointer kept
                    // alive across the chunk loop was what the register allocator spilled)
                    const unsigned at = (unsigned)base + 2u * (unsigned)lane;
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        if (128 * i + 2 * lane < valid && base + 128 * i + 2 * lane >= first_out)
                            gstv4_leaf(reinterpret_cast<float4 *>(out + (at + 128u * i)), *reinterpret_cast<const v4f *>(src + 144 * i));
                }
            }
            continue;
        }

        // 3. stage 0 in registers
        halo_stage0(car0, ext0, lane);
        v2f ext1[10 + 8]; // ext1[10 + t] = y[t] (stage-0 outputs of this lane), ext1[0..9] halo
        v2f *y = ext1 + 10;
        hb_regs<EXACT, 8>(ext0, y);
        if (save && lane == lv) // next frame's stage-0 history: x[size-1-k], k = 1..10
#pragma unroll
            for (int k = 1; k <= kHbHist; ++k)
                gstv2(hb_save + 0 * kHbHist + k - 1, x[15 - k]);

        if (dd == 1) {
            if (emit_l && active) {
                const int g = (base >> 1) + lane * 8;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const size_t pos = otiled ? tile_pos(g + 2 * i) : (size_t)(g + 2 * i);
                    gstv4(reinterpret_cast<float4 *>(out + pos), cat2(y[2 * i], y[2 * i + 1]));
                }
            }
            continue;
        }

        // 4. stage 1 in registers
        halo_stage1(car1, ext1, lane);
        if (save) { // next frame's stage-1 history: y[size1-1-k]
            if (lane == lv)
#pragma unroll
                for (int k = 1; k <= 7; ++k)
                    gstv2(hb_save + 1 * kHbHist + k - 1, y[7 - k]);
            if (lane == lv - 1)
#pragma unroll
                for (int k = 8; k <= kHbHist; ++k)
                    gstv2(hb_save + 1 * kHbHist + k - 1, y[15 - k]);
        }
        v2f z[4];
        hb_regs<EXACT, 4>(ext1, z);

        if (dd == 2) {
            if (emit_l && active) {
                const int g = (base >> 2) + lane * 4;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    if (otiled)
                        gstv4(reinterpret_cast<float4 *>(out + tile_pos(g + 2 * i)), cat2(z[2 * i], z[2 * i + 1]));
                }
            }
            if constexpr (DM) {
                float2 *dm_tap = ldc(&Dp->tap[par]);
                if (dm_tap && emit_l && active) { // decimate[2] is wanted too (fftVFOSlot, parity tests): natural order
                    const int g = (base >> 2) + lane * 4;
                    gstv4(reinterpret_cast<float4 *>(dm_tap + (size_t)g), cat2(z[0], z[1]));
                    gstv4(reinterpret_cast<float4 *>(dm_tap + (size_t)(g + 2)), cat2(z[2], z[3]));
                }
                demod_chunk<EXACT>(dm, dmc, z, lane, valid >> 2, base >> 2, first_out >> 2, save);
            } else if constexpr (kShape) { // its four outputs wait (in LDS) for the next chunk's loads to be on their way
                v4f *park = reinterpret_cast<v4f *>(lds) + 2 * lane;
                park[0] = cat2(z[0], z[1]);
                park[1] = cat2(z[2], z[3]);
                held.units = 2, held.base = base, held.lv = lv;
            } else if (!otiled && emit_l && active) {
                const int g = (base >> 2) + lane * 4;
                gstv4_leaf(reinterpret_cast<float4 *>(out + (size_t)g), cat2(z[0], z[1]));
                gstv4_leaf(reinterpret_cast<float4 *>(out + (size_t)(g + 2)), cat2(z[2], z[3]));
            }
            continue;
        }

        // 5. stages >= 2 in LDS: this lane's 4 stage-2 inputs go to A_2, then the generic stage
        {
            v2f *A2 = lds + stage_offset(2) + kCarry + lane * 4;
            *reinterpret_cast<v4f *>(A2) = cat2(z[0], z[1]);
            *reinterpret_cast<v4f *>(A2 + 2) = cat2(z[2], z[3]);
        }
#if SDRX_FIXED_STAGES
        if (kShape && valid == kChunk && !save && dd == kFixedDepth) { // (uniform) a full chunk, not the frame's last (the shaped body of a d = 5 leaf only)
#if SDRX_PAIR_STAGES
            // stage 2 every chunk; stages 3 and 4 every second chunk on both chunks' stage-2 outputs -- when the NEXT chunk
            // of this item is such a chunk too (otherwise this one is finished alone: nothing pending ever meets another path)
            if (pair_base < 0 && base + kChunk < W.s_end && base + 2 * kChunk < D.n_in) {
                hb_stage_fixed1<EXACT, 2, kFixedDepth, 1>(lds, held, 0, 0, lane);
                pair_base = base;
                continue;
            }
            if (pair_base >= 0) {
                hb_stage_fixed1<EXACT, 2, kFixedDepth, 1, (kChunk >> 3)>(lds, held, 0, 0, lane);
                // outputs below jmin belong to the warm-up of a segment that starts inside the frame
                hb_stage_fixed<EXACT, 3, kFixedDepth, 2>(lds, held, pair_base >> dd, max(0, (first_out - pair_base) >> dd), lane);
                pair_base = -1;
                if constexpr (!kShape)
                    flush_held();
                continue;
            }
#endif
            const int jmin = max(0, (first_out - base) >> dd), gbase = base >> dd;
            hb_stage_fixed<EXACT, 2, kFixedDepth, 1>(lds, held, gbase, jmin, lane);
            if constexpr (!kShape)
                flush_held();
            continue;
        }
#endif
        for (int s = kRegStages; s < dd; ++s) {
            hb_stage_lds<EXACT>(lds + stage_offset(s), lds + stage_offset(s + 1), out, base >> dd, otiled != 0, s + 1 == dd,
                                max(0, (first_out - base) >> dd), valid >> s, lane, save, hb_save + s * kHbHist);
        }
    }
    flush_held(); // the last chunk's outputs
}

// ------------------------------------------------------------------------------------ late_item
// vfo::usb_decimdemod's first half (vfo.cpp:334-356) for a d = 0 leaf, INSIDE the mix wave: NCO + mixer as in mix_item,
// then the decimating low-pass on the mixed samples while they are still on the CU,
//     z'[k] = sum_{t < Nd} hd[t] * x[L k - Nd + t]         (FIR::FIRUpdateAndProcess: the (Nd+1)-slot ring leaves the newest
//                                                            sample x[L k] out, dsp.cpp:59-71; FIRUpdate for the samples
//                                                            in between, dsp.cpp:150-154; the phase counter restarts with
//                                                            every frame, vfo.cpp:338-339, and frames are multiples of L)
// so that only z' (1 / L of the bytes) goes to HBM: the two-kernel form wrote the mixed 240 kS/s stream for
// k_late_decimate4 to read back, 246 MB per frame on BASELINE config 4.  One accumulator per output and component, taps in
// ascending order, every product and sum rounded (EXACT): the reference's summation, bit for bit.
//
// A chunk is LateGeom<L>::kChunkLen samples (960 | 1008): kMixLanes lanes replay the NCO and mix 16 consecutive samples each,
// write them to LDS as rows of 3 L samples (row stride kStride: the b64 reads below are conflict-free), and lane l then
// computes the three outputs k = base / L + 3 l + r from ONE pass over the 3 L - 1 + Nd samples that end in its row: window
// sample u of lane l sits at l * kStride + const(u), the taps are wave-uniform scalars.  The last kCarryRows rows of a chunk
// are the next chunk's history; the frame's last late_hist<L>() mixed samples are the next frame's (hb[]).
template <bool EXACT, int LD, bool ROT>
__device__ __forceinline__ void late_item(const K1Vfo *__restrict__ vfos, const K1Work W, unsigned long long frame_no, unsigned char *smem,
                                          int lane)
{
    using G = LateGeom<LD>;
    constexpr int N = G::kTaps, kRow = G::kRow, kStride = G::kStride, kHc = late_hist<LD>();
    constexpr int kCarryElems = G::kCarryRows * kStride;
    v2f *buf = reinterpret_cast<v2f *>(smem); // [(kCarryRows + kRows) * kStride]; sample p of the chunk (p >= -kHc) sits in row (p + kHc) / kRow
    float *sh = reinterpret_cast<float *>(smem + late_window_bytes<LD>()); // the taps, zero-padded to kLateTapPad floats
    const int par = (int)(frame_no & 1ull);
    const K1Vfo *Dp = vfos + W.vfo;
    struct {
        const float2 *cp;
        int n_in, L;
    } D = {ldc(&Dp->cp), ldc(&Dp->n_in), ldc(&Dp->L)};
    const v2f rot = {ldc(&Dp->rot_re), ldc(&Dp->rot_im)};
    const float4 *in = reinterpret_cast<const float4 *>(ldc(&Dp->in[par]));
    float2 *zout = ldc(&Dp->out[par]);
    float2 *tap = ldc(&Dp->tap[par]);
    const float2 *hist_load = ldc(&Dp->hb[par]);
    float2 *hist_save = ldc(&Dp->hb[par ^ 1]);
    const float *taps = ldc(&Dp->late_taps);
    const v2f zero2 = {0.f, 0.f};

    // history in front of the segment's first sample: the previous frame's last kHc mixed samples (zero at start-up,
    // dsp.cpp:40-49), or zeros for a segment that starts inside the frame (its first kWarm samples are warm-up)
    for (int t = lane; t < kHc; t += 64)
        buf[(t / kRow) * kStride + t % kRow] = W.s_begin == 0 ? gldv2(hist_load + t) : zero2;
    for (int t = lane; t < kLateTapPad; t += 64)
        sh[t] = t < N ? gld(taps + t) : 0.f;
    // where this lane's 16-sample run goes: it starts in row q0 at column r0 and crosses into row q0 + 1 at sample wr_T
    const int q0 = (16 * lane) / kRow, r0 = 16 * lane - kRow * q0;
    const int wr_lo = (G::kCarryRows + q0) * kStride + r0;
    int wr_T = kRow - r0;
    const v2f *rd = buf + lane * kStride; // window sample u of this lane: rd[(kCarryRows + floor(u / kRow)) * kStride + u mod kRow]
    const int phase_frame = (int)((frame_no * (unsigned long long)D.n_in) % (unsigned long long)D.L);
    // As in mix_item: the checkpoint the whole NCO hangs on is requested a chunk ahead (always issued: a load the compiler can
    // count).  (Holding the chunk's outputs back behind the next chunk's loads, as mix_item's shaped bodies do, measured
    // +2 % on config 4 here -- the parking in the window rows' padding and the three buffer stores cost more than the
    // waits they avoid -- so this body stores where the outputs arise.)
    auto cp_of = [&](int at) {
        int ix = phase_frame + at; // both < L
        ix -= ix >= D.L ? D.L : 0;
        ix += lane * kRun;
        ix -= ix >= D.L ? D.L : 0;
        return D.cp + (ix >> 4);
    };
    v2f o_next = gldv2(cp_of(W.s_begin));
    int kb = W.s_begin / LD; // index of the first output of the chunk (s_begin is a multiple of L)
    for (int base = W.s_begin; base < W.s_end; base += G::kChunkLen, kb += G::kChunkLen / LD) {
        const int valid = min(G::kChunkLen, D.n_in - base);
        // 1. this lane's run of 16 consecutive samples out of the parent's tile-layout stream.  (Lanes past the chunk or
        //    the frame read on into the stream's spare tile: finite values nobody uses.)
        const unsigned run = (unsigned)(base >> 4) + (unsigned)lane;
        const float4 *src = in + ((run >> 6) * 512u + (run & 63u));
        v2f x[kRun];
        load_run_tile(src, x);
        // 2. NCO and mixer, exactly as in mix_item
        const v2f o = o_next; // (its load was issued a chunk ago)
        o_next = gldv2(cp_of(base + G::kChunkLen < W.s_end ? base + G::kChunkLen : base)); // (always issued, hence counted)
        {
            int i0 = phase_frame + base;
            i0 -= i0 >= D.L ? D.L : 0;
            // (kChunk, not kChunkLen: the lanes past the chunk replay entries nobody uses -- up to 64 x 16 of them must not wrap either)
            nco_mix<EXACT, ROT>(o, rot, Dp->rk, D.cp + (D.L >> 4), frame_no == 0 && base == 0 && lane == 0, i0, kChunk, D.L, x);
        }
        if (tap && lane < G::kMixLanes && 16 * lane < valid && base + 16 * lane >= W.s_first_out) {
            // decimate[0] is wanted (the GUI's spectrum tap, parity tests)
            float4 *t4 = reinterpret_cast<float4 *>(tap + (size_t)(base + 16 * lane));
#pragma unroll
            for (int i = 0; i < 8; ++i)
                gstv4(t4 + i, cat2(x[2 * i], x[2 * i + 1]));
        }
        // 3. to LDS, rows of 3 L samples
        wave_sync(); // the previous chunk's window reads and its carry rows are done
        if (LD == 6 || !EXACT) // (16 selects per chunk instead of 16 addresses kept across the loop -- there they spill; exact /5 keeps them)
            asm volatile("" : "+v"(wr_T));
        if (lane < G::kMixLanes) {
#pragma unroll
            for (int i = 0; i < kRun; ++i)
                buf[wr_lo + i + (i >= wr_T ? kStride - kRow : 0)] = x[i];
        }
        wave_sync();
        // 4. three outputs per lane from one pass over the window, a row of it at a time.  The taps come as broadcast b128
        //    reads (the same address in every lane): as scalars they would be SGPR PAIRS (a packed instruction's scalar
        //    operand is 64 bits wide), twice the SGPR file.  Window samples are read two at a time (b128) where the row
        //    stride keeps pairs 16-byte aligned; single b64 reads otherwise.
        if (lane < G::kRows) {
            v2f acc[3] = {zero2, zero2, zero2};
            static_for<G::kCarryRows + 1>([&](auto row_ic) {
                constexpr int fl = decltype(row_ic)::value - G::kCarryRows; // window row, -kCarryRows .. 0
                constexpr int u_lo = kRow * fl < -N ? -N : kRow * fl, u_hi = kRow * (fl + 1) > 2 * LD ? 2 * LD : kRow * (fl + 1);
                // taps this row touches: t = u - L r + N for u in [u_lo, u_hi), r in 0..2
                constexpr int t_lo = (u_lo + N - 2 * LD < 0 ? 0 : u_lo + N - 2 * LD) & ~3;
                constexpr int kHv = (kRow + 2 * LD + 8) / 2;
                v2f hv[kHv]; // tap PAIRS (t_lo + 2 q, + 1): a packed multiply takes either half for both of its lanes
#pragma unroll
                for (int g = 0; g < kHv / 2; ++g)
                    if (t_lo + 4 * g < N && t_lo + 4 * g < u_hi + N) {
                        const v4f q = *reinterpret_cast<const v4f *>(sh + t_lo + 4 * g);
                        hv[2 * g] = lo2(q), hv[2 * g + 1] = hi2(q);
                    }
                const v2f *row = rd + (G::kCarryRows + fl) * kStride;
                v2f w[kRow];
#pragma unroll
                for (int col = 0; col < kRow; ++col) {
                    const int u = kRow * fl + col;
                    if (u < u_lo || u >= u_hi)
                        continue;
                    const bool pairs = kStride % 2 == 0;
                    if (pairs && col % 2 == 0 && col + 1 < kRow && u + 1 < u_hi) {
                        const v4f q = *reinterpret_cast<const v4f *>(row + col);
                        w[col] = lo2(q), w[col + 1] = hi2(q);
                    } else if (!(pairs && col % 2 == 1 && u - 1 >= u_lo)) {
                        w[col] = row[col];
                    }
                }
                static_for<kRow>([&](auto col_ic) {
                    constexpr int col = decltype(col_ic)::value, u = kRow * fl + col;
                    if constexpr (u >= u_lo && u < u_hi) {
                        // output r takes tap t_r = u - L r + N of this sample if 0 <= t_r < N: r in [r_lo, r_hi]
                        constexpr int i0 = u + N - t_lo, i1 = i0 - LD, i2 = i0 - 2 * LD; // tap index - t_lo per output
                        constexpr bool on0 = u + N >= 0 && u + N < N, on1 = u - LD + N >= 0 && u - LD + N < N, on2 = u - 2 * LD + N >= 0 && u - 2 * LD + N < N;
                        if constexpr (on0 && on1 && on2)
                            fir_mac3<EXACT, i0 & 1, i1 & 1, i2 & 1>(acc[0], acc[1], acc[2], hv[i0 >> 1], hv[i1 >> 1], hv[i2 >> 1], w[col]);
                        else if constexpr (on0 && on1)
                            fir_mac2<EXACT, i0 & 1, i1 & 1>(acc[0], acc[1], hv[i0 >> 1], hv[i1 >> 1], w[col]);
                        else if constexpr (on1 && on2)
                            fir_mac2<EXACT, i1 & 1, i2 & 1>(acc[1], acc[2], hv[i1 >> 1], hv[i2 >> 1], w[col]);
                        else if constexpr (on0)
                            fir_mac1<EXACT, i0 & 1>(acc[0], hv[i0 >> 1], w[col]);
                        else if constexpr (on2)
                            fir_mac1<EXACT, i2 & 1>(acc[2], hv[i2 >> 1], w[col]);
                        else
                            static_assert(!on1, "a window sample feeds a contiguous range of outputs");
                    }
                });
                __builtin_amdgcn_sched_barrier(0); // a row's reads stay in the row (all hoisted to the top they spill)
            });
            const int pos = base + kRow * lane; // input position L k of the lane's first output
#pragma unroll
            for (int r = 0; r < 3; ++r)
                if (pos + LD * r >= W.s_first_out && kRow * lane + LD * r < valid)
                    gstv2_leaf(zout + (size_t)(kb + 3 * lane + r), acc[r]);
        }
        wave_sync();
        if (base + valid == D.n_in) {
            // the frame's last kHc mixed samples are the next frame's history (sample valid - kHc + t of this chunk)
            int lane_here = lane; // (the per-lane address is computed HERE: hoisted above the chunk loop it is one live 64-bit value too many)
            asm volatile("" : "+v"(lane_here));
            for (int t = lane_here; t < kHc; t += 64) {
                const int pp = valid + t;
                gstv2(hist_save + t, buf[(pp / kRow) * kStride + pp % kRow]);
            }
        } else {
            // the last kCarryRows rows become the next chunk's history rows
            v2f c0 = zero2, c1 = zero2;
            if (lane < kCarryElems)
                c0 = buf[G::kRows * kStride + lane];
            if (kCarryElems > 64 && lane + 64 < kCarryElems)
                c1 = buf[G::kRows * kStride + lane + 64];
            wave_sync();
            if (lane < kCarryElems)
                buf[lane] = c0;
            if (kCarryElems > 64 && lane + 64 < kCarryElems)
                buf[lane + 64] = c1;
        }
    }
}

// which body a work item runs: the fused late decimation for the leaves marked so at finalize, the half-band cascade otherwise
template <bool EXACT, bool ROT>
__device__ __forceinline__ void run_item(const K1Vfo *__restrict__ vfos, const K1Work W, unsigned long long frame_no,
                                         const void *__restrict__ raw, int raw_mode, bool level0, unsigned char *smem, int lane)
{
    const int late = ldc(&vfos[W.vfo].late_L);
    if (late == 5)
        late_item<EXACT, 5, ROT>(vfos, W, frame_no, smem, lane);
    else if (late == 6)
        late_item<EXACT, 6, ROT>(vfos, W, frame_no, smem, lane);
    else {
        // a leaf fed from a tile-layout stream: the two shapes of the reference's sub VFOs have bodies of their own (0: any VFO)
        const bool leaf_on_tiles = !(level0 && raw_mode != kRawTiled) && !ldc(&vfos[W.vfo].out_tiled);
        const int d = ldc(&vfos[W.vfo].d);
        const int shape = leaf_on_tiles && d == kFixedDepth ? kFixedDepth : leaf_on_tiles && d == 2 ? 2 : 0;
        if (shape == 2 && ldc(&vfos[W.vfo].dm) != nullptr) // (set at finalize for exactly such leaves: option fuse_demod)
            mix_item<EXACT, 2, ROT, true>(vfos, W, frame_no, raw, raw_mode, false, smem, lane);
        else if (shape == kFixedDepth)
            mix_item<EXACT, kFixedDepth, ROT>(vfos, W, frame_no, raw, raw_mode, false, smem, lane);
        else if (shape == 2)
            mix_item<EXACT, 2, ROT>(vfos, W, frame_no, raw, raw_mode, false, smem, lane);
        else
            mix_item<EXACT, -1, ROT>(vfos, W, frame_no, raw, raw_mode, level0, smem, lane);
    }
}

// One wave per workgroup, one workgroup per K1Work.  LEVEL only gives the root launch and the sub
// launches distinct kernel names in profiles.
template <bool EXACT, int LEVEL, bool ROT = !EXACT>
__global__ __launch_bounds__(64, SDRX_K1_MIN_WAVES) void k_mix_decimate(const K1Vfo *__restrict__ vfos, const K1Work *__restrict__ work,
                                                     unsigned long long frame_no, const void *__restrict__ raw, int raw_mode)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    run_item<EXACT, ROT>(vfos, work[blockIdx.x], frame_no, raw, raw_mode, LEVEL == 0, smem, (int)threadIdx.x);
}

// ------------------------------------------------------------------------------------ demod tail (kernels of its own)
// Late decimation by L in {5,6} (vfo.cpp:334-387 with FIR::FIRUpdateAndProcess/FIRUpdate,
// dsp.cpp:59-71,150-154): z'[k] = sum_i hd[i] * x[L k - Nd + i]  -- the newest sample x[L k] is
// NOT part of the sum ((N+1)-slot ring).  The phase counter restarts every frame and frames are
// multiples of L, so k is frame-local.  256 outputs per block; the input window sits in LDS.
template <bool EXACT>
__global__ __launch_bounds__(256) void k_late_decimate(const K2aVfo *__restrict__ vfos, const BlockWork *__restrict__ work,
                                                       unsigned long long frame_no)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float2 *sx = reinterpret_cast<float2 *>(smem);
    const BlockWork bw = work[blockIdx.x];
    const K2aVfo *Dp = vfos + bw.vfo;
    const int blk = bw.blk;
    const int par = (int)(frame_no & 1ull);
    const int tid = threadIdx.x;
    struct {
        const float *taps;
        int Hx, n, ndec, L, n_out;
    } D = {Dp->taps, Dp->Hx, Dp->n, Dp->ndec, Dp->L, Dp->n_out};
    const float2 *xbase = Dp->x[par];
    float2 *xnext = Dp->x_next[par];
    float2 *zout = Dp->z[par];
    const float2 *x = xbase + D.Hx; // sample 0 of this frame
    const int k0 = blk * 256;
    if (k0 >= D.n_out && blk != 0)
        return;
    // history for the next frame: the last Hx entries of [hist | data]
    if (blk == 0)
        for (int j = tid; j < D.Hx; j += 256)
            gst2(xnext + j, gld2(xbase + D.n + j));
    const int lo = D.L * k0 - D.ndec;              // first input index needed (>= -Hx)
    const int span = D.L * 255 + D.ndec;           // indices lo .. lo+span-1 feed the 256 outputs
    for (int t = tid; t < span; t += 256) {
        const int idx = lo + t;
        sx[t] = idx < D.n ? gld2(x + idx) : make_float2(0.f, 0.f);
    }
    __syncthreads();
    const int k = k0 + tid;
    if (k >= D.n_out)
        return;
    const float2 *w = sx + D.L * tid; // w[i] = x[L k - Nd + i]
    float ar = 0.f, ai = 0.f;
    if (EXACT) {
        for (int i = 0; i < D.ndec; ++i) {
            const float h = gld(D.taps + i);
            ar = ar + h * w[i].x;
            ai = ai + h * w[i].y;
        }
    } else {
        for (int i = 0; i < D.ndec; ++i) {
            const float h = gld(D.taps + i);
            ar = fmaf(h, w[i].x, ar);
            ai = fmaf(h, w[i].y, ai);
        }
    }
    gst2(zout + k, make_float2(ar, ai));
}

// The same for L in {5,6} (all the reference configures, mainwindow.cpp:196-216) and Nd <= 96: ONE
// wave per 64 R outputs, R consecutive outputs per lane.  The R windows of a lane overlap
// (Nd + (R-1) L samples instead of R Nd), so a lane streams ITS window once (two ds_read_b128 per
// 4 samples) and feeds R accumulator pairs; output r uses tap j - L r for window sample j, read
// as aligned ds_read_b128 from a copy of the taps shifted by L r (zero outside [0, Nd): adding
// 0*x never changes a float sum).  The lane stride in LDS is padded to an odd multiple of 16
// bytes so that 8 lanes' b128 reads cover all 32 banks.  Same summation order as the reference:
// one accumulator per output and component, taps in ascending order.
constexpr int kLateMaxTaps = 96;
constexpr int kLateTapRow = 124; // shifted tap copies: u = i + L r < Nd + 3 L = 114 (+ b128 overrun), a multiple of 4
__host__ __device__ constexpr int late4_pad(int R, int L) { return (2 * R * L) % 8 == 4 ? 0 : 2; } // float2 per R L samples
__host__ __device__ constexpr int late4_lds_bytes(int R, int L, int ndec) // for the launch's largest L and Nd
{
    const int span = L * (64 * R - 1) + ndec + (R - 1) * L;
    return 4 * R * kLateTapRow + 8 * (span + 2 * (span / (R * L)) + 16);
}

template <bool EXACT, int L, int R>
__device__ __forceinline__ void late4_body(const float2 *__restrict__ x, float2 *__restrict__ zout, const float *__restrict__ taps, int n, int ndec,
                                           int n_out, int k0, int lane, v2f *__restrict__ sx, float *__restrict__ sh)
{
    constexpr int kPad = late4_pad(R, L);
    constexpr int kStride = R * L + kPad;                                          // float2 per lane
    constexpr int kIters = (L * (64 * R - 1) + kLateMaxTaps + (R - 1) * L + 8 + 63) / 64; // window loads per lane (+8: zero tail)
    constexpr int kTapIters = (R * kLateTapRow + 63) / 64;
    const int lo = L * k0 - ndec;                       // first input index of the tile's window (>= -Hx)
    const int span = L * (64 * R - 1) + ndec + (R - 1) * L; // lanes' windows end at lo + R L lane + ndec + (R-1) L
    // every load of the lane is issued before the first LDS write (see k_usb_demod)
    v2f stage[kIters];
#pragma unroll
    for (int it = 0; it < kIters; ++it) {
        const int t = lane + 64 * it, idx = lo + t;
        const v2f z2 = {0.f, 0.f};
        stage[it] = (t < span && idx < n) ? gldv2(x + idx) : z2;
    }
    float hst[kTapIters];
#pragma unroll
    for (int it = 0; it < kTapIters; ++it) {
        const int u = lane + 64 * it;
        const int r = u / kLateTapRow, i = u - r * kLateTapRow - L * r;
        hst[it] = (u < R * kLateTapRow && i >= 0 && i < ndec) ? gld(taps + i) : 0.f;
    }
#pragma unroll
    for (int it = 0; it < kIters; ++it) {
        const int t = lane + 64 * it;
        if (t < span + 8) // the entries just past the span are read under zero taps only: they must be finite
            sx[t + kPad * (t / (R * L))] = stage[it];
    }
#pragma unroll
    for (int it = 0; it < kTapIters; ++it) {
        const int u = lane + 64 * it;
        if (u < R * kLateTapRow)
            sh[u] = hst[it];
    }
    wave_sync();
    const int k = k0 + R * lane;
    if (k >= n_out)
        return;
    const v2f *w = sx + kStride * lane; // window sample j sits at w[j + kPad * (j / (R L))]
    v2f acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r)
        acc[r] = v2f{0.f, 0.f};
    const int groups = (ndec + (R - 1) * L + 3) / 4;
    for (int g = 0; g < groups; ++g) {
        const int j = 4 * g;
        const v4f a = *reinterpret_cast<const v4f *>(w + j + kPad * (j / (R * L)));
        const v4f b = *reinterpret_cast<const v4f *>(w + j + 2 + kPad * ((j + 2) / (R * L)));
        const v2f xs[4] = {lo2(a), hi2(a), lo2(b), hi2(b)};
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const v4f h4 = *reinterpret_cast<const v4f *>(sh + r * kLateTapRow + j);
            const float h[4] = {h4.x, h4.y, h4.z, h4.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const v2f hh = {h[e], h[e]};
                if (EXACT)
                    acc[r] = acc[r] + hh * xs[e];
                else
                    acc[r] = __builtin_elementwise_fma(hh, xs[e], acc[r]);
            }
        }
    }
    if (k + R - 1 < n_out) {
#pragma unroll
        for (int r = 0; r < R; r += 2)
            gstv4(reinterpret_cast<float4 *>(zout + k + r), cat2(acc[r], acc[r + 1]));
    } else {
        for (int r = 0; r < R && k + r < n_out; ++r)
            gstv2(zout + k + r, acc[r]);
    }
}

#ifndef SDRX_LATE4_WAVES
#define SDRX_LATE4_WAVES 6 // 76 VGPRs; measured 34.0 us vs 35.4 (5 waves) and 35.5 (8 waves) on config 4
#endif
template <bool EXACT, int R>
__global__ __launch_bounds__(64, SDRX_LATE4_WAVES) void k_late_decimate4(const K2aVfo *__restrict__ vfos, const BlockWork *__restrict__ work,
                                                       unsigned long long frame_no)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *sh = reinterpret_cast<float *>(smem);
    v2f *sx = reinterpret_cast<v2f *>(smem + 4 * R * kLateTapRow);
    const BlockWork bw = work[blockIdx.x];
    const K2aVfo *Dp = vfos + bw.vfo;
    const int par = (int)(frame_no & 1ull);
    const int lane = threadIdx.x;
    struct {
        const float *taps;
        int Hx, n, ndec, L, n_out;
    } D = {Dp->taps, Dp->Hx, Dp->n, Dp->ndec, Dp->L, Dp->n_out};
    const float2 *xbase = Dp->x[par];
    float2 *xnext = Dp->x_next[par];
    const int k0 = bw.blk * 64 * R;
    if (k0 >= D.n_out && bw.blk != 0)
        return;
    if (bw.blk == 0) // history for the next frame: the last Hx entries of [hist | data]
        for (int j = lane; j < D.Hx; j += 64)
            gst2(xnext + j, gld2(xbase + D.n + j));
    if (D.L == 5)
        late4_body<EXACT, 5, R>(xbase + D.Hx, Dp->z[par], D.taps, D.n, D.ndec, D.n_out, k0, lane, sx, sh);
    else
        late4_body<EXACT, 6, R>(xbase + D.Hx, Dp->z[par], D.taps, D.n, D.ndec, D.n_out, k0, lane, sx, sh);
}

// USB demodulation + optional audio low-pass + int16 (vfo.cpp:300-332):
//   usb[m]  = I[m-62] - sum_{i<125} hp[i] Q[m-124+i]      DelayThing (dsp.h:101-106) and
//                                                         FIRHilbert (dsp.cpp:218-231, newest
//                                                         sample included; float sum, the
//                                                         subtraction in double)
//   usb'[m] = sum_{i<N} hu[i] usb[m-N+i]                  FIR::FIRUpdateAndProcess, newest excluded
//   out[m]  = short(usb' * gain * 32768.0)                float product, then double
// Only the 62 odd-index Hilbert taps are non-zero (the even ones are exactly 0.0f and adding
// 0*x leaves a float sum unchanged), so the sum runs over those, in index order:
//   sum = sum_{s<62} hnz[s] Q[m - 123 + 2 s].
//
// Blocking: one 256-thread block = 1024 outputs of one VFO.  Outputs of equal parity share their
// Q samples, so Q is staged in LDS as two parity planes and a thread computes 4 same-parity
// outputs from one contiguous run of 65 plane entries (17 ds_read_b128 for 248 MACs); the taps
// are wave-uniform scalar loads held in SGPRs.  The low-pass then takes 4 consecutive outputs per
// thread from the usb values parked in LDS, taps broadcast from LDS.  EXACT keeps one accumulator
// per output and the reference's summation order (4 independent chains per thread give the ILP).
constexpr int kDemodTile = 1024;
constexpr int kPlaneLen = (kDemodTile + kMaxFir + kHilbert + 1) / 2 + 8;

// ---- the non-exact arithmetics' MACs, two outputs per instruction ----------------------------------------------------------
// Four outputs r = 0..3 of a thread share a window: at window entry k output r takes tap h[k - r] (the Hilbert sum: outputs
// t0 + 2 r on one parity plane; the audio low-pass: four consecutive outputs).  As scalar FMAs that is four instructions per
// entry, and a v_fma_f32 pairs with another wave's instruction far less often than a two-operand v_mul / v_add does (37 % of
// k_usb_demod's instructions in the tolerance arithmetic against 80 % in the exact one: profiles/pmc_config3_tolerance.json).  As
// PACKED FMAs it is two: A = (out0, out1) += (h[k], h[k-1]) * v[k], B = (out2, out3) += (h[k-2], h[k-3]) * v[k].  The pair
// (h[k-1], h[k]) is an aligned register pair of one of two copies of the taps -- E: pairs (h[2j-1], h[2j]), O: pairs
// (h[2j], h[2j+1]) -- taken SWAPPED (op_sel / op_sel_hi on the tap operand), v[k] either half of an aligned pair for both
// lanes.  Every lane of every instruction is the same fma(h, v, acc) in the same order as the scalar form: bit-identical.
template <int H>
__device__ __forceinline__ void pk_fma2_s(v2f &A, v2f &B, v2f pa, v2f pb, v2f vv) // taps: scalar register pairs
{
    asm("v_pk_fma_f32 %0, %2, %4, %0 op_sel:[1,%5,0] op_sel_hi:[0,%5,1]\n\t"
        "v_pk_fma_f32 %1, %3, %4, %1 op_sel:[1,%5,0] op_sel_hi:[0,%5,1]"
        : "+v"(A), "+v"(B)
        : "s"(pa), "s"(pb), "v"(vv), "n"(H));
}
// the same in the exact arithmetic: product and sum rounded separately (v_pk_mul_f32, v_pk_add_f32), the two products first
template <int H>
__device__ __forceinline__ void pk_mac2_s(v2f &A, v2f &B, v2f pa, v2f pb, v2f vv)
{
    v2f ta, tb;
    asm("v_pk_mul_f32 %2, %4, %6 op_sel:[1,%7] op_sel_hi:[0,%7]\n\t"
        "v_pk_mul_f32 %3, %5, %6 op_sel:[1,%7] op_sel_hi:[0,%7]\n\t"
        "v_pk_add_f32 %0, %0, %2\n\t"
        "v_pk_add_f32 %1, %1, %3"
        : "+v"(A), "+v"(B), "=&v"(ta), "=&v"(tb)
        : "s"(pa), "s"(pb), "v"(vv), "n"(H));
}
template <int H>
__device__ __forceinline__ void pk_fma2_v(v2f &A, v2f &B, v2f pa, v2f pb, v2f vv) // taps: vector register pairs
{
    asm("v_pk_fma_f32 %0, %2, %4, %0 op_sel:[1,%5,0] op_sel_hi:[0,%5,1]\n\t"
        "v_pk_fma_f32 %1, %3, %4, %1 op_sel:[1,%5,0] op_sel_hi:[0,%5,1]"
        : "+v"(A), "+v"(B)
        : "v"(pa), "v"(pb), "v"(vv), "n"(H));
}
// pair i (0..8) of the 18 floats (x16, x2) a pass has loaded
template <int I>
__device__ __forceinline__ v2f pair_of(v16f a, v2f b)
{
    if constexpr (I < 8)
        return __builtin_shufflevector(a, a, 2 * I, 2 * I + 1);
    else
        return b;
}
// The Hilbert sum of four same-parity outputs: plane[k], k = 0 .. 64 (17 aligned b128 reads), taps hE[m] = h[m - 3], hO[m] =
// h[m - 2] (zeros outside 0 .. 61; K2Vfo::hnz_e / hnz_o, 96 floats each): pair j' of hE = (h[2j'-3], h[2j'-2]), of hO =
// (h[2j'-2], h[2j'-1]).  Entry k takes P(k) = (h[k-1], h[k]) for A and P(k-2) for B: P(k) = pair k/2 + 1 of hE (k even) or
// of hO (k odd).  Sixteen entries per pass, the 2 x 18 floats of taps a pass needs requested in front of it.
template <bool EXACT>
__device__ __forceinline__ void hilbert4_packed(const float *plane, const float *hE, const float *hO, float (&acc)[4])
{
    v2f A = {0.f, 0.f}, B = {0.f, 0.f};
    static_for<5>([&](auto pass_ic) {
        constexpr int pass = decltype(pass_ic)::value, nk = pass < 4 ? 16 : 1; // (entry 64 feeds out3 alone: h[61])
        const float *pe = hE + 16 * pass, *po = hO + 16 * pass;
        asm volatile("" : "+s"(pe), "+s"(po)); // (a pass's taps are requested here, not five passes' worth up front)
        const v16f e16 = *(const SDRX_AS4 v16f *)pe, o16 = *(const SDRX_AS4 v16f *)po;
        const v2f e2 = *(const SDRX_AS4 v2f *)(pe + 16), o2 = *(const SDRX_AS4 v2f *)(po + 16);
        static_for<(nk + 3) / 4>([&](auto g_ic) {
            constexpr int g = decltype(g_ic)::value;
            const v4f v4 = *reinterpret_cast<const v4f *>(plane + 16 * pass + 4 * g);
            static_for<(nk < 4 ? nk : 4)>([&](auto e_ic) {
                constexpr int e = decltype(e_ic)::value, kk = 4 * g + e, iA = kk / 2 + 1, iB = iA - 1;
                const v2f vv = e < 2 ? lo2(v4) : hi2(v4);
                if constexpr (EXACT) {
                    if constexpr (kk % 2 == 0)
                        pk_mac2_s<0>(A, B, pair_of<iA>(e16, e2), pair_of<iB>(e16, e2), vv);
                    else
                        pk_mac2_s<1>(A, B, pair_of<iA>(o16, o2), pair_of<iB>(o16, o2), vv);
                } else {
                    if constexpr (kk % 2 == 0)
                        pk_fma2_s<0>(A, B, pair_of<iA>(e16, e2), pair_of<iB>(e16, e2), vv);
                    else
                        pk_fma2_s<1>(A, B, pair_of<iA>(o16, o2), pair_of<iB>(o16, o2), vv);
                }
            });
        });
        __builtin_amdgcn_sched_barrier(0);
    });
    acc[0] = A.x, acc[1] = A.y, acc[2] = B.x, acc[3] = B.y;
}

#ifndef SDRX_PACKED_EXACT_HILBERT
#define SDRX_PACKED_EXACT_HILBERT 0 // (A/B: 1 = the exact arithmetic's Hilbert sum as packed v_pk_mul / v_pk_add on output pairs as well)
#endif
#ifndef SDRX_PACKED_LPF
#define SDRX_PACKED_LPF 0 // (A/B: 1 = the audio low-pass of the non-exact arithmetics as packed FMAs too -- measured slower on config 4, profiles/README.md round 6)
#endif
struct DemodLds { // LDS of one 256-thread demodulation block
    alignas(16) float sP0[kPlaneLen + 4]; // even offsets from `lo`, stored shifted by +3
    alignas(16) float sP1[kPlaneLen + 4]; // odd offsets
    alignas(16) float sI[kDemodTile + kMaxFir + 8];
    alignas(16) float sU[kDemodTile + kMaxFir + 16];
    alignas(16) float sH[kMaxFir + 16];  // sH[m] = hu[m - 3]: the E pairs of the packed form, (hu[2j'-3], hu[2j'-2])
    alignas(16) float sH1[SDRX_PACKED_LPF ? kMaxFir + 16 : 4]; // sH1[m] = sH[m + 1] = hu[m - 2]: the O pairs (the packed low-pass only)
};
static_assert(sizeof(DemodLds) % 16 == 0, "DemodLds is a whole number of 16-byte units");

template <bool EXACT>
__device__ __forceinline__ void demod_block(const K2Vfo *__restrict__ vfos, const BlockWork bw, unsigned long long frame_no, DemodLds &S,
                                            int tid)
{
    float *sP0 = S.sP0, *sP1 = S.sP1, *sI = S.sI, *sU = S.sU, *sH = S.sH, *sH1 = S.sH1;
    const K2Vfo *Dp = vfos + bw.vfo;
    const int blk = bw.blk;
    const int par = (int)(frame_no & 1ull);
    struct {
        const float *hnz, *lpf;
        short *pay;
        float *prequant;
        float gain;
        int H, n, nlpf, tile;
    } D = {ldc(&Dp->hnz), ldc(&Dp->lpf_pad), ldc(&Dp->pay[par]), ldc(&Dp->prequant), ldc(&Dp->gain),
           ldc(&Dp->H),   ldc(&Dp->n),       ldc(&Dp->nlpf),     ldc(&Dp->tile)};
    float *usb_out = ldc(&Dp->usb_out[par]); // non-null: a long low-pass follows in k_lpf_long
    const float2 *sbase = ldc(&Dp->s[par]);
    float2 *snext = ldc(&Dp->s_next[par]);
    const int m0 = blk * D.tile;
    if (m0 >= D.n && blk != 0)
        return;
    // The 62 Hilbert taps, read before this kernel has stored anything so the compiler can use
    // wave-uniform scalar loads and keep them in SGPRs for the whole block.  (The non-exact arithmetics take them pass by
    // pass as register PAIRS instead: hilbert4_packed.)
    constexpr bool kPackedHilbert = !EXACT || SDRX_PACKED_EXACT_HILBERT;
    float hnz[kHilbertNz];
    if constexpr (!kPackedHilbert) {
#pragma unroll
        for (int s = 0; s < kHilbertNz; ++s)
            hnz[s] = ldc(D.hnz + s);
    }
    if (D.nlpf > 0)
        for (int j = tid; j < D.nlpf + 15; j += 256) {
            sH[j] = gld(D.lpf + j);
            if constexpr (!EXACT && SDRX_PACKED_LPF)
                sH1[j] = j + 1 < D.nlpf + 15 ? gld(D.lpf + j + 1) : 0.f;
        }
    if (blk == 0) // history for the next frame: the last H entries of [hist | data]
        for (int j = tid; j < D.H; j += 256)
            gst2(snext + j, gld2(sbase + D.n + j));
    const float2 *z = sbase + D.H; // sample 0 of this frame
    const int N = D.nlpf;
    const int E = N + (N & 1);                  // usb values m0-E .. m0+1023 are computed (t = 0 .. ntv-1)
    const int ntv = D.tile + E;                 // <= 1024 when the host shortens the tile of a low-pass VFO by E: one usb pass
    const int lo = m0 - E - (kHilbert - 1);     // first stream index touched (>= -H)
    const int nr = ntv + (kHilbert - 1);        // offsets r = idx - lo in [0, nr)
    // all of this thread's loads are issued before the first LDS write (a rolled load -> wait ->
    // write loop would pay the memory latency once per iteration)
    constexpr int kStageIters = (kDemodTile + kMaxFir + kHilbert + 255) / 256;
    float2 stage[kStageIters];
#pragma unroll
    for (int it = 0; it < kStageIters; ++it) {
        const int r = tid + 256 * it, idx = lo + r;
        stage[it] = (r < nr && idx < D.n) ? gld2_once(z + idx) : make_float2(0.f, 0.f);
    }
#pragma unroll
    for (int it = 0; it < kStageIters; ++it) {
        const int r = tid + 256 * it;
        if (r < nr) {
            if (r & 1)
                sP1[r >> 1] = stage[it].y;
            else
                sP0[(r >> 1) + 3] = stage[it].y;
            const int t = r - kDelay;           // I[u-62] for u = m0 - E + t
            if (t >= 0 && t < ntv)
                sI[t] = stage[it].x;
        }
    }
    __syncthreads();

    // ---- usb: Q index of tap s for value t is r = t + 1 + 2 s
    const int p = tid >> 7, q = tid & 127;      // waves 0-1 take even t, waves 2-3 odd t
    const int soff = (4 - (E - N)) & 3;         // sU shift that makes the low-pass reads 16-byte aligned
    for (int tb = 0; tb < ntv; tb += kDemodTile) {
        const int t0 = tb + p + 8 * q;          // this thread: t0, t0+2, t0+4, t0+6
        if (t0 >= ntv)
            continue;
        const float *plane = p == 0 ? sP1 + (t0 >> 1) : sP0 + 3 + ((t0 + 1) >> 1);
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        if constexpr (!kPackedHilbert) {
#pragma unroll
            for (int g = 0; g < 17; ++g) {
                const float4 v4 = *reinterpret_cast<const float4 *>(plane + 4 * g);
                const float v[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        const int s = 4 * g + e - rr;
                        if (s >= 0 && s < kHilbertNz)
                            acc[rr] = acc[rr] + hnz[s] * v[e];
                    }
            }
        } else {
            hilbert4_packed<EXACT>(plane, ldc(&Dp->hnz_e), ldc(&Dp->hnz_o), acc);
        }
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int t = t0 + 2 * rr;
            if (t < ntv)
                sU[t + soff] = usb_difference(sI[t], acc[rr]);
        }
    }
    if (tid < 8) // read by the low-pass only under its zero padding taps: must be finite (0 * NaN = NaN)
        sU[ntv + soff + tid] = 0.f; // <= 1280 + 3 + 7, inside sU
    __syncthreads();

    // ---- audio low-pass (newest sample excluded) on 4 consecutive outputs, then int16
    const int j0 = 4 * tid;
    const int m = m0 + j0;
    if (j0 >= D.tile || m >= D.n)
        return;
    float u4[4];
    if (N > 0) {
        // output j0+rr, tap i reads usb t = j0 + rr + (E-N) + i; sH[i+3] = hu[i], zeros around
        const float *w = sU + soff + (E - N) + j0; // 16-byte aligned by the choice of soff
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        const int groups = (N + 6) / 4;
        if constexpr (EXACT || !SDRX_PACKED_LPF) {
            for (int g = 0; g < groups; ++g) {
                const float4 v4 = *reinterpret_cast<const float4 *>(w + 4 * g);
                const float v[4] = {v4.x, v4.y, v4.z, v4.w};
                const float4 ha = *reinterpret_cast<const float4 *>(sH + 4 * g), hb4 = *reinterpret_cast<const float4 *>(sH + 4 * g + 4);
                const float h[8] = {ha.x, ha.y, ha.z, ha.w, hb4.x, hb4.y, hb4.z, hb4.w}; // h[k] = hu[4g + k - 3]
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        if (EXACT)
                            acc[rr] = acc[rr] + h[e - rr + 3] * v[e];
                        else
                            acc[rr] = fmaf(h[e - rr + 3], v[e], acc[rr]);
                    }
            }
        } else {
            // the packed form (see hilbert4_packed): window entry q = 4 g + e gives out_r the tap hu[q - r]; quad t of sH holds
            // the E pairs 2 t, 2 t + 1 -- (hu[4t-3], hu[4t-2]), (hu[4t-1], hu[4t]) --, of sH1 the O pairs (hu[4t-2], hu[4t-1]),
            // (hu[4t], hu[4t+1]); a group needs the pairs 2 g, 2 g + 1 (the quad before) and 2 g + 2 (this one's low half)
            v2f A = {0.f, 0.f}, B = {0.f, 0.f};
            v4f ep = *reinterpret_cast<const v4f *>(sH), op = *reinterpret_cast<const v4f *>(sH1);
            for (int g = 0; g < groups; ++g) {
                const v4f w4 = *reinterpret_cast<const v4f *>(w + 4 * g);
                const v4f ec = *reinterpret_cast<const v4f *>(sH + 4 * g + 4), oc = *reinterpret_cast<const v4f *>(sH1 + 4 * g + 4);
                pk_fma2_v<0>(A, B, hi2(ep), lo2(ep), lo2(w4)); // q = 4g:     (hu[4g-1], hu[4g])   | (hu[4g-3], hu[4g-2])
                pk_fma2_v<1>(A, B, hi2(op), lo2(op), lo2(w4)); // q = 4g + 1: (hu[4g],   hu[4g+1]) | (hu[4g-2], hu[4g-1])
                pk_fma2_v<0>(A, B, lo2(ec), hi2(ep), hi2(w4)); // q = 4g + 2: (hu[4g+1], hu[4g+2]) | (hu[4g-1], hu[4g])
                pk_fma2_v<1>(A, B, lo2(oc), hi2(op), hi2(w4)); // q = 4g + 3: (hu[4g+2], hu[4g+3]) | (hu[4g],   hu[4g+1])
                ep = ec, op = oc;
            }
            acc[0] = A.x, acc[1] = A.y, acc[2] = B.x, acc[3] = B.y;
        }
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
            u4[rr] = acc[rr];
    } else {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
            u4[rr] = sU[soff + j0 + rr];
    }
    if (usb_out) { // hand the unfiltered usb values on (k_lpf_long quantises)
        for (int rr = 0; rr < 4 && m + rr < D.n; ++rr)
            *(SDRX_AS1 float *)(usb_out + m + rr) = u4[rr];
        return;
    }
    short o4[4];
    float pq[4];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        pq[rr] = quantise(u4[rr] * D.gain, o4[rr]);
    }
    if (m + 3 < D.n) {
        v4s o = {o4[0], o4[1], o4[2], o4[3]};
        *(SDRX_AS1 v4s *)(D.pay + m) = o;
        if (D.prequant)
            gst4(reinterpret_cast<float4 *>(D.prequant + m), make_float4(pq[0], pq[1], pq[2], pq[3]));
    } else {
        for (int rr = 0; rr < 4 && m + rr < D.n; ++rr) {
            *(SDRX_AS1 short *)(D.pay + m + rr) = o4[rr];
            if (D.prequant)
                *(SDRX_AS1 float *)(D.prequant + m + rr) = pq[rr];
        }
    }
}

#ifndef SDRX_DEMOD_MIN_WAVES
#define SDRX_DEMOD_MIN_WAVES 1 // (A/B: 8 = at most 64 VGPRs, eight 256-thread blocks per CU)
#endif
template <bool EXACT>
__global__ __launch_bounds__(256, SDRX_DEMOD_MIN_WAVES) void k_usb_demod(const K2Vfo *__restrict__ vfos, const BlockWork *__restrict__ work,
                                                   unsigned long long frame_no)
{
    __shared__ __attribute__((aligned(16))) DemodLds S;
    demod_block<EXACT>(vfos, work[blockIdx.x], frame_no, S, (int)threadIdx.x);
}

// ------------------------------------------------------------------------------------ k_mix_levels
// ONE launch for the mix/decimate items of EVERY tree level, each level working on the frame that has
// reached it: level l on frame k - l (a software pipeline over consecutive frames: the levels of one
// launch are independent of each other).  Why: on BASELINE config 3 the level-0 launch -- 2-3 main VFOs,
// ~770 one-chunk items -- leaves most of the chip idle for its whole duration (6-9 us of a ~110 us
// frame); as part of the same grid its short items run beside the sub VFOs' long ones, without the
// cross-stream events that cost more than they gain on this runtime (profiles/README.md).
// The list is [level 0 items | level 1 items | ...], every part starting at a multiple of 8 entries
// (workgroup i runs on XCD i mod 8: a part keeps its XCD mapping whichever range a launch covers);
// -1 = padding.  One wave per workgroup, exactly as k_mix_decimate.
struct LevelArgs {
    unsigned long long frame_level[kMaxLevels]; // frame each tree level works on in this launch
    const void *raw;                            // level 0's raw frame ...
    int raw_mode;                               // ... and its form (kRaw*)
    int pad_;
};
template <bool EXACT, bool ROT = !EXACT>
__global__ __launch_bounds__(64, SDRX_K1_MIN_WAVES) void k_mix_levels(const K1Vfo *__restrict__ k1, const K1Work *__restrict__ items,
                                                   const int *__restrict__ item_level, const int *__restrict__ list, LevelArgs A)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int it = ldc(list + blockIdx.x);
    if (it < 0)
        return;
    const int lv = ldc(item_level + it);
    run_item<EXACT, ROT>(k1, K1Work{ldc(&items[it].vfo), ldc(&items[it].s_begin), ldc(&items[it].s_first_out), ldc(&items[it].s_end)},
                    A.frame_level[lv], A.raw, A.raw_mode, lv == 0, smem, (int)threadIdx.x);
}

// An audio low-pass of more than kMaxFir taps (the reference accepts any filter_bandwidth: 500 Hz at
// 48 kS/s is 925 taps, firfilter.cpp:108-119): usb'[m] = sum_{i<N} hu[i] usb[m-N+i], newest sample
// excluded (FIR::FIRUpdateAndProcess, dsp.cpp:59-71), one accumulator per output in tap order, then the
// same quantisation as k_usb_demod.  256 outputs per block; the block's window of N + 256 usb values sits
// in LDS, the taps are wave-uniform scalar loads.  A rare configuration: correctness first.
template <bool EXACT>
__global__ __launch_bounds__(256) void k_lpf_long(const K4Vfo *__restrict__ vfos, const BlockWork *__restrict__ work, unsigned long long frame_no)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *sw = reinterpret_cast<float *>(smem);
    const K4Vfo *Dp = vfos + ldc(&work[blockIdx.x].vfo);
    const int blk = ldc(&work[blockIdx.x].blk);
    const int par = (int)(frame_no & 1ull);
    const int tid = threadIdx.x;
    const float *taps = ldc(&Dp->taps);
    const int Hu = ldc(&Dp->Hu), n = ldc(&Dp->n), N = ldc(&Dp->nlpf);
    const float gain = ldc(&Dp->gain);
    const float *ubase = ldc(&Dp->u[par]);
    float *unext = ldc(&Dp->u_next[par]);
    short *pay = ldc(&Dp->pay[par]);
    float *prequant = ldc(&Dp->prequant);
    if (blk == 0) // history for the next frame: the last Hu entries of [hist | data]
        for (int j = tid; j < Hu; j += 256)
            *(SDRX_AS1 float *)(unext + j) = gld(ubase + n + j);
    const float *u = ubase + Hu; // sample 0 of this frame
    const int m0 = blk * 256;
    for (int t = tid; t < N + 256; t += 256) { // window: usb[m0 - N .. m0 + 255]
        const int idx = m0 - N + t;
        sw[t] = idx < n ? gld(u + idx) : 0.f;
    }
    __syncthreads();
    const int m = m0 + tid;
    if (m >= n)
        return;
    const float *w = sw + tid; // w[i] = usb[m - N + i]
    float acc = 0.f;
    for (int i = 0; i < N; ++i) {
        const float h = ldc(taps + i);
        if (EXACT)
            acc = acc + h * w[i];
        else
            acc = fmaf(h, w[i], acc);
    }
    short q;
    const float pre = quantise(acc * gain, q);
    *(SDRX_AS1 short *)(pay + m) = q;
    if (prequant)
        *(SDRX_AS1 float *)(prequant + m) = pre;
}

// vfo::compress (vfo.cpp:389-424): cstyle 1 packs the high nibbles of (re/scalecomp)*128 and
// (im/scalecomp)*128 into one byte; otherwise two int8 per sample.
__device__ __forceinline__ int to_schar(float f) // cvttss2si + the low 8 bits: out of int32 range or NaN -> INT32_MIN -> 0
{
    const int t = (f >= -2147483648.0f && f < 2147483648.0f) ? (int)f : (int)0x80000000;
    return (int)(signed char)(unsigned char)(unsigned)t;
}
__global__ __launch_bounds__(256) void k_compress(const K3Vfo *__restrict__ vfos, const BlockWork *__restrict__ work,
                                                  unsigned long long frame_no)
{
    const BlockWork bw = work[blockIdx.x];
    const K3Vfo *Dp = vfos + bw.vfo;
    const int blk = bw.blk;
    const int par = (int)(frame_no & 1ull);
    struct {
        signed char *pay;
        int n, cstyle, scalecomp;
    } D = {Dp->pay[par], Dp->n, Dp->cstyle, Dp->scalecomp};
    const float2 *z = Dp->s[par];
    for (int i = blk * 4096 + threadIdx.x; i < min(D.n, (blk + 1) * 4096); i += 256) {
        const float2 v = gld2(z + i);
        if (D.cstyle == 1) {
            const float sc = (float)D.scalecomp;
            const int re = to_schar((v.x / sc) * 128.0f);
            const int im = to_schar((v.y / sc) * 128.0f);
            D.pay[i] = (signed char)((re & 0xF0) | ((im & 0xF0) >> 4));
        } else {
            D.pay[2 * i] = (signed char)to_schar(v.x * 128.0f);
            D.pay[2 * i + 1] = (signed char)to_schar(v.y * 128.0f);
        }
    }
}

} // namespace sdrx
