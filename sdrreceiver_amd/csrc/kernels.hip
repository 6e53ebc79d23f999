// kernels.hip -- hand-written HIP kernels of the per-VFO IQ chain for gfx950 (MI355X, CDNA4).
//
// Compiled with -ffp-contract=off: every a*b+c below is two roundings unless fmaf() is
// spelled out.  That is what makes the EXACT variants reproduce the reference's -O2 x86-64
// arithmetic bit for bit; the FAST variants use explicit fmaf().
//
// Kernel map (reference function -> kernel):
//   Oscillator::Oscillator            oscillator.cpp:4-32      -> k_nco_init
//   vfo::process mix loop             vfo.cpp:237-245          \
//   HalfBandDecimator::decimate       halfbanddecimator.cpp:43-72 } k_mix_decimate
//     FIR::...HalfBandQueue           dsp.cpp:96-173           /
//   vfo::usb_decimdemod (FIR part)    vfo.cpp:334-387          -> k_late_decimate
//   vfo::usb_demod / demod tail       vfo.cpp:300-332,350-364  -> k_usb_demod
//   vfo::compress                     vfo.cpp:389-424          -> k_compress
//
// Execution model: 64-wide wavefronts.  k_mix_decimate runs ONE wave per workgroup and one
// workgroup per (VFO, time segment); a wave owns all the LDS it touches, so its
// __syncthreads() are wave-local (no cross-wave barrier traffic) and the segment walks the
// frame chunk by chunk with the filter state resident in LDS.
#include "sdrx_dev.h"

namespace sdrx {

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

// LDS layout of k_mix_decimate (bytes).  `raw`: the input chunk as 512 float4 units (2 samples
// each), one pad unit after every 8 so that the per-lane 128-byte runs are bank-conflict free
// for ds_read_b128.  Stage arrays A_s: [16 carry | (1024 >> s) data] float2 in a linear index
// space p; A_0 alone is stored through pad0() because it is WRITTEN as 16-sample lane runs.
constexpr int kRawUnits = 512 + 512 / 8;                 // 576 float4
constexpr int kRawBytes = kRawUnits * 16;                // 9216
__host__ __device__ constexpr int pad0(int p) { return p + 2 * (p >> 4); }
constexpr int kA0Elems = kCarry + kChunk + 2 * ((kCarry + kChunk) >> 4) + 2; // 1172
__host__ __device__ constexpr int stage_elems(int s) { return s == 0 ? kA0Elems : kCarry + (kChunk >> s); }
__host__ __device__ constexpr int stage_offset(int s)
{
    int o = 0;
    for (int t = 0; t < s; ++t)
        o += stage_elems(t);
    return o;
}
__host__ __device__ constexpr int k1_lds_bytes(int d)
{
    return kRawBytes + 8 * stage_offset(d < 1 ? 1 : d);
}

template <int S>
__device__ __forceinline__ int amap(int p)
{
    return S == 0 ? pad0(p) : p;
}

// One half-band stage of one chunk, fully in LDS: A (stage S input, linear space with carry)
// -> B (stage S+1 input) or global `out` when S is the last stage.
template <bool EXACT, int S>
__device__ __forceinline__ void hb_stage(float2 *__restrict__ A, float2 *__restrict__ B, float2 *__restrict__ gout,
                                         bool last, bool emit, int cnt, int lane, bool save, float2 *__restrict__ hbsave)
{
    __syncthreads(); // stage input (written by the previous phase) is visible
    const int nout = cnt >> 1;
    for (int j = lane; j < nout; j += 64) {
        const int p = kCarry + 2 * j - 10; // window p .. p+10, newest = input sample 2j of this chunk
        float2 w0 = A[amap<S>(p)], w2 = A[amap<S>(p + 2)], w4 = A[amap<S>(p + 4)], w5 = A[amap<S>(p + 5)],
               w6 = A[amap<S>(p + 6)], w8 = A[amap<S>(p + 8)], w10 = A[amap<S>(p + 10)];
        float2 y;
        y.x = hb_dot<EXACT>(w0.x, w2.x, w4.x, w5.x, w6.x, w8.x, w10.x);
        y.y = hb_dot<EXACT>(w0.y, w2.y, w4.y, w5.y, w6.y, w8.y, w10.y);
        if (!last)
            B[amap<S + 1>(kCarry + j)] = y;
        else if (emit)
            gout[j] = y;
    }
    __syncthreads(); // all window reads done before the carry is overwritten
    // FIRQueueBackToFront (dsp.cpp:163-173) at the end of the FRAME: the 10 samples before the
    // LAST one become the next frame's history: x[-k] := x[size-1-k].
    if (save && lane < kHbHist)
        hbsave[lane] = A[amap<S>(kCarry + cnt - 2 - lane)];
    // Between chunks of one frame the stream is simply continuous: keep the last 16 samples.
    float2 t;
    if (lane < kCarry)
        t = A[amap<S>(cnt + lane)];
    __syncthreads();
    if (lane < kCarry)
        A[amap<S>(lane)] = t;
}

template <bool EXACT, int S>
struct StageChain {
    static __device__ __forceinline__ void run(float2 *lds, int d, float2 *gout, bool emit, int valid, int lane,
                                               bool save, float2 *hbsave)
    {
        if (S < d) {
            hb_stage<EXACT, S>(lds + stage_offset(S), lds + stage_offset(S + 1), gout, S + 1 == d, emit, valid >> S,
                               lane, save, hbsave + S * kHbHist);
            StageChain<EXACT, S + 1>::run(lds, d, gout, emit, valid, lane, save, hbsave);
        }
    }
};
template <bool EXACT>
struct StageChain<EXACT, kMaxStages> {
    static __device__ __forceinline__ void run(float2 *, int, float2 *, bool, int, int, bool, float2 *) {}
};

// Fused NCO + mixer + half-band cascade.  One wave per workgroup, one workgroup per K1Work.
template <bool EXACT>
__global__ __launch_bounds__(64) void k_mix_decimate(const K1Vfo *__restrict__ vfos, const K1Work *__restrict__ work,
                                                     const float2 *__restrict__ raw_frame,
                                                     unsigned long long frame_no)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float4 *raw = reinterpret_cast<float4 *>(smem);
    float2 *lds = reinterpret_cast<float2 *>(smem + kRawBytes);

    const K1Work W = work[blockIdx.x];
    const int lane = threadIdx.x;
    const int par = (int)(frame_no & 1ull);
    // Scalar (wave-uniform) loads of the descriptor; the parity-indexed pointers are read
    // straight from memory so the struct never becomes a runtime-indexed private array.
    const K1Vfo *Dp = vfos + W.vfo;
    struct {
        const float2 *cp;
        float rot_re, rot_im;
        int n_in, d, L;
    } D = {Dp->cp, Dp->rot_re, Dp->rot_im, Dp->n_in, Dp->d, Dp->L};
    const float2 *in = Dp->in[par] ? Dp->in[par] : raw_frame;
    float2 *out = Dp->out[par];
    float2 *hb_load = Dp->hb[par];
    float2 *hb_save = Dp->hb[par ^ 1];
    const int nchunks = (D.n_in + kChunk - 1) / kChunk;

    // Filter state at the start of this segment.  Segment 0 continues from the previous frame's
    // saved history (zero at start-up, dsp.cpp:40-49); a later segment starts from zeros and
    // runs warm-up chunks until every stage's window holds real samples again.
    const int nst = D.d < 1 ? 1 : D.d;
    for (int s = 0; s < nst; ++s) {
        if (lane < kCarry) {
            float2 v = make_float2(0.f, 0.f);
            const int k = kCarry - lane; // position `lane` of the carry is x[-k]
            if (W.c_begin == 0 && k <= kHbHist && s < D.d)
                v = hb_load[s * kHbHist + (k - 1)];
            const int p = lane;
            lds[stage_offset(s) + (s == 0 ? pad0(p) : p)] = v;
        }
    }
    const int phase_frame = (int)((frame_no * (unsigned long long)D.n_in) % (unsigned long long)D.L);

    for (int c = W.c_begin; c < W.c_end; ++c) {
        const int base = c * kChunk;
        const int valid = min(kChunk, D.n_in - base);
        const bool emit = c >= W.c_first_out;
        const bool last_chunk = c == nchunks - 1;

        // 1. input chunk, coalesced 16 B per lane, into the padded raw tile
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int u = i * 64 + lane;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (2 * u < valid)
                v = *reinterpret_cast<const float4 *>(in + base + 2 * u);
            raw[u + (u >> 3)] = v;
        }
        __syncthreads();

        // 2. this lane's run of 16 consecutive samples
        float2 x[kRun];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float4 v = raw[9 * lane + i];
            x[2 * i] = make_float2(v.x, v.y);
            x[2 * i + 1] = make_float2(v.z, v.w);
        }

        // 3. NCO: regenerate table[idx .. idx+16) from the checkpoint before it, and mix
        //    (vfo.cpp:241: osc * sample, re = ac - bd, im = ad + bc).  The very first sample
        //    after start-up is multiplied by the LAST table entry (oscillator.cpp:30,39-50).
        int idx = phase_frame + base; // both < L
        idx -= idx >= D.L ? D.L : 0;
        idx += lane * kRun;           // L >= kChunk (checked by sdrx_finalize)
        idx -= idx >= D.L ? D.L : 0;
        float2 o = D.cp[idx >> 4];
        const bool first_ever = frame_no == 0 && base == 0 && lane == 0;
#pragma unroll
        for (int i = 0; i < kRun; ++i) {
            o = nco_step(o, D.rot_re, D.rot_im);
            float2 m = o;
            if (i == 0 && first_ever)
                m = D.cp[D.L >> 4];
            const float a = m.x, b = m.y, cc = x[i].x, dd = x[i].y;
            if (EXACT) {
                x[i].x = a * cc - b * dd;
                x[i].y = a * dd + b * cc;
            } else {
                x[i].x = fmaf(a, cc, -(b * dd));
                x[i].y = fmaf(a, dd, b * cc);
            }
        }

        // 4. mixed samples into A_0 (lane runs, 144-byte lane stride: conflict-free b128 writes)
        float2 *A0 = lds;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int p = kCarry + lane * kRun + 2 * i;
            *reinterpret_cast<float4 *>(A0 + pad0(p)) = make_float4(x[2 * i].x, x[2 * i].y, x[2 * i + 1].x, x[2 * i + 1].y);
        }

        if (D.d == 0) {
            // no decimation: decimate[0] is the mixed stream itself
            __syncthreads();
            if (emit) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int u = i * 64 + lane;
                    if (2 * u < valid)
                        *reinterpret_cast<float4 *>(out + base + 2 * u) =
                            *reinterpret_cast<const float4 *>(A0 + pad0(kCarry + 2 * u));
                }
            }
        } else {
            // 5. the cascade
            StageChain<EXACT, 0>::run(lds, D.d, out + (base >> D.d), emit, valid, lane,
                                      last_chunk && W.c_end == nchunks, hb_save);
        }
    }
}

// ------------------------------------------------------------------------------------ demod tail
// `short = double` of the reference's x86-64 build (vfo.cpp:328,364): truncate toward zero to
// int32, keep the low 16 bits.  v_cvt_i32_f64 truncates and saturates; in-range values (all
// that the reference defines) agree.
__device__ __forceinline__ short to_short(double d)
{
    return (short)(unsigned short)(unsigned)(int)d;
}

// Late decimation by L in {5,6} (vfo.cpp:334-387 with FIR::FIRUpdateAndProcess/FIRUpdate,
// dsp.cpp:59-71,150-154): z'[k] = sum_i hd[i] * x[L k - Nd + i]  -- the newest sample x[L k] is
// NOT part of the sum ((N+1)-slot ring).  The phase counter restarts every frame and frames are
// multiples of L, so k is frame-local.  256 outputs per block; the input window sits in LDS.
template <bool EXACT>
__global__ __launch_bounds__(256) void k_late_decimate(const K2aVfo *__restrict__ vfos, int blocks_per_vfo,
                                                       unsigned long long frame_no)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float2 *sx = reinterpret_cast<float2 *>(smem);
    const K2aVfo *Dp = vfos + blockIdx.x / blocks_per_vfo;
    const int blk = blockIdx.x % blocks_per_vfo;
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
            xnext[j] = xbase[D.n + j];
    const int lo = D.L * k0 - D.ndec;              // first input index needed (>= -Hx)
    const int span = D.L * 255 + D.ndec;           // indices lo .. lo+span-1 feed the 256 outputs
    for (int t = tid; t < span; t += 256) {
        const int idx = lo + t;
        sx[t] = idx < D.n ? x[idx] : make_float2(0.f, 0.f);
    }
    __syncthreads();
    const int k = k0 + tid;
    if (k >= D.n_out)
        return;
    const float2 *w = sx + D.L * tid; // w[i] = x[L k - Nd + i]
    float ar = 0.f, ai = 0.f;
    if (EXACT) {
        for (int i = 0; i < D.ndec; ++i) {
            const float h = D.taps[i];
            ar = ar + h * w[i].x;
            ai = ai + h * w[i].y;
        }
    } else {
        for (int i = 0; i < D.ndec; ++i) {
            const float h = D.taps[i];
            ar = fmaf(h, w[i].x, ar);
            ai = fmaf(h, w[i].y, ai);
        }
    }
    zout[k] = make_float2(ar, ai);
}

// USB demodulation + optional audio low-pass + int16 (vfo.cpp:300-332):
//   usb[m]  = I[m-62] - sum_{i<125} hp[i] Q[m-124+i]      DelayThing (dsp.h:101-106) and
//                                                         FIRHilbert (dsp.cpp:218-231, newest
//                                                         sample included; float sum, the
//                                                         subtraction in double)
//   usb'[m] = sum_{i<N} hu[i] usb[m-N+i]                  FIR::FIRUpdateAndProcess, newest excluded
//   out[m]  = short(usb' * gain * 32768.0)                float product, then double
// Only the 62 odd-index Hilbert taps are non-zero (the even ones are exactly 0.0f and adding
// 0*x leaves a float sum unchanged), so the sum runs over those, in index order.
// 256 outputs per block; stream window (I and Q planes) and the usb window sit in LDS.
template <bool EXACT>
__global__ __launch_bounds__(256) void k_usb_demod(const K2Vfo *__restrict__ vfos, int blocks_per_vfo,
                                                   unsigned long long frame_no)
{
    __shared__ float sI[256 + kMaxFir + 128];
    __shared__ float sQ[256 + kMaxFir + 128];
    __shared__ float sU[256 + kMaxFir];
    const K2Vfo *Dp = vfos + blockIdx.x / blocks_per_vfo;
    const int blk = blockIdx.x % blocks_per_vfo;
    const int par = (int)(frame_no & 1ull);
    const int tid = threadIdx.x;
    struct {
        const float *hilbert, *lpf;
        short *pay;
        float *prequant;
        float gain;
        int H, n, nlpf;
    } D = {Dp->hilbert, Dp->lpf, Dp->pay, Dp->prequant, Dp->gain, Dp->H, Dp->n, Dp->nlpf};
    const float2 *sbase = Dp->s[par];
    float2 *snext = Dp->s_next[par];
    const int m0 = blk * 256;
    if (m0 >= D.n && blk != 0)
        return;
    if (blk == 0)
        for (int j = tid; j < D.H; j += 256)
            snext[j] = sbase[D.n + j];
    const float2 *z = sbase + D.H; // sample 0 of this frame
    const int N = D.nlpf;
    const int lo = m0 - N - (kHilbert - 1); // first stream index needed (>= -H)
    const int span = 256 + N + (kHilbert - 1);
    for (int t = tid; t < span; t += 256) {
        const int idx = lo + t;
        float2 v = idx < D.n ? z[idx] : make_float2(0.f, 0.f);
        sI[t] = v.x;
        sQ[t] = v.y;
    }
    __syncthreads();
    // usb for indices m0-N .. m0+255  (u = index - (m0-N))
    for (int u = tid; u < 256 + N; u += 256) {
        const float *q = sQ + u; // q[i] = Q[mu - 124 + i]
        float acc = 0.f;
        if (EXACT) {
#pragma unroll 4
            for (int i = 1; i < kHilbert; i += 2)
                acc = acc + D.hilbert[i] * q[i];
        } else {
            float a0 = 0.f, a1 = 0.f;
#pragma unroll 4
            for (int i = 1; i < kHilbert - 2; i += 4) {
                a0 = fmaf(D.hilbert[i], q[i], a0);
                a1 = fmaf(D.hilbert[i + 2], q[i + 2], a1);
            }
            acc = a0 + a1; // 31 pairs (i, i+2), i = 1,5,..,121: all 62 odd taps 1..123
        }
        const float usb = (float)((double)sI[u + kDelay] - (double)acc);
        sU[u] = usb;
    }
    __syncthreads();
    const int m = m0 + tid;
    if (m >= D.n)
        return;
    float usb;
    if (N > 0) {
        const float *w = sU + tid; // w[i] = usb[m - N + i]
        float acc = 0.f;
        if (EXACT) {
            for (int i = 0; i < N; ++i)
                acc = acc + D.lpf[i] * w[i];
        } else {
            for (int i = 0; i < N; ++i)
                acc = fmaf(D.lpf[i], w[i], acc);
        }
        usb = acc;
    } else {
        usb = sU[tid];
    }
    const float scaled = usb * D.gain;
    const double pre = (double)scaled * 32768.0;
    D.pay[m] = to_short(pre);
    if (D.prequant)
        D.prequant[m] = (float)pre; // exact: float * 2^15
}

// vfo::compress (vfo.cpp:389-424): cstyle 1 packs the high nibbles of (re/scalecomp)*128 and
// (im/scalecomp)*128 into one byte; otherwise two int8 per sample.
__device__ __forceinline__ int to_schar(float f)
{
    return (int)(signed char)(unsigned char)(unsigned)(int)f;
}
__global__ __launch_bounds__(256) void k_compress(const K3Vfo *__restrict__ vfos, int blocks_per_vfo,
                                                  unsigned long long frame_no)
{
    const K3Vfo *Dp = vfos + blockIdx.x / blocks_per_vfo;
    const int blk = blockIdx.x % blocks_per_vfo;
    const int par = (int)(frame_no & 1ull);
    struct {
        signed char *pay;
        int n, cstyle, scalecomp;
    } D = {Dp->pay, Dp->n, Dp->cstyle, Dp->scalecomp};
    const float2 *z = Dp->s[par];
    for (int i = blk * 256 + threadIdx.x; i < D.n; i += blocks_per_vfo * 256) {
        const float2 v = z[i];
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
