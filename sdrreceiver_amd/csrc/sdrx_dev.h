// sdrx_dev.h -- device-visible descriptors shared by the kernels and the host orchestration.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sdrx {

constexpr int kRun = 16;            // input samples per lane per chunk (NCO checkpoint spacing)
constexpr int kChunk = 64 * kRun;   // 1024 input samples per wave-iteration
constexpr int kCarry = 16;          // per-stage history window kept in LDS between chunks
constexpr int kHbHist = 10;         // half-band history carried between frames (x[-1..-10])
constexpr int kMaxStages = 8;       // vfo.h:63
constexpr int kMaxLevels = 4;       // tree depth up to which k_mix_levels (all levels in one launch) is used (the reference builds 2)
constexpr int kHilbert = 125;       // vfo.cpp:137
constexpr int kHilbertNz = 62;      // non-zero Hilbert taps (odd indices 1..123)
constexpr int kDelay = 62;          // vfo.cpp:136
constexpr int kMaxFir = 256;        // longest low-pass k_usb_demod applies itself (staged in LDS); longer ones: k_lpf_long
constexpr int kMaxFirLong = 8192;   // longest low-pass at all (window + block in LDS: 33 KB)
// where the parent-less VFOs of a launch read the raw frame from
constexpr int kRawTiled = 0;        // the context's tile-layout copy (an ingest kernel wrote it)
constexpr int kRawF32 = 1;          // the caller's cf32 frame, natural order
constexpr int kRawU8 = 2;           // the dongle's interleaved bytes, natural order

struct K2Vfo;
// ---- mix + half-band cascade (one per VFO) ---------------------------------------------------
struct K1Vfo {
    const float2 *in[2];   // input stream in TILE LAYOUT (see kernels.hip) per frame parity
    float2 *out[2];        // decimate[d] of this frame, per frame parity (tile layout iff out_tiled)
    const float2 *cp;      // NCO checkpoints: cp[j] = table[16 j - 1], cp[0] = (1,0), cp[L/16] = table[L-1]
    float2 *hb[2];         // half-band history per frame parity: [d][10], entry k-1 = x[-k]
    float rot_re, rot_im;  // NCO rotation (float cos, float sin of the double angle)
    int n_in;              // complex samples per frame
    int d;                 // half-band stages
    int L;                 // NCO table length = (int)fs
    int out_tiled;         // 1: children consume the output (tile layout); 0: natural order for the demod
    // A leaf whose /5 or /6 low-pass (vfo::usb_decimdemod, vfo.cpp:334-387) runs in the mix wave itself (late_item, kernels.hip):
    // d == 0, `out` = the DECIMATED stream z' (behind its demodulation history), hb[] = the last kLateHist mixed samples of
    // the previous frame, and the mixed 240 kS/s stream decimate[0] is never written to HBM -- unless it is wanted:
    int late_L;            // 0: not such a leaf; 5 | 6
    int pad_;
    const float *late_taps; // the Nd = LateGeom<L>::kTaps taps of the decimating low-pass
    float2 *tap[2];        // non-null: also keep decimate[0] of this frame here, natural order (sdrx_set_tap, option keep_streams)
    // Tolerance arithmetic only (option exact = 0; kernels.hip "NCO in the tolerance arithmetic"): rk[j] = u^(j+1), u = the
    // rotation by the angle of (rot_re, rot_im) at unit modulus, computed in double and stored as floats -- 1 .. 4 steps of
    // the table's recurrence as one rotation, once its amplitude has settled (entries >= kNcoSettle).
    float2 rk[4];
    // A USB leaf that demodulates in the mix wave itself (demod_chunk, kernels.hip; option fuse_demod): its demodulation
    // descriptor.  `out` is then unused -- decimate[d] of such a leaf goes to HBM only through `tap`.
    const K2Vfo *dm;
};
static_assert(sizeof(K1Vfo) % 8 == 0, "K1Vfo array stride");
constexpr int kNcoSettle = 512; // table entries below this still carry the start-up ringing of the amplitude stabiliser (oscillator.cpp:20-28): always replayed exactly

// Geometry of the fused late decimation for L in {5, 6}.  The decimating low-pass is low_pass(2, rate L, rate / 2, rate / (L - 1))
// (vfo.cpp:82-87): its length (int)(53 fs / (22 tw)) made odd depends on fs / tw = L (L - 1) only -- 49 taps for L = 5, 73 for
// L = 6, whatever the rate (checked at finalize; anything else takes the two-kernel path).  A wave walks its segment in chunks
// of kChunkLen samples (a multiple of 16 L, so that a chunk starts on an output and a lane's 16-sample run on a run of the
// tile layout) = kMixLanes lanes of 16 consecutive samples for the NCO and the mixer, and = kRows rows of 3 L samples for the
// FIR: lane l then owns the three outputs whose newest-but-one sample lies in row l.
#ifndef SDRX_LATE5_STRIDE
#define SDRX_LATE5_STRIDE 18 // (experiment: 17 -> 9.3 KB of LDS per wave = 17 instead of 16 waves per CU, single b64 window reads; profiles/README.md round 6)
#endif
template <int L>
struct LateGeom;
template <>
struct LateGeom<5> {
    static constexpr int kTaps = 49, kRow = 15, kRows = 64, kChunkLen = 960, kMixLanes = 60;
    static constexpr int kStride = SDRX_LATE5_STRIDE; // LDS row stride in samples (18): 9 slots of 16 bytes -> the 16 lanes of a ds_read_b128 group hit 16 distinct slots,
                                          //   and the 16 lanes of a ds_write_b64 group (pad 3: 6 dwords per row crossed) 16 distinct bank pairs
    static constexpr int kCarryRows = 4;  // rows of the previous chunk a window reaches back into: 4 x 15 >= 49
    static constexpr int kWarm = 80;      // a segment that starts inside the frame walks this many samples first: >= kTaps, a multiple of 16 L
};
template <>
struct LateGeom<6> {
    static constexpr int kTaps = 73, kRow = 18, kRows = 56, kChunkLen = 1008, kMixLanes = 63;
    static constexpr int kStride = 19;    // 38 dwords
    static constexpr int kCarryRows = 5;  // 5 x 18 >= 73
    static constexpr int kWarm = 96;
};
template <int L>
__host__ __device__ constexpr int late_hist() { return LateGeom<L>::kCarryRows * LateGeom<L>::kRow; } // samples carried between frames
constexpr int kLateTapPad = 80; // the taps in LDS, zero-padded to whole float4s
template <int L>
__host__ __device__ constexpr int late_window_bytes() // [carry rows | chunk rows], rounded up to 16 bytes
{
    return (8 * (LateGeom<L>::kCarryRows + LateGeom<L>::kRows) * LateGeom<L>::kStride + 15) / 16 * 16;
}
template <int L>
__host__ __device__ constexpr int late_lds_bytes() { return late_window_bytes<L>() + 4 * kLateTapPad; }

// One wave's job: samples [s_begin, s_end) of one VFO-frame, walked in 1024-sample chunks (all three
// are multiples of 16 and of 2^d; s_end - s_begin is a whole number of chunks except at the frame's
// end).  A segment that starts inside the frame starts from zero filter state: its outputs are exact
// from input position s_first_out = s_begin + warm-up on, and only those are emitted.
struct K1Work {
    int vfo;
    int s_begin, s_first_out, s_end;
};

// One block's job in the block-per-tile kernels (late decimation, demod, compress).
struct BlockWork {
    int vfo; // index into that kernel's descriptor array
    int blk; // tile index within the VFO-frame
};

// ---- late decimation by L (vfo::usb_decimdemod, vfo.cpp:334-387) ------------------------------
struct K2aVfo {
    const float2 *x[2];     // [hist Hx | data n] per parity
    float2 *x_next[2];      // the other parity's buffer (history for the next frame)
    float2 *z[2];           // output stream data base (after ITS history) per parity
    const float *taps;      // Nd taps
    int Hx, n, ndec, L;
    int n_out;              // n / L
    int pad_;
};

// ---- USB demod + audio low-pass + int16 (vfo::usb_demod, vfo.cpp:300-332) ---------------------
struct K2Vfo {
    const float2 *s[2];     // [hist H | data n] per parity
    float2 *s_next[2];
    const float *hnz;       // the 62 non-zero Hilbert taps hp[1], hp[3], ..., hp[123]
    const float *lpf_pad;   // audio low-pass taps with 3 zeros in front and >= 8 behind, or null
    short *pay[2];          // n int16, per frame parity (frame f's payload is copied out while f+1 is computed)
    float *prequant;        // optional n floats
    float gain;
    int H, n, nlpf;
    int tile;               // outputs per block: 1024, or 1024 - E with the low-pass (E = nlpf rounded up to even)
    int pad_;
    float *usb_out[2];      // a low-pass longer than kMaxFir: the unfiltered usb floats go here per frame parity
                            //   (behind that stream's history) and k_lpf_long does the rest; else null
    const float *hnz_e, *hnz_o; // the same taps for the packed MACs of the non-exact arithmetics (hilbert4_packed): hnz_e[m] = h[m-3],
                            //   hnz_o[m] = h[m-2] with h[s] = hnz[s], zeros outside 0 .. 61; 96 floats each
    float *state[2];        // a leaf that demodulates in its mix wave: its demodulation history per frame parity (256 floats:
                            //   the last 62 odd / 62 even Q, 62 I and Nh usb values at 64-float strides); else null
};
static_assert(sizeof(K2Vfo) % 8 == 0, "K2Vfo array stride");

// ---- compress() for childless non-USB VFOs (vfo.cpp:389-424) ----------------------------------
struct K3Vfo {
    const float2 *s[2];
    signed char *pay[2];    // per frame parity
    int n, cstyle, scalecomp;
    int pad_;
};

// ---- audio low-pass of more than kMaxFir taps (FIR::FIRUpdateAndProcess, dsp.cpp:59-71) + int16 --------
struct K4Vfo {
    const float *u[2];      // [hist Hu | data n] usb floats per frame parity
    float *u_next[2];       // the other parity's buffer (history for the next frame)
    const float *taps;      // nlpf taps
    short *pay[2];
    float *prequant;
    float gain;
    int Hu, n, nlpf;
};

struct NcoInit {
    float2 *cp;
    float rot_re, rot_im;
    int L;
    int pad_;
};

} // namespace sdrx
