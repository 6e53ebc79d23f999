// sdrx.hip -- host side of libsdrx.so: the C ABI of include/sdrx.h, VFO-tree bookkeeping,
// HBM layout, and the per-frame launch sequence.  The arithmetic lives in kernels.hip.
//
// Data layout in HBM (all in one arena, zeroed at finalize == the reference's zero start state):
//   per VFO   NCO checkpoints  (L/16+1) cf32           cp[j] = table[16j-1]
//             half-band state  2 x d x 10 cf32         ping-pong by frame parity
//             stream           2 x (H + n/2^d) cf32    decimate[d] of the current frame behind H
//                                                       history samples of the previous frame
//                                                       (H = 0 for VFOs that only feed children)
//             late-dec stream  2 x (H' + n_out) cf32   only for lateDecimate leaves
//   per leaf  payload          n_out int16 | n (or 2n) int8, packed in one buffer => one D2H copy
// Frame f reads state[f&1] and writes state[(f+1)&1]; nothing is copied between frames.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <utility>
#include <vector>

#include "../../include/sdrx.h"
#include "kernels.hip"
#include "tapdesign.h"

using namespace sdrx;

static_assert(sizeof(sdrx_vfo_desc) == 56 && offsetof(sdrx_vfo_desc, topic) == 48, "sdrx_vfo_desc ABI layout");
static_assert(sizeof(sdrx_stats) == 80, "sdrx_stats ABI layout");

namespace {

thread_local std::string g_create_error;

enum Kind { KIND_MIX_ROOT = 0, KIND_MIX_SUB = 1, KIND_LATE_DEC = 2, KIND_DEMOD = 3, KIND_COMPRESS = 4, KIND_INGEST = 5, KIND_LEVELS = 6, KIND_LPF_LONG = 7 };
const char *kKindNames[SDRX_NKERNELS] = {"k_mix_decimate(level0)", "k_mix_decimate(sub)", "k_late_decimate", "k_usb_demod",
                                         "k_compress",             "k_ingest",            "k_mix_levels",    "k_lpf_long"};

struct Node {
    sdrx_vfo_desc d;
    std::vector<int> children;
    int level = 0;
    int n_f = 0;       // samples of decimate[d] per frame
    int n_out = 0;     // after late decimation
    unsigned rate = 0; // outputRate
    bool leaf = false;
    // designed taps (host copies for sdrx_get_taps)
    std::vector<float> lpf, dec, hilbert;
    std::vector<float> lpf_pad, hnz; // device forms: zero-padded low-pass, compacted Hilbert
    std::vector<float> hnz_e, hnz_o; //   ... and the compacted Hilbert taps shifted by 3 / 2 in 96 zero-padded floats (hilbert4_packed)
    size_t off_hnz_e = 0, off_hnz_o = 0;
    int demod_tile = 1024;           // outputs per k_usb_demod block
    bool long_lpf = false;           // audio low-pass of more than kMaxFir taps: applied by k_lpf_long
    int Hu = 0;                      // its history length (usb floats of the previous frame)
    size_t off_u[2] = {0, 0};        // its input: [hist Hu | data n_out] usb floats per frame parity
    // device placement (byte offsets into the arena)
    size_t off_cp = 0, off_hb[2] = {0, 0}, off_stream[2] = {0, 0}, off_z[2] = {0, 0}, off_preq = 0;
    size_t off_lpf = 0, off_dec = 0, off_hilbert = 0, off_hnz = 0;
    int H = 0, Hx = 0;
    size_t pay_off = 0; // into the payload buffer
    uint32_t pay_len = 0;
    float rot_re = 0, rot_im = 0;
    int fused_late = 0;     // 5 | 6: the late decimation runs inside the mix wave (late_item); 0: not
    bool fused_demod = false; // the USB demodulation runs inside the mix wave (demod_chunk): the leaf writes its payload itself
    size_t off_dstate[2] = {0, 0}; //   ... its demodulation history per frame parity (kDemodStateFloats floats)
    int d2_index = -1;      // its K2Vfo in the demodulation descriptor array
    bool has_stream = true; // decimate[d] of every frame is kept in HBM (false: a fused late decimation writes only z', a fused demodulation only the payload)
};

struct Launch1 { // one k_mix_decimate launch (a tree level)
    int kind;
    int level;
    int n_work;
    int lds_bytes;
    size_t off_work; // arena offset of K1Work[]
    int64_t alg_bytes;
};
struct LaunchB { // block-per-tile launches (late decimate / demod / compress): one launch per kernel
    int kind;
    int n_blocks;
    size_t off_desc, off_work;
    int lds_bytes;
    int64_t alg_bytes;
};

// The one-launch levels (k_mix_levels): the list is [level 0 items | level 1 items | ...], every part
// starting at a multiple of 8 entries.  A launch covers the contiguous range of the levels that have a
// frame to work on.
struct LevelPlan {
    bool usable = false;
    size_t off_items = 0, off_item_level = 0, off_list = 0; // arena offsets
    std::vector<int> part_begin, part_end;                   // list range of every level
    std::vector<int64_t> part_bytes;                         // SURVEY 8d share of every level
    int lds_bytes = 0;
};
struct InFlight { // a frame inside the software pipeline: `next` = the level that runs it in the next launch
    unsigned long long f;
    int next;
};

struct TimedEvent {
    hipEvent_t a, b;
    int kind;
    int64_t bytes;
};

} // namespace

struct sdrx_ctx {
    int device = 0;
    std::string err;
    std::vector<Node> nodes;
    bool finalized = false;
    int opt_exact = 1, opt_prequant = 0, opt_segments = 0, opt_dc_blocked = 0, opt_pipeline = 0, opt_dc_speculative = 1;
    int opt_fuse = 1, opt_frame_pipeline = 1, opt_fuse_late = 1, opt_keep_streams = 0, opt_fuse_demod = 0;
    // sdrx_set_tap / sdrx_add_tap: the fused late-decimation leaves that keep decimate[0] because they are taps (vfo::fftVFOSlot
    // sets emitFFT on EVERY VFO whose topic matches, vfo.cpp:492-509): node -> its buffers per frame parity and the first
    // frame that fills them.  The first such leaf uses the arena's buffer, further ones buffers of their own (hipMalloc).
    struct TapBuf {
        float2 *buf[2] = {nullptr, nullptr};
        unsigned long long since = 0;
        bool own = false;
    };
    std::map<int, TapBuf> taps;
    size_t tap_len = 0, off_tapbuf[2] = {0, 0}; // the arena's tap buffer (sized for the longest fused leaf), per frame parity
    LevelPlan fp;
    std::vector<InFlight> pipe; // oldest first
    sdrx_publish_fn cb = nullptr;
    void *cb_user = nullptr;

    // Streams.  `stream` (the context's own or the caller's) carries the ingest and the
    // mix/decimate launches of every tree level; the leaf tail of a frame runs on `tail_stream`
    // when option "pipeline" is on (off by default: measured slower, profiles/README.md); payloads leave on
    // `copy_stream` for frames that came in through sdrx_submit*.  Cross-stream order is by the
    // per-parity events below (measured on this runtime, tools/event_probe.hip: a record costs its
    // stream ~3-5 us, a wait on an event that completed long ago ~2.5 us, a tight hop ~11 us).
    hipStream_t own_stream = nullptr, stream = nullptr, tail_stream = nullptr, copy_stream = nullptr, copy_stream2 = nullptr;
    int late_copy = -1;                    // payload copy issued by sdrx_wait instead of queued behind the frame: -1 = for frames that carry the DC recurrence, 0 never, 1 always (SDRX_LATE_COPY)
    bool copy_owed[2] = {false, false};    //   ... and not issued yet
    bool upload_kernel = false; // SDRX_UPLOAD_KERNEL=1: host frames go up with k_copy16 instead of hipMemcpyAsync (A/B switch; slower)
    int download_blocks = 0;    // workgroups of a payload copy KERNEL for the frames that carry the recurrence (SDRX_DOWNLOAD_BLOCKS with SDRX_LATE_COPY=0; 0: hipMemcpyAsync) ...
    bool long_frame = false;    // ... which carries the payloads of frames whose kernels outlast the copy (the DC-bias recurrence)
    hipEvent_t ev_levels[2] = {nullptr, nullptr}; // levels of frame f done (recorded on `stream`)
    hipEvent_t ev_tail[2] = {nullptr, nullptr};   // tail of frame f done (recorded on the tail's stream)
    hipEvent_t ev_copied[2] = {nullptr, nullptr}; // payloads of frame f are in h_pay[f & 1]
    bool tail_recorded[2] = {false, false};
    unsigned char *arena = nullptr;
    size_t arena_bytes = 0;
    unsigned char *d_pay[2] = {nullptr, nullptr}, *h_pay[2] = {nullptr, nullptr}; // per frame parity
    size_t pay_bytes = 0;
    unsigned char *h_in[2] = {nullptr, nullptr}; // pinned staging of host-fed frames, per frame parity
    size_t h_in_bytes = 0;
    int in_flight = 0;               // frames submitted (sdrx_submit*) and not yet delivered (sdrx_wait)
    bool broken = false;             // fault injection (SDRX_FAULT_WAIT): every frame call fails from here on, like after a HIP error
    int host_slot = -1;              // which h_pay holds the payloads sdrx_get_output serves
    float2 *d_raw[2] = {nullptr, nullptr}; // host-fed frames on the device (natural order), per frame parity: frame f's
                                           //   buffer stays untouched until f+2 is staged (another context on this device may
                                           //   be working on it: sdrx_submit_shared)
    hipEvent_t ev_staged[2] = {nullptr, nullptr}; // the host frame of parity p is complete on the device
    // other contexts that ran on this context's uploaded frame of parity p (sdrx_submit_shared): each left an event behind its
    // kernels, and this context's next upload into that buffer waits for them (events owned, and reused, by this context).
    // No lock: `ctx` and `src` of a sharing call must be driven from ONE thread (sdrx.h).
    struct SharedReader {
        const sdrx_ctx *who; // (identity only: never dereferenced)
        hipEvent_t ev;       // behind who's kernels on this context's frame of that parity; owned by THIS context
        bool pending;        // recorded since this context last waited for it
    };
    std::vector<SharedReader> shared_readers[2]; // at most one entry per (reader, parity): re-recorded, never piled up
    float2 *d_raw_tiled = nullptr; // the raw frame in tile layout: input of the parent-less VFOs
    int last_raw = -1;             // how the last frame reached level 0 (kRaw*; -1: caller-owned device memory)
    bool late4 = false;            // k_late_decimate4 serves the late-decimation launch
    int late4_r = 4;               // outputs per lane of that kernel
    bool root_direct = false;      // level 0 reads the caller's natural-order frame itself (few VFOs)
    unsigned char *d_raw_u8[2] = {nullptr, nullptr}; // the same for dongle bytes
    float *d_dc_state = nullptr;   // DC-bias accumulator (exact: [2]; fast: [parity][2])
    float *d_dc_work = nullptr;    // exact DC-bias removal: products P[2][stride] and estimates A[2][stride] of one frame
    unsigned long long *d_dc_counters = nullptr; // k_dc_chain_spec: [0] blocks walked, [1] blocks redone with the sequential operations, [2] blocks taken again on their own
    int dc_waves = 8;                            // k_dc_chain_spec: blocks per step = waves per workgroup (SDRX_DC_WAVES: 1, 2, 4, 8)
    int dc_rounds = 0;                           //   ... its limit of rounds per step (SDRX_DC_ROUNDS; 0 = kDcMaxIter)
    int dc_work_stride = 0;
    double *d_dc_tab = nullptr;    // fast DC scan: powers of the decay + per-chunk sums behind them
    unsigned long long dc_frames = 0; // frames the fast scan has run on (its state ping-pongs)
    size_t dc_tab_sums = 0;        // offset (in doubles) of the double2 sums[] inside d_dc_tab
    size_t raw_cap = 0;
    int root_frame = 0; // samples_per_buffer of the parent-less VFOs
    size_t off_k1vfo = 0;
    std::vector<Launch1> l1;
    std::vector<LaunchB> lb;
    std::vector<int> publish_order;
    unsigned long long frame_no = 0;
    bool pending_fetch = false;
    int64_t alg_bytes = 0, vfo_samples = 0, mix_chunks = 0;
    int n_levels = 0;

    bool timing = false;
    std::vector<TimedEvent> pending_events;
    std::vector<hipEvent_t> event_pool;
    double t_ms[SDRX_NKERNELS] = {0};
    int64_t t_n[SDRX_NKERNELS] = {0};
    int64_t t_bytes[SDRX_NKERNELS] = {0};
};

namespace {

int fail(sdrx_ctx *c, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c)
        c->err = buf;
    else
        g_create_error = buf;
    return code;
}

#define HIPCHK(c, expr)                                                                         \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return fail((c), SDRX_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct ArenaPlan {
    size_t size = 0;
    size_t take(size_t bytes)
    {
        size_t o = align_up(size, 256);
        size = o + bytes;
        return o;
    }
};

// A segment that starts inside the frame starts from zero filter state.  An output of stage d with
// index j (counted from the segment's first sample) depends on the inputs 2^d j - 10 (2^d - 1) ... 2^d j,
// so it is exact once j >= 10 - 10 / 2^d: the first ceil(..) outputs of a segment are warm-up and are
// not emitted.  Returned in input samples, rounded up to a multiple of 16 (the emit test of the
// register stages is per lane = per 16 samples); always a multiple of 2^d.
int warmup_samples(int d)
{
    if (d <= 0)
        return 0;
    const int vd = (10 * ((1 << d) - 1) + (1 << d) - 1) >> d;        // ceil(10 (2^d - 1) / 2^d)
    int w = vd << d;
    while (w & 15)
        w += 1 << d;
    return w;
}

hipEvent_t get_event(sdrx_ctx *c)
{
    if (!c->event_pool.empty()) {
        hipEvent_t e = c->event_pool.back();
        c->event_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess)
        return nullptr; // the launch is then simply not timed
    return e;
}

void drain_events(sdrx_ctx *c)
{
    for (auto &te : c->pending_events) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, te.a, te.b) == hipSuccess) {
            c->t_ms[te.kind] += ms;
            c->t_n[te.kind] += 1;
            c->t_bytes[te.kind] += te.bytes;
        }
        c->event_pool.push_back(te.a);
        c->event_pool.push_back(te.b);
    }
    c->pending_events.clear();
}

struct Bracket { // RAII: event pair around one launch when timing is on, on the launch's stream
    sdrx_ctx *c;
    hipStream_t st;
    TimedEvent te{};
    bool on;
    Bracket(sdrx_ctx *ctx, hipStream_t stream, int kind, int64_t bytes) : c(ctx), st(stream), on(ctx->timing)
    {
        if (!on)
            return;
        te.kind = kind;
        te.bytes = bytes;
        te.a = get_event(c);
        te.b = get_event(c);
        if (!te.a || !te.b) {
            on = false;
            return;
        }
        (void)hipEventRecord(te.a, st);
    }
    ~Bracket()
    {
        if (!on)
            return;
        (void)hipEventRecord(te.b, st);
        c->pending_events.push_back(te);
    }
};

int pipeline_step(sdrx_ctx *c, bool have_new, const void *raw, int raw_mode);
int pipeline_flush(sdrx_ctx *c);
inline int pipeline_flush_unless(sdrx_ctx *c, bool keep) { return keep ? SDRX_OK : pipeline_flush(c); }
void launch_block_kernel(sdrx_ctx *c, const LaunchB &L, hipStream_t ts, unsigned long long frame, bool exact);

// One frame: [wait for the tail of frame f-2] -> ingest -> one k_mix_decimate launch per tree level on
// `stream`; then the leaf tail (late decimation, demodulation, compress) -- on `tail_stream` behind an
// event when the pipeline option is on, so that it runs beside the NEXT frame's levels -- and, for a
// frame that came through sdrx_submit*, the payload copy on `copy_stream` behind the tail.
//
// What makes the two-stream form safe (frame f, parity p = f & 1):
//   * the leaf streams of parity p are written by the levels of f and read by the tail of f: the tail
//     waits for ev_levels[p]; their next writer is frame f+2, whose levels wait for ev_tail[p] first;
//   * the history prefix of the parity-(p^1) leaf streams is written by the tail of f and read by the
//     tail of f+1: same stream, in order (the levels of f+1 write only the data part behind it);
//   * half-band state, NCO tables and the parents' streams are touched by the levels only;
//   * d_pay[p] is written by the tail of f and read by the copy of f; its next writer is the tail of
//     f+2, which the host does not submit before frame f was delivered (SDRX_MAX_IN_FLIGHT = 2).
// ARITH = option "exact": 1 the exact arithmetic, 0 the tolerance arithmetic (NCO as rotations), 2 the robust one (exact NCO,
// FMA mixer and filters) -- kernels.hip, nco_mix.
template <int ARITH>
int enqueue_frame_as(sdrx_ctx *c, const void *raw, int raw_mode, bool egress)
{
    constexpr bool EXACT = ARITH == 1, ROT = ARITH == 0;
    const K1Vfo *k1 = reinterpret_cast<const K1Vfo *>(c->arena + c->off_k1vfo);
    const int p = (int)(c->frame_no & 1ull);
    const bool pipe = c->opt_pipeline != 0;
    if (pipe && c->tail_recorded[p])
        HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_tail[p], 0));
    // A few parent-less VFOs (the reference's 2-3 mains) read the caller's frame as it is; a wide
    // level 0 (the flat workloads) is bandwidth bound and wants coalesced reads: one layout pass
    // natural order -> tile layout first.
    if (raw_mode != kRawTiled && !c->root_direct) {
        Bracket b(c, c->stream, KIND_INGEST, 0);
        const int n_pairs = c->root_frame / 2;
        if (raw_mode == kRawF32)
            hipLaunchKernelGGL(k_ingest_f32, dim3((n_pairs + 255) / 256), dim3(256), 0, c->stream,
                               reinterpret_cast<const float4 *>(raw), reinterpret_cast<float4 *>(c->d_raw_tiled), n_pairs);
        else
            hipLaunchKernelGGL(k_ingest_u8, dim3((n_pairs + 255) / 256), dim3(256), 0, c->stream,
                               reinterpret_cast<const unsigned *>(raw), reinterpret_cast<float4 *>(c->d_raw_tiled), n_pairs);
        raw_mode = kRawTiled;
    }
    if (int rc = pipeline_flush_unless(c, c->fp.usable && c->opt_fuse && !pipe && !egress))
        return rc;
    if (c->fp.usable && c->opt_fuse && !pipe && !egress) {
        // Frames that stay on the device and are queued back to back share launches: level l of frame
        // k - l runs in the launch that frame k enters with, and the frame that leaves the last level
        // gets its leaf tail right behind it.  (A frame whose payloads must leave now -- sdrx_process*,
        // sdrx_submit* -- runs through its own launches below: nothing to overlap it with.)
        const int rc = pipeline_step(c, true, raw, raw_mode);
        if (rc)
            return rc;
        if (!c->opt_frame_pipeline)
            if (int rc2 = pipeline_flush(c))
                return rc2;
        c->frame_no++;
        c->pending_fetch = true;
        return SDRX_OK;
    }
    for (const Launch1 &L : c->l1) {
        Bracket b(c, c->stream, L.kind, L.alg_bytes);
        const K1Work *w = reinterpret_cast<const K1Work *>(c->arena + L.off_work);
        if (L.level == 0)
            hipLaunchKernelGGL((k_mix_decimate<EXACT, 0, ROT>), dim3(L.n_work), dim3(64), L.lds_bytes, c->stream, k1, w, c->frame_no, raw, raw_mode);
        else
            hipLaunchKernelGGL((k_mix_decimate<EXACT, 1, ROT>), dim3(L.n_work), dim3(64), L.lds_bytes, c->stream, k1, w, c->frame_no,
                               (const void *)nullptr, kRawTiled);
    }
    hipStream_t ts = pipe ? c->tail_stream : c->stream;
    if (pipe) {
        HIPCHK(c, hipEventRecord(c->ev_levels[p], c->stream));
        HIPCHK(c, hipStreamWaitEvent(ts, c->ev_levels[p], 0));
    }
    for (const LaunchB &L : c->lb)
        launch_block_kernel(c, L, ts, c->frame_no, EXACT);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
        return fail(c, SDRX_EHIP, "kernel launch failed: %s", hipGetErrorString(e));
    if (pipe || egress) {
        HIPCHK(c, hipEventRecord(c->ev_tail[p], ts));
        c->tail_recorded[p] = pipe;
    }
    // How the payloads leave (measured, traced: profiles/README.md round 5; tools/copy_overlap_probe.hip).  Normally queued here,
    // behind the frame's last kernel: hipMemcpyAsync on a copy stream, which the runtime hands to an SDMA engine -- the kernels of
    // the next frame run beside it, config 3 from host floats goes at PCIe speed (0.30 ms per pipelined frame = the 15 MB copy).
    // A frame that carries the DC-bias recurrence did not get to run beside a copy queued that way (0.59-0.82 ms per frame,
    // about the SUM of its parts), nor beside a copy kernel of ours (a kernel cannot retire beside one): for those frames the copy
    // is issued by sdrx_wait, when the host has seen the frame's last kernel end -- 0.34 ms per frame.  (The float path would lose
    // by that, 0.38 vs 0.30: between two waits the copy engine idles.)  SDRX_LATE_COPY=0 / 1: never / always.
    if (egress && (c->late_copy == 1 || (c->late_copy < 0 && c->long_frame))) {
        c->copy_owed[p] = true;
        c->in_flight++;
    } else if (egress) {
        hipStream_t cs = (p && c->copy_stream2) ? c->copy_stream2 : c->copy_stream;
        HIPCHK(c, hipStreamWaitEvent(cs, c->ev_tail[p], 0));
        // (SDRX_DOWNLOAD_BLOCKS=n with SDRX_LATE_COPY=0: a copy kernel of ours on n workgroups for the frames that carry the recurrence -- A/B switch)
        if (c->download_blocks > 0 && c->long_frame) {
            const size_t n16 = (c->pay_bytes + 15) / 16; // (both buffers are allocated in whole 16-byte units)
            hipLaunchKernelGGL(k_copy16, dim3(c->download_blocks), dim3(256), 0, cs, reinterpret_cast<const uint4 *>(c->d_pay[p]),
                               reinterpret_cast<uint4 *>(c->h_pay[p]), n16);
        } else {
            HIPCHK(c, hipMemcpyAsync(c->h_pay[p], c->d_pay[p], c->pay_bytes, hipMemcpyDeviceToHost, cs));
        }
        HIPCHK(c, hipEventRecord(c->ev_copied[p], cs));
        c->in_flight++;
    }
    c->frame_no++;
    c->pending_fetch = !egress;
    return SDRX_OK;
}

int enqueue_frame(sdrx_ctx *c, const void *raw, int raw_mode, bool egress)
{
    return c->opt_exact == 1 ? enqueue_frame_as<1>(c, raw, raw_mode, egress)
           : c->opt_exact == 2 ? enqueue_frame_as<2>(c, raw, raw_mode, egress)
                               : enqueue_frame_as<0>(c, raw, raw_mode, egress);
}

void launch_block_kernel(sdrx_ctx *c, const LaunchB &L, hipStream_t ts, unsigned long long frame, bool exact)
{
    Bracket b(c, ts, L.kind, L.alg_bytes);
    const dim3 grid(L.n_blocks);
    const BlockWork *w = reinterpret_cast<const BlockWork *>(c->arena + L.off_work);
    const K2aVfo *k2a = reinterpret_cast<const K2aVfo *>(c->arena + L.off_desc);
    if (L.kind == KIND_LATE_DEC && c->late4) {
        if (c->late4_r == 2) {
            if (exact)
                hipLaunchKernelGGL((k_late_decimate4<true, 2>), grid, dim3(64), L.lds_bytes, ts, k2a, w, frame);
            else
                hipLaunchKernelGGL((k_late_decimate4<false, 2>), grid, dim3(64), L.lds_bytes, ts, k2a, w, frame);
        } else {
            if (exact)
                hipLaunchKernelGGL((k_late_decimate4<true, 4>), grid, dim3(64), L.lds_bytes, ts, k2a, w, frame);
            else
                hipLaunchKernelGGL((k_late_decimate4<false, 4>), grid, dim3(64), L.lds_bytes, ts, k2a, w, frame);
        }
    } else if (L.kind == KIND_LATE_DEC) {
        if (exact)
            hipLaunchKernelGGL(k_late_decimate<true>, grid, dim3(256), L.lds_bytes, ts, k2a, w, frame);
        else
            hipLaunchKernelGGL(k_late_decimate<false>, grid, dim3(256), L.lds_bytes, ts, k2a, w, frame);
    } else if (L.kind == KIND_DEMOD) {
        const K2Vfo *k2 = reinterpret_cast<const K2Vfo *>(c->arena + L.off_desc);
        if (exact)
            hipLaunchKernelGGL(k_usb_demod<true>, grid, dim3(256), 0, ts, k2, w, frame);
        else
            hipLaunchKernelGGL(k_usb_demod<false>, grid, dim3(256), 0, ts, k2, w, frame);
    } else if (L.kind == KIND_LPF_LONG) {
        const K4Vfo *k4 = reinterpret_cast<const K4Vfo *>(c->arena + L.off_desc);
        if (exact)
            hipLaunchKernelGGL(k_lpf_long<true>, grid, dim3(256), L.lds_bytes, ts, k4, w, frame);
        else
            hipLaunchKernelGGL(k_lpf_long<false>, grid, dim3(256), L.lds_bytes, ts, k4, w, frame);
    } else {
        hipLaunchKernelGGL(k_compress, grid, dim3(256), 0, ts, reinterpret_cast<const K3Vfo *>(c->arena + L.off_desc), w, frame);
    }
}

// One step of the frame pipeline: every in-flight frame (and the new one, if `have_new`) moves through
// the tree level it has reached -- ONE k_mix_levels launch over the contiguous range of those levels --
// and the frame that thereby leaves the last level gets its leaf tail (late decimation, demodulation,
// compress) right behind that launch.
int pipeline_step(sdrx_ctx *c, bool have_new, const void *raw, int raw_mode)
{
    const LevelPlan &P = c->fp;
    const int n_levels = (int)P.part_begin.size();
    if (have_new)
        c->pipe.push_back({c->frame_no, 0});
    if (c->pipe.empty())
        return SDRX_OK;
    LevelArgs A;
    memset(&A, 0, sizeof A);
    A.raw = raw;
    A.raw_mode = raw_mode;
    int lo = n_levels, hi = -1;
    for (const InFlight &q : c->pipe) {
        lo = std::min(lo, q.next);
        hi = std::max(hi, q.next);
        A.frame_level[q.next] = q.f;
    }
    const int first = std::min(P.part_begin[(size_t)lo], P.part_begin[(size_t)hi]), last = std::max(P.part_end[(size_t)lo], P.part_end[(size_t)hi]);
    {
        int64_t bytes = 0;
        for (int j = lo; j <= hi; ++j)
            bytes += P.part_bytes[(size_t)j];
        Bracket b(c, c->stream, lo != hi ? KIND_LEVELS : lo == 0 ? KIND_MIX_ROOT : KIND_MIX_SUB, bytes);
        const K1Vfo *k1 = reinterpret_cast<const K1Vfo *>(c->arena + c->off_k1vfo);
        const K1Work *items = reinterpret_cast<const K1Work *>(c->arena + P.off_items);
        const int *item_level = reinterpret_cast<const int *>(c->arena + P.off_item_level);
        const int *list = reinterpret_cast<const int *>(c->arena + P.off_list) + first;
        if (c->opt_exact == 1)
            hipLaunchKernelGGL((k_mix_levels<true, false>), dim3(last - first), dim3(64), P.lds_bytes, c->stream, k1, items, item_level, list, A);
        else if (c->opt_exact == 2)
            hipLaunchKernelGGL((k_mix_levels<false, false>), dim3(last - first), dim3(64), P.lds_bytes, c->stream, k1, items, item_level, list, A);
        else
            hipLaunchKernelGGL((k_mix_levels<false, true>), dim3(last - first), dim3(64), P.lds_bytes, c->stream, k1, items, item_level, list, A);
    }
    for (InFlight &q : c->pipe)
        q.next++;
    if (c->pipe.front().next >= n_levels) { // the oldest frame has passed its last level: its leaf tail, now
        const unsigned long long f = c->pipe.front().f;
        for (const LaunchB &L : c->lb)
            launch_block_kernel(c, L, c->stream, f, c->opt_exact == 1);
        c->pipe.erase(c->pipe.begin());
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
        return fail(c, SDRX_EHIP, "kernel launch failed: %s", hipGetErrorString(e));
    return SDRX_OK;
}

// run every in-flight frame to its end
int pipeline_flush(sdrx_ctx *c)
{
    while (!c->pipe.empty()) {
        const int rc = pipeline_step(c, false, nullptr, kRawTiled);
        if (rc)
            return rc;
    }
    return SDRX_OK;
}

// every frame handed to the context is complete and every stream of the context idle afterwards
int drain(sdrx_ctx *c)
{
    if (int rc = pipeline_flush(c))
        return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipStreamSynchronize(c->tail_stream));
    HIPCHK(c, hipStreamSynchronize(c->copy_stream));
    if (c->copy_stream2)
        HIPCHK(c, hipStreamSynchronize(c->copy_stream2));
    drain_events(c);
    return SDRX_OK;
}

// vfo::transmitData for every leaf, in the reference's order (vfo.cpp:426-453, sdrj.cpp:288-294)
void publish_all(sdrx_ctx *c, int slot)
{
    c->host_slot = slot;
    if (!c->cb)
        return;
    for (int i : c->publish_order) {
        const Node &n = c->nodes[(size_t)i];
        // USB leaves always publish; an IQ leaf only with a topic; ZmqPublisher::publish sends
        // nothing for len 0 (zmqpublisher.cpp:88).
        if (n.pay_len == 0)
            continue;
        if (!n.d.demod_usb && n.d.topic[0] == 0)
            continue;
        char topic[5] = {0, 0, 0, 0, 0};
        for (int k = 0; k < 5 && n.d.topic[k]; ++k)
            topic[k] = n.d.topic[k];
        c->cb(c->cb_user, topic, n.rate, c->h_pay[slot] + n.pay_off, n.pay_len);
    }
}

void free_device_state(sdrx_ctx *c)
{
    auto dfree = [](auto *&p) {
        if (p)
            (void)hipFree(p);
        p = nullptr;
    };
    auto hfree = [](unsigned char *&p) {
        if (p)
            (void)hipHostFree(p);
        p = nullptr;
    };
    for (auto &kv : c->taps)
        if (kv.second.own)
            for (int p = 0; p < 2; ++p)
                (void)hipFree(kv.second.buf[p]);
    c->taps.clear();
    dfree(c->arena);
    for (int p = 0; p < 2; ++p) {
        dfree(c->d_pay[p]);
        hfree(c->h_pay[p]);
        hfree(c->h_in[p]);
    }
    c->h_in_bytes = 0;
    for (int p = 0; p < 2; ++p) {
        dfree(c->d_raw[p]);
        dfree(c->d_raw_u8[p]);
    }
    dfree(c->d_raw_tiled);
    dfree(c->d_dc_state);
    dfree(c->d_dc_work);
    dfree(c->d_dc_counters);
    c->dc_work_stride = 0;
    dfree(c->d_dc_tab);
    c->raw_cap = 0;
}

int ensure_raw(sdrx_ctx *c, size_t n_complex)
{
    if (c->raw_cap >= n_complex)
        return SDRX_OK;
    for (int p = 0; p < 2; ++p) {
        if (c->d_raw[p])
            (void)hipFree(c->d_raw[p]);
        if (c->d_raw_u8[p])
            (void)hipFree(c->d_raw_u8[p]);
        c->d_raw[p] = nullptr;
        c->d_raw_u8[p] = nullptr;
        HIPCHK(c, hipMalloc(&c->d_raw[p], n_complex * sizeof(float2)));
        HIPCHK(c, hipMalloc(&c->d_raw_u8[p], n_complex * 2));
    }
    c->raw_cap = n_complex;
    return SDRX_OK;
}

} // namespace

// ================================================================================ C ABI
extern "C" {

int sdrx_abi_version(void) { return SDRX_ABI_VERSION; }

#ifndef SDRX_SOURCE_HASH
#define SDRX_SOURCE_HASH "unknown"
#endif
const char *sdrx_build_id(void) { return SDRX_SOURCE_HASH; }

const char *sdrx_last_error(const sdrx_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

const char *sdrx_kernel_name(int kind) { return kind >= 0 && kind < SDRX_NKERNELS ? kKindNames[kind] : ""; }

int sdrx_create(sdrx_ctx **out, int device)
{
    if (!out)
        return fail(nullptr, SDRX_EINVAL, "sdrx_create: null output pointer");
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(nullptr, SDRX_EHIP, "sdrx_create: no HIP device (%s); libsdrx has no CPU fallback",
                    e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    if (device < 0 || device >= ndev)
        return fail(nullptr, SDRX_EINVAL, "sdrx_create: device %d out of range (0..%d)", device, ndev - 1);
    e = hipSetDevice(device);
    if (e != hipSuccess)
        return fail(nullptr, SDRX_EHIP, "hipSetDevice(%d): %s", device, hipGetErrorString(e));
    sdrx_ctx *c = new sdrx_ctx();
    c->device = device;
    e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete c;
        return fail(nullptr, SDRX_EHIP, "hipStreamCreate: %s", hipGetErrorString(e));
    }
    c->stream = c->own_stream;
    bool ok = hipStreamCreateWithFlags(&c->tail_stream, hipStreamNonBlocking) == hipSuccess &&
              hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking) == hipSuccess;
    if (getenv("SDRX_LATE_COPY"))
        c->late_copy = atoi(getenv("SDRX_LATE_COPY")) != 0;
    c->upload_kernel = getenv("SDRX_UPLOAD_KERNEL") && atoi(getenv("SDRX_UPLOAD_KERNEL")) != 0;
    // experiment switches of the exact DC-bias removal, read ONCE, here, and validated like the option they shadow
    if (const char *e = getenv("SDRX_DC_WAVES")) {
        const int v = atoi(e);
        if (v != 1 && v != 2 && v != 4 && v != 8) {
            sdrx_destroy(c);
            return fail(nullptr, SDRX_EINVAL, "SDRX_DC_WAVES=%s: 1, 2, 4 or 8 (option dc_blocks_per_step)", e);
        }
        c->dc_waves = v;
    }
    if (const char *e = getenv("SDRX_DC_ROUNDS")) {
        const int v = atoi(e);
        if (v < 1 || v > 1000) {
            sdrx_destroy(c);
            return fail(nullptr, SDRX_EINVAL, "SDRX_DC_ROUNDS=%s: 1 .. 1000 rounds per step of k_dc_chain_spec", e);
        }
        c->dc_rounds = v;
    }
    if (getenv("SDRX_DOWNLOAD_BLOCKS"))
        c->download_blocks = std::max(0, std::min(4096, atoi(getenv("SDRX_DOWNLOAD_BLOCKS"))));
    // odd frames' payloads leave on a copy stream of their own: the next copy's set-up then overlaps the
    // current copy's tail (measured through the ABI on config 3: 0.296 vs 0.306 ms per frame;
    // SDRX_TWO_COPY_STREAMS=0 for A/B runs)
    if (ok && !(getenv("SDRX_TWO_COPY_STREAMS") && atoi(getenv("SDRX_TWO_COPY_STREAMS")) == 0))
        ok = hipStreamCreateWithFlags(&c->copy_stream2, hipStreamNonBlocking) == hipSuccess;
    for (int p = 0; p < 2 && ok; ++p)
        ok = hipEventCreateWithFlags(&c->ev_levels[p], hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&c->ev_tail[p], hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&c->ev_copied[p], hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&c->ev_staged[p], hipEventDisableTiming) == hipSuccess;
    if (!ok) {
        sdrx_destroy(c);
        return fail(nullptr, SDRX_EHIP, "sdrx_create: could not create the streams / events of the context");
    }
    *out = c;
    return SDRX_OK;
}

int sdrx_destroy(sdrx_ctx *c)
{
    if (!c)
        return SDRX_EINVAL;
    (void)hipSetDevice(c->device);
    if (c->stream)
        (void)hipStreamSynchronize(c->stream);
    if (c->tail_stream)
        (void)hipStreamSynchronize(c->tail_stream);
    if (c->copy_stream)
        (void)hipStreamSynchronize(c->copy_stream);
    if (c->copy_stream2)
        (void)hipStreamSynchronize(c->copy_stream2);
    drain_events(c);
    for (hipEvent_t e : c->event_pool)
        (void)hipEventDestroy(e);
    for (auto *v : {&c->shared_readers[0], &c->shared_readers[1]})
        for (const auto &r : *v)
            (void)hipEventDestroy(r.ev);
    for (int p = 0; p < 2; ++p)
        for (hipEvent_t e : {c->ev_levels[p], c->ev_tail[p], c->ev_copied[p], c->ev_staged[p]})
            if (e)
                (void)hipEventDestroy(e);
    free_device_state(c);
    for (hipStream_t st : {c->own_stream, c->tail_stream, c->copy_stream, c->copy_stream2})
        if (st)
            (void)hipStreamDestroy(st);
    delete c;
    return SDRX_OK;
}

int sdrx_set_option(sdrx_ctx *c, const char *name, int value)
{
    if (!c || !name)
        return SDRX_EINVAL;
    if (c->finalized)
        return fail(c, SDRX_ESTATE, "sdrx_set_option after sdrx_finalize");
    if (!strcmp(name, "exact"))
        c->opt_exact = value == 2 ? 2 : value != 0; // 1 exact (default) | 0 tolerance | 2 robust
    else if (!strcmp(name, "keep_prequant"))
        c->opt_prequant = value != 0;
    else if (!strcmp(name, "segments"))
        c->opt_segments = value < 0 ? 0 : value;
    else if (!strcmp(name, "dc_blocked_scan"))
        c->opt_dc_blocked = value != 0;
    else if (!strcmp(name, "dc_speculative"))
        c->opt_dc_speculative = value != 0;
    else if (!strcmp(name, "dc_blocks_per_step")) {
        if (value != 1 && value != 2 && value != 4 && value != 8)
            return fail(c, SDRX_EINVAL, "dc_blocks_per_step: 1, 2, 4 or 8");
        c->dc_waves = value;
    }
    else if (!strcmp(name, "pipeline"))
        c->opt_pipeline = value != 0;
    else if (!strcmp(name, "fuse"))
        c->opt_fuse = value != 0;
    else if (!strcmp(name, "frame_pipeline"))
        c->opt_frame_pipeline = value != 0;
    else if (!strcmp(name, "fuse_late"))
        c->opt_fuse_late = value != 0;
    else if (!strcmp(name, "keep_streams"))
        c->opt_keep_streams = value != 0;
    else if (!strcmp(name, "fuse_demod"))
        c->opt_fuse_demod = value != 0;
    else
        return fail(c, SDRX_EINVAL, "unknown option '%s'", name);
    return SDRX_OK;
}

int sdrx_add_vfo(sdrx_ctx *c, const sdrx_vfo_desc *d, int *id_out)
{
    if (!c || !d)
        return SDRX_EINVAL;
    if (c->finalized)
        return fail(c, SDRX_ESTATE, "sdrx_add_vfo after sdrx_finalize");
    const int id = (int)c->nodes.size();
    if (d->parent_id >= id || d->parent_id < -1)
        return fail(c, SDRX_EINVAL, "vfo %d: parent_id %d must name an earlier vfo or be -1", id, d->parent_id);
    if (d->decimate_count < 0 || d->decimate_count > kMaxStages)
        return fail(c, SDRX_EINVAL, "vfo %d: decimate_count %d outside 0..8 (vfo.h:63)", id, d->decimate_count);
    if (d->fs <= 0 || d->samples_per_buffer <= 0)
        return fail(c, SDRX_EINVAL, "vfo %d: fs and samples_per_buffer must be positive", id);
    if (d->late_decimate < 0 || d->late_decimate == 1)
        return fail(c, SDRX_EINVAL, "vfo %d: late_decimate must be 0 or >= 2 (the reference uses 5 and 6)", id);
    if (d->parent_id >= 0 && c->nodes[(size_t)d->parent_id].d.demod_usb)
        return fail(c, SDRX_EINVAL, "vfo %d: parent %d is a USB leaf", id, d->parent_id);
    Node n;
    n.d = *d;
    c->nodes.push_back(n);
    if (d->parent_id >= 0)
        c->nodes[(size_t)d->parent_id].children.push_back(id);
    if (id_out)
        *id_out = id;
    return SDRX_OK;
}

int sdrx_check_vfo(const sdrx_vfo_desc *d, char *msg, size_t cap)
{
    auto say = [&](int code, const char *fmt, ...) {
        if (msg && cap) {
            va_list ap;
            va_start(ap, fmt);
            vsnprintf(msg, cap, fmt, ap);
            va_end(ap);
        }
        return code;
    };
    if (!d)
        return SDRX_EINVAL;
    if (msg && cap)
        msg[0] = 0;
    if (d->decimate_count < 0 || d->decimate_count > kMaxStages)
        return say(SDRX_EINVAL, "decimate_count %d outside 0..8 (vfo.h:63)", d->decimate_count);
    if (d->fs <= 0 || d->samples_per_buffer <= 0)
        return say(SDRX_EINVAL, "fs and samples_per_buffer must be positive");
    if (d->late_decimate < 0 || d->late_decimate == 1)
        return say(SDRX_EINVAL, "late_decimate must be 0 or >= 2 (the reference uses 5 and 6)");
    // the order of vfo::init (vfo.cpp:60-124): late-decimation low-pass first, then the audio low-pass
    int target = (int)(d->fs / std::pow(2, d->decimate_count));
    if (d->demod_usb && d->late_decimate > 0) {
        target /= d->late_decimate;
        if (const char *why = low_pass_rejection((double)target * d->late_decimate, (double)(target / 2), (double)target / (d->late_decimate - 1)))
            return say(SDRX_EFILTER, "%s", why);
    }
    if (d->filter_bw_hz > 0) // (vfo::init designs this filter for ANY vfo with filterbw > 0, vfo.cpp:106-124, USB or not)
        if (const char *why = low_pass_rejection((double)target, (double)d->filter_bw_hz, (double)d->filter_bw_hz / 4))
            return say(SDRX_EFILTER, "%s", why);
    if (d->fs % kRun || d->samples_per_buffer % kRun || d->fs < kChunk)
        return say(SDRX_EUNSUPPORTED, "fs (%d) and samples_per_buffer (%d) must be multiples of 16 and fs >= 1024", d->fs, d->samples_per_buffer);
    if (d->samples_per_buffer % (1 << d->decimate_count))
        return say(SDRX_EUNSUPPORTED, "samples_per_buffer %d not a multiple of 2^%d", d->samples_per_buffer, d->decimate_count);
    return SDRX_OK;
}

int sdrx_set_publish_callback(sdrx_ctx *c, sdrx_publish_fn fn, void *user)
{
    if (!c)
        return SDRX_EINVAL;
    c->cb = fn;
    c->cb_user = user;
    return SDRX_OK;
}

} // extern "C"

// ================================================================================ sdrx_finalize
// = vfo::init for every node (vfo.cpp:60-176) plus everything the launches need, in phases that hand a `Built` to each
// other: derive_nodes (rates, tap designs, tree levels, which leaves take the fused late decimation) -> plan_buffers (HBM
// placement) -> build_mix_work (the (VFO, time segment) items of the mix/decimate launches) -> build_tail_work (the
// block-per-tile launches of the leaf tail) -> build_level_plan (k_mix_levels' list) -> allocate_and_upload.
namespace {

struct Built { // host copies of what goes to the arena, and where
    ArenaPlan plan;
    std::map<std::vector<float>, size_t> tap_offsets; // identical tap sets are stored once
    size_t pay = 0;                                   // bytes of the packed payload buffer
    std::vector<std::vector<K1Work>> works;           // mix/decimate items per tree level
    std::vector<int> level_count, level_maxd;
    std::vector<K2aVfo> d2a;
    std::vector<K2Vfo> d2;
    std::vector<K3Vfo> d3;
    std::vector<K4Vfo> d4;
    std::vector<BlockWork> w2a, w2, w3, w4;
    std::vector<int> n2a, n2, n3, n4; // node index of each descriptor
    size_t o2a = 0, o2 = 0, o3 = 0, o4 = 0, ow2a = 0, ow2 = 0, ow3 = 0, ow4 = 0;
    std::vector<K1Work> all_items; // k_mix_levels: every level's items in one array ...
    std::vector<int> all_item_level, llist; // ... their levels, and the launch list over them
    size_t off_nco_jobs = 0;

    size_t place_taps(const std::vector<float> &t)
    {
        auto it = tap_offsets.find(t);
        if (it != tap_offsets.end())
            return it->second;
        const size_t o = plan.take(t.size() * sizeof(float));
        tap_offsets.emplace(t, o);
        return o;
    }
};

// Does this leaf run its /5 or /6 low-pass inside the mix wave (late_item)?  d = 0 below a parent (a tile-layout input),
// and the tap count the geometry was laid out for -- which is what vfo::init's design formula yields at every rate.
int fused_late_of(const sdrx_ctx *c, const Node &n)
{
    if (!c->opt_fuse_late || !n.leaf || !n.d.demod_usb || n.d.decimate_count != 0 || n.d.parent_id < 0)
        return 0;
    if (n.d.late_decimate == 5 && (int)n.dec.size() == LateGeom<5>::kTaps && n.d.samples_per_buffer >= LateGeom<5>::kChunkLen)
        return 5;
    if (n.d.late_decimate == 6 && (int)n.dec.size() == LateGeom<6>::kTaps && n.d.samples_per_buffer >= LateGeom<6>::kChunkLen)
        return 6;
    return 0;
}

// Does this leaf demodulate inside its mix wave (demod_chunk, kernels.hip)?  The reference's 48 kS/s sub VFO: two half-band
// stages below a parent (a tile-layout input, 256 stream samples per 1024-sample chunk), no late decimation, an audio low-pass of
// at most kDmMaxLpf taps (the 10 kHz filter at 48 kS/s has 47).  Everything else keeps k_usb_demod.
bool fused_demod_of(const sdrx_ctx *c, const Node &n)
{
    return c->opt_fuse_demod && n.leaf && n.d.demod_usb && n.d.late_decimate == 0 && n.d.decimate_count == 2 && n.d.parent_id >= 0 &&
           !n.long_lpf && (int)n.lpf.size() <= kDmMaxLpf && n.d.samples_per_buffer >= kChunk;
}
constexpr int kDemodStateFloats = 256; // K2Vfo::state: QO | QE | I | U at 64-float strides

// ---- per-node derived quantities: everything vfo::init computes (vfo.cpp:60-176)
int derive_nodes(sdrx_ctx *c)
{
    const int N = (int)c->nodes.size();
    c->root_frame = 0;
    int max_level = 0;
    for (int i = 0; i < N; ++i) {
        Node &n = c->nodes[(size_t)i];
        const sdrx_vfo_desc &d = n.d;
        n.leaf = n.children.empty();
        {
            char why[200];
            const int rc = sdrx_check_vfo(&d, why, sizeof why); // what a binding may already have asked at init() time
            if (rc != SDRX_OK)
                return fail(c, rc, "vfo %d: %s", i, why);
        }
        if (d.samples_per_buffer % kChunk != 0 && d.samples_per_buffer % kChunk < 256)
            return fail(c, SDRX_EUNSUPPORTED, "vfo %d: samples_per_buffer %d leaves a last chunk shorter than 256 samples", i,
                        d.samples_per_buffer);
        if ((long long)d.samples_per_buffer > (long long)d.fs)
            return fail(c, SDRX_EUNSUPPORTED, "vfo %d: a frame longer than one second of signal is not supported", i);
        n.n_f = d.samples_per_buffer >> d.decimate_count;
        int target = (int)(d.fs / std::pow(2, d.decimate_count)); // vfo.cpp:66
        n.n_out = n.n_f;
        const bool late = d.demod_usb && d.late_decimate > 0; // vfo.cpp:70
        if (late) {
            if (n.n_f % d.late_decimate)
                return fail(c, SDRX_EUNSUPPORTED, "vfo %d: %d samples per frame is not a multiple of late_decimate %d", i, n.n_f,
                            d.late_decimate);
            target /= d.late_decimate;
            n.n_out = n.n_f / d.late_decimate;
            if (!design_low_pass(2, (double)target * d.late_decimate, (double)(target / 2),
                                 (double)target / (d.late_decimate - 1), n.dec)) // vfo.cpp:82-87
                return fail(c, SDRX_EFILTER, "vfo %d: late-decimation low-pass rejected (firfilter.cpp:122-134)", i);
            if ((int)n.dec.size() > kMaxFir)
                return fail(c, SDRX_EUNSUPPORTED, "vfo %d: %zu-tap late-decimation filter exceeds %d", i, n.dec.size(), kMaxFir);
        }
        n.rate = (unsigned)target;
        if (d.demod_usb && d.filter_bw_hz > 0) { // vfo.cpp:106-124
            if (!design_low_pass(2, (double)target, (double)d.filter_bw_hz, (double)d.filter_bw_hz / 4, n.lpf))
                return fail(c, SDRX_EFILTER, "vfo %d: filter_bw %d Hz rejected at %d S/s (firfilter.cpp:122-134)", i, d.filter_bw_hz,
                            target);
            if ((int)n.lpf.size() > kMaxFirLong)
                return fail(c, SDRX_EUNSUPPORTED, "vfo %d: %zu-tap audio filter exceeds %d", i, n.lpf.size(), kMaxFirLong);
            n.long_lpf = (int)n.lpf.size() > kMaxFir;
        }
        if (d.demod_usb) {
            design_hilbert(kHilbert, n.n_out, n.hilbert); // vfo.cpp:137: "Fs" = samplesOut
            n.hnz.clear();
            for (int t = 0; t < kHilbert; ++t) {
                if (t & 1)
                    n.hnz.push_back(n.hilbert[(size_t)t]);
                else if (n.hilbert[(size_t)t] != 0.0f)
                    return fail(c, SDRX_EUNSUPPORTED, "vfo %d: even Hilbert tap %d is not zero", i, t);
            }
            n.hnz_e.assign(96, 0.0f);
            n.hnz_o.assign(96, 0.0f);
            std::copy(n.hnz.begin(), n.hnz.end(), n.hnz_e.begin() + 3);
            std::copy(n.hnz.begin(), n.hnz.end(), n.hnz_o.begin() + 2);
            if (!n.lpf.empty() && !n.long_lpf) {
                n.lpf_pad.assign(n.lpf.size() + 3 + 12, 0.0f);
                std::copy(n.lpf.begin(), n.lpf.end(), n.lpf_pad.begin() + 3);
            }
        }
        nco_rotation((double)d.fs, d.mixer_freq_hz, n.rot_re, n.rot_im);
        if (d.parent_id < 0) {
            n.level = 0;
            if (c->root_frame == 0)
                c->root_frame = d.samples_per_buffer;
            else if (c->root_frame != d.samples_per_buffer)
                return fail(c, SDRX_EINVAL, "vfo %d: all parent-less VFOs must share samples_per_buffer", i);
        } else {
            const Node &p = c->nodes[(size_t)d.parent_id];
            n.level = p.level + 1;
            if (d.samples_per_buffer != p.n_f)
                return fail(c, SDRX_EUNSUPPORTED, "vfo %d: samples_per_buffer %d != parent's output frame %d", i,
                            d.samples_per_buffer, p.n_f);
        }
        if (!n.leaf && d.demod_usb)
            return fail(c, SDRX_EINVAL, "vfo %d has children but demod_usb set", i);
        max_level = std::max(max_level, n.level);
    }
    c->n_levels = max_level + 1;
    for (Node &n : c->nodes) {
        n.fused_late = fused_late_of(c, n);
        n.fused_demod = fused_demod_of(c, n);
    }
    return SDRX_OK;
}

// ---- where everything lives in the arena; the payload buffer; SURVEY.md 8d's byte count
void plan_buffers(sdrx_ctx *c, Built &B)
{
    const int N = (int)c->nodes.size();
    ArenaPlan &plan = B.plan;
    c->off_k1vfo = plan.take(sizeof(K1Vfo) * (size_t)N);
    c->alg_bytes = 0;
    c->vfo_samples = 0;
    size_t tap_len = 0; // longest decimate[0] a fused leaf would have to keep (sdrx_set_tap)
    for (int i = 0; i < N; ++i) {
        Node &n = c->nodes[(size_t)i];
        const sdrx_vfo_desc &d = n.d;
        const bool late = d.demod_usb && d.late_decimate > 0;
        n.off_cp = plan.take(sizeof(float2) * (size_t)(d.fs / kRun + 1));
        // half-band history -- or, for a fused late decimation, the previous frame's last mixed samples
        const size_t hist = n.fused_late == 5 ? (size_t)late_hist<5>() : n.fused_late == 6 ? (size_t)late_hist<6>() : (size_t)std::max(1, d.decimate_count * kHbHist);
        for (int p = 0; p < 2; ++p)
            n.off_hb[p] = plan.take(sizeof(float2) * hist);
        n.H = n.Hx = 0;
        if (n.leaf && d.demod_usb) {
            const int Hdemod = (int)align_up((size_t)((n.long_lpf ? 0 : n.lpf.size()) + 1 + kHilbert - 1), 4);
            if (late) {
                n.Hx = n.fused_late ? 0 : (int)align_up(n.dec.size(), 4);
                n.H = Hdemod;
            } else {
                n.Hx = Hdemod; // the stream itself feeds the demodulator
            }
        }
        n.has_stream = !(n.fused_late || n.fused_demod) || c->opt_keep_streams;
        if (n.fused_demod)
            for (int p = 0; p < 2; ++p)
                n.off_dstate[p] = plan.take(sizeof(float) * kDemodStateFloats);
        if (n.has_stream)
            for (int p = 0; p < 2; ++p) // a stream that feeds children is kept in whole 1024-sample tiles
                n.off_stream[p] = plan.take(sizeof(float2) * (n.leaf ? (size_t)(n.Hx + n.n_f) : align_up((size_t)n.n_f, kChunk) + kChunk)); // (+1 tile: a shifted walk's idle lanes read past the last one)
        else
            tap_len = std::max(tap_len, (size_t)n.n_f);
        if (late)
            for (int p = 0; p < 2; ++p)
                n.off_z[p] = plan.take(sizeof(float2) * (size_t)(n.H + n.n_out));
        if (!n.lpf_pad.empty())
            n.off_lpf = B.place_taps(n.lpf_pad);
        if (n.long_lpf) {
            n.off_lpf = B.place_taps(n.lpf);
            n.Hu = (int)align_up(n.lpf.size(), 4);
            for (int p = 0; p < 2; ++p)
                n.off_u[p] = plan.take(sizeof(float) * (size_t)(n.Hu + n.n_out));
        }
        if (!n.hnz.empty()) {
            n.off_hnz = B.place_taps(n.hnz);
            n.off_hnz_e = B.place_taps(n.hnz_e);
            n.off_hnz_o = B.place_taps(n.hnz_o);
        }
        if (!n.dec.empty())
            n.off_dec = B.place_taps(n.dec);
        if (!n.hilbert.empty())
            n.off_hilbert = B.place_taps(n.hilbert);
        if (n.leaf) {
            n.pay_off = B.pay;
            if (d.demod_usb)
                n.pay_len = (uint32_t)(n.n_out * 2);
            else
                n.pay_len = (uint32_t)(d.cstyle == 1 ? n.n_f : 2 * n.n_f); // vfo.cpp:143-150
            B.pay = align_up(B.pay + n.pay_len, 64);
            if (c->opt_prequant && d.demod_usb)
                n.off_preq = plan.take(sizeof(float) * (size_t)n.n_out);
        }
        // SURVEY.md 8d algorithmic bytes: cf32 consumed + what this VFO hands on
        c->alg_bytes += 8ll * d.samples_per_buffer + (n.leaf ? (int64_t)n.pay_len : 8ll * n.n_f);
        c->vfo_samples += d.samples_per_buffer;
    }
    c->tap_len = tap_len;
    for (int p = 0; p < 2; ++p)
        c->off_tapbuf[p] = tap_len ? plan.take(sizeof(float2) * tap_len) : 0;
}

// ---- work lists for the mix/decimate launches, one launch per tree level
// A work item is one wave walking a run of chunks of one VFO-frame (+ a warm-up when it starts mid-frame).  Measured on
// config 3 (profiles/README.md): the same NUMBER of segments for every VFO of a level, 32 work items per CU in total, in
// VFO creation order (the long d=5 items of the first parent first, the short d=2 items of the second parent back-filling
// the tail) beats one resident round of equal-length items (86 vs 91.5 us), equal-length short items (94-99 us),
// class-interleaved order (103 us) and segment-major order (96-102 us).
int build_mix_work(sdrx_ctx *c, Built &B)
{
    const int N = (int)c->nodes.size();
    B.works.assign((size_t)c->n_levels, {});
    B.level_count.assign((size_t)c->n_levels, 0);
    B.level_maxd.assign((size_t)c->n_levels, 0);
    for (const Node &n : c->nodes) {
        B.level_count[(size_t)n.level]++;
        B.level_maxd[(size_t)n.level] = std::max(B.level_maxd[(size_t)n.level], n.d.decimate_count);
    }
    int ncu = 256;
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, c->device) == hipSuccess && prop.multiProcessorCount > 0)
            ncu = prop.multiProcessorCount;
    }
    std::vector<int> level_nseg((size_t)c->n_levels, 1);
    // (experiment switches, read here: SDRX_ITEMS_PER_CU = work items per CU a level is cut into, default 32; SDRX_MIN_SEG = the
    // fewest chunks of useful work a segment of a many-VFO level may have, default 4)
    const int items_per_cu = getenv("SDRX_ITEMS_PER_CU") ? std::max(1, atoi(getenv("SDRX_ITEMS_PER_CU"))) : 32;
    const int min_seg_chunks = getenv("SDRX_MIN_SEG") ? std::max(1, atoi(getenv("SDRX_MIN_SEG"))) : 4;
    for (int lv = 0; lv < c->n_levels; ++lv) // 32 work items per CU (= the hardware's wave slots per CU)
        level_nseg[(size_t)lv] = std::max(1, (ncu * items_per_cu + B.level_count[(size_t)lv] - 1) / B.level_count[(size_t)lv]);
    c->mix_chunks = 0;
    for (int i = 0; i < N; ++i) {
        const Node &n = c->nodes[(size_t)i];
        const int n_in = n.d.samples_per_buffer;
        // the walk's chunk and what a segment that starts inside the frame must walk before its first exact output:
        // the half-band cascade's dependency cone, or the decimating low-pass's length (a multiple of 16 L: a segment of a
        // fused late decimation starts on an output AND on a 16-sample run)
        const int chunk = n.fused_late == 5 ? LateGeom<5>::kChunkLen : n.fused_late == 6 ? LateGeom<6>::kChunkLen : kChunk;
        // (a leaf that demodulates in its wave: behind the half-band warm-up another 124 stream samples until the Hilbert window
        // holds real samples and N more until the audio low-pass does -- usb'[m] reads usb[m - N .. m - 1] --, 4 input samples each)
        const int warm_demod = n.fused_demod ? (int)align_up((size_t)(warmup_samples(n.d.decimate_count) + ((kHilbert - 1 + (int)n.lpf.size()) << n.d.decimate_count)), 16) : 0;
        const int warm = n.fused_late == 5 ? LateGeom<5>::kWarm : n.fused_late == 6 ? LateGeom<6>::kWarm : n.fused_demod ? warm_demod : warmup_samples(n.d.decimate_count);
        const int nchunks = (n_in + chunk - 1) / chunk;
        const int wch = (warm + chunk - 1) / chunk; // chunks a segment spends before its first exact output
        // few VFOs in the level (the 2-3 mains): segments as short as the warm-up allows;
        // otherwise at least 4 chunks of useful work per segment
        const bool few = (long long)B.level_count[(size_t)n.level] * nchunks < (long long)ncu * 16;
        // (a fused late decimation: measured on config 4, interleaved: 2 / 3 / 4 / 6 / 8 chunks per segment = 0.0481 / 0.0484 /
        // 0.0471 / 0.0482 / 0.0509 ms per step; SDRX_LATE_MINSEG for A/B runs)
        const int late_min_seg = getenv("SDRX_LATE_MINSEG") ? std::max(1, atoi(getenv("SDRX_LATE_MINSEG"))) : 4;
        const int min_seg = few ? std::max(1, wch) : n.fused_late ? late_min_seg : std::max(min_seg_chunks, min_seg_chunks * wch);
        int nseg = c->opt_segments > 0 ? c->opt_segments : std::min(level_nseg[(size_t)n.level], std::max(1, nchunks / min_seg));
        nseg = std::max(1, std::min(nseg, nchunks / std::max(1, wch)));
        // Segment s > 0 starts `warm` samples before its first emitted output and ends on a chunk
        // boundary of ITS OWN walk (s_begin + a whole number of chunks), so the warm-up costs the
        // first `warm / 16` lanes of its first chunk instead of a whole extra chunk; the boundaries
        // between segments are therefore not multiples of 1024.
        // ... unless the whole-chunk warm-up costs little anyway (long VFO-frames cut into few segments:
        // < 4 % extra chunks): then segments stay tile aligned, which keeps the kernel's uniform
        // walk on the tiles (measured on the memory-bound flat workload: a walk that straddles two
        // tiles per chunk costs 5 %).
        const bool shifted = n.fused_late || (long long)(nseg - 1) * wch * 25 > nchunks;
        const int lead = shifted ? warm : wch * chunk; // samples a segment walks before its first emitted output
        const long long target = ((long long)n_in + nseg - 1) / nseg; // samples a segment should emit
        int first_out = 0;                                            // input position of the first output the next segment emits
        while (first_out < n_in) {
            K1Work w;
            w.vfo = i;
            w.s_first_out = first_out;
            w.s_begin = first_out == 0 ? 0 : first_out - lead;
            if (w.s_begin < 0)
                return fail(c, SDRX_EUNSUPPORTED, "vfo %d: %d segments do not leave room for the %d-sample warm-up", i, nseg, warm);
            long long k = ((long long)(first_out - w.s_begin) + target + chunk / 2) / chunk; // chunks of this segment's walk
            k = std::max<long long>(k, lead / chunk + 1);                                   // it must emit something
            long long end = w.s_begin + k * chunk;
            if (end + lead + chunk / 2 >= n_in) // what would be left is not worth a segment of its own
                end = n_in;
            w.s_end = (int)std::min<long long>(n_in, end);
            if (w.s_end == n_in && w.s_begin > 0 && !n.fused_late) {
                // The chunk that holds the frame's last sample saves the filter history for the next
                // frame from the registers of its last TWO lanes and from the tail of the LDS stages:
                // like a tile-aligned frame (checked above), a shifted walk must end in a chunk of
                // at least 256 samples.  Start earlier if it does not -- more warm-up is always exact.
                // (A fused late decimation saves its history from LDS rows that hold the previous chunk's tail
                // as well, and its first chunk is longer than that history: nothing to adjust.)
                const int r = (n_in - w.s_begin) & (kChunk - 1);
                if (r != 0 && r < 256) {
                    const int unit = std::max(16, 1 << n.d.decimate_count);
                    const int delta = (256 - r + unit - 1) / unit * unit;
                    w.s_begin = std::max(0, w.s_begin - delta); // (0 = walk from the frame's start with the real history)
                }
            }
            B.works[(size_t)n.level].push_back(w);
            c->mix_chunks += (w.s_end - w.s_begin + chunk - 1) / chunk;
            first_out = w.s_end;
        }
    }
    // Order of the work items inside a launch (experiment switch SDRX_ORDER, default = VFO-major,
    // i.e. creation order; measured: spreading d=5 and d=2 items evenly through the list is 12 %
    // SLOWER than keeping each VFO's -- and each parent's -- items together).
    if (const char *e = getenv("SDRX_ORDER")) {
        if (atoi(e) == 1) // segment-major: all first segments, then all second segments, ...
            for (auto &wl : B.works)
                std::stable_sort(wl.begin(), wl.end(), [](const K1Work &a, const K1Work &b) { return a.s_first_out < b.s_first_out; });
    }
    c->l1.clear();
    for (int lv = 0; lv < c->n_levels; ++lv) {
        Launch1 L;
        L.kind = lv == 0 ? KIND_MIX_ROOT : KIND_MIX_SUB;
        L.level = lv;
        L.n_work = (int)B.works[(size_t)lv].size();
        bool need_tr = false;
        int lds_late = 0;
        L.alg_bytes = 0;
        for (const Node &n : c->nodes) {
            if (n.level != lv)
                continue;
            need_tr |= n.d.decimate_count == 0 && n.leaf && !n.fused_late;
            lds_late = std::max(lds_late, n.fused_late == 5 ? late_lds_bytes<5>() : n.fused_late == 6 ? late_lds_bytes<6>() : 0);
            lds_late = std::max(lds_late, n.fused_demod ? demod_lds_bytes() : 0);
            // SURVEY.md 8d share of this launch: cf32 consumed (+ cf32 handed to children; + the int16 payload of a leaf that
            // demodulates in its wave)
            L.alg_bytes += 8ll * n.d.samples_per_buffer + (n.leaf ? (n.fused_demod ? (int64_t)n.pay_len : 0ll) : 8ll * n.n_f);
        }
        L.lds_bytes = std::max(k1_lds_bytes(B.level_maxd[(size_t)lv], need_tr), lds_late);
        L.off_work = B.plan.take(sizeof(K1Work) * B.works[(size_t)lv].size());
        c->l1.push_back(L);
    }
    return SDRX_OK;
}

// ---- block-per-tile launches of the leaf tail: one launch per kernel, driven by a (vfo, tile) work list
void build_tail_work(sdrx_ctx *c, Built &B)
{
    const int N = (int)c->nodes.size();
    int64_t b2 = 0, b3 = 0;
    int lds2a = 0;
    auto two_kernel_late = [](const Node &n) { return n.leaf && n.d.demod_usb && n.d.late_decimate > 0 && !n.fused_late; };
    // every late-decimating VFO left to a kernel of its own has L in {5,6} and <= 96 taps: one-wave tiles, R outputs per lane
    bool late4 = !getenv("SDRX_NO_LATE4");
    int late_lmax = 5, late_ndec = 0;
    for (const Node &n : c->nodes)
        if (two_kernel_late(n)) {
            late4 = late4 && (n.d.late_decimate == 5 || n.d.late_decimate == 6) && (int)n.dec.size() <= kLateMaxTaps;
            late_lmax = std::max(late_lmax, n.d.late_decimate);
            late_ndec = std::max(late_ndec, (int)n.dec.size());
        }
    c->late4 = late4;
    // 2 outputs per lane (8 KB of LDS per wave, ~20 waves per CU) measured 38 us on config 4, 4 outputs
    // per lane (fewer LDS reads, 16 KB, 10 waves per CU) 48 us, the one-output-per-thread kernel 54 us
    c->late4_r = getenv("SDRX_LATE4_R") ? atoi(getenv("SDRX_LATE4_R")) : 2;
    if (c->late4_r != 4)
        c->late4_r = 2;
    const int late_tile = late4 ? 64 * c->late4_r : 256;
    for (int i = 0; i < N; ++i) {
        Node &n = c->nodes[(size_t)i];
        if (!n.leaf)
            continue;
        if (n.d.demod_usb) {
            if (two_kernel_late(n)) {
                for (int b = 0; b < (n.n_out + late_tile - 1) / late_tile; ++b)
                    B.w2a.push_back({(int)B.d2a.size(), b});
                B.n2a.push_back(i);
                B.d2a.push_back(K2aVfo{});
                lds2a = std::max(lds2a, (int)sizeof(float2) * (n.d.late_decimate * 255 + (int)n.dec.size()));
            }
            {
                // a block computes E = nlpf (rounded up to even) extra usb values as history for its low-pass:
                // its tile is shortened by E so that usb stays ONE pass of <= 1024 values (a second pass would
                // keep two of the four waves busy for a whole Hilbert loop on ~50 values)
                const int nl = n.long_lpf ? 0 : (int)n.lpf.size();
                n.demod_tile = (nl > 0 && !getenv("SDRX_DEMOD_FULL_TILE")) ? ((kDemodTile - (nl + (nl & 1))) & ~3) : kDemodTile;
            }
            // (a leaf that demodulates in its mix wave has a descriptor -- the wave reads it -- but no blocks in this launch)
            for (int b = 0; !n.fused_demod && b < (n.n_out + n.demod_tile - 1) / n.demod_tile; ++b)
                B.w2.push_back({(int)B.d2.size(), b});
            n.d2_index = (int)B.d2.size();
            B.n2.push_back(i);
            B.d2.push_back(K2Vfo{});
            if (!n.fused_demod)
                b2 += n.pay_len; // W_out of SURVEY.md 8d
            if (n.long_lpf) {
                for (int b = 0; b < (n.n_out + 255) / 256; ++b)
                    B.w4.push_back({(int)B.d4.size(), b});
                B.n4.push_back(i);
                B.d4.push_back(K4Vfo{});
            }
        } else {
            for (int b = 0; b < (n.n_f + 4095) / 4096; ++b)
                B.w3.push_back({(int)B.d3.size(), b});
            B.n3.push_back(i);
            B.d3.push_back(K3Vfo{});
            b3 += n.pay_len;
        }
    }
    c->lb.clear();
    ArenaPlan &plan = B.plan;
    if (!B.d2a.empty()) {
        B.o2a = plan.take(sizeof(K2aVfo) * B.d2a.size());
        B.ow2a = plan.take(sizeof(BlockWork) * B.w2a.size());
        c->lb.push_back({KIND_LATE_DEC, (int)B.w2a.size(), B.o2a, B.ow2a, c->late4 ? late4_lds_bytes(c->late4_r, late_lmax, late_ndec) : lds2a, 0});
    }
    if (!B.d2.empty() && !getenv("SDRX_NO_LPT")) {
        // Blocks are independent and the launch is a few resident rounds deep, so its tail is set by
        // what is dispatched last: longest blocks first (a block with the audio low-pass does about
        // twice the work; the last block of a VFO-frame may be nearly empty).
        auto cost = [&](const BlockWork &b) -> long long {
            const Node &n = c->nodes[(size_t)B.n2[(size_t)b.vfo]];
            const int outs = std::min(n.demod_tile, n.n_out - b.blk * n.demod_tile);
            return (long long)outs * (kHilbertNz + (long long)(n.long_lpf ? 0 : n.lpf.size()));
        };
        std::stable_sort(B.w2.begin(), B.w2.end(), [&](const BlockWork &a, const BlockWork &b) { return cost(a) > cost(b); });
    }
    if (!B.d2.empty()) {
        B.o2 = plan.take(sizeof(K2Vfo) * B.d2.size());
        B.ow2 = plan.take(sizeof(BlockWork) * std::max<size_t>(1, B.w2.size()));
        if (!B.w2.empty()) // (every USB leaf may demodulate in its own mix wave: no k_usb_demod launch at all then)
            c->lb.push_back({KIND_DEMOD, (int)B.w2.size(), B.o2, B.ow2, 0, b2});
    }
    if (!B.d4.empty()) {
        int lds4 = 0;
        for (int i : B.n4)
            lds4 = std::max(lds4, (int)sizeof(float) * ((int)c->nodes[(size_t)i].lpf.size() + 256));
        B.o4 = plan.take(sizeof(K4Vfo) * B.d4.size());
        B.ow4 = plan.take(sizeof(BlockWork) * B.w4.size());
        c->lb.push_back({KIND_LPF_LONG, (int)B.w4.size(), B.o4, B.ow4, lds4, 0});
    }
    if (!B.d3.empty()) {
        B.o3 = plan.take(sizeof(K3Vfo) * B.d3.size());
        B.ow3 = plan.take(sizeof(BlockWork) * B.w3.size());
        c->lb.push_back({KIND_COMPRESS, (int)B.w3.size(), B.o3, B.ow3, 0, b3});
    }
}

// ---- the one-launch levels (k_mix_levels): unified item array and list
void build_level_plan(sdrx_ctx *c, Built &B)
{
    LevelPlan &P = c->fp;
    P = LevelPlan();
    P.usable = c->n_levels >= 2 && c->n_levels <= kMaxLevels; // (one level: nothing to share a launch with)
    // Frame k passes level l in launch k + l and gets its leaf tail behind launch k + n_levels - 1; the streams are double
    // buffered by frame parity.  A leaf at level l is written in launch k + l and overwritten by frame k + 2 in launch
    // k + l + 2: its tail must have run by then, i.e. l >= n_levels - 2 -- true for every tree the reference builds (two
    // levels).  A deeper tree with a shallower leaf runs one launch per level instead (found by the 600-seed soak run of
    // test_frame_pipeline_on_random_trees: seed 213, a parent-less leaf beside a three-level tree).
    for (const Node &n : c->nodes)
        if (n.leaf && n.level < c->n_levels - 2)
            P.usable = false;
    if (!P.usable)
        return;
    // deepest level first: in the steady state of the reference's two-level trees the long sub-VFO
    // items are dispatched first and the short level-0 items fill the launch's tail
    // (SDRX_LEVEL_ORDER=1: level 0 first, for A/B runs)
    const bool root_first = getenv("SDRX_LEVEL_ORDER") && atoi(getenv("SDRX_LEVEL_ORDER")) == 1;
    P.part_begin.assign((size_t)c->n_levels, 0);
    P.part_end.assign((size_t)c->n_levels, 0);
    P.part_bytes.assign((size_t)c->n_levels, 0);
    for (int q = 0; q < c->n_levels; ++q) {
        const int lv = root_first ? q : c->n_levels - 1 - q;
        while (B.llist.size() % 8)
            B.llist.push_back(-1);
        P.part_begin[(size_t)lv] = (int)B.llist.size();
        const int base = (int)B.all_items.size(), cnt = (int)B.works[(size_t)lv].size();
        B.all_items.insert(B.all_items.end(), B.works[(size_t)lv].begin(), B.works[(size_t)lv].end());
        B.all_item_level.insert(B.all_item_level.end(), (size_t)cnt, lv);
        for (int i = 0; i < cnt; ++i)
            B.llist.push_back(base + i);
        P.part_end[(size_t)lv] = (int)B.llist.size();
        P.part_bytes[(size_t)lv] = c->l1[(size_t)lv].alg_bytes;
        P.lds_bytes = std::max(P.lds_bytes, c->l1[(size_t)lv].lds_bytes);
    }
    P.off_items = B.plan.take(sizeof(K1Work) * B.all_items.size());
    P.off_item_level = B.plan.take(sizeof(int) * B.all_item_level.size());
    P.off_list = B.plan.take(sizeof(int) * B.llist.size());
}

// ---- allocate, zero (= the reference's zero-initialised filter state, dsp.cpp:40-49), fill the descriptors, build the NCO tables
int allocate_and_upload(sdrx_ctx *c, Built &B)
{
    const int N = (int)c->nodes.size();
    B.off_nco_jobs = B.plan.take(sizeof(NcoInit) * (size_t)N);
    c->arena_bytes = align_up(B.plan.size, 256);
    HIPCHK(c, hipMalloc(&c->arena, c->arena_bytes));
    HIPCHK(c, hipMemsetAsync(c->arena, 0, c->arena_bytes, c->stream));
    c->pay_bytes = std::max<size_t>(B.pay, 64);
    for (int p = 0; p < 2; ++p) {
        HIPCHK(c, hipMalloc(&c->d_pay[p], align_up(c->pay_bytes, 16))); // (whole 16-byte units: k_copy16)
        HIPCHK(c, hipMemsetAsync(c->d_pay[p], 0, c->pay_bytes, c->stream));
        HIPCHK(c, hipHostMalloc(&c->h_pay[p], align_up(c->pay_bytes, 16), hipHostMallocDefault));
        memset(c->h_pay[p], 0, c->pay_bytes);
    }
    {
        const size_t raw_tiles = align_up((size_t)c->root_frame, kChunk) + kChunk; // (+1 tile, as for the parents' streams)
        HIPCHK(c, hipMalloc(&c->d_raw_tiled, raw_tiles * sizeof(float2)));
        HIPCHK(c, hipMemsetAsync(c->d_raw_tiled, 0, raw_tiles * sizeof(float2), c->stream));
        c->root_direct = B.level_count[0] <= 4 && !getenv("SDRX_NO_ROOT_DIRECT"); // the reference allows 3 mains (mainwindow.h:82)
    }
    auto P = [&](size_t off) { return c->arena + off; };
    std::vector<K1Vfo> k1((size_t)N);
    std::vector<NcoInit> jobs((size_t)N);
    for (int i = 0; i < N; ++i) {
        Node &n = c->nodes[(size_t)i];
        K1Vfo &k = k1[(size_t)i];
        memset(&k, 0, sizeof k);
        for (int p = 0; p < 2; ++p) {
            if (n.d.parent_id >= 0) {
                const Node &pn = c->nodes[(size_t)n.d.parent_id];
                k.in[p] = reinterpret_cast<const float2 *>(P(pn.off_stream[p])) + pn.Hx;
            } else {
                k.in[p] = c->d_raw_tiled;
            }
            if (n.fused_late) { // the wave writes the decimated stream itself; decimate[0] only where it is kept
                k.out[p] = reinterpret_cast<float2 *>(P(n.off_z[p])) + n.H;
                k.tap[p] = n.has_stream ? reinterpret_cast<float2 *>(P(n.off_stream[p])) : nullptr;
            } else if (n.fused_demod) { // the wave writes the int16 payload itself; decimate[d] only where it is kept
                k.out[p] = nullptr;
                k.tap[p] = n.has_stream ? reinterpret_cast<float2 *>(P(n.off_stream[p])) + n.Hx : nullptr;
            } else {
                k.out[p] = reinterpret_cast<float2 *>(P(n.off_stream[p])) + n.Hx;
            }
            k.hb[p] = reinterpret_cast<float2 *>(P(n.off_hb[p]));
        }
        k.cp = reinterpret_cast<const float2 *>(P(n.off_cp));
        k.rot_re = n.rot_re;
        k.rot_im = n.rot_im;
        {
            // the tolerance arithmetic's NCO: 1 .. 4 steps of the recurrence as ONE rotation (the stabiliser holds |v|, so a
            // step is the rotation by arg(rot) at unit modulus: oscillator.cpp:20-28), in double, stored as floats
            const double ang = std::atan2((double)n.rot_im, (double)n.rot_re);
            for (int t = 0; t < 4; ++t)
                k.rk[t] = make_float2((float)std::cos(ang * (t + 1)), (float)std::sin(ang * (t + 1)));
        }
        k.n_in = n.d.samples_per_buffer;
        k.d = n.d.decimate_count;
        k.L = n.d.fs;
        k.out_tiled = n.leaf ? 0 : 1;
        k.late_L = n.fused_late;
        k.late_taps = n.fused_late ? reinterpret_cast<const float *>(P(n.off_dec)) : nullptr;
        k.dm = n.fused_demod ? reinterpret_cast<const K2Vfo *>(P(B.o2)) + n.d2_index : nullptr;
        jobs[(size_t)i] = NcoInit{reinterpret_cast<float2 *>(P(n.off_cp)), n.rot_re, n.rot_im, n.d.fs, 0};
    }
    for (size_t q = 0; q < B.d2a.size(); ++q) {
        Node &n = c->nodes[(size_t)B.n2a[q]];
        K2aVfo &k = B.d2a[q];
        for (int p = 0; p < 2; ++p) {
            k.x[p] = reinterpret_cast<const float2 *>(P(n.off_stream[p]));
            k.x_next[p] = reinterpret_cast<float2 *>(P(n.off_stream[p ^ 1]));
            k.z[p] = reinterpret_cast<float2 *>(P(n.off_z[p])) + n.H;
        }
        k.taps = reinterpret_cast<const float *>(P(n.off_dec));
        k.Hx = n.Hx;
        k.n = n.n_f;
        k.ndec = (int)n.dec.size();
        k.L = n.d.late_decimate;
        k.n_out = n.n_out;
    }
    for (size_t q = 0; q < B.d2.size(); ++q) {
        Node &n = c->nodes[(size_t)B.n2[q]];
        K2Vfo &k = B.d2[q];
        const bool late = n.d.late_decimate > 0;
        for (int p = 0; p < 2; ++p) {
            k.s[p] = n.fused_demod ? nullptr : reinterpret_cast<const float2 *>(P(late ? n.off_z[p] : n.off_stream[p]));
            k.s_next[p] = n.fused_demod ? nullptr : reinterpret_cast<float2 *>(P(late ? n.off_z[p ^ 1] : n.off_stream[p ^ 1]));
        }
        k.hnz = reinterpret_cast<const float *>(P(n.off_hnz));
        k.hnz_e = reinterpret_cast<const float *>(P(n.off_hnz_e));
        k.hnz_o = reinterpret_cast<const float *>(P(n.off_hnz_o));
        k.lpf_pad = (n.lpf.empty() || n.long_lpf) ? nullptr : reinterpret_cast<const float *>(P(n.off_lpf));
        for (int p = 0; p < 2; ++p) {
            k.usb_out[p] = n.long_lpf ? reinterpret_cast<float *>(P(n.off_u[p])) + n.Hu : nullptr;
            k.state[p] = n.fused_demod ? reinterpret_cast<float *>(P(n.off_dstate[p])) : nullptr;
        }
        for (int p = 0; p < 2; ++p)
            k.pay[p] = reinterpret_cast<short *>(c->d_pay[p] + n.pay_off);
        k.prequant = (c->opt_prequant) ? reinterpret_cast<float *>(P(n.off_preq)) : nullptr;
        k.gain = n.d.gain;
        k.H = late ? n.H : n.Hx;
        k.n = n.n_out;
        k.nlpf = n.long_lpf ? 0 : (int)n.lpf.size();
        k.tile = n.demod_tile;
    }
    for (size_t q = 0; q < B.d4.size(); ++q) {
        Node &n = c->nodes[(size_t)B.n4[q]];
        K4Vfo &k = B.d4[q];
        for (int p = 0; p < 2; ++p) {
            k.u[p] = reinterpret_cast<const float *>(P(n.off_u[p]));
            k.u_next[p] = reinterpret_cast<float *>(P(n.off_u[p ^ 1]));
            k.pay[p] = reinterpret_cast<short *>(c->d_pay[p] + n.pay_off);
        }
        k.taps = reinterpret_cast<const float *>(P(n.off_lpf));
        k.prequant = (c->opt_prequant) ? reinterpret_cast<float *>(P(n.off_preq)) : nullptr;
        k.gain = n.d.gain;
        k.Hu = n.Hu;
        k.n = n.n_out;
        k.nlpf = (int)n.lpf.size();
    }
    for (size_t q = 0; q < B.d3.size(); ++q) {
        Node &n = c->nodes[(size_t)B.n3[q]];
        K3Vfo &k = B.d3[q];
        for (int p = 0; p < 2; ++p) {
            k.s[p] = reinterpret_cast<const float2 *>(P(n.off_stream[p])) + n.Hx;
            k.pay[p] = reinterpret_cast<signed char *>(c->d_pay[p] + n.pay_off);
        }
        k.n = n.n_f;
        k.cstyle = n.d.cstyle;
        k.scalecomp = n.d.scalecomp;
    }
    auto up = [&](size_t off, const void *src, size_t bytes) -> hipError_t {
        return bytes ? hipMemcpyAsync(P(off), src, bytes, hipMemcpyHostToDevice, c->stream) : hipSuccess;
    };
    HIPCHK(c, up(c->off_k1vfo, k1.data(), sizeof(K1Vfo) * k1.size()));
    HIPCHK(c, up(B.off_nco_jobs, jobs.data(), sizeof(NcoInit) * jobs.size()));
    for (int lv = 0; lv < c->n_levels; ++lv)
        HIPCHK(c, up(c->l1[(size_t)lv].off_work, B.works[(size_t)lv].data(), sizeof(K1Work) * B.works[(size_t)lv].size()));
    HIPCHK(c, up(B.o2a, B.d2a.data(), sizeof(K2aVfo) * B.d2a.size()));
    HIPCHK(c, up(B.ow2a, B.w2a.data(), sizeof(BlockWork) * B.w2a.size()));
    HIPCHK(c, up(B.o2, B.d2.data(), sizeof(K2Vfo) * B.d2.size()));
    HIPCHK(c, up(B.ow2, B.w2.data(), sizeof(BlockWork) * B.w2.size()));
    HIPCHK(c, up(B.o4, B.d4.data(), sizeof(K4Vfo) * B.d4.size()));
    HIPCHK(c, up(B.ow4, B.w4.data(), sizeof(BlockWork) * B.w4.size()));
    HIPCHK(c, up(B.o3, B.d3.data(), sizeof(K3Vfo) * B.d3.size()));
    HIPCHK(c, up(B.ow3, B.w3.data(), sizeof(BlockWork) * B.w3.size()));
    if (c->fp.usable) {
        HIPCHK(c, up(c->fp.off_items, B.all_items.data(), sizeof(K1Work) * B.all_items.size()));
        HIPCHK(c, up(c->fp.off_item_level, B.all_item_level.data(), sizeof(int) * B.all_item_level.size()));
        HIPCHK(c, up(c->fp.off_list, B.llist.data(), sizeof(int) * B.llist.size()));
    }
    for (auto &kv : B.tap_offsets)
        HIPCHK(c, up(kv.second, kv.first.data(), kv.first.size() * sizeof(float)));
    HIPCHK(c, hipStreamSynchronize(c->stream)); // the host vectors above go out of scope

    // NCO tables: Oscillator::Oscillator for every node, on the device
    hipLaunchKernelGGL(k_nco_init, dim3((N + 63) / 64), dim3(64), 0, c->stream, reinterpret_cast<const NcoInit *>(P(B.off_nco_jobs)), N);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SDRX_OK;
}

// ---- publish order: main VFOs in list order, their subs in list order (vfo.cpp:257-263)
void build_publish_order(sdrx_ctx *c)
{
    c->publish_order.clear();
    std::vector<int> stack;
    for (int i = (int)c->nodes.size() - 1; i >= 0; --i)
        if (c->nodes[(size_t)i].d.parent_id < 0)
            stack.push_back(i);
    while (!stack.empty()) {
        const int i = stack.back();
        stack.pop_back();
        const Node &n = c->nodes[(size_t)i];
        if (n.leaf)
            c->publish_order.push_back(i);
        else
            for (auto it = n.children.rbegin(); it != n.children.rend(); ++it)
                stack.push_back(*it);
    }
}

int finalize_impl(sdrx_ctx *c)
{
    if (int rc = derive_nodes(c))
        return rc;
    Built B;
    plan_buffers(c, B);
    if (int rc = build_mix_work(c, B))
        return rc;
    build_tail_work(c, B);
    build_level_plan(c, B);
    if (int rc = allocate_and_upload(c, B))
        return rc;
    build_publish_order(c);
    c->taps.clear();
    c->finalized = true;
    return SDRX_OK;
}

} // namespace

extern "C" {

int sdrx_finalize(sdrx_ctx *c)
{
    if (!c)
        return SDRX_EINVAL;
    if (c->finalized)
        return fail(c, SDRX_ESTATE, "sdrx_finalize called twice");
    if (c->nodes.empty())
        return fail(c, SDRX_ESTATE, "sdrx_finalize: no VFOs");
    HIPCHK(c, hipSetDevice(c->device));
    const int rc = finalize_impl(c);
    if (rc != SDRX_OK) { // nothing of a half-built tree stays behind: a later call starts clean
        (void)hipStreamSynchronize(c->stream);
        free_device_state(c);
        c->l1.clear();
        c->lb.clear();
        c->publish_order.clear();
    }
    return rc;
}

// The reference's fftVFOSlot(topic) (vfo.cpp:492-509): from the next frame on, decimate[decimateCount] of node `id` is what
// sdrx_get_stream serves.  Every VFO keeps that stream in HBM anyway -- except a leaf whose late decimation runs inside the
// mix wave (it writes only the decimated stream): for such a leaf this call makes the wave keep decimate[0] as well.
// sdrx_set_tap REPLACES the selection (id = -1: none), sdrx_add_tap adds to it: fftVFOSlot sets emitFFT on every VFO whose
// topic equals the selected string, so two VFOs with one topic are two taps.
static int tap_change(sdrx_ctx *c, int id, bool replace, const char *what)
{
    if (!c)
        return SDRX_EINVAL;
    if (!c->finalized)
        return fail(c, SDRX_ESTATE, "%s before sdrx_finalize", what);
    if (id < (replace ? -1 : 0) || id >= (int)c->nodes.size())
        return fail(c, SDRX_EINVAL, "bad vfo id %d", id);
    if (c->in_flight > 0)
        return fail(c, SDRX_ESTATE, "%s: %d submitted frame(s) not yet delivered -- call sdrx_wait first", what, c->in_flight);
    HIPCHK(c, hipSetDevice(c->device));
    if (int rc = drain(c)) // frames inside the software pipeline run to their end with the taps they were queued under
        return rc;
    auto point = [&](int node, float2 *b0, float2 *b1) -> hipError_t { // K1Vfo::tap of `node` (synchronous: the pointers live on this stack)
        float2 *ptrs[2] = {b0, b1};
        return hipMemcpy(c->arena + c->off_k1vfo + sizeof(K1Vfo) * (size_t)node + offsetof(K1Vfo, tap), ptrs, sizeof ptrs, hipMemcpyHostToDevice);
    };
    if (replace) {
        for (auto it = c->taps.begin(); it != c->taps.end();) {
            if (it->first == id) { // stays what it is (and keeps serving the frames it has)
                ++it;
                continue;
            }
            HIPCHK(c, point(it->first, nullptr, nullptr));
            if (it->second.own)
                for (int p = 0; p < 2; ++p)
                    (void)hipFree(it->second.buf[p]);
            it = c->taps.erase(it);
        }
    }
    if (id < 0 || c->nodes[(size_t)id].has_stream || c->taps.count(id))
        return SDRX_OK; // every other node keeps decimate[d] in HBM anyway
    sdrx_ctx::TapBuf t;
    t.since = c->frame_no;
    bool arena_free = c->tap_len > 0;
    for (const auto &kv : c->taps)
        arena_free = arena_free && kv.second.own;
    if (arena_free) {
        for (int p = 0; p < 2; ++p)
            t.buf[p] = reinterpret_cast<float2 *>(c->arena + c->off_tapbuf[p]);
    } else {
        t.own = true;
        for (int p = 0; p < 2; ++p)
            if (hipMalloc(&t.buf[p], sizeof(float2) * (size_t)c->nodes[(size_t)id].n_f) != hipSuccess) {
                if (t.buf[0])
                    (void)hipFree(t.buf[0]);
                return fail(c, SDRX_ENOMEM, "%s: no device memory for another tap buffer", what);
            }
    }
    HIPCHK(c, point(id, t.buf[0], t.buf[1]));
    c->taps.emplace(id, t);
    return SDRX_OK;
}

int sdrx_set_tap(sdrx_ctx *c, int id) { return tap_change(c, id, true, "sdrx_set_tap"); }
int sdrx_add_tap(sdrx_ctx *c, int id) { return tap_change(c, id, false, "sdrx_add_tap"); }

int sdrx_set_stream(sdrx_ctx *c, void *s)
{
    if (!c)
        return SDRX_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    int rc = drain(c);
    if (rc)
        return rc;
    c->stream = s ? reinterpret_cast<hipStream_t>(s) : c->own_stream;
    return SDRX_OK;
}

} // extern "C"

namespace {

int check_frame_call(sdrx_ctx *c, const char *what, const void *ptr, int n_complex, bool sync_call)
{
    if (c && c->broken)
        return fail(c, SDRX_EHIP, "%s: injected fault (SDRX_FAULT_WAIT): the context is unusable", what);
    if (!c)
        return SDRX_EINVAL;
    if (!ptr)
        return fail(c, SDRX_EINVAL, "%s: null frame pointer", what);
    if (!c->finalized)
        return fail(c, SDRX_ESTATE, "%s before sdrx_finalize", what);
    if (n_complex != c->root_frame)
        return fail(c, SDRX_EINVAL, "frame of %d samples, VFOs were initialised for %d (vfo::init samplesPerBuffer)", n_complex,
                    c->root_frame);
    if (sync_call && c->in_flight > 0)
        return fail(c, SDRX_ESTATE, "%s: %d submitted frame(s) not yet delivered -- call sdrx_wait first", what, c->in_flight);
    if (!sync_call && c->in_flight >= SDRX_MAX_IN_FLIGHT)
        return fail(c, SDRX_ESTATE, "%s: %d frames in flight -- call sdrx_wait before submitting another", what, c->in_flight);
    HIPCHK(c, hipSetDevice(c->device));
    return SDRX_OK;
}

// host frame -> the library's pinned staging buffer of this frame parity -> device.  The pinned buffer's
// previous user is frame f-2, which has been delivered (at most two frames are in flight), so its copy is
// long done; the device buffer is written and read on `stream` only, in order.
int stage_host_frame(sdrx_ctx *c, const void *src, size_t bytes, void *dst_dev)
{
    const int p = (int)(c->frame_no & 1ull);
    if (c->h_in_bytes < (size_t)c->root_frame * sizeof(float2)) {
        for (int q = 0; q < 2; ++q) {
            if (c->h_in[q])
                (void)hipHostFree(c->h_in[q]);
            c->h_in[q] = nullptr;
            HIPCHK(c, hipHostMalloc(&c->h_in[q], (size_t)c->root_frame * sizeof(float2), hipHostMallocDefault));
        }
        c->h_in_bytes = (size_t)c->root_frame * sizeof(float2);
    }
    memcpy(c->h_in[p], src, bytes);
    for (auto &r : c->shared_readers[p]) // whoever shared frame f-2 of this buffer has read it before it is overwritten
        if (r.pending) {
            HIPCHK(c, hipStreamWaitEvent(c->stream, r.ev, 0));
            r.pending = false;
        }
    // (the runtime moves host-to-device copies with the DMA engine: concurrent with kernels.  SDRX_UPLOAD_KERNEL=1: a copy kernel
    // reading the pinned buffer over PCIe instead -- measured slower, 0.353 vs 0.302 ms per pipelined frame on config 3: it sits
    // in the compute stream's way)
    if (!c->upload_kernel) {
        HIPCHK(c, hipMemcpyAsync(dst_dev, c->h_in[p], bytes, hipMemcpyHostToDevice, c->stream));
    } else {
        const size_t n16 = (bytes + 15) / 16; // (both buffers are whole 16-byte units long: frames are multiples of 16 samples)
        hipLaunchKernelGGL(k_copy16, dim3(64), dim3(256), 0, c->stream, reinterpret_cast<const uint4 *>(c->h_in[p]),
                           reinterpret_cast<uint4 *>(dst_dev), n16);
    }
    HIPCHK(c, hipEventRecord(c->ev_staged[p], c->stream)); // (for a context that shares this frame: sdrx_submit_shared)
    return SDRX_OK;
}

int enqueue_f32(sdrx_ctx *c, const float *iq, int n_complex, bool egress)
{
    int rc = ensure_raw(c, (size_t)c->root_frame);
    if (rc)
        return rc;
    float2 *dst = c->d_raw[c->frame_no & 1ull];
    rc = stage_host_frame(c, iq, (size_t)n_complex * sizeof(float2), dst);
    if (rc)
        return rc;
    rc = enqueue_frame(c, dst, kRawF32, egress);
    if (rc == SDRX_OK)
        c->last_raw = kRawF32;
    return rc;
}

// `dev_bytes`: the frame's dongle bytes, already on this context's device and complete in the order of
// c->stream.  LUT (+ the DC-bias IIR with this context's own accumulator) and the frame itself.
int enqueue_u8_device(sdrx_ctx *c, const void *dev_bytes, int n_complex, int correct_dc, bool egress)
{
    if (correct_dc && !c->d_dc_state) {
        HIPCHK(c, hipMalloc(&c->d_dc_state, 4 * sizeof(float)));
        HIPCHK(c, hipMemsetAsync(c->d_dc_state, 0, 4 * sizeof(float), c->stream)); // `static cpx_typef avept=0`, sdrj.cpp:279
    }
    int mode = kRawU8;
    const int nchunks = (n_complex + kChunk - 1) / kChunk;
    if (correct_dc && c->opt_dc_blocked && !c->d_dc_tab) {
        // powers of the decay A = (float)(1 - 1e-6): [0..16] A^k, [32..95] A^(16 l), [96 + k] A^(1024 k)
        const double A = (double)(1.0f - 0.000001f);
        std::vector<double> tab(96 + (size_t)nchunks + 1 + 2 * (size_t)nchunks + 2, 0.0);
        for (int k = 0; k <= 16; ++k)
            tab[(size_t)k] = std::pow(A, k);
        for (int l = 0; l < 64; ++l)
            tab[32 + (size_t)l] = std::pow(A, 16.0 * l);
        for (int k = 0; k <= nchunks; ++k)
            tab[96 + (size_t)k] = std::pow(A, 1024.0 * k);
        c->dc_tab_sums = (96 + (size_t)nchunks + 1 + 1) & ~(size_t)1; // 16-byte aligned double2[]
        HIPCHK(c, hipMalloc(&c->d_dc_tab, tab.size() * sizeof(double)));
        HIPCHK(c, hipMemcpy(c->d_dc_tab, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    if (correct_dc && !c->opt_dc_blocked) {
        // products (parallel) -> the two recurrences (one wave each, nearly alone with their dependent chain) -> subtract (parallel)
        const int stride = (int)align_up((size_t)n_complex + kDcPad, 64);
        if (c->dc_work_stride < stride) {
            if (c->d_dc_work)
                (void)hipFree(c->d_dc_work);
            c->d_dc_work = nullptr;
            HIPCHK(c, hipMalloc(&c->d_dc_work, sizeof(float) * 4 * (size_t)stride)); // P[2][stride] | A[2][stride]
            HIPCHK(c, hipMemsetAsync(c->d_dc_work, 0, sizeof(float) * 4 * (size_t)stride, c->stream));
            c->dc_work_stride = stride;
        }
        float *Pp = c->d_dc_work, *Ap = c->d_dc_work + 2 * (size_t)c->dc_work_stride;
        const int words = n_complex / 2;
        Bracket b(c, c->stream, KIND_INGEST, 0);
        hipLaunchKernelGGL(k_dc_products, dim3((words + 255) / 256), dim3(256), 0, c->stream, reinterpret_cast<const unsigned *>(dev_bytes), Pp,
                           n_complex, c->dc_work_stride);
        if (c->opt_dc_speculative) {
            if (!c->d_dc_counters) {
                HIPCHK(c, hipMalloc(&c->d_dc_counters, 4 * sizeof(unsigned long long)));
                HIPCHK(c, hipMemsetAsync(c->d_dc_counters, 0, 4 * sizeof(unsigned long long), c->stream));
            }
            // one workgroup per component, dc_waves consecutive 1024-sample blocks per step
            auto chain = c->dc_waves >= 8 ? k_dc_chain_spec<8> : c->dc_waves >= 4 ? k_dc_chain_spec<4> : c->dc_waves >= 2 ? k_dc_chain_spec<2> : k_dc_chain_spec<1>;
            const int waves = c->dc_waves >= 8 ? 8 : c->dc_waves >= 4 ? 4 : c->dc_waves >= 2 ? 2 : 1;
            hipLaunchKernelGGL(chain, dim3(2), dim3(64 * waves), 0, c->stream, Pp, Ap, n_complex, c->dc_work_stride, c->d_dc_state, c->d_dc_counters,
                               c->dc_rounds > 0 ? c->dc_rounds : kDcMaxIter);
        } else {
            hipLaunchKernelGGL(k_dc_chain, dim3(2), dim3(64), 0, c->stream, Pp, Ap, n_complex, c->dc_work_stride, c->d_dc_state);
        }
        hipLaunchKernelGGL(k_dc_apply, dim3((words + 255) / 256), dim3(256), 0, c->stream, reinterpret_cast<const unsigned *>(dev_bytes), Ap,
                           reinterpret_cast<float4 *>(c->d_raw_tiled), n_complex, c->dc_work_stride);
        mode = kRawTiled;
    } else if (correct_dc) {
        Bracket b(c, c->stream, KIND_INGEST, 0);
        const int par = (int)(c->dc_frames++ & 1ull);
        double2 *sums = reinterpret_cast<double2 *>(c->d_dc_tab + c->dc_tab_sums);
        hipLaunchKernelGGL(k_dc_block_sums, dim3(nchunks), dim3(64), 0, c->stream, reinterpret_cast<const unsigned *>(dev_bytes), n_complex,
                           c->d_dc_tab, sums);
        hipLaunchKernelGGL(k_ingest_u8_dc_fast, dim3(nchunks), dim3(64), 0, c->stream, reinterpret_cast<const unsigned *>(dev_bytes),
                           reinterpret_cast<float4 *>(c->d_raw_tiled), n_complex, c->d_dc_state + 2 * par, c->d_dc_state + 2 * (par ^ 1),
                           c->d_dc_tab, sums);
        mode = kRawTiled;
    }
    c->long_frame = correct_dc && !c->opt_dc_blocked;
    const int rc = enqueue_frame(c, dev_bytes, mode, egress);
    c->long_frame = false;
    if (rc == SDRX_OK)
        c->last_raw = mode;
    return rc;
}

int enqueue_u8(sdrx_ctx *c, const uint8_t *bytes, int n_complex, int correct_dc, bool egress)
{
    int rc = ensure_raw(c, (size_t)c->root_frame);
    if (rc)
        return rc;
    unsigned char *dst = c->d_raw_u8[c->frame_no & 1ull];
    rc = stage_host_frame(c, bytes, (size_t)n_complex * 2, dst);
    return rc ? rc : enqueue_u8_device(c, dst, n_complex, correct_dc, egress);
}

} // namespace

extern "C" {

int sdrx_process_device(sdrx_ctx *c, const void *dev_iq, int n_complex)
{
    int rc = check_frame_call(c, "sdrx_process_device", dev_iq, n_complex, true);
    if (rc)
        return rc;
    c->last_raw = -1;
    return enqueue_frame(c, dev_iq, kRawF32, false);
}

int sdrx_submit_device(sdrx_ctx *c, const void *dev_iq, int n_complex)
{
    int rc = check_frame_call(c, "sdrx_submit_device", dev_iq, n_complex, false);
    if (rc)
        return rc;
    c->last_raw = -1;
    return enqueue_frame(c, dev_iq, kRawF32, true);
}

int sdrx_submit(sdrx_ctx *c, const float *iq, int n_complex)
{
    int rc = check_frame_call(c, "sdrx_submit", iq, n_complex, false);
    return rc ? rc : enqueue_f32(c, iq, n_complex, true);
}

int sdrx_submit_u8(sdrx_ctx *c, const uint8_t *bytes, int n_complex, int correct_dc)
{
    int rc = check_frame_call(c, "sdrx_submit_u8", bytes, n_complex, false);
    return rc ? rc : enqueue_u8(c, bytes, n_complex, correct_dc, true);
}

// The frame `src` staged LAST (host floats or dongle bytes handed to sdrx_process* / sdrx_submit* of `src`) once
// more, through the tree of `c` -- two contexts on one device fed the same raw frame, as sdrj::demodData feeds
// every main VFO the same `samples` (sdrj.cpp:288-294) -- without a second host-to-device copy: `c` waits for
// src's upload event and reads src's device buffer.  That buffer is per frame parity: it stays untouched until
// `src` stages the frame after next, by which time the caller must have waited for this one on `c`.
// `same_as` (may be null): host cf32 the caller believes to BE that frame -- compared byte for byte with src's pinned staging
// copy first; SDRX_DIFFERENT and nothing queued when it is not.
static int submit_shared(sdrx_ctx *c, sdrx_ctx *src, const char *what, bool sync_call, const float *same_as = nullptr, int same_n = 0)
{
    if (!c || !src || c == src)
        return c ? fail(c, SDRX_EINVAL, "%s: needs another context as the source", what) : SDRX_EINVAL;
    if (!src->finalized || src->frame_no == 0 || (src->last_raw != kRawF32 && src->last_raw != kRawU8))
        return fail(c, SDRX_ESTATE, "%s: the source context has staged no host frame (floats or bytes without DC removal) yet", what);
    if (src->device != c->device)
        return fail(c, SDRX_EINVAL, "%s: the source context lives on device %d, this one on %d", what, src->device, c->device);
    const int p = (int)((src->frame_no - 1) & 1ull);
    if (same_as) {
        if (src->last_raw != kRawF32 || same_n != src->root_frame || !src->h_in[p] ||
            memcmp(src->h_in[p], same_as, (size_t)same_n * sizeof(float2)) != 0)
            return SDRX_DIFFERENT;
    }
    const void *frame = src->last_raw == kRawF32 ? (const void *)src->d_raw[p] : (const void *)src->d_raw_u8[p];
    int rc = check_frame_call(c, what, frame, src->root_frame, sync_call);
    if (rc)
        return rc;
    if (src->last_raw == kRawU8 && !c->root_direct)
        return fail(c, SDRX_EUNSUPPORTED, "%s: a wide level 0 (more than 4 parent-less VFOs) shares float frames only", what);
    // src's NEXT upload into this buffer (its frame after next) must not overtake this context's kernels: an event behind
    // them, which src's staging waits for.  One event per (reader, parity), acquired BEFORE anything is queued -- a failure
    // here leaves no frame in flight -- and re-recorded for every shared frame (a source that never restages, or a reader
    // fed through sdrx_process_device, does not pile events up).
    sdrx_ctx::SharedReader *slot = nullptr;
    for (auto &r : src->shared_readers[p])
        if (r.who == c)
            slot = &r;
    if (!slot) {
        hipEvent_t e = nullptr;
        HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        src->shared_readers[p].push_back({c, e, false});
        slot = &src->shared_readers[p].back();
    }
    HIPCHK(c, hipStreamWaitEvent(c->stream, src->ev_staged[p], 0));
    c->last_raw = -1; // (not this context's buffer: sdrx_get_raw is served by `src`)
    const int src_raw = src->last_raw;
    rc = enqueue_frame(c, frame, src_raw, true);
    if (rc)
        return rc;
    if (hipEventRecord(slot->ev, c->stream) == hipSuccess)
        slot->pending = true;
    else
        (void)hipStreamSynchronize(c->stream); // (the frame IS queued: order it the blunt way rather than report a failure)
    return SDRX_OK;
}

int sdrx_submit_shared(sdrx_ctx *c, sdrx_ctx *src) { return submit_shared(c, src, "sdrx_submit_shared", false); }

int sdrx_process_shared(sdrx_ctx *c, sdrx_ctx *src)
{
    const int rc = submit_shared(c, src, "sdrx_process_shared", true);
    return rc ? rc : sdrx_wait(c);
}

int sdrx_submit_if_same(sdrx_ctx *c, sdrx_ctx *src, const float *iq, int n_complex)
{
    if (!iq)
        return c ? fail(c, SDRX_EINVAL, "sdrx_submit_if_same: null frame pointer") : SDRX_EINVAL;
    return submit_shared(c, src, "sdrx_submit_if_same", false, iq, n_complex);
}

int sdrx_process_if_same(sdrx_ctx *c, sdrx_ctx *src, const float *iq, int n_complex)
{
    if (!iq)
        return c ? fail(c, SDRX_EINVAL, "sdrx_process_if_same: null frame pointer") : SDRX_EINVAL;
    const int rc = submit_shared(c, src, "sdrx_process_if_same", true, iq, n_complex);
    return rc ? rc : sdrx_wait(c);
}

int sdrx_in_flight(sdrx_ctx *c) { return c ? c->in_flight : SDRX_EINVAL; }

} // extern "C"

namespace {
// the oldest undelivered frame's payload copy, if sdrx_wait is the one to issue it (enqueue_frame: frames that carry the DC
// recurrence): the host waits for the frame's last kernel, then the copy goes out with nothing to wait for
int start_owed_copy(sdrx_ctx *c)
{
    if (c->in_flight <= 0)
        return SDRX_OK;
    const int p = (int)((c->frame_no - (unsigned long long)c->in_flight) & 1ull);
    if (!c->copy_owed[p])
        return SDRX_OK;
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t cs = (p && c->copy_stream2) ? c->copy_stream2 : c->copy_stream;
    HIPCHK(c, hipEventSynchronize(c->ev_tail[p]));
    HIPCHK(c, hipMemcpyAsync(c->h_pay[p], c->d_pay[p], c->pay_bytes, hipMemcpyDeviceToHost, cs));
    HIPCHK(c, hipEventRecord(c->ev_copied[p], cs));
    // only now: a wait retried after one of the calls above failed must find the copy still owed -- ev_copied[p] is still the
    // event of frame f - 2, which completed long ago, and h_pay[p] still holds THAT frame's payloads
    c->copy_owed[p] = false;
    return SDRX_OK;
}
// the oldest undelivered frame's payloads are in host memory afterwards (slot returned); no callbacks
// Fault injection for the hosts' error paths (tests/test_dropin_qt.py): SDRX_FAULT_WAIT=k makes the k-th sdrx_wait of the PROCESS
// fail like a HIP error does -- before the frame leaves the queue, and for good: every later frame call of that context fails
// too (HIP errors are sticky).  One shot per process, so that a host which recovers by building a new context gets a sound one.
bool injected_fault(sdrx_ctx *c, bool at_wait)
{
    static std::atomic<long> countdown{getenv("SDRX_FAULT_WAIT") ? atol(getenv("SDRX_FAULT_WAIT")) : 0};
    if (c->broken)
        return true;
    if (at_wait && countdown.load() > 0 && countdown.fetch_sub(1) == 1)
        c->broken = true;
    return c->broken;
}

int wait_frame(sdrx_ctx *c, int *slot)
{
    if (c->in_flight <= 0)
        return fail(c, SDRX_ESTATE, "sdrx_wait: no submitted frame is in flight");
    if (injected_fault(c, true))
        return fail(c, SDRX_EHIP, "sdrx_wait: injected fault (SDRX_FAULT_WAIT): the context is unusable from here on");
    int rc = start_owed_copy(c);
    if (rc)
        return rc;
    HIPCHK(c, hipSetDevice(c->device));
    const unsigned long long f = c->frame_no - (unsigned long long)c->in_flight; // the oldest undelivered frame
    const int p = (int)(f & 1ull);
    HIPCHK(c, hipEventSynchronize(c->ev_copied[p]));
    c->in_flight--;
    c->host_slot = p;
    if (c->in_flight == 0)
        drain_events(c);
    *slot = p;
    return SDRX_OK;
}
} // namespace

extern "C" {

int sdrx_wait(sdrx_ctx *c)
{
    if (!c)
        return SDRX_EINVAL;
    int p = 0;
    const int rc = wait_frame(c, &p);
    if (rc)
        return rc;
    publish_all(c, p);
    return SDRX_OK;
}

int sdrx_sync(sdrx_ctx *c)
{
    if (!c)
        return SDRX_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    return drain(c);
}

int sdrx_fetch(sdrx_ctx *c)
{
    if (!c)
        return SDRX_EINVAL;
    if (!c->finalized)
        return fail(c, SDRX_ESTATE, "sdrx_fetch before sdrx_finalize");
    if (c->in_flight > 0)
        return fail(c, SDRX_ESTATE, "sdrx_fetch: %d submitted frame(s) not yet delivered -- call sdrx_wait", c->in_flight);
    if (c->frame_no == 0)
        return fail(c, SDRX_ESTATE, "sdrx_fetch: no frame processed yet");
    HIPCHK(c, hipSetDevice(c->device));
    const int p = (int)((c->frame_no - 1) & 1ull); // the last frame's payloads
    hipStream_t ts = c->opt_pipeline ? c->tail_stream : c->stream;
    if (int rc = pipeline_flush(c)) // frames still inside the software pipeline run to their end first
        return rc;
    if (c->pending_fetch)
        HIPCHK(c, hipMemcpyAsync(c->h_pay[p], c->d_pay[p], c->pay_bytes, hipMemcpyDeviceToHost, ts));
    int rc = drain(c);
    if (rc)
        return rc;
    c->pending_fetch = false;
    publish_all(c, p);
    return SDRX_OK;
}

int sdrx_process(sdrx_ctx *c, const float *iq, int n_complex)
{
    int rc = check_frame_call(c, "sdrx_process", iq, n_complex, true);
    if (rc)
        return rc;
    rc = enqueue_f32(c, iq, n_complex, true);
    return rc ? rc : sdrx_wait(c);
}

int sdrx_process_u8(sdrx_ctx *c, const uint8_t *bytes, int n_complex, int correct_dc)
{
    int rc = check_frame_call(c, "sdrx_process_u8", bytes, n_complex, true);
    if (rc)
        return rc;
    rc = enqueue_u8(c, bytes, n_complex, correct_dc, true);
    return rc ? rc : sdrx_wait(c);
}

int sdrx_get_output(sdrx_ctx *c, int id, const void **buf, uint32_t *len, uint32_t *rate)
{
    if (!c || id < 0 || id >= (int)c->nodes.size())
        return fail(c, SDRX_EINVAL, "bad vfo id %d", id);
    if (!c->finalized)
        return fail(c, SDRX_ESTATE, "sdrx_get_output before sdrx_finalize");
    const Node &n = c->nodes[(size_t)id];
    if (!n.leaf)
        return fail(c, SDRX_EINVAL, "vfo %d has children and publishes nothing (vfo.cpp:253-266)", id);
    if (c->in_flight > 0 && c->host_slot < 0)
        return fail(c, SDRX_ESTATE, "sdrx_get_output: %d submitted frame(s), none delivered yet -- call sdrx_wait first", c->in_flight);
    if (c->pending_fetch) { // frames queued with sdrx_process_device: bring the last one's payloads over
        int rc = sdrx_fetch(c);
        if (rc)
            return rc;
    }
    if (buf) {
        if (c->host_slot < 0)
            return fail(c, SDRX_ESTATE, "sdrx_get_output: no frame has been delivered yet");
        *buf = c->h_pay[c->host_slot] + n.pay_off;
    }
    if (len)
        *len = n.pay_len;
    if (rate)
        *rate = n.rate;
    return SDRX_OK;
}

int sdrx_get_raw(sdrx_ctx *c, float *out, int max_complex, int *n_ret)
{
    if (!c || !out)
        return SDRX_EINVAL;
    if (!c->finalized || c->frame_no == 0)
        return fail(c, SDRX_ESTATE, "sdrx_get_raw: no frame processed yet");
    if (c->in_flight > 0)
        return fail(c, SDRX_ESTATE, "sdrx_get_raw: %d submitted frame(s) not yet delivered -- call sdrx_wait first", c->in_flight);
    if (c->last_raw < 0)
        return fail(c, SDRX_ESTATE, "sdrx_get_raw: the last frame was caller-owned device memory (sdrx_process_device)");
    HIPCHK(c, hipSetDevice(c->device));
    const int n = std::min(max_complex, c->root_frame);
    if (int rc = drain(c))
        return rc;
    if (c->last_raw == kRawF32) {
        HIPCHK(c, hipMemcpy(out, c->d_raw[(c->frame_no - 1) & 1ull], (size_t)n * sizeof(float2), hipMemcpyDeviceToHost));
    } else if (c->last_raw == kRawU8) { // floats[b] = b - 127, jonti/sdr.cpp:43-49
        std::vector<uint8_t> b((size_t)n * 2);
        HIPCHK(c, hipMemcpy(b.data(), c->d_raw_u8[(c->frame_no - 1) & 1ull], b.size(), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < b.size(); ++i)
            out[i] = (float)((int)b[i] - 127);
    } else { // tile layout (the DC-bias kernels wrote it): unit (chunk, i2, lane) = samples 16 lane + 2 i2, +1
        const size_t total = align_up((size_t)c->root_frame, kChunk);
        std::vector<float2> t(total);
        HIPCHK(c, hipMemcpy(t.data(), c->d_raw_tiled, total * sizeof(float2), hipMemcpyDeviceToHost));
        for (int g = 0; g < n; ++g) {
            const int ch = g >> 10, r = g & 1023, ln = r >> 4, i = r & 15;
            const float2 v = t[(size_t)ch * 1024 + (size_t)(i >> 1) * 128 + (size_t)ln * 2 + (size_t)(i & 1)];
            out[2 * g] = v.x;
            out[2 * g + 1] = v.y;
        }
    }
    if (n_ret)
        *n_ret = n;
    return SDRX_OK;
}

int sdrx_get_stream(sdrx_ctx *c, int id, float *out, int max_complex, int *n_ret)
{
    if (!c || id < 0 || id >= (int)c->nodes.size())
        return fail(c, SDRX_EINVAL, "bad vfo id %d", id);
    if (!c->finalized || c->frame_no == 0)
        return fail(c, SDRX_ESTATE, "sdrx_get_stream: no frame processed yet");
    if (c->in_flight > 0) // the stream buffers already belong to the newest submitted frame, not to the last delivered one
        return fail(c, SDRX_ESTATE, "sdrx_get_stream: %d submitted frame(s) not yet delivered -- call sdrx_wait first", c->in_flight);
    const Node &n = c->nodes[(size_t)id];
    const int par = (int)((c->frame_no - 1) & 1ull);
    const int cnt = std::min(max_complex, n.n_f);
    HIPCHK(c, hipSetDevice(c->device));
    if (int rc = drain(c))
        return rc;
    const auto tap = c->taps.find(id);
    if (!n.has_stream && !(tap != c->taps.end() && c->frame_no > tap->second.since))
        return fail(c, SDRX_ENOSTREAM,
                    "sdrx_get_stream: vfo %d %s inside the mix wave and keeps no decimate[%d] -- select it with sdrx_set_tap / "
                    "sdrx_add_tap before the frame (the reference's fftVFOSlot), or set option keep_streams=1 / %s=0",
                    id, n.fused_late ? "decimates by 5 / 6" : "demodulates", n.d.decimate_count, n.fused_late ? "fuse_late" : "fuse_demod");
    if (out && cnt > 0) {
        if (!n.has_stream) {
            HIPCHK(c, hipMemcpy(out, tap->second.buf[par], sizeof(float2) * (size_t)cnt, hipMemcpyDeviceToHost));
        } else if (n.leaf) {
            HIPCHK(c, hipMemcpy(out, c->arena + n.off_stream[par] + sizeof(float2) * (size_t)n.Hx, sizeof(float2) * (size_t)cnt,
                                hipMemcpyDeviceToHost));
        } else {
            // tile layout on the device: undo it for the caller (fftData carries natural order)
            const size_t total = align_up((size_t)n.n_f, kChunk);
            std::vector<float2> tmp(total);
            HIPCHK(c, hipMemcpy(tmp.data(), c->arena + n.off_stream[par], sizeof(float2) * total, hipMemcpyDeviceToHost));
            float2 *o = reinterpret_cast<float2 *>(out);
            for (int g = 0; g < cnt; ++g) {
                const int ch = g >> 10, r = g & 1023, ln = r >> 4, i = r & 15;
                o[g] = tmp[(size_t)ch * 1024 + (size_t)(i >> 1) * 128 + (size_t)ln * 2 + (size_t)(i & 1)];
            }
        }
    }
    if (n_ret)
        *n_ret = n.n_f;
    return SDRX_OK;
}

int sdrx_get_prequant(sdrx_ctx *c, int id, float *out, int max, int *n_ret)
{
    if (!c || id < 0 || id >= (int)c->nodes.size())
        return fail(c, SDRX_EINVAL, "bad vfo id %d", id);
    const Node &n = c->nodes[(size_t)id];
    if (!c->finalized || !c->opt_prequant || !n.leaf || !n.d.demod_usb)
        return fail(c, SDRX_ESTATE, "sdrx_get_prequant: set option keep_prequant=1 before finalize; USB leaves only");
    if (c->in_flight > 0)
        return fail(c, SDRX_ESTATE, "sdrx_get_prequant: %d submitted frame(s) not yet delivered -- call sdrx_wait first", c->in_flight);
    const int cnt = std::min(max, n.n_out);
    HIPCHK(c, hipSetDevice(c->device));
    if (int rc = drain(c))
        return rc;
    if (out && cnt > 0)
        HIPCHK(c, hipMemcpy(out, c->arena + n.off_preq, sizeof(float) * (size_t)cnt, hipMemcpyDeviceToHost));
    if (n_ret)
        *n_ret = n.n_out;
    return SDRX_OK;
}

int sdrx_get_taps(sdrx_ctx *c, int id, int which, float *out, int max, int *n_ret)
{
    if (!c || id < 0 || id >= (int)c->nodes.size())
        return fail(c, SDRX_EINVAL, "bad vfo id %d", id);
    if (!c->finalized)
        return fail(c, SDRX_ESTATE, "sdrx_get_taps before sdrx_finalize");
    const Node &n = c->nodes[(size_t)id];
    const std::vector<float> *t = which == 0 ? &n.lpf : which == 1 ? &n.dec : which == 2 ? &n.hilbert : nullptr;
    if (!t)
        return fail(c, SDRX_EINVAL, "which must be 0, 1 or 2");
    // read back what the kernels actually use (device copy), not the host vector
    const size_t off = which == 0 ? n.off_lpf + (n.long_lpf ? 0 : 3 * sizeof(float)) : which == 1 ? n.off_dec : n.off_hilbert;
    const int cnt = std::min(max, (int)t->size());
    HIPCHK(c, hipSetDevice(c->device));
    if (out && cnt > 0)
        HIPCHK(c, hipMemcpy(out, c->arena + off, sizeof(float) * (size_t)cnt, hipMemcpyDeviceToHost));
    if (n_ret)
        *n_ret = (int)t->size();
    return SDRX_OK;
}

int sdrx_get_nco(sdrx_ctx *c, int id, long first, long count, float *out)
{
    if (!c || id < 0 || id >= (int)c->nodes.size())
        return fail(c, SDRX_EINVAL, "bad vfo id %d", id);
    if (!c->finalized)
        return fail(c, SDRX_ESTATE, "sdrx_get_nco before sdrx_finalize");
    const Node &n = c->nodes[(size_t)id];
    if (first < 0 || count < 0 || first + count > n.d.fs)
        return fail(c, SDRX_EINVAL, "table range [%ld,%ld) outside 0..%d", first, first + count, n.d.fs);
    if (count == 0)
        return SDRX_OK;
    HIPCHK(c, hipSetDevice(c->device));
    float2 *tmp = nullptr;
    HIPCHK(c, hipMalloc(&tmp, sizeof(float2) * (size_t)count));
    hipLaunchKernelGGL(k_nco_dump, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, c->stream,
                       reinterpret_cast<const float2 *>(c->arena + n.off_cp), n.rot_re, n.rot_im, first, count, tmp);
    hipError_t e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess)
        e = hipMemcpy(out, tmp, sizeof(float2) * (size_t)count, hipMemcpyDeviceToHost);
    (void)hipFree(tmp);
    if (e != hipSuccess)
        return fail(c, SDRX_EHIP, "sdrx_get_nco: %s", hipGetErrorString(e));
    return SDRX_OK;
}

int sdrx_get_stats(sdrx_ctx *c, sdrx_stats *s)
{
    if (!c || !s)
        return SDRX_EINVAL;
    memset(s, 0, sizeof *s);
    s->n_vfos = (int)c->nodes.size();
    for (const Node &n : c->nodes)
        s->n_leaves += n.children.empty();
    s->n_levels = c->n_levels;
    s->exact = c->opt_exact;
    s->algorithmic_bytes_per_frame = c->alg_bytes;
    s->vfo_samples_per_frame = c->vfo_samples;
    s->device_bytes = (int64_t)(c->arena_bytes + 2 * c->pay_bytes + c->raw_cap * 10);
    s->frames = (int64_t)c->frame_no;
    s->mix_chunks_per_frame = c->mix_chunks;
    if (c->d_dc_counters) { // (waits for what is queued: a measurement call)
        unsigned long long h[3] = {0, 0, 0};
        HIPCHK(c, hipSetDevice(c->device));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        HIPCHK(c, hipMemcpy(h, c->d_dc_counters, sizeof h, hipMemcpyDeviceToHost));
        s->dc_blocks = (int64_t)h[0];
        s->dc_fallback_blocks = (int64_t)h[1];
        s->dc_retried_blocks = (int64_t)h[2];
    }
    return SDRX_OK;
}

int sdrx_enable_kernel_timing(sdrx_ctx *c, int enable)
{
    if (!c)
        return SDRX_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    if (int rc = drain(c))
        return rc;
    c->timing = enable != 0;
    for (int k = 0; k < SDRX_NKERNELS; ++k) {
        c->t_ms[k] = 0;
        c->t_n[k] = 0;
        c->t_bytes[k] = 0;
    }
    return SDRX_OK;
}

int sdrx_get_kernel_times(sdrx_ctx *c, double ms[SDRX_NKERNELS], int64_t launches[SDRX_NKERNELS],
                          int64_t alg_bytes[SDRX_NKERNELS])
{
    if (!c)
        return SDRX_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    if (int rc = drain(c))
        return rc;
    for (int k = 0; k < SDRX_NKERNELS; ++k) {
        if (ms)
            ms[k] = c->t_ms[k];
        if (launches)
            launches[k] = c->t_n[k];
        if (alg_bytes)
            alg_bytes[k] = c->t_bytes[k];
    }
    return SDRX_OK;
}

} // extern "C"

#include "sdrx_group.hip"
