/* include/sdrx.h -- C ABI of the MI355X-native per-VFO IQ chain (libsdrx.so).
 *
 * Drop-in boundary for ONE path of jeroenbeijer/SDRReceiver: the per-VFO chain
 *   table-NCO complex mix -> cascaded 11-tap half-band decimation -> USB demodulation
 *   (62-sample delay minus 125-tap Hilbert) -> optional Hamming low-pass -> int16
 * i.e. the arithmetic of vfo.cpp / oscillator.cpp / halfbanddecimator.cpp / jonti/dsp.cpp /
 * gnuradio/firfilter.cpp.  Everything around it (Qt GUI, RTL-SDR / rtl_tcp ingest, the ZeroMQ
 * socket) stays on the host side of this boundary; INTEGRATION.md shows the reference-side
 * binding.  All citations are file:line in the reference repository.
 *
 * Conventions: plain C types only, every function returns 0 on success and a negative
 * SDRX_E* code on failure (no exception crosses the ABI; sdrx_last_error() has the text).
 * A context is single-caller and not re-entrant -- like the reference, where all VFOs run on
 * one thread (sdrj.cpp:288-294).  There is no CPU fallback: without a usable HIP device
 * sdrx_create() fails.
 */
#ifndef SDRX_H
#define SDRX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SDRX_ABI_VERSION 5

enum {
    SDRX_OK = 0,
    SDRX_EINVAL = -1,   /* bad argument / bad descriptor                               */
    SDRX_ESTATE = -2,   /* call order violated (e.g. process before finalize)            */
    SDRX_EFILTER = -3,  /* tap design rejected: where the reference throws out_of_range  */
                        /* from firfilter::sanity_check_1f (firfilter.cpp:122-134)       */
    SDRX_EHIP = -4,     /* HIP runtime error                                             */
    SDRX_EUNSUPPORTED = -5, /* geometry outside what the kernels handle (see sdrx_finalize) */
    SDRX_ENOMEM = -6,
    SDRX_ENOSTREAM = -7 /* sdrx_get_stream: this VFO keeps no decimate[d] (a fused /5 | /6 leaf that is not a tap)  */
};

typedef struct sdrx_ctx sdrx_ctx;

/* One VFO node.  Each field replaces one reference setter; sdrx_add_vfo + sdrx_finalize
 * together replace `new vfo` + setters + vfo::init (vfo.cpp:60-176, called from
 * mainwindow.cpp:105-136 for main VFOs and 150-225 for sub VFOs). */
typedef struct sdrx_vfo_desc {
    int32_t fs;                 /* vfo::setFs                 vfo.cpp:189-193: input rate of THIS vfo */
    int32_t decimate_count;     /* vfo::setDecimationCount    vfo.cpp:194-197: half-band stages, 0..8 */
    double mixer_freq_hz;       /* vfo::setMixerFreq          vfo.cpp:199-204: integer Hz, may be < 0 */
    int32_t demod_usb;          /* vfo::setDemodUSB           vfo.cpp:468-472: 1 = USB audio leaf      */
    int32_t late_decimate;      /* vfo::init(..,lateDecimate) vfo.cpp:70-101: 0, 5 or 6               */
    int32_t filter_bw_hz;       /* vfo::setFilterBandwidth    vfo.cpp:223-227: 0 = no audio low-pass   */
    float gain;                 /* vfo::setGain               vfo.cpp:229-233                          */
    int32_t cstyle;             /* vfo::setCompressonStyle    vfo.cpp:455-460 (compress(), 389-424)    */
    int32_t scalecomp;          /* vfo::setScaleComp          vfo.cpp:462-467                          */
    int32_t parent_id;          /* vfo::setVFOs on the parent vfo.cpp:485-490; -1 = fed by the raw     */
                                /*   stream, i.e. a member of sdrj's main-VFO list (sdrj.cpp:288-294)  */
    int32_t samples_per_buffer; /* vfo::init(samplesPerBuffer) vfo.cpp:60: complex samples per frame   */
    char topic[8];              /* vfo::setZmqTopic: the first 5 bytes go on the wire                  */
                                /*   (zmqpublisher.cpp:91)                                              */
} sdrx_vfo_desc;

/* Replaces the call `ZmqPublisher::publish(buf, len, topic, sampleRate)` made by
 * vfo::transmitData (vfo.cpp:426-453, zmqpublisher.h:16).  Invoked once per publishing leaf per
 * frame, in the reference's order: main VFOs in list order, their sub VFOs in list order
 * (sdrj.cpp:288-294, vfo.cpp:257-263).  `buf` is owned by the library and valid until the next
 * sdrx_process*() / sdrx_wait() / sdrx_fetch() / sdrx_destroy() -- libzmq copies on zmq_send
 * (zmqpublisher.cpp:91-93), so the reference's publisher needs it no longer than the callback.
 * Not invoked for an empty payload (zmqpublisher.cpp:88). */
typedef void (*sdrx_publish_fn)(void *user, const char topic[5], uint32_t sample_rate, const void *buf,
                                uint32_t len_bytes);

/* ---- lifetime ------------------------------------------------------------------------------ */
int sdrx_abi_version(void);
/* A short hash of the sources this library was compiled from (csrc/Makefile): what profiles and bench lines name as the
 * build they measured. */
const char *sdrx_build_id(void);
int sdrx_create(sdrx_ctx **ctx, int device_ordinal);
int sdrx_destroy(sdrx_ctx *ctx);
const char *sdrx_last_error(const sdrx_ctx *ctx); /* ctx may be NULL: error of a failed create */

/* ---- configuration ( = MainWindow building the VFO tree) ------------------------------------ */
int sdrx_add_vfo(sdrx_ctx *ctx, const sdrx_vfo_desc *desc, int *id_out);
/* Options, all before sdrx_finalize:
 *   "exact"  1 (default): every fp32 operation rounded like the reference's -O2 x86-64 build
 *            (no FMA contraction, reference summation order) -> results bit-identical to it.
 *            0: the TOLERANCE arithmetic -- what BASELINE.json's north_star grants ("within 1e-5 relative float
 *            tolerance"): every final complex stream and pre-quantisation float within 1e-5 of max|reference| per
 *            VFO-frame, int16 within 1 LSB (tests: every VFO of configs 3 / 4 and of the 10 240-VFO workload, 18 s runs
 *            without drift, the -Ofast build's fixtures).  The NCO table entries (oscillator.cpp:20-28) become rotations
 *            of the table's EXACT 16-entry checkpoints (<= 1e-6 from the table, never accumulating; the table's first 512
 *            entries and the first sample ever are still replayed exactly), the mixer (vfo.cpp:241) a packed multiply +
 *            FMA, the half-band / FIR dot products FMAs: ~230 instead of ~390 vector instructions per 1024 samples.
 *            Its NCO error (~1e-6 of |v|) multiplies the TOTAL input power while the bar is relative to the VFO's own
 *            output: a strong out-of-band carrier over a quiet channel eats the margin -- 100 LSB of carrier over +-1 LSB of
 *            noise: 6.4e-6 of max|stream|, 8.3e-6 on the pre-quantisation float
 *            (tests/test_gpu_parity.py::test_strong_carrier_over_quiet_channels).
 *            2: the ROBUST arithmetic -- the table entries from the exact recurrence (bit-identical to the reference's
 *            table: no NCO error at all), the mixer and every filter as FMAs as with 0 (~260 instructions per 1024 samples).
 *            What is left is FMA versus two roundings -- the kind of difference the reference's own -Ofast build (as shipped)
 *            has against its -O2 build, and on that adversarial input the same size: 4.3e-6 against their 4.0e-6.
 *   "keep_prequant" 1: also keep the pre-quantisation float `usb*gain*32768` per leaf
 *            (parity tests; sdrx_get_prequant).   default 0
 *   "segments" n: force n time-segments per VFO-frame in the decimation kernel (0 = auto).
 *   "pipeline" 0 (default) | 1: with 1 the leaf tail of a frame (late decimation, USB
 *            demodulation, compress) runs on a second HIP stream behind an event, so that it may
 *            overlap the mix/decimate launches of the NEXT frame when frames are queued back to
 *            back; results are bit-identical.  Measured on MI355X (profiles/README.md): the two
 *            kernels then share a VALU-bound machine and both stretch -- 0.119 vs 0.109 ms per
 *            frame on BASELINE config 3 -- so it is off by default and kept as an A/B switch.
 *   "fuse" 1 (default) | 0: frames queued on the device (sdrx_process_device) run the mix/decimate items
 *            of ALL tree levels in one launch (k_mix_levels); 0 = one launch per tree level (A/B switch).
 *   "frame_pipeline" 1 (default) | 0: with "fuse", level l of that one launch works on frame k - l (a
 *            software pipeline over the frames queued back to back), so a frame queued with
 *            sdrx_process_device is only COMPLETE after the next such call or after sdrx_sync /
 *            sdrx_fetch / sdrx_get_* (which run what is outstanding).  0 = every call runs its frame through
 *            all levels at once.  Results are bit-identical either way.
 *   "fuse_late" 1 (default) | 0: a USB leaf with decimate_count 0 and late_decimate 5 | 6 below a parent
 *            (vfo::usb_decimdemod, vfo.cpp:334-387: the reference mixes, low-passes and keeps every 5th / 6th
 *            sample in one pass) runs its decimating low-pass inside the mix wave and writes only the decimated
 *            stream to HBM; decimate[0] of such a leaf is kept only while it is the tap (sdrx_set_tap) or with
 *            "keep_streams".  0 = the two-kernel form (the mixed stream goes to HBM and comes back): A/B switch.
 *            Results are bit-identical either way.
 *   "fuse_demod" 0 (default) | 1: with 1 a USB leaf with decimate_count 2 below a parent and an audio low-pass of at most 64
 *            taps (the reference's 48 kS/s sub VFOs, vfo::usb_demod, vfo.cpp:300-332) demodulates inside its mix wave: 256
 *            stream samples per 1024-sample chunk go from registers through the wave's LDS to int16 and the leaf's cf32
 *            stream never reaches HBM (decimate[2] of such a leaf is kept only while it is the tap, sdrx_set_tap, or with
 *            "keep_streams").  Results are bit-identical.  Measured on MI355X (round 6, profiles/README.md): it removes
 *            138 MB of HBM traffic per frame of BASELINE config 3 and is SLOWER -- 0.123 vs 0.112 ms per frame exact, 0.102 vs
 *            0.091 tolerance: the frame is bound by VALU issue, not by HBM, and the demodulation's plain fp32 MACs issue in
 *            pairs only beside other waves doing the same (k_usb_demod: 80 % paired at 7 waves per SIMD) -- inside the mix
 *            wave, five waves per SIMD most of which are in packed-fp32 phases, they do not.  Off by default; an A/B switch.
 *   "keep_streams" 0 (default) | 1: every such leaf also keeps decimate[d] of every frame (parity tests that
 *            compare every stream of the tree).
 *   "dc_blocked_scan" 0|1 (default 0): how sdrx_process_u8 removes the DC bias.  0 = the
 *                 reference's sequentially rounded fp32 recurrence, bit for bit (below).  1 = the same linear filter as a
 *                 blocked parallel scan (~15 us): the true IIR response.  The reference's recurrence
 *                 wanders around that by up to ~3e-3 of the DC offset (its rounding errors are
 *                 correlated from step to step), so 1 is NOT within the 1e-5 parity tolerance
 *                 of the reference unless the offset is well below 1 LSB.
 *   "dc_speculative" 1 (default) | 0: how the bit-exact recurrence is evaluated.  1 = blocks of 1024 samples as integer
 *                 prefix sums of the mantissa, several blocks side by side per step, each step VERIFIED (binade, sign
 *                 and the rounding of avept * (1 - 1e-6) unchanged through it, no exact ties), taken again block by
 *                 block where that fails and redone with the sequential operations where a single block fails: bit-exact
 *                 by construction, at worst the sequential time (sdrx_stats.dc_blocks / dc_retried_blocks /
 *                 dc_fallback_blocks).
 *   "dc_blocks_per_step" 1 | 2 | 4 | 8: how many 1024-sample blocks one step of that evaluation takes side by
 *                 side (= waves of the one workgroup per component).  Same results for every value; anything else is
 *                 SDRX_EINVAL.  (The sequential recurrence for every sample -- ~2.0 ms per frame: two waves, each alone
 *                 with its dependent chain -- is option "dc_speculative" = 0, not a value of this one.) */
int sdrx_set_option(sdrx_ctx *ctx, const char *name, int value);
/* All of vfo::init for every node: NCO tables (oscillator.cpp:4-32), low-pass designs
 * (firfilter.cpp:64-119), Hilbert taps (dsp.cpp:184-217), zeroed filter state, buffers.
 * SDRX_EFILTER where the reference would throw; SDRX_EUNSUPPORTED unless for every node
 * fs % 16 == 0, samples_per_buffer % 16 == 0 and samples_per_buffer % 2^decimate_count == 0
 * (true for every rate the reference accepts, mainwindow.h:29), and a child's
 * samples_per_buffer equals its parent's samples_per_buffer / 2^decimate_count. */
int sdrx_finalize(sdrx_ctx *ctx);
int sdrx_set_publish_callback(sdrx_ctx *ctx, sdrx_publish_fn fn, void *user);
/* What vfo::init would do with this descriptor, decided on the host without touching a device (so a
 * binding can fail at init() time exactly where the reference does): SDRX_OK; SDRX_EFILTER where
 * firfilter::sanity_check_1f throws std::out_of_range (firfilter.cpp:122-134) -- `msg` then holds the
 * reference's what() text, e.g. "firdes check failed: 0 < fa <= sampling_freq / 2"; SDRX_EINVAL /
 * SDRX_EUNSUPPORTED for descriptors sdrx_add_vfo / sdrx_finalize refuse (`msg` says why).  `msg` may
 * be NULL. */
int sdrx_check_vfo(const sdrx_vfo_desc *desc, char *msg, size_t msg_cap);

/* ---- per frame ( = sdrj::demodData, sdrj.cpp:266-305) ---------------------------------------- */
/* `iq`: n_complex interleaved (I,Q) float pairs on the HOST, as sdr::audio_signal_out /
 * sdrj::readyRead deliver them (values b-127, jonti/sdr.cpp:43-49).  n_complex must equal the
 * samples_per_buffer of the parent-less VFOs.  Synchronous: on return every payload is in host
 * memory and the publish callback has run for every leaf.  The DC-bias IIR of sdrj.cpp:271-286
 * stays on the caller's side of this entry point (or use sdrx_process_u8). */
int sdrx_process(sdrx_ctx *ctx, const float *iq, int n_complex);
/* Raw dongle bytes (2 per complex sample, unsigned, offset 127) with the byte->float LUT
 * (jonti/sdr.cpp:43-49,122-129; sdrj.cpp:155-160) and, if correct_dc != 0, the DC-bias IIR
 * (sdrj.cpp:271-286) done on the device.  Otherwise like sdrx_process. */
int sdrx_process_u8(sdrx_ctx *ctx, const uint8_t *iq_bytes, int n_complex, int correct_dc);

/* Device-resident variant for pipelines that already hold the frame in HBM (bench.py, the
 * multi-GPU path where the frame arrives by RCCL broadcast): `dev_iq` is a DEVICE pointer to
 * n_complex cf32.  Asynchronous on the context's stream; sdrx_fetch() waits, copies the payloads
 * to the host and runs the callbacks; sdrx_sync() only waits. */
int sdrx_process_device(sdrx_ctx *ctx, const void *dev_iq, int n_complex);
int sdrx_fetch(sdrx_ctx *ctx);
int sdrx_sync(sdrx_ctx *ctx);
/* Run on a caller-provided hipStream_t (e.g. torch's current stream) instead of the context's
 * own; NULL restores the default.  The frame is consumed on that stream (work the caller queues on
 * it after sdrx_process_device may overwrite the frame); with option "pipeline" the leaf tail runs
 * on a stream of the library's own, which sdrx_sync / sdrx_fetch / sdrx_wait also wait for. */
int sdrx_set_stream(sdrx_ctx *ctx, void *hip_stream);

/* ---- pipelined per-frame interface ("optional async submit/wait pair") ---------------
 * sdrx_submit* enqueue one frame and return at once: host -> device copy of the frame (staged through
 * pinned memory of the library's own: `iq` is borrowed for the duration of the call only, like the
 * argument of sdrj::demodData), kernels, and the device -> host copy of that frame's payloads on a
 * copy stream -- which therefore overlaps the kernels of the next frame.  (Frames of sdrx_submit_u8 with correct_dc: that
 * copy is issued by sdrx_wait instead, once the frame's kernels have ended -- measured, it is the order in which the next
 * frame's DC recurrence and the copy do run side by side)  sdrx_wait delivers the
 * OLDEST frame not yet delivered: it blocks until that frame's payloads are in host memory, then
 * runs the publish callback for every leaf in the reference's order (= ZmqPublisher::publish per
 * leaf, vfo.cpp:426-453); afterwards sdrx_get_output serves that frame.  At most
 * SDRX_MAX_IN_FLIGHT frames may be submitted and not yet delivered (SDRX_ESTATE otherwise).
 * The steady state of a streaming host is  submit(f+1); wait() -> f;  i.e. one frame of latency in
 * exchange for PCIe and kernels running concurrently.  sdrx_process* are submit + wait of one frame.
 * While frames are in flight sdrx_get_output keeps serving the last DELIVERED frame (its payloads sit in
 * host memory); the synchronous calls (sdrx_process*, sdrx_fetch) and the device read-backs
 * (sdrx_get_stream, sdrx_get_raw, sdrx_get_prequant) return SDRX_ESTATE -- the device buffers behind them
 * already belong to a newer frame.  sdrx_get_kernel_times waits for everything queued. */
#define SDRX_MAX_IN_FLIGHT 2
int sdrx_submit(sdrx_ctx *ctx, const float *iq, int n_complex);
int sdrx_submit_u8(sdrx_ctx *ctx, const uint8_t *iq_bytes, int n_complex, int correct_dc);
int sdrx_submit_device(sdrx_ctx *ctx, const void *dev_iq, int n_complex);
/* Two contexts on ONE device fed the same raw frame -- sdrj::demodData hands every main VFO the same `samples`
 * (sdrj.cpp:288-294), and a binding that keeps one context per main VFO (host/qt/vfo_adapter.cpp) would otherwise
 * upload the frame once per main: run through `ctx` the frame `src` staged LAST (host floats or dongle bytes given
 * to sdrx_process* / sdrx_submit* of `src`; not after a DC-bias removal on the device) without another
 * host-to-device copy.  `ctx` waits for src's upload on the device.  src's frame buffers are per frame parity: the
 * shared frame stays valid until `src` stages the frame after next -- wait for it on `ctx` before that.  `ctx` and `src`
 * must be driven from one thread (the call touches both; contexts carry no locks -- like the reference, where every VFO
 * runs on the one thread of sdrj::demodData). */
int sdrx_submit_shared(sdrx_ctx *ctx, sdrx_ctx *src);
int sdrx_process_shared(sdrx_ctx *ctx, sdrx_ctx *src); /* = sdrx_submit_shared + sdrx_wait */
/* The same for a binding that cannot KNOW that two main VFOs were handed the same samples (host/qt/vfo_adapter.cpp: `class
 * vfo` only sees process(samples) calls): `iq` is compared byte for byte with the pinned staging copy of the frame `src`
 * staged last (a memcmp of the frame instead of its upload).  Equal: exactly sdrx_submit_shared / sdrx_process_shared.  Not
 * equal (or src staged bytes, not floats): SDRX_DIFFERENT (> 0), nothing queued -- submit the frame normally. */
#define SDRX_DIFFERENT 1
int sdrx_submit_if_same(sdrx_ctx *ctx, sdrx_ctx *src, const float *iq, int n_complex);
int sdrx_process_if_same(sdrx_ctx *ctx, sdrx_ctx *src, const float *iq, int n_complex);
int sdrx_wait(sdrx_ctx *ctx);
int sdrx_in_flight(sdrx_ctx *ctx); /* >= 0: frames submitted and not yet delivered; < 0: error */

/* ---- results -------------------------------------------------------------------------------- */
/* Payload of leaf `id` after the last frame: int16 audio (USB leaf) or packed int8 IQ
 * (compress(), vfo.cpp:389-424).  *rate = outputRate (vfo.cpp:102). */
int sdrx_get_output(sdrx_ctx *ctx, int id, const void **buf, uint32_t *len_bytes, uint32_t *rate);
/* decimate[decimateCount] of node `id` (public member vfo.h:39 -- what the fftData signal
 * carries, vfo.cpp:290-293): copies up to max_complex cf32 to `out`, returns the count in *n. */
int sdrx_get_stream(sdrx_ctx *ctx, int id, float *out_iq, int max_complex, int *n);
/* fftVFOSlot(topic) (vfo.cpp:492-509, sdrj.cpp:84-101): the GUI names the VFO(s) whose decimate[decimateCount] it wants
 * from the next frame on -- every VFO whose topic equals the selected string gets emitFFT, so an INI with one topic on two
 * VFOs (or several empty topics) has several taps.  Every node keeps that stream in HBM anyway, with one exception: a leaf
 * whose late decimation is fused into the mix wave (option "fuse_late") writes only its decimated stream -- sdrx_get_stream
 * on it returns SDRX_ENOSTREAM unless it was selected here before the frame was processed (or "keep_streams" is set).
 * sdrx_set_tap REPLACES the selection by `id` (-1: nothing selected: a deselected leaf stops writing its decimate[0]);
 * sdrx_add_tap adds `id` to it.  Not while submitted frames are in flight; contexts are single-caller (one thread). */
int sdrx_set_tap(sdrx_ctx *ctx, int id);
int sdrx_add_tap(sdrx_ctx *ctx, int id);
/* The raw frame exactly as the parent-less VFOs consumed it -- `samples` of sdrj::demodData
 * (sdrj.cpp:266-305) after the byte LUT and the DC-bias removal, what sdrj's own fftData signal
 * carries (sdrj.cpp:296-303) -- natural order, cf32.  Available after sdrx_process and
 * sdrx_process_u8; after sdrx_process_device the frame was the caller's own device memory and
 * the call returns SDRX_ESTATE. */
int sdrx_get_raw(sdrx_ctx *ctx, float *out_iq, int max_complex, int *n);
int sdrx_get_prequant(sdrx_ctx *ctx, int id, float *out, int max, int *n);
/* Designed tap sets, for parity checks: which = 0 audio low-pass, 1 late-decimation low-pass,
 * 2 Hilbert. */
int sdrx_get_taps(sdrx_ctx *ctx, int id, int which, float *out, int max, int *n);
/* NCO table entries [first, first+count) of node `id` as the device generated them. */
int sdrx_get_nco(sdrx_ctx *ctx, int id, long first, long count, float *out_iq);

/* ---- one tree on several GPUs, one host process ---------------------------------------------------
 * The fan-out the reference does on one thread -- sdrj::demodData over the main VFOs (sdrj.cpp:288-294),
 * each main over its sub VFOs (vfo.cpp:253-264) -- sharded over the devices of one node:
 * a group owns one sdrx_ctx per device.  sdrx_group_add_vfo describes the WHOLE tree exactly like
 * sdrx_add_vfo; sdrx_group_finalize gives device k of W, for every parent-less VFO with children, the
 * block [K k / W, K (k+1) / W) of its K children together with everything below them, plus a replica of
 * the parent where that block is not empty (parent-less leaves are block-partitioned among themselves).
 * Per frame the raw IQ lands on the FIRST device of the list and is fanned out to the others by one
 * peer-to-peer copy each (hipMemcpyPeerAsync on the receiving device's stream: over xGMI every peer is
 * one direct link from the source, the copies run on different links at once), double-buffered by frame
 * parity; every device then runs its shard and copies its payloads back itself.  The publish callback is
 * invoked in the reference's order over the whole tree, whichever device computed a leaf.  Ids are those
 * of sdrx_group_add_vfo.  `devices` may name a device more than once (two shards on one GPU: tests).
 * sdrx_group_submit* / _wait / _process mirror sdrx_submit* / sdrx_wait / sdrx_process.  For
 * sdrx_group_submit_device the frame (cf32, on the first device) must be complete in the order of
 * `producer_stream` (a hipStream_t of that device; NULL: complete already) and stay untouched until
 * it was waited for.  A device that ends up without VFOs (more devices than sub VFOs) stays idle. */
typedef struct sdrx_group sdrx_group;
int sdrx_group_create(sdrx_group **grp, const int *device_ordinals, int n_devices);
int sdrx_group_destroy(sdrx_group *grp);
const char *sdrx_group_last_error(const sdrx_group *grp); /* grp may be NULL: error of a failed create */
int sdrx_group_size(const sdrx_group *grp);
/* 1: every member reaches the first device's frame buffer directly (same device, or peer access enabled: one
 * xGMI link per peer, the N-1 copies run at once); 0: the runtime stages at least one peer copy through host
 * memory (still correct, slower) -- sdrx_group_last_error() right after sdrx_group_create names the device. */
int sdrx_group_peer_access(const sdrx_group *grp);
int sdrx_group_add_vfo(sdrx_group *grp, const sdrx_vfo_desc *desc, int *id_out);
int sdrx_group_set_option(sdrx_group *grp, const char *name, int value); /* applied to every member */
int sdrx_group_set_publish_callback(sdrx_group *grp, sdrx_publish_fn fn, void *user);
int sdrx_group_finalize(sdrx_group *grp);
int sdrx_group_process(sdrx_group *grp, const float *iq, int n_complex);
int sdrx_group_submit(sdrx_group *grp, const float *iq, int n_complex);
/* Dongle bytes: a quarter of the bytes cross PCIe and xGMI, every device applies the b - 127 LUT itself
 * (jonti/sdr.cpp:43-49) and, if correct_dc != 0, the DC-bias IIR of sdrj.cpp:271-286 on the whole frame with
 * an accumulator of its own -- identical bytes and identical start state keep the devices' estimates
 * identical, so only the bytes travel.  What the shipped sdr_25E.ini (correct_dc_bias=1) needs. */
int sdrx_group_submit_u8(sdrx_group *grp, const uint8_t *iq_bytes, int n_complex, int correct_dc);
int sdrx_group_process_u8(sdrx_group *grp, const uint8_t *iq_bytes, int n_complex, int correct_dc);
int sdrx_group_submit_device(sdrx_group *grp, const void *dev_iq_on_first_device, int n_complex, void *producer_stream);
/* like sdrx_process_device: kernels only, asynchronous; the frame must stay untouched until sdrx_group_sync */
int sdrx_group_process_device(sdrx_group *grp, const void *dev_iq_on_first_device, int n_complex, void *producer_stream);
int sdrx_group_wait(sdrx_group *grp);
int sdrx_group_in_flight(sdrx_group *grp);
int sdrx_group_sync(sdrx_group *grp);
int sdrx_group_get_output(sdrx_group *grp, int id, const void **buf, uint32_t *len_bytes, uint32_t *rate);
/* Where a VFO lives: *member = index into the device list (the owner of a leaf; the first replica of
 * a VFO with children), *local_id = its id inside that member's context; sdrx_group_member hands out
 * that context (NULL for a member that holds no VFOs) for sdrx_get_stream / sdrx_get_stats / kernel
 * timing. */
int sdrx_group_locate(sdrx_group *grp, int id, int *member, int *local_id);
int sdrx_group_member(sdrx_group *grp, int k, sdrx_ctx **ctx, int *device_ordinal);

/* ---- introspection / measurement ------------------------------------------------------------- */
typedef struct sdrx_stats {
    int32_t n_vfos, n_leaves, n_levels;
    int32_t exact;
    int64_t algorithmic_bytes_per_frame; /* sum of 8*n_in + W_out              */
    int64_t vfo_samples_per_frame;       /* sum of n_in over all VFOs                        */
    int64_t device_bytes;                /* HBM allocated by this context                    */
    int64_t frames;                      /* frames processed so far                          */
    int64_t mix_chunks_per_frame;        /* 1024-sample chunks the k_mix_decimate waves walk  */
                                         /*   per frame, warm-up chunks of segments included  */
    int64_t dc_blocks;                   /* exact DC-bias removal (sdrx_process_u8 .. correct_dc): 1024-sample blocks of one */
    int64_t dc_fallback_blocks;          /*   component walked so far / of those, redone with the sequential operations     */
    int64_t dc_retried_blocks;           /*   / taken again on their own because the step of several blocks they were part of */
                                         /*   did not verify as a whole (and then did: not counted as redone)                */
} sdrx_stats;
/* Everything but the three dc_* counters is host-side bookkeeping and costs nothing.  Once the context has run a frame with
 * correct_dc the dc_* counters live on the device: the call then WAITS for whatever is queued on the context's stream and
 * copies them back (a measurement call: a host that polls it between sdrx_submit and sdrx_wait serialises the frame it has
 * just queued).  They count what has EXECUTED, so they can lag `frames` by the frames still inside the launch pipeline
 * of sdrx_process_device (sdrx_sync / sdrx_fetch first for a consistent reading). */
int sdrx_get_stats(sdrx_ctx *ctx, sdrx_stats *out);
/* Per-kernel GPU time from HIP events recorded on the launch stream.  enable=1 brackets every
 * kernel launch with events (small overhead: use for profiling runs, not for throughput runs).
 * sdrx_get_kernel_times: accumulated milliseconds and launch counts since enabling, for
 * kernel kinds 0..SDRX_NKERNELS-1 (names from sdrx_kernel_name). */
#define SDRX_NKERNELS 8
int sdrx_enable_kernel_timing(sdrx_ctx *ctx, int enable);
int sdrx_get_kernel_times(sdrx_ctx *ctx, double ms[SDRX_NKERNELS], int64_t launches[SDRX_NKERNELS],
                          int64_t alg_bytes[SDRX_NKERNELS]);
const char *sdrx_kernel_name(int kind);

#ifdef __cplusplus
}
#endif
#endif /* SDRX_H */
