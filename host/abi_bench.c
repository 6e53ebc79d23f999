/* host/abi_bench.c -- what a C host sees through the ABI, PCIe both ways: BASELINE config 3 (the two
 * sdr_25E main VFOs + n sub VFOs by the config-3 rule of SURVEY.md 8d) fed from HOST buffers,
 *   (1) sdrx_process per frame (synchronous: H2D, kernels, D2H, callbacks in line), and
 *   (2) the pipelined pair  sdrx_submit(f+1); sdrx_wait() -> f  (frame f's payload copy beside f+1's kernels),
 * each on a pageable (malloc) input buffer.  Prints one JSON line.  C99, no HIP headers: plain ABI only.
 *   host/abi_bench [n_subs=1024] [frames=24] [device=0] */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "../include/sdrx.h"

static unsigned long n_messages, n_bytes;
static void on_publish(void *user, const char topic[5], uint32_t rate, const void *buf, uint32_t len)
{
    (void)user, (void)topic, (void)rate, (void)buf;
    ++n_messages;
    n_bytes += len;
}

static double now(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static int spread_mixer(int k, int K, int rate) /* topology._spread_mixer */
{
    return (int)nearbyint((k + 0.5 - K / 2.0) * 0.8 * rate / K) + 37;
}

#define CHECK(call)                                                                      \
    do {                                                                                 \
        int rc_ = (call);                                                                \
        if (rc_ != SDRX_OK) {                                                            \
            fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, sdrx_last_error(ctx));   \
            return 5;                                                                    \
        }                                                                                \
    } while (0)

int main(int argc, char **argv)
{
    const int n_subs = argc > 1 ? atoi(argv[1]) : 1024, frames = argc > 2 ? atoi(argv[2]) : 24, device = argc > 3 ? atoi(argv[3]) : 0;
    const int frame = 384000;
    sdrx_ctx *ctx = NULL;
    sdrx_vfo_desc d;
    int mains[2], i, k, f, rep;
    float *iq;
    uint32_t x = 1;
    double t_sync[3], t_pipe[3];
    if (sdrx_create(&ctx, device) != SDRX_OK) {
        fprintf(stderr, "sdrx_create failed: %s\n", sdrx_last_error(NULL));
        return 3;
    }
    for (i = 0; i < 2; ++i) { /* the sdr_25E main VFOs: +484 000 Hz d=2, -496 000 Hz d=3 */
        memset(&d, 0, sizeof d);
        d.fs = 1536000, d.decimate_count = i == 0 ? 2 : 3, d.mixer_freq_hz = i == 0 ? 484000.0 : -496000.0;
        d.cstyle = 1, d.scalecomp = 1, d.gain = 0.01f, d.parent_id = -1, d.samples_per_buffer = frame;
        CHECK(sdrx_add_vfo(ctx, &d, &mains[i]));
    }
    for (i = 0; i < 2; ++i) {
        const int K = i == 0 ? n_subs / 2 : n_subs - n_subs / 2, rate = i == 0 ? 384000 : 192000;
        for (k = 0; k < K; ++k) {
            memset(&d, 0, sizeof d);
            d.fs = rate, d.decimate_count = i == 0 ? 5 : 2, d.mixer_freq_hz = (double)spread_mixer(k, K, rate);
            d.demod_usb = 1, d.filter_bw_hz = (i == 1 && (k & 1)) ? 10000 : 0, d.gain = 0.05f, d.cstyle = 1, d.scalecomp = 1;
            d.parent_id = mains[i], d.samples_per_buffer = rate / 4;
            snprintf(d.topic, sizeof d.topic, "%c%04d", i == 0 ? 'A' : 'B', k % 10000);
            CHECK(sdrx_add_vfo(ctx, &d, NULL));
        }
    }
    CHECK(sdrx_set_publish_callback(ctx, on_publish, NULL));
    CHECK(sdrx_finalize(ctx));
    iq = (float *)malloc(sizeof(float) * 2 * (size_t)frame); /* pageable, like the reference's ring slots */
    for (i = 0; i < 2 * frame; ++i) {                           /* BASELINE.md's LCG */
        x = x * 1664525u + 1013904223u;
        iq[i] = (float)((int)((x >> 24) % 17u) - 8);
    }
    for (f = 0; f < 400; ++f) /* clocks up: ~50 ms of GPU time */
        CHECK(sdrx_process(ctx, iq, frame));
    for (rep = 0; rep < 3; ++rep) {
        double t0 = now();
        for (f = 0; f < frames; ++f)
            CHECK(sdrx_process(ctx, iq, frame));
        t_sync[rep] = (now() - t0) / frames;
        t0 = now();
        CHECK(sdrx_submit(ctx, iq, frame));
        for (f = 1; f < frames; ++f) {
            CHECK(sdrx_submit(ctx, iq, frame));
            CHECK(sdrx_wait(ctx));
        }
        CHECK(sdrx_wait(ctx));
        t_pipe[rep] = (now() - t0) / frames;
    }
    /* median of three */
    for (i = 0; i < 2; ++i)
        for (k = 0; k < 2 - i; ++k) {
            if (t_sync[k] > t_sync[k + 1]) { double t = t_sync[k]; t_sync[k] = t_sync[k + 1]; t_sync[k + 1] = t; }
            if (t_pipe[k] > t_pipe[k + 1]) { double t = t_pipe[k]; t_pipe[k] = t_pipe[k + 1]; t_pipe[k + 1] = t; }
        }
    printf("{\"host\": \"C99 over the ABI, pageable input, publish callback counting bytes\", \"n_subs\": %d, \"frames\": %d, "
           "\"sdrx_process_ms\": %.4f, \"sdrx_submit_wait_ms\": %.4f, \"messages_per_frame\": %lu, \"payload_bytes_per_frame\": %lu}\n",
           n_subs, frames, t_sync[1] * 1e3, t_pipe[1] * 1e3, n_messages / (unsigned long)(400 + 6 * frames),
           n_bytes / (unsigned long)(400 + 6 * frames));
    free(iq);
    sdrx_destroy(ctx);
    return 0;
}
