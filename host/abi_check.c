/* host/abi_check.c -- include/sdrx.h is a plain C header: this file is compiled as C99 with
 * -pedantic and linked against libsdrx.so.  Run without a GPU it must fail LOUDLY at sdrx_create
 * (there is no CPU fallback); with one it builds the sdr_25E.ini VFO01 chain and pushes one frame. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/sdrx.h"

static int n_messages;
static void on_publish(void *user, const char topic[5], uint32_t rate, const void *buf, uint32_t len)
{
    (void)user;
    (void)buf;
    printf("published %.5s rate %u bytes %u\n", topic, (unsigned)rate, (unsigned)len);
    ++n_messages;
}

int main(void)
{
    sdrx_ctx *ctx = NULL;
    sdrx_vfo_desc main_vfo, sub;
    int id_main = -1, id_sub = -1, rc, i;
    float *iq;
    printf("sdrx ABI version %d, sizeof(sdrx_vfo_desc) = %u\n", sdrx_abi_version(), (unsigned)sizeof(sdrx_vfo_desc));
    rc = sdrx_create(&ctx, 0);
    if (rc != SDRX_OK) {
        fprintf(stderr, "sdrx_create failed (%d): %s\n", rc, sdrx_last_error(NULL));
        return 3;
    }
    memset(&main_vfo, 0, sizeof main_vfo);
    main_vfo.fs = 1536000, main_vfo.decimate_count = 2, main_vfo.mixer_freq_hz = 484000.0, main_vfo.parent_id = -1;
    main_vfo.samples_per_buffer = 384000, main_vfo.scalecomp = 1, main_vfo.gain = 0.01f;
    memset(&sub, 0, sizeof sub);
    sub.fs = 384000, sub.decimate_count = 5, sub.mixer_freq_hz = 110854.0, sub.demod_usb = 1, sub.filter_bw_hz = 4000;
    sub.gain = 0.05f, sub.scalecomp = 1, sub.samples_per_buffer = 96000;
    memcpy(sub.topic, "VFO01", 5);
    if ((rc = sdrx_add_vfo(ctx, &main_vfo, &id_main)) != SDRX_OK)
        goto fail;
    sub.parent_id = id_main;
    if ((rc = sdrx_add_vfo(ctx, &sub, &id_sub)) != SDRX_OK || (rc = sdrx_set_publish_callback(ctx, on_publish, NULL)) != SDRX_OK ||
        (rc = sdrx_finalize(ctx)) != SDRX_OK)
        goto fail;
    iq = (float *)calloc(2 * 384000, sizeof(float));
    for (i = 0; i < 2 * 384000; ++i)
        iq[i] = (float)((i * 7) % 17 - 8);
    rc = sdrx_process(ctx, iq, 384000);
    free(iq);
    if (rc != SDRX_OK)
        goto fail;
    sdrx_destroy(ctx);
    ctx = NULL;
    if (n_messages != 1)
        return 4;
    /* the same chain on a device LIST (two shards on the one GPU), pipelined: submit(f+1); wait() -> f */
    {
        sdrx_group *grp = NULL;
        const int devices[2] = {0, 0};
        int f, rc2 = sdrx_group_create(&grp, devices, 2);
        if (rc2 == SDRX_OK)
            rc2 = sdrx_group_add_vfo(grp, &main_vfo, &id_main);
        sub.parent_id = id_main;
        if (rc2 == SDRX_OK)
            rc2 = sdrx_group_add_vfo(grp, &sub, &id_sub);
        if (rc2 == SDRX_OK)
            rc2 = sdrx_group_set_publish_callback(grp, on_publish, NULL);
        if (rc2 == SDRX_OK)
            rc2 = sdrx_group_finalize(grp);
        iq = (float *)calloc(2 * 384000, sizeof(float));
        for (f = 0; f < 3 && rc2 == SDRX_OK; ++f) {
            rc2 = sdrx_group_submit(grp, iq, 384000);
            if (rc2 == SDRX_OK && f > 0)
                rc2 = sdrx_group_wait(grp);
        }
        if (rc2 == SDRX_OK)
            rc2 = sdrx_group_wait(grp);
        free(iq);
        if (rc2 != SDRX_OK) {
            fprintf(stderr, "sdrx_group error %d: %s\n", rc2, sdrx_group_last_error(grp));
            sdrx_group_destroy(grp);
            return 6;
        }
        sdrx_group_destroy(grp);
    }
    return n_messages == 4 ? 0 : 7;
fail:
    fprintf(stderr, "sdrx error %d: %s\n", rc, sdrx_last_error(ctx));
    sdrx_destroy(ctx);
    return 5;
}
