/* oracle/san_driver.c -- TEST INFRASTRUCTURE.  Runs the CPU restatement (vfo_oracle.c) under
 * AddressSanitizer + UndefinedBehaviorSanitizer on the three shapes of tree the tests rely on
 * (25E-like with the low-pass, 54W late decimation, a childless main with compress()), 3 frames
 * each, and prints a checksum per leaf.  GPU sanitizers are not available on the pool; the checker
 * at least is clean.  Built and run by tests/test_oracle_sanitizers.py:
 *   gcc -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all -ffp-contract=off \
 *       oracle/san_driver.c oracle/vfo_oracle.c -lm */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "vfo_oracle.h"

static orc_vfo *mk(int fs, int d, double mix, int usb, double bw, float gain, int spb, int late, const char *topic)
{
    orc_vfo *v = orc_vfo_new();
    orc_vfo_set_fs(v, fs);
    orc_vfo_set_decimation_count(v, d);
    orc_vfo_set_mixer_freq(v, mix);
    orc_vfo_set_demod_usb(v, usb);
    orc_vfo_set_filter_bandwidth(v, bw);
    orc_vfo_set_gain(v, gain);
    orc_vfo_set_compression_style(v, 1);
    orc_vfo_set_scale_comp(v, 1);
    orc_vfo_set_topic(v, topic);
    if (orc_vfo_init(v, spb, late) != 0) {
        fprintf(stderr, "init failed for %s\n", topic);
        exit(2);
    }
    return v;
}

static void report(const orc_vfo *v, const char *name, int usb_leaf)
{
    static short usb[65536];
    static signed char iq[2 * 65536];
    int n = usb_leaf ? orc_vfo_get_usb(v, usb, 65536) : 0, i;
    unsigned long sum = 0;
    if (usb_leaf)
        for (i = 0; i < n; ++i)
            sum = sum * 31u + (unsigned long)(long)usb[i];
    else {
        n = orc_vfo_get_iq(v, iq, 2 * 65536);
        for (i = 0; i < n; ++i)
            sum = sum * 31u + (unsigned long)(long)iq[i];
    }
    printf("%s n=%d sum=%lu\n", name, n, sum);
}

int main(void)
{
    const int frame25 = 384000, frame54 = 480000;
    float *iq = (float *)malloc(sizeof(float) * 2 * (size_t)frame54);
    float dc[2] = {0.f, 0.f};
    uint32_t x = 1;
    int f, i;
    /* 25E-like: main d=2 -> VFO01 (d=5, 4 kHz low-pass) and VFO19-like (fs 192 k is another main; here d=3 main) */
    orc_vfo *m0 = mk(1536000, 2, 484000.0, 0, 0, 0.01f, frame25, 0, "");
    orc_vfo *s0 = mk(384000, 5, 110854.0, 1, 4000.0, 0.05f, frame25 / 4, 0, "VFO01");
    orc_vfo *m1 = mk(1536000, 3, -496000.0, 0, 0, 0.01f, frame25, 0, "");
    orc_vfo *s1 = mk(192000, 2, -41300.0, 1, 10000.0, 0.03f, frame25 / 8, 0, "VFO19");
    orc_vfo *m2 = mk(1536000, 4, 12345.0, 0, 0, 0.01f, frame25, 0, "IQ00"); /* childless: compress() */
    /* 54W-like: main d=3 -> late-decimating subs (d=0 L=5, d=2 L=5) */
    orc_vfo *w0 = mk(1920000, 3, 819000.0, 0, 0, 0.01f, frame54, 0, "");
    orc_vfo *w1 = mk(240000, 0, 74578.0, 1, 10000.0, 0.04f, frame54 / 8, 5, "VFO51");
    orc_vfo *w2 = mk(240000, 2, 105571.0, 1, 0, 0.04f, frame54 / 8, 5, "VFO41");
    orc_vfo *roots25[3], *roots54[1];
    orc_vfo_add_child(m0, s0);
    orc_vfo_add_child(m1, s1);
    orc_vfo_add_child(w0, w1);
    orc_vfo_add_child(w0, w2);
    roots25[0] = m0, roots25[1] = m1, roots25[2] = m2;
    roots54[0] = w0;
    for (f = 0; f < 3; ++f) {
        for (i = 0; i < 2 * frame54; ++i) {
            x = x * 1664525u + 1013904223u;
            iq[i] = (float)((int)((x >> 24) % 17u) - 8);
        }
        orc_dc_correct(iq, frame25, dc);
        for (i = 0; i < 3; ++i)
            orc_vfo_process(roots25[i], iq, frame25);
        orc_vfo_process(roots54[0], iq, frame54);
    }
    report(s0, "VFO01", 1);
    report(s1, "VFO19", 1);
    report(m2, "IQ00", 0);
    report(w1, "VFO51", 1);
    report(w2, "VFO41", 1);
    orc_vfo_free(m0), orc_vfo_free(m1), orc_vfo_free(m2), orc_vfo_free(w0);
    free(iq);
    return 0;
}
