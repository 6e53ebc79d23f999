/* oracle/vfo_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C, scalar, CPU restatement of the reference's per-VFO IQ chain (table-NCO mix ->
 * cascaded 11-tap half-band decimation -> USB demod by 62-sample delay minus 125-tap
 * Hilbert -> optional Hamming low-pass -> int16), written from the algorithm, not copied:
 * every function cites the reference file:line (relative to /root/reference) it follows.
 *
 * Parity status: PINNED.  tests/test_oracle_vs_reference.py checks this file sample for
 * sample (bit-exact) against oracle/_ref/libsdrref.so -- the reference's own sources
 * compiled unmodified in this container by oracle/ref/Makefile -- and
 * tests/test_oracle_golden.py checks it against the committed fixtures under
 * tests/golden/ that were generated from that reference build (tests/golden/make_golden.py).
 * The reference itself ships no tests, golden vectors or recorded IQ.
 *
 * Who may use it: tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, as the
 * checker / the timed CPU baseline only.  The product (sdrreceiver_amd/) never links,
 * imports or calls anything in oracle/.
 *
 * Build: gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC (oracle/Makefile).  The
 * reference arithmetic is IEEE fp32 with a few double sub-expressions; -ffp-contract=off
 * keeps every product and sum separately rounded like the -O2 x86-64 reference build.
 */
#include "vfo_oracle.h"

#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846264338327950288
#endif

#define HB_TAPS 11
#define HILBERT_LEN 125
#define DELAY_LEN ((HILBERT_LEN - 1) / 2) /* vfo.cpp:136 */
#define MAX_STAGES 8                     /* vfo.h:63 hdecimator[8] */

/* ======================================================================= primitives */

/* Oscillator::Oscillator, oscillator.cpp:4-32.  Table of (int)fs entries from a
 * sequential fp32 recurrence: v *= rot (complex product, re = ac-bd, im = ad+bc as
 * libstdc++/GCC expand it), then v *= 1.95f - |v|^2.  The rotation is cos/sin of a
 * double angle rounded to float. */
long orc_osc_table(double fs, double f, float *t)
{
    double angle = 2.0 * M_PI * f / fs;
    float rc = (float)cos(angle), rs = (float)sin(angle);
    float vr = 1.0f, vi = 0.0f;
    long len = (long)(int)fs;
    for (long i = 0; i < len; ++i) {
        float nr = vr * rc - vi * rs;
        float ni = vr * rs + vi * rc;
        float norm = 1.95f - (nr * nr + ni * ni);
        vr = nr * norm;
        vi = ni * norm;
        t[2 * i] = vr;
        t[2 * i + 1] = vi;
    }
    return len;
}

/* Which table entry multiplies sample k (k counted from start-up): after the ctor
 * _vector is the LAST entry and queuePtr is 0; tick() pre-increments
 * (oscillator.cpp:30,39-50), so k=0 -> L-1, k>=1 -> k mod L. */
static inline long osc_index(long long k, long L) { return k == 0 ? L - 1 : (long)(k % L); }

void orc_osc_sequence(double fs, double f, long n_ticks, float *out)
{
    long L = (long)(int)fs;
    float *t = (float *)malloc(sizeof(float) * 2 * (size_t)L);
    orc_osc_table(fs, f, t);
    for (long k = 0; k < n_ticks; ++k) {
        long i = osc_index(k, L);
        out[2 * k] = t[2 * i];
        out[2 * k + 1] = t[2 * i + 1];
    }
    free(t);
}

/* firfilter::low_pass with WIN_HAMMING, firfilter.cpp:64-106; compute_ntaps 108-119
 * (max_attenuation(HAMMING)=53, 141-171); hamming 212-220; sanity_check_1f 122-134
 * (the reference throws; here -1).  Window and taps are stored as float, the
 * sub-expressions are double, the DC-gain normalisation sums the float taps in double. */
int orc_low_pass(double gain, double fs, double fc, double tw, float *taps, int max)
{
    if (fs <= 0.0 || fc <= 0.0 || fc > fs / 2 || tw <= 0)
        return -1;
    int ntaps = (int)(53.0 * fs / (22.0 * tw));
    if ((ntaps & 1) == 0)
        ntaps++;
    if (ntaps > max)
        return -2;
    float *w = (float *)malloc(sizeof(float) * (size_t)ntaps);
    float Mw = (float)(ntaps - 1);
    for (int n = 0; n < ntaps; ++n)
        w[n] = (float)(0.54 - 0.46 * cos((2 * M_PI * n) / Mw));
    int M = (ntaps - 1) / 2;
    double fwT0 = 2 * M_PI * fc / fs;
    for (int n = -M; n <= M; ++n) {
        if (n == 0)
            taps[n + M] = (float)(fwT0 / M_PI * w[n + M]);
        else
            taps[n + M] = (float)(sin(n * fwT0) / (n * M_PI) * w[n + M]);
    }
    double fmax = taps[M];
    for (int n = 1; n <= M; ++n)
        fmax += 2 * taps[n + M]; /* int*float is a float product, then widened */
    gain /= fmax;
    for (int i = 0; i < ntaps; ++i)
        taps[i] = (float)(taps[i] * gain);
    free(w);
    return ntaps;
}

/* FIRHilbert::FIRHilbert, dsp.cpp:184-217: t[n] = Fs/(pi (n-c)) (1-cos(pi (n-c))),
 * float-stored, float sum of squares, reversed and divided by the square root of that sum --
 * taken in FLOAT: dsp.cpp is C++ with `using namespace std`, so `sqrt(float)` is the float
 * overload and only its result is widened to the `double gain`. */
void orc_hilbert_taps(int len, int fs, float *taps)
{
    float *tmp = (float *)malloc(sizeof(float) * (size_t)len);
    float sumsq = 0;
    for (int n = 0; n < len; ++n) {
        if (n == len / 2)
            tmp[n] = 0;
        else
            tmp[n] = (float)(fs / (M_PI * (n - len / 2)) * (1 - cos(M_PI * (n - len / 2))));
        sumsq += tmp[n] * tmp[n];
    }
    double g = (double)sqrtf(sumsq);
    for (int i = 0; i < len; ++i)
        taps[i] = (float)(tmp[len - i - 1] / g);
    free(tmp);
}

/* DC-bias removal on the shared raw stream, sdrj.cpp:277-283: a first-order IIR whose
 * accumulator is function-static there (lives for the whole process); here the caller
 * keeps state[2] = {re, im}, zero at start.  (sdrj.cpp is not buildable in this image --
 * it needs librtlsdr's header -- so these four lines are restated from the text only.) */
void orc_dc_correct(float *iq, int n, float state[2])
{
    const float keep = 1.0f - 0.000001f, k = 0.000001f;
    float ar = state[0], ai = state[1];
    for (int i = 0; i < n; ++i) {
        ar = ar * keep + k * iq[2 * i];
        ai = ai * keep + k * iq[2 * i + 1];
        iq[2 * i] -= ar;
        iq[2 * i + 1] -= ai;
    }
    state[0] = ar;
    state[1] = ai;
}

/* Dongle bytes -> float, jonti/sdr.cpp:43-49,122-129 and sdrj.cpp:155-160: b - 127. */
void orc_u8_to_float(const unsigned char *b, int n, float *out)
{
    for (int i = 0; i < n; ++i)
        out[i] = (float)((int)b[i] - 127);
}

/* `short = double` as the x86-64 reference build performs it (vfo.cpp:328,364): truncate
 * toward zero into int32 (cvttsd2si: out of range or NaN gives INT32_MIN), keep the low
 * 16 bits.  In-range values are plain truncation. */
short orc_double_to_short(double d)
{
    int32_t t = (d >= 2147483648.0 || d < -2147483648.0 || d != d) ? INT32_MIN : (int32_t)d;
    return (short)(uint16_t)(uint32_t)t;
}

/* `signed char = float` the same way (vfo.cpp:401-402,417-418). */
signed char orc_float_to_schar(float f)
{
    int32_t t = (f >= 2147483648.0f || f < -2147483648.0f || f != f) ? INT32_MIN : (int32_t)f;
    return (signed char)(uint8_t)(uint32_t)t;
}

/* ======================================================================= FIR pieces */

/* jonti FIR with an (N+1)-slot ring, dsp.cpp:32-50,59-71,150-154: after the newest sample
 * is pushed the sum runs over the N samples BEFORE it (oldest first), so the newest sample
 * is excluded.  Kept here as a 2x linear history so the dot product is one contiguous run. */
typedef struct {
    int n;        /* taps */
    float *taps;  /* [n] */
    float *hist;  /* [2*(n+1)] mirrored ring */
    int pos;      /* next write slot in [0, n+1) */
} orc_fir;

static void fir_init(orc_fir *f, int n, const float *taps)
{
    f->n = n;
    f->taps = (float *)malloc(sizeof(float) * (size_t)n);
    memcpy(f->taps, taps, sizeof(float) * (size_t)n);
    f->hist = (float *)calloc((size_t)(2 * (n + 1)), sizeof(float));
    f->pos = 0;
}
static void fir_free(orc_fir *f)
{
    free(f->taps);
    free(f->hist);
    f->taps = f->hist = 0;
}
static inline void fir_push(orc_fir *f, float x) /* FIR::FIRUpdate, dsp.cpp:150-154 */
{
    int r = f->n + 1;
    f->hist[f->pos] = x;
    f->hist[f->pos + r] = x;
    f->pos = f->pos + 1 == r ? 0 : f->pos + 1;
}
static inline float fir_push_process(orc_fir *f, float x) /* FIRUpdateAndProcess, dsp.cpp:59-71 */
{
    fir_push(f, x);
    const float *w = f->hist + f->pos; /* oldest of the ring; w[n] would be the newest */
    float acc = 0;
    for (int i = 0; i < f->n; ++i)
        acc += f->taps[i] * w[i];
    return acc;
}

/* FIRHilbert::FIRUpdateAndProcess, dsp.cpp:218-231: N-slot ring, newest sample included,
 * float accumulation, returned as double. */
typedef struct {
    float taps[HILBERT_LEN];
    float hist[2 * HILBERT_LEN];
    int pos;
} orc_hilbert;

static inline double hilbert_push_process(orc_hilbert *h, float x)
{
    h->hist[h->pos] = x;
    h->hist[h->pos + HILBERT_LEN] = x;
    h->pos = h->pos + 1 == HILBERT_LEN ? 0 : h->pos + 1;
    const float *w = h->hist + h->pos; /* w[0] oldest ... w[124] newest */
    float acc = 0;
    for (int i = 0; i < HILBERT_LEN; ++i)
        acc += h->taps[i] * w[i];
    return acc;
}

/* DelayThing<float>::update_dont_touch, dsp.h:87-94,101-106: setLength(62) makes a
 * 63-slot ring, output is the sample written 62 calls earlier. */
typedef struct {
    float buf[DELAY_LEN + 1];
    int pos;
} orc_delay;

static inline float delay_push(orc_delay *d, float x)
{
    d->buf[d->pos] = x;
    d->pos = (d->pos + 1) % (DELAY_LEN + 1);
    return d->buf[d->pos];
}

/* One 11-tap half-band /2 stage on a complex stream.
 * HalfBandDecimator::decimate, halfbanddecimator.cpp:43-72, on top of
 * FIR::FIRUpdateAndProcessHalfBandQueue (dsp.cpp:96-149, case 11: 137-143),
 * FIRUpdateQueue (156-160) and FIRQueueBackToFront (163-173); taps hbcoeff11,
 * halfbanddecimator.h:66-79.
 *
 * The reference keeps, per component, a linear queue [11 history | frame].  An output is
 * produced for every even frame index i from the 11 newest queue entries (symmetric-pair
 * form, evaluated left to right, then added to 0).  At the end of the frame the 11
 * entries ending ONE BEFORE the newest are copied to the front: the last sample of each
 * frame never becomes history and the history is shifted by one sample.  `hist` mirrors
 * queue[0..10]. */
static const float HB0 = 0.0060431029837374152f;
static const float HB2 = -0.049372515458761493f;
static const float HB4 = 0.29332944952052842f;
static const float HB5 = 0.5f;

typedef struct {
    float hist[HB_TAPS][2];
} orc_halfband;

static inline float hb_dot(float w0, float w2, float w4, float w5, float w6, float w8, float w10)
{
    float s = HB0 * (w0 + w10) + HB2 * (w2 + w8) + HB4 * (w4 + w6) + HB5 * w5;
    return 0.0f + s; /* `outsum = 0; outsum += ...` */
}

static void halfband_decimate(orc_halfband *hb, const float *in, int n_in, float *out, float *scratch)
{
    /* scratch: [(11 + n_in) * 2] = history followed by the frame, like the reference queue */
    float (*q)[2] = (float (*)[2])scratch;
    memcpy(q, hb->hist, sizeof(hb->hist));
    memcpy(q + HB_TAPS, in, sizeof(float) * 2 * (size_t)n_in);
    int step = 0;
    for (int i = 0; i < n_in; i += 2) {
        const float (*w)[2] = (const float (*)[2])(q + i + 1); /* w[10] is frame sample i */
        out[2 * step] = hb_dot(w[0][0], w[2][0], w[4][0], w[5][0], w[6][0], w[8][0], w[10][0]);
        out[2 * step + 1] = hb_dot(w[0][1], w[2][1], w[4][1], w[5][1], w[6][1], w[8][1], w[10][1]);
        ++step;
    }
    if (n_in > 0)
        memcpy(hb->hist, q + (n_in - 1), sizeof(hb->hist)); /* queue[(qp-1)-11 .. qp-1) */
}

/* ======================================================================= the VFO node */

struct orc_vfo {
    /* configuration, vfo.h:53-114 and the setters vfo.cpp:177-233,455-490 */
    int fs;
    int decimate_count;
    double mixer_freq;
    int demod_usb;
    int filterbw;
    float gain;
    int cstyle;
    int scalecomp;
    char topic[64];

    /* built by init */
    int samples_per_buffer;
    int late_decimate;
    int discard;
    unsigned output_rate;
    int samples_out;
    long osc_len;
    float *osc_table;
    long long sample_count;
    orc_halfband hb[MAX_STAGES];
    int has_fir_usb, has_fir_dec;
    orc_fir fir_usb, fir_dec_i, fir_dec_q;
    orc_hilbert hilbert;
    orc_delay delay;
    float *stream[MAX_STAGES + 1]; /* decimate[0..d], vfo.h:39 */
    int stream_len[MAX_STAGES + 1];
    float *scratch;
    short *transmit_usb;
    double *usb_prequant;
    signed char *transmit_iq;
    int transmit_iq_len;
    int published;

    orc_vfo **children;
    int n_children, cap_children;
};

orc_vfo *orc_vfo_new(void)
{
    orc_vfo *v = (orc_vfo *)calloc(1, sizeof(orc_vfo));
    v->gain = 0.01f; /* vfo.cpp:9 (the double literal 0.01 stored to float) */
    v->demod_usb = 1; /* vfo.cpp:15 */
    v->scalecomp = 1; /* vfo.cpp:24 */
    return v;
}

void orc_vfo_free(orc_vfo *v)
{
    if (!v)
        return;
    for (int i = 0; i < v->n_children; ++i)
        orc_vfo_free(v->children[i]);
    free(v->children);
    free(v->osc_table);
    for (int i = 0; i <= MAX_STAGES; ++i)
        free(v->stream[i]);
    free(v->scratch);
    free(v->transmit_usb);
    free(v->usb_prequant);
    free(v->transmit_iq);
    if (v->has_fir_usb)
        fir_free(&v->fir_usb);
    if (v->has_fir_dec) {
        fir_free(&v->fir_dec_i);
        fir_free(&v->fir_dec_q);
    }
    free(v);
}

void orc_vfo_set_fs(orc_vfo *v, int fs) { v->fs = fs; }
void orc_vfo_set_decimation_count(orc_vfo *v, int c) { v->decimate_count = c; }
void orc_vfo_set_mixer_freq(orc_vfo *v, double f) { v->mixer_freq = f; }
void orc_vfo_set_demod_usb(orc_vfo *v, int usb) { v->demod_usb = usb != 0; }
void orc_vfo_set_filter_bandwidth(orc_vfo *v, double bw) { v->filterbw = (int)bw; } /* int member, vfo.h:104 */
void orc_vfo_set_gain(orc_vfo *v, float g) { v->gain = g; }
void orc_vfo_set_compression_style(orc_vfo *v, int st) { v->cstyle = st; }
void orc_vfo_set_scale_comp(orc_vfo *v, int s) { v->scalecomp = s; }
void orc_vfo_set_topic(orc_vfo *v, const char *t)
{
    strncpy(v->topic, t, sizeof(v->topic) - 1);
    v->topic[sizeof(v->topic) - 1] = 0;
}

void orc_vfo_add_child(orc_vfo *p, orc_vfo *c)
{
    if (p->n_children == p->cap_children) {
        p->cap_children = p->cap_children ? 2 * p->cap_children : 8;
        p->children = (orc_vfo **)realloc(p->children, sizeof(orc_vfo *) * (size_t)p->cap_children);
    }
    p->children[p->n_children++] = c;
}

/* vfo::init, vfo.cpp:60-176 (the ZMQ bind/connect part, 160-172, is the boundary and not
 * restated).  Returns -1 where the reference would throw out of firfilter::low_pass. */
int orc_vfo_init(orc_vfo *v, int samples_per_buffer, int late_decimate)
{
    float taps[4096];
    int d = v->decimate_count;
    if (d < 0 || d > MAX_STAGES)
        return -3;
    v->samples_per_buffer = samples_per_buffer;
    v->osc_len = (long)(int)(double)v->fs;
    v->osc_table = (float *)malloc(sizeof(float) * 2 * (size_t)v->osc_len);
    orc_osc_table((double)v->fs, v->mixer_freq, v->osc_table);
    v->sample_count = 0;

    int target_rate = (int)(v->fs / pow(2, d));             /* vfo.cpp:66 */
    int samples_out = (int)(samples_per_buffer / pow(2, d)); /* vfo.cpp:67 */
    v->late_decimate = 0;
    if (v->demod_usb && late_decimate > 0) { /* vfo.cpp:70-101 */
        v->late_decimate = late_decimate;
        v->discard = late_decimate - 1;
        target_rate = target_rate / late_decimate;
        samples_out = samples_out / late_decimate;
        int n = orc_low_pass(2, target_rate * late_decimate, target_rate / 2,
                             (double)target_rate / (late_decimate - 1), taps, 4096);
        if (n < 0)
            return -1;
        fir_init(&v->fir_dec_i, n, taps);
        fir_init(&v->fir_dec_q, n, taps);
        v->has_fir_dec = 1;
    }
    v->output_rate = (unsigned)target_rate;
    v->samples_out = samples_out;
    if (v->filterbw > 0) { /* vfo.cpp:106-124 */
        int n = orc_low_pass(2, target_rate, v->filterbw, (double)v->filterbw / 4, taps, 4096);
        if (n < 0)
            return -1;
        fir_init(&v->fir_usb, n, taps);
        v->has_fir_usb = 1;
    }
    memset(v->hb, 0, sizeof(v->hb));                            /* vfo.cpp:127-133, dsp.cpp:40-49 */
    memset(&v->delay, 0, sizeof(v->delay));                     /* vfo.cpp:136 */
    memset(&v->hilbert, 0, sizeof(v->hilbert));
    orc_hilbert_taps(HILBERT_LEN, samples_out, v->hilbert.taps); /* vfo.cpp:137: "Fs" = samplesOut */

    v->transmit_usb = (short *)calloc((size_t)(samples_out > 0 ? samples_out : 1), sizeof(short));
    v->usb_prequant = (double *)calloc((size_t)(samples_out > 0 ? samples_out : 1), sizeof(double));
    v->transmit_iq_len = v->cstyle == 1 ? samples_out : 2 * samples_out; /* vfo.cpp:143-150 */
    v->transmit_iq = (signed char *)calloc((size_t)(v->transmit_iq_len > 0 ? v->transmit_iq_len : 1), 1);
    v->stream_len[0] = samples_per_buffer; /* vfo.cpp:152-158 */
    for (int a = 1; a <= d; ++a)
        v->stream_len[a] = v->stream_len[a - 1] / 2;
    for (int a = 0; a <= d; ++a)
        v->stream[a] = (float *)calloc((size_t)(2 * v->stream_len[a] + 2), sizeof(float));
    v->scratch = (float *)malloc(sizeof(float) * 2 * (size_t)(samples_per_buffer + HB_TAPS));
    return 0;
}

/* vfo::usb_demod, vfo.cpp:300-332 (the osc_bfo branch, 307-312, is dead: offsetbw is
 * never set). */
static void usb_demod(orc_vfo *v)
{
    const float *z = v->stream[v->decimate_count];
    int n = v->stream_len[v->decimate_count];
    for (int i = 0; i < n; ++i) {
        double diff = (double)delay_push(&v->delay, z[2 * i]) - hilbert_push_process(&v->hilbert, z[2 * i + 1]);
        float usb = (float)diff;
        if (v->filterbw > 0)
            usb = fir_push_process(&v->fir_usb, usb);
        double pre = usb * v->gain * 32768.0; /* (float*float) then *double, vfo.cpp:328 */
        v->usb_prequant[i] = pre;
        v->transmit_usb[i] = orc_double_to_short(pre);
    }
}

/* vfo::usb_decimdemod, vfo.cpp:334-387: the phase counter restarts at every frame; one
 * sample in L goes through the late-decimation low-pass (newest excluded) and on to the
 * demodulator, the others are only pushed; the audio low-pass comes after the demod. */
static void usb_decimdemod(orc_vfo *v)
{
    const float *z = v->stream[v->decimate_count];
    int n = v->stream_len[v->decimate_count];
    int mark = 0, check = 0;
    for (int i = 0; i < n; ++i) {
        float re = z[2 * i], im = z[2 * i + 1];
        if (check == 0) {
            float fr = fir_push_process(&v->fir_dec_i, re);
            float fi = fir_push_process(&v->fir_dec_q, im);
            float usb = (float)((double)delay_push(&v->delay, fr) - hilbert_push_process(&v->hilbert, fi));
            if (v->filterbw > 0)
                usb = fir_push_process(&v->fir_usb, usb);
            double pre = usb * v->gain * 32768.0;
            if (mark < v->samples_out) {
                v->usb_prequant[mark] = pre;
                v->transmit_usb[mark] = orc_double_to_short(pre);
            }
            mark++;
            check++;
        } else if (check == v->discard) {
            fir_push(&v->fir_dec_i, re);
            fir_push(&v->fir_dec_q, im);
            check = 0;
        } else {
            fir_push(&v->fir_dec_i, re);
            fir_push(&v->fir_dec_q, im);
            check++;
        }
    }
}

/* vfo::compress, vfo.cpp:389-424. */
static void compress_iq(orc_vfo *v)
{
    const float *z = v->stream[v->decimate_count];
    int n = v->stream_len[v->decimate_count];
    if (v->cstyle == 1) {
        for (int i = 0; i < n && i < v->transmit_iq_len; ++i) {
            signed char re = orc_float_to_schar((z[2 * i] / v->scalecomp) * 128);
            signed char im = orc_float_to_schar((z[2 * i + 1] / v->scalecomp) * 128);
            v->transmit_iq[i] = (signed char)((re & 0xF0) | ((im & 0xF0) >> 4));
        }
    } else {
        for (int i = 0; i < n && 2 * i + 1 < v->transmit_iq_len; ++i) {
            v->transmit_iq[2 * i] = orc_float_to_schar(z[2 * i] * 128);
            v->transmit_iq[2 * i + 1] = orc_float_to_schar(z[2 * i + 1] * 128);
        }
    }
}

/* The part of vfo::process that belongs to this node alone: mix loop (vfo.cpp:237-245)
 * and the decimation cascade (247-251). */
static void mix_and_decimate(orc_vfo *v, const float *iq, int n)
{
    float *x = v->stream[0];
    const float *t = v->osc_table;
    long L = v->osc_len;
    long long k = v->sample_count;
    for (int i = 0; i < n; ++i, ++k) {
        long j = osc_index(k, L);
        float a = t[2 * j], b = t[2 * j + 1], c = iq[2 * i], d = iq[2 * i + 1];
        x[2 * i] = a * c - b * d;
        x[2 * i + 1] = a * d + b * c;
    }
    v->sample_count = k;
    for (int s = 0; s < v->decimate_count; ++s)
        halfband_decimate(&v->hb[s], v->stream[s], v->stream_len[s], v->stream[s + 1], v->scratch);
}

static void leaf_tail(orc_vfo *v) /* vfo.cpp:267-287 */
{
    if (v->demod_usb) {
        if (!v->late_decimate)
            usb_demod(v);
        else
            usb_decimdemod(v);
    } else {
        compress_iq(v);
    }
    v->published = 1; /* transmitData, vfo.cpp:426-453 -> orc_vfo_get_publish */
}

/* vfo::process, vfo.cpp:235-296.  n must equal samples_per_buffer (the reference
 * indexes decimate[0] without a bound check). */
void orc_vfo_process(orc_vfo *v, const float *iq, int n)
{
    if (n > v->samples_per_buffer)
        n = v->samples_per_buffer;
    mix_and_decimate(v, iq, n);
    if (v->n_children > 0) {
        for (int a = 0; a < v->n_children; ++a)
            orc_vfo_process(v->children[a], v->stream[v->decimate_count], v->stream_len[v->decimate_count]);
    } else {
        leaf_tail(v);
    }
}

/* The loop of sdrj::demodData over the main VFOs, sdrj.cpp:288-294, `frames` times over the
 * same input.  threads<=1: serial, exactly the reference's single-thread order.
 * threads>1: children of each root are processed with OpenMP -- legal because no VFO reads
 * another VFO's state (vfo.cpp:253-264) -- as the "all host cores" CPU baseline. */
void orc_process_roots(orc_vfo **roots, int n_roots, const float *iq, int n, int frames, int threads)
{
    if (threads <= 1) {
        for (int f = 0; f < frames; ++f)
            for (int r = 0; r < n_roots; ++r)
                orc_vfo_process(roots[r], iq, n);
        return;
    }
    /* VFOs do not see each other (vfo.cpp:253-264: every child is handed the same read-only buffer), so the order the
     * reference walks them in -- sdrj.cpp:288-294 over the mains, each main over its children -- is free: per frame first
     * every root (a root with children: its own mix + cascade only), then all children of all roots, both in parallel. */
    int n_kids = 0;
    for (int r = 0; r < n_roots; ++r)
        n_kids += roots[r]->n_children;
    orc_vfo **kid = (orc_vfo **)malloc(sizeof(orc_vfo *) * (size_t)(n_kids ? n_kids : 1));
    int *kid_root = (int *)malloc(sizeof(int) * (size_t)(n_kids ? n_kids : 1));
    for (int r = 0, k = 0; r < n_roots; ++r)
        for (int a = 0; a < roots[r]->n_children; ++a, ++k) {
            kid[k] = roots[r]->children[a];
            kid_root[k] = r;
        }
    for (int f = 0; f < frames; ++f) {
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
        for (int r = 0; r < n_roots; ++r) {
            orc_vfo *v = roots[r];
            if (v->n_children == 0)
                orc_vfo_process(v, iq, n);
            else
                mix_and_decimate(v, iq, n < v->samples_per_buffer ? n : v->samples_per_buffer);
        }
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
        for (int k = 0; k < n_kids; ++k) {
            const orc_vfo *v = roots[kid_root[k]];
            orc_vfo_process(kid[k], v->stream[v->decimate_count], v->stream_len[v->decimate_count]);
        }
    }
    free(kid);
    free(kid_root);
}

/* ======================================================================= accessors */
int orc_vfo_decimate_count(const orc_vfo *v) { return v->decimate_count; }
unsigned orc_vfo_output_rate(const orc_vfo *v) { return v->output_rate; }

int orc_vfo_get_stream(const orc_vfo *v, int stage, float *out, int max_complex)
{
    if (stage < 0 || stage > v->decimate_count)
        return -1;
    int n = v->stream_len[stage] < max_complex ? v->stream_len[stage] : max_complex;
    memcpy(out, v->stream[stage], sizeof(float) * 2 * (size_t)n);
    return v->stream_len[stage];
}
int orc_vfo_get_usb(const orc_vfo *v, short *out, int max)
{
    int n = v->samples_out < max ? v->samples_out : max;
    memcpy(out, v->transmit_usb, sizeof(short) * (size_t)n);
    return v->samples_out;
}
int orc_vfo_get_usb_prequant(const orc_vfo *v, double *out, int max)
{
    int n = v->samples_out < max ? v->samples_out : max;
    memcpy(out, v->usb_prequant, sizeof(double) * (size_t)n);
    return v->samples_out;
}
int orc_vfo_get_iq(const orc_vfo *v, signed char *out, int max)
{
    int n = v->transmit_iq_len < max ? v->transmit_iq_len : max;
    memcpy(out, v->transmit_iq, (size_t)n);
    return v->transmit_iq_len;
}
static int copy_taps(const float *t, int n, float *out, int max)
{
    memcpy(out, t, sizeof(float) * (size_t)(n < max ? n : max));
    return n;
}
int orc_vfo_get_fir_usb_taps(const orc_vfo *v, float *out, int max)
{
    return v->has_fir_usb ? copy_taps(v->fir_usb.taps, v->fir_usb.n, out, max) : 0;
}
int orc_vfo_get_fir_dec_taps(const orc_vfo *v, float *out, int max)
{
    return v->has_fir_dec ? copy_taps(v->fir_dec_i.taps, v->fir_dec_i.n, out, max) : 0;
}
int orc_vfo_get_hilbert_taps(const orc_vfo *v, float *out, int max)
{
    return copy_taps(v->hilbert.taps, HILBERT_LEN, out, max);
}

/* What vfo::transmitData hands to ZmqPublisher::publish (vfo.cpp:426-453,
 * zmqpublisher.cpp:82-96): frame 1 = exactly 5 topic bytes, frame 2 = uint32 rate in
 * native byte order, frame 3 = payload; nothing is sent for an empty payload, and a
 * non-USB VFO only publishes when it has a topic.  Returns 1 if a message would be sent. */
int orc_vfo_get_publish(const orc_vfo *v, char topic5[5], unsigned *rate, const unsigned char **payload,
                        unsigned *len)
{
    if (!v->published || v->n_children > 0)
        return 0;
    memset(topic5, 0, 5);
    memcpy(topic5, v->topic, strlen(v->topic) < 5 ? strlen(v->topic) : 5);
    *rate = v->output_rate;
    if (v->demod_usb) {
        *payload = (const unsigned char *)v->transmit_usb;
        *len = (unsigned)(v->samples_out * (int)sizeof(short));
    } else {
        if (strlen(v->topic) == 0)
            return 0;
        *payload = (const unsigned char *)v->transmit_iq;
        *len = (unsigned)v->transmit_iq_len;
    }
    return *len != 0;
}
