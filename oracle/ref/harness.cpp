// oracle/ref/harness.cpp -- TEST INFRASTRUCTURE, not product code.
//
// Thin extern "C" driver around the UNMODIFIED reference classes, compiled from the
// sources where they lie under /root/reference (see oracle/ref/Makefile; nothing is
// copied into this repository).  The resulting oracle/_ref/libsdrref.so is the real
// reference hot path (vfo.cpp, oscillator.cpp, halfbanddecimator.cpp, jonti/dsp.cpp,
// gnuradio/firfilter.cpp, zmqpublisher.cpp) linked against the real Qt5 and the real
// libzmq that ship in this image.  It is used to
//   * pin oracle/vfo_oracle.c (the plain-C restatement) sample for sample,
//   * generate the committed golden fixtures under tests/golden/,
//   * optionally serve as bench.py's cpu_baseline of kind "reference".
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
//
// The reference keeps its results in private members (vfo::transmit_usb, ...).  The
// harness reads them by compiling the reference *headers* with `private` spelled
// `public` -- the reference sources themselves are compiled untouched, so layout and
// behaviour are exactly upstream's.
#include <cassert>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <complex>
#include <complex.h>
#include <math.h>
#include <iostream>
#include <stdexcept>
#include <string>
#include <vector>
#include <QObject>
#include <QString>
#include <QVector>
#include <QSettings>
#include <QStringList>
#include <zmq.h>

#define private public
#define protected public
#include "vfo.h"
#include "gnuradio/firfilter.h"
#undef private
#undef protected

namespace {
std::string g_bind_address = "inproc://sdrref";
}

extern "C" {

// ---------------------------------------------------------------- vfo tree
void ref_set_bind_address(const char *addr) { g_bind_address = addr; }

void *ref_vfo_new() { return new vfo(); }
// Deleting a vfo deletes its children too (vfo.cpp:34-59): only free roots.
void ref_vfo_free(void *v) { delete static_cast<vfo *>(v); }

void ref_vfo_set_fs(void *v, int fs) { static_cast<vfo *>(v)->setFs(fs); }
void ref_vfo_set_decimation_count(void *v, int c) { static_cast<vfo *>(v)->setDecimationCount(c); }
void ref_vfo_set_mixer_freq(void *v, double f) { static_cast<vfo *>(v)->setMixerFreq(f); }
void ref_vfo_set_demod_usb(void *v, int usb) { static_cast<vfo *>(v)->setDemodUSB(usb != 0); }
void ref_vfo_set_filter_bandwidth(void *v, double bw) { static_cast<vfo *>(v)->setFilterBandwidth(bw); }
void ref_vfo_set_gain(void *v, float g) { static_cast<vfo *>(v)->setGain(g); }
void ref_vfo_set_compression_style(void *v, int st) { static_cast<vfo *>(v)->setCompressonStyle(st); }
void ref_vfo_set_scale_comp(void *v, int s) { static_cast<vfo *>(v)->setScaleComp(s); }
void ref_vfo_set_zmq_topic(void *v, const char *t) { static_cast<vfo *>(v)->setZmqTopic(QString::fromUtf8(t)); }
void ref_vfo_set_zmq_address(void *v, const char *a) { static_cast<vfo *>(v)->setZmqAddress(QString::fromUtf8(a)); }

// vfo::init (vfo.cpp:60-176).  A bound VFO without an address would pop a modal
// QMessageBox on bind failure (zmqpublisher.cpp:46-56); give it the harness default.
int ref_vfo_init(void *vv, int samples_per_buffer, int bind, int late_decimate)
{
    vfo *v = static_cast<vfo *>(vv);
    if (bind && v->zmqAddress.isEmpty())
        v->setZmqAddress(QString::fromStdString(g_bind_address));
    try {
        v->init(samples_per_buffer, bind != 0, late_decimate);
    } catch (const std::exception &) {
        return -1; // firfilter::sanity_check_1f threw (firfilter.cpp:122-134)
    }
    return 0;
}

// setVFOs (vfo.cpp:485-490): the parent keeps a pointer to a QVector it does not own.
void ref_vfo_add_child(void *parent, void *child)
{
    vfo *p = static_cast<vfo *>(parent);
    if (p->mpVFOs == 0)
        p->setVFOs(new QVector<vfo *>());
    p->mpVFOs->push_back(static_cast<vfo *>(child));
}

// One frame through vfo::process (vfo.cpp:235-296).  Building the
// std::vector<complex<float>> from interleaved floats is what sdrj::demodData does
// before calling process (sdrj.cpp:268-286, without the DC branch).
void ref_vfo_process(void *v, const float *iq, int n_complex)
{
    std::vector<cpx_typef> samples(n_complex);
    for (int i = 0; i < n_complex; ++i)
        samples[i] = cpx_typef(iq[2 * i], iq[2 * i + 1]);
    static_cast<vfo *>(v)->process(samples);
}

// Same, for several root VFOs in list order and `frames` repetitions of the same
// input: the loop of sdrj.cpp:288-294, used for timing.
void ref_process_roots(void **roots, int n_roots, const float *iq, int n_complex, int frames)
{
    std::vector<cpx_typef> samples(n_complex);
    for (int f = 0; f < frames; ++f) {
        for (int i = 0; i < n_complex; ++i)
            samples[i] = cpx_typef(iq[2 * i], iq[2 * i + 1]);
        for (int a = 0; a < n_roots; ++a)
            static_cast<vfo *>(roots[a])->process(samples);
    }
}

int ref_vfo_decimate_count(void *v) { return static_cast<vfo *>(v)->decimateCount; }
unsigned ref_vfo_output_rate(void *v) { return static_cast<vfo *>(v)->outputRate; }
int ref_vfo_get_out_rate(void *v) { return static_cast<vfo *>(v)->getOutRate(); }

// decimate[stage] (public member, vfo.h:39) as interleaved floats.
int ref_vfo_get_stream(void *vv, int stage, float *out, int max_complex)
{
    vfo *v = static_cast<vfo *>(vv);
    if (stage < 0 || stage > 8)
        return -1;
    const std::vector<cpx_typef> &s = v->decimate[stage];
    int n = (int)s.size() < max_complex ? (int)s.size() : max_complex;
    for (int i = 0; i < n; ++i) {
        out[2 * i] = s[i].real();
        out[2 * i + 1] = s[i].imag();
    }
    return (int)s.size();
}

int ref_vfo_get_usb(void *vv, short *out, int max)
{
    vfo *v = static_cast<vfo *>(vv);
    int n = (int)v->transmit_usb.size() < max ? (int)v->transmit_usb.size() : max;
    std::memcpy(out, v->transmit_usb.data(), n * sizeof(short));
    return (int)v->transmit_usb.size();
}

int ref_vfo_get_iq(void *vv, signed char *out, int max)
{
    vfo *v = static_cast<vfo *>(vv);
    int n = (int)v->transmit_iq.size() < max ? (int)v->transmit_iq.size() : max;
    std::memcpy(out, v->transmit_iq.data(), n);
    return (int)v->transmit_iq.size();
}

// Tap sets the reference designed inside vfo::init.
int ref_vfo_get_fir_usb_taps(void *vv, float *out, int max)
{
    vfo *v = static_cast<vfo *>(vv);
    if (!v->fir_usb)
        return 0;
    int n = v->fir_usb->NumberOfPoints;
    for (int i = 0; i < n && i < max; ++i)
        out[i] = v->fir_usb->points[i];
    return n;
}
int ref_vfo_get_fir_dec_taps(void *vv, float *out, int max)
{
    vfo *v = static_cast<vfo *>(vv);
    if (!v->fir_decI)
        return 0;
    int n = v->fir_decI->NumberOfPoints;
    for (int i = 0; i < n && i < max; ++i)
        out[i] = v->fir_decI->points[i];
    return n;
}
int ref_vfo_get_hilbert_taps(void *vv, float *out, int max)
{
    vfo *v = static_cast<vfo *>(vv);
    if (!v->philbert)
        return 0;
    int n = v->philbert->NumberOfPoints;
    for (int i = 0; i < n && i < max; ++i)
        out[i] = v->philbert->points[i];
    return n;
}

// ---------------------------------------------------------------- primitives
// Oscillator (oscillator.cpp:4-50): out[0] = _vector right after the ctor, then the
// value after each of n_ticks-1 ticks -- i.e. exactly the multiplier sequence the mix
// loop of vfo.cpp:237-245 sees for samples 0..n_ticks-1.
void ref_osc_sequence(double fs, double f, long n_ticks, float *out)
{
    Oscillator o(fs, f);
    for (long i = 0; i < n_ticks; ++i) {
        out[2 * i] = o._vector.real();
        out[2 * i + 1] = o._vector.imag();
        o.tick();
    }
}
// Raw table entries queue[first .. first+count) (private member, oscillator.h:20).
int ref_osc_table(double fs, double f, long first, long count, float *out)
{
    Oscillator o(fs, f);
    if (first < 0 || first + count > o.length)
        return -1;
    for (long i = 0; i < count; ++i) {
        out[2 * i] = o.queue[first + i].real();
        out[2 * i + 1] = o.queue[first + i].imag();
    }
    return o.length;
}

int ref_low_pass(double gain, double fs, double fc, double tw, float *out, int max)
{
    firfilter filt;
    try {
        QVector<float> t = filt.low_pass(gain, fs, fc, tw, firfilter::win_type::WIN_HAMMING, 0);
        for (int i = 0; i < t.length() && i < max; ++i)
            out[i] = t[i];
        return t.length();
    } catch (const std::exception &) {
        return -1;
    }
}

int ref_hilbert_taps(int len, int fs, float *out)
{
    FIRHilbert h(len, fs);
    for (int i = 0; i < len; ++i)
        out[i] = h.points[i];
    return len;
}

void *ref_halfband_new(int taps, int inlen) { return new HalfBandDecimator(taps, inlen); }
void ref_halfband_free(void *h) { delete static_cast<HalfBandDecimator *>(h); }
void ref_halfband_decimate(void *h, const float *in, int n_in, float *out)
{
    std::vector<cpx_typef> vin(n_in), vout(n_in / 2);
    for (int i = 0; i < n_in; ++i)
        vin[i] = cpx_typef(in[2 * i], in[2 * i + 1]);
    static_cast<HalfBandDecimator *>(h)->decimate(vin, vout);
    for (int i = 0; i < n_in / 2; ++i) {
        out[2 * i] = vout[i].real();
        out[2 * i + 1] = vout[i].imag();
    }
}

// FIR::FIRUpdateAndProcess / FIRUpdate (dsp.cpp:59-71,150-154) on a scalar stream:
// process[i] != 0 -> UpdateAndProcess (result stored), else FIRUpdate (out[i] = 0).
void ref_fir_run(const float *taps, int ntaps, const float *in, const unsigned char *process, int n, float *out)
{
    FIR f(ntaps, 0);
    for (int i = 0; i < ntaps; ++i)
        f.FIRSetPoint(i, taps[i]);
    for (int i = 0; i < n; ++i) {
        if (process == 0 || process[i])
            out[i] = f.FIRUpdateAndProcess(in[i]);
        else {
            f.FIRUpdate(in[i]);
            out[i] = 0.0f;
        }
    }
}

void ref_hilbert_run(int len, int fs, const float *in, int n, double *out)
{
    FIRHilbert h(len, fs);
    for (int i = 0; i < n; ++i)
        out[i] = h.FIRUpdateAndProcess(in[i]);
}

void ref_delay_run(int length, const float *in, int n, float *out)
{
    DelayThing<float> d;
    d.setLength(length);
    for (int i = 0; i < n; ++i)
        out[i] = d.update_dont_touch(in[i]);
}

// ---------------------------------------------------------------- ZMQ framing
// ZmqPublisher::publish (zmqpublisher.cpp:82-96) through the real libzmq: a PUB bound
// on `addr`, a SUB connected to it, one message published and received back as its
// three frames.  Returns number of frames received (3) or <0.
int ref_publish_roundtrip(const char *addr, const unsigned char *payload, unsigned len, const char *topic,
                          unsigned rate, unsigned char *f0, int *n0, unsigned char *f1, int *n1,
                          unsigned char *f2, int *n2, int max)
{
    ZmqPublisher pub;
    pub.setAddress(QString::fromUtf8(addr));
    pub.setBind(true);
    pub.connect();
    if (pub.zmqStatus < 0)
        return -2;
    void *ctx = zmq_ctx_new();
    void *sub = zmq_socket(ctx, ZMQ_SUB);
    int timeout = 200;
    zmq_setsockopt(sub, ZMQ_RCVTIMEO, &timeout, sizeof(timeout));
    zmq_setsockopt(sub, ZMQ_SUBSCRIBE, "", 0);
    if (zmq_connect(sub, addr) != 0)
        return -3;
    unsigned char *bufs[3] = {f0, f1, f2};
    int *lens[3] = {n0, n1, n2};
    int got = -4;
    // PUB/SUB joins asynchronously: keep publishing until the subscriber sees one.
    for (int attempt = 0; attempt < 50 && got < 0; ++attempt) {
        pub.publish(const_cast<unsigned char *>(payload), len, QString::fromUtf8(topic), rate);
        int r = zmq_recv(sub, bufs[0], max, 0);
        if (r < 0)
            continue;
        *lens[0] = r;
        got = 1;
        for (int k = 1; k < 3; ++k) {
            int more = 0;
            size_t sz = sizeof(more);
            zmq_getsockopt(sub, ZMQ_RCVMORE, &more, &sz);
            if (!more)
                break;
            r = zmq_recv(sub, bufs[k], max, 0);
            if (r < 0)
                break;
            *lens[k] = r;
            ++got;
        }
    }
    zmq_close(sub);
    zmq_ctx_term(ctx);
    return got;
}

// ---------------------------------------------------------------- INI parsing
// What QSettings(IniFormat) -- the parser MainWindow uses (mainwindow.cpp:27) -- makes of a
// profile: every key with its string value as "key=value\n".  Pins sdrreceiver_amd's INI parser.
int ref_qsettings_dump(const char *path, char *out, int max)
{
    QSettings settings(QString::fromUtf8(path), QSettings::IniFormat);
    QByteArray all;
    const QStringList keys = settings.allKeys();
    for (int i = 0; i < keys.size(); ++i) {
        all += keys.at(i).toUtf8();
        all += '=';
        all += settings.value(keys.at(i)).toString().toUtf8();
        all += '\n';
    }
    int n = all.size() < max - 1 ? all.size() : max - 1;
    std::memcpy(out, all.constData(), n);
    out[n] = 0;
    return all.size();
}

} // extern "C"
