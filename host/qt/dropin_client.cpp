// host/qt/dropin_client.cpp -- TEST INFRASTRUCTURE for the drop-in claim.
//
// A client of `class vfo` that uses ONLY the public interface of the reference's vfo.h
// (vfo.h:16-49), the way MainWindow (mainwindow.cpp:98-233) and sdrj::demodData
// (sdrj.cpp:288-294) do: setters, init(samplesPerBuffer, bind, lateDecimate), setVFOs,
// process(samples) on every main VFO, the fftVFOSlot slot and the fftData signal.  What the
// receiver publishes is observed where a real subscriber sees it: on a ZMQ SUB socket.
//
// The same file is linked twice (host/qt/Makefile):
//   oracle/_ref/libdropin_ref.so   with the reference's own vfo.cpp / DSP sources
//   oracle/_ref/libdropin_sdrx.so  with host/qt/vfo_adapter.cpp over libsdrx.so (the GPU)
// Same client, same header, same ZmqPublisher: the message streams must be byte-identical.
#include <QVector>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include <time.h>
#include <unistd.h>

#include "vfo.h"
#include "zmq.h"

#include "../../include/sdrx.h" // only for the plain description struct sdrx_vfo_desc

namespace {
void put32(std::vector<unsigned char> &o, uint32_t v)
{
    unsigned char b[4];
    std::memcpy(b, &v, 4);
    o.insert(o.end(), b, b + 4);
}
uint64_t fnv1a(const void *p, size_t n)
{
    const unsigned char *b = static_cast<const unsigned char *>(p);
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; ++i) {
        h ^= b[i];
        h *= 1099511628211ull;
    }
    return h;
}
} // namespace

extern "C" {

// Build the tree described by descs[0..n) (parents before children, parent_id = index or -1),
// run `frames` frames of the BASELINE LCG signal through every main VFO and return everything a
// subscriber of `addr` received:  for every message  u32 nparts, then per part u32 len + bytes.
// fft_topic (may be empty) is handed to every vfo's fftVFOSlot; each fftData emission is logged
// to fft_out as "frame topic count fnv1a64\n".  Returns the number of bytes written to out, or
// < 0 on error.
// `copies` > 1 builds that many independent receivers from the same description (MainWindow builds
// one; nothing in vfo.h forbids more) and feeds every frame to each in turn: the subscriber then sees
// the same messages `copies` times per frame.
int dropin_run(const sdrx_vfo_desc *descs, int n, const char *addr, int frames, const char *fft_topic, unsigned char *out, int cap,
               char *fft_out, int fft_cap, int copies)
{
    if (copies < 1)
        copies = 1;
    const int n1 = n;
    n *= copies;
    std::vector<vfo *> nodes((size_t)n);
    std::vector<QVector<vfo *> *> kids((size_t)n, nullptr);
    QVector<vfo *> mains;
    std::string fftlog;
    int frame_no = 0;
    int root_frame = 0;
    for (int i = 0; i < n; ++i) {
        sdrx_vfo_desc d = descs[i % n1];
        if (d.parent_id >= 0)
            d.parent_id += (i / n1) * n1;
        vfo *v = new vfo();
        v->setZmqAddress(QString::fromUtf8(addr));
        v->setZmqTopic(QString::fromLatin1(d.topic));
        v->setFs(d.fs);
        v->setDecimationCount(d.decimate_count);
        v->setMixerFreq(d.mixer_freq_hz);
        v->setDemodUSB(d.demod_usb != 0);
        v->setFilterBandwidth(d.filter_bw_hz);
        v->setGain(d.gain);
        v->setCompressonStyle(d.cstyle);
        v->setScaleComp(d.scalecomp);
        v->init(d.samples_per_buffer, true, d.late_decimate);
        nodes[(size_t)i] = v;
        if (d.parent_id < 0) {
            mains.append(v);
            root_frame = d.samples_per_buffer;
        } else {
            if (!kids[(size_t)d.parent_id])
                kids[(size_t)d.parent_id] = new QVector<vfo *>();
            kids[(size_t)d.parent_id]->append(v);
        }
        const std::string topic = d.topic;
        QObject::connect(v, &vfo::fftData, [&fftlog, &frame_no, topic](const std::vector<cpx_typef> &data) {
            char line[128];
            snprintf(line, sizeof line, "%d %s %zu %016llx\n", frame_no, topic.c_str(), data.size(),
                     (unsigned long long)fnv1a(data.data(), data.size() * sizeof(cpx_typef)));
            fftlog += line;
        });
    }
    for (int i = 0; i < n; ++i)
        if (kids[(size_t)i])
            nodes[(size_t)i]->setVFOs(kids[(size_t)i]);
    if (fft_topic && fft_topic[0])
        for (vfo *v : nodes)
            v->fftVFOSlot(QString::fromLatin1(fft_topic));

    // the subscriber; PUB/SUB joins asynchronously, so give the connection time before frame 0
    void *zctx = zmq_ctx_new();
    void *sub = zmq_socket(zctx, ZMQ_SUB);
    int hwm = 0, timeout = 300;
    zmq_setsockopt(sub, ZMQ_RCVHWM, &hwm, sizeof hwm);
    zmq_setsockopt(sub, ZMQ_RCVTIMEO, &timeout, sizeof timeout);
    zmq_setsockopt(sub, ZMQ_SUBSCRIBE, "", 0);
    if (zmq_connect(sub, addr) != 0)
        return -2;
    usleep(500 * 1000);

    std::vector<unsigned char> all;
    std::vector<cpx_typef> samples((size_t)root_frame);
    std::vector<unsigned char> part(1 << 20);
    // what the subscriber has received so far, in order; gives up after `first_ms` of silence (20 ms
    // once a message of the burst has arrived: the rest of the frame's messages are already queued)
    auto drain = [&](int first_ms) {
        int timeout = first_ms;
        zmq_setsockopt(sub, ZMQ_RCVTIMEO, &timeout, sizeof timeout);
        for (;;) {
            int r = zmq_recv(sub, part.data(), part.size(), 0);
            if (r < 0)
                break;
            std::vector<std::vector<unsigned char>> parts;
            parts.emplace_back(part.begin(), part.begin() + r);
            for (;;) {
                int more = 0;
                size_t sz = sizeof more;
                zmq_getsockopt(sub, ZMQ_RCVMORE, &more, &sz);
                if (!more)
                    break;
                r = zmq_recv(sub, part.data(), part.size(), 0);
                if (r < 0)
                    break;
                parts.emplace_back(part.begin(), part.begin() + r);
            }
            put32(all, (uint32_t)parts.size());
            for (auto &p : parts) {
                put32(all, (uint32_t)p.size());
                all.insert(all.end(), p.begin(), p.end());
            }
            timeout = 20;
            zmq_setsockopt(sub, ZMQ_RCVTIMEO, &timeout, sizeof timeout);
        }
    };
    uint32_t x = 1; // BASELINE.md's LCG: x <- x*1664525 + 1013904223, component ((x >> 24) % 17) - 8
    for (frame_no = 0; frame_no < frames; ++frame_no) {
        for (auto &s : samples) {
            x = x * 1664525u + 1013904223u;
            const float re = (float)((int)((x >> 24) % 17u) - 8);
            x = x * 1664525u + 1013904223u;
            const float im = (float)((int)((x >> 24) % 17u) - 8);
            s = cpx_typef(re, im);
        }
        // DROPIN_MUTATE (test switch): bit 0 -- the samples CHANGE between two main VFOs' process() calls (same vector, same
        // address; every 97th sample from the 7th on, sparing the 64 positions a spot check of the frame would look at);
        // bit 1 -- on odd frames the first main VFO is not called at all.  Whatever `class vfo` sits behind the header must
        // process what it is HANDED: an implementation that reuses an earlier upload of "the same" vector would not.
        const int mutate = std::getenv("DROPIN_MUTATE") ? std::atoi(std::getenv("DROPIN_MUTATE")) : 0;
        for (int mi = 0; mi < mains.size(); ++mi) { // sdrj.cpp:288-294
            if ((mutate & 2) && mi == 0 && (frame_no & 1) && mains.size() > 1)
                continue;
            mains[mi]->process(samples);
            if (mutate & 1) {
                const size_t nf = 2 * samples.size();
                for (size_t i = 7 + (size_t)mi; i < samples.size(); i += 97) {
                    bool probe = false;
                    for (size_t k = 0; k < 64 && !probe; ++k)
                        probe = (k * (nf - 1) / 63) / 2 == i;
                    if (!probe)
                        samples[i] = cpx_typef(1.0f - samples[i].real(), samples[i].imag() + 2.0f);
                }
            }
        }
        drain(300); // everything published for this frame (the first recv waits up to the timeout)
    }
    // MainWindow's stop: a vfo owns its children (vfo.cpp:49-57).  Before the subscriber goes: an implementation that
    // delivers a frame late (the adapter with SDRX_PIPELINE=1) hands over what it still owes when its tree is deleted;
    // the reference publishes nothing here.
    for (vfo *m : mains)
        delete m;
    drain(1500); // messages still on their way when the last frame's drain gave up (a loaded host)
    zmq_close(sub);
    zmq_ctx_term(zctx);
    if ((int)all.size() > cap)
        return -3;
    std::memcpy(out, all.data(), all.size());
    if (fft_out && fft_cap > 0) {
        const size_t k = std::min(fftlog.size(), (size_t)fft_cap - 1);
        std::memcpy(fft_out, fftlog.data(), k);
        fft_out[k] = 0;
    }
    return (int)all.size();
}

// Wall time per frame of the loop `for every main VFO: process(samples)` (sdrj.cpp:288-294) through the public
// interface of vfo.h -- setters, init, setVFOs, process -- with whatever implementation of `class vfo` is linked
// behind it: what a Qt host pays per frame, transmitData / ZmqPublisher::publish of every leaf included (a PUB
// socket without subscribers drops the messages in libzmq).  `warm` untimed frames, then `frames` timed ones.
// Returns milliseconds per frame, < 0 on error.
double dropin_time(const sdrx_vfo_desc *descs, int n, const char *addr, int warm, int frames)
{
    std::vector<vfo *> nodes((size_t)n);
    std::vector<QVector<vfo *> *> kids((size_t)n, nullptr);
    QVector<vfo *> mains;
    int root_frame = 0;
    for (int i = 0; i < n; ++i) {
        const sdrx_vfo_desc &d = descs[i];
        vfo *v = new vfo();
        v->setZmqAddress(QString::fromUtf8(addr));
        v->setZmqTopic(QString::fromLatin1(d.topic));
        v->setFs(d.fs);
        v->setDecimationCount(d.decimate_count);
        v->setMixerFreq(d.mixer_freq_hz);
        v->setDemodUSB(d.demod_usb != 0);
        v->setFilterBandwidth(d.filter_bw_hz);
        v->setGain(d.gain);
        v->setCompressonStyle(d.cstyle);
        v->setScaleComp(d.scalecomp);
        v->init(d.samples_per_buffer, true, d.late_decimate);
        nodes[(size_t)i] = v;
        if (d.parent_id < 0) {
            mains.append(v);
            root_frame = d.samples_per_buffer;
        } else {
            if (!kids[(size_t)d.parent_id])
                kids[(size_t)d.parent_id] = new QVector<vfo *>();
            kids[(size_t)d.parent_id]->append(v);
        }
    }
    for (int i = 0; i < n; ++i)
        if (kids[(size_t)i])
            nodes[(size_t)i]->setVFOs(kids[(size_t)i]);
    std::vector<cpx_typef> samples((size_t)root_frame);
    uint32_t x = 1;
    for (auto &s : samples) {
        x = x * 1664525u + 1013904223u;
        const float re = (float)((int)((x >> 24) % 17u) - 8);
        x = x * 1664525u + 1013904223u;
        s = cpx_typef(re, (float)((int)((x >> 24) % 17u) - 8));
    }
    struct timespec t0, t1;
    for (int f = 0; f < warm + frames; ++f) {
        if (f == warm)
            clock_gettime(CLOCK_MONOTONIC, &t0);
        for (vfo *m : mains)
            m->process(samples);
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    for (vfo *m : mains)
        delete m;
    return ((t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) * 1e-6) / (frames > 0 ? frames : 1);
}

// Does vfo::init throw for this description, and what?  Returns 1 and the exception's what() text where
// the reference's firfilter::sanity_check_1f throws std::out_of_range (vfo.cpp:82-87,110-115), 0 where
// init returns normally, 2 for any other exception.
int dropin_init_probe(const sdrx_vfo_desc *d, const char *addr, char *what, int cap)
{
    vfo *v = new vfo();
    v->setZmqAddress(QString::fromUtf8(addr));
    v->setZmqTopic(QString::fromLatin1(d->topic));
    v->setFs(d->fs);
    v->setDecimationCount(d->decimate_count);
    v->setMixerFreq(d->mixer_freq_hz);
    v->setDemodUSB(d->demod_usb != 0);
    v->setFilterBandwidth(d->filter_bw_hz);
    v->setGain(d->gain);
    int rc = 0;
    try {
        v->init(d->samples_per_buffer, true, d->late_decimate);
    } catch (const std::out_of_range &e) {
        rc = 1;
        if (what && cap > 0) {
            std::strncpy(what, e.what(), (size_t)cap - 1);
            what[cap - 1] = 0;
        }
    } catch (...) {
        rc = 2;
    }
    // (an object whose init threw is not deleted: the reference's destructor would free members init never set)
    if (rc == 0)
        delete v;
    return rc;
}

} // extern "C"
