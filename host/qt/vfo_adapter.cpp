// host/qt/vfo_adapter.cpp -- `class vfo` of the reference's UNMODIFIED vfo.h, implemented over
// libsdrx.so.  This is the file a maintainer puts in place of vfo.cpp:
// mainwindow.cpp, sdrj.cpp and zmqpublisher.cpp stay as they are; oscillator.cpp,
// halfbanddecimator.cpp, jonti/dsp.cpp and gnuradio/firfilter.cpp are no longer linked (only
// their headers are still included by vfo.h for the now unused private members).
//
// How the reference's per-object interface maps onto the whole-tree C ABI:
//   * setters only record parameters in the object's own (private) members, as in vfo.cpp:177-233;
//   * init() records the frame length, computes outputRate and does the ZMQ bind/connect exactly
//     as vfo.cpp:160-172 does (the sockets stay on the host);
//   * the first process() call commits the tree to the GPU: main VFOs are the initialised objects
//     nobody holds in an mpVFOs list, in creation order (the order sdrj's list has,
//     mainwindow.cpp:98-147), children in list order -- ids are creation order = publish order;
//   * sdrj::demodData calls process() on every main VFO with the same frame (sdrj.cpp:288-294):
//     the first main VFO submits the frame, the calls on the other mains return at once;
//   * every leaf's payload comes back through the library's publish callback in the reference's
//     order and goes out through transmitData() -> the unchanged ZmqPublisher::publish;
//   * fftData carries decimate[decimateCount] of the node fftVFOSlot selected (vfo.cpp:290-293).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <unordered_map>
#include <unordered_set>

#include "vfo.h"

#include "../../include/sdrx.h"

ZmqPublisher vfo::bind_publisher; // vfo.h:66 (static, shared by all binding VFOs)

namespace {
struct NodeState {
    int id = -1;
    int samples_per_buffer = 0;
    int late_decimate = 0;
};
struct Registry {
    sdrx_ctx *ctx = nullptr;
    bool committed = false;
    std::vector<vfo *> created;                       // initialised objects, in init() order
    std::unordered_map<const vfo *, NodeState> state;
    std::vector<vfo *> nodes;                         // by library id
    std::vector<vfo *> leaves;                        // in publish order
    size_t cursor = 0;                                // next leaf the publish callback serves
    vfo *first_main = nullptr;
} g;

void fatal(const char *what)
{
    qFatal("sdrx adapter: %s: %s", what, sdrx_last_error(g.ctx));
}
} // namespace

vfo::vfo(QObject *parent) : QObject(parent)
{
    // vfo.cpp:6-31
    gain = 0.01f;
    demodUSB = true;
    filterAudio = false;
    filterbw = 0;
    offsetbw = 0;
    mpVFOs = nullptr;
    emitFFT = false;
    FFTcount = 0;
    scalecomp = 1;
    decimateCount = 0;
    Fs = 0;
    mixer_freq = 0;
    outputRate = 0;
    zmqBind = false;
    laststageDecimate = false;
    discard = 0;
    fir_decI = fir_decQ = fir_usb = nullptr;
    osc_mix = osc_bfo = nullptr;
    philbert = nullptr;
}

vfo::~vfo()
{
    if (mpVFOs) // a vfo owns its children (vfo.cpp:49-57)
        for (int a = 0; a < mpVFOs->length(); ++a)
            delete mpVFOs->at(a);
    g.state.erase(this);
    g.created.erase(std::remove(g.created.begin(), g.created.end(), this), g.created.end());
    if (g.created.empty() && g.ctx) { // the receiver was stopped: the next start builds a new tree
        sdrx_destroy(g.ctx);
        g = Registry();
    }
}

void vfo::setZmqAddress(QString address) { zmqAddress = address; }
void vfo::setZmqTopic(QString topic) { zmqTopic = topic; }
void vfo::setScaleComp(int scale) { scalecomp = scale; }
void vfo::setFs(int samplerate) { Fs = samplerate; }
void vfo::setDecimationCount(int count) { decimateCount = count; }
void vfo::setMixerFreq(double freq) { mixer_freq = freq; }
double vfo::getMixerFreq() { return mixer_freq; }
int vfo::getOutRate() { return Fs / (pow(2, decimateCount)); }
void vfo::setOffsetBandwidth(double bw) { offsetbw = bw; }
void vfo::setFilterBandwidth(double bw) { filterbw = bw; }
void vfo::setGain(float gn) { gain = gn; }
void vfo::setDemodUSB(bool usb) { demodUSB = usb; }
bool vfo::getDemodUSB() { return demodUSB; }
void vfo::setCompressonStyle(int st) { cstyle = st; }
void vfo::setFilter(bool filter, int bw)
{
    filterAudio = filter;
    filterbw = bw;
}
void vfo::setVFOs(QVector<vfo *> *pVFOs) { mpVFOs = pVFOs; }
void vfo::fftVFOSlot(QString topic) // vfo.cpp:492-509
{
    emitFFT = topic.compare(zmqTopic) == 0;
    FFTcount = 0;
}

void vfo::init(int samplesPerBuffer, bool bind, int lateDecimate)
{
    if (g.committed)
        qFatal("sdrx adapter: vfo::init after the first process() -- delete the VFOs and build the tree again");
    NodeState &st = g.state[this];
    st.samples_per_buffer = samplesPerBuffer;
    st.late_decimate = (demodUSB && lateDecimate > 0) ? lateDecimate : 0; // vfo.cpp:69
    if (std::find(g.created.begin(), g.created.end(), this) == g.created.end())
        g.created.push_back(this);
    int targetRate = Fs / (pow(2, decimateCount)); // vfo.cpp:65,74,102
    if (st.late_decimate > 0)
        targetRate = targetRate / lateDecimate;
    outputRate = targetRate;
    // the sockets stay where they were (vfo.cpp:160-172)
    if (!vfo::bind_publisher.connected && bind) {
        vfo::bind_publisher.setAddress(zmqAddress);
        vfo::bind_publisher.setBind(bind);
        vfo::bind_publisher.connect();
    } else if (!bind) {
        connect_publisher.setBind(false);
        connect_publisher.setAddress(zmqAddress);
        connect_publisher.connect();
    }
    zmqBind = bind;
}

void vfo::transmitData() // vfo.cpp:426-453: which publisher, which buffer
{
    ZmqPublisher &pub = zmqBind ? vfo::bind_publisher : connect_publisher;
    if (demodUSB)
        pub.publish((unsigned char *)transmit_usb.data(), transmit_usb.size() * sizeof(short), zmqTopic, outputRate);
    else if (zmqTopic.length() > 0)
        pub.publish((unsigned char *)transmit_iq.data(), transmit_iq.size() * sizeof(char), zmqTopic, outputRate);
}

// not on any path of the adapter (the GPU does this work); defined so the class is complete
void vfo::usb_demod() {}
void vfo::usb_decimdemod() {}
void vfo::compress() {}

void vfo::process(const std::vector<cpx_typef> &samples)
{
    if (!g.committed) {
        if (sdrx_create(&g.ctx, 0) != SDRX_OK)
            qFatal("sdrx adapter: sdrx_create: %s", sdrx_last_error(nullptr));
        std::unordered_set<const vfo *> children;
        for (vfo *v : g.created)
            if (v->mpVFOs)
                for (vfo *c : *v->mpVFOs)
                    children.insert(c);
        struct Walk {
            static void add(vfo *v, int parent)
            {
                auto it = g.state.find(v);
                if (it == g.state.end())
                    qFatal("sdrx adapter: a VFO of the tree was never initialised (vfo::init)");
                sdrx_vfo_desc d;
                std::memset(&d, 0, sizeof d);
                d.fs = v->Fs;
                d.decimate_count = v->decimateCount;
                d.mixer_freq_hz = v->mixer_freq;
                d.demod_usb = v->demodUSB ? 1 : 0;
                d.late_decimate = it->second.late_decimate;
                d.filter_bw_hz = v->filterbw;
                d.gain = v->gain;
                d.cstyle = v->cstyle;
                d.scalecomp = v->scalecomp;
                d.parent_id = parent;
                d.samples_per_buffer = it->second.samples_per_buffer;
                const QByteArray t = v->zmqTopic.toUtf8();
                std::memcpy(d.topic, t.constData(), std::min<size_t>((size_t)t.size(), sizeof(d.topic) - 1));
                if (sdrx_add_vfo(g.ctx, &d, &it->second.id) != SDRX_OK)
                    fatal("sdrx_add_vfo");
                g.nodes.push_back(v);
                const bool leaf = !v->mpVFOs || v->mpVFOs->isEmpty();
                if (leaf)
                    g.leaves.push_back(v);
                else
                    for (vfo *c : *v->mpVFOs)
                        add(c, it->second.id);
            }
        };
        for (vfo *v : g.created)
            if (!children.count(v)) {
                if (!g.first_main)
                    g.first_main = v;
                Walk::add(v, -1);
            }
        // a lambda inside a member function may touch private members: the payload lands in the
        // object's own transmit buffer and leaves through its own transmitData()
        sdrx_set_publish_callback(
            g.ctx,
            [](void *, const char *, uint32_t, const void *buf, uint32_t len) {
                // leaves that publish nothing (non-USB without a topic) are skipped by the library too
                while (g.cursor < g.leaves.size() && !g.leaves[g.cursor]->demodUSB && g.leaves[g.cursor]->zmqTopic.length() == 0)
                    ++g.cursor;
                if (g.cursor >= g.leaves.size())
                    return;
                vfo *v = g.leaves[g.cursor++];
                if (v->demodUSB)
                    v->transmit_usb.assign((const short *)buf, (const short *)buf + len / sizeof(short));
                else
                    v->transmit_iq.assign((const signed char *)buf, (const signed char *)buf + len);
                v->transmitData();
            },
            nullptr);
        if (sdrx_finalize(g.ctx) != SDRX_OK) // the reference throws std::out_of_range from init here (firfilter.cpp:122-134)
            fatal("sdrx_finalize");
        g.committed = true;
    }
    if (this != g.first_main)
        return; // the first main VFO's call processed the whole tree for this frame
    g.cursor = 0;
    if (sdrx_process(g.ctx, reinterpret_cast<const float *>(samples.data()), (int)samples.size()) != SDRX_OK)
        fatal("sdrx_process");
    for (vfo *v : g.nodes) // vfo.cpp:290-293
        if (v->emitFFT) {
            int n = 0;
            const int id = g.state[v].id;
            if (sdrx_get_stream(g.ctx, id, nullptr, 0, &n) != SDRX_OK)
                fatal("sdrx_get_stream");
            std::vector<cpx_typef> &dst = v->decimate[v->decimateCount];
            dst.resize((size_t)n);
            if (sdrx_get_stream(g.ctx, id, reinterpret_cast<float *>(dst.data()), n, &n) != SDRX_OK)
                fatal("sdrx_get_stream");
            emit v->fftData(dst);
        }
}
