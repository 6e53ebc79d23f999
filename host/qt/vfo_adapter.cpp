// host/qt/vfo_adapter.cpp -- `class vfo` of the reference's UNMODIFIED vfo.h, implemented over
// libsdrx.so.  This is the file a maintainer puts in place of vfo.cpp:
// zmqpublisher.cpp stays as it is and is still linked; oscillator.cpp, halfbanddecimator.cpp,
// jonti/dsp.cpp and gnuradio/firfilter.cpp are no longer linked (only their headers are still
// included by vfo.h for the now unused private members).
//
// How the reference's per-object interface maps onto the whole-tree C ABI:
//   * setters only record parameters in the object's own (private) members, as in vfo.cpp:177-233;
//   * init() records the frame length, computes outputRate, does the ZMQ bind/connect exactly as
//     vfo.cpp:160-172 does (the sockets stay on the host) and validates the filter specification on
//     the host: where the reference's init throws std::out_of_range from firfilter::sanity_check_1f
//     (vfo.cpp:82-87,110-115 -> firfilter.cpp:122-134) this init throws the same exception with the
//     same what() text;
//   * a TREE is keyed by its ROOT object: the first process() call on a vfo nobody holds in an mpVFOs
//     list commits that vfo and everything below it (children in list order: ids are creation order
//     = the reference's publish order, vfo.cpp:257-263) to a context of its own.  sdrj::demodData
//     calls process() on every main VFO in turn (sdrj.cpp:288-294): each main runs its own subtree,
//     exactly the reference's semantics, so any number of receivers can live in one process, and
//     deleting a root (MainWindow's stop, vfo.cpp:34-59) frees its context -- the next start builds a
//     new one;
//   * every leaf's payload comes back through the library's publish callback in the reference's order
//     and goes out through the publisher transmitData() would pick -> the unchanged ZmqPublisher::publish, straight from
//     the library's pinned buffer (no copy into the private transmit_usb / transmit_iq: zmq_send copies anyway);
//   * fftData carries decimate[decimateCount] of the node fftVFOSlot selected (vfo.cpp:290-293).
//
// Device selection (environment, read when a tree is committed): SDRX_DEVICE=<ordinal> (default 0), or
// SDRX_DEVICES=<a,b,...>: the tree sharded over several GPUs of the node (sdrx_group_*).
//
// One upload per frame: sdrj::demodData hands every main VFO the SAME `samples` (sdrj.cpp:288-294).  The root that
// is called first uploads the frame; a root of another tree on the same device that is then called runs on that
// uploaded copy (sdrx_process_if_same / sdrx_submit_if_same) if -- and only if -- its samples ARE the uploaded frame:
// the library compares them byte for byte with the uploader's pinned staging copy (a memcmp of the frame instead of
// its upload; no address or spot-check heuristics).  One PCIe crossing per frame instead of one per main VFO -- which,
// measured on config 3 (two mains, bench.py through_abi.qt_adapter), buys nothing: 1.833 ms per frame with it, 1.830
// without (the comparison costs the host what the upload's staging copy did; the copy engine was idle anyway).  So it
// is OFF unless SDRX_SHARE_UPLOAD=1 asks for it.
//
// SDRX_PIPELINE=1: process() only SUBMITS its frame (sdrx_submit*) and delivers the PREVIOUS frame's payloads --
// the frame's kernels and the payload copy of the one before run concurrently, 0.30 instead of 0.55 ms per frame
// and main on BASELINE config 3 -- at the price of one frame (250 ms) of latency; the last frame is delivered when
// the tree is deleted (MainWindow's stop).  While an fftData tap is selected anywhere, frames are delivered at once.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <unordered_map>

#include "vfo.h"

#include "../../include/sdrx.h"

ZmqPublisher vfo::bind_publisher; // vfo.h:66 (static, shared by all binding VFOs)

namespace {

// One committed tree = one receiver rooted at one main VFO.
struct Tree {
    sdrx_ctx *ctx = nullptr;   // one device ...
    sdrx_group *grp = nullptr; // ... or several
    std::vector<vfo *> nodes;  // by library id
    std::vector<vfo *> leaves; // in publish order
    size_t cursor = 0;         // next leaf the publish callback serves
    int device = 0;
    bool pipelined = false;    // SDRX_PIPELINE=1
    int in_flight = 0;         // frames submitted and not yet delivered
    std::vector<const vfo *> taps;      // the nodes the library was last told about (fftVFOSlot -> sdrx_add_tap) ...
    std::vector<sdrx_ctx *> tap_ctxs;   // ... and the contexts that hold them
    long dropped = 0;                   // frames lost to a runtime error of the device path (logged, never fatal)
    int consecutive = 0;                // ... in a row
    bool lost = false;                  // the device path does not come back by itself: the next process() builds a new context
    ~Tree();
    const char *error() const { return grp ? sdrx_group_last_error(grp) : sdrx_last_error(ctx); }
    // deliver the oldest submitted frame: payloads -> transmit buffers -> ZmqPublisher, in the reference's order.
    // The reference's hot path cannot fail (every function on it is void; a buffer the dongle thread could not hand over is
    // a time gap and a qDebug line, jonti/sdr.cpp:105-110): a runtime error of the device path is the same kind of event --
    // the frame is dropped and logged, the receiver (and the GUI around it) keeps running.
    void deliver_one()
    {
        cursor = 0;
        const int before = in_flight;
        const int rc = grp ? sdrx_group_wait(grp) : sdrx_wait(ctx);
        in_flight = before - 1;
        if (rc == SDRX_OK) {
            consecutive = 0;
        } else {
            drop("sdrx_wait", rc);
            // The library's own count is the truth: a wait that failed before the library took the frame off its queue has NOT
            // delivered it, and a count of ours that ran ahead of the library's would make every later submit fail with
            // "2 frames in flight" while the `while (in_flight > ...) deliver_one()` loops never drain the library.
            const int q = grp ? sdrx_group_in_flight(grp) : sdrx_in_flight(ctx);
            if (q >= before)
                lost = true; // nothing left the queue: waiting again would spin -- the tree is rebuilt instead
            else if (q >= 0)
                in_flight = q;
        }
        if (lost)
            in_flight = 0;
    }
    // A transient error costs a frame.  HIP errors are sticky, though: after one, every later call of the context fails too.  So a
    // HIP error -- or four failures in a row of any kind -- gives the tree up: the next process() on its root commits the same
    // vfo objects to a NEW context (filter state restarts from zero: a time gap, as after MainWindow's stop / start).
    void drop(const char *what, int rc)
    {
        ++dropped;
        ++consecutive;
        if (rc == SDRX_EHIP || consecutive >= 4)
            lost = true;
        qWarning("sdrx adapter: %s failed (%d), frame dropped (%ld so far)%s: %s", what, rc, dropped,
                 lost ? "; the device context is given up and rebuilt with the next frame" : "", error());
    }
};

// The tree that uploaded a host frame to a device last (see "One upload per frame" above).
struct Upload {
    Tree *owner = nullptr;
};
std::unordered_map<int, Upload> &uploads()
{
    static auto *u = new std::unordered_map<int, Upload>();
    return *u;
}
int g_fft_taps = 0; // VFOs anywhere in the process with an fftData tap selected (vfo::fftVFOSlot)

Tree::~Tree()
{
    auto it = uploads().find(device);
    if (it != uploads().end() && it->second.owner == this)
        uploads().erase(it);
    if (ctx)
        sdrx_destroy(ctx);
    if (grp)
        sdrx_group_destroy(grp);
}

// What the adapter must remember per object and the unmodified header has no member for.
struct NodeState {
    int id = -1;
    int samples_per_buffer = 0;
    int late_decimate = 0;
    bool initialised = false;
    std::shared_ptr<Tree> tree; // the tree this object was committed to (shared by all its nodes)
};

// (both registries are deliberately never destroyed: vfo objects -- and through them trees -- may outlive any
// static of this file at process exit)
std::unordered_map<const vfo *, NodeState> &side()
{
    static auto *s = new std::unordered_map<const vfo *, NodeState>();
    return *s;
}

std::vector<int> devices_from_env()
{
    std::vector<int> d;
    if (const char *e = std::getenv("SDRX_DEVICES")) {
        std::stringstream ss(e);
        std::string tok;
        while (std::getline(ss, tok, ','))
            if (!tok.empty())
                d.push_back(std::atoi(tok.c_str()));
    }
    if (d.empty())
        d.push_back(std::getenv("SDRX_DEVICE") ? std::atoi(std::getenv("SDRX_DEVICE")) : 0);
    return d;
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
    {
        // a root in pipelined mode still owes its subscribers the last frame(s): delivered now, while the leaves exist
        auto it = side().find(this);
        if (it != side().end() && it->second.tree && it->second.id == 0)
            while (it->second.tree->in_flight > 0)
                it->second.tree->deliver_one();
    }
    if (mpVFOs) // a vfo owns its children (vfo.cpp:49-57)
        for (int a = 0; a < mpVFOs->length(); ++a)
            delete mpVFOs->at(a);
    if (emitFFT)
        --g_fft_taps;
    // the tree (context, device memory) goes when its last node does: the root is deleted last
    side().erase(this);
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
    const bool was = emitFFT;
    emitFFT = topic.compare(zmqTopic) == 0;
    g_fft_taps += (emitFFT ? 1 : 0) - (was ? 1 : 0);
    FFTcount = 0;
}

void vfo::init(int samplesPerBuffer, bool bind, int lateDecimate)
{
    NodeState &st = side()[this];
    if (st.tree)
        qFatal("sdrx adapter: vfo::init on a VFO whose tree is already running -- delete the tree and build it again");
    st.samples_per_buffer = samplesPerBuffer;
    st.late_decimate = (demodUSB && lateDecimate > 0) ? lateDecimate : 0; // vfo.cpp:69
    // vfo::init designs its filters here and lets firfilter::sanity_check_1f throw (vfo.cpp:82-87,110-115)
    {
        sdrx_vfo_desc d;
        std::memset(&d, 0, sizeof d);
        d.fs = Fs;
        d.decimate_count = decimateCount;
        d.demod_usb = demodUSB ? 1 : 0;
        d.late_decimate = st.late_decimate;
        d.filter_bw_hz = filterbw;
        d.samples_per_buffer = samplesPerBuffer;
        d.parent_id = -1;
        char why[160];
        if (sdrx_check_vfo(&d, why, sizeof why) == SDRX_EFILTER)
            throw std::out_of_range(why);
    }
    st.initialised = true;
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
    NodeState &me = side()[this];
    if (me.tree && me.tree->lost && me.id == 0) {
        // the device path of this tree failed for good (Tree::drop): let go of the context and commit the tree anew, below
        const std::vector<vfo *> nodes = me.tree->nodes;
        for (vfo *v : nodes) {
            NodeState &st = side()[v];
            st.tree.reset(); // (the last one destroys the Tree and with it the context)
            st.id = -1;
        }
    }
    if (!me.tree) {
        // first frame for this root: commit it and everything below it
        std::shared_ptr<Tree> T = std::make_shared<Tree>();
        const std::vector<int> devices = devices_from_env();
        T->device = devices[0];
        T->pipelined = std::getenv("SDRX_PIPELINE") && std::atoi(std::getenv("SDRX_PIPELINE")) != 0;
        if (devices.size() > 1) {
            if (sdrx_group_create(&T->grp, devices.data(), (int)devices.size()) != SDRX_OK)
                qFatal("sdrx adapter: sdrx_group_create: %s", sdrx_group_last_error(nullptr));
        } else if (sdrx_create(&T->ctx, devices[0]) != SDRX_OK) {
            qFatal("sdrx adapter: sdrx_create(%d): %s", devices[0], sdrx_last_error(nullptr));
        }
        struct Walk {
            static void add(Tree &T, const std::shared_ptr<Tree> &sp, vfo *v, int parent)
            {
                NodeState &st = side()[v];
                if (!st.initialised)
                    qFatal("sdrx adapter: a VFO of the tree was never initialised (vfo::init)");
                if (st.tree)
                    qFatal("sdrx adapter: a VFO is reachable from two roots");
                sdrx_vfo_desc d;
                std::memset(&d, 0, sizeof d);
                d.fs = v->Fs;
                d.decimate_count = v->decimateCount;
                d.mixer_freq_hz = v->mixer_freq;
                d.demod_usb = v->demodUSB ? 1 : 0;
                d.late_decimate = st.late_decimate;
                d.filter_bw_hz = v->filterbw;
                d.gain = v->gain;
                d.cstyle = v->cstyle;
                d.scalecomp = v->scalecomp;
                d.parent_id = parent;
                d.samples_per_buffer = st.samples_per_buffer;
                const QByteArray t = v->zmqTopic.toUtf8();
                std::memcpy(d.topic, t.constData(), std::min<size_t>((size_t)t.size(), sizeof(d.topic) - 1));
                const int rc = T.grp ? sdrx_group_add_vfo(T.grp, &d, &st.id) : sdrx_add_vfo(T.ctx, &d, &st.id);
                if (rc != SDRX_OK)
                    qFatal("sdrx adapter: sdrx_add_vfo: %s", T.error());
                st.tree = sp;
                T.nodes.push_back(v);
                const bool leaf = !v->mpVFOs || v->mpVFOs->isEmpty();
                if (leaf)
                    T.leaves.push_back(v);
                else
                    for (vfo *c : *v->mpVFOs)
                        add(T, sp, c, st.id);
            }
        };
        Walk::add(*T, T, this, -1);
        // a lambda inside a member function may touch private members: the payload leaves through the publisher
        // transmitData() would pick (vfo.cpp:426-453), straight from the library's pinned host buffer -- zmq_send copies
        // what it is handed (zmqpublisher.cpp:91-93), and transmit_usb / transmit_iq are private members nobody else reads,
        // so the 15 MB per frame (config 3) are not copied into them first: that copy was two thirds of a frame's host time
        sdrx_publish_fn deliver = [](void *user, const char *, uint32_t, const void *buf, uint32_t len) {
            Tree &T = *static_cast<Tree *>(user);
            // leaves that publish nothing (non-USB without a topic) are skipped by the library too
            while (T.cursor < T.leaves.size() && !T.leaves[T.cursor]->demodUSB && T.leaves[T.cursor]->zmqTopic.length() == 0)
                ++T.cursor;
            if (T.cursor >= T.leaves.size())
                return;
            vfo *v = T.leaves[T.cursor++];
            ZmqPublisher &pub = v->zmqBind ? vfo::bind_publisher : v->connect_publisher;
            if (v->demodUSB || v->zmqTopic.length() > 0)
                pub.publish(static_cast<unsigned char *>(const_cast<void *>(buf)), len, v->zmqTopic, v->outputRate);
        };
        if (T->grp)
            sdrx_group_set_publish_callback(T->grp, deliver, T.get());
        else
            sdrx_set_publish_callback(T->ctx, deliver, T.get());
        // (filter specifications were validated in init(); what can still fail here is a geometry the
        // kernels do not handle -- not an error path of the reference)
        if ((T->grp ? sdrx_group_finalize(T->grp) : sdrx_finalize(T->ctx)) != SDRX_OK)
            qFatal("sdrx adapter: sdrx_finalize: %s", T->error());
    }
    if (me.id != 0)
        qFatal("sdrx adapter: vfo::process called on a VFO that is a child in a running tree");
    Tree &T = *me.tree;
    T.cursor = 0;
    const float *iq = reinterpret_cast<const float *>(samples.data());
    const int n = (int)samples.size();
    // (any tap anywhere makes EVERY tree deliver at once: the trees of one receiver stay in step with each other)
    const bool want_fft = g_fft_taps > 0;
    // fftVFOSlot selected a node of this tree (vfo.cpp:492-509): the library must know BEFORE the frame runs -- a leaf whose
    // late decimation is fused into the mix wave keeps decimate[0] only while it is the tap (sdrx_set_tap)
    {
        std::vector<const vfo *> want; // (vfo::fftVFOSlot sets emitFFT on EVERY VFO whose topic matches, vfo.cpp:492-509)
        for (vfo *v : T.nodes)
            if (v->emitFFT)
                want.push_back(v);
        if (want != T.taps) {
            while (T.in_flight > 0)
                T.deliver_one();
            bool ok = true;
            for (sdrx_ctx *c : T.tap_ctxs)
                ok = sdrx_set_tap(c, -1) == SDRX_OK && ok;
            T.tap_ctxs.clear();
            for (const vfo *w : want) {
                int id = side()[w].id;
                sdrx_ctx *c = T.ctx;
                if (T.grp) {
                    int member = -1;
                    if (sdrx_group_locate(T.grp, id, &member, &id) != SDRX_OK || sdrx_group_member(T.grp, member, &c, nullptr) != SDRX_OK || !c)
                        qFatal("sdrx adapter: sdrx_group_locate: %s", T.error()); // (a VFO of this very tree: misuse, not a runtime event)
                }
                ok = sdrx_add_tap(c, id) == SDRX_OK && ok;
                if (std::find(T.tap_ctxs.begin(), T.tap_ctxs.end(), c) == T.tap_ctxs.end())
                    T.tap_ctxs.push_back(c);
            }
            if (!ok) // (the selection is retried with the next frame: T.taps stays what it was)
                qWarning("sdrx adapter: selecting the spectrum tap failed: %s", T.error());
            else
                T.taps = want;
        }
    }
    // Who uploads: this tree -- unless the tree that uploaded last on this device holds exactly these samples (sdrj::demodData
    // hands every main VFO the same vector, sdrj.cpp:288-294, but `class vfo` cannot know that): the library compares the
    // caller's frame byte for byte with that tree's pinned staging copy (sdrx_*_if_same: a memcmp instead of an upload) and
    // runs this tree on the frame already on the device only if they are equal.
    const bool may_share = T.ctx && n > 0 && std::getenv("SDRX_SHARE_UPLOAD") && std::atoi(std::getenv("SDRX_SHARE_UPLOAD")) != 0;
    Upload *U = may_share ? &uploads()[T.device] : nullptr;
    sdrx_ctx *from = (U && U->owner && U->owner != &T) ? U->owner->ctx : nullptr;
    bool shared = false;
    if (!T.pipelined) {
        int rc = from ? sdrx_process_if_same(T.ctx, from, iq, n) : SDRX_DIFFERENT;
        shared = rc == SDRX_OK;
        if (rc == SDRX_DIFFERENT)
            rc = T.grp ? sdrx_group_process(T.grp, iq, n) : sdrx_process(T.ctx, iq, n);
        if (rc != SDRX_OK) {
            T.drop("sdrx_process", rc);
            return;
        }
        T.consecutive = 0;
    } else {
        // submit(f); deliver f-1 -- or everything, while a spectrum tap wants this very frame's streams
        int rc = from ? sdrx_submit_if_same(T.ctx, from, iq, n) : SDRX_DIFFERENT;
        shared = rc == SDRX_OK;
        if (rc == SDRX_DIFFERENT)
            rc = T.grp ? sdrx_group_submit(T.grp, iq, n) : sdrx_submit(T.ctx, iq, n);
        if (rc != SDRX_OK) {
            T.drop("sdrx_submit", rc);
            return;
        }
        ++T.in_flight;
        while (T.in_flight > (want_fft ? 0 : 1))
            T.deliver_one();
    }
    if (U && !shared)
        U->owner = &T; // (this tree's staging copy now holds the newest frame on the device)
    for (vfo *v : T.nodes) // vfo.cpp:290-293
        if (v->emitFFT) {
            int n = 0, id = side()[v].id;
            sdrx_ctx *c = T.ctx;
            if (T.grp) {
                int member = -1;
                if (sdrx_group_locate(T.grp, id, &member, &id) != SDRX_OK || sdrx_group_member(T.grp, member, &c, nullptr) != SDRX_OK || !c)
                    qFatal("sdrx adapter: sdrx_group_locate: %s", T.error());
            }
            if (sdrx_get_stream(c, id, nullptr, 0, &n) != SDRX_OK) { // (e.g. selected too late for this frame: no spectrum this time)
                qWarning("sdrx adapter: sdrx_get_stream: %s", sdrx_last_error(c));
                continue;
            }
            std::vector<cpx_typef> &dst = v->decimate[v->decimateCount];
            dst.resize((size_t)n);
            if (sdrx_get_stream(c, id, reinterpret_cast<float *>(dst.data()), n, &n) != SDRX_OK) {
                qWarning("sdrx adapter: sdrx_get_stream: %s", sdrx_last_error(c));
                continue;
            }
            emit v->fftData(dst);
        }
}
