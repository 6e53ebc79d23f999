// host/sdrx_host.hpp -- C++ host layer over the C ABI (include/sdrx.h), Qt-free, header-only.
//
// The reference's host is Qt/C++.  This header gives a C++ host the reference's own interface for
// the path, minus Qt:
//   sdrx_host::vfo   -- setters / init / setVFOs of `class vfo`           (vfo.h:16-49)
//   sdrx_host::sdrj  -- setVFOs / setDCCorrection / demodData of `sdrj`   (sdrj.h, sdrj.cpp:266-305)
//   sdrx_host::load_profile -- the INI -> VFO tree rules of MainWindow    (mainwindow.cpp:27-233)
// A Qt front-end wraps these; tests drive them through host/demo.cpp.
#pragma once
#include <algorithm>
#include <cmath>
#include <complex>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../include/sdrx.h"

namespace sdrx_host {

// ------------------------------------------------------------------------------------------ vfo
class vfo {
public:
    vfo()
    {
        std::memset(&d, 0, sizeof d);
        d.gain = 0.01f;   // vfo.cpp:9
        d.demod_usb = 1;  // vfo.cpp:15
        d.scalecomp = 1;  // vfo.cpp:24
        d.parent_id = -1;
    }
    void setFs(int samplerate) { d.fs = samplerate; }
    void setDecimationCount(int count) { d.decimate_count = count; }
    void setMixerFreq(double freq) { d.mixer_freq_hz = freq; }
    double getMixerFreq() const { return d.mixer_freq_hz; }
    int getOutRate() const { return (int)(d.fs / std::pow(2, d.decimate_count)); } // vfo.cpp:212-217
    void setFilterBandwidth(double bw) { d.filter_bw_hz = (int)bw; }
    void setGain(float g) { d.gain = g; }
    void setDemodUSB(bool usb) { d.demod_usb = usb ? 1 : 0; }
    bool getDemodUSB() const { return d.demod_usb != 0; }
    void setCompressonStyle(int st) { d.cstyle = st; } // sic, vfo.h:36
    void setScaleComp(int scale) { d.scalecomp = scale; }
    void setZmqTopic(const std::string &t)
    {
        topic = t;
        std::memset(d.topic, 0, sizeof d.topic);
        std::memcpy(d.topic, t.data(), std::min<size_t>(t.size(), sizeof(d.topic) - 1));
    }
    void setZmqAddress(const std::string &a) { zmqAddress = a; } // the socket stays on the host side
    void init(int samplesPerBuffer, bool /*bind*/, int lateDecimate = 0)
    {
        d.samples_per_buffer = samplesPerBuffer;
        d.late_decimate = lateDecimate;
        initialised = true;
    }
    void setVFOs(std::vector<vfo *> *pVFOs) { mpVFOs = pVFOs; }
    // vfo.cpp:492-509: decimate[decimateCount] goes to fftData after every frame while the selected
    // topic is this VFO's own (the Qt signal of vfo.h:46 is a plain callback here)
    void fftVFOSlot(const std::string &t) { emitFFT = t == topic; }
    std::function<void(const std::vector<std::complex<float>> &)> fftData;
    bool emitFFT = false;

    sdrx_vfo_desc d;
    std::string topic, zmqAddress;
    std::vector<vfo *> *mpVFOs = nullptr;
    int id = -1;
    bool initialised = false;
};

// ------------------------------------------------------------------------------------------ sdrj
// publish(topic5, rate, payload, len): what vfo::transmitData hands to ZmqPublisher::publish.
using publish_fn = std::function<void(const char topic[5], uint32_t rate, const void *buf, uint32_t len)>;

class sdrj {
public:
    explicit sdrj(int device = 0) : devices_(1, device) {}
    // One tree on several GPUs of the node (sdrx_group_*): sub VFOs block-partitioned per main VFO, the raw
    // frame fanned out from the first device over xGMI.  A device may be named twice (two shards on one GPU).
    explicit sdrj(std::vector<int> devices) : devices_(std::move(devices))
    {
        if (devices_.empty())
            devices_.push_back(0);
    }
    ~sdrj()
    {
        if (ctx_)
            sdrx_destroy(ctx_);
        if (grp_)
            sdrx_group_destroy(grp_);
    }
    sdrj(const sdrj &) = delete;
    sdrj &operator=(const sdrj &) = delete;

    void setVFOs(std::vector<vfo *> *vfos) { mpVFOs = vfos; }
    void setDCCorrection(bool dc) { correctDC = dc; }
    void setPublisher(publish_fn f) { publish_ = std::move(f); }
    void setOption(const std::string &name, int value) { options_[name] = value; }
    // sdrj.cpp:84-101: the raw spectrum is selected by the topic "Main"; any selection restarts
    // the every-4th-call counter of demodData (sdrj.cpp:296-303)
    void fftVFOSlot(const std::string &topic)
    {
        emitFFT = topic == "Main";
        count = 0;
    }
    std::function<void(const std::vector<std::complex<float>> &)> fftData; // signal of sdrj.h:40

    // Commit the tree to the GPU (== all vfo::init work).  Called by the first demodData.
    void start()
    {
        if (!mpVFOs || mpVFOs->empty())
            throw std::runtime_error("sdrj: no main VFOs");
        if (devices_.size() > 1) {
            check(sdrx_group_create(&grp_, devices_.data(), (int)devices_.size()), "sdrx_group_create");
            for (auto &kv : options_)
                check(sdrx_group_set_option(grp_, kv.first.c_str(), kv.second), "sdrx_group_set_option");
            for (vfo *m : *mpVFOs)
                add(m, -1);
            check(sdrx_group_set_publish_callback(grp_, &sdrj::trampoline, this), "sdrx_group_set_publish_callback");
            check(sdrx_group_finalize(grp_), "sdrx_group_finalize");
            return;
        }
        check(sdrx_create(&ctx_, devices_[0]), "sdrx_create");
        for (auto &kv : options_)
            check(sdrx_set_option(ctx_, kv.first.c_str(), kv.second), "sdrx_set_option");
        for (vfo *m : *mpVFOs)
            add(m, -1); // parents first: ids are creation order = the reference's publish order
        check(sdrx_set_publish_callback(ctx_, &sdrj::trampoline, this), "sdrx_set_publish_callback");
        check(sdrx_finalize(ctx_), "sdrx_finalize");
    }

    // sdrj::demodData(const float*, int) (sdrj.cpp:266-305): `len` floats, interleaved I/Q.
    void demodData(const float *data, int len)
    {
        if (!started())
            start();
        const float *in = data;
        if (correctDC || (grp_ && emitFFT)) // (a group keeps the host's copy for the raw spectrum tap)
            samples_.assign(data, data + len);
        if (correctDC) { // sdrj.cpp:271-286, on the host exactly where the reference has it
            dc_correct(samples_);
            in = samples_.data();
        }
        select_tap();
        if (grp_)
            check(sdrx_group_process(grp_, in, len / 2), "sdrx_group_process");
        else
            check(sdrx_process(ctx_, in, len / 2), "sdrx_process");
        after_frame(len / 2);
    }
    // rtl_tcp / dongle bytes (sdrj.cpp:149-165): LUT and DC correction on the device.  On several devices
    // the bytes themselves are fanned out (a quarter of the traffic); with the DC-bias IIR on, every device runs
    // the recurrence itself on the whole frame (same bytes, same start state: the same estimate everywhere).
    void demodBytes(const uint8_t *bytes, int n_complex)
    {
        if (!started())
            start();
        select_tap();
        if (!grp_) {
            check(sdrx_process_u8(ctx_, bytes, n_complex, correctDC ? 1 : 0), "sdrx_process_u8");
        } else {
            if (emitFFT && !correctDC) { // the raw spectrum tap of a group without DC removal: the LUT, here
                samples_.resize((size_t)2 * n_complex);
                for (size_t i = 0; i < samples_.size(); ++i)
                    samples_[i] = (float)((int)bytes[i] - 127); // jonti/sdr.cpp:43-49
            } else {
                samples_.clear();
            }
            check(sdrx_group_process_u8(grp_, bytes, n_complex, correctDC ? 1 : 0), "sdrx_group_process_u8");
        }
        after_frame(n_complex);
    }
    sdrx_ctx *context() { return ctx_; }
    sdrx_group *group() { return grp_; }

private:
    bool started() const { return ctx_ || grp_; }
    void dc_correct(std::vector<float> &x)
    {
        const float keep = 1.0f - 0.000001f, k = 0.000001f;
        for (size_t i = 0; i + 1 < x.size(); i += 2) {
            avept_[0] = avept_[0] * keep + k * x[i];
            avept_[1] = avept_[1] * keep + k * x[i + 1];
            x[i] -= avept_[0];
            x[i + 1] -= avept_[1];
        }
    }
    // the context that holds VFO `id` and its id there
    sdrx_ctx *locate(int id, int *local)
    {
        if (!grp_) {
            *local = id;
            return ctx_;
        }
        int member = -1;
        sdrx_ctx *c = nullptr;
        check(sdrx_group_locate(grp_, id, &member, local), "sdrx_group_locate");
        check(sdrx_group_member(grp_, member, &c, nullptr), "sdrx_group_member");
        return c;
    }
    // fftVFOSlot selected VFOs (vfo.cpp:492-509: EVERY VFO whose topic equals the selected string): tell the library before the
    // frame runs -- a leaf whose late decimation is fused into the mix wave keeps decimate[0] only while it is a tap
    // (sdrx_set_tap replaces the selection of a context, sdrx_add_tap adds to it)
    void select_tap()
    {
        std::vector<vfo *> want;
        for (vfo *v : all_)
            if (v->emitFFT && v->fftData)
                want.push_back(v);
        if (want == taps_)
            return;
        for (sdrx_ctx *c : tap_ctxs_)
            check_ctx(c, sdrx_set_tap(c, -1), "sdrx_set_tap");
        tap_ctxs_.clear();
        for (vfo *v : want) {
            int lid = -1;
            sdrx_ctx *c = locate(v->id, &lid);
            check_ctx(c, sdrx_add_tap(c, lid), "sdrx_add_tap");
            if (std::find(tap_ctxs_.begin(), tap_ctxs_.end(), c) == tap_ctxs_.end())
                tap_ctxs_.push_back(c);
        }
        taps_ = want;
    }
    // vfo::process ends with `if (emitFFT) emit fftData(decimate[decimateCount])` (vfo.cpp:290-293);
    // demodData with `if (count == 4 && emitFFT) { emit fftData(samples); count = 0; } count++`.
    void after_frame(int n_complex)
    {
        for (vfo *v : all_)
            if (v->emitFFT && v->fftData) {
                int n = 0, lid = -1;
                sdrx_ctx *c = locate(v->id, &lid);
                check_ctx(c, sdrx_get_stream(c, lid, nullptr, 0, &n), "sdrx_get_stream");
                tap_.resize((size_t)n);
                check_ctx(c, sdrx_get_stream(c, lid, reinterpret_cast<float *>(tap_.data()), n, &n), "sdrx_get_stream");
                v->fftData(tap_);
            }
        if (count == 4 && emitFFT) {
            if (fftData) {
                int n = 0;
                tap_.resize((size_t)n_complex);
                if (grp_) { // the frame as the host handed it over (after its own LUT / DC removal) ...
                    if (samples_.size() == (size_t)2 * n_complex) {
                        std::memcpy(static_cast<void *>(tap_.data()), samples_.data(), sizeof(float) * samples_.size());
                    } else { // ... or, bytes with the DC removal done on the devices, as the first one holding VFOs kept it
                        sdrx_ctx *c = nullptr;
                        for (int k = 0; k < sdrx_group_size(grp_) && !c; ++k)
                            check(sdrx_group_member(grp_, k, &c, nullptr), "sdrx_group_member");
                        if (c && sdrx_get_raw(c, reinterpret_cast<float *>(tap_.data()), n_complex, &n) == SDRX_OK)
                            tap_.resize((size_t)n);
                        else
                            tap_.clear();
                    }
                } else {
                    check(sdrx_get_raw(ctx_, reinterpret_cast<float *>(tap_.data()), n_complex, &n), "sdrx_get_raw");
                    tap_.resize((size_t)n);
                }
                if (!tap_.empty())
                    fftData(tap_);
            }
            count = 0;
        }
        count++;
    }
    void add(vfo *v, int parent)
    {
        all_.push_back(v);
        if (!v->initialised)
            throw std::runtime_error("vfo::init was not called");
        v->d.parent_id = parent;
        if (grp_)
            check(sdrx_group_add_vfo(grp_, &v->d, &v->id), "sdrx_group_add_vfo");
        else
            check(sdrx_add_vfo(ctx_, &v->d, &v->id), "sdrx_add_vfo");
        if (v->mpVFOs)
            for (vfo *c : *v->mpVFOs)
                add(c, v->id);
    }
    void check(int rc, const char *what)
    {
        if (rc != SDRX_OK)
            throw std::runtime_error(std::string(what) + ": " + (grp_ ? sdrx_group_last_error(grp_) : sdrx_last_error(ctx_)));
    }
    static void check_ctx(sdrx_ctx *c, int rc, const char *what)
    {
        if (rc != SDRX_OK)
            throw std::runtime_error(std::string(what) + ": " + sdrx_last_error(c));
    }
    static void trampoline(void *user, const char topic[5], uint32_t rate, const void *buf, uint32_t len)
    {
        sdrj *self = static_cast<sdrj *>(user);
        if (self->publish_)
            self->publish_(topic, rate, buf, len);
    }
    std::vector<int> devices_;
    sdrx_ctx *ctx_ = nullptr;
    sdrx_group *grp_ = nullptr;
    std::vector<vfo *> *mpVFOs = nullptr;
    bool correctDC = false;
    bool emitFFT = false;
    int count = 0;
    std::vector<vfo *> all_;
    std::vector<std::complex<float>> tap_;
    std::vector<vfo *> taps_;          // the VFOs the library was last told about (sdrx_add_tap) ...
    std::vector<sdrx_ctx *> tap_ctxs_; // ... and the contexts that hold them
    float avept_[2] = {0.f, 0.f};
    std::vector<float> samples_;
    publish_fn publish_;
    std::map<std::string, int> options_;
};

// ------------------------------------------------------------------------------------------ INI
// The subset of QSettings::IniFormat the shipped profiles use: [section] headers, key=value with
// blanks trimmed, `N\key` array members (backslash = group separator), top-level keys before any
// section, ';' comment lines.  A leading '#' is NOT a comment for QSettings: such a line defines a
// key nobody reads (sdr_25E.ini:5-9).
inline std::map<std::string, std::string> parse_ini(std::istream &in)
{
    auto trim = [](std::string s) {
        size_t a = s.find_first_not_of(" \t\r\n"), b = s.find_last_not_of(" \t\r\n");
        return a == std::string::npos ? std::string() : s.substr(a, b - a + 1);
    };
    std::map<std::string, std::string> kv;
    std::string line, section;
    while (std::getline(in, line)) {
        line = trim(line);
        if (line.empty() || line[0] == ';')
            continue;
        if (line.front() == '[' && line.back() == ']') {
            section = trim(line.substr(1, line.size() - 2));
            std::string low = section;
            std::transform(low.begin(), low.end(), low.begin(), ::tolower);
            if (low == "general")
                section.clear();
            continue;
        }
        size_t eq = line.find('=');
        if (eq == std::string::npos)
            continue;
        std::string k = trim(line.substr(0, eq)), v = trim(line.substr(eq + 1));
        std::replace(k.begin(), k.end(), '\\', '/');
        if (v.size() >= 2 && v.front() == '"' && v.back() == '"')
            v = v.substr(1, v.size() - 2);
        kv[(section.empty() ? "" : section + "/") + k] = v;
    }
    return kv;
}

// A whole receiver profile: owns the vfo objects; `mains` is what sdrj::setVFOs receives.
struct Profile {
    int fs = 0, frame = 0, bufsplit = 4, center_frequency = 0;
    bool correct_dc = false;
    std::string zmq_address;
    std::vector<std::unique_ptr<vfo>> all; // creation order: mains, then subs in INI order
    std::vector<vfo *> mains;
    std::vector<std::vector<vfo *>> subs; // per main (VFOsub[i], mainwindow.h:82)
};

// ---- the INI front door ----------------------------------------------------------
// The rules by which the reference turns its settings file into a VFO tree (mainwindow.cpp:27-233), restated as
// three small pure steps -- settings access, frame geometry, sub-VFO placement -- and an instantiation step.

// QSettings' view of the parsed file: missing or malformed numbers read as 0 (QVariant::toInt / toFloat).
class Settings {
public:
    explicit Settings(std::map<std::string, std::string> kv) : kv_(std::move(kv)) {}
    std::string text(const std::string &key) const
    {
        auto it = kv_.find(key);
        return it == kv_.end() ? std::string() : it->second;
    }
    int integer(const std::string &key) const
    {
        const std::string s = text(key);
        if (s.empty())
            return 0;
        char *end = nullptr;
        const long long v = std::strtoll(s.c_str(), &end, 10);
        return (*end != 0 || v < INT32_MIN || v > INT32_MAX) ? 0 : (int)v;
    }
    float real(const std::string &key) const
    {
        const std::string s = text(key);
        char *end = nullptr;
        const float v = std::strtof(s.c_str(), &end);
        return (s.empty() || *end != 0) ? 0.0f : v;
    }
    // `N\\key` array members of QSettings::beginReadArray(group): "<group>/<i>/<key>", i = 1 .. size
    std::string item(const std::string &group, int i, const char *key) const { return group + "/" + std::to_string(i) + "/" + key; }

private:
    std::map<std::string, std::string> kv_;
};

// How many floats the ingest delivers per callback: a quarter of a second of I/Q, or a fifth where a quarter
// is not a multiple of 512 floats (288 kS/s) -- mainwindow.cpp:65-80.
struct FrameGeometry {
    int floats_per_callback, callbacks_per_second;
    int complex_per_frame() const { return floats_per_callback / 2; }
};
inline FrameGeometry frame_geometry(int sample_rate)
{
    const int quarter = 2 * sample_rate / 4;
    if (quarter % 512 == 0)
        return {quarter, 4};
    return {2 * sample_rate / 5, 5};
}

// Where a sub VFO hangs and how it decimates, decided from the mains already configured
// (mainwindow.cpp:179-216): the first IQ main whose output band covers the channel; below a 240 k / 288 k main
// the last step to 48 k is the /5 or /6 FIR ("late decimation"), the half-bands do the rest.
struct SubPlacement {
    int main_index = 0, input_rate = 0, parent_mixer = 0, halfband_stages = 0, late_decimate = 0;
};
inline SubPlacement place_sub(const std::vector<vfo *> &mains, int receiver_rate, int center, int channel_freq, int out_rate)
{
    SubPlacement pl;
    pl.input_rate = receiver_rate;
    for (size_t a = 0; a < mains.size(); ++a) {
        vfo &m = *mains[a];
        const int main_centre = center - (int)m.getMixerFreq();
        if (std::abs(main_centre - channel_freq) < m.getOutRate() && !m.getDemodUSB()) {
            pl.main_index = (int)a;
            pl.parent_mixer = (int)m.getMixerFreq();
            pl.input_rate = m.getOutRate();
            break;
        }
    }
    const int fir_step = pl.input_rate / 48000; // 5 below a 240 k main, 6 below a 288 k one
    if (fir_step == 5 || fir_step == 6) {
        pl.late_decimate = fir_step;
        pl.halfband_stages = (int)std::log2(pl.input_rate / (fir_step * out_rate));
    } else {
        pl.halfband_stages = (int)std::log2(receiver_rate / out_rate) - (int)std::log2(receiver_rate / pl.input_rate);
    }
    return pl;
}

inline std::unique_ptr<Profile> load_profile(std::istream &in)
{
    const Settings ini(parse_ini(in));
    auto prof = std::unique_ptr<Profile>(new Profile());
    Profile &P = *prof;
    P.fs = ini.integer("sample_rate");
    if (P.fs == 0)
        throw std::runtime_error("sample_rate ini file key not found or equal to zero"); // mainwindow.cpp:31-37
    if (P.fs != 288000 && P.fs != 1536000 && P.fs != 1920000)                            // mainwindow.h:29
        throw std::runtime_error("sample_rate " + std::to_string(P.fs) + " not supported");
    const FrameGeometry geo = frame_geometry(P.fs);
    P.frame = geo.complex_per_frame();
    P.bufsplit = geo.callbacks_per_second;
    P.center_frequency = ini.integer("center_frequency");
    P.zmq_address = ini.text("zmq_address");
    P.correct_dc = ini.text("correct_dc_bias") == "1";
    const int channel_offset = ini.integer("mix_offset");

    // main VFOs: IQ only, compress() style 1, fed by the raw stream (mainwindow.cpp:98-138)
    const int n_mains = std::max(0, ini.integer("main_vfos/size"));
    P.subs.resize((size_t)n_mains);
    for (int i = 1; i <= n_mains; ++i) {
        auto key = [&](const char *k) { return ini.item("main_vfos", i, k); };
        const int out_rate = ini.integer(key("out_rate"));
        if (out_rate <= 0)
            throw std::runtime_error(key("out_rate") + " missing");
        P.all.emplace_back(new vfo());
        vfo &m = *P.all.back();
        if (const int scale = ini.integer(key("compress_scale")))
            if (scale > 0)
                m.setScaleComp(scale);
        if (!ini.text(key("zmq_address")).empty() && !ini.text(key("zmq_topic")).empty()) {
            m.setZmqAddress(ini.text(key("zmq_address")));
            m.setZmqTopic(ini.text(key("zmq_topic")));
        }
        m.setFs(P.fs);
        m.setDecimationCount(P.fs / out_rate == 1 ? 0 : (int)std::log2(P.fs / out_rate));
        m.setMixerFreq(P.center_frequency - ini.integer(key("frequency")));
        m.setDemodUSB(false);
        m.setCompressonStyle(1);
        m.init(P.frame, false);
        m.setVFOs(&P.subs[(size_t)i - 1]);
        P.mains.push_back(&m);
    }

    // sub VFOs: USB audio leaves under the main that covers them (mainwindow.cpp:141-233)
    const int n_subs = ini.integer("vfos/size");
    for (int i = 1; i <= n_subs; ++i) {
        auto key = [&](const char *k) { return ini.item("vfos", i, k); };
        int out_rate = ini.integer(key("out_rate"));
        if (out_rate == 0) { // the older profiles give the data rate of the channel instead
            const int data_rate = ini.integer(key("data_rate"));
            if (data_rate > 0)
                out_rate = data_rate == 600 ? 12000 : data_rate == 1200 ? 24000 : 48000;
        }
        if (out_rate <= 0)
            throw std::runtime_error(ini.item("vfos", i, "") + ": neither out_rate nor data_rate given");
        if (P.mains.empty())
            throw std::runtime_error("profile has sub VFOs but no main VFO");
        const int channel = ini.integer(key("frequency")) + channel_offset;
        const SubPlacement pl = place_sub(P.mains, P.fs, P.center_frequency, channel, out_rate);
        P.all.emplace_back(new vfo());
        vfo &v = *P.all.back();
        v.setZmqTopic(ini.text(key("topic")));
        v.setZmqAddress(P.zmq_address);
        v.setDecimationCount(pl.halfband_stages);
        v.setFilterBandwidth(ini.integer(key("filter_bandwidth")));
        v.setGain(ini.real(key("gain")) / 100);
        v.setMixerFreq((P.center_frequency - pl.parent_mixer) - channel);
        v.setFs(pl.input_rate);
        v.setCompressonStyle(1);
        v.init(pl.input_rate / P.bufsplit, true, pl.late_decimate);
        P.subs[(size_t)pl.main_index].push_back(&v);
    }
    return prof;
}

inline std::unique_ptr<Profile> load_profile_file(const std::string &path)
{
    std::ifstream f(path);
    if (!f)
        throw std::runtime_error("Given settings ini file doesn't exist: " + path); // mainwindow.cpp:20-25
    return load_profile(f);
}

} // namespace sdrx_host
