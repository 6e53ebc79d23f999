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

// MainWindow::MainWindow's configuration part (mainwindow.cpp:27-233) on an INI stream.
inline std::unique_ptr<Profile> load_profile(std::istream &in)
{
    const auto kv = parse_ini(in);
    auto str = [&](const std::string &k) { auto it = kv.find(k); return it == kv.end() ? std::string() : it->second; };
    auto toInt = [&](const std::string &k) { // QVariant(QString).toInt(): 0 when missing / not an integer
        const std::string s = str(k);
        if (s.empty())
            return 0;
        char *end = nullptr;
        long long v = std::strtoll(s.c_str(), &end, 10);
        if (*end != 0 || v < INT32_MIN || v > INT32_MAX)
            return 0;
        return (int)v;
    };
    auto toFloat = [&](const std::string &k) {
        const std::string s = str(k);
        char *end = nullptr;
        float v = std::strtof(s.c_str(), &end);
        return (s.empty() || *end != 0) ? 0.0f : v;
    };
    auto P = std::unique_ptr<Profile>(new Profile());
    const int Fs = toInt("sample_rate");
    if (Fs == 0)
        throw std::runtime_error("sample_rate ini file key not found or equal to zero"); // 31-37
    if (Fs != 288000 && Fs != 1536000 && Fs != 1920000)                                  // mainwindow.h:29
        throw std::runtime_error("sample_rate " + std::to_string(Fs) + " not supported");
    P->fs = Fs;
    P->center_frequency = toInt("center_frequency");
    const int mix_offset = toInt("mix_offset");
    int buflen; // "usually 4 buffers per Fs but in some cases 5 due to multiple of 512", 65-80
    if (((2 * Fs) / 4) % 512 > 0) {
        buflen = (2 * Fs) / 5;
        P->bufsplit = 5;
    } else {
        buflen = (2 * Fs) / 4;
    }
    P->frame = buflen / 2;
    P->zmq_address = str("zmq_address");
    P->correct_dc = str("correct_dc_bias") == "1";
    const int center = P->center_frequency;

    const int msize = toInt("main_vfos/size"); // 98-138
    P->subs.resize((size_t)std::max(msize, 0));
    for (int i = 1; i <= msize; ++i) {
        const std::string p = "main_vfos/" + std::to_string(i) + "/";
        const int vfo_freq = toInt(p + "frequency"), out_rate = toInt(p + "out_rate");
        if (out_rate <= 0)
            throw std::runtime_error(p + "out_rate missing");
        vfo *v = new vfo();
        P->all.emplace_back(v);
        const int compscale = toInt(p + "compress_scale");
        if (compscale > 0)
            v->setScaleComp(compscale);
        if (!str(p + "zmq_address").empty() && !str(p + "zmq_topic").empty()) {
            v->setZmqAddress(str(p + "zmq_address"));
            v->setZmqTopic(str(p + "zmq_topic"));
        }
        v->setFs(Fs);
        v->setDecimationCount(Fs / out_rate == 1 ? 0 : (int)std::log2(Fs / out_rate));
        v->setMixerFreq(center - vfo_freq);
        v->setDemodUSB(false);
        v->setCompressonStyle(1);
        v->init(buflen / 2, false);
        v->setVFOs(&P->subs[(size_t)i - 1]);
        P->mains.push_back(v);
    }
    const int size = toInt("vfos/size"); // 141-233
    for (int i = 1; i <= size; ++i) {
        const std::string p = "vfos/" + std::to_string(i) + "/";
        const int vfo_freq = toInt(p + "frequency") + mix_offset;
        const int data_rate = toInt(p + "data_rate");
        int out_rate = toInt(p + "out_rate");
        if (out_rate == 0 && data_rate > 0)
            out_rate = data_rate == 600 ? 12000 : data_rate == 1200 ? 24000 : 48000;
        if (out_rate <= 0)
            throw std::runtime_error(p + ": neither out_rate nor data_rate given");
        if (P->mains.empty())
            throw std::runtime_error("profile has sub VFOs but no main VFO");
        int main_vfo_freq = 0, main_vfo_out_rate = Fs, main_idx = 0;
        for (size_t a = 0; a < P->mains.size(); ++a) { // first main whose band covers the VFO, 179-191
            const int diff = std::abs((center - (int)P->mains[a]->getMixerFreq()) - vfo_freq);
            if (diff < P->mains[a]->getOutRate() && !P->mains[a]->getDemodUSB()) {
                main_idx = (int)a;
                main_vfo_freq = (int)P->mains[a]->getMixerFreq();
                main_vfo_out_rate = P->mains[a]->getOutRate();
                break;
            }
        }
        vfo *v = new vfo();
        P->all.emplace_back(v);
        v->setZmqTopic(str(p + "topic"));
        v->setZmqAddress(P->zmq_address);
        int lateDecimate = 0; // 196-216
        if (main_vfo_out_rate / 48000 == 5) {
            v->setDecimationCount((int)std::log2(main_vfo_out_rate / (5 * out_rate)));
            lateDecimate = 5;
        } else if (main_vfo_out_rate / 48000 == 6) {
            v->setDecimationCount((int)std::log2(main_vfo_out_rate / (6 * out_rate)));
            lateDecimate = 6;
        } else {
            v->setDecimationCount((int)std::log2(Fs / out_rate) - (int)std::log2(Fs / main_vfo_out_rate));
        }
        v->setFilterBandwidth(toInt(p + "filter_bandwidth"));
        v->setGain(toFloat(p + "gain") / 100);
        v->setMixerFreq((center - main_vfo_freq) - vfo_freq);
        v->setFs(main_vfo_out_rate);
        v->setCompressonStyle(1);
        v->init(main_vfo_out_rate / P->bufsplit, true, lateDecimate);
        P->subs[(size_t)main_idx].push_back(v);
    }
    return P;
}

inline std::unique_ptr<Profile> load_profile_file(const std::string &path)
{
    std::ifstream f(path);
    if (!f)
        throw std::runtime_error("Given settings ini file doesn't exist: " + path); // mainwindow.cpp:20-25
    return load_profile(f);
}

} // namespace sdrx_host
