// host/demo.cpp -- drives the C++ host layer (host/sdrx_host.hpp) the way the reference's
// MainWindow + sdrj do: read a profile INI, build the VFO tree through the vfo setters, feed
// frames to sdrj::demodData, receive every leaf's payload through the publish hook.
//
//   sdrx_demo <profile.ini> --dump              descriptors as JSON lines (no GPU needed)
//   sdrx_demo <profile.ini> --frames N [--u8] [--fft TOPIC] [--devices 0,1,...]
//                                               N synthetic LCG frames; one line per published
//                                               message: frame topic rate bytes fnv1a64(payload)
#include <cinttypes>
#include <cstdio>
#include <iostream>

#include "sdrx_host.hpp"

using namespace sdrx_host;

static uint64_t fnv1a(const void *p, size_t n)
{
    const unsigned char *b = static_cast<const unsigned char *>(p);
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; ++i) {
        h ^= b[i];
        h *= 1099511628211ull;
    }
    return h;
}

int main(int argc, char **argv)
{
    if (argc < 3) {
        std::fprintf(stderr, "usage: %s profile.ini --dump | --frames N [--u8]\n", argv[0]);
        return 2;
    }
    try {
        auto P = load_profile_file(argv[1]);
        const std::string mode = argv[2];
        if (mode == "--dump") {
            std::printf("{\"fs\": %d, \"frame\": %d, \"bufsplit\": %d, \"correct_dc\": %d, \"n\": %zu}\n", P->fs, P->frame, P->bufsplit,
                        (int)P->correct_dc, P->all.size());
            // creation order with parents resolved: mains, then each main's subs
            for (size_t m = 0; m < P->mains.size(); ++m) {
                const sdrx_vfo_desc &d = P->mains[m]->d;
                std::printf("{\"main\": %zu, \"fs\": %d, \"d\": %d, \"mixer\": %.1f, \"usb\": %d, \"spb\": %d, \"scalecomp\": %d}\n", m, d.fs,
                            d.decimate_count, d.mixer_freq_hz, d.demod_usb, d.samples_per_buffer, d.scalecomp);
            }
            for (size_t m = 0; m < P->subs.size(); ++m)
                for (vfo *v : P->subs[m]) {
                    const sdrx_vfo_desc &d = v->d;
                    std::printf("{\"sub_of\": %zu, \"topic\": \"%s\", \"fs\": %d, \"d\": %d, \"late\": %d, \"mixer\": %.1f, \"bw\": %d, "
                                "\"gain\": %.9g, \"spb\": %d}\n",
                                m, v->topic.c_str(), d.fs, d.decimate_count, d.late_decimate, d.mixer_freq_hz, d.filter_bw_hz, (double)d.gain,
                                d.samples_per_buffer);
                }
            return 0;
        }
        if (mode != "--frames" || argc < 4)
            throw std::runtime_error("bad arguments");
        const int frames = std::atoi(argv[3]);
        bool u8 = false;
        std::string fft_topic;
        std::vector<int> devices;
        for (int a = 4; a < argc; ++a) {
            if (std::string(argv[a]) == "--u8")
                u8 = true;
            else if (std::string(argv[a]) == "--fft" && a + 1 < argc)
                fft_topic = argv[++a];
            else if (std::string(argv[a]) == "--devices" && a + 1 < argc) { // one tree sharded over several GPUs
                std::stringstream ss(argv[++a]);
                std::string tok;
                while (std::getline(ss, tok, ','))
                    devices.push_back(std::atoi(tok.c_str()));
            }
        }
        if (devices.empty())
            devices.push_back(0);
        sdrj radio(devices);
        radio.setVFOs(&P->mains);
        radio.setDCCorrection(P->correct_dc);
        int frame_no = 0;
        radio.setPublisher([&](const char topic[5], uint32_t rate, const void *buf, uint32_t len) {
            char t[6] = {0, 0, 0, 0, 0, 0};
            std::memcpy(t, topic, 5);
            std::printf("%d %s %u %u %016" PRIx64 "\n", frame_no, t, rate, len, fnv1a(buf, len));
        });
        // --fft TOPIC: what the GUI's spectrum selector does -- every vfo and the sdrj get fftVFOSlot(TOPIC)
        auto tap = [&](const char *who) {
            return [&frame_no, who](const std::vector<std::complex<float>> &d) {
                std::printf("fft %d %s %zu %016" PRIx64 "\n", frame_no, who, d.size(), fnv1a(d.data(), d.size() * sizeof(d[0])));
            };
        };
        if (!fft_topic.empty()) {
            for (auto &up : P->all) {
                vfo *v = up.get();
                v->fftData = tap(v->topic.c_str());
                v->fftVFOSlot(fft_topic);
            }
            radio.fftData = tap("Main");
            radio.fftVFOSlot(fft_topic);
        }
        // synthetic IQ of BASELINE.md: LCG x <- x*1664525 + 1013904223, component ((x >> 24) % 17) - 8
        uint32_t x = 1;
        std::vector<float> iq((size_t)2 * P->frame);
        std::vector<uint8_t> bytes((size_t)2 * P->frame);
        for (frame_no = 0; frame_no < frames; ++frame_no) {
            for (size_t i = 0; i < iq.size(); ++i) {
                x = x * 1664525u + 1013904223u;
                const int c = (int)((x >> 24) % 17u) - 8;
                iq[i] = (float)c;
                bytes[i] = (uint8_t)(c + 127);
            }
            if (u8)
                radio.demodBytes(bytes.data(), P->frame);
            else
                radio.demodData(iq.data(), (int)iq.size());
        }
        return 0;
    } catch (const std::exception &e) {
        std::fprintf(stderr, "sdrx_demo: %s\n", e.what());
        return 1;
    }
}
