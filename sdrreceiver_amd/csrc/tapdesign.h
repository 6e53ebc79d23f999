// tapdesign.h -- init-time filter design on the host (product code; no dependency on oracle/).
//
// Restates, from the algorithm, the only tap designs vfo::init ever requests:
//   firfilter::low_pass(gain, fs, fc, tw, WIN_HAMMING)  gnuradio/firfilter.cpp:64-119,212-220
//   FIRHilbert::FIRHilbert(125, samplesOut)              jonti/dsp.cpp:184-217
//   Oscillator rotation                                  oscillator.cpp:9-11
// with the reference's exact mix of double sub-expressions and float stores, so the taps are
// bit-identical to the reference's (checked against golden vectors in tests/).
#pragma once
#include <cmath>
#include <vector>

namespace sdrx {

#ifndef SDRX_PI
#define SDRX_PI 3.14159265358979323846264338327950288
#endif

// firfilter::sanity_check_1f (firfilter.cpp:122-134): the what() text of the std::out_of_range it
// throws for this specification, or nullptr if it passes.
inline const char *low_pass_rejection(double fs, double fc, double tw)
{
    if (fs <= 0.0)
        return "firdes check failed: sampling_freq > 0";
    if (fc <= 0.0 || fc > fs / 2)
        return "firdes check failed: 0 < fa <= sampling_freq / 2";
    if (tw <= 0)
        return "firdes check failed: transition_width > 0";
    return nullptr;
}

// Returns false where firfilter::sanity_check_1f would throw.
inline bool design_low_pass(double gain, double fs, double fc, double tw, std::vector<float> &taps)
{
    if (low_pass_rejection(fs, fc, tw) || !(fs > 0.0) || !(tw > 0)) // (the second half also catches NaNs)
        return false;
    int ntaps = (int)(53.0 * fs / (22.0 * tw)); // max_attenuation(HAMMING) = 53, compute_ntaps 108-119
    if ((ntaps & 1) == 0)
        ntaps++;
    std::vector<float> w((size_t)ntaps);
    const float Mw = (float)(ntaps - 1);
    for (int n = 0; n < ntaps; ++n) // hamming(), 212-220: double expression, float store
        w[(size_t)n] = (float)(0.54 - 0.46 * std::cos((2 * SDRX_PI * n) / Mw));
    taps.assign((size_t)ntaps, 0.f);
    const int M = (ntaps - 1) / 2;
    const double fwT0 = 2 * SDRX_PI * fc / fs;
    for (int n = -M; n <= M; ++n) {
        if (n == 0)
            taps[(size_t)(n + M)] = (float)(fwT0 / SDRX_PI * w[(size_t)(n + M)]);
        else
            taps[(size_t)(n + M)] = (float)(std::sin(n * fwT0) / (n * SDRX_PI) * w[(size_t)(n + M)]);
    }
    double fmax = taps[(size_t)M]; // DC gain from the float-stored taps, summed in double
    for (int n = 1; n <= M; ++n)
        fmax += 2 * taps[(size_t)(n + M)];
    gain /= fmax;
    for (int i = 0; i < ntaps; ++i)
        taps[(size_t)i] = (float)(taps[(size_t)i] * gain);
    return true;
}

// 125-tap Hilbert transformer normalised to unit energy; `fs` is the reference's odd choice of
// samplesOut (vfo.cpp:137).  The square root is the FLOAT overload in the reference (dsp.cpp is
// C++ with `using namespace std`), widened afterwards.
inline void design_hilbert(int len, int fs, std::vector<float> &taps)
{
    std::vector<float> tmp((size_t)len);
    float sumsq = 0;
    for (int n = 0; n < len; ++n) {
        if (n == len / 2)
            tmp[(size_t)n] = 0;
        else
            tmp[(size_t)n] = (float)(fs / (SDRX_PI * (n - len / 2)) * (1 - std::cos(SDRX_PI * (n - len / 2))));
        sumsq += tmp[(size_t)n] * tmp[(size_t)n];
    }
    const double g = (double)std::sqrt(sumsq);
    taps.resize((size_t)len);
    for (int i = 0; i < len; ++i)
        taps[(size_t)i] = (float)(tmp[(size_t)(len - i - 1)] / g);
}

inline void nco_rotation(double fs, double f, float &rc, float &rs)
{
    const double angle = 2.0 * SDRX_PI * f / fs;
    rc = (float)std::cos(angle);
    rs = (float)std::sin(angle);
}

} // namespace sdrx
