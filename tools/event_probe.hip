// tools/event_probe.hip -- what do HIP events, cross-stream waits, graph fork/join and a copy
// stream cost on this runtime?  Decides how libsdrx pipelines frames (profiles/README.md).
//   hipcc --offload-arch=gfx950 -O2 -o tools/event_probe tools/event_probe.hip && tools/event_probe
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x)                                                                               \
    do {                                                                                    \
        hipError_t e_ = (x);                                                                \
        if (e_ != hipSuccess) {                                                             \
            fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__);     \
            exit(1);                                                                        \
        }                                                                                   \
    } while (0)

// every thread spins until `ticks` of the 100 MHz wall clock have passed: a kernel of known length
// whatever the grid
__global__ void spin(long long ticks, int *sink)
{
    const long long t0 = wall_clock64();
    int k = 0;
    while (wall_clock64() - t0 < ticks)
        ++k;
    if (k == -1)
        *sink = k;
}

static double now()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main()
{
    CK(hipSetDevice(0));
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    int *sink;
    CK(hipMalloc(&sink, 4));
    const int N = 300;
    const long long T20 = 2000, T40 = 4000, T5 = 500; // 20 us, 40 us, 5 us
    const dim3 full(2048), part(256), blk(256);
    std::vector<hipEvent_t> evd(16), evn(16);
    for (auto &e : evd)
        CK(hipEventCreate(&e));
    for (auto &e : evn)
        CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));

    auto run = [&](const char *name, auto body) {
        for (int i = 0; i < 20; ++i)
            body(i);
        CK(hipDeviceSynchronize());
        const double t0 = now();
        for (int i = 0; i < N; ++i)
            body(i);
        CK(hipDeviceSynchronize());
        const double us = (now() - t0) / N * 1e6;
        printf("%-78s %8.2f us / iteration\n", name, us);
        return us;
    };

    run("1  one 20 us kernel per iteration, stream A", [&](int) { hipLaunchKernelGGL(spin, full, blk, 0, s1, T20, sink); });
    run("2  + hipEventRecord (default flags) after it", [&](int i) {
        hipLaunchKernelGGL(spin, full, blk, 0, s1, T20, sink);
        CK(hipEventRecord(evd[i & 15], s1));
    });
    run("3  + hipEventRecord (hipEventDisableTiming) after it", [&](int i) {
        hipLaunchKernelGGL(spin, full, blk, 0, s1, T20, sink);
        CK(hipEventRecord(evn[i & 15], s1));
    });
    run("4  + record (no timing) + stream B waits + 5 us kernel on B (a copy stand-in)", [&](int i) {
        hipLaunchKernelGGL(spin, full, blk, 0, s1, T20, sink);
        CK(hipEventRecord(evn[i & 15], s1));
        CK(hipStreamWaitEvent(s2, evn[i & 15], 0));
        hipLaunchKernelGGL(spin, part, blk, 0, s2, T5, sink);
    });
    run("5  chain A:20us -> B:20us -> A ... through events (2 hops per iteration; ideal 40)", [&](int i) {
        hipLaunchKernelGGL(spin, full, blk, 0, s1, T20, sink);
        CK(hipEventRecord(evn[(2 * i) & 15], s1));
        CK(hipStreamWaitEvent(s2, evn[(2 * i) & 15], 0));
        hipLaunchKernelGGL(spin, full, blk, 0, s2, T20, sink);
        CK(hipEventRecord(evn[(2 * i + 1) & 15], s2));
        CK(hipStreamWaitEvent(s1, evn[(2 * i + 1) & 15], 0));
    });
    run("6  two independent streams, one 20 us HALF-chip kernel each (ideal 20 if concurrent)", [&](int) {
        hipLaunchKernelGGL(spin, part, blk, 0, s1, T20, sink);
        hipLaunchKernelGGL(spin, part, blk, 0, s2, T20, sink);
    });
    run("7  A: 20us; A: 40us || B(after event): 20us half-chip; no join (ideal 60)", [&](int i) {
        hipLaunchKernelGGL(spin, full, blk, 0, s1, T20, sink);
        CK(hipEventRecord(evn[i & 15], s1));
        CK(hipStreamWaitEvent(s2, evn[i & 15], 0));
        hipLaunchKernelGGL(spin, part, blk, 0, s1, T40, sink);
        hipLaunchKernelGGL(spin, part, blk, 0, s2, T20, sink);
    });

    // The frame pipeline libsdrx would use: A runs root(f) + sub(f) (9 + 85 us), B runs demod(f)
    // (40 us) behind an event of A; sub(f+2) on A must not start before demod(f) on B is done --
    // an event that completed a whole frame earlier.  Half-chip grids so that A and B can co-run.
    {
        std::vector<hipEvent_t> ea(8), eb(8);
        for (auto &e : ea)
            CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        for (auto &e : eb)
            CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        const long long T9 = 900, T85 = 8500;
        run("17a A only: 9 us + 85 us kernels (half chip), nothing else (ideal 94)", [&](int) {
            hipLaunchKernelGGL(spin, part, blk, 0, s1, T9, sink);
            hipLaunchKernelGGL(spin, part, blk, 0, s1, T85, sink);
        });
        run("17b A: 9 + 85 + 40 us serial on one stream (ideal 134)", [&](int) {
            hipLaunchKernelGGL(spin, part, blk, 0, s1, T9, sink);
            hipLaunchKernelGGL(spin, part, blk, 0, s1, T85, sink);
            hipLaunchKernelGGL(spin, part, blk, 0, s1, T40, sink);
        });
        int it = 0;
        for (int i = 0; i < 8; ++i) { // every event recorded once so that the first waits are legal
            CK(hipEventRecord(ea[i], s1));
            CK(hipEventRecord(eb[i], s2));
        }
        CK(hipDeviceSynchronize());
        run("17c pipeline: A: 9; wait eB[f-2]; 85; rec eA[f] | B: wait eA[f]; 40; rec eB[f] (ideal 94)", [&](int) {
            const int f = it++;
            hipLaunchKernelGGL(spin, part, blk, 0, s1, T9, sink);
            if (f >= 2)
                CK(hipStreamWaitEvent(s1, eb[(f - 2) & 7], 0));
            hipLaunchKernelGGL(spin, part, blk, 0, s1, T85, sink);
            CK(hipEventRecord(ea[f & 7], s1));
            CK(hipStreamWaitEvent(s2, ea[f & 7], 0));
            hipLaunchKernelGGL(spin, part, blk, 0, s2, T40, sink);
            CK(hipEventRecord(eb[f & 7], s2));
        });
        CK(hipDeviceSynchronize());
        it = 0;
        run("17d the same without the B -> A wait (unsafe; isolates its cost)", [&](int) {
            const int f = it++;
            hipLaunchKernelGGL(spin, part, blk, 0, s1, T9, sink);
            hipLaunchKernelGGL(spin, part, blk, 0, s1, T85, sink);
            CK(hipEventRecord(ea[f & 7], s1));
            CK(hipStreamWaitEvent(s2, ea[f & 7], 0));
            hipLaunchKernelGGL(spin, part, blk, 0, s2, T40, sink);
            CK(hipEventRecord(eb[f & 7], s2));
        });
    }

    // graph: A(20) -> { B(40) || C(20) } -> join, captured from streams
    {
        hipGraph_t g;
        hipGraphExec_t ge;
        hipEvent_t f, j;
        CK(hipEventCreateWithFlags(&f, hipEventDisableTiming));
        CK(hipEventCreateWithFlags(&j, hipEventDisableTiming));
        CK(hipStreamBeginCapture(s1, hipStreamCaptureModeGlobal));
        hipLaunchKernelGGL(spin, full, blk, 0, s1, T20, sink);
        CK(hipEventRecord(f, s1));
        CK(hipStreamWaitEvent(s2, f, 0));
        hipLaunchKernelGGL(spin, part, blk, 0, s1, T40, sink);
        hipLaunchKernelGGL(spin, part, blk, 0, s2, T20, sink);
        CK(hipEventRecord(j, s2));
        CK(hipStreamWaitEvent(s1, j, 0));
        CK(hipStreamEndCapture(s1, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        run("8  the same fork/join as a captured hipGraph, one replay per iteration (ideal 60)", [&](int) { CK(hipGraphLaunch(ge, s1)); });
        // serial graph of three kernels (no fork): the replay floor
        hipGraph_t g2;
        hipGraphExec_t ge2;
        CK(hipStreamBeginCapture(s1, hipStreamCaptureModeGlobal));
        hipLaunchKernelGGL(spin, full, blk, 0, s1, T20, sink);
        hipLaunchKernelGGL(spin, part, blk, 0, s1, T40, sink);
        hipLaunchKernelGGL(spin, part, blk, 0, s1, T20, sink);
        CK(hipStreamEndCapture(s1, &g2));
        CK(hipGraphInstantiate(&ge2, g2, nullptr, nullptr, 0));
        run("9  serial graph 20 + 40 + 20 (ideal 80)", [&](int) { CK(hipGraphLaunch(ge2, s1)); });
        run("10 the same three launches eagerly on one stream (ideal 80)", [&](int) {
            hipLaunchKernelGGL(spin, full, blk, 0, s1, T20, sink);
            hipLaunchKernelGGL(spin, part, blk, 0, s1, T40, sink);
            hipLaunchKernelGGL(spin, part, blk, 0, s1, T20, sink);
        });
    }

    // payload egress: 15 MB D2H on stream B behind an event, kernels continue on A
    {
        const size_t bytes = 15400000;
        unsigned char *d, *h;
        CK(hipMalloc(&d, 2 * bytes));
        CK(hipHostMalloc(&h, 2 * bytes, hipHostMallocDefault));
        run("11 15.4 MB D2H alone on B (pinned)", [&](int i) { CK(hipMemcpyAsync(h + (i & 1) * bytes, d + (i & 1) * bytes, bytes, hipMemcpyDeviceToHost, s2)); });
        run("12 A: 140 us of kernels; record; B waits; B: 15.4 MB D2H (ideal = max)", [&](int i) {
            for (int k = 0; k < 7; ++k)
                hipLaunchKernelGGL(spin, full, blk, 0, s1, T20, sink);
            CK(hipEventRecord(evn[i & 15], s1));
            CK(hipStreamWaitEvent(s2, evn[i & 15], 0));
            CK(hipMemcpyAsync(h + (i & 1) * bytes, d + (i & 1) * bytes, bytes, hipMemcpyDeviceToHost, s2));
        });
        run("13 A: 140 us of kernels then the D2H on A itself (serial: sum)", [&](int i) {
            for (int k = 0; k < 7; ++k)
                hipLaunchKernelGGL(spin, full, blk, 0, s1, T20, sink);
            CK(hipMemcpyAsync(h + (i & 1) * bytes, d + (i & 1) * bytes, bytes, hipMemcpyDeviceToHost, s1));
        });
        // H2D of the raw frame: pinned vs pageable source
        const size_t raw = 384000 * 8;
        float *dr, *hp;
        CK(hipMalloc(&dr, raw));
        CK(hipHostMalloc(&hp, raw, hipHostMallocDefault));
        std::vector<float> pageable(raw / 4, 1.0f);
        run("14 3.07 MB H2D from pinned memory, async on A", [&](int) { CK(hipMemcpyAsync(dr, hp, raw, hipMemcpyHostToDevice, s1)); });
        run("15 3.07 MB H2D from pageable memory, async call on A", [&](int) { CK(hipMemcpyAsync(dr, pageable.data(), raw, hipMemcpyHostToDevice, s1)); });
        run("16 memcpy pageable -> pinned on the host + H2D from pinned", [&](int) {
            memcpy(hp, pageable.data(), raw);
            CK(hipMemcpyAsync(dr, hp, raw, hipMemcpyHostToDevice, s1));
        });
    }
    return 0;
}
