// copy_overlap_probe.hip -- do short kernels on one stream run beside a 15 MB device-to-host copy on another?
// (profiles/README.md, round 5: the payload copy of frame f and the first kernels of frame f + 1 were seen NOT to overlap.)
// Stream B: 24 short kernels (~8-10 us each; pure ALU, or ALU + a 3 MB write, or one wave per CU only), stream A: the copy, as
//   none | a copy kernel of ours (64 workgroups, 16-byte units, plain or non-temporal stores) | hipMemcpyAsync (the runtime's
//   choice: a blit kernel here) | hsa_amd_memory_async_copy (an SDMA engine).
// Printed: when B's kernels were done and when the copy was, from a common start.
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(256) void k_spin(float *out, int iters, int write)
{
    float a = threadIdx.x * 0.25f, b = blockIdx.x * 0.5f;
    for (int i = 0; i < iters; ++i) {
        a = a * 1.0001f + b;
        b = b * 0.9999f + a;
    }
    if (write || a == 12345.678f)
        out[(size_t)blockIdx.x * 256 + threadIdx.x] = a + b;
}
template <bool NT>
__global__ __launch_bounds__(256) void k_copy(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        typedef unsigned v4u __attribute__((ext_vector_type(4)));
        const v4u v = reinterpret_cast<const v4u *>(src)[i];
        if (NT)
            __builtin_nontemporal_store(v, reinterpret_cast<v4u *>(dst) + i);
        else
            reinterpret_cast<v4u *>(dst)[i] = v;
    }
}
static hsa_agent_t g_gpu, g_cpu;
static int g_have_gpu = 0, g_have_cpu = 0;
static hsa_status_t on_agent(hsa_agent_t a, void *)
{
    hsa_device_type_t t;
    hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t);
    if (t == HSA_DEVICE_TYPE_GPU && !g_have_gpu) g_gpu = a, g_have_gpu = 1;
    if (t == HSA_DEVICE_TYPE_CPU && !g_have_cpu) g_cpu = a, g_have_cpu = 1;
    return HSA_STATUS_SUCCESS;
}
static double now_us()
{
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
int main()
{
    const size_t bytes = 15360000, n16 = bytes / 16;
    uint4 *d_src, *h_dst;
    float *d_out;
    CK(hipMalloc(&d_src, bytes));
    CK(hipMemset(d_src, 1, bytes));
    CK(hipHostMalloc(&h_dst, bytes, hipHostMallocDefault));
    CK(hipMalloc(&d_out, 1024 * 256 * sizeof(float)));
    hipStream_t A, B;
    CK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&B, hipStreamNonBlocking));
    hsa_init();
    hsa_iterate_agents(on_agent, nullptr);
    hsa_signal_t sig;
    hsa_signal_create(1, 0, nullptr, &sig);
    hipEvent_t ev;
    CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    const char *copies[] = {"no copy", "copy kernel, 64 workgroups", "copy kernel, non-temporal stores", "hipMemcpyAsync", "hsa_amd_memory_async_copy (SDMA)",
                            "hipMemcpyAsync behind a wait for B's event"};
    const char *works[] = {"ALU only, 1024 workgroups", "ALU + 1 MB written, 1024 workgroups", "ALU only, 2 workgroups"};
    for (int w = 0; w < 3; ++w)
        for (int c = 0; c < 6; ++c) {
            if (c == 4 && !(g_have_gpu && g_have_cpu))
                continue;
            double tb = 0, ta = 0;
            for (int rep = 0; rep < 4; ++rep) { // the last repetition counts
                CK(hipDeviceSynchronize());
                const double t0 = now_us();
                if (c == 1)
                    hipLaunchKernelGGL(k_copy<false>, dim3(64), dim3(256), 0, A, d_src, h_dst, n16);
                else if (c == 2)
                    hipLaunchKernelGGL(k_copy<true>, dim3(64), dim3(256), 0, A, d_src, h_dst, n16);
                else if (c == 3)
                    CK(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, A));
                else if (c == 5) { // (as the library does it: the copy stream waits for the event behind the frame's last kernel)
                    hipLaunchKernelGGL(k_spin, dim3(1024), dim3(256), 0, B, d_out, 400, 0);
                    CK(hipEventRecord(ev, B));
                    CK(hipStreamWaitEvent(A, ev, 0));
                    CK(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, A));
                }
                else if (c == 4) {
                    hsa_signal_store_relaxed(sig, 1);
                    hsa_status_t s = hsa_amd_memory_async_copy(h_dst, g_cpu, d_src, g_gpu, bytes, 0, nullptr, sig);
                    if (s != HSA_STATUS_SUCCESS) {
                        printf("hsa_amd_memory_async_copy: status %d\n", (int)s);
                        break;
                    }
                }
                for (int k = 0; k < 24; ++k)
                    hipLaunchKernelGGL(k_spin, dim3(w == 2 ? 2 : 1024), dim3(256), 0, B, d_out, w == 2 ? 1500 : 400, w == 1);
                CK(hipStreamSynchronize(B));
                tb = now_us() - t0;
                if (c == 4)
                    hsa_signal_wait_scacquire(sig, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_BLOCKED);
                else
                    CK(hipStreamSynchronize(A));
                ta = now_us() - t0;
            }
            printf("%-36s | %-42s | kernels done %7.1f us, copy done %7.1f us\n", works[w], copies[c], tb, ta);
        }
    return 0;
}
