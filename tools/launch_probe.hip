// launch_probe.hip -- what does a launch of N small workgroups cost on MI355X before any work?
// Kernels: empty / with dynamic LDS / with a high VGPR count / with scratch / with a dependent
// 3-load chain, each timed back to back (HIP events around 200 launches) for several grids.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(64) void k_empty(int *p) { if (p && threadIdx.x == 9999) *p = 1; }

__global__ __launch_bounds__(64) void k_lds(int *p)
{
    extern __shared__ int sm[];
    sm[threadIdx.x] = threadIdx.x;
    __syncthreads();
    if (p && sm[(threadIdx.x + 1) & 63] == 9999) *p = 1;
}

// ~96 live VGPRs
__global__ __launch_bounds__(64, 5) void k_vgpr(const float *in, float *out, int iters)
{
    extern __shared__ int sm[];
    float a[88];
#pragma unroll
    for (int i = 0; i < 88; ++i) a[i] = threadIdx.x * 0.5f + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 88; ++i) a[i] = a[i] * 1.0001f + a[(i + 1) % 88];
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 88; ++i) s += a[i];
    sm[threadIdx.x] = (int)s;
    if (s == 12345.678f) out[0] = s;
}

// private array indexed at run time -> scratch
__global__ __launch_bounds__(64) void k_scratch(const int *idx, float *out)
{
    float a[16];
    for (int i = 0; i < 16; ++i) a[i] = i * 1.5f + threadIdx.x;
    int j = idx[0];
    a[j & 15] += 1.f;
    float s = a[(j + 3) & 15];
    if (s == 12345.678f) out[0] = s;
}

struct Desc { const int *next; int pad[6]; };
__global__ __launch_bounds__(64) void k_chain(const int *work, const Desc *descs, int *out)
{
    int w = work[blockIdx.x];          // work item
    const int *p = descs[w].next;      // descriptor
    int v = p[threadIdx.x];            // data
    if (v == 99999) out[0] = v;
}

template <typename F> static float time_us(F launch, int reps = 200)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 10; ++i) launch();
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms * 1000.f / reps;
}

int main()
{
    int *d; CK(hipMalloc(&d, 1 << 20)); CK(hipMemset(d, 0, 1 << 20));
    float *f; CK(hipMalloc(&f, 1 << 20));
    const int maxg = 65536;
    std::vector<int> hw(maxg); for (int i = 0; i < maxg; ++i) hw[i] = i % 1024;
    int *work; CK(hipMalloc(&work, maxg * 4)); CK(hipMemcpy(work, hw.data(), maxg * 4, hipMemcpyHostToDevice));
    std::vector<Desc> hd(1024); for (auto &x : hd) x.next = d;
    Desc *descs; CK(hipMalloc(&descs, 1024 * sizeof(Desc))); CK(hipMemcpy(descs, hd.data(), 1024 * sizeof(Desc), hipMemcpyHostToDevice));
    for (int grid : {256, 1152, 4608, 9216, 18432, 49152}) {
        float e = time_us([&] { hipLaunchKernelGGL(k_empty, dim3(grid), dim3(64), 0, 0, (int *)nullptr); });
        float e256 = time_us([&] { hipLaunchKernelGGL(k_empty, dim3(grid / 4), dim3(256), 0, 0, (int *)nullptr); });
        float l = time_us([&] { hipLaunchKernelGGL(k_lds, dim3(grid), dim3(64), 4352, 0, (int *)nullptr); });
        float l26 = time_us([&] { hipLaunchKernelGGL(k_lds, dim3(grid), dim3(64), 26000, 0, (int *)nullptr); });
        float v0 = time_us([&] { hipLaunchKernelGGL(k_vgpr, dim3(grid), dim3(64), 4352, 0, f, f, 0); });
        float v10 = time_us([&] { hipLaunchKernelGGL(k_vgpr, dim3(grid), dim3(64), 4352, 0, f, f, 10); });
        float v100 = time_us([&] { hipLaunchKernelGGL(k_vgpr, dim3(grid), dim3(64), 4352, 0, f, f, 100); });
        float s = time_us([&] { hipLaunchKernelGGL(k_scratch, dim3(grid), dim3(64), 0, 0, d, f); });
        float c = time_us([&] { hipLaunchKernelGGL(k_chain, dim3(grid), dim3(64), 0, 0, work, descs, d); });
        printf("grid %6d: empty64 %.1f  empty256(grid/4) %.1f  lds4k %.1f  lds26k %.1f  vgpr96 %.1f  vgpr96+10it %.1f  +100it %.1f  scratch %.1f  chain %.1f us\n",
               grid, e, e256, l, l26, v0, v10, v100, s, c);
    }
    return 0;
}
