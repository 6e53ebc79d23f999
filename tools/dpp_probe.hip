// tools/dpp_probe.hip -- hardware probe: lane semantics of the GFX9 whole-wave DPP shifts on gfx950.
// Build: hipcc --offload-arch=gfx950 -O2 dpp_probe.hip -o dpp_probe ; run on an MI355X.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const int *in, int *out)
{
    int l = threadIdx.x;
    int v = in[l], carry = in[64 + l];
    out[l] = __builtin_amdgcn_update_dpp(carry, v, 0x138, 0xf, 0xf, false);        // wave_shr:1, old=carry
    out[64 + l] = __builtin_amdgcn_update_dpp(-1, v, 0x13C, 0xf, 0xf, false);      // wave_ror:1
    out[128 + l] = __builtin_amdgcn_update_dpp(carry, v, 0x130, 0xf, 0xf, false);  // wave_shl:1
    out[192 + l] = __builtin_amdgcn_update_dpp(-1, v, 0x134, 0xf, 0xf, false);     // wave_rol:1
}
int main()
{
    int h[128], o[256];
    for (int i = 0; i < 64; ++i) { h[i] = 100 + i; h[64 + i] = 900 + i; }
    int *di, *dout;
    hipMalloc(&di, sizeof h); hipMalloc(&dout, sizeof o);
    hipMemcpy(di, h, sizeof h, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, di, dout);
    hipMemcpy(o, dout, sizeof o, hipMemcpyDeviceToHost);
    const char *names[4] = {"wave_shr1(old=carry)", "wave_ror1", "wave_shl1(old=carry)", "wave_rol1"};
    int ok = 1;
    for (int t = 0; t < 4; ++t) {
        printf("%s:", names[t]);
        for (int i = 0; i < 64; ++i) printf(" %d", o[64 * t + i]);
        printf("\n");
    }
    for (int i = 0; i < 64; ++i) {
        ok &= o[i] == (i == 0 ? 900 : 100 + i - 1);
        ok &= o[64 + i] == 100 + ((i + 63) & 63);
        ok &= o[128 + i] == (i == 63 ? 963 : 100 + i + 1);
        ok &= o[192 + i] == 100 + ((i + 1) & 63);
    }
    printf("DPP_SEMANTICS_%s\n", ok ? "OK" : "MISMATCH");
    return ok ? 0 : 1;
}
