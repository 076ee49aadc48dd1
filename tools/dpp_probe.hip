// Prints which source lane a lane reads under the DPP controls the kernels rely on (gfx950):
//   0x134 wave_rol:1, 0x13C wave_ror:1 (kernels_kpm_wave.hip), 0x121 row_ror:1, 0x12F row_ror:15 (kernels_kpm.hip, kernels_fdm_own.hip).
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O2 tools/dpp_probe.hip -o /tmp/dpp_probe && /tmp/dpp_probe
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CTRL>
__global__ void probe(int *o) { o[threadIdx.x] = __builtin_amdgcn_update_dpp(-1, (int)threadIdx.x, CTRL, 0xf, 0xf, false); }
template <int CTRL>
void run(const char *name)
{
    int *d, h[64];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(probe<CTRL>, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-12s lane 0 <- %d, lane 1 <- %d, lane 15 <- %d, lane 16 <- %d, lane 62 <- %d, lane 63 <- %d\n", name, h[0], h[1], h[15], h[16], h[62], h[63]);
    hipFree(d);
}
int main()
{
    run<0x134>("wave_rol:1");
    run<0x13C>("wave_ror:1");
    run<0x121>("row_ror:1");
    run<0x12F>("row_ror:15");
    return 0;
}
