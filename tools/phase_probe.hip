// What one dependent "phase" of the latency-bound kernels costs on gfx950 with ONE wavefront (or one 256-lane workgroup) on an otherwise
// idle device: the DPP wavefront sum, fp64 sqrt, fp64 division, an LDS round trip, a barrier-synchronised LDS exchange.  Each primitive
// is run 1000 times in a dependent chain between two reads of the constant 100 MHz clock (s_memrealtime) and of the shader clock (s_memtime).
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O2 tools/phase_probe.hip -o /tmp/phase_probe && /tmp/phase_probe
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_add(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, true);
    return v + __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wsum_k(double v)
{
    v = dpp_add<0x111, 0xf>(v); v = dpp_add<0x112, 0xf>(v); v = dpp_add<0x114, 0xf>(v); v = dpp_add<0x118, 0xf>(v);
    v = dpp_add<0x142, 0xa>(v); v = dpp_add<0x143, 0xc>(v);
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}
template <int WHAT>
__global__ void probe(double *out, unsigned long long *t, double seed)
{
    __shared__ double lds[1024];
    double x = seed + threadIdx.x * 1e-3, y = 1.0;
    lds[threadIdx.x] = x;
    __syncthreads();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 1000; ++i) {
        if (WHAT == 0) x = wsum_k(x) * 1e-2 + 1.0;
        if (WHAT == 1) x = sqrt(x) + 1.0;
        if (WHAT == 2) x = y / x + 2.0;
        if (WHAT == 3) { lds[threadIdx.x] = x; x = lds[(threadIdx.x + 1) & 63] + 1e-9; }                       // one wavefront: in-order LDS queue, no barrier
        if (WHAT == 4) { lds[threadIdx.x] = x; __syncthreads(); x = lds[(threadIdx.x + 65) & (blockDim.x - 1)] + 1e-9; __syncthreads(); }
        if (WHAT == 5) x = x * 1.0000001 + 1e-9;                                                                 // one dependent fp64 FMA
        if (WHAT == 6) { double a = x, b = x + 1, c = x + 2, d = x + 3; a = a * 1.1 + b; b = b * 1.1 + c; c = c * 1.1 + d; d = d * 1.1 + a; x = (a + b) + (c + d); x = x * 1e-3 + 1; }
    }
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime(), c1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) { t[0] = r1 - r0; t[1] = c1 - c0; }
}
template <int WHAT>
void run(const char *name, int threads)
{
    double *d; unsigned long long *t, h[2];
    (void)hipMalloc(&d, 1024 * sizeof(double)); (void)hipMalloc(&t, 2 * sizeof(unsigned long long));
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(probe<WHAT>, dim3(1), dim3(threads), 0, 0, d, t, 2.0);
    (void)hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-44s %4d lanes: %7.1f ns per iteration, %7.1f shader-clock ticks -> %5.2f GHz\n", name, threads, h[0] * 10.0 / 1000.0, h[1] / 1000.0, (double)h[1] / (h[0] * 10.0));
    (void)hipFree(d); (void)hipFree(t);
}
// the shader clock a stream of small kernels runs at: `wgs` workgroups of 256 lanes spin on dependent FMAs for ~10 us, launched back to back
// (or with a host synchronisation after each launch: the "trickle" of a polling driver); workgroup 0 stamps both clocks
__global__ void spin(double *out, unsigned long long *t, int iters, int slot)
{
    double x = 1.0 + threadIdx.x * 1e-9;
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) x = x * 1.0000001 + 1e-9;
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime(), c1 = __builtin_amdgcn_s_memtime();
    if (x == 0.5) out[0] = x;
    if (blockIdx.x == 0 && threadIdx.x == 0) { t[2 * slot] = r1 - r0; t[2 * slot + 1] = c1 - c0; }
}
void clocks(const char *name, int wgs, bool sync_each)
{
    const int n = 2000;
    double *d; unsigned long long *t;
    (void)hipMalloc(&d, 8); (void)hipMalloc(&t, 2 * n * sizeof(unsigned long long));
    unsigned long long *h = new unsigned long long[2 * n];
    for (int i = 0; i < n; ++i) {
        hipLaunchKernelGGL(spin, dim3(wgs), dim3(256), 0, 0, d, t, 4000, i);
        if (sync_each) (void)hipDeviceSynchronize();
    }
    (void)hipMemcpy(h, t, 2 * n * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    printf("%-58s", name);
    const int at[5] = {0, 10, 100, 1000, 1999};
    for (int q = 0; q < 5; ++q) printf("  #%d: %5.1f us %4.2f GHz", at[q], h[2 * at[q]] * 0.01, (double)h[2 * at[q] + 1] / (h[2 * at[q]] * 10.0));
    printf("\n");
    delete[] h; (void)hipFree(d); (void)hipFree(t);
}
int main()
{
    clocks("64 workgroups, back to back", 64, false);
    clocks("64 workgroups, host synchronisation after each launch", 64, true);
    clocks("1 workgroup, back to back", 1, false);
    clocks("1 workgroup, host synchronisation after each launch", 1, true);
    clocks("2048 workgroups, back to back", 2048, false);
    run<5>("dependent fp64 FMA", 64);
    run<6>("four independent FMAs + sum", 64);
    run<0>("DPP wavefront sum (wsum_k) + FMA", 64);
    run<1>("fp64 sqrt + add", 64);
    run<2>("fp64 division + add", 64);
    run<3>("LDS write + read, one wavefront", 64);
    run<4>("LDS write, barrier, read, barrier", 256);
    run<4>("LDS write, barrier, read, barrier", 1024);
    return 0;
}
