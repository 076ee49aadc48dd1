// Stream-copy ceiling probe for MI355X: several copy kernel shapes over 1 GiB buffers.  hipcc --offload-arch=gfx950 -O3 tools/copy_probe.hip -o gpurun_out/copy_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int U>
__global__ void __launch_bounds__(256) copy_strided(double2 *__restrict__ dst, const double2 *__restrict__ src, size_t n)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        double2 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = src[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) dst[i + u * stride] = v[u];
    }
    for (; i < n; i += stride) dst[i] = src[i];
}

// each workgroup owns contiguous tiles of U*256 elements
template <int U, bool NT>
__global__ void __launch_bounds__(256) copy_tiled(double2 *__restrict__ dst, const double2 *__restrict__ src, size_t n)
{
    const size_t tile = (size_t)U * 256, ntile = n / tile;
    for (size_t t = blockIdx.x; t < ntile; t += gridDim.x) {
        const size_t base = t * tile + threadIdx.x;
        typedef double v2d __attribute__((ext_vector_type(2)));
        const v2d *s2 = reinterpret_cast<const v2d *>(src);
        v2d *d2 = reinterpret_cast<v2d *>(dst);
        v2d v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(&s2[base + u * 256]) : s2[base + u * 256];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (NT) __builtin_nontemporal_store(v[u], &d2[base + u * 256]);
            else d2[base + u * 256] = v[u];
        }
    }
}

template <typename F>
static double timeit(F f, int reps)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    f();
    hipEventRecord(a);
    for (int r = 0; r < reps; ++r) f();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}

int main()
{
    const size_t bytes = (size_t)1 << 30, n = bytes / sizeof(double2);
    double2 *src, *dst;
    hipMalloc(&src, bytes); hipMalloc(&dst, bytes);
    hipMemset(src, 1, bytes); hipMemset(dst, 0, bytes);
    const int grids[] = {256, 512, 1024, 2048, 4096, 8192, 16384, 65536};
    auto report = [&](const char *name, int g, double ms) { printf("%-28s grid %6d  %8.1f us  %7.1f GB/s\n", name, g, ms * 1e3, 2.0 * bytes / (ms * 1e-3) / 1e9); };
    for (int g : grids) {
        report("strided U=4", g, timeit([&] { hipLaunchKernelGGL(copy_strided<4>, dim3(g), dim3(256), 0, 0, dst, src, n); }, 10));
        report("tiled U=4", g, timeit([&] { hipLaunchKernelGGL((copy_tiled<4, false>), dim3(g), dim3(256), 0, 0, dst, src, n); }, 10));
        report("tiled U=8", g, timeit([&] { hipLaunchKernelGGL((copy_tiled<8, false>), dim3(g), dim3(256), 0, 0, dst, src, n); }, 10));
        report("tiled U=4 nt", g, timeit([&] { hipLaunchKernelGGL((copy_tiled<4, true>), dim3(g), dim3(256), 0, 0, dst, src, n); }, 10));
        report("tiled U=8 nt", g, timeit([&] { hipLaunchKernelGGL((copy_tiled<8, true>), dim3(g), dim3(256), 0, 0, dst, src, n); }, 10));
    }
    report("hipMemcpyDtoD", 0, timeit([&] { hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, 0); }, 10));
    // read-only and write-only ceilings
    report("hipMemset (write only, x0.5)", 0, timeit([&] { hipMemsetAsync(dst, 0, bytes, 0); }, 10));
    hipFree(src); hipFree(dst);
    return 0;
}
