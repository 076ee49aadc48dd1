// C ABI of libsmoqy_hip.so (include/smoqy_hip.h), part "bench": measurement aids (timers, launch sampling, copy ceiling, algorithmic bytes).
// gfx950 / ROCm only; there is no CPU path.  Split out of one api.hip in round 4; the handle and the shared internals are in ctx.h.
#include "ctx.h"

extern "C" {

// ---- measurement aids -------------------------------------------------------------------------------

int smoqy_timer_start(smoqy_ctx *c)
{
    CHECK_CTX(c);
    HIPCHK(c, hipEventRecord(c->ev0, c->stream));
    return 0;
}

int smoqy_timer_stop(smoqy_ctx *c, double *ms)
{
    CHECK_CTX(c);
    HIPCHK(c, hipEventRecord(c->ev1, c->stream));
    HIPCHK(c, hipEventSynchronize(c->ev1));
    float f = 0;
    HIPCHK(c, hipEventElapsedTime(&f, c->ev0, c->ev1));
    *ms = f;
    return 0;
}

int smoqy_matvec_timing(smoqy_ctx *c, int sample_every, int max_samples)
{
    CHECK_CTX(c);
    auto &T = c->mvt;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (sample_every < 0 || max_samples < 0 || max_samples > 65536) FAIL(c, 1, "invalid sampling parameters");
    while ((int)T.ev.size() < max_samples) {
        hipEvent_t a = nullptr, b = nullptr;
        HIPCHK(c, hipEventCreate(&a));
        HIPCHK(c, hipEventCreate(&b));
        T.ev.push_back({a, b});
    }
    {   // one (start, end) slot per workgroup per sampled launch, at most 1024 launches (16 MiB at 1024 workgroups)
        const int cap = std::min(max_samples, 1024), wgs = c->g.Lt * c->g.nsys;  // nchunk <= Lt
        if (cap > T.stamp_cap || wgs != T.stamp_wgs) {
            if (T.d_stamp) (void)hipFree(T.d_stamp);
            T.d_stamp = nullptr;
            T.stamp_cap = T.stamp_wgs = 0;
            if (cap > 0) {
                HIPCHK(c, hipMalloc(&T.d_stamp, 2 * (size_t)cap * wgs * sizeof(unsigned long long)));
                T.stamp_cap = cap;
                T.stamp_wgs = wgs;
            }
        }
        if (T.d_stamp) HIPCHK(c, hipMemset(T.d_stamp, 0, 2 * (size_t)T.stamp_cap * T.stamp_wgs * sizeof(unsigned long long)));
    }
    T.every = sample_every;
    T.seen = T.used = 0;
    return 0;
}

// the same sampled launches by the device's own clock: mean of (last workgroup's end - first workgroup's start), the interval
// rocprofv3 --kernel-trace reports for a dispatch.  Call BEFORE smoqy_matvec_timing_read (which ends the sampling).
int smoqy_matvec_timing_read_device(smoqy_ctx *c, double *avg_us, int *samples)
{
    CHECK_CTX(c);
    auto &T = c->mvt;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const int n = std::min(T.used, T.stamp_cap);
    const size_t per = 2 * (size_t)T.stamp_wgs;
    std::vector<unsigned long long> st(per * (size_t)std::max(n, 1));
    if (n) HIPCHK(c, hipMemcpy(st.data(), T.d_stamp, per * n * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    double sum = 0.0;
    int cnt = 0;
    for (int k = 0; k < n; ++k) {
        unsigned long long t0 = ~0ull, t1 = 0ull;
        for (int b = 0; b < T.stamp_wgs; ++b) {
            const unsigned long long s0 = st[per * k + 2 * b], s1 = st[per * k + 2 * b + 1];
            if (s0) t0 = std::min(t0, s0);   // slots of workgroups that never ran (smaller grid) stay 0
            t1 = std::max(t1, std::max(s0, s1));  // a workgroup that retired at entry (converged system) only has a start
        }
        if (t0 != ~0ull && t1 > t0) { sum += (double)(t1 - t0) * 0.01; ++cnt; }  // 100 MHz ticks -> µs
    }
    if (avg_us) *avg_us = cnt ? sum / cnt : 0.0;
    if (samples) *samples = cnt;
    return 0;
}

int smoqy_matvec_timing_read(smoqy_ctx *c, double *avg_us, int *samples)
{
    CHECK_CTX(c);
    auto &T = c->mvt;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    double sum = 0.0;
    for (int k = 0; k < T.used; ++k) {
        float f = 0;
        HIPCHK(c, hipEventElapsedTime(&f, T.ev[k].first, T.ev[k].second));
        sum += f;
    }
    if (avg_us) *avg_us = T.used ? 1e3 * sum / T.used : 0.0;
    if (samples) *samples = T.used;
    T.every = 0;
    T.seen = T.used = 0;
    return 0;
}

// per-kernel durations of the fused CG iteration inside real solves: the next `iterations` full-batch iterations on the handle's stream get an
// event in front of each of their four launches (MᵀM, forward τ-FFT, Chebyshev, inverse τ-FFT) and one behind the last
int smoqy_cg_iteration_timing(smoqy_ctx *c, int iterations)
{
    CHECK_CTX(c);
    if (iterations < 0 || iterations > 4096) FAIL(c, 1, "invalid number of iterations");
    HIPCHK(c, hipStreamSynchronize(c->stream));
    auto &IT = c->itt;
    while (IT.ev.size() < (size_t)5 * iterations) {
        hipEvent_t e = nullptr;
        HIPCHK(c, hipEventCreate(&e));
        IT.ev.push_back(e);
    }
    IT.want = iterations;
    IT.used = 0;
    return 0;
}

// us[0..3] = mean event-to-event time of the four launches over the sampled iterations (dependent launches on one stream: the kernel plus
// the hand-over to the next one); ends the sampling
int smoqy_cg_iteration_timing_read(smoqy_ctx *c, double *us, int *iterations)
{
    CHECK_CTX(c);
    if (!us) FAIL(c, 1, "us is NULL");
    auto &IT = c->itt;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    double sum[4] = {0, 0, 0, 0};
    for (int k = 0; k < IT.used; ++k)
        for (int q = 0; q < 4; ++q) {
            float f = 0;
            HIPCHK(c, hipEventElapsedTime(&f, IT.ev[(size_t)5 * k + q], IT.ev[(size_t)5 * k + q + 1]));
            sum[q] += f;
        }
    for (int q = 0; q < 4; ++q) us[q] = IT.used ? 1e3 * sum[q] / IT.used : 0.0;
    if (iterations) *iterations = IT.used;
    IT.want = IT.used = 0;
    return 0;
}

int smoqy_bench_matvec(smoqy_ctx *c, int op, int out, int in, int reps, double *ms)
{
    CHECK_CTX(c);
    if (int rc = check_vec(c, out)) return rc;
    if (int rc = check_vec(c, in)) return rc;
    if (out == in) FAIL(c, 1, "bench_matvec needs distinct vectors");
    if (int rc = matvec_dev(c, op, c->vecs[out], c->vecs[in], nullptr, nullptr, 0, c->g.nsys)) return rc;  // warm-up
    HIPCHK(c, hipEventRecord(c->ev0, c->stream));
    for (int r = 0; r < reps; ++r)
        if (int rc = matvec_dev(c, op, c->vecs[out], c->vecs[in], nullptr, nullptr, 0, c->g.nsys)) return rc;
    HIPCHK(c, hipEventRecord(c->ev1, c->stream));
    HIPCHK(c, hipEventSynchronize(c->ev1));
    float f = 0;
    HIPCHK(c, hipEventElapsedTime(&f, c->ev0, c->ev1));
    *ms = f;
    return check_launch(c, "bench_matvec");
}

// device stream-copy ceiling: `reps` copies of `bytes` bytes (src -> dst, both allocated here, far larger than the 256 MiB
// Infinity Cache when bytes >= 1 GiB) between two HIP events on the handle's stream; moved bytes = 2 * bytes per copy
int smoqy_bench_copy(smoqy_ctx *c, size_t bytes, int reps, double *ms)
{
    CHECK_CTX(c);
    if (bytes < 16 || reps < 1) FAIL(c, 1, "invalid copy benchmark parameters");
    const size_t n = bytes / sizeof(double2);
    double2 *src = nullptr, *dst = nullptr;
    HIPCHK(c, hipMalloc(&src, n * sizeof(double2)));
    if (hipMalloc(&dst, n * sizeof(double2)) != hipSuccess) { (void)hipFree(src); FAIL(c, 2, "out of device memory for the copy benchmark"); }
    int rc = 0;
    float f = 0;
    do {
        if (hipMemsetAsync(src, 1, n * sizeof(double2), c->stream) != hipSuccess) { rc = 2; break; }
        launch_stream_copy(c->stream, dst, src, n);  // warm-up (page faults, clocks)
        if (hipEventRecord(c->ev0, c->stream) != hipSuccess) { rc = 2; break; }
        for (int r = 0; r < reps; ++r) launch_stream_copy(c->stream, dst, src, n);
        if (hipEventRecord(c->ev1, c->stream) != hipSuccess || hipEventSynchronize(c->ev1) != hipSuccess) { rc = 2; break; }
        if (hipEventElapsedTime(&f, c->ev0, c->ev1) != hipSuccess) { rc = 2; break; }
    } while (0);
    (void)hipFree(src);
    (void)hipFree(dst);
    if (rc) FAIL(c, rc, "HIP error in the copy benchmark: %s", hipGetErrorString(hipGetLastError()));
    *ms = f;
    return check_launch(c, "bench_copy");
}

int smoqy_algorithmic_bytes(const smoqy_ctx *c, int op, double *bytes)
{
    CHECK_CTX(c);
    const Geometry &g = c->g;
    const double V = (double)g.Lt * g.N;
    const double S = 16.0 * V, F = 8.0 * V + 16.0 * g.Lt * g.Nh;
    const double one = g.nsys * 2.0 * S + g.nw * F;
    *bytes = (op == SMOQY_OP_MTM || op == SMOQY_OP_MMT) ? 2.0 * one : one;
    return 0;
}


}  // extern "C"
