// read-bandwidth probe: what a pure streaming read reaches on this part (tools probe, not product code)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
template <int NT, int UNROLL>
__global__ __launch_bounds__(256) void k_read(const f4 *__restrict__ p, size_t n, float *out)
{
    float acc = 0.f;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < n; i += UNROLL * stride) {
        f4 v[UNROLL];
#pragma unroll
        for (int k = 0; k < UNROLL; ++k) v[k] = NT ? __builtin_nontemporal_load(&p[i + k * stride]) : p[i + k * stride];
#pragma unroll
        for (int k = 0; k < UNROLL; ++k) acc += v[k].x + v[k].y + v[k].z + v[k].w;
    }
    for (; i < n; i += stride) { f4 v = p[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 1.2345f) out[0] = acc;
}
// contiguous chunk per workgroup (like the matrix kernels: a work item streams a contiguous range)
template <int NT>
__global__ __launch_bounds__(1024) void k_read_chunked(const f4 *__restrict__ p, size_t n, size_t per_wg, float *out)
{
    float acc = 0.f;
    const size_t b = (size_t)blockIdx.x * per_wg, e = b + per_wg < n ? b + per_wg : n;
    for (size_t i = b + threadIdx.x; i < e; i += 4 * 1024) {
        f4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { const size_t j = i + (size_t)k * 1024; v[k] = j < e ? (NT ? __builtin_nontemporal_load(&p[j]) : p[j]) : f4{0, 0, 0, 0}; }
#pragma unroll
        for (int k = 0; k < 4; ++k) acc += v[k].x + v[k].y + v[k].z + v[k].w;
    }
    if (acc == 1.2345f) out[0] = acc;
}
// the matrix kernels' access pattern without their arithmetic: per wave and 512-entry chunk two 16-byte loads per lane of a 32-byte
// lane record (vals), one 12-byte load per lane (slots) and 64 bytes of scalar loads (masks), from three arrays
template <int SPLIT, int WHAT>
__global__ __launch_bounds__(1024) void k_read3(const float *__restrict__ vals, const unsigned *__restrict__ slots, const unsigned long long *__restrict__ masks,
                                                 size_t nchunks, float *out)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float acc = 0.f;
    unsigned long long macc = 0;
    const size_t per = (nchunks + gridDim.x - 1) / gridDim.x, c0 = blockIdx.x * per, c1 = c0 + per < nchunks ? c0 + per : nchunks;
    for (size_t c = c0 + wave; c < c1; c += 16) {
        const float *vp = vals + c * 512;
        f4 a, b;
        if (SPLIT) { a = __builtin_nontemporal_load((const f4 *)(vp + lane * 8)); b = __builtin_nontemporal_load((const f4 *)(vp + lane * 8 + 4)); }
        else { a = __builtin_nontemporal_load((const f4 *)(vp + lane * 4)); b = __builtin_nontemporal_load((const f4 *)(vp + 256 + lane * 4)); }
        unsigned w0 = 0, w1 = 0, w2 = 0;
        if (WHAT == 3) { const unsigned *sp = slots + c * 192 + lane; w0 = __builtin_nontemporal_load(sp); w1 = __builtin_nontemporal_load(sp + 64); w2 = __builtin_nontemporal_load(sp + 128); }
        else if (WHAT >= 1) { const unsigned *sp = slots + c * 192 + lane * 3; w0 = __builtin_nontemporal_load(sp); w1 = __builtin_nontemporal_load(sp + 1); w2 = __builtin_nontemporal_load(sp + 2); }
        if (WHAT >= 2) {
            const unsigned long long *mp = masks + c * 8;
#pragma unroll
            for (int k = 0; k < 8; ++k) macc += mp[k];
        }
        acc += a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w + (float)(w0 ^ w1 ^ w2);
    }
    if (acc == 1.2345f || macc == 77) out[0] = acc;
}
// ONE interleaved stream: a chunk is a contiguous record [values 2048 B | slot stream SB bytes | masks 64 B]; the wave reads its record with
// the same instructions as k_read3 (two 16-byte loads per lane at a 16-byte lane stride, the slot stream as dwordx3 (SB = 768) or as
// dwordx2 + one byte per lane (SB = 576: 16-bit base + seven 8-bit deltas), masks by scalar loads)
template <int SB, int UNR>
__global__ __launch_bounds__(1024) void k_read_rec(const char *__restrict__ base, size_t nchunks, float *out)
{
    constexpr size_t REC = 2048 + (SB == 577 ? 576 : SB) + 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float acc = 0.f;
    unsigned long long macc = 0;
    const size_t per = (nchunks + gridDim.x - 1) / gridDim.x, c0 = blockIdx.x * per, c1 = c0 + per < nchunks ? c0 + per : nchunks;
#pragma unroll UNR
    for (size_t c = c0 + wave; c < c1; c += 16) {
        const char *r = base + c * REC;
        const float *vp = (const float *)r;
        const f4 a = __builtin_nontemporal_load((const f4 *)(vp + lane * 4)), b = __builtin_nontemporal_load((const f4 *)(vp + 256 + lane * 4));
        unsigned w0 = 0, w1 = 0, w2 = 0;
        if (SB == 768) {
            const unsigned *sp = (const unsigned *)(r + 2048) + lane * 3;
            w0 = __builtin_nontemporal_load(sp); w1 = __builtin_nontemporal_load(sp + 1); w2 = __builtin_nontemporal_load(sp + 2);
        } else if (SB == 576) {
            const unsigned *sp = (const unsigned *)(r + 2048) + lane * 2;
            w0 = __builtin_nontemporal_load(sp); w1 = __builtin_nontemporal_load(sp + 1);
            w2 = __builtin_nontemporal_load((const unsigned char *)(r + 2048 + 512) + lane);
        } else if (SB == 577) {      // (576 bytes; the seventh delta as a dword shared by four lanes)
            const unsigned *sp = (const unsigned *)(r + 2048) + lane * 2;
            w0 = __builtin_nontemporal_load(sp); w1 = __builtin_nontemporal_load(sp + 1);
            w2 = (__builtin_nontemporal_load((const unsigned *)(r + 2048 + 512) + (lane >> 2)) >> ((lane & 3) * 8)) & 0xffu;
        } else {                     // 512: exactly 8 bytes per lane
            const unsigned *sp = (const unsigned *)(r + 2048) + lane * 2;
            w0 = __builtin_nontemporal_load(sp); w1 = __builtin_nontemporal_load(sp + 1);
        }
        const unsigned long long *mp = (const unsigned long long *)(r + 2048 + SB);
#pragma unroll
        for (int k = 0; k < 8; ++k) macc += mp[k];
        acc += a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w + (float)(w0 ^ w1 ^ w2);
    }
    if (acc == 1.2345f || macc == 77) out[0] = acc;
}
// record-size sweep: [values 2048 B | slot stream as dwordx2 per lane (512 B) + optional dwordx1 per lane (256 B) | masks 64 B], padded to REC bytes
template <int REC, int THIRD>
__global__ __launch_bounds__(1024) void k_read_sweep(const char *__restrict__ base, size_t nchunks, float *out)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float acc = 0.f;
    unsigned long long macc = 0;
    const size_t per = (nchunks + gridDim.x - 1) / gridDim.x, c0 = blockIdx.x * per, c1 = c0 + per < nchunks ? c0 + per : nchunks;
    for (size_t c = c0 + wave; c < c1; c += 16) {
        const char *r = base + c * (size_t)REC;
        const float *vp = (const float *)r;
        const f4 a = __builtin_nontemporal_load((const f4 *)(vp + lane * 4)), b = __builtin_nontemporal_load((const f4 *)(vp + 256 + lane * 4));
        const unsigned *sp = (const unsigned *)(r + 2048) + lane * 2;
        unsigned w0 = __builtin_nontemporal_load(sp), w1 = __builtin_nontemporal_load(sp + 1), w2 = 0;
        if (THIRD) w2 = __builtin_nontemporal_load((const unsigned *)(r + 2048 + 512) + lane);
        const unsigned long long *mp = (const unsigned long long *)(r + 2048 + 512 + (THIRD ? 256 : 0));
#pragma unroll
        for (int k = 0; k < 8; ++k) macc += mp[k];
        acc += a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w + (float)(w0 ^ w1 ^ w2);
    }
    if (acc == 1.2345f || macc == 77) out[0] = acc;
}
// one launch that reads the same buffer `reps` times (no launch ramp between the passes)
__global__ __launch_bounds__(256) void k_read_loop(const f4 *__restrict__ p, size_t n, int reps, float *out)
{
    float acc = 0.f;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (int r = 0; r < reps; ++r) {
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i + 3 * stride < n; i += 4 * stride) {
            const f4 a = p[i], b = p[i + stride], c = p[i + 2 * stride], d = p[i + 3 * stride];
            acc += a.x + b.y + c.z + d.w;
        }
        __syncthreads();
    }
    if (acc == 1.2345f) out[0] = acc;
}
__global__ void k_copy(const f4 *__restrict__ a, f4 *__restrict__ b, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
int main()
{
    const size_t bytes = (size_t)48 << 30, n = bytes / 16;
    f4 *p, *q; float *out;
    CK(hipMalloc(&p, bytes)); CK(hipMalloc(&q, bytes)); CK(hipMalloc(&out, 4));
    CK(hipMemset(p, 1, bytes)); CK(hipMemset(q, 0, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char *name, auto launch, double nbytes) {
        launch(); hipDeviceSynchronize();
        float best = 1e30f;
        for (int r = 0; r < 5; ++r) { hipEventRecord(e0, 0); launch(); hipEventRecord(e1, 0); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; }
        printf("%-46s %8.3f ms  %7.1f GB/s\n", name, best, nbytes / best / 1e6);
    };
    for (int g : {1024, 2048, 4096, 8192, 16384}) {
        char nm[96];
        snprintf(nm, 96, "read f4 grid-stride x4   grid %5d", g); timeit(nm, [&]() { hipLaunchKernelGGL((k_read<0, 4>), dim3(g), dim3(256), 0, 0, p, n, out); }, (double)bytes);
        snprintf(nm, 96, "read f4 grid-stride x4 nt grid %5d", g); timeit(nm, [&]() { hipLaunchKernelGGL((k_read<1, 4>), dim3(g), dim3(256), 0, 0, p, n, out); }, (double)bytes);
        snprintf(nm, 96, "read f4 grid-stride x8 nt grid %5d", g); timeit(nm, [&]() { hipLaunchKernelGGL((k_read<1, 8>), dim3(g), dim3(256), 0, 0, p, n, out); }, (double)bytes);
    }
    for (int wgs : {256, 512, 1024, 4096}) {
        char nm[96];
        const size_t per = (n + wgs - 1) / wgs;
        snprintf(nm, 96, "read f4 contiguous per WG nt, %4d WGs x1024", wgs); timeit(nm, [&]() { hipLaunchKernelGGL((k_read_chunked<1>), dim3(wgs), dim3(1024), 0, 0, p, n, per, out); }, (double)bytes);
    }
    {
        const size_t nch = (size_t)16 << 20;        // 16 Mi chunks: 32 GB of values, 12 GB of slots, 1 GB of masks
        float *v; unsigned *sl; unsigned long long *mk;
        CK(hipMalloc(&v, nch * 2048)); CK(hipMalloc(&sl, nch * 768)); CK(hipMalloc(&mk, nch * 64));
        CK(hipMemset(v, 0, nch * 2048)); CK(hipMemset(sl, 0, nch * 768)); CK(hipMemset(mk, 0, nch * 64));
        const double by = (double)nch * (2048 + 768 + 64);
        for (int wgs : {256, 512, 4096}) {
            char nm[96];
            snprintf(nm, 96, "3 streams, 32-B lane records, %4d WGs", wgs); timeit(nm, [&]() { hipLaunchKernelGGL((k_read3<1, 2>), dim3(wgs), dim3(1024), 0, 0, v, sl, mk, nch, out); }, by);
            snprintf(nm, 96, "3 streams, 16-B lane stride, %4d WGs", wgs); timeit(nm, [&]() { hipLaunchKernelGGL((k_read3<0, 2>), dim3(wgs), dim3(1024), 0, 0, v, sl, mk, nch, out); }, by);
            snprintf(nm, 96, "3 streams, slots as 3 dword planes, %4d WGs", wgs); timeit(nm, [&]() { hipLaunchKernelGGL((k_read3<0, 3>), dim3(wgs), dim3(1024), 0, 0, v, sl, mk, nch, out); }, by);
            snprintf(nm, 96, "values + slots (no masks), %4d WGs", wgs); timeit(nm, [&]() { hipLaunchKernelGGL((k_read3<0, 1>), dim3(wgs), dim3(1024), 0, 0, v, sl, mk, nch, out); }, (double)nch * (2048 + 768));
            snprintf(nm, 96, "values only, %4d WGs", wgs); timeit(nm, [&]() { hipLaunchKernelGGL((k_read3<0, 0>), dim3(wgs), dim3(1024), 0, 0, v, sl, mk, nch, out); }, (double)nch * 2048);
        }
    }
    {
        const size_t nch = (size_t)16 << 20;
        char *rec;
        CK(hipMalloc(&rec, nch * 2880)); CK(hipMemset(rec, 0, nch * 2880));
        for (int wgs : {512, 4096}) {
            char nm[96];
            snprintf(nm, 96, "1 interleaved stream, 2880-B records, %4d WGs", wgs); timeit(nm, [&]() { hipLaunchKernelGGL((k_read_rec<768, 1>), dim3(wgs), dim3(1024), 0, 0, rec, nch, out); }, (double)nch * 2880);
            snprintf(nm, 96, "  same, 2 chunks in flight per wave, %4d WGs", wgs); timeit(nm, [&]() { hipLaunchKernelGGL((k_read_rec<768, 2>), dim3(wgs), dim3(1024), 0, 0, rec, nch, out); }, (double)nch * 2880);
            snprintf(nm, 96, "1 interleaved stream, 2688-B records, %4d WGs", wgs); timeit(nm, [&]() { hipLaunchKernelGGL((k_read_rec<576, 1>), dim3(wgs), dim3(1024), 0, 0, rec, nch, out); }, (double)nch * 2688);
            snprintf(nm, 96, "  same, 2 chunks in flight per wave, %4d WGs", wgs); timeit(nm, [&]() { hipLaunchKernelGGL((k_read_rec<576, 2>), dim3(wgs), dim3(1024), 0, 0, rec, nch, out); }, (double)nch * 2688);
            snprintf(nm, 96, "2688-B records, d7 as shared dword, %4d WGs", wgs); timeit(nm, [&]() { hipLaunchKernelGGL((k_read_rec<577, 1>), dim3(wgs), dim3(1024), 0, 0, rec, nch, out); }, (double)nch * 2688);
            snprintf(nm, 96, "2624-B records (8 B slots per lane), %4d WGs", wgs); timeit(nm, [&]() { hipLaunchKernelGGL((k_read_rec<512, 1>), dim3(wgs), dim3(1024), 0, 0, rec, nch, out); }, (double)nch * 2624);
            snprintf(nm, 96, "  same, 4 chunks in flight per wave, %4d WGs", wgs); timeit(nm, [&]() { hipLaunchKernelGGL((k_read_rec<576, 4>), dim3(wgs), dim3(1024), 0, 0, rec, nch, out); }, (double)nch * 2688);
        }
    }
    {
        const size_t nch = (size_t)16 << 20;
        char *rec;
        CK(hipMalloc(&rec, nch * 3072)); CK(hipMemset(rec, 0, nch * 3072));
        const int wgs = 4096;
        char nm[96];
#define SWEEP(REC, THIRD) snprintf(nm, 96, "sweep: %d-B records, %s", REC, THIRD ? "x4 x4 x2 x1 + masks" : "x4 x4 x2 + masks"); \
        timeit(nm, [&]() { hipLaunchKernelGGL((k_read_sweep<REC, THIRD>), dim3(wgs), dim3(1024), 0, 0, rec, nch, out); }, (double)nch * REC);
        SWEEP(2624, 0) SWEEP(2688, 0) SWEEP(2752, 0) SWEEP(2816, 0) SWEEP(2880, 0) SWEEP(3072, 0)
        SWEEP(2880, 1) SWEEP(2944, 1) SWEEP(3072, 1)
#undef SWEEP
    }
    // Infinity Cache (256 MiB): the same buffer read again and again - how fast is a re-read that can come from the memory-side cache?
    for (size_t mb : {32, 64, 96, 128, 160, 192, 256, 384, 1024}) {
        const size_t nb = mb << 20, ne = nb / 16;
        char nm[96];
        snprintf(nm, 96, "re-read of a %4zu MB buffer, plain loads", mb); timeit(nm, [&]() { hipLaunchKernelGGL((k_read<0, 8>), dim3(4096), dim3(256), 0, 0, p, ne, out); }, (double)nb);
        snprintf(nm, 96, "re-read of a %4zu MB buffer, nt loads", mb); timeit(nm, [&]() { hipLaunchKernelGGL((k_read<1, 8>), dim3(4096), dim3(256), 0, 0, p, ne, out); }, (double)nb);
    }
    for (size_t mb : {16, 64, 128, 192, 256, 512, 4096}) {
        const size_t nb = mb << 20, ne = nb / 16;
        const int reps = mb >= 4096 ? 4 : 40;
        char nm[96];
        snprintf(nm, 96, "one launch, %d passes over %zu MB", reps, mb);
        timeit(nm, [&]() { hipLaunchKernelGGL(k_read_loop, dim3(2048), dim3(256), 0, 0, p, ne, reps, out); }, (double)nb * reps);
    }
    // ... and a slab read once by one kernel (nt or plain), then by a second kernel: what the second reader sees
    for (size_t mb : {64, 128}) {
        const size_t nb = mb << 20, ne = nb / 16;
        char nm[96];
        for (int first_nt = 0; first_nt < 2; ++first_nt) {
            // walk through 8 GB so that every slab is cold for its first reader
            float best = 1e30f;
            for (int r = 0; r < 5; ++r) {
                const f4 *slab = p + (size_t)(r * 7 + first_nt * 3 + 1) * (ne * 4);
                if (first_nt) hipLaunchKernelGGL((k_read<1, 8>), dim3(4096), dim3(256), 0, 0, slab, ne, out);
                else hipLaunchKernelGGL((k_read<0, 8>), dim3(4096), dim3(256), 0, 0, slab, ne, out);
                hipEventRecord(e0, 0);
                hipLaunchKernelGGL((k_read<0, 8>), dim3(4096), dim3(256), 0, 0, slab, ne, out);
                hipEventRecord(e1, 0); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
            }
            snprintf(nm, 96, "second reader of a %zu MB slab (first: %s)", mb, first_nt ? "nt" : "plain");
            printf("%-46s %8.3f ms  %7.1f GB/s\n", nm, best, (double)nb / best / 1e6);
        }
    }
    timeit("copy f4 (read + write)", [&]() { hipLaunchKernelGGL(k_copy, dim3(8192), dim3(256), 0, 0, p, q, n); }, 2.0 * bytes);
    return 0;
}
