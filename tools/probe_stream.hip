// Read-bandwidth probe: what a streaming read of the benchmark's head tensors
// reaches on this chip with different launch shapes.  Standalone; not part of
// the library.  Build: hipcc --offload-arch=gfx950 -O3 tools/probe_stream.hip -o tools/probe_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int UNROLL>
__global__ __launch_bounds__(256) void k_chunked(const float4 *src, long long nslots, int chunk_slots, float *sink)
{
    const long long q_lo = (long long)blockIdx.x * chunk_slots;
    long long q_hi = q_lo + chunk_slots;
    if (q_hi > nslots) q_hi = nslots;
    const int n = (int)(q_hi - q_lo);
    const float4 *p = src + q_lo;
    float acc = 0.f;
    for (int it = 0; it < n; it += 256 * UNROLL) {
        float4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            int i = it + u * 256 + threadIdx.x;
            i = i < n ? i : n - 1;
            v[u] = p[i];
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc += fmaxf(fmaxf(v[u].x, v[u].y), fmaxf(v[u].z, v[u].w));
    }
    if (acc == 12345.678f) sink[0] = acc;
}

template <int UNROLL>
__global__ __launch_bounds__(256) void k_gridstride(const float4 *src, long long nslots, float *sink)
{
    float acc = 0.f;
    const long long stride = (long long)gridDim.x * 256 * UNROLL;
    for (long long base = (long long)blockIdx.x * 256 * UNROLL; base < nslots; base += stride) {
        float4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            long long i = base + u * 256 + threadIdx.x;
            i = i < nslots ? i : nslots - 1;
            v[u] = src[i];
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc += fmaxf(fmaxf(v[u].x, v[u].y), fmaxf(v[u].z, v[u].w));
    }
    if (acc == 12345.678f) sink[0] = acc;
}

int main()
{
    const long long bytes = 231678720;            // config 2: 64 images x 3,619,980 B
    const long long nslots = bytes / 16;
    const int NBUF = 4;
    std::vector<float4 *> bufs(NBUF);
    for (auto &b : bufs) { CK(hipMalloc(&b, bytes)); CK(hipMemset(b, 0x3c, bytes)); }
    float *sink; CK(hipMalloc(&sink, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time_it = [&](const char *name, auto launch) {
        for (int i = 0; i < 8; ++i) launch(bufs[i % NBUF]);
        CK(hipDeviceSynchronize());
        const int R = 40;
        CK(hipEventRecord(e0));
        for (int i = 0; i < R; ++i) launch(bufs[i % NBUF]);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-34s %8.2f us  %7.1f GB/s\n", name, 1e3 * ms / R, bytes / (ms / R * 1e-3) / 1e9);
    };
    char name[128];
    for (int cs : {2048, 4096, 8192, 16384}) {
        snprintf(name, sizeof name, "chunked u8 chunk=%d slots", cs);
        const int blocks = (int)((nslots + cs - 1) / cs);
        time_it(name, [&](float4 *b) { hipLaunchKernelGGL(k_chunked<8>, dim3(blocks), dim3(256), 0, 0, b, nslots, cs, sink); });
    }
    for (int cs : {1024, 2048, 4096, 8192}) {
        snprintf(name, sizeof name, "chunked u4 chunk=%d slots", cs);
        const int blocks = (int)((nslots + cs - 1) / cs);
        time_it(name, [&](float4 *b) { hipLaunchKernelGGL(k_chunked<4>, dim3(blocks), dim3(256), 0, 0, b, nslots, cs, sink); });
    }
    for (int bpc : {2, 4, 6, 8}) {
        snprintf(name, sizeof name, "gridstride u8 %d blocks/CU", bpc);
        time_it(name, [&](float4 *b) { hipLaunchKernelGGL(k_gridstride<8>, dim3(256 * bpc), dim3(256), 0, 0, b, nslots, sink); });
        snprintf(name, sizeof name, "gridstride u4 %d blocks/CU", bpc);
        time_it(name, [&](float4 *b) { hipLaunchKernelGGL(k_gridstride<4>, dim3(256 * bpc), dim3(256), 0, 0, b, nslots, sink); });
    }
    return 0;
}
