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

typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;
typedef __attribute__((address_space(3))) float lds_float_t;

// LDS-DMA, workgroup tiles of TILE_KB KiB between barriers, one LDS read per thread and tile.
template <int TILE_KB>
__global__ __launch_bounds__(256) void k_dma_block(const float4 *src, long long nslots, int tiles_per_block, float *sink)
{
    __shared__ __attribute__((aligned(16))) float tile[TILE_KB * 256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float acc = 0.f;
    for (int t = 0; t < tiles_per_block; ++t) {
        const long long q0 = ((long long)blockIdx.x * tiles_per_block + t) * (TILE_KB * 64);
        if (q0 >= nslots) break;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < TILE_KB / 4; ++j) {
            const int slot0 = (j * 4 + wave) * 64;
            long long q = q0 + slot0 + lane;
            q = q < nslots ? q : nslots - 1;
            __builtin_amdgcn_global_load_lds((gptr_t *)(src + q), (lptr_t *)(tile + slot0 * 4), 16, 0, 0);
        }
        __syncthreads();
        acc += tile[threadIdx.x * 85 % (TILE_KB * 256)];
    }
    if (acc == 12345.678f) sink[0] = acc;
}

// LDS-DMA, one wavefront per workgroup with a ring of NBUF tiles of NI KiB and counted waits.
template <int NI, int NBUF>
__global__ __launch_bounds__(64) void k_dma_wave(const float4 *src, long long nslots, int tiles_per_wave, float *sink)
{
    __shared__ __attribute__((aligned(16))) float tile[NBUF][NI * 256];
    const int lane = threadIdx.x;
    float acc = 0.f;
    const long long qb = (long long)blockIdx.x * tiles_per_wave * (NI * 64);
    auto issue = [&](int t, float *tb) {
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            long long q = qb + (long long)t * (NI * 64) + j * 64 + lane;
            q = q < nslots ? q : nslots - 1;
            __builtin_amdgcn_global_load_lds((gptr_t *)(src + q), (lptr_t *)(tb + j * 256), 16, 0, 0);
        }
    };
#pragma unroll
    for (int i = 0; i < NBUF; ++i)
        if (i < tiles_per_wave) issue(i, tile[i]);
    int buf = 0;
    for (int t = 0; t < tiles_per_wave; ++t) {
        const int later = tiles_per_wave - 1 - t;
        if (later == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (NBUF == 2 || later == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI) : "memory");
        else if (NBUF == 3 || later == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NI) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * NI) : "memory");
        float v;
        const uint32_t addr = (uint32_t)(uintptr_t)(lds_float_t *)tile[buf] + (uint32_t)(lane * 85 % (NI * 256)) * 4u;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
        acc += v;
        if (t + NBUF < tiles_per_wave) issue(t + NBUF, tile[buf]);
        buf = buf + 1 == NBUF ? 0 : buf + 1;
    }
    if (acc == 12345.678f) sink[0] = acc;
}

// Register loads, one wavefront per workgroup walking `iters` groups of U KiB.
template <int U>
__global__ __launch_bounds__(64) void k_vgpr_wave(const float4 *src, long long nslots, int iters, float *sink)
{
    float acc = 0.f;
    const long long qb = (long long)blockIdx.x * iters * (U * 64);
    for (int it = 0; it < iters; ++it) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            long long q = qb + ((long long)it * U + u) * 64 + threadIdx.x;
            q = q < nslots ? q : nslots - 1;
            v[u] = src[q];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc += fmaxf(fmaxf(v[u].x, v[u].y), fmaxf(v[u].z, v[u].w));
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
    for (int tpb : {1, 2, 4, 7}) {
        snprintf(name, sizeof name, "dma block 32K tiles/block=%d", tpb);
        const int blocks = (int)((nslots + 2048LL * tpb - 1) / (2048LL * tpb));
        time_it(name, [&](float4 *b) { hipLaunchKernelGGL(k_dma_block<32>, dim3(blocks), dim3(256), 0, 0, b, nslots, tpb, sink); });
        snprintf(name, sizeof name, "dma block 16K tiles/block=%d", 2 * tpb);
        time_it(name, [&](float4 *b) { hipLaunchKernelGGL(k_dma_block<16>, dim3(blocks), dim3(256), 0, 0, b, nslots, 2 * tpb, sink); });
    }
    for (int tpw : {7, 14, 28}) {
        const int b4 = (int)((nslots + 256LL * tpw - 1) / (256LL * tpw));
        const int b8 = (int)((nslots + 512LL * tpw - 1) / (512LL * tpw));
        snprintf(name, sizeof name, "dma wave ni4 nbuf2 tiles/wave=%d", tpw);
        time_it(name, [&](float4 *b) { hipLaunchKernelGGL((k_dma_wave<4, 2>), dim3(b4), dim3(64), 0, 0, b, nslots, tpw, sink); });
        snprintf(name, sizeof name, "dma wave ni4 nbuf3 tiles/wave=%d", tpw);
        time_it(name, [&](float4 *b) { hipLaunchKernelGGL((k_dma_wave<4, 3>), dim3(b4), dim3(64), 0, 0, b, nslots, tpw, sink); });
        snprintf(name, sizeof name, "dma wave ni8 nbuf2 tiles/wave=%d", tpw);
        time_it(name, [&](float4 *b) { hipLaunchKernelGGL((k_dma_wave<8, 2>), dim3(b8), dim3(64), 0, 0, b, nslots, tpw, sink); });
        snprintf(name, sizeof name, "dma wave ni8 nbuf3 tiles/wave=%d", tpw);
        time_it(name, [&](float4 *b) { hipLaunchKernelGGL((k_dma_wave<8, 3>), dim3(b8), dim3(64), 0, 0, b, nslots, tpw, sink); });
    }
    for (int it : {1, 2, 4, 8}) {
        const int b8 = (int)((nslots + 512LL * it - 1) / (512LL * it));
        const int b16 = (int)((nslots + 1024LL * it - 1) / (1024LL * it));
        snprintf(name, sizeof name, "vgpr wave u8 iters=%d", it);
        time_it(name, [&](float4 *b) { hipLaunchKernelGGL(k_vgpr_wave<8>, dim3(b8), dim3(64), 0, 0, b, nslots, it, sink); });
        snprintf(name, sizeof name, "vgpr wave u16 iters=%d", it);
        time_it(name, [&](float4 *b) { hipLaunchKernelGGL(k_vgpr_wave<16>, dim3(b16), dim3(64), 0, 0, b, nslots, it, sink); });
    }
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
