// Measurement aid: how fast can one float per row (the objectness logit, stride K floats) be
// read out of a head tensor, compared with reading the whole tensor?  Config 2 shape.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int U>
__global__ __launch_bounds__(256) void k_strided(const float *src, long long rows, int K, int ch, float *sink)
{
    const long long base = ((long long)blockIdx.x * 256 * U) + threadIdx.x;
    float v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const long long r = base + (long long)u * 256;
        v[u] = r < rows ? src[r * K + ch] : 0.0f;
    }
    float acc = 0.0f;
#pragma unroll
    for (int u = 0; u < U; ++u) acc = fmaxf(acc, v[u]);
    if (acc == 12345.678f) sink[0] = acc;
}

template <int U>
__global__ __launch_bounds__(256) void k_strided_nt(const float *src, long long rows, int K, int ch, float *sink)
{
    const long long base = ((long long)blockIdx.x * 256 * U) + threadIdx.x;
    float v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const long long r = base + (long long)u * 256;
        v[u] = r < rows ? __builtin_nontemporal_load(src + r * K + ch) : 0.0f;
    }
    float acc = 0.0f;
#pragma unroll
    for (int u = 0; u < U; ++u) acc = fmaxf(acc, v[u]);
    if (acc == 12345.678f) sink[0] = acc;
}

// sc1 sc0: system-scope, bypassing the caches' allocation
template <int U>
__global__ __launch_bounds__(256) void k_strided_sc(const float *src, long long rows, int K, int ch, float *sink)
{
    const long long base = ((long long)blockIdx.x * 256 * U) + threadIdx.x;
    float v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const long long r = base + (long long)u * 256;
        const float *ptr = src + (r < rows ? r : rows - 1) * K + ch;
        asm volatile("global_load_dword %0, %1, off sc0 sc1 nt" : "=v"(v[u]) : "v"(ptr) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float acc = 0.0f;
#pragma unroll
    for (int u = 0; u < U; ++u) acc = fmaxf(acc, v[u]);
    if (acc == 12345.678f) sink[0] = acc;
}

int main()
{
    const int K = 85;
    const long long rows = 64LL * 10647, bytes = rows * K * 4;
    const int NBUF = 4;
    std::vector<float *> bufs(NBUF);
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
        printf("%-34s %8.2f us\n", name, 1e3 * ms / R);
    };
    char name[128];
#define RUN(U) do { \
        snprintf(name, sizeof name, "strided obj, %d rows/lane", U); \
        const int blocks = (int)((rows + 256LL * U - 1) / (256LL * U)); \
        time_it(name, [&](float *b) { hipLaunchKernelGGL(k_strided<U>, dim3(blocks), dim3(256), 0, 0, b, rows, K, K - 1, sink); }); \
    } while (0)
    RUN(1); RUN(2); RUN(4); RUN(8);
#define RUN2(KERN, LABEL, U) do { \
        snprintf(name, sizeof name, LABEL ", %d rows/lane", U); \
        const int blocks = (int)((rows + 256LL * U - 1) / (256LL * U)); \
        time_it(name, [&](float *b) { hipLaunchKernelGGL(KERN<U>, dim3(blocks), dim3(256), 0, 0, b, rows, K, K - 1, sink); }); \
    } while (0)
    RUN2(k_strided_nt, "nontemporal", 1); RUN2(k_strided_nt, "nontemporal", 2);
    RUN2(k_strided_sc, "sc0 sc1 nt", 1); RUN2(k_strided_sc, "sc0 sc1 nt", 2);
    return 0;
}
