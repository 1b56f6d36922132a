// Measurement aid: what does a 4-byte read that misses the L2 cost the memory system, a 128-byte
// line or a 64-byte half of it?  681,408 reads (the objectness column of config 2) laid out three
// ways: one per 128-byte line, one per 64-byte half line (half as many lines), and two per line
// (both halves).  If lines are the unit, the half-line layout takes half the time of the first and
// the both-halves layout the same time; if 64-byte sectors are, the other way round.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// lane i reads `per` floats at byte offsets i * stride + k * gap (gap >= 0; the buffers hold n * 340 + 4096 bytes,
// so stride <= 340 and gap <= 4092 stay inside)
__global__ __launch_bounds__(256) void k_read(const char *src, long long n, long long stride, int per, long long gap, float *sink)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    float acc = 0.0f;
    if (i < n) {
        float v[2] = {0.f, 0.f};
        for (int k = 0; k < per; ++k) v[k] = *reinterpret_cast<const float *>(src + i * stride + k * gap);
        acc = fmaxf(v[0], v[1]);
    }
    if (acc == 12345.678f) sink[0] = acc;
}

int main()
{
    const long long n = 64LL * 10647;
    const long long bytes = n * 340 + 4096;
    const int NBUF = 6;
    std::vector<char *> bufs(NBUF);
    for (auto &b : bufs) { CK(hipMalloc(&b, bytes)); CK(hipMemset(b, 0x3c, bytes)); }
    float *sink; CK(hipMalloc(&sink, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time_it = [&](const char *name, long long cnt, long long stride, int per, long long gap) {
        const int blocks = (int)((cnt + 255) / 256);
        for (int i = 0; i < 8; ++i) hipLaunchKernelGGL(k_read, dim3(blocks), dim3(256), 0, 0, bufs[i % NBUF], cnt, stride, per, gap, sink);
        CK(hipDeviceSynchronize());
        const int R = 60;
        CK(hipEventRecord(e0));
        for (int i = 0; i < R; ++i) hipLaunchKernelGGL(k_read, dim3(blocks), dim3(256), 0, 0, bufs[i % NBUF], cnt, stride, per, gap, sink);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-58s %8.2f us\n", name, 1e3 * ms / R);
    };
    time_it("objectness column: stride 340 B, 1 read per row", n, 340, 1, 0);
    time_it("one read per 128-B line (stride 128)", n, 128, 1, 0);
    time_it("one read per 64-B half line (stride 64): half the lines", n, 64, 1, 0);
    time_it("two reads per 128-B line, both halves (stride 128, gap 64)", n, 128, 2, 64);
    time_it("two reads per 128-B line, same half (stride 128, gap 32)", n, 128, 2, 32);
    time_it("one read per 256 B (stride 256): every other line", n, 256, 1, 0);
    return 0;
}
