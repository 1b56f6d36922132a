// Ablation probe for the detect stream kernel: compiles the library's kernel
// source with YH_ABLATE=n and times it alone on config-2 shaped data.
//   n=0 full kernel; 1 no phase 2; 2 no staging/parking and no phase 2; 3 as 2 plus no flush
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude -Iyolo-tf_amd/csrc -DYH_ABLATE=n tools/probe_detect.hip -o tools/probe_detect_n
#include "../yolo-tf_amd/csrc/yh_stream.hip"
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

static uint64_t rng_state = 88172645463325252ull;
static inline double urand() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return (rng_state >> 11) * (1.0 / 9007199254740992.0); }
static inline float nrand(float m, float s) { double u1 = urand() + 1e-12, u2 = urand(); return m + s * (float)(std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2)); }

int main()
{
    const int B = 64, C = 80, P = 3, K = C + 5, S = 3;
    const int grid[3] = {13, 26, 52};
    const int NSETS = 3;
    yh::StreamParams p; std::memset(&p, 0, sizeof p);
    p.S = S; p.B = B; p.C = C; p.P = P; p.K = K;
    p.magic_k = (uint32_t)(((uint64_t)1 << 32) / K) + 1u;
    p.thr = 0.5f; p.prefilter = -0.001f; p.cand_cap = 53235; p.sub_trips = 4;
    std::vector<std::vector<float *>> sets(NSETS, std::vector<float *>(S));
    uint32_t row_base = 0; size_t total_bytes = 0;
    for (int s = 0; s < S; ++s) {
        const uint32_t rows = grid[s] * grid[s] * P;
        p.sc[s].rows_img = rows; p.sc[s].span = rows * K; p.sc[s].grid_w = grid[s]; p.sc[s].row_base = row_base;
        p.sc[s].mult = 32.f;
        for (int q = 0; q < P; ++q) { p.sc[s].prior_hw[q][0] = 1.5f + q; p.sc[s].prior_hw[q][1] = 2.f + q; }
        row_base += rows;
        const size_t n = (size_t)B * rows * K;
        std::vector<float> h(n);
        for (size_t r = 0; r < (size_t)B * rows; ++r) {
            float *t = &h[r * K];
            for (int c = 0; c < C; ++c) t[c] = nrand(-6.f, 1.f);
            t[(int)(urand() * C) % C] = nrand(2.f, 1.f);
            t[C] = nrand(0, 1); t[C + 1] = nrand(0, 1); t[C + 2] = nrand(0, .5f); t[C + 3] = nrand(0, .5f); t[C + 4] = nrand(-4.f, 2.f);
        }
        for (int i = 0; i < NSETS; ++i) { CK(hipMalloc(&sets[i][s], n * 4)); CK(hipMemcpy(sets[i][s], h.data(), n * 4, hipMemcpyHostToDevice)); }
        total_bytes += n * 4;
    }
    p.N = row_base;
    CK(hipMalloc(&p.cand, (size_t)B * p.cand_cap * 8)); CK(hipMalloc(&p.img_count, B * 4)); CK(hipMalloc(&p.boxes, (size_t)B * p.N * 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int want : {1200, 2400, 3600, 5000, 8000}) {
        uint64_t total_slots = 0;
        for (int s = 0; s < S; ++s) total_slots += ((uint64_t)p.sc[s].span + 3) / 4 * B;
        uint64_t target = total_slots / want; if (target < 2048) target = 2048;
        uint32_t blocks = 0;
        for (int s = 0; s < S; ++s) {
            const uint64_t slots_img = ((uint64_t)p.sc[s].span + 3) / 4 + 1;
            uint64_t chunks = (slots_img + target / 2) / target; if (chunks < 1) chunks = 1;
            uint64_t cs = (slots_img + chunks - 1) / chunks; cs = (cs + 255) / 256 * 256;
            p.sc[s].chunk_slots = cs; p.sc[s].chunks_per_img = (slots_img + cs - 1) / cs; p.sc[s].blk_begin = blocks;
            blocks += p.sc[s].chunks_per_img * B;
        }
        auto launch = [&](int i) {
            for (int s = 0; s < S; ++s) p.sc[s].src = sets[i % NSETS][s];
            CK(hipMemsetAsync(p.img_count, 0, B * 4, 0));
            hipLaunchKernelGGL(yh::k_stream<yh::kModeDetect>, dim3(blocks), dim3(256), 0, 0, p);
        };
        for (int i = 0; i < 6; ++i) launch(i);
        CK(hipDeviceSynchronize());
        const int R = 30; float tot = 0;
        for (int i = 0; i < R; ++i) {
            for (int s = 0; s < S; ++s) p.sc[s].src = sets[i % NSETS][s];
            CK(hipMemsetAsync(p.img_count, 0, B * 4, 0));
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(yh::k_stream<yh::kModeDetect>, dim3(blocks), dim3(256), 0, 0, p);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); tot += ms;
        }
        std::vector<int> cnt(B); CK(hipMemcpy(cnt.data(), p.img_count, B * 4, hipMemcpyDeviceToHost));
        long sum = 0; for (int c : cnt) sum += c;
        printf("ablate=%d blocks=%5u  %8.2f us  %7.1f GB/s  candidates/img %.1f\n", YH_ABLATE, blocks, 1e3 * tot / R,
               total_bytes / (tot / R * 1e-3) / 1e9, (double)sum / B);
    }
    return 0;
}
