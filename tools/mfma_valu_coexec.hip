// Do f32-input MFMAs and f32 vector FMAs of two waves on ONE SIMD overlap on gfx950, or do they share the datapath?
// One 512-thread workgroup per CU: waves w and w + 4 share a SIMD. Each wave runs a fixed instruction count of one kind
// and reports its own elapsed shader cycles (s_memtime):
//   mode 0: all eight waves MFMA          (two MFMA waves per SIMD)
//   mode 1: all eight waves VALU          (two VALU waves per SIMD)
//   mode 2: waves 0..3 MFMA, 4..7 VALU    (one of each per SIMD)       <- the question
//   mode 3: waves 0..3 MFMA, 4..7 idle    (one MFMA wave per SIMD alone)
//   mode 4: waves 0..3 idle, 4..7 VALU    (one VALU wave per SIMD alone)
// If the pipes were separate, mode 2 would cost each wave what it costs alone (modes 3, 4); if the f32 MFMA runs on the
// vector FMA lanes, mode 2 costs each wave about the SUM.
// build: hipcc -O3 --offload-arch=gfx950 -o mfma_valu_coexec mfma_valu_coexec.hip ; run: ./mfma_valu_coexec
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef float f4v __attribute__((ext_vector_type(4)));
constexpr int ITERS = 2000;

__global__ __launch_bounds__(1024) void k(int mode, unsigned long long *out, float *sink) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool mfma = mode == 0 || ((mode == 2 || mode == 3) && wave < 4);
    const bool valu = mode == 1 || ((mode == 2 || mode == 4) && wave >= 4);
    float a = 1.0f + lane * 1e-3f, b = 1.0f - lane * 1e-3f;
    f4v c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    float v0 = a, v1 = b, v2 = a + 1, v3 = b + 1, v4 = a + 2, v5 = b + 2, v6 = a + 3, v7 = b + 3;
    __syncthreads();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (mfma) {
#pragma unroll 1
        for (int i = 0; i < ITERS; ++i) {          // 8 MFMAs, 4 independent accumulators
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c3, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, c3, 0, 0, 0);
        }
    } else if (valu) {
#pragma unroll 1
        for (int i = 0; i < ITERS; ++i) {          // 64 v_fma_f32, 8 independent chains
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                v0 = fmaf(v0, a, b); v1 = fmaf(v1, a, b); v2 = fmaf(v2, a, b); v3 = fmaf(v3, a, b);
                v4 = fmaf(v4, a, b); v5 = fmaf(v5, a, b); v6 = fmaf(v6, a, b); v7 = fmaf(v7, a, b);
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0 && wave < 8 && blockIdx.x < 256) { out[blockIdx.x * 8 + wave] = t1 - t0; out[2048 + blockIdx.x * 8 + wave] = r1 - r0; }
    const float r = c0[0] + c1[1] + c2[2] + c3[3] + v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
    if (r == 12345.678f) sink[0] = r;
}

int main() {
    unsigned long long *d; float *s;
    hipMalloc(&d, 2 * 256 * 8 * sizeof(unsigned long long)); hipMalloc(&s, 4);
    std::vector<unsigned long long> h(2 * 256 * 8);
    const char *names[5] = {"8 MFMA waves", "8 VALU waves", "4 MFMA + 4 VALU (SIMD partners)", "4 MFMA waves alone", "4 VALU waves alone"};
    // occupancy check of the MFMA rate: all-MFMA blocks of 4 / 8 / 16 waves, one or two blocks per CU
    for (int cfg = 0; cfg < 5; ++cfg) {
        const int thr[5] = {256, 512, 1024, 512, 256}, grid[5] = {256, 256, 256, 512, 1024};
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k, dim3(grid[cfg]), dim3(thr[cfg]), 0, 0, 0, d, s);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k, dim3(grid[cfg]), dim3(thr[cfg]), 0, 0, 0, d, s);
        hipEventRecord(e1, 0); hipDeviceSynchronize();
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        const double flop = (double)grid[cfg] * (thr[cfg] / 64) * ITERS * 8.0 * 2048.0;
        printf("all-MFMA, %4d blocks x %4d threads: %8.1f us by events -> %6.1f TFLOP/s\n", grid[cfg], thr[cfg], ms * 1e3, flop / (ms * 1e-3) / 1e12);
    }
    for (int rep = 0; rep < 2; ++rep)
        for (int mode = 0; mode < 5; ++mode) {
            hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, mode, d, s);
            hipDeviceSynchronize();
            hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
            std::vector<double> m, v, rt, tk;
            for (int i = 0; i < 2048; ++i) if (h[i]) { tk.push_back((double)h[i]); rt.push_back((double)h[2048 + i]); }
            std::sort(tk.begin(), tk.end()); std::sort(rt.begin(), rt.end());
            const double ghz = tk[tk.size() / 2] / (rt[rt.size() / 2] * 10.0);     // s_memtime ticks per ns (realtime = 100 MHz)
            for (int b = 0; b < 256; ++b) for (int w = 0; w < 8; ++w) (w < 4 ? m : v).push_back((double)h[b * 8 + w]);
            std::sort(m.begin(), m.end()); std::sort(v.begin(), v.end());
            if (rep == 1)
                printf("mode %d  %-34s waves 0-3: %8.1f cycles per MFMA (8 per iteration)   waves 4-7: %6.2f cycles per v_fma (64 per iteration)\n",
                       mode, names[mode], m[m.size() / 2] / (ITERS * 8.0), v[v.size() / 2] / (ITERS * 64.0));
            if (rep == 1) printf("        s_memtime runs at %.3f GHz against the 100 MHz real-time counter\n", ghz);
        }
    return 0;
}
