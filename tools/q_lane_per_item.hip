// q_lane_per_item.hip — feasibility micro-benchmark for a different loop A (open items):
// Q_k(s', .) with ONE LANE PER ITEM instead of one lane per 21 features. Every lane walks all features of its
// own item, the weights are wave-uniform (scalar loads -> SGPR operands of v_pk_fma_f32), so there are no
// tables in LDS, no LDS reads in the loop and no butterflies. A workgroup = 8 waves = 256 items: wave w takes
// items 64 (w & 3) .. +64 and half (w >> 2) of the 36 c12 values; halves are added through LDS.
// Prints cycles per workgroup pass (256 items x 1296 features x 5 actions) for comparison with the fused
// kernel's loop A (tools/stamp_report.py: "root: loop A" + "wait A"). Build: make -C tools.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));
constexpr int NF = 1296, NACT = 5;
#ifndef PIPELINED
#define PIPELINED 0
#endif

__device__ __forceinline__ v2f cmul(v2f a, v2f b) { return (v2f){a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }

__global__ __launch_bounds__(512, 2) void q_kernel(const float *__restrict__ W, const float4 *__restrict__ st,
                                                   float *__restrict__ q_out, unsigned long long *cyc, int reps) {
    __shared__ float s_half[256][NACT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = __builtin_amdgcn_readfirstlane(wave >> 2);      // wave-uniform, and the compiler knows it
    const int item = (wave & 3) * 64 + lane;
    const float4 s = st[blockIdx.x * 256 + item];
    // unit-complex powers Z_d^0..5 of the four state variables
    v2f Z[4][6];
    const float sv[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        float sn, cs;
        __sincosf(3.14159265f * sv[d], &sn, &cs);
        Z[d][0] = (v2f){1.0f, 0.0f};
        Z[d][1] = (v2f){cs, sn};
#pragma unroll
        for (int k = 2; k < 6; ++k) Z[d][k] = cmul(Z[d][k - 1], Z[d][1]);
    }
    // CD table of this item in registers: 36 complex values as re/im pairs over adjacent c34
    v2f cdre[18], cdim[18];
#pragma unroll
    for (int c3 = 0; c3 < 6; ++c3) {
#pragma unroll
        for (int c4 = 0; c4 < 6; c4 += 2) {
            const v2f p0 = cmul(Z[2][c3], Z[3][c4]), p1 = cmul(Z[2][c3], Z[3][c4 + 1]);
            cdre[(c3 * 6 + c4) >> 1] = (v2f){p0.x, p1.x};
            cdim[(c3 * 6 + c4) >> 1] = (v2f){p0.y, p1.y};
        }
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    v2f q[NACT];
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int a = 0; a < NACT; ++a) q[a] = (v2f){0.0f, 0.0f};
#if PIPELINED
        // chunks of 6 adjacent features x 5 actions = 30 scalars; the next chunk's scalar loads are issued before
        // the current chunk's 21 packed ops (double-buffered SGPR sets)
        const int ch0 = half * 108;                                        // 18 c12 values x 6 chunks
        float wn[NACT][6];
#pragma unroll
        for (int a = 0; a < NACT; ++a) {
#pragma unroll
            for (int u = 0; u < 6; ++u) wn[a][u] = W[a * NF + ch0 * 6 + u];
        }
        for (int c12 = half * 18; c12 < half * 18 + 18; ++c12) {          // wave-uniform
            const int c1 = c12 / 6, c2 = c12 - 6 * c1;
            v2f z0 = Z[0][0], z1 = Z[1][0];
#pragma unroll
            for (int k = 1; k < 6; ++k) { if (c1 == k) z0 = Z[0][k]; if (c2 == k) z1 = Z[1][k]; }
            const v2f ab = cmul(z0, z1);
            const v2f abre = {ab.x, ab.x}, abim = {-ab.y, -ab.y};
#pragma unroll
            for (int m = 0; m < 6; ++m) {
                float wc[NACT][6];
#pragma unroll
                for (int a = 0; a < NACT; ++a) {
#pragma unroll
                    for (int u = 0; u < 6; ++u) wc[a][u] = wn[a][u];
                }
                const int nxt = (c12 * 6 + m + 1 < half * 108 + 108) ? c12 * 6 + m + 1 : ch0;      // wave-uniform
#pragma unroll
                for (int a = 0; a < NACT; ++a) {
#pragma unroll
                    for (int u = 0; u < 6; ++u) wn[a][u] = W[a * NF + nxt * 6 + u];
                }
#pragma unroll
                for (int pp = 0; pp < 3; ++pp) {
                    const int p = 3 * m + pp;
                    const v2f phi = __builtin_elementwise_fma(abim, cdim[p], abre * cdre[p]);
#pragma unroll
                    for (int a = 0; a < NACT; ++a) {
                        const v2f w = {wc[a][2 * pp], wc[a][2 * pp + 1]};
                        q[a] = __builtin_elementwise_fma(w, phi, q[a]);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
#else
        for (int c12 = half * 18; c12 < half * 18 + 18; ++c12) {          // wave-uniform
            const int c1 = c12 / 6, c2 = c12 - 6 * c1;
            v2f z0 = Z[0][0], z1 = Z[1][0];
#pragma unroll
            for (int k = 1; k < 6; ++k) { if (c1 == k) z0 = Z[0][k]; if (c2 == k) z1 = Z[1][k]; }
            const v2f ab = cmul(z0, z1);
            const v2f abre = {ab.x, ab.x}, abim = {-ab.y, -ab.y};
            const float *Wc = W + c12 * 36;                                // wave-uniform address
#pragma unroll
            for (int p = 0; p < 18; ++p) {
                const v2f phi = __builtin_elementwise_fma(abim, cdim[p], abre * cdre[p]);
#pragma unroll
                for (int a = 0; a < NACT; ++a) {
                    const v2f w = {Wc[a * NF + 2 * p], Wc[a * NF + 2 * p + 1]};
                    q[a] = __builtin_elementwise_fma(w, phi, q[a]);
                }
            }
        }
    }
#endif
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (half == 1) {
#pragma unroll
        for (int a = 0; a < NACT; ++a) s_half[item][a] = q[a].x + q[a].y;
    }
    __syncthreads();
    if (half == 0) {
#pragma unroll
        for (int a = 0; a < NACT; ++a) q_out[(size_t)(blockIdx.x * 256 + item) * NACT + a] = (q[a].x + q[a].y) + s_half[item][a];
    }
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    hipDeviceProp_t pr;
    if (hipGetDeviceProperties(&pr, 0) != hipSuccess) { printf("no device\n"); return 1; }
    const int cus = pr.multiProcessorCount, n = cus * 256, reps = 50;
    std::vector<float> hW(NACT * NF), hs(n * 4);
    for (size_t i = 0; i < hW.size(); ++i) hW[i] = 1e-3f * (float)((i * 2654435761u) % 1000) - 0.5f;
    for (size_t i = 0; i < hs.size(); ++i) hs[i] = (float)((i * 40503u) % 1000) * 1e-3f;
    float *W, *q; float4 *st; unsigned long long *cyc;
    (void)hipMalloc(&W, hW.size() * 4); (void)hipMalloc(&st, hs.size() * 4); (void)hipMalloc(&q, (size_t)n * NACT * 4); (void)hipMalloc(&cyc, cus * 8);
    (void)hipMemcpy(W, hW.data(), hW.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(st, hs.data(), hs.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(q_kernel, dim3(cus), dim3(512), 0, 0, W, st, q, cyc, reps);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    std::vector<unsigned long long> hc(cus);
    (void)hipMemcpy(hc.data(), cyc, cus * 8, hipMemcpyDeviceToHost);
    double mean = 0; for (auto c : hc) mean += (double)c; mean /= cus;
    std::vector<float> hq(8);
    (void)hipMemcpy(hq.data(), q, 32, hipMemcpyDeviceToHost);
    printf("device %s, %d CUs\n", pr.name, cus);
    printf("lane-per-item Q(s',.): %.0f s_memtime ticks per workgroup pass (256 items, 8 waves, wave 0), %.1f us per pass by events\n",
           mean / reps, ms * 1e3 / reps);
    printf("check q[0..4] = %g %g %g %g %g\n", hq[0], hq[1], hq[2], hq[3], hq[4]);
    return 0;
}
