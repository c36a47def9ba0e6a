// Does a table BUILD (dependent complex products on the vector pipe + LDS stores) run beside MFMA-bound PRODUCT loops (LDS operand
// reads + v_mfma_f32_16x16x4_f32) of other waves of the same SIMD, or do the two interleave badly? This is the shape of the step
// kernel's U2 stage; round 4's pipelined U2 (profiles/r04_u2_two_table_pipeline.diff) lost 9 % where the arithmetic said it would
// gain 5 %. One 1024-thread workgroup per CU (16 waves, 4 per SIMD), three arrangements of the SAME work (every wave: STAGES builds
// and STAGES product stretches):
//   mode 0  all build, barrier, all multiply, barrier                      (the shipped form)
//   mode 1  per stage: waves 0-3 and 8-11 multiply then build, waves 4-7 and 12-15 build then multiply; one barrier per stage
//   mode 2  as mode 1, builds at s_setprio 3
// build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -o build_beside_mfma build_beside_mfma.hip ; run: ./build_beside_mfma
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef float f4v __attribute__((ext_vector_type(4)));
constexpr int STAGES = 6, GROUPS = 7, ROW = 260;      // product stretch: 7 groups x 12 MFMAs; table rows of 260 floats (128 slots; 260 = 4 mod 64)

template <bool STORE>
__device__ __forceinline__ float build(float *tab, int wave, int lane, float seed) {
    float keep = 0.0f;
    // ~ item_entries: powers of four unit complex numbers by repeated complex products (dependent chains), 12 float2 results stored
    float2 z[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) z[d] = make_float2(0.6f + 0.01f * d + seed, 0.8f - 0.01f * d);
    float2 p[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) p[d] = z[d];
    const int slot = 8 * wave + lane % 8, cp = lane / 8;
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        float2 ab = make_float2(p[0].x * p[1].x - p[0].y * p[1].y, p[0].x * p[1].y + p[0].y * p[1].x);
        float2 cd = make_float2(p[2].x * p[3].x - p[2].y * p[3].y, p[2].x * p[3].y + p[2].y * p[3].x);
        if (STORE) {
            if (cp < 6) {
                *reinterpret_cast<float2 *>(tab + (cp + 6 * c) * ROW + 2 * slot) = ab;
                *reinterpret_cast<float2 *>(tab + (36 + cp + 6 * c) * ROW + 2 * slot) = cd;
            }
        } else keep += ab.x + ab.y + cd.x + cd.y;
#pragma unroll
        for (int d = 0; d < 4; ++d) p[d] = make_float2(p[d].x * z[d].x - p[d].y * z[d].y, p[d].x * z[d].y + p[d].y * z[d].x);
    }
    return keep;
}

template <bool LOAD>
__device__ __forceinline__ void products(const float *tab, int wave, int lane, f4v (&acc)[6]) {
    const int n16 = lane & 15, g = lane >> 4, so = 8 * (wave % 5);
    const float *pa = tab + n16 * ROW + 2 * g + 2 * so, *pb = tab + (36 + n16) * ROW + 2 * g + 2 * so;
    for (int gi = 0; gi < GROUPS; ++gi) {
        float2 a2, b2, c0, c1, c2;
        if (LOAD) {
            a2 = *reinterpret_cast<const float2 *>(pa + 8 * gi); b2 = *reinterpret_cast<const float2 *>(pa + 16 * ROW + 8 * gi);
            c0 = *reinterpret_cast<const float2 *>(pb + 8 * gi); c1 = *reinterpret_cast<const float2 *>(pb + 16 * ROW + 8 * gi);
            c2 = *reinterpret_cast<const float2 *>(pb + 20 * ROW + 8 * gi);
        } else {
            a2 = make_float2(1.0f + lane * 1e-3f, 0.5f); b2 = make_float2(0.25f, 1.0f - lane * 1e-3f);
            c0 = a2; c1 = b2; c2 = make_float2(a2.y, b2.x);
            asm volatile("" : "+v"(a2.x), "+v"(b2.y), "+v"(gi));
        }
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(c0.x, a2.x, acc[0], 0, 0, 0); acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(c1.x, a2.x, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(c2.x, a2.x, acc[2], 0, 0, 0); acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(c0.y, a2.y, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(c1.y, a2.y, acc[1], 0, 0, 0); acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(c2.y, a2.y, acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(c0.x, b2.x, acc[3], 0, 0, 0); acc[4] = __builtin_amdgcn_mfma_f32_16x16x4f32(c1.x, b2.x, acc[4], 0, 0, 0);
        acc[5] = __builtin_amdgcn_mfma_f32_16x16x4f32(c2.x, b2.x, acc[5], 0, 0, 0); acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(c0.y, b2.y, acc[3], 0, 0, 0);
        acc[4] = __builtin_amdgcn_mfma_f32_16x16x4f32(c1.y, b2.y, acc[4], 0, 0, 0); acc[5] = __builtin_amdgcn_mfma_f32_16x16x4f32(c2.y, b2.y, acc[5], 0, 0, 0);
    }
}

__global__ __launch_bounds__(1024) void k(int mode, unsigned long long *out, float *sink) {
    extern __shared__ float lds[];                         // two table buffers of 72 rows x 260 floats (150 KB)
    float *buf0 = lds, *buf1 = lds + 72 * ROW;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    f4v acc[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) acc[i] = (f4v){0.0f, 0.0f, 0.0f, 0.0f};
    build<true>(buf0, wave, lane, 0.0f);
    build<true>(buf1, wave, lane, 0.5f);
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const bool mul_first = ((wave >> 2) & 1) == 0;
    const int arrangement = mode & 3;
    const bool store = !(mode & 16), load = !(mode & 32);
    float keep = 0.0f;
    auto B = [&](float *t, float sd) { keep += store ? build<true>(t, wave, lane, sd) : build<false>(t, wave, lane, sd); };
    auto P = [&](const float *t) { if (load) products<true>(t, wave, lane, acc); else products<false>(t, wave, lane, acc); };
    for (int s = 0; s < STAGES; ++s) {
        float *cur = (s & 1) ? buf1 : buf0, *nxt = (s & 1) ? buf0 : buf1;
        if (arrangement == 0) {
            P(cur);
            __syncthreads();
            B(cur, 0.001f * s);
            __syncthreads();
        } else {
            if (mul_first) P(cur);
            if (arrangement == 2) __builtin_amdgcn_s_setprio(3);
            B(nxt, 0.001f * s);
            if (arrangement == 2) __builtin_amdgcn_s_setprio(0);
            if (!mul_first) P(cur);
            __syncthreads();
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = keep;
#pragma unroll
    for (int i = 0; i < 6; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    sink[blockIdx.x * 1024 + threadIdx.x] = r;
    if (lane == 0) out[blockIdx.x * 16 + wave] = t1 - t0;
}

int main() {
    const int n = 256;
    unsigned long long *out; float *sink;
    (void)hipMalloc(&out, n * 16 * 8); (void)hipMalloc(&sink, n * 1024 * 4);
    const size_t lds = 2 * 72 * ROW * 4;
    (void)hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const char *names[3] = {"all build | barrier | all multiply | barrier", "half of every SIMD multiplies first, half builds first; one barrier per stage",
                            "the same, builds at s_setprio 3"};
    const int modes[] = {0, 1, 2, 16, 17, 32, 33, 48, 49};
    for (int mode : modes) {
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k, dim3(n), dim3(1024), lds, 0, mode, out, sink);
        if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) { printf("launch failed\n"); return 1; }
        std::vector<unsigned long long> c(n * 16); (void)hipMemcpy(c.data(), out, n * 16 * 8, hipMemcpyDeviceToHost);
        double mean = 0; unsigned long long mx = 0;
        for (auto v : c) { mean += v; mx = std::max(mx, v); }
        printf("%-34s %-78s %7.0f cycles per stage (mean over waves; slowest wave %7.0f)\n",
               (mode & 48) == 0 ? "LDS stores + LDS operand reads:" : (mode & 48) == 16 ? "build WITHOUT its LDS stores:" : (mode & 48) == 32 ? "products with REGISTER operands:" : "neither touches LDS:",
               names[mode & 3], mean / c.size() / STAGES, (double)mx / STAGES);
    }
    printf("per stage and wave: one build (6 x (2 + 4) complex products, 12 ds_write_b64) and one product stretch (%d groups x (5 ds_read_b64 + 12 MFMAs)); "
           "MFMA floor per stage: 4 waves x %d MFMAs x 32 cycles = %d cycles per SIMD\n", GROUPS, GROUPS * 12, 4 * GROUPS * 12 * 32);
    return 0;
}
