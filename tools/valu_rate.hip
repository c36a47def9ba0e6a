// valu_rate.hip — micro-benchmark behind DESIGN.md's VALU roofline: sustained wave64 issue rate of
// v_fma_f32 and v_pk_fma_f32 on gfx950 at 1, 2 and 4 waves per SIMD. Build: make -C tools.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int PK>
__global__ void k(float *out, int iters, float a, float b) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    typedef float v2 __attribute__((ext_vector_type(2)));
    v2 p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}, p4 = {x1, x3}, p5 = {x5, x7}, p6 = {x0, x2}, p7 = {x4, x6};
    v2 av = {a, a}, bv = {b, b};
    for (int i = 0; i < iters; ++i) {
        if (PK) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n"
                             "v_pk_fma_f32 %3, %3, %8, %9\n v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n"
                             "v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7)
                             : "v"(av), "v"(bv));
            }
        } else {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n"
                             "v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n"
                             "v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)
                             : "v"(a), "v"(b));
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = PK ? (p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y)
                                                    : (x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7);
}

int main() {
    hipDeviceProp_t pr;
    if (hipGetDeviceProperties(&pr, 0) != hipSuccess) { printf("no device\n"); return 1; }
    const int cus = pr.multiProcessorCount;
    float *out;
    hipMalloc(&out, sizeof(float) * cus * 16 * 64 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    printf("device %s, %d CUs, clock %d MHz\n", pr.name, cus, pr.clockRate / 1000);
    for (int pk = 0; pk < 2; ++pk)
        for (int wps : {1, 2, 4}) {   // waves per SIMD
            dim3 grid(cus), block(64 * 4 * wps);
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (pk) hipLaunchKernelGGL(k<1>, grid, block, 0, 0, out, iters, 1.0001f, 0.5f);
                else hipLaunchKernelGGL(k<0>, grid, block, 0, 0, out, iters, 1.0001f, 0.5f);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
            }
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double instr = (double)iters * 64 * cus * 4 * wps;       // wave-instructions
            const double flop = instr * 64 * 2 * (pk ? 2 : 1);
            printf("%s waves/SIMD=%d: %.3f ms, %.2f wave-instr/ns chip, %.2f cycles/instr/SIMD @2.4GHz, %.1f TFLOP/s\n",
                   pk ? "v_pk_fma_f32" : "v_fma_f32   ", wps, ms, instr / (ms * 1e6),
                   (ms * 1e-3 * 2.4e9) / ((double)iters * 64 * wps), flop / (ms * 1e9));
        }
    return 0;
}
