// Why do U2's heavy waves need 55-80 cycles per v_mfma_f32_16x16x4_f32 (32 = the pipe's rate)? The kernel's product loop in isolation:
// one 1024-thread workgroup, ONE wave per SIMD works (like waves 12..14 of the step kernel while the others wait at the barrier), 24 groups
// per call: four ds_read_b64 operand fetches + six MFMAs on three accumulators per group.   hipcc --offload-arch=gfx950 -O3 -o u2_loop_probe u2_loop_probe.hip
#include <hip/hip_runtime.h>
#pragma clang diagnostic ignored "-Wunused-value"
#include <cstdio>
typedef float f4v __attribute__((ext_vector_type(4)));
constexpr int USX = 292, NG = 24;
template <int VARIANT, bool WAITERS, bool RANDOM_DATA>
__global__ __launch_bounds__(1024, 4) void probe(float *out, unsigned long long *cyc, int ngrp) {
    extern __shared__ float tab[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, n16 = lane & 15, g = lane >> 4;
    for (int i = tid; i < 3 * 36 * USX; i += 1024) {
        unsigned h = (unsigned)i * 2654435761u + (unsigned)ngrp * 40503u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        tab[i] = RANDOM_DATA ? __uint_as_float((h & 0x807fffffu) | 0x3f000000u) - ((h >> 8) & 1 ? 0.75f : 0.0f)      // full mantissas, both signs, |x| < 1
                             : 1e-3f * (float)(i % 97);
    }
    __syncthreads();
    const bool idle = VARIANT >= 5 ? !((0x703f >> wave) & 1) : (wave < 12 || wave > 14);
    if (idle && !WAITERS) return;
    if (idle) { __builtin_amdgcn_s_barrier(); return; }    // WAITERS: the other waves wait at the barrier the workers reach after their loop, as in the kernel      // (the other waves are gone: the barrier wait of the kernel; variants 5, 6: nine waves, 3 + 2 + 2 + 2 per SIMD)
    const float *paA = tab + n16 * USX + 2 * g, *pb0 = tab + 2 * 36 * USX + n16 * USX + 2 * g, *pb1 = pb0 + 16 * USX, *pb2 = pb0 + 20 * USX;
    f4v a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (VARIANT == 0) {                                     // the product's loop
        for (int gi = 0; gi < ngrp; ++gi) {
            const float2 p = *reinterpret_cast<const float2 *>(paA + 8 * gi);
            const float2 c0 = *reinterpret_cast<const float2 *>(pb0 + 8 * gi), c1 = *reinterpret_cast<const float2 *>(pb1 + 8 * gi), c2 = *reinterpret_cast<const float2 *>(pb2 + 8 * gi);
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(c0.x, p.x, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(c1.x, p.x, a1, 0, 0, 0); a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(c2.x, p.x, a2, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(c0.y, p.y, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(c1.y, p.y, a1, 0, 0, 0); a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(c2.y, p.y, a2, 0, 0, 0);
        }
    } else if (VARIANT == 1) {                              // operands from registers: the MFMA stream alone
        float2 p = *reinterpret_cast<const float2 *>(paA), c0 = *reinterpret_cast<const float2 *>(pb0), c1 = *reinterpret_cast<const float2 *>(pb1), c2 = *reinterpret_cast<const float2 *>(pb2);
        for (int gi = 0; gi < ngrp; ++gi) {
            asm volatile("" : "+v"(p.x), "+v"(c0.x));
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(c0.x, p.x, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(c1.x, p.x, a1, 0, 0, 0); a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(c2.x, p.x, a2, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(c0.y, p.y, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(c1.y, p.y, a1, 0, 0, 0); a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(c2.y, p.y, a2, 0, 0, 0);
        }
    } else if (VARIANT == 3) {                              // the next group's operands fetched before this group's products are issued
        float2 p = *reinterpret_cast<const float2 *>(paA), c0 = *reinterpret_cast<const float2 *>(pb0), c1 = *reinterpret_cast<const float2 *>(pb1), c2 = *reinterpret_cast<const float2 *>(pb2);
        for (int gi = 0; gi < ngrp; ++gi) {
            const int gn = min(gi + 1, ngrp - 1);
            const float2 np = *reinterpret_cast<const float2 *>(paA + 8 * gn);
            const float2 n0 = *reinterpret_cast<const float2 *>(pb0 + 8 * gn), n1 = *reinterpret_cast<const float2 *>(pb1 + 8 * gn), n2 = *reinterpret_cast<const float2 *>(pb2 + 8 * gn);
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(c0.x, p.x, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(c1.x, p.x, a1, 0, 0, 0); a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(c2.x, p.x, a2, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(c0.y, p.y, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(c1.y, p.y, a1, 0, 0, 0); a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(c2.y, p.y, a2, 0, 0, 0);
            p = np; c0 = n0; c1 = n1; c2 = n2;
        }
    } else if (VARIANT == 4) {                              // ping-pong operand sets, the fetches in inline asm (the compiler undoes a source-level rotation)
        float2 p, c0, c1, c2, q, d0, d1, d2;
        unsigned aA = (unsigned)(size_t)paA, a0_ = (unsigned)(size_t)pb0, a1_ = (unsigned)(size_t)pb1, a2_ = (unsigned)(size_t)pb2;
#define FETCH(P, C0, C1, C2) asm volatile("ds_read_b64 %0, %7\n\tds_read_b64 %1, %8\n\tds_read_b64 %2, %9\n\tds_read_b64 %3, %10" \
                 : "=&v"(P), "=&v"(C0), "=&v"(C1), "=&v"(C2), "+v"(a0), "+v"(a1), "+v"(a2) : "v"(aA), "v"(a0_), "v"(a1_), "v"(a2_)); aA += 32; a0_ += 32; a1_ += 32; a2_ += 32
#define LANDED(P, C0, C1, C2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(P), "+v"(C0), "+v"(C1), "+v"(C2), "+v"(a0), "+v"(a1), "+v"(a2))
#define SIX(P, C0, C1, C2) a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(C0.x, P.x, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(C1.x, P.x, a1, 0, 0, 0); a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(C2.x, P.x, a2, 0, 0, 0); \
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(C0.y, P.y, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(C1.y, P.y, a1, 0, 0, 0); a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(C2.y, P.y, a2, 0, 0, 0)
        FETCH(p, c0, c1, c2); LANDED(p, c0, c1, c2);
        for (int gi = 0; gi < ngrp; gi += 2) {              // (the last fetch of an even count reads one group past the end: inside the table)
            FETCH(q, d0, d1, d2);
            SIX(p, c0, c1, c2);
            LANDED(q, d0, d1, d2);
            FETCH(p, c0, c1, c2);
            SIX(q, d0, d1, d2);
            LANDED(p, c0, c1, c2);
        }
    } else if (VARIANT == 7) {                              // calibration: 2048 dependent v_add_f32 (the same chain runs on an idle wave inside the step kernel's U2)
        float v = (float)lane;
        for (int i = 0; i < 256; ++i) asm volatile("v_add_f32 %0, %0, %0\n\tv_add_f32 %0, %0, %0\n\tv_add_f32 %0, %0, %0\n\tv_add_f32 %0, %0, %0\n\tv_add_f32 %0, %0, %0\n\tv_add_f32 %0, %0, %0\n\tv_add_f32 %0, %0, %0\n\tv_add_f32 %0, %0, %0" : "+v"(v));
        a0[0] = v;
    } else if (VARIANT == 5 || VARIANT == 6) {              // round 5's re-mapped loop: one tile, one P + one C (+ one B) operand, two (+ two) dependent MFMAs per group
        const float *paB = tab + 36 * USX + n16 * USX + 2 * g;
        for (int gi = 0; gi < ngrp; ++gi) {
            const float2 p = *reinterpret_cast<const float2 *>(paA + 8 * gi), c = *reinterpret_cast<const float2 *>(pb0 + 8 * gi);
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(c.x, p.x, a0, 0, 0, 0); a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(c.y, p.y, a0, 0, 0, 0);
            if (VARIANT == 6) {
                const float2 b = *reinterpret_cast<const float2 *>(paB + 8 * gi);
                a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(c.x, b.x, a1, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(c.y, b.y, a1, 0, 0, 0);
            }
        }
    } else {                                                // two groups per iteration, eight fetches up front
        for (int gi = 0; gi + 1 < ngrp; gi += 2) {
            const float2 p = *reinterpret_cast<const float2 *>(paA + 8 * gi), q = *reinterpret_cast<const float2 *>(paA + 8 * gi + 8);
            const float2 c0 = *reinterpret_cast<const float2 *>(pb0 + 8 * gi), c1 = *reinterpret_cast<const float2 *>(pb1 + 8 * gi), c2 = *reinterpret_cast<const float2 *>(pb2 + 8 * gi);
            const float2 d0 = *reinterpret_cast<const float2 *>(pb0 + 8 * gi + 8), d1 = *reinterpret_cast<const float2 *>(pb1 + 8 * gi + 8), d2 = *reinterpret_cast<const float2 *>(pb2 + 8 * gi + 8);
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(c0.x, p.x, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(c1.x, p.x, a1, 0, 0, 0); a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(c2.x, p.x, a2, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(c0.y, p.y, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(c1.y, p.y, a1, 0, 0, 0); a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(c2.y, p.y, a2, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(d0.x, q.x, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(d1.x, q.x, a1, 0, 0, 0); a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(d2.x, q.x, a2, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(d0.y, q.y, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(d1.y, q.y, a1, 0, 0, 0); a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(d2.y, q.y, a2, 0, 0, 0);
        }
    }
    asm volatile("s_nop 0" :: "v"(a0), "v"(a1), "v"(a2));
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (WAITERS) __builtin_amdgcn_s_barrier();
    out[(blockIdx.x * 1024 + tid) * 3] = a0[0] + a1[1] + a2[2];
    if (lane == 0) cyc[blockIdx.x * 16 + wave] = t1 - t0;
}
template <int V, bool WT = false, bool RD = false> static void run(const char *name, int nblk) {
    float *out; unsigned long long *cyc;
    hipMalloc(&out, (size_t)nblk * 1024 * 3 * 4); hipMalloc(&cyc, (size_t)nblk * 16 * 8);
    hipFuncSetAttribute(reinterpret_cast<const void *>(&probe<V, WT, RD>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 36 * USX * 4);
    for (int r = 0; r < 3; ++r) probe<V, WT, RD><<<nblk, 1024, 3 * 36 * USX * 4>>>(out, cyc, NG);
    hipDeviceSynchronize();
    unsigned long long *h = new unsigned long long[nblk * 16];
    hipMemcpy(h, cyc, (size_t)nblk * 16 * 8, hipMemcpyDeviceToHost);
    double s = 0; int nw = 0; for (int b = 0; b < nblk; ++b) for (int w = 0; w < 16; ++w) if (V >= 5 ? ((0x703f >> w) & 1) : (w >= 12 && w < 15)) { s += (double)h[b * 16 + w]; ++nw; }
    s /= nw;
    const int per = V == 5 ? 2 : V == 6 ? 4 : 6;
    printf("%-58s %7.0f cycles per call of %d groups = %5.1f per group, %5.1f per MFMA (%d blocks)\n", name, s, NG, s / NG, s / (per * (double)NG), nblk);
    hipFree(out); hipFree(cyc); delete[] h;
}
int main() {
    for (int nblk : {1, 256}) {
        run<0>("the kernel's loop: 4 ds_read_b64 + 6 MFMA per group", nblk);
        run<1>("operands in registers", nblk);
        run<2>("two groups per iteration: 8 ds_read_b64 + 12 MFMA", nblk);
        run<3>("next group's operands fetched ahead", nblk);
        run<4>("ping-pong operand sets, fetches in inline asm", nblk);
        run<5>("re-mapped: 9 waves, 2 reads + 2 dependent MFMAs", nblk);
        run<6>("re-mapped: 9 waves, 3 reads + 2 + 2 dependent MFMAs", nblk);
        run<7>("calibration: 2048 dependent v_add_f32, one wave per SIMD", nblk);
        run<0, false, true>("the kernel's loop, RANDOM operands (full mantissas, both signs)", nblk);
        run<4, false, true>("ping-pong, RANDOM operands", nblk);
        run<1, false, true>("operands in registers, RANDOM operands", nblk);
        run<0, true>("the kernel's loop, the other 13 waves WAITING at a barrier", nblk);
        run<6, true>("re-mapped, 3 reads + 2 + 2, the other 7 waves WAITING", nblk);
    }
    return 0;
}
