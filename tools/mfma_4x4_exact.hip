// mfma_4x4_exact.hip — is v_mfma_f32_4x4x1_16b_f32 (16 independent 4x4 blocks, K = 1: D[b][i][j] += A[b][i] B[b][j]) the
// same fmaf as v_mfma_f32_16x16x4_f32's chain steps, bit for bit (subnormals, cancellation), and what does it cost on
// the matrix pipe? Layout probed: A lane l -> block l>>2, row l&3; B lane l -> block l>>2, col l&3; D reg v of lane l ->
// block l>>2, row v, col l&3. Used to decide U2's 4-wide border strips of the 36 x 36 gradient.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
typedef float f4v __attribute__((ext_vector_type(4)));
constexpr int K = 64;

__global__ void chain(const float *A /*[K][64]*/, const float *B /*[K][64]*/, float *D /*[64][4]*/, long long *cyc) {
    const int l = threadIdx.x;
    float a[K], b[K];
#pragma unroll
    for (int k = 0; k < K; ++k) { a[k] = A[k * 64 + l]; b[k] = B[k * 64 + l]; }
    // timed sections: the accumulators pass through an asm after the start stamp and are read by one before the end
    // stamp, so the chain can neither start early nor still be in flight when the clock is read
#define T_START(t) asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory")
#define PIN(x) asm volatile("" : "+v"(x))
#define T_END(t, x) do { float x0_ = (x)[0]; asm volatile("v_mov_b32 %1, %1\n s_nop 4\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t), "+v"(x0_) :: "memory"); (x)[0] = x0_; } while (0)
    long long t0, t1;
    f4v c = {0.0f, 0.0f, 0.0f, 0.0f};
    __builtin_amdgcn_s_waitcnt(0);
    T_START(t0); PIN(c);
#pragma unroll
    for (int k = 0; k < K; ++k) c = __builtin_amdgcn_mfma_f32_4x4x1f32(a[k], b[k], c, 0, 0, 0);
    T_END(t1, c);
    const long long d1 = t1 - t0;
    f4v e[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    T_START(t0); PIN(e[0]); PIN(e[1]);
#pragma unroll
    for (int k = 0; k < K; ++k) e[k & 1] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[k], b[k], e[k & 1], 0, 0, 0);
    T_END(t1, e[0]); T_END(t1, e[1]);
    const long long d2 = t1 - t0;
    e[0] = e[1] = (f4v){0, 0, 0, 0};
    T_START(t0); PIN(e[0]); PIN(e[1]); PIN(e[2]); PIN(e[3]);
#pragma unroll
    for (int k = 0; k < K; ++k) e[k & 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[k], b[k], e[k & 3], 0, 0, 0);
    T_END(t1, e[0]); T_END(t1, e[1]); T_END(t1, e[2]); T_END(t1, e[3]);
    const long long d3 = t1 - t0;
    f4v f = {0.0f, 0.0f, 0.0f, 0.0f}, f2 = f;
    T_START(t0); PIN(f);
#pragma unroll
    for (int k = 0; k < K; ++k) f = __builtin_amdgcn_mfma_f32_16x16x4f32(a[k], b[k], f, 0, 0, 0);
    T_END(t1, f);
    const long long d4 = t1 - t0;
    f = (f4v){0, 0, 0, 0};
    T_START(t0); PIN(f); PIN(f2);
#pragma unroll
    for (int k = 0; k < K; k += 2) { f = __builtin_amdgcn_mfma_f32_16x16x4f32(a[k], b[k], f, 0, 0, 0); f2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[k + 1], b[k + 1], f2, 0, 0, 0); }
    T_END(t1, f); T_END(t1, f2);
    const long long d5 = t1 - t0;
    if (l == 0) { cyc[0] = d1; cyc[1] = d2; cyc[2] = d3; cyc[3] = d4; cyc[4] = d5; }
    float sink = e[0][0] + e[1][0] + e[2][0] + e[3][0] + f[0] + f2[0];
    if (sink == 12345.678f) D[0] = 0.0f;
#pragma unroll
    for (int v = 0; v < 4; ++v) D[l * 4 + v] = c[v];
}

int main() {
    std::vector<float> A(K * 64), B(K * 64), D(256);
    unsigned s = 2463534242u;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return s; };
    int bad_total = 0;
    for (int trial = 0; trial < 4; ++trial) {
        for (int i = 0; i < K * 64; ++i) {
            float x = (float)(rnd() & 0xffffff) / 8388608.0f - 1.0f, y = (float)(rnd() & 0xffffff) / 8388608.0f - 1.0f;
            if (trial == 1) { x *= 0x1p-70f; y *= 0x1p-70f; }                      // products and sums in the subnormal range
            if (trial == 2 && (i & 3) == 3) { x = -A[i - 1]; y = B[i - 1]; }       // exact cancellation -> signed zeros
            if (trial == 3) { x *= (i & 1) ? 0x1p60f : 0x1p-60f; y *= (i & 2) ? 0x1p60f : 0x1p-64f; }
            A[i] = x; B[i] = y;
        }
        float *dA, *dB, *dD; long long *dC, cyc[5];
        (void)hipMalloc(&dA, K * 256); (void)hipMalloc(&dB, K * 256); (void)hipMalloc(&dD, 1024); (void)hipMalloc(&dC, 40);
        (void)hipMemcpy(dA, A.data(), K * 256, hipMemcpyHostToDevice);
        (void)hipMemcpy(dB, B.data(), K * 256, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(chain, dim3(1), dim3(64), 0, 0, dA, dB, dD, dC);
        if (hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost) != hipSuccess) { printf("no device\n"); return 1; }
        (void)hipMemcpy(cyc, dC, 40, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int l = 0; l < 64; ++l)
            for (int v = 0; v < 4; ++v) {
                const int blk = l >> 2, j = l & 3, i = v;
                float acc = 0.0f;
                for (int k = 0; k < K; ++k) acc = fmaf(A[k * 64 + 4 * blk + i], B[k * 64 + 4 * blk + j], acc);
                if (memcmp(&acc, &D[l * 4 + v], 4)) { if (bad < 4) printf("  lane %d reg %d: mfma %a  fmaf chain %a\n", l, v, D[l * 4 + v], acc); ++bad; }
            }
        printf("trial %d: %d of 256 outputs differ from the k-ordered fmaf chain (K = %d)\n", trial, bad, K);
        if (trial == 0)
            printf("cycles per instruction (one wave alone, %d back-to-back): 4x4x1 on one accumulator %.1f, on two %.1f, on four %.1f; "
                   "16x16x4 on one accumulator %.1f, on two %.1f\n", K, (double)cyc[0] / K, (double)cyc[1] / K, (double)cyc[2] / K,
                   (double)cyc[3] / K, (double)cyc[4] / K);
        bad_total += bad;
    }
    printf(bad_total ? "NOT exact\n" : "exact: v_mfma_f32_4x4x1_16b_f32 == fmaf per step, all trials\n");
    return bad_total != 0;
}
