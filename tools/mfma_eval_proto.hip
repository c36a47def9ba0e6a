// mfma_eval_proto.hip — prototype + exactness check of the round-2 evaluation loop (SPEC §3.1, round 2):
//   T[row][col] = chain_{c34} fma(W[row][c34], CD[c34][col], T)   on v_mfma_f32_16x16x4_f32
//   Q[a][item]  = sum over lane groups / (re, im) of chains  fma(T[a,c12][col], ABsel[c12][col], q)
// (1) numerics: one 16x16x4 MFMA with C != 0 and subnormal inputs against the k-ordered fmaf chain;
// (2) the whole per-wave loop (tables from Z^1, 108 MFMAs per 8 items, stage 2, butterflies) against a scalar CPU
//     model of the same definition, bit for bit;
// (3) timing: cycles per 8-item column block with 1 and 2 waves per SIMD, every CU busy.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_one(const float *A /*[16][4]*/, const float *B /*[4][16]*/, const float *C /*[16][16]*/, float *D) {
    const int l = threadIdx.x;
    const float a = A[(l & 15) * 4 + (l >> 4)];
    const float b = B[(l >> 4) * 16 + (l & 15)];
    f4 c;
#pragma unroll
    for (int v = 0; v < 4; ++v) c[v] = C[(4 * (l >> 4) + v) * 16 + (l & 15)];
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
#pragma unroll
    for (int v = 0; v < 4; ++v) D[(4 * (l >> 4) + v) * 16 + (l & 15)] = c[v];
}

// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(fmaf(-a.y, b.y, a.x * b.x), fmaf(a.x, b.y, a.y * b.x));
}
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ float swz_xor4(float v) { return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x101F)); }

constexpr int WAVES = 4;
constexpr int TAB_FLOATS = 36 * 16 + 16 * 36;     // CDk[36][16] + ABq[16][36]

// power c (0..5, lane-dependent) of z: sequential chain Z^k = cmul(Z^(k-1), Z^1)
__device__ __forceinline__ float2 zpow_sel(float2 z, int c) {
    float2 cur = z, out = make_float2(1.0f, 0.0f);
#pragma unroll
    for (int j = 1; j <= 5; ++j) {
        if (c == j) out = cur;
        if (j < 5) cur = cmul(cur, z);
    }
    return out;
}

// tables of one 8-item column block: z1[item][4] (float2) -> CDk[c34][16], ABq[col][36]
__device__ __forceinline__ void build_tables(const float2 *z1 /*[8][4]*/, float *cdk, float *abq, int lane) {
    const int i = lane & 7, cp = lane >> 3;
    if (cp < 6) {
        const float2 z0 = z1[i * 4 + 0], zb = z1[i * 4 + 1], z2 = z1[i * 4 + 2], z3 = z1[i * 4 + 3];
        const float2 pb = zpow_sel(zb, cp), pd = zpow_sel(z3, cp);
        const int cre = 8 * (i >> 2) + (i & 3), cim = cre + 4;
        float2 pa = make_float2(1.0f, 0.0f), pc = make_float2(1.0f, 0.0f);
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            if (c == 1) { pa = z0; pc = z2; }
            if (c > 1) { pa = cmul(pa, z0); pc = cmul(pc, z2); }
            const float2 ab = cmul(pa, pb), cd = cmul(pc, pd);
            abq[cre * 36 + 6 * c + cp] = ab.x; abq[cim * 36 + 6 * c + cp] = -ab.y;
            cdk[(6 * c + cp) * 16 + cre] = cd.x; cdk[(6 * c + cp) * 16 + cim] = cd.y;
        }
    }
}

template <bool TIMING>
__global__ __launch_bounds__(WAVES * 64, 2) void k_eval(const float *W /*[5][36][36]*/, const float2 *Z1 /*[items][4]*/,
                                                        int n_items, float *Q /*[items][5]*/, int reps,
                                                        unsigned long long *cyc) {
    __shared__ __attribute__((aligned(16))) float s_tab[WAVES][TAB_FLOATS];
    __shared__ float2 s_z1[WAVES][8 * 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, n = lane & 15;
    float *cdk = s_tab[wave], *abq = s_tab[wave] + 36 * 16;
    // A operands: W rows 16t + (lane & 15), k = 4 kb + g
    float Wr[12][9];
#pragma unroll
    for (int t = 0; t < 12; ++t) {
        const int row = 16 * t + n;
#pragma unroll
        for (int kb = 0; kb < 9; ++kb) Wr[t][kb] = row < 180 ? W[row * 36 + 4 * kb + g] : 0.0f;
    }
    const int nblk = (n_items + 7) / 8;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int rep = 0; rep < reps; ++rep) {
        for (int cb = blockIdx.x * WAVES + wave; cb < nblk; cb += gridDim.x * WAVES) {
            if (lane < 32) {
                const int it = cb * 8 + (lane >> 2);
                s_z1[wave][lane] = it < n_items ? Z1[it * 4 + (lane & 3)] : make_float2(1.0f, 0.0f);
            }
            wave_lds_sync();
            build_tables(s_z1[wave], cdk, abq, lane);
            wave_lds_sync();
            float B[9];
#pragma unroll
            for (int kb = 0; kb < 9; ++kb) B[kb] = cdk[64 * kb + lane];
            f4 acc[12];
#pragma unroll
            for (int t = 0; t < 12; ++t) acc[t] = (f4){0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int kb = 0; kb < 9; ++kb) {
#pragma unroll
                for (int t = 0; t < 12; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(Wr[t][kb], B[kb], acc[t], 0, 0, 0);
            }
            // stage 2: rows rho = 16 t + 4 g + v -> action rho / 36, c12 = rho % 36
            float q[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int t = 0; t < 12; ++t) {
                constexpr int dummy = 0; (void)dummy;
                const int Ct = (16 * t) % 36, At = (16 * t) / 36;
                const bool wrap = Ct + 4 * g >= 36;
                const int c0 = Ct + 4 * g - (wrap ? 36 : 0);
                const f4 ab = *reinterpret_cast<const f4 *>(abq + n * 36 + c0);
                const bool mixed = Ct + 12 >= 36;
                if (!mixed) {
#pragma unroll
                    for (int v = 0; v < 4; ++v) q[At] = fmaf(acc[t][v], ab[v], q[At]);
                } else {
                    float x = wrap ? q[At + 1] : q[At];
#pragma unroll
                    for (int v = 0; v < 4; ++v) x = fmaf(acc[t][v], ab[v], x);
                    q[At] = wrap ? q[At] : x;
                    q[At + 1] = wrap ? x : q[At + 1];
                }
            }
#pragma unroll
            for (int a = 0; a < 5; ++a) q[a] = q[a] + swz_xor4(q[a]);                  // re + im
#pragma unroll
            for (int a = 0; a < 5; ++a) {
                const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(q[a]), __float_as_uint(q[a]), false, false);
                q[a] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
            }
#pragma unroll
            for (int a = 0; a < 5; ++a) {
                const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(q[a]), __float_as_uint(q[a]), false, false);
                q[a] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
            }
            if (g == 0 && !(n & 4)) {
                const int it = cb * 8 + (n >> 3) * 4 + (n & 3);
                if (it < n_items) {
#pragma unroll
                    for (int a = 0; a < 5; ++a) Q[it * 5 + a] = q[a];
                }
            }
            wave_lds_sync();
        }
    }
    if (TIMING && lane == 0) cyc[blockIdx.x * WAVES + wave] = __builtin_amdgcn_s_memtime() - t0;
}


// ---------------------------------------------------------------------------------------------------------
// variant: W staged in LDS in A-operand order ([tile][k-block group of 4][lane][4] + [tile][lane] for kb = 8), A operands
// streamed per MFMA; <= 128 VGPRs so that 4 waves share a SIMD (8-wave workgroups, two per CU)
constexpr int WAVES2 = 8;
template <bool TIMING>
__global__ __launch_bounds__(WAVES2 * 64, 4) void k_eval_lds(const float *W, const float2 *Z1, int n_items, float *Q, int reps,
                                                            unsigned long long *cyc) {
    __shared__ __attribute__((aligned(16))) float s_w[12 * 9 * 64];
    __shared__ __attribute__((aligned(16))) float s_tab[WAVES2][TAB_FLOATS];
    __shared__ float2 s_z1[WAVES2][8 * 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, n = lane & 15;
    float *cdk = s_tab[wave], *abq = s_tab[wave] + 36 * 16;
    // stage: entry (t, kb, lane) at ((t*3 + kb/4)*64 + lane)*4 + kb%4 for kb < 8, and 12*2*64*4 + (t*64 + lane) for kb = 8
    for (int i = tid; i < 12 * 9 * 64; i += WAVES2 * 64) {
        const int l = i & 63, kb = (i >> 6) % 9, t = (i >> 6) / 9;
        const int row = 16 * t + (l & 15), gg = l >> 4;
        const float v = row < 180 ? W[row * 36 + 9 * gg + kb] : 0.0f;
        if (kb < 8) s_w[((t * 2 + (kb >> 2)) * 64 + l) * 4 + (kb & 3)] = v;
        else s_w[12 * 2 * 64 * 4 + t * 64 + l] = v;
    }
    __syncthreads();
    const int nblk = (n_items + 7) / 8;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int rep = 0; rep < reps; ++rep) {
        for (int cb = blockIdx.x * WAVES2 + wave; cb < nblk; cb += gridDim.x * WAVES2) {
            if (lane < 32) {
                const int it = cb * 8 + (lane >> 2);
                s_z1[wave][lane] = it < n_items ? Z1[it * 4 + (lane & 3)] : make_float2(1.0f, 0.0f);
            }
            wave_lds_sync();
            build_tables(s_z1[wave], cdk, abq, lane);
            wave_lds_sync();
            float B[9];
#pragma unroll
            for (int kb = 0; kb < 9; ++kb) B[kb] = cdk[64 * kb + lane];
            f4 acc[12];
#pragma unroll
            for (int t = 0; t < 12; ++t) acc[t] = (f4){0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int t = 0; t < 12; ++t) {
                const f4 a0 = *reinterpret_cast<const f4 *>(s_w + ((t * 2 + 0) * 64 + lane) * 4);
                const f4 a1 = *reinterpret_cast<const f4 *>(s_w + ((t * 2 + 1) * 64 + lane) * 4);
                const float a8 = s_w[12 * 2 * 64 * 4 + t * 64 + lane];
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[kb], B[kb], acc[t], 0, 0, 0);
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[kb], B[4 + kb], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a8, B[8], acc[t], 0, 0, 0);
            }
            float q[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int t = 0; t < 12; ++t) {
                const int Ct = (16 * t) % 36, At = (16 * t) / 36;
                const bool wrap = Ct + 4 * g >= 36;
                const int c0 = Ct + 4 * g - (wrap ? 36 : 0);
                const f4 ab = *reinterpret_cast<const f4 *>(abq + n * 36 + c0);
                const bool mixed = Ct + 12 >= 36;
                if (!mixed) {
#pragma unroll
                    for (int v = 0; v < 4; ++v) q[At] = fmaf(acc[t][v], ab[v], q[At]);
                } else {
                    float x = wrap ? q[At + 1] : q[At];
#pragma unroll
                    for (int v = 0; v < 4; ++v) x = fmaf(acc[t][v], ab[v], x);
                    q[At] = wrap ? q[At] : x;
                    q[At + 1] = wrap ? x : q[At + 1];
                }
            }
#pragma unroll
            for (int a = 0; a < 5; ++a) q[a] = q[a] + swz_xor4(q[a]);
#pragma unroll
            for (int a = 0; a < 5; ++a) {
                const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(q[a]), __float_as_uint(q[a]), false, false);
                q[a] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
            }
#pragma unroll
            for (int a = 0; a < 5; ++a) {
                const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(q[a]), __float_as_uint(q[a]), false, false);
                q[a] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
            }
            if (g == 0 && !(n & 4)) {
                const int it = cb * 8 + (n >> 3) * 4 + (n & 3);
                if (it < n_items) {
#pragma unroll
                    for (int a = 0; a < 5; ++a) Q[it * 5 + a] = q[a];
                }
            }
            wave_lds_sync();
        }
    }
    if (TIMING && lane == 0) cyc[blockIdx.x * WAVES2 + wave] = __builtin_amdgcn_s_memtime() - t0;
}

// ---------------------------------------------------------------------------------------------------------
struct cf { float re, im; };
static cf h_cmul(cf a, cf b) { return {fmaf(-a.im, b.im, a.re * b.re), fmaf(a.re, b.im, a.im * b.re)}; }

static void cpu_q(const float *W, const cf z1[4], float Q[5]) {
    cf Z[4][6];
    for (int d = 0; d < 4; ++d) {
        Z[d][0] = {1.0f, 0.0f}; Z[d][1] = z1[d];
        for (int k = 2; k < 6; ++k) Z[d][k] = h_cmul(Z[d][k - 1], z1[d]);
    }
    cf AB[36], CD[36];
    for (int a = 0; a < 6; ++a)
        for (int b = 0; b < 6; ++b) { AB[a * 6 + b] = h_cmul(Z[0][a], Z[1][b]); CD[a * 6 + b] = h_cmul(Z[2][a], Z[3][b]); }
    for (int a = 0; a < 5; ++a) {
        float qg[4][2] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
        for (int c12 = 0; c12 < 36; ++c12) {
            float tre = 0.0f, tim = 0.0f;
            for (int c34 = 0; c34 < 36; ++c34) {
                const float w = W[(a * 36 + c12) * 36 + c34];
                tre = fmaf(w, CD[c34].re, tre); tim = fmaf(w, CD[c34].im, tim);
            }
            const int g = ((36 * a + c12) % 16) / 4;
            qg[g][0] = fmaf(tre, AB[c12].re, qg[g][0]);
            qg[g][1] = fmaf(tim, -AB[c12].im, qg[g][1]);
        }
        float u[4];
        for (int g = 0; g < 4; ++g) u[g] = qg[g][0] + qg[g][1];
        Q[a] = (u[0] + u[1]) + (u[2] + u[3]);
    }
}

int main() {
    // ---- (1) one MFMA against the chain, C != 0, subnormals in A, B, C
    {
        std::vector<float> A(64), B(64), Cc(256), D(256);
        unsigned s = 777u;
        auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)((s >> 8) & 0xffffff) / 16777216.0f * 2.0f - 1.0f; };
        for (auto &x : A) x = rnd();
        for (auto &x : B) x = rnd();
        for (auto &x : Cc) x = rnd() * 3.0f;
        A[5] = 1e-41f; B[7] = -3e-40f; Cc[9] = 2e-42f; A[20] = 1e-30f; B[33] = 1e-12f; Cc[100] = 0.0f; Cc[101] = -0.0f;
        float *dA, *dB, *dC, *dD;
        CHECK(hipMalloc(&dA, 256)); CHECK(hipMalloc(&dB, 256)); CHECK(hipMalloc(&dC, 1024)); CHECK(hipMalloc(&dD, 1024));
        CHECK(hipMemcpy(dA, A.data(), 256, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dB, B.data(), 256, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(dC, Cc.data(), 1024, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_one, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
        CHECK(hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost));
        int ok = 0;
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 16; ++j) {
                float c = Cc[i * 16 + j];
                for (int k = 0; k < 4; ++k) c = fmaf(A[i * 4 + k], B[k * 16 + j], c);
                ok += !memcmp(&c, &D[i * 16 + j], 4);
            }
        printf("(1) v_mfma_f32_16x16x4_f32 vs k-ordered fmaf chain from C (subnormals included): %d / 256 bit-identical\n", ok);
    }
    // ---- (2) the loop against the scalar model
    const int n_items = 8 * 4 * 2 * 256 * 4 - 3;            // ragged tail
    std::vector<float> W(5 * 36 * 36);
    std::vector<cf> Z(n_items * 4);
    {
        unsigned s = 4242u;
        auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)((s >> 8) & 0xffffff) / 16777216.0f; };
        for (auto &x : W) x = (rnd() * 2.0f - 1.0f) * 0.05f;
        for (auto &z : Z) { const float th = rnd() * 6.2831853f; z = {cosf(th), sinf(th)}; }
    }
    float *dW, *dQ; float2 *dZ; unsigned long long *dcyc;
    CHECK(hipMalloc(&dW, W.size() * 4)); CHECK(hipMalloc(&dQ, (size_t)n_items * 5 * 4)); CHECK(hipMalloc(&dZ, Z.size() * 8));
    CHECK(hipMalloc(&dcyc, 4096 * 8));
    CHECK(hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dZ, Z.data(), Z.size() * 8, hipMemcpyHostToDevice));
    CHECK(hipMemset(dQ, 0xff, (size_t)n_items * 5 * 4));
    hipLaunchKernelGGL(k_eval<false>, dim3(512), dim3(WAVES * 64), 0, 0, dW, dZ, n_items, dQ, 1, dcyc);
    CHECK(hipDeviceSynchronize());
    std::vector<float> Qg((size_t)n_items * 5);
    CHECK(hipMemcpy(Qg.data(), dQ, Qg.size() * 4, hipMemcpyDeviceToHost));
    {
        int bad = 0, checked = 0;
        for (int it = 0; it < n_items; it += 37) {
            float q[5];
            cpu_q(W.data(), &Z[it * 4], q);
            for (int a = 0; a < 5; ++a) { bad += memcmp(&q[a], &Qg[(size_t)it * 5 + a], 4) != 0; ++checked; }
        }
        float q[5];
        cpu_q(W.data(), &Z[(n_items - 1) * 4], q);
        for (int a = 0; a < 5; ++a) { bad += memcmp(&q[a], &Qg[(size_t)(n_items - 1) * 5 + a], 4) != 0; ++checked; }
        printf("(2) evaluation loop vs scalar model: %d of %d Q values differ (sample Q = %g gpu %g)\n", bad, checked, q[0], Qg[(size_t)(n_items - 1) * 5]);
    }
    // ---- (2b) the LDS-streamed variant against the same model
    CHECK(hipMemset(dQ, 0xff, (size_t)n_items * 5 * 4));
    hipLaunchKernelGGL(k_eval_lds<false>, dim3(512), dim3(WAVES2 * 64), 0, 0, dW, dZ, n_items, dQ, 1, dcyc);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(Qg.data(), dQ, Qg.size() * 4, hipMemcpyDeviceToHost));
    {
        int bad = 0, checked = 0;
        for (int it = 0; it < n_items; it += 37) {
            float q[5];
            cpu_q(W.data(), &Z[it * 4], q);
            for (int a = 0; a < 5; ++a) { bad += memcmp(&q[a], &Qg[(size_t)it * 5 + a], 4) != 0; ++checked; }
        }
        printf("(2b) A operands streamed from LDS, 4 waves/SIMD: %d of %d Q values differ\n", bad, checked);
    }
    {
        const int grid = 512, reps = 8;
        const int items = grid * WAVES2 * 8 * 2;             // 2 column blocks per wave per rep
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k_eval_lds<true>, dim3(grid), dim3(WAVES2 * 64), 0, 0, dW, dZ, items, dQ, reps, dcyc);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_eval_lds<true>, dim3(grid), dim3(WAVES2 * 64), 0, 0, dW, dZ, items, dQ, reps, dcyc);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        float ms = 0.0f;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> cyc(grid * WAVES2);
        CHECK(hipMemcpy(cyc.data(), dcyc, cyc.size() * 8, hipMemcpyDeviceToHost));
        double sum = 0;
        for (auto c : cyc) sum += (double)c;
        const double per_blk = sum / cyc.size() / (reps * 2);
        const double mfma = (double)items / 8 * reps * 108;
        printf("(3b) LDS-streamed A, 2 x 8-wave workgroups/CU (4 waves/SIMD): %.0f cycles per 8-item block per wave, wall %.3f ms, "
               "%.1f us per 65536+48000 items, MFMA pipe %.0f %% of 2.4 GHz peak\n",
               per_blk, ms, ms * 1e3 * (65536.0 + 48000.0) / ((double)items * reps),
               100.0 * mfma * 32.0 / (1024.0 * 2.4e9 * ms * 1e-3));
    }
    // ---- (3) timing
    for (int wgs_per_cu = 1; wgs_per_cu <= 2; ++wgs_per_cu) {
        const int grid = 256 * wgs_per_cu, reps = 8;
        const int items = grid * WAVES * 8 * 4;              // 4 column blocks per wave per rep
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k_eval<true>, dim3(grid), dim3(WAVES * 64), 0, 0, dW, dZ, items, dQ, reps, dcyc);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_eval<true>, dim3(grid), dim3(WAVES * 64), 0, 0, dW, dZ, items, dQ, reps, dcyc);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        float ms = 0.0f;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> cyc(grid * WAVES);
        CHECK(hipMemcpy(cyc.data(), dcyc, cyc.size() * 8, hipMemcpyDeviceToHost));
        double sum = 0;
        for (auto c : cyc) sum += (double)c;
        const double per_blk = sum / cyc.size() / (reps * 4);
        const double mfma = (double)items / 8 * reps * 108;
        printf("(3) %d workgroup(s)/CU: %.0f cycles per 8-item block per wave (108 MFMAs = 3456 issue cycles), wall %.3f ms, "
               "%.1f us per 65536+48000 items, MFMA pipe %.0f %% of 2.4 GHz peak\n",
               wgs_per_cu, per_blk, ms, ms * 1e3 * (65536.0 + 48000.0) / ((double)items * reps),
               100.0 * mfma * 32.0 / (1024.0 * 2.4e9 * ms * 1e-3));
    }
    return 0;
}
