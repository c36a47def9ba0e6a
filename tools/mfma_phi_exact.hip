// mfma_phi_exact.hip — does v_mfma_f32_32x32x2_f32 reproduce SPEC §3's phi = fma(-Im(AB), Im(CD), Re(AB)*Re(CD))
// bit for bit? One MFMA forms the rank-2 outer product D[i][j] = A[i][0] B[0][j] + A[i][1] B[1][j] for 32 values
// of c12 (rows) and 32 of c34 (columns) of ONE item — 1024 of the 1296 Fourier features per instruction, on the
// matrix pipe instead of 16 packed VALU ops. Prints the accumulator layout and compares against both fma orders.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

typedef float f16v __attribute__((ext_vector_type(16)));

__global__ void k(const float *A /*[32][2]*/, const float *B /*[2][32]*/, float *D /*[64][16]*/) {
    const int l = threadIdx.x;
    const float a = A[(l & 31) * 2 + (l >> 5)];     // A[i][k]: lane = i + 32 k
    const float b = B[(l >> 5) * 32 + (l & 31)];    // B[k][j]: lane = j + 32 k
    f16v c;
#pragma unroll
    for (int v = 0; v < 16; ++v) c[v] = 0.0f;
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
#pragma unroll
    for (int v = 0; v < 16; ++v) D[l * 16 + v] = c[v];
}

int main() {
    std::vector<float> A(64), B(64), D(1024);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)((s >> 8) & 0xffff) / 65536.0f * 2.0f - 1.0f; };
    for (auto &x : A) x = rnd();
    for (auto &x : B) x = rnd();
    float *dA, *dB, *dD;
    (void)hipMalloc(&dA, 256); (void)hipMalloc(&dB, 256); (void)hipMalloc(&dD, 4096);
    (void)hipMemcpy(dA, A.data(), 256, hipMemcpyHostToDevice);
    (void)hipMemcpy(dB, B.data(), 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    if (hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost) != hipSuccess) { printf("no device\n"); return 1; }
    // find the layout: for lane l, register v, which (i, j) does the value belong to?
    int n01 = 0, n10 = 0, nplain = 0, nfound = 0;
    bool layout_ok = true;
    for (int l = 0; l < 64; ++l)
        for (int v = 0; v < 16; ++v) {
            const int j = l & 31, i = 8 * (v >> 2) + 4 * (l >> 5) + (v & 3);       // expected layout
            const float d = D[l * 16 + v];
            const float p0 = A[i * 2 + 0] * B[j], p1 = A[i * 2 + 1] * B[32 + j];
            const float c01 = fmaf(A[i * 2 + 1], B[32 + j], p0);   // k = 0 first: fma(a1, b1, round(a0 b0))
            const float c10 = fmaf(A[i * 2 + 0], B[j], p1);        // k = 1 first
            const float cpl = p0 + p1;                             // both products rounded
            const bool e01 = !memcmp(&d, &c01, 4), e10 = !memcmp(&d, &c10, 4), epl = !memcmp(&d, &cpl, 4);
            n01 += e01; n10 += e10; nplain += epl; nfound += (e01 || e10 || epl);
            if (fabsf(d - c01) > 1e-5f) layout_ok = false;
        }
    printf("layout D[lane l][reg v] = (i = 8 (v>>2) + 4 (l>>5) + (v&3), j = l&31): %s\n", layout_ok ? "confirmed" : "WRONG");
    printf("of 1024 outputs: %d == fma(a1,b1,round(a0*b0))   %d == fma(a0,b0,round(a1*b1))   %d == round(a0*b0)+round(a1*b1)   (%d match one of them)\n",
           n01, n10, nplain, nfound);
    return 0;
}
