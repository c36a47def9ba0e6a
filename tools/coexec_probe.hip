// Round 5, VERDICT r4 item 1(b): what does a vector stream get beside f32 MFMAs of other waves of the SAME SIMD on gfx950, as a
// function of the stream's SHAPE? profiles/r03_mfma_valu_coexec.txt (pure streams, 1 + 1 waves per SIMD) says the v_fma wave keeps
// 58 % of its solo rate beside a full-rate MFMA wave; profiles/r04_build_beside_mfma.txt (U2's build beside U2's products, 2 + 2
// waves, barrier per stage) found < 10 % overlap. This probe varies ONE property at a time between the two:
//   * the wave split per SIMD (MFMA waves + vector waves: 1+1, 1+3, 2+2, 3+1, 2+1, 1+2),
//   * independent accumulators per MFMA wave (3 / 6 / 12; 1 = a dependent chain),
//   * the vector stream: 8 independent v_fma chains (r03's), or the table build's complex-product chains over 1 / 2 / 3
//     interleaved items (ILP), with and without its ds_write_b64,
//   * the MFMA stream: register operands, or U2's product loop (5 ds_read_b64 + s_waitcnt per 12 MFMAs).
// One 1024-thread workgroup per CU (16 waves, wave w on SIMD w % 4), every wave loops over its role's unit of work for a fixed
// WINDOW of shader cycles (s_memtime) and reports how many units it finished: all streams overlap for the whole window, nobody
// runs alone at the end. Reported per role: units finished -> MFMA pipe busy % (32 cycles per 16x16x4 f32 MFMA) and vector
// instructions per cycle and SIMD, each also as a fraction of the same stream running WITHOUT the other role (same wave count).
// build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -o coexec_probe coexec_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
typedef float f4v __attribute__((ext_vector_type(4)));
constexpr unsigned long long WINDOW = 300000;      // shader cycles per measurement
constexpr int ROW = 260;

enum Role { IDLE = 0, M1, M3, M6, M12, M6_LDS, V_FMA8, V_B1, V_B2, V_B3, V_B1_ST, V_B2_ST, X_FMA2, X_FMA4, X_FMA6, X_FMA8, X_B36, X_B24, N_ROLES };
static const char *role_name[N_ROLES] = {"idle", "MFMA 1 acc", "MFMA 3 acc", "MFMA 6 acc", "MFMA 12 acc", "MFMA 6 acc + ds_read/waitcnt per 12",
                                         "v_fma x8 chains", "build ILP1", "build ILP2", "build ILP3", "build ILP1 + ds_write", "build ILP2 + ds_write",
                                         "ONE wave: MFMA + 2 v_fma each", "ONE wave: MFMA + 4 v_fma each", "ONE wave: MFMA + 6 v_fma each", "ONE wave: MFMA + 8 v_fma each",
                                         "ONE wave: 36 MFMA + 1 build", "ONE wave: 24 MFMA + 1 build"};
// vector instructions per unit of a role (counted from the source: mul/add/sub of the complex products; checked against the ISA)
static int valu_per_unit(int r) {
    switch (r) {
    case V_FMA8: return 64;
    case V_B1: return 234;       // 6 x 36 products' mul/add/sub + the sums that keep them alive (ISA count of the loop / units per trip)
    case V_B2: return 446;
    case V_B3: return 665;
    case V_B1_ST: return 210;
    case V_B2_ST: return 405;
    case X_FMA2: return 24; case X_FMA4: return 48; case X_FMA6: return 72; case X_FMA8: return 96;
    case X_B36: case X_B24: return 246;
    default: return 0;
    }
}
static int mfma_per_unit(int r) { return (r >= M1 && r <= M6_LDS) ? 12 : (r >= X_FMA2 && r <= X_FMA8) ? 12 : r == X_B36 ? 36 : r == X_B24 ? 24 : 0; }

struct Cfg { unsigned char role[16]; unsigned char prio[16]; };

// the table build of the step kernel's item_entries, reduced to its dependency structure: per item four unit complex numbers
// raised to powers 1..6 by repeated complex products (4 dependent chains), per power two products of pairs (ab, cd)
template <int ILP, bool STORE>
__device__ __forceinline__ float build_unit(float *tab, const float2 (&zin)[4], int wave, int lane, float seed) {
    float2 z[ILP][4], p[ILP][4];
    float keep = 0.0f;
#pragma unroll
    for (int i = 0; i < ILP; ++i) {
#pragma unroll
        for (int d = 0; d < 4; ++d) { z[i][d] = make_float2(zin[d].x + seed, zin[(d + i) & 3].y); p[i][d] = z[i][d]; }
    }
    const int slot = 8 * wave + (lane & 7), cp = lane >> 3;
#pragma unroll
    for (int c = 0; c < 6; ++c) {
#pragma unroll
        for (int i = 0; i < ILP; ++i) {
            const float2 ab = make_float2(p[i][0].x * p[i][1].x - p[i][0].y * p[i][1].y, p[i][0].x * p[i][1].y + p[i][0].y * p[i][1].x);
            const float2 cd = make_float2(p[i][2].x * p[i][3].x - p[i][2].y * p[i][3].y, p[i][2].x * p[i][3].y + p[i][2].y * p[i][3].x);
            if (STORE) {
                if (cp < 6) {
                    *reinterpret_cast<float2 *>(tab + (cp + 6 * c) * ROW + 2 * slot + 4 * i) = ab;
                    *reinterpret_cast<float2 *>(tab + (36 + cp + 6 * c) * ROW + 2 * slot + 4 * i) = cd;
                }
            } else keep += (ab.x + ab.y) + (cd.x + cd.y);
#pragma unroll
            for (int d = 0; d < 4; ++d)
                p[i][d] = make_float2(p[i][d].x * z[i][d].x - p[i][d].y * z[i][d].y, p[i][d].x * z[i][d].y + p[i][d].y * z[i][d].x);
        }
    }
#pragma unroll
    for (int i = 0; i < ILP; ++i) keep += p[i][0].x + p[i][1].y + p[i][2].x + p[i][3].y;
    return keep;
}

template <int NACC>
__device__ __forceinline__ void mfma_unit(f4v (&acc)[12], float a, float b) {      // 12 MFMAs over NACC independent accumulators
#pragma unroll
    for (int i = 0; i < 12; ++i) acc[i % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32((i & 1) ? a : b, (i & 2) ? a : b, acc[i % NACC], 0, 0, 0);
}
__device__ __forceinline__ void mfma_unit_lds(f4v (&acc)[12], const float *tab, int lane, int gi) {      // U2's product loop, one group
    const int n16 = lane & 15, g = lane >> 4;
    const float *pa = tab + n16 * ROW + 2 * g + 8 * gi, *pb = tab + (36 + n16) * ROW + 2 * g + 8 * gi;
    const float2 a2 = *reinterpret_cast<const float2 *>(pa), b2 = *reinterpret_cast<const float2 *>(pa + 16 * ROW);
    const float2 c0 = *reinterpret_cast<const float2 *>(pb), c1 = *reinterpret_cast<const float2 *>(pb + 16 * ROW),
                 c2 = *reinterpret_cast<const float2 *>(pb + 20 * ROW);
    acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(c0.x, a2.x, acc[0], 0, 0, 0); acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(c1.x, a2.x, acc[1], 0, 0, 0);
    acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(c2.x, a2.x, acc[2], 0, 0, 0); acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(c0.y, a2.y, acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(c1.y, a2.y, acc[1], 0, 0, 0); acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(c2.y, a2.y, acc[2], 0, 0, 0);
    acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(c0.x, b2.x, acc[3], 0, 0, 0); acc[4] = __builtin_amdgcn_mfma_f32_16x16x4f32(c1.x, b2.x, acc[4], 0, 0, 0);
    acc[5] = __builtin_amdgcn_mfma_f32_16x16x4f32(c2.x, b2.x, acc[5], 0, 0, 0); acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(c0.y, b2.y, acc[3], 0, 0, 0);
    acc[4] = __builtin_amdgcn_mfma_f32_16x16x4f32(c1.y, b2.y, acc[4], 0, 0, 0); acc[5] = __builtin_amdgcn_mfma_f32_16x16x4f32(c2.y, b2.y, acc[5], 0, 0, 0);
}

// ONE wave's stream holds both kinds: after every MFMA, NV independent v_fma (8 chains) — the order is pinned with
// sched_group_barrier (1 MFMA, NV VALU, ...), so the vector instructions issue while the wave's own MFMA occupies the matrix pipe
template <int NV>
__device__ __forceinline__ void mix_unit(f4v (&acc)[12], float (&v)[8], float a, float b) {
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        acc[i % 6] = __builtin_amdgcn_mfma_f32_16x16x4f32((i & 1) ? a : b, (i & 2) ? a : b, acc[i % 6], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NV; ++j) v[(i * NV + j) & 7] = fmaf(v[(i * NV + j) & 7], a, b);
    }
#pragma unroll
    for (int i = 0; i < 12; ++i) { __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); __builtin_amdgcn_sched_group_barrier(0x2, NV, 0); }
}
// ... or NM MFMAs spread over one table build (234 mul / add / sub of dependent complex products): the kernel's own ratio is 5.5 vector
// instructions per MFMA; 36 -> 6.5, 24 -> 9.75
template <int NM>
__device__ __forceinline__ float mix_build_unit(f4v (&acc)[12], const float2 (&zin)[4], float a, float b, float seed) {
    float2 z[4], p[4];
    float keep = 0.0f;
#pragma unroll
    for (int d = 0; d < 4; ++d) { z[d] = make_float2(zin[d].x + seed, zin[d].y); p[d] = z[d]; }
    int im = 0;
    auto M = [&]() {        // NM / 6 MFMAs per power c, one behind every segment of the build (sched_barrier: the scheduler keeps the order)
        __builtin_amdgcn_sched_barrier(0);
        acc[im % 6] = __builtin_amdgcn_mfma_f32_16x16x4f32((im & 1) ? a : b, (im & 2) ? a : b, acc[im % 6], 0, 0, 0);
        ++im;
        __builtin_amdgcn_sched_barrier(0);
    };
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        const float2 ab = make_float2(p[0].x * p[1].x - p[0].y * p[1].y, p[0].x * p[1].y + p[0].y * p[1].x);
        M();
        const float2 cd = make_float2(p[2].x * p[3].x - p[2].y * p[3].y, p[2].x * p[3].y + p[2].y * p[3].x);
        M();
        keep += (ab.x + ab.y) + (cd.x + cd.y);
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            p[d] = make_float2(p[d].x * z[d].x - p[d].y * z[d].y, p[d].x * z[d].y + p[d].y * z[d].x);
            if (NM == 36 || d < 2) M();
        }
    }
    keep += p[0].x + p[1].y + p[2].x + p[3].y;
    return keep;
}

struct Out { float keep; unsigned units; unsigned cyc; };
// every role's loop is its own (non-inlined) function: the ISA of one role is one symbol, countable by tools/coexec_isa_count.py
template <int ROLE>
__device__ __attribute__((noinline)) Out run_role(const float *tab_r, float *tab_w, int wave, int lane, unsigned long long t0) {
    f4v acc[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) acc[i] = (f4v){0.0f, 0.0f, 0.0f, 0.0f};
    float a = 1.0f + lane * 1e-3f, b = 1.0f - lane * 1e-3f, keep = 0.0f;
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = a + i;
    float2 zin[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) zin[d] = *reinterpret_cast<const float2 *>(tab_r + 2 * lane + 130 * d);
    unsigned long long t1 = t0;
    unsigned units = 0;
    constexpr int REPS = (ROLE == X_B36 || ROLE == X_B24) ? 1 : (ROLE == V_B2_ST || ROLE == V_B3) ? 2 : 4;      // (unrolled units per loop trip: kept below the register budget)
#pragma unroll 1
    for (;;) {
#pragma unroll
        for (int rep = 0; rep < REPS; ++rep) {
            if (ROLE == M1) mfma_unit<1>(acc, a, b);
            if (ROLE == M3) mfma_unit<3>(acc, a, b);
            if (ROLE == M6) mfma_unit<6>(acc, a, b);
            if (ROLE == M12) mfma_unit<12>(acc, a, b);
            if (ROLE == M6_LDS) mfma_unit_lds(acc, tab_r, lane, (int)((units + rep) % 7));
            if (ROLE == V_FMA8) {
#pragma unroll
                for (int r = 0; r < 8; ++r) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = fmaf(v[i], a, b);
                }
            }
            if (ROLE == V_B1) keep += build_unit<1, false>(tab_w, zin, wave, lane, keep * 1e-30f);
            if (ROLE == V_B2) keep += build_unit<2, false>(tab_w, zin, wave, lane, keep * 1e-30f);
            if (ROLE == V_B3) keep += build_unit<3, false>(tab_w, zin, wave, lane, keep * 1e-30f);
            if (ROLE == V_B1_ST) keep += build_unit<1, true>(tab_w, zin, wave, lane, 1e-3f * ((units + rep) & 7));
            if (ROLE == V_B2_ST) keep += build_unit<2, true>(tab_w, zin, wave, lane, 1e-3f * ((units + rep) & 7));
            if (ROLE == X_FMA2) mix_unit<2>(acc, v, a, b);
            if (ROLE == X_FMA4) mix_unit<4>(acc, v, a, b);
            if (ROLE == X_FMA6) mix_unit<6>(acc, v, a, b);
            if (ROLE == X_FMA8) mix_unit<8>(acc, v, a, b);
            if (ROLE == X_B36) keep += mix_build_unit<36>(acc, zin, a, b, keep * 1e-30f);
            if (ROLE == X_B24) keep += mix_build_unit<24>(acc, zin, a, b, keep * 1e-30f);
        }
        units += REPS;
        t1 = __builtin_amdgcn_s_memtime();
        if (t1 - t0 >= WINDOW) break;
    }
#pragma unroll
    for (int i = 0; i < 12; ++i) keep += acc[i][0] + acc[i][3];
#pragma unroll
    for (int i = 0; i < 8; ++i) keep += v[i];
    return {keep, units, (unsigned)(t1 - t0)};
}

__global__ __launch_bounds__(1024) void k(const Cfg cfg, unsigned *units_out, unsigned *cycles_out, float *sink) {
    extern __shared__ float lds[];                         // one operand table (72 rows) + one build target (72 rows)
    float *tab_r = lds, *tab_w = lds + 72 * ROW;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int role = __builtin_amdgcn_readfirstlane((int)cfg.role[wave]);
    for (int i = threadIdx.x; i < 2 * 72 * ROW; i += 1024) lds[i] = 0.6f + 1e-4f * (i & 1023);
    __syncthreads();
    switch (__builtin_amdgcn_readfirstlane((int)cfg.prio[wave])) {      // (s_setprio takes an immediate)
    case 1: __builtin_amdgcn_s_setprio(1); break;
    case 2: __builtin_amdgcn_s_setprio(2); break;
    case 3: __builtin_amdgcn_s_setprio(3); break;
    default: break;
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    Out o = {0.0f, 0u, 0u};
    switch (role) {
    case M1: o = run_role<M1>(tab_r, tab_w, wave, lane, t0); break;
    case M3: o = run_role<M3>(tab_r, tab_w, wave, lane, t0); break;
    case M6: o = run_role<M6>(tab_r, tab_w, wave, lane, t0); break;
    case M12: o = run_role<M12>(tab_r, tab_w, wave, lane, t0); break;
    case M6_LDS: o = run_role<M6_LDS>(tab_r, tab_w, wave, lane, t0); break;
    case V_FMA8: o = run_role<V_FMA8>(tab_r, tab_w, wave, lane, t0); break;
    case V_B1: o = run_role<V_B1>(tab_r, tab_w, wave, lane, t0); break;
    case V_B2: o = run_role<V_B2>(tab_r, tab_w, wave, lane, t0); break;
    case V_B3: o = run_role<V_B3>(tab_r, tab_w, wave, lane, t0); break;
    case V_B1_ST: o = run_role<V_B1_ST>(tab_r, tab_w, wave, lane, t0); break;
    case V_B2_ST: o = run_role<V_B2_ST>(tab_r, tab_w, wave, lane, t0); break;
    case X_FMA2: o = run_role<X_FMA2>(tab_r, tab_w, wave, lane, t0); break;
    case X_FMA4: o = run_role<X_FMA4>(tab_r, tab_w, wave, lane, t0); break;
    case X_FMA6: o = run_role<X_FMA6>(tab_r, tab_w, wave, lane, t0); break;
    case X_FMA8: o = run_role<X_FMA8>(tab_r, tab_w, wave, lane, t0); break;
    case X_B36: o = run_role<X_B36>(tab_r, tab_w, wave, lane, t0); break;
    case X_B24: o = run_role<X_B24>(tab_r, tab_w, wave, lane, t0); break;
    default: break;
    }
    sink[blockIdx.x * 1024 + threadIdx.x] = o.keep;
    if (lane == 0) { units_out[blockIdx.x * 16 + wave] = o.units; cycles_out[blockIdx.x * 16 + wave] = o.cyc; }
}

struct Res { double mfma_busy, valu_ipc; };      // per SIMD: fraction of cycles the matrix pipe is busy; vector instructions per cycle
static unsigned *d_units, *d_cyc; static float *d_sink;
static Res run(const Cfg &c) {
    const int n = 256; const size_t lds = 2 * 72 * ROW * 4;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k, dim3(n), dim3(1024), lds, 0, c, d_units, d_cyc, d_sink);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); exit(1); }
    std::vector<unsigned> u(n * 16), cy(n * 16);
    (void)hipMemcpy(u.data(), d_units, n * 16 * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(cy.data(), d_cyc, n * 16 * 4, hipMemcpyDeviceToHost);
    double m = 0, v = 0;
    for (int bk = 0; bk < n; ++bk) for (int w = 0; w < 16; ++w) {
        const int r = c.role[w]; if (!r) continue;
        const double un = u[bk * 16 + w], cyc = cy[bk * 16 + w];
        m += un * mfma_per_unit(r) * 32.0 / cyc; v += un * valu_per_unit(r) / cyc;
    }
    return {m / (n * 4.0), v / (n * 4.0)};      // 4 SIMDs per CU; wave w runs on SIMD w % 4, so per-SIMD = sum over its waves
}
// nm MFMA waves + nv vector waves on EVERY SIMD (waves w, w+4, w+8, w+12 share one); vfirst: the vector waves take the lower wave ids
// (the older waves of the SIMD); vprio / mprio: s_setprio of the two kinds
static Cfg make(int nm, int mrole, int nv, int vrole, bool vfirst = false, int vprio = 0, int mprio = 0) {
    Cfg c; memset(&c, 0, sizeof c);
    for (int s = 0; s < 4; ++s) {
        int slot = 0;
        if (vfirst) for (int i = 0; i < nv; ++i) { c.role[s + 4 * slot] = (unsigned char)vrole; c.prio[s + 4 * slot++] = (unsigned char)vprio; }
        for (int i = 0; i < nm; ++i) { c.role[s + 4 * slot] = (unsigned char)mrole; c.prio[s + 4 * slot++] = (unsigned char)mprio; }
        if (!vfirst) for (int i = 0; i < nv; ++i) { c.role[s + 4 * slot] = (unsigned char)vrole; c.prio[s + 4 * slot++] = (unsigned char)vprio; }
    }
    return c;
}

int main() {
    (void)hipMalloc(&d_units, 256 * 16 * 4); (void)hipMalloc(&d_cyc, 256 * 16 * 4); (void)hipMalloc(&d_sink, 256 * 1024 * 4);
    (void)hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 72 * ROW * 4);
    printf("per SIMD, one 16-wave workgroup per CU, %llu-cycle window; 'alone' = the same waves of that role with the other role idle\n", WINDOW);
    printf("%-40s %-26s | %-22s | %-34s | %s\n", "split per SIMD (MFMA role + vector role)", "", "matrix pipe busy", "vector instr / cycle (x4 = lane-cycles/16)", "combined (pipe busy + vector/alone-of-4)");
    const int mroles[] = {M6, M3, M12, M1, M6_LDS};
    const int vroles[] = {V_FMA8, V_B1, V_B2, V_B3, V_B1_ST, V_B2_ST};
    // solo rates
    double v_solo[N_ROLES][5] = {}, m_solo[N_ROLES][5] = {};
    for (int vr : vroles) for (int nv = 1; nv <= 4; ++nv) v_solo[vr][nv] = run(make(0, 0, nv, vr)).valu_ipc;
    for (int mr : mroles) for (int nm = 1; nm <= 3; ++nm) m_solo[mr][nm] = run(make(nm, mr, 0, 0)).mfma_busy;
    for (int vr : vroles) printf("solo %-26s 1..4 waves per SIMD: %.3f %.3f %.3f %.3f vector instr / cycle\n", role_name[vr], v_solo[vr][1], v_solo[vr][2], v_solo[vr][3], v_solo[vr][4]);
    for (int mr : mroles) printf("solo %-40s 1..3 waves per SIMD: %.1f %% %.1f %% %.1f %% pipe busy\n", role_name[mr], 100 * m_solo[mr][1], 100 * m_solo[mr][2], 100 * m_solo[mr][3]);
    const int splits[][2] = {{1, 1}, {1, 3}, {2, 2}, {3, 1}, {2, 1}, {1, 2}};
    auto line = [&](int nm, int mr, int nv, int vr) {
        const Res r = run(make(nm, mr, nv, vr));
        printf("%d x %-36s + %d x %-22s | %5.1f %% (alone %5.1f %%) | %.3f (alone %.3f = %3.0f %% kept) | %.2f\n", nm, role_name[mr], nv, role_name[vr],
               100 * r.mfma_busy, 100 * m_solo[mr][nm], r.valu_ipc, v_solo[vr][nv], 100 * r.valu_ipc / v_solo[vr][nv], r.mfma_busy + r.valu_ipc / v_solo[vr][4]);
    };
    printf("--- wave split, pure streams (MFMA 6 acc + v_fma x8)\n");
    for (auto &s : splits) line(s[0], M6, s[1], V_FMA8);
    printf("--- wave split, the build's chains (ILP1) beside MFMA 6 acc\n");
    for (auto &s : splits) line(s[0], M6, s[1], V_B1);
    printf("--- ILP of the build (2 + 2 and 1 + 3)\n");
    for (int vr : {V_B1, V_B2, V_B3}) { line(2, M6, 2, vr); line(1, M6, 3, vr); }
    printf("--- independent accumulators per MFMA wave (2 + 2, build ILP1 and v_fma)\n");
    for (int mr : {M1, M3, M6, M12}) { line(2, mr, 2, V_B1); line(2, mr, 2, V_FMA8); }
    printf("--- LDS on either side (2 + 2)\n");
    line(2, M6, 2, V_B1_ST); line(2, M6_LDS, 2, V_B1); line(2, M6_LDS, 2, V_B1_ST); line(2, M6_LDS, 2, V_B2_ST); line(3, M6_LDS, 1, V_B1_ST); line(1, M6_LDS, 3, V_B1_ST);
    printf("--- who wins the issue port: wave age (vector waves = the lower wave ids) and s_setprio\n");
    auto line2 = [&](int nm, int mr, int nv, int vr, bool vfirst, int vprio, int mprio) {
        const Res r = run(make(nm, mr, nv, vr, vfirst, vprio, mprio));
        printf("%d x %-22s + %d x %-22s %-22s prio M %d V %d | %5.1f %% (alone %5.1f %%) | %.3f (alone %.3f = %3.0f %% kept) | %.2f\n", nm, role_name[mr], nv, role_name[vr],
               vfirst ? "vector waves older" : "MFMA waves older", mprio, vprio, 100 * r.mfma_busy, 100 * m_solo[mr][nm], r.valu_ipc, v_solo[vr][nv], 100 * r.valu_ipc / v_solo[vr][nv],
               r.mfma_busy + r.valu_ipc / v_solo[vr][4]);
    };
    for (int vr : {V_FMA8, V_B1}) {
        line2(2, M6, 2, vr, true, 0, 0); line2(2, M6, 2, vr, false, 3, 0); line2(2, M6, 2, vr, true, 3, 0); line2(2, M6, 2, vr, false, 0, 3);
        line2(1, M6, 3, vr, true, 0, 0); line2(1, M6, 3, vr, true, 3, 0); line2(1, M6, 1, vr, true, 3, 0); line2(3, M6, 1, vr, true, 3, 0);
    }
    printf("--- both kinds in ONE wave's instruction stream (n waves per SIMD all running the mixed stream)\n");
    for (int xr : {X_FMA2, X_FMA4, X_FMA6, X_FMA8, X_B36, X_B24})
        for (int n = 1; n <= 4; ++n) {
            const Res r = run(make(n, xr, 0, 0));
            printf("%d x %-32s | pipe busy %5.1f %% | %.3f vector instr / cycle (%.0f %% of the 4-wave solo rate of that stream) | combined %.2f\n", n, role_name[xr], 100 * r.mfma_busy, r.valu_ipc,
                   100 * r.valu_ipc / v_solo[xr >= X_B36 ? V_B1 : V_FMA8][4], r.mfma_busy + r.valu_ipc / v_solo[xr >= X_B36 ? V_B1 : V_FMA8][4]);
        }
    return 0;
}
