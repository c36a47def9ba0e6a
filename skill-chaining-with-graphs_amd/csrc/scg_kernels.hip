// scg_kernels.hip — gfx950 kernels + the C-ABI of include/scg_abi.h.
//
// One step-batch (SPEC §5) = td_kernel<FUSED> -> reduce_kernel (+ sort_hist / sort_scatter when no env order is
// prepared). td_kernel: a workgroup = 8 wavefronts owns 128 consecutive positions of the option-sorted env order; two
// workgroups per CU, <= 128 VGPRs, four waves per SIMD.
//   phase P  (waves 0..1, one lane per env)  act from qcache, Pinball physics (cell mask -> exact refine; envs without a
//            candidate edge fly free in place, the (env, edge) pairs of the others are dealt to the lanes of waves 0..3: one
//            intercept per lane and sub-step, hits combined by ballot), reset/bookkeeping, option logic; MEANWHILE waves 4..7
//            stage W_0, take Z_d^1 of the entry states, build the root's update list and run U1 of the root pass
//   phase Z  Z_d^1 = sincospi of the four normalised state variables of s_next -> LDS
//   phase TD, value function by value function, on the matrix pipe (v_mfma_f32_16x16x4_f32 is bit for bit a k-ordered
//            fmaf chain, so the CPU oracle reproduces every sum):
//     E   Q_k(s_next, .) of 8 items per wave-iteration: T[(a,c12)][item re|im] = W_k (180 x 36, staged in LDS in
//         A-operand order, one ds_read_b128 per four MFMAs) x CD (36 x 16, from a per-wave LDS table) = 108 MFMAs, then
//         per lane 48 fmas with the AB factors and a 3-stage butterfly (SPEC §3.1)
//     U1  Q_k(s, a_t) per action run: the same contraction on the 3 row tiles of action a_t
//     U2  the block partial G_b,k[a] (36 x 36) += P (36 x 2n, delta-scaled AB factors) x C^T (2n x 36, CD factors):
//         9 output tiles per action dealt over the 8 waves, accumulators stay in registers for the whole pass and
//         go straight to the block's slab (no cross-wave reduction)
//   tail     value functions that only have envs ENTERING them here: the same chains on the vector pipe, per wave
//   reduce_kernel  slabs -> 16-block segment sums -> G, n_k, W += alpha/n_k * scale * G; commit + next env order (+ an
//                  announced example trigger's row totals)
// fit_kernel: SPEC §6 on 8 workgroups x 1024 chains per option behind tagged-word exchanges; a fit whose workgroups cannot run
// together gives up after a wall-clock wait, leaves its row untouched and raises the ctx's asynchronous status word.
// Every sum has the pinned order of SPEC §3.1 / §5 / §6 (no atomics on data): the CPU oracle reproduces every bit.
// No upstream code exists to cite (reference = README.md:1-2); sections cite SPEC.md.
#include "scg_device.hpp"
#include "../../include/scg_abi.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <new>
#include <type_traits>
#include <vector>

using namespace scg;
typedef float f4v __attribute__((ext_vector_type(4)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------
constexpr int P_WAVES = BLOCK_ENVS / 64;      // phase P: one lane per env on full waves
constexpr int HELPER0 = WAVES / 2;            // waves HELPER0.. work under phase P (learning steps)
constexpr int P_POOL = HELPER0;               // waves 0..P_POOL-1 share the physics' (env, edge) pair groups
static_assert(P_WAVES == 2, "the pair-group owner lookup below is written for two env waves");
// LDS map of the step kernel (bytes). 8 wavefronts per workgroup, two workgroups per CU (80 KB each).
constexpr int OFF_RC = 0;                                      // float r0,c0,ro,co (per env), rk,ck (per env, this pass) [128]
constexpr int OFF_INT = OFF_RC + 6 * BLOCK_ENVS * 4;           // uint8 a, ot, on, gs, ia [128]
constexpr int OFF_Z1 = OFF_INT + 5 * BLOCK_ENVS;               // float2 z1[128][2][4]: Z_d^1 of s and s_next
// region R, used by one phase at a time:
//   P, Z  : s[4][128], sn[4][128] (the envs' states) and the edge table [256][8]
//   E, U1 : W_k staged in A-operand order (12 row tiles x 9 k-blocks x 64 lanes) + per wave CDk[36][16] + ABq[16][AS]
//   U2    : PT[36][US], CDT[36][US] (one chunk of U2_CH padded slots = 2 U2_CH K-steps)
constexpr int W_FLOATS = 12 * 9 * 64;
constexpr int W_TAIL = 12 * 2 * 64 * 4;                        // k-block 8 of every tile sits behind the two float4 groups
constexpr int AS = 40;                                         // row stride of ABq (floats): 16-byte rows, and the ds_read_b128 of the
                                                               // fold (16-lane groups mixing row groups g, g + 1) conflict-free: 36 gave 2-way
constexpr int E_TAB_FLOATS = 36 * 16 + 16 * AS;
constexpr int U2_CH = 72;                                      // U2 chunk: padded slots (9 per wave: 54 of 64 builder lanes busy, and the root's
                                                               // <= 143 slots are always TWO chunks; 64-slot chunks left a third one of ~8 slots)
constexpr int US = 2 * U2_CH + 4;                              // row stride of the chunk tables (floats): rows 20 banks apart, operand reads conflict-free
constexpr int R_TAB = W_FLOATS;                                // private tables start behind the staged W_k
constexpr int R_S = R_TAB, R_EDGES = R_TAB + 8 * BLOCK_ENVS, R_PITEMS = R_EDGES + MAX_EDGES * 8;   // phases P / Z: inside the table area of waves 0..3 (the helper
                                                               // waves 4..7 use region W and their own tables meanwhile)
constexpr int cmax3(int a, int b, int c) { return a > b ? (a > c ? a : c) : (b > c ? b : c); }
constexpr int R_FLOATS = cmax3(W_FLOATS + WAVES * E_TAB_FLOATS, 2 * 36 * US, R_EDGES + MAX_EDGES * 8);
constexpr int OFF_R = OFF_Z1 + BLOCK_ENVS * 2 * 4 * 8;
constexpr int OFF_ELIST = OFF_R + R_FLOATS * 4;                // uint16 eval list[128]
constexpr int OFF_ULIST = OFF_ELIST + BLOCK_ENVS * 2;          // uint16 update list[128] (5 action runs)
constexpr int OFF_MAXQ = OFF_ULIST + BLOCK_ENVS * 2;           // float maxq[128] (per env)
constexpr int OFF_QSA = OFF_MAXQ + BLOCK_ENVS * 4;             // float qsa[128] (per update-list position)
constexpr int OFF_ENV = OFF_QSA + BLOCK_ENVS * 4;              // int env[128]: env index of each block slot
constexpr int OFF_CLF = OFF_ENV + BLOCK_ENVS * 4;              // float clf[6][8]
constexpr int OFF_MISC = OFF_CLF + MAX_VF * CLF_STRIDE * 4;    // int misc[32]
#ifdef SCG_STAMPS
constexpr int OFF_STAMP = OFF_MISC + 128;                      // unsigned stamp[32] (diagnostic build)
constexpr int LDS_BYTES = OFF_STAMP + 128;
#else
constexpr int LDS_BYTES = OFF_MISC + 128;
#endif
static_assert(LDS_BYTES <= 80 * 1024, "LDS budget: two workgroups per CU");
static_assert(OFF_Z1 % 16 == 0 && OFF_R % 16 == 0 && (R_TAB * 4) % 16 == 0 && (R_EDGES * 4) % 16 == 0, "LDS alignment");
static_assert(8 * BLOCK_ENVS + MAX_EDGES * 8 + P_WAVES * PITEMS <= HELPER0 * E_TAB_FLOATS, "states + edges + the physics pair lists fit the table area of the waves below the helpers");

enum { MODE_FUSED = 0, MODE_TRANS = 1, MODE_QVAL = 2 };

// Diagnostic build only (make stamps -> libscg_hip_stamps.so, tools/stamp_report.py): wave 0 of every
// workgroup accumulates s_memtime deltas per kernel section into A.stamps[block][section]. The shipped
// library is built without SCG_STAMPS and contains none of this.
#ifdef SCG_STAMPS

#define SCG_STAMP(SEC)                                                                   \
    do {                                                                                 \
        if (MODE == MODE_FUSED && A.stamps && tid == 0) {                                \
            const unsigned long long t_ = __builtin_amdgcn_s_memtime();                  \
            s_stamp[(SEC)] += (unsigned)(t_ - stamp_prev);                               \
            stamp_prev = t_;                                                             \
        }                                                                                \
    } while (0)
#else
#define SCG_STAMP(SEC) do { } while (0)
#endif

template <typename T>
__device__ __forceinline__ void gstore(T *p, T v) { *p = v; }

// 16-byte WRITE-THROUGH store (sc1) for data that only the NEXT launch reads — the slabs: 27 MB per step-batch. A plain store
// leaves the lines dirty in the XCD's L2 and the kernel boundary then waits for their write-back (MI355X_MICROARCH.md, "boundary":
// + B / 6 TB/s); written through, they are in the memory-side cache by then, where the reduce launch (other XCDs) reads them
// anyway: +1.1 % env-steps/s. (`nt` stores, round 2, bypassed that cache too and cost the reduce launch 5.7 us.)
__device__ __forceinline__ void store_wt(float *p, f4v v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
}

struct StepArgs {
    // env state (FUSED: in/out; TRANS/QVAL: in)
    float *x, *y, *vx, *vy;
    int32_t *option_id, *opt_steps, *ep_steps;
    int32_t *hist_next;        // [rows of 256 envs][8] counts of the option ids this step leaves (null = off)
    float4 *outrec;            // FUSED: [positions][4] per-env results in env-ORDER position (one 64-byte line:
                               // state', {reward, bits, counters}, Q(s', .) of the VF acting next), committed to the
                               // caller's arrays by commit_row (coalesced) instead of 4-byte scatters from here
    float *qcache;                 // [5][n]  (QVAL: output q)
    uint8_t *action;               // FUSED: out; TRANS: in
    float *reward;                 // FUSED: out; TRANS: in (r)
    uint8_t *done;
    const float *cont_in;          // TRANS
    const float *xn, *yn, *vxn, *vyn;   // TRANS
    const float *W;                // [n_vf][5][1296] (QVAL: one VF)
    const float *clf;              // [n_vf][8]
    const float *edges;            // device [n_edges][8]
    const uint64_t *cellmask;      // device [32*32][4] candidate-edge masks per grid cell
    const int32_t *perm;           // FUSED: envs in (option_id, env) order (SPEC §5); NULL = identity
    float *ring_x, *ring_y;        // SPEC §7 trace buffers (NULL = off)
    uint8_t *events;
    int32_t *ev_len;
    int32_t ring_mask;             // ring_len - 1
    const float *starts;           // device [n_starts][2]
    float *slabs;                  // [nblk][n_vf][5][1296]
    int32_t *cnts;                 // [nblk][n_vf]
    unsigned long long *stamps;    // diagnostic build only
    int32_t n, n_vf, k_lo, k_hi;
    uint32_t enabled, learn;
    uint32_t gest;                 // SPEC §4.4: options in gestation (classifier known, not selectable, learning off-policy)
    int32_t *gest_succ;            // [n_vf] successes seen from inside a gestating option's initiation set (atomic counts)
    uint32_t parents;              // 3 bits per option k at [3k, 3k+3): target option of k (0 = the task goal)
    uint64_t t, seed;
    int64_t env_base;
    float gamma, epsilon, r_succ;
    int32_t max_ep, max_opt;
    MapScalars ms;
};

__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Workgroup barrier for LDS hand-offs only. __syncthreads() carries a workgroup-scope release fence, which on
// gfx9 means s_waitcnt vmcnt(0): every barrier after a global store waits for the store to be acknowledged.
// Nothing in td_kernel passes data between threads through global memory, so the barriers only need this wave's
// LDS traffic done.
__device__ __forceinline__ void block_lds_sync() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

__device__ __forceinline__ bool in_set(const StepArgs &A, int k, float x, float y) {
    if (k < 1 || k >= A.n_vf) return false;
    if (!((A.enabled >> k) & 1u)) return false;
    return clf_z(A.clf + CLF_STRIDE * k, x, y) > 0.0f;
}

// ------------------------------------------------------------------------------------------------
// SPEC §3 tables of one item from its four Z_d^1 (z1p: 4 float2 in LDS). The calling lane owns second index
// `cp` (c2 of AB, c4 of CD; cp < 6) and gets, for the first index c = 0..5, AB[6c + cp] and CD[6c + cp]:
// AB[c2] = Z_1^c2, AB[c1*6 + c2] = cmul(AB[(c1-1)*6 + c2], Z_0^1) (CD likewise from Z_3, Z_2), Z^0 = (1, 0), Z^k = cmul(Z^(k-1), Z^1):
// five chained products per table column instead of a power chain plus a product per entry.
__device__ __forceinline__ float2 zpow_sel(float2 z, int c) {
    float2 cur = z, out = make_float2(1.0f, 0.0f);
#pragma unroll
    for (int j = 1; j <= 5; ++j) {
        if (c == j) out = cur;
        if (j < 5) cur = cmul(cur, z);
    }
    return out;
}
__device__ __forceinline__ void item_entries(const float2 *z1p, int cp, float2 (&ab)[6], float2 (&cd)[6]) {
    const float4 za = *reinterpret_cast<const float4 *>(z1p), zc = *reinterpret_cast<const float4 *>(z1p + 2);
    const float2 z0 = make_float2(za.x, za.y), z1 = make_float2(za.z, za.w);
    const float2 z2 = make_float2(zc.x, zc.y), z3 = make_float2(zc.z, zc.w);
    ab[0] = zpow_sel(z1, cp); cd[0] = zpow_sel(z3, cp);
#pragma unroll
    for (int c = 1; c < 6; ++c) {
        ab[c] = cmul(ab[c - 1], z0);
        cd[c] = cmul(cd[c - 1], z2);
    }
}

// SPEC §3.1 butterfly over the 16 partial sums of one item (4 row groups x re|im, the item's 8 columns hold
// [re x 4 items, im x 4 items]): u_g = q_re + q_im (lane xor 4), then (u_0 + u_1) + (u_2 + u_3) (lane xor 16, 32)
template <int M>
__device__ __forceinline__ void item_tree_sum(float (&q)[M]) {
#pragma unroll
    for (int a = 0; a < M; ++a) q[a] = q[a] + swz_xor4(q[a]);
#pragma unroll
    for (int a = 0; a < M; ++a) {
        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(q[a]), __float_as_uint(q[a]), false, false);
        q[a] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
#pragma unroll
    for (int a = 0; a < M; ++a) {
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(q[a]), __float_as_uint(q[a]), false, false);
        q[a] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
}

template <int MODE>
__global__ __launch_bounds__(THREADS, 4) void td_kernel(const StepArgs A) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *s_s = reinterpret_cast<float *>(smem + OFF_R) + R_S;        // [8][128]: s then sn   (region R, phases P and Z)
    float *s_edges = reinterpret_cast<float *>(smem + OFF_R) + R_EDGES;  // [n_edges][8]        (region R, phase P)
    uint32_t *s_pitems = reinterpret_cast<uint32_t *>(smem + OFF_R) + R_PITEMS;   // [P_WAVES][PITEMS] (env, edge) pairs of the physics
    float *s_r0 = reinterpret_cast<float *>(smem + OFF_RC);
    float *s_c0 = s_r0 + BLOCK_ENVS, *s_ro = s_c0 + BLOCK_ENVS, *s_co = s_ro + BLOCK_ENVS;
    float *s_rk = s_co + BLOCK_ENVS, *s_ck = s_rk + BLOCK_ENVS;       // reward / continuation of the pass's value function
    uint8_t *s_a = reinterpret_cast<uint8_t *>(smem + OFF_INT);
    uint8_t *s_ot = s_a + BLOCK_ENVS, *s_on = s_ot + BLOCK_ENVS;
    uint8_t *s_gs = s_on + BLOCK_ENVS;      // bit k: gestating option k holds s in its initiation set (off-policy item)
    uint8_t *s_ia = s_gs + BLOCK_ENVS;      // bit 0: goal; bit k: in_k(s')
    float2 *s_z1 = reinterpret_cast<float2 *>(smem + OFF_Z1);
    float *s_R = reinterpret_cast<float *>(smem + OFF_R);
    uint16_t *s_elist = reinterpret_cast<uint16_t *>(smem + OFF_ELIST);
    uint16_t *s_ulist = reinterpret_cast<uint16_t *>(smem + OFF_ULIST);
    float *s_maxq = reinterpret_cast<float *>(smem + OFF_MAXQ);
    float *s_qsa = reinterpret_cast<float *>(smem + OFF_QSA);
    float *s_W = s_R;                                                   // region W = the head of region R
    int *s_env = reinterpret_cast<int *>(smem + OFF_ENV);
    float *s_clf = reinterpret_cast<float *>(smem + OFF_CLF);
    int *s_misc = reinterpret_cast<int *>(smem + OFF_MISC);

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x;
    const int e0 = b * BLOCK_ENVS;
    const int nb = min(BLOCK_ENVS, A.n - e0);
    const int N = A.n;
#ifdef SCG_STAMPS
    unsigned *s_stamp = reinterpret_cast<unsigned *>(smem + OFF_STAMP);
    if (tid < 32) s_stamp[tid] = 0;
    unsigned long long stamp_prev = __builtin_amdgcn_s_memtime();
#endif

    if (MODE == MODE_FUSED) {
        for (int i = tid; i < A.ms.n_edges * 8; i += THREADS) s_edges[i] = A.edges[i];
        if (tid == 0) { s_misc[31] = 1; s_misc[30] = 1; s_misc[29] = 0; s_misc[28] = 0; s_misc[25] = 0; s_misc[24] = 0; }  // [31]/[30] bit k: some env here has an item /
                                                           // an UPDATE item for VF k; [29]/[28]: hand-off counters
        if (tid < A.n_vf * CLF_STRIDE) s_clf[tid] = A.clf[tid];
        block_lds_sync();
    }

    // Lane roles. As an MFMA operand lane (16x16x4): n16 = lane & 15 is the tile row (A) / column (B, C, D), g = lane >> 4
    // the k index (A, B) / the row group (C, D: rows 4 g + v). As a table builder: bi = lane & 7 is the item of an
    // 8-item column block, cp = lane >> 3 the second index (c2 / c4; lanes with cp >= 6 idle).
    // Columns of an 8-item block: item j = 4 h + i  (h = 0, 1; i = 0..3) has its real-part column at 8 h + i and its
    // imaginary-part column at 8 h + 4 + i.
    const int n16 = lane & 15, g = lane >> 4;
    const int bi = lane & 7, cp = lane >> 3;
    const int bcol = 8 * (bi >> 2) + (bi & 3);               // builder: real-part column of item bi
    const int ocol_item = 4 * (n16 >> 3) + (n16 & 3);         // operand lane: item of column n16 within the block
    const bool out_lane = (g == 0) && !(n16 & 4);             // lanes that hold an item's finished sums
    float *cdk = s_R + R_TAB + wave * E_TAB_FLOATS, *abq = cdk + 36 * 16;     // this wave's private tables
    const float *ab_lane = abq + n16 * AS + 4 * g;

    // private tables of one 8-item column block: items lst[i0 .. i0 + cnt) (a short block repeats its last item),
    // state sg (0 = s, 1 = s_next): CDk[c34][col], ABq[col][c12] with ABsel = (Re AB | -Im AB)
    auto build_block = [&](const uint16_t *lst, int i0, int cnt, int sg) {
        if (cp < 6) {
            const int it = lst[i0 + min(bi, cnt - 1)];
            float2 ab[6], cd[6];
            item_entries(s_z1 + (it * 2 + sg) * 4, cp, ab, cd);
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                abq[bcol * AS + 6 * c + cp] = ab[c].x; abq[(bcol + 4) * AS + 6 * c + cp] = -ab[c].y;
                cdk[(6 * c + cp) * 16 + bcol] = cd[c].x; cdk[(6 * c + cp) * 16 + bcol + 4] = cd[c].y;
            }
        }
    };

    // A operands come from the staged W_k, per row tile two ds_read_b128 (k-blocks 0..3, 4..7) and one ds_read_b32
    // (k-block 8): the kernel stays below 128 VGPRs, four waves share a SIMD, and while one wave builds tables or
    // folds its accumulators another one keeps the matrix pipe busy
    const f4v *w4 = reinterpret_cast<const f4v *>(s_W) + lane;
    const float *w8 = s_W + W_TAIL + lane;

    // U1: Q_k(s, a_t) of the update items, per action run, 8 items per wave-iteration: the contraction of E on the 3 row
    // tiles that hold action a's rows -> s_qsa[list position]. The column blocks of all runs are dealt to `nw` waves.
    auto run_u1 = [&](int wv, int nw, const int (&rl)[NACT], const int (&ro)[NACT], int base) {
#pragma unroll 1
        for (int a = 0; a < NACT; ++a) {
            const int t0 = (36 * a) >> 4;                        // first of the 3 row tiles holding action a's rows
            const int cnt = rl[a];
            const uint16_t *lst = s_ulist + ro[a];
            for (int cb = ((wv - base) & (nw - 1)); 8 * cb < cnt; cb += nw) {
                build_block(lst, 8 * cb, min(8, cnt - 8 * cb), 0);
                wave_lds_sync();
                float B[9];
#pragma unroll
                for (int kb = 0; kb < 9; ++kb) B[kb] = cdk[(9 * g + kb) * 16 + n16];
                float qs = 0.0f;
#pragma unroll
                for (int tt = 0; tt < 3; ++tt) {
                    const int t = t0 + tt;
                    const f4v a0 = w4[(t * 2) * 64], a1 = w4[(t * 2 + 1) * 64];
                    const float a8 = w8[t * 64];
                    f4v c = (f4v){0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                    for (int kb = 0; kb < 4; ++kb) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[kb], B[kb], c, 0, 0, 0);
#pragma unroll
                    for (int kb = 0; kb < 4; ++kb) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[kb], B[4 + kb], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a8, B[8], c, 0, 0, 0);
                    const int r0 = 16 * t + 4 * g - 36 * a;             // c12 of the lane's first row, if in [0, 36)
                    const bool in = r0 >= 0 && r0 < 36;
                    const f4v ab4 = *reinterpret_cast<const f4v *>(abq + n16 * AS + (in ? r0 : 0));
                    float xq = qs;
#pragma unroll
                    for (int v = 0; v < 4; ++v) xq = fmaf(c[v], ab4[v], xq);
                    qs = in ? xq : qs;
                }
                float qo[1] = {qs};
                item_tree_sum<1>(qo);
                if (out_lane && 8 * cb + ocol_item < cnt) s_qsa[ro[a] + 8 * cb + ocol_item] = qo[0];
                wave_lds_sync();
            }
            base += (cnt + 7) >> 3;
        }
    };
    // W_k -> region W in A-operand order (12 row tiles of the 180 x 36 matrix; entry (tile t, k-block kb, lane (n16, g)) =
    // W[16 t + n16][9 g + kb], rows >= 180 zero; per tile and lane the k-blocks 0..3 and 4..7 form two float4 — one
    // ds_read_b128 feeds four MFMAs — and k-block 8 sits apart), by `nth` threads with index `ht`. One coalesced read per
    // workgroup instead of a 27 KB gather per wave: every workgroup of the chip wants the same 26 KB at the same moment,
    // and the L2 channels holding them were the bottleneck.
    auto stage_w = [&](const float *Wk, int ht, int nth) {
        // one thread per DESTINATION float4 (row tile t, k-block half hk, lane (nn, gg)): its four values W[16 t + nn][9 gg + 4 hk ..
        // + 3] are consecutive in the source row — a 16-byte load at a 4-byte-aligned address — and go out as one linear
        // ds_write_b128. (Round 2 walked the SOURCE float4s and scattered each into four places: ~45 instructions of index
        // arithmetic per float4 and 2.5-way bank conflicts on the stores.)
        struct __attribute__((packed, aligned(4))) F4U { float x, y, z, w; };
        f4v *dst4 = reinterpret_cast<f4v *>(s_W);
        for (int d = ht; d < 12 * 2 * 64; d += nth) {
            const int t2h = d >> 6, ln = d & 63, row = 16 * (t2h >> 1) + (ln & 15), col = 9 * (ln >> 4) + 4 * (t2h & 1);
            f4v v = (f4v){0.0f, 0.0f, 0.0f, 0.0f};
            if (row < NACT * 36) {
                const F4U w = *reinterpret_cast<const F4U *>(Wk + row * 36 + col);
                v = (f4v){w.x, w.y, w.z, w.w};
            }
            dst4[d] = v;
        }
        for (int z = ht; z < 12 * 64; z += nth) {                                 // k-block 8 of every tile
            const int ln = z & 63, row = 16 * (z >> 6) + (ln & 15);
            s_W[W_TAIL + z] = row < NACT * 36 ? Wk[row * 36 + 9 * (ln >> 4) + 8] : 0.0f;
        }
    };
    // counters in LDS for hand-offs between SUBSETS of the workgroup's waves (s_barrier takes all eight): a producer
    // publishes with lds_arrive, a consumer polls with lds_await. Every awaited count is reached by waves that never
    // wait on the waiter, so the polls terminate.
    auto lds_arrive = [&](int *ctr, int amount) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == __builtin_ctzll(__ballot(true))) atomicAdd(ctr, amount);
    };
    auto lds_await = [&](int *ctr, int want) {
        for (int spins = 0; __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < want && spins < (1 << 20); ++spins)
            __builtin_amdgcn_s_sleep(2);                      // (the bound only guards the GPU against a logic error)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    };
    // With learning on, waves 4..7 ("helpers") have nothing to do in phase P (one lane per env on waves 0..1, pairs on 0..3):
    // they stage W_0, take Z_d^1 of the entry states, build the root's update list and run U1 of the root pass under it.
    const bool helpers = MODE == MODE_FUSED && A.learn && A.k_hi >= 0;

    // ------------------------------------------------------------------ phase P
    if (helpers && wave < P_WAVES) __builtin_amdgcn_s_setprio(2);   // phase P is the critical path: its waves issue
                                                                     // ahead of the helper waves (which have slack)
    if (wave < P_WAVES) {                             // one lane per env on P_WAVES full waves; the physics deals (env, edge)
        const int i = tid;                            // pairs to the lanes of the same wave (pinball_step_wave)
        const bool valid = i < nb;
        const int e = (MODE == MODE_FUSED && A.perm && valid) ? A.perm[e0 + i] : e0 + i;
        s_env[i] = e;
        if (MODE == MODE_FUSED) {
            uint32_t u[4] = {0u, 0u, 0u, 0u};
            int a = NACT - 1, ep0 = 0, o = 0, osteps = 0;
            float sx = 0.5f, sy = 0.5f, svx = 0.0f, svy = 0.0f;
            if (valid) {
                // act (SPEC §2, §4.3)
                const uint64_t g = (uint64_t)(A.env_base + e);
                philox4x32_10((uint32_t)g, (uint32_t)(A.t & 0xffffffffu), (uint32_t)(A.t >> 32), 0u,
                              (uint32_t)(A.seed & 0xffffffffu), (uint32_t)(A.seed >> 32), u);
                const bool explore = (float)(u[0] >> 8) * 0x1p-24f < A.epsilon;
                const int a_rand = (int)__umulhi(u[1], 5u);
                int a_greedy = 0;
                float best = A.qcache[e];
#pragma unroll
                for (int a = 1; a < NACT; ++a) {
                    const float q = A.qcache[(size_t)a * N + e];
                    if (q > best) { best = q; a_greedy = a; }
                }
                a = explore ? a_rand : a_greedy;
                SCG_STAMP(16);                                        // P: perm + qcache gathers, Philox, action
                sx = A.x[e]; sy = A.y[e]; svx = A.vx[e]; svy = A.vy[e];
                ep0 = A.ep_steps[e]; o = A.option_id[e]; osteps = A.opt_steps[e];   // early: latency hides under the physics
                s_s[0 * BLOCK_ENVS + i] = sx; s_s[1 * BLOCK_ENVS + i] = sy;
                s_s[2 * BLOCK_ENVS + i] = svx; s_s[3 * BLOCK_ENVS + i] = svy;
                s_a[i] = (uint8_t)a;
            }
            if (helpers) lds_arrive(&s_misc[29], __popcll(__ballot(valid)));      // state and action of these envs are out
            // physics (SPEC §1.3), the whole wave together
            bool goal;
            SCG_STAMP(17);                                        // P: state gathers
            // the envs' own wave settles free flight and lists the (env, candidate edge) pairs of the others in groups of 64;
            // the groups of BOTH env waves are then dealt to waves 0..P_POOL-1 (a wave whose 64 envs sit near walls would
            // otherwise need two or three rounds of 20 sub-steps while its neighbours idle)
            bool par;
            float *xs_mine = s_s + 4 * BLOCK_ENVS + wave * 64;
            const int groups = pinball_wave_prepare_any(s_edges, A.cellmask, A.ms, valid, sx, sy, svx, svy, a, goal, par,
                                                        s_pitems + wave * PITEMS, xs_mine, BLOCK_ENVS);
            if (lane == 0) s_misc[22 + wave] = groups;
            lds_arrive(&s_misc[24], 1);
            SCG_STAMP(18);                                        // P: physics, own part (refinement, free flight, pair lists)
            {
                lds_await(&s_misc[24], P_WAVES);
                int gsum[P_WAVES + 1];
                gsum[0] = 0;
#pragma unroll
                for (int w2 = 0; w2 < P_WAVES; ++w2) gsum[w2 + 1] = gsum[w2] + s_misc[22 + w2];
                for (int q = wave; q < gsum[P_WAVES]; q += P_POOL) {
                    int owner = 0;
#pragma unroll
                    for (int w2 = 1; w2 < P_WAVES; ++w2) owner += q >= gsum[w2] ? 1 : 0;
                    pinball_wave_group(s_edges, A.ms, s_pitems + owner * PITEMS + 64 * (q - (owner ? gsum[1] : 0)),
                                       s_s + 4 * BLOCK_ENVS + owner * 64, BLOCK_ENVS, s_ia + owner * 64);
                }
                lds_arrive(&s_misc[25], 1);
                lds_await(&s_misc[25], P_POOL);
            }
            const float rew = pinball_wave_finish(par, sx, sy, svx, svy, a, goal, xs_mine, BLOCK_ENVS, s_ia + wave * 64);
            SCG_STAMP(2);                                         // P: physics, the pooled pair groups + hand-offs
            if (valid) {
                // bookkeeping (SPEC §1.4)
                const int eps1 = ep0 + 1;
                const bool timeout = !goal && eps1 >= A.max_ep;
                const int dn = goal ? 1 : (timeout ? 2 : 0);
                float nx = sx, ny = sy, nvx = svx, nvy = svy;
                if (dn) {
                    const uint32_t si = __umulhi(u[2], (uint32_t)A.ms.n_starts);
                    nx = A.starts[2 * si]; ny = A.starts[2 * si + 1]; nvx = 0.0f; nvy = 0.0f;
                }
                s_s[4 * BLOCK_ENVS + i] = nx; s_s[5 * BLOCK_ENVS + i] = ny;
                s_s[6 * BLOCK_ENVS + i] = nvx; s_s[7 * BLOCK_ENVS + i] = nvy;
                // options (SPEC §4.2), branch-free: membership bit masks of s' and s_next over all options
                // (classifier rows come from LDS), then the success / failure / selection rules on the masks
                unsigned inA = 0, inB = 0, inS = 0;   // bit k: in_k(s'), in_k(s_next); gestating k only: in_k(s)
                const unsigned known = A.enabled | A.gest;
#pragma unroll
                for (int k = 1; k < MAX_VF; ++k) {
                    if (k < A.n_vf && ((known >> k) & 1u)) {
                        const float *w = s_clf + CLF_STRIDE * k;
                        if (clf_z(w, sx, sy) > 0.0f) inA |= 1u << k;
                        if (clf_z(w, nx, ny) > 0.0f) inB |= 1u << k;
                        if (((A.gest >> k) & 1u) && clf_z(w, s_s[0 * BLOCK_ENVS + i], s_s[1 * BLOCK_ENVS + i]) > 0.0f) inS |= 1u << k;
                    }
                }
                bool keep = false;
                float ro = 0.0f, co = 0.0f;
                if (o >= 1) {
                    const unsigned par = (A.parents >> (3 * (o & 7))) & 7u;       // SPEC §4.2: target of option o
                    const bool succ = (par == 0) ? goal : ((inA >> par) & 1u);
                    const bool fail = !succ && !((inA >> (o & 31)) & 1u);
                    const bool otime = osteps + 1 >= A.max_opt;
                    const bool term = (dn != 0) || succ || fail || otime;
                    ro = rew + (succ ? A.r_succ : 0.0f);
                    co = term ? 0.0f : A.gamma;
                    keep = !term;
                }
                // smallest k with in_k(s_next) and s_next outside k's target region
                unsigned tgtB = 0;                    // bit k: s_next already lies in option k's target region
#pragma unroll
                for (int k = 1; k < MAX_VF; ++k) {
                    const unsigned par = (A.parents >> (3 * k)) & 7u;
                    if (par != 0 && ((inB >> par) & 1u)) tgtB |= 1u << k;
                }
                const unsigned sel = inB & ~tgtB & A.enabled;         // a gestating option is never selected
                const int on = keep ? o : (sel ? __builtin_ctz(sel) : 0);
                s_a[i] = (uint8_t)a; s_ot[i] = (uint8_t)o; s_on[i] = (uint8_t)on;
                s_gs[i] = (uint8_t)inS; s_ia[i] = (uint8_t)((inA & 0x3Eu) | (goal ? 1u : 0u));
                {   // bit masks of the VFs with items / update items here: OR over the wave's lanes first (7 ballots), then
                    // one LDS atomic per wave instead of one per env on a single word
                    const unsigned pm = (1u << (o & 31)) | (1u << (on & 31)) | inS, um = (1u << (o & 31)) | inS;
                    unsigned pw = 0, uw = 0;
#pragma unroll
                    for (int k = 0; k < MAX_VF + 1; ++k) {
                        if (__ballot((pm >> k) & 1u)) pw |= 1u << k;
                        if (__ballot((um >> k) & 1u)) uw |= 1u << k;
                    }
                    if (lane == __builtin_ctzll(__ballot(true))) {
                        atomicOr(reinterpret_cast<unsigned *>(&s_misc[31]), pw);
                        atomicOr(reinterpret_cast<unsigned *>(&s_misc[30]), uw);
                    }
                }
                if (inS && A.gest_succ) {                             // SPEC §4.4: gestation successes (integer counts: order-free)
#pragma unroll
                    for (int k = 1; k < MAX_VF; ++k) {
                        const unsigned par = (A.parents >> (3 * k)) & 7u;
                        if (((inS >> k) & 1u) && ((par == 0) ? goal : (bool)((inA >> par) & 1u))) atomicAdd(&A.gest_succ[k], 1);
                    }
                }
                s_r0[i] = rew; s_c0[i] = dn ? 0.0f : A.gamma; s_ro[i] = ro; s_co[i] = co;
                // results -> staging record at this env's POSITION (full-line stores); commit_row scatters them
                // to the caller's arrays in env order. (Direct 4-byte stores from here dirtied every 64-byte line
                // from ~6 workgroups on different XCDs: ≈ 9 us of partial-line writes per step.)
                {
                    const int osn = keep ? osteps + 1 : 0, epn = dn ? 0 : eps1;
                    float4 *orec = A.outrec + (size_t)(e0 + i) * 4;
                    orec[0] = make_float4(nx, ny, nvx, nvy);
                    orec[1] = make_float4(rew, __uint_as_float((unsigned)a | ((unsigned)dn << 8) | ((unsigned)on << 16)),
                                          __int_as_float(osn), __int_as_float(epn));
                }
                SCG_STAMP(19);                                        // P: bookkeeping, option logic, result line
                if (A.ring_x) {                                       // SPEC §7: trajectory ring + events
                    const size_t row = (size_t)(ep0 & A.ring_mask) * N + e;
                    A.ring_x[row] = s_s[0 * BLOCK_ENVS + i]; A.ring_y[row] = s_s[1 * BLOCK_ENVS + i];
                }
                if (A.events) { A.events[e] = (uint8_t)((goal ? 1u : 0u) | (inA & 0x3Eu)); A.ev_len[e] = eps1; }
                if (A.hist_next) atomicAdd(&A.hist_next[(e >> 8) * 8 + on], 1);   // next step's counting sort
            } else {
                s_a[i] = 0; s_ot[i] = 255; s_on[i] = 255; s_gs[i] = 0; s_ia[i] = 0;
            }
        } else if (valid) {
            if (MODE == MODE_TRANS) {
                s_s[0 * BLOCK_ENVS + i] = A.x[e]; s_s[1 * BLOCK_ENVS + i] = A.y[e];
                s_s[2 * BLOCK_ENVS + i] = A.vx[e]; s_s[3 * BLOCK_ENVS + i] = A.vy[e];
                s_s[4 * BLOCK_ENVS + i] = A.xn[e]; s_s[5 * BLOCK_ENVS + i] = A.yn[e];
                s_s[6 * BLOCK_ENVS + i] = A.vxn[e]; s_s[7 * BLOCK_ENVS + i] = A.vyn[e];
                s_a[i] = A.action[e]; s_ot[i] = (uint8_t)A.k_lo; s_on[i] = 255; s_gs[i] = 0; s_ia[i] = 0;
                const float r = A.reward[e], c = A.cont_in[e];
                s_r0[i] = r; s_c0[i] = c; s_ro[i] = r; s_co[i] = c;
            } else {
                s_s[4 * BLOCK_ENVS + i] = A.x[e]; s_s[5 * BLOCK_ENVS + i] = A.y[e];
                s_s[6 * BLOCK_ENVS + i] = A.vx[e]; s_s[7 * BLOCK_ENVS + i] = A.vy[e];
                s_a[i] = 0; s_ot[i] = 255; s_on[i] = (uint8_t)A.k_lo; s_gs[i] = 0; s_ia[i] = 0;
                s_r0[i] = 0.0f; s_c0[i] = 0.0f; s_ro[i] = 0.0f; s_co[i] = 0.0f;
            }
        } else {
            s_a[i] = 0; s_ot[i] = 255; s_on[i] = 255; s_gs[i] = 0; s_ia[i] = 0;
        }
    } else if (MODE == MODE_FUSED && wave < P_POOL) {   // no envs of its own: takes its share of the physics' pair groups
        lds_await(&s_misc[24], P_WAVES);
        int gsum[P_WAVES + 1];
        gsum[0] = 0;
#pragma unroll
        for (int w2 = 0; w2 < P_WAVES; ++w2) gsum[w2 + 1] = gsum[w2] + s_misc[22 + w2];
        for (int q = wave; q < gsum[P_WAVES]; q += P_POOL) {
            int owner = 0;
#pragma unroll
            for (int w2 = 1; w2 < P_WAVES; ++w2) owner += q >= gsum[w2] ? 1 : 0;
            pinball_wave_group(s_edges, A.ms, s_pitems + owner * PITEMS + 64 * (q - (owner ? gsum[1] : 0)),
                               s_s + 4 * BLOCK_ENVS + owner * 64, BLOCK_ENVS, s_ia + owner * 64);
        }
        lds_arrive(&s_misc[25], 1);
    } else if (helpers && wave >= HELPER0) {
        const int ht = tid - HELPER0 * 64, hw = wave - HELPER0;       // helper thread / wave index (256 threads, 4 waves)
        stage_w(A.W, ht, THREADS - HELPER0 * 64);
        lds_await(&s_misc[29], nb);                                                    // the P waves have published s and a
        if (ht < nb) {                                                                 // Z_d^1 of the entry states
            const float sv2 = fmaf(s_s[2 * BLOCK_ENVS + ht], 0.25f, 0.5f), sv3 = fmaf(s_s[3 * BLOCK_ENVS + ht], 0.25f, 0.5f);
            const float2 za = sincospi_cs(s_s[ht]), zb = sincospi_cs(s_s[BLOCK_ENVS + ht]), zc = sincospi_cs(sv2), zd = sincospi_cs(sv3);
            float4 *dst = reinterpret_cast<float4 *>(s_z1 + (ht * 2 + 0) * 4);
            dst[0] = make_float4(za.x, za.y, zb.x, zb.y);
            dst[1] = make_float4(zc.x, zc.y, zd.x, zd.y);
        }
        // the root's update list (every env, one run per action, block order inside a run): each helper wave derives the
        // run geometry itself from two rounds of ballots; helper wave 0 writes the list
        int rl[NACT], ro[NACT];
        {
            uint64_t m0[NACT], m1[NACT];
            const int at0 = lane < nb ? s_a[lane] : -1, at1 = 64 + lane < nb ? s_a[64 + lane] : -1;
            int off = 0;
#pragma unroll
            for (int a = 0; a < NACT; ++a) {
                m0[a] = __ballot(at0 == a); m1[a] = __ballot(at1 == a);
                rl[a] = __popcll(m0[a]) + __popcll(m1[a]); ro[a] = off; off += rl[a];
            }
            if (hw == 0) {
                const uint64_t below = (1ull << lane) - 1ull;
#pragma unroll
                for (int a = 0; a < NACT; ++a) {
                    if (at0 == a) s_ulist[ro[a] + __popcll(m0[a] & below)] = (uint16_t)lane;
                    if (at1 == a) s_ulist[ro[a] + __popcll(m0[a]) + __popcll(m1[a] & below)] = (uint16_t)(64 + lane);
                }
            }
        }
        lds_arrive(&s_misc[28], 1);
        lds_await(&s_misc[28], WAVES - HELPER0);                               // W_0, Z(s) and the list are complete
        run_u1(hw, WAVES - HELPER0, rl, ro, 0);
#ifdef SCG_STAMPS
        if (ht == 0 && A.stamps) s_stamp[28] += (unsigned)(__builtin_amdgcn_s_memtime() - stamp_prev);   // helper wave 0: start -> done
#endif
    }
    // From here on waves 0..1 run at priority 1 and the others at 0: waves 0 and 1 build every pass's lists (the other
    // six wait for them at the next barrier). (Round 2 had waves 0..3 at 1; 0..1 measured +0.3..0.45 % in round 3.) Measured against no priority:
    // +2.0 % env-steps/s; the same for waves 0..1 only; 0 % for waves 4..7, odd waves or one whole workgroup of the CU.
    if (helpers) { if (wave < P_WAVES) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
    block_lds_sync();

    SCG_STAMP(0);   // phase P
    // ------------------------------------------------------------------ phase Z (SPEC §3): Z_d^1 of s and s_next
    // one sincospi per thread and round: thread -> (env i, state sg, variable d)
    for (int u = tid; u < BLOCK_ENVS * 8; u += THREADS) {
        const int i = u & (BLOCK_ENVS - 1), d = (u / BLOCK_ENVS) & 3, sg = u / (4 * BLOCK_ENVS);
        if (i < nb && (MODE != MODE_QVAL || sg == 1) && !(helpers && sg == 0)) {
            const float sv = s_s[(4 * sg + d) * BLOCK_ENVS + i];
            s_z1[(i * 2 + sg) * 4 + d] = sincospi_cs(d < 2 ? sv : fmaf(sv, 0.25f, 0.5f));
        }
    }

    // ------------------------------------------------------------------ phase TD (SPEC §3.1, §5) on the matrix pipe
    const unsigned present = (MODE == MODE_FUSED) ? (unsigned)__builtin_amdgcn_readfirstlane(s_misc[31]) : ~0u;
    // value functions that only have envs ENTERING them here (evaluation-only: no update item, a handful of items — 1.6
    // such VFs with 3.5 items each per workgroup on the bench workload) skip the pass machinery: see the tail of the kernel
    const unsigned eval_only = (MODE == MODE_FUSED && A.learn) ? (present & ~(unsigned)__builtin_amdgcn_readfirstlane(s_misc[30])) : 0u;
    for (int k = A.k_lo; k <= A.k_hi; ++k) {
        if (!((present >> k) & 1u) || ((eval_only >> k) & 1u)) {   // nobody here runs or enters option k: skip the pass outright
            if (tid == 0 && A.cnts) A.cnts[(size_t)b * A.n_vf + k] = 0;
            continue;
        }
        block_lds_sync();
        SCG_STAMP(k == 0 ? 5 : 12);   // (diagnostic) wait at the pass's first barrier
        // ---- the workgroup's item lists for VF k (SPEC §5), built by ballot + popcount over the 128 envs:
        //   eval list   : envs that need Q_k(s_next, .) (bootstrap target and/or next action), block order
        //   update lists: envs that update VF k, one run per action a_t, block order inside a run
        bool ev = false, up = false;
        int at = -1;
        if (tid < nb) {
            const int ot = s_ot[tid], on = s_on[tid];
            const bool own = (MODE == MODE_FUSED && k == 0) || ot == k;
            const bool gst = MODE == MODE_FUSED && !own && ((s_gs[tid] >> k) & 1);       // SPEC §4.4 off-policy item
            up = (MODE != MODE_QVAL) && A.learn && (own || gst);
            float rk = s_r0[tid], cont = s_c0[tid];
            if (MODE == MODE_FUSED && k != 0) {
                if (ot == k) { rk = s_ro[tid]; cont = s_co[tid]; }
                else {                                    // as if the env ran option k: no time-out, no selection
                    const unsigned ia = s_ia[tid], par = (A.parents >> (3 * k)) & 7u;
                    const bool succ = (par == 0) ? (ia & 1u) : ((ia >> par) & 1u);
                    const bool fail = !succ && !((ia >> k) & 1u);
                    rk = rk + (succ ? A.r_succ : 0.0f);
                    cont = (cont == 0.0f || succ || fail) ? 0.0f : A.gamma;
                }
            }
            s_rk[tid] = rk; s_ck[tid] = cont;
            ev = (on == k) || (up && cont > 0.0f);
            at = s_a[tid];
        }
        uint64_t mb[1 + NACT];
        if (wave < LIST_WAVES) {
            mb[0] = __ballot(ev);
#pragma unroll
            for (int a = 0; a < NACT; ++a) mb[1 + a] = __ballot(up && at == a);
            if (lane < 1 + NACT) {
                const uint64_t mine = lane == 0 ? mb[0] : lane == 1 ? mb[1] : lane == 2 ? mb[2] : lane == 3 ? mb[3]
                                                        : lane == 4 ? mb[4] : mb[5];
                s_misc[wave * 8 + lane] = __popcll(mine);
            }
        }
        SCG_STAMP(23);                                       // (diagnostic) flags + ballots
        const float *Wk = A.W + (MODE == MODE_QVAL ? 0 : (size_t)k * NACT * NF);
        if (!(helpers && k == 0)) stage_w(Wk, tid, THREADS);   // (the helper waves staged W_0 under phase P)
        SCG_STAMP(24);                                       // (diagnostic) W staging
        block_lds_sync();
        SCG_STAMP(25);                                       // (diagnostic) wait at the barrier behind the staging
        int n_ev = 0, run_len[NACT], run_off[NACT];
        {
            int se = 0;
#pragma unroll
            for (int w2 = 0; w2 < LIST_WAVES; ++w2) se += s_misc[w2 * 8];
            n_ev = __builtin_amdgcn_readfirstlane(se);
            int off = 0;
#pragma unroll
            for (int a = 0; a < NACT; ++a) {      // wave-uniform: keep them in SGPRs
                int sr = 0;
#pragma unroll
                for (int w2 = 0; w2 < LIST_WAVES; ++w2) sr += s_misc[w2 * 8 + 1 + a];
                run_len[a] = __builtin_amdgcn_readfirstlane(sr);
                run_off[a] = off;
                off += run_len[a];
            }
        }
        const int nupd = run_off[NACT - 1] + run_len[NACT - 1];
        if (wave < LIST_WAVES) {
            const uint64_t below = (1ull << lane) - 1ull;
            if (ev) {
                int off = 0;
                for (int w2 = 0; w2 < wave; ++w2) off += s_misc[w2 * 8];
                s_elist[off + __popcll(mb[0] & below)] = (uint16_t)tid;
            }
            if (up) {
                int off = 0;
                const uint64_t mine = at == 0 ? mb[1] : at == 1 ? mb[2] : at == 2 ? mb[3] : at == 3 ? mb[4] : mb[5];
                for (int w2 = 0; w2 < wave; ++w2) off += s_misc[w2 * 8 + 1 + at];
                const int ro = at == 0 ? run_off[0] : at == 1 ? run_off[1] : at == 2 ? run_off[2] : at == 3 ? run_off[3] : run_off[4];
                s_ulist[ro + off + __popcll(mine & below)] = (uint16_t)tid;
            }
        }
        block_lds_sync();
        if (tid == 0 && A.cnts) A.cnts[(size_t)b * A.n_vf + k] = nupd;
        SCG_STAMP(k == 0 ? 1 : 8);    // phase Z (first pass only) + list build + W staging
        if (n_ev + nupd == 0) continue;

        // column blocks of this pass: nqe of the eval list, then per action run ceil(run_len / 8) of the update list;
        // block index i goes to wave i & 7
        const int nqe = (n_ev + 7) >> 3;
        if (wave < nqe) {
            // ---- E: Q_k(s_next, .) of the eval list, one 8-item column block per wave-iteration (SPEC §3.1)
            for (int cb = wave; cb < nqe; cb += WAVES) {
                build_block(s_elist, 8 * cb, min(8, n_ev - 8 * cb), 1);
                wave_lds_sync();
                float B[9];
#pragma unroll
                for (int kb = 0; kb < 9; ++kb) B[kb] = cdk[(9 * g + kb) * 16 + n16];
                f4v acc[12];
#pragma unroll
                for (int t = 0; t < 12; ++t) {
                    const f4v a0 = w4[(t * 2) * 64], a1 = w4[(t * 2 + 1) * 64];
                    const float a8 = w8[t * 64];
                    f4v c = (f4v){0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                    for (int kb = 0; kb < 4; ++kb) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[kb], B[kb], c, 0, 0, 0);
#pragma unroll
                    for (int kb = 0; kb < 4; ++kb) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[kb], B[4 + kb], c, 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a8, B[8], c, 0, 0, 0);
                }
                // rows 16 t + 4 g + v -> action rho / 36, c12 = rho % 36; a lane's four rows never straddle actions
                float q[NACT + 1] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int t = 0; t < 12; ++t) {
                    const int Ct = (16 * t) % 36, At = (16 * t) / 36;
                    if (Ct + 12 < 36) {                          // the tile's 16 rows belong to one action
                        const f4v ab4 = *reinterpret_cast<const f4v *>(ab_lane + Ct);
#pragma unroll
                        for (int v = 0; v < 4; ++v) q[At] = fmaf(acc[t][v], ab4[v], q[At]);
                    } else {                                     // row groups g >= (36 - Ct) / 4 belong to the next action
                        const bool wrap = 4 * g >= 36 - Ct;
                        const f4v ab4 = *reinterpret_cast<const f4v *>(ab_lane + (wrap ? Ct - 36 : Ct));
                        float xq = wrap ? q[At + 1] : q[At];
#pragma unroll
                        for (int v = 0; v < 4; ++v) xq = fmaf(acc[t][v], ab4[v], xq);
                        q[At] = wrap ? q[At] : xq;
                        q[At + 1] = wrap ? xq : q[At + 1];
                    }
                }
                float qo[NACT] = {q[0], q[1], q[2], q[3], q[4]};
                item_tree_sum<NACT>(qo);
                if (out_lane && 8 * cb + ocol_item < n_ev) {
                    const int il = s_elist[8 * cb + ocol_item];
                    if (s_on[il] == k) {
                        if (MODE == MODE_FUSED) {         // into the env's result line; commit_row writes qcache
                            float4 *orec = A.outrec + (size_t)(e0 + il) * 4;
                            orec[2] = make_float4(qo[0], qo[1], qo[2], qo[3]);
                            orec[3].x = qo[4];
                        } else {
#pragma unroll
                            for (int a = 0; a < NACT; ++a) gstore(&A.qcache[(size_t)a * N + s_env[il]], qo[a]);
                        }
                    }
                    float mx = qo[0];
#pragma unroll
                    for (int a = 1; a < NACT; ++a) mx = fmaxf(mx, qo[a]);
                    s_maxq[il] = mx;
                }
                wave_lds_sync();
            }
        }
        SCG_STAMP(k == 0 ? 3 : 10);   // E (wave 0's share)
        // ---- U1 (the root's ran under phase P on the helper waves)
        if (MODE != MODE_QVAL && nupd > 0 && !(helpers && k == 0)) run_u1(wave, WAVES, run_len, run_off, nqe);   // dealt on behind E's blocks
        if (MODE == MODE_QVAL || nupd == 0) continue;
        SCG_STAMP(k == 0 ? 4 : 11);   // U1 (wave 0's share)
        block_lds_sync();                                   // s_maxq, s_qsa cross waves; the staging area changes hands
        SCG_STAMP(k == 0 ? 7 : 14);   // wait for the other waves

        // ---- U2: the block partial (SPEC §5). Padded slots: every action run is padded with null items to a multiple of
        // 4, run a occupying slots [off4[a], off4[a] + len4[a]). Chunks of U2_CH = 72 slots (144 K-steps, kap = 2 slot + part):
        //   build  PT[c12][kap] = delta * ABsel, CDT[c34][kap] = CD  (null items: +0); wave w owns slots 9 w .. 9 w + 8
        //   MFMA   G[a] += PT x CDT^T, groups of 4 items: one MFMA over their real parts, one over the imaginary parts;
        //          the 9 output tiles of action a are dealt to the 8 waves (tile q -> wave (q + a) & 7, so wave a holds two),
        //          accumulators stay in registers for the whole pass and go straight to the block's slab
        int off4[NACT + 1];
        off4[0] = 0;
#pragma unroll
        for (int a = 0; a < NACT; ++a) off4[a + 1] = off4[a] + ((run_len[a] + 3) & ~3);
        int wave_u = wave;
        asm volatile("" : "+s"(wave_u));                    // keeps the per-(wave, action) tile geometry inside the pass
        f4v accU[NACT][2];
#pragma unroll
        for (int a = 0; a < NACT; ++a) {
#pragma unroll
            for (int s = 0; s < 2; ++s) accU[a][s] = (f4v){0.0f, 0.0f, 0.0f, 0.0f};
        }
        float *ptab = s_R, *ctab = s_R + 36 * US;
        const int bi9 = lane % 9, cp9 = lane / 9;           // builder lanes of a chunk: (slot 9 wave + bi9, second index cp9 < 6)
        for (int ch0 = 0; ch0 < off4[NACT]; ch0 += U2_CH) {
            if (ch0 > 0) block_lds_sync();                                    // previous chunk's operands consumed
            SCG_STAMP(20);                                                    // (diagnostic) U2: MFMAs of the previous chunk + wait
            {
                const int slot = 9 * wave + bi9, ps = ch0 + slot;
                const int cp = cp9;                                           // (shadows the 8-item builders' cp in this scope)
                if (cp < 6 && ps < off4[NACT]) {
                    int a_ = 0;
#pragma unroll
                    for (int a = 1; a < NACT; ++a) a_ += ps >= off4[a] ? 1 : 0;
                    const int o4 = a_ == 0 ? off4[0] : a_ == 1 ? off4[1] : a_ == 2 ? off4[2] : a_ == 3 ? off4[3] : off4[4];
                    const int rl = a_ == 0 ? run_len[0] : a_ == 1 ? run_len[1] : a_ == 2 ? run_len[2] : a_ == 3 ? run_len[3] : run_len[4];
                    const int ro = a_ == 0 ? run_off[0] : a_ == 1 ? run_off[1] : a_ == 2 ? run_off[2] : a_ == 3 ? run_off[3] : run_off[4];
                    const int j = ps - o4;
                    float *pd = ptab + cp * US + 2 * slot, *cdst = ctab + cp * US + 2 * slot;
                    if (j < rl) {
                        const int li = ro + j, il = s_ulist[li];
                        const float rr = s_rk[il], cont = s_ck[il];
                        const float target = cont > 0.0f ? fmaf(cont, s_maxq[il], rr) : rr;
                        const float d = target - s_qsa[li];
                        float2 ab[6], cd[6];
                        item_entries(s_z1 + (il * 2 + 0) * 4, cp, ab, cd);
#pragma unroll
                        for (int c = 0; c < 6; ++c) {
                            *reinterpret_cast<float2 *>(pd + 6 * c * US) = make_float2(d * ab[c].x, d * (-ab[c].y));
                            *reinterpret_cast<float2 *>(cdst + 6 * c * US) = make_float2(cd[c].x, cd[c].y);
                        }
                    } else {                                     // null item padding a run to a multiple of 4
#pragma unroll
                        for (int c = 0; c < 6; ++c) {
                            *reinterpret_cast<float2 *>(pd + 6 * c * US) = make_float2(0.0f, 0.0f);
                            *reinterpret_cast<float2 *>(cdst + 6 * c * US) = make_float2(0.0f, 0.0f);
                        }
                    }
                }
            }
            SCG_STAMP(21);                                       // (diagnostic) U2: build
            block_lds_sync();                                    // operands visible
            SCG_STAMP(22);                                       // (diagnostic) U2: wait for the other waves' build
            auto run_u2 = [&](auto aa_c) {
                constexpr int AA = decltype(aa_c)::value;
                const int lo = max(off4[AA], ch0), hi = min(off4[AA + 1], ch0 + U2_CH);   // the run's slots in this chunk
                if (lo >= hi) return;
                const bool two = wave_u == AA;                   // this wave's tiles: (wave - AA) & 7, and tile 8 on wave AA
                const float *pa[2], *pb[2];
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const int q = s == 0 ? ((wave_u - AA) & 7) : 8;
                    const int mi = (q * 11) >> 5, ni = q - 3 * mi;             // q / 3, q % 3
                    pa[s] = ptab + min(16 * mi + n16, 35) * US + 2 * g + 2 * (lo - ch0);
                    pb[s] = ctab + min(16 * ni + n16, 35) * US + 2 * g + 2 * (lo - ch0);
                }
                const int ngrp = (hi - lo) >> 2;
                if (two) {
                    for (int gi = 0; gi < ngrp; ++gi) {
                        float2 a2[2], b2[2];
#pragma unroll
                        for (int s = 0; s < 2; ++s) { a2[s] = *reinterpret_cast<const float2 *>(pa[s] + 8 * gi); b2[s] = *reinterpret_cast<const float2 *>(pb[s] + 8 * gi); }
#pragma unroll
                        for (int s = 0; s < 2; ++s) accU[AA][s] = __builtin_amdgcn_mfma_f32_16x16x4f32(b2[s].x, a2[s].x, accU[AA][s], 0, 0, 0);
#pragma unroll
                        for (int s = 0; s < 2; ++s) accU[AA][s] = __builtin_amdgcn_mfma_f32_16x16x4f32(b2[s].y, a2[s].y, accU[AA][s], 0, 0, 0);
                    }
                } else {
                    for (int gi = 0; gi < ngrp; ++gi) {
                        const float2 a2 = *reinterpret_cast<const float2 *>(pa[0] + 8 * gi), b2 = *reinterpret_cast<const float2 *>(pb[0] + 8 * gi);
                        accU[AA][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(b2.x, a2.x, accU[AA][0], 0, 0, 0);
                        accU[AA][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(b2.y, a2.y, accU[AA][0], 0, 0, 0);
                    }
                }
            };
            run_u2(std::integral_constant<int, 0>{}); run_u2(std::integral_constant<int, 1>{});
            run_u2(std::integral_constant<int, 2>{}); run_u2(std::integral_constant<int, 3>{});
            run_u2(std::integral_constant<int, 4>{});
        }
        SCG_STAMP(k == 0 ? 6 : 13);   // U2
        // the block partial P_b,k straight from the accumulators (zeros for an empty run). The tiles were accumulated
        // TRANSPOSED (A operand = CDT rows, B operand = PT rows; fma(a, b, c) = fma(b, a, c)), so register v of lane (n16, g)
        // of tile (mi, ni) is G[a][c12 = 16 mi + n16][c34 = 16 ni + 4 g + v]: four consecutive floats per lane, one 16-byte
        // store instead of four 4-byte ones (the epilogue is store-issue-bound)
        {
            float *slab_lane = A.slabs + ((size_t)b * A.n_vf + k) * NACT * NF + n16 * 36 + 4 * g;
#pragma unroll
            for (int a = 0; a < NACT; ++a) {
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const int q = s == 0 ? ((wave_u - a) & 7) : 8;
                    if (s == 0 || wave_u == a) {
                        const int mi = (q * 11) >> 5, ni = q - 3 * mi;         // wave-uniform
                        const bool okl = (mi < 2 || n16 < 4) && (ni < 2 || g == 0);
                        if (okl) store_wt(slab_lane + a * NF + (16 * mi) * 36 + 16 * ni, accU[a][s]);
                    }
                }
            }
        }
        SCG_STAMP(15);                // slab stores issued
    }
    // ------------------------------------------------------------------ evaluation-only value functions, on the vector pipe
    // Q_k(s_next, .) of the few envs entering an option nobody in this workgroup runs. The (value function, env) pairs are
    // enumerated in a fixed order by every wave and dealt round-robin; a wave evaluates its pair alone, with W_k read
    // straight from global memory (one 36-float row per lane and round) — no staging, no lists, no workgroup barrier. The
    // arithmetic is SPEC §3.1 operation for operation (the MFMA is the same fmaf chain), so the result is bit-identical to
    // what a pass would have produced; a full pass for 3.5 items cost ~16k cycles, 1.6 times per workgroup.
    if (MODE == MODE_FUSED && eval_only && A.k_hi >= 0) {
        block_lds_sync();                                   // region R is free again
        SCG_STAMP(26);
        float2 *t_ab = reinterpret_cast<float2 *>(cdk), *t_cd = t_ab + 36, *t_T = t_cd + 36;      // this wave's table area: 36 + 36 + 180 float2
        int pair = 0;
        for (int k = 1; k < A.n_vf; ++k) {
            if (!((eval_only >> k) & 1u)) continue;
            const float *Wk = A.W + (size_t)k * NACT * NF;
            for (int h = 0; h < BLOCK_ENVS / 64; ++h) {
                const int ii = 64 * h + lane;
                uint64_t m = __ballot(ii < nb && s_on[ii] == k);
                while (m) {
                    const int il = 64 * h + (int)__builtin_ctzll(m);
                    m &= m - 1;
                    if ((pair++ & (WAVES - 1)) != wave) continue;
                    // tables of s_next: lane c < 36 owns AB[c] and CD[c] (c = 6 hi + lo)
                    if (lane < 36) {
                        const float4 *zp = reinterpret_cast<const float4 *>(s_z1 + (il * 2 + 1) * 4);
                        const float4 za = zp[0], zc = zp[1];
                        const int hi = (lane * 43) >> 8, lo = lane - 6 * hi;                  // lane / 6, lane % 6 for lane < 36
                        float2 ab = zpow_sel(make_float2(za.z, za.w), lo), cd = zpow_sel(make_float2(zc.z, zc.w), lo);
#pragma unroll
                        for (int c = 1; c < 6; ++c) {                                         // row hi: hi chained products
                            const float2 abn = cmul(ab, make_float2(za.x, za.y)), cdn = cmul(cd, make_float2(zc.x, zc.y));
                            if (c <= hi) { ab = abn; cd = cdn; }
                        }
                        t_ab[lane] = make_float2(ab.x, -ab.y);
                        t_cd[lane] = cd;
                    }
                    wave_lds_sync();
                    // T[row][re | im], row = 36 a + c12: the fmaf chain over c34 = 9 g + kb (kb outer, g inner)
#pragma unroll 1
                    for (int r0 = 0; r0 < 192; r0 += 64) {
                        const int row = r0 + lane;
                        if (row < 180) {
                            const float4 *wr = reinterpret_cast<const float4 *>(Wk + row * 36);
                            float wv[36];
#pragma unroll
                            for (int q4 = 0; q4 < 9; ++q4) {
                                const float4 w = wr[q4];
                                wv[4 * q4] = w.x; wv[4 * q4 + 1] = w.y; wv[4 * q4 + 2] = w.z; wv[4 * q4 + 3] = w.w;
                            }
                            float tre = 0.0f, tim = 0.0f;
#pragma unroll
                            for (int kb = 0; kb < 9; ++kb) {
#pragma unroll
                                for (int gg = 0; gg < 4; ++gg) {
                                    const float2 cdv = t_cd[9 * gg + kb];
                                    tre = fmaf(wv[9 * gg + kb], cdv.x, tre);
                                    tim = fmaf(wv[9 * gg + kb], cdv.y, tim);
                                }
                            }
                            t_T[row] = make_float2(tre, tim);
                        }
                    }
                    wave_lds_sync();
                    // q[a][g][part]: lane = 8 a + 2 g + part chains over its nine c12 in increasing order, then the tree
                    float qv = 0.0f;
                    {
                        const int a = min(lane >> 3, NACT - 1), gq = (lane >> 1) & 3, part = lane & 1;
                        // rows 36 a + c12 of group gq: c12 = 4 i + v with ((36 a) / 4 + i) % 4 == gq  ->  i = (gq - 9 a) & 3, + 4, + 8
                        const int i0 = (gq - 9 * a) & 3;
#pragma unroll
                        for (int ii3 = 0; ii3 < 3; ++ii3) {
                            const int i = i0 + 4 * ii3;                                  // i = 0..8: quad of rows 4 i .. 4 i + 3
                            if (i < 9) {
#pragma unroll
                                for (int v = 0; v < 4; ++v) {
                                    const float2 tv = t_T[36 * a + 4 * i + v], av = t_ab[4 * i + v];
                                    qv = fmaf(part ? tv.y : tv.x, part ? av.y : av.x, qv);
                                }
                            }
                        }
                    }
                    qv = qv + __shfl_xor(qv, 1, 64);                      // u_g = q_re + q_im
                    qv = qv + __shfl_xor(qv, 2, 64);                      // u_0 + u_1 | u_2 + u_3
                    qv = qv + __shfl_xor(qv, 4, 64);                      // (u_0 + u_1) + (u_2 + u_3)
                    if (lane < 8 * NACT && (lane & 7) == 0)               // into the env's result line (orec[2].xyzw, orec[3].x)
                        reinterpret_cast<float *>(A.outrec + (size_t)(e0 + il) * 4)[8 + (lane >> 3)] = qv;
                    wave_lds_sync();
                }
            }
        }
        SCG_STAMP(27);
    }
#ifdef SCG_STAMPS
    if (MODE == MODE_FUSED && A.stamps) {
        __syncthreads();
        if (tid < 32) A.stamps[(size_t)blockIdx.x * 32 + tid] += s_stamp[tid];
    }
#endif
}

// ------------------------------------------------------------------------------------------------
// slabs -> G (SPEC §5 two-level block order), n_k, optional apply; the next step's env order rides along
struct ReduceArgs {
    const float *slabs;
    const int32_t *cnts;
    float *G;
    int32_t *n_k;
    float *nk_f;             // packed operand: the counts again, as floats right after G (null = off)
    float *W;
    const float *scale;
    int32_t nblk, n_vf;
    float alpha;
    uint32_t apply;
    // next step's env order (SPEC §5) as extra workgroups (option_id null = off): the fused kernel has counted
    // the new option ids per row of 256 envs into `hist`; `hist_zero` is the other buffer, cleared for the next step
    const int32_t *option_id;
    int32_t *hist, *hist_zero, *perm;
    int32_t n, nrow;
    // commit of the fused kernel's per-position results to the caller's arrays (outrec null = off), one row of
    // 256 envs per wave; with `sort` the same wave then places its row in the next env order
    const float4 *outrec;
    int32_t *invperm;              // [n] position of env e in the current order (in: this step's, out: the next's)
    float *x, *y, *vx, *vy, *reward;
    int32_t *option_id_out, *opt_steps, *ep_steps;
    uint8_t *action, *done;
    float *qcache;                 // [5][n], null = the step ran no TD pass (diagnostic): leave it alone
    int32_t sort;
    // an announced example trigger (scg_arm_collect; c_rows null = none): the commit rows leave what collect_count_kernel would
    const uint8_t *c_events, *c_prev;
    const int32_t *c_evlen, *c_count;
    int32_t *c_rows;
    uint32_t c_bits;
    int32_t c_L, c_ring_len;
};

constexpr int SEG = 16;            // SPEC §5: blocks per first-level segment
constexpr int RED_WAVES = 16;      // one wave per segment, 16 segments per round
constexpr int RED_THREADS = 64 * RED_WAVES;
constexpr int RED_SPW = 2;         // segments per wave and round
constexpr int RED_COLS = NACT * NF / 4;                              // float4 columns per value function
constexpr int RED_NCOL = (RED_COLS + 63) / 64;

__device__ __forceinline__ int sort_key(const int32_t *option_id, int e, int n, int n_vf) {
    int o = e < n ? option_id[e] : -1;
    if (o < 0 || o >= n_vf) o = e < n ? n_vf : -1;
    return o;
}

// SPEC §5 env order from the key totals (all lanes compute the same few integers). Runs of the keys 1..6 follow one
// another in key order; envs of key 0 (running no option) are the filler:
//  * chunked layout (the normal case): every workgroup gets at most c envs of one option's run at its start and
//    key-0 envs behind them, with c = ceil(S / (full workgroups - non-empty runs)) — so the option work is spread
//    evenly over ALL workgroups instead of leaving the key-0 workgroups idle after their root pass (one workgroup
//    per CU: the launch lasts as long as its slowest), and no workgroup ever holds two options' runs (a third
//    pass). What is left of key 0 comes last.
//  * padded layout (when the option runs alone need more workgroups than there are full ones): runs back to back,
//    each padded with key-0 envs to the next workgroup boundary while any are left.
// floor(r / d) for r < 2^24, d <= 2^30, with m = ceil(2^32 / d) (m wraps to 0 for d = 1)
__device__ __forceinline__ uint32_t div_magic(uint32_t d) { return 0xFFFFFFFFu / d + 1u; }
__device__ __forceinline__ int div_by(int r, int d, uint32_t m) { return d == 1 ? r : (int)__umulhi((uint32_t)r, m); }
__device__ __forceinline__ int collect_v(int e, int n, const uint8_t *events, const uint8_t *prev_in, uint32_t bits,
                                         const int32_t *ev_len, int ring_len, int L, bool &in_out);
struct OrderLayout {
    int chunked, c, g, U, Ftot;
    uint32_t mc, mg;               // ceil(2^32 / c), ceil(2^32 / g): exact division of ranks (< 2^24) by mul-high
    int start[7];                  // position of run k's first env
    int cnt[7], n[7], F[7];        // chunked: workgroups of run k, its size, key-0 fill slots before it
    int pad_lo[7], pad_n[7], pad_pos[7], tail_lo, tail_pos;     // padded layout
};
__device__ __forceinline__ void order_layout(const int tot[7], int n_envs, OrderLayout &L) {
    int S = 0, Rn = 0;
#pragma unroll
    for (int k = 1; k < 7; ++k) { S += tot[k]; Rn += tot[k] > 0 ? 1 : 0; L.n[k] = tot[k]; }
    L.n[0] = tot[0];
    const int Bf = n_envs / BLOCK_ENVS;
    int c = BLOCK_ENVS;
    if (Bf > Rn && S > 0) c = min(BLOCK_ENVS, (S + (Bf - Rn) - 1) / (Bf - Rn));
    int U = 0, F = 0;
    L.start[0] = 0; L.cnt[0] = 0; L.F[0] = 0;
    L.mc = div_magic((uint32_t)c);
    L.mg = div_magic((uint32_t)max(BLOCK_ENVS - c, 1));
#pragma unroll
    for (int k = 1; k < 7; ++k) {
        L.cnt[k] = div_by(tot[k] + c - 1, c, L.mc);
        L.start[k] = U * BLOCK_ENVS; L.F[k] = F;
        U += L.cnt[k]; F += L.cnt[k] * BLOCK_ENVS - tot[k];
    }
    L.c = c; L.g = BLOCK_ENVS - c; L.U = U; L.Ftot = F;
    L.chunked = (U * BLOCK_ENVS <= n_envs) ? 1 : 0;
    if (!L.chunked) {
        int P = 0, used = 0;
        L.c = 1 << 30;                                  // one "chunk" per run: pos = start + rank
        L.mc = div_magic(1u << 30);
        L.pad_lo[0] = 0; L.pad_n[0] = 0; L.pad_pos[0] = 0;
#pragma unroll
        for (int k = 1; k < 7; ++k) {
            L.start[k] = P; P += tot[k];
            const int need = tot[k] > 0 ? (BLOCK_ENVS - P % BLOCK_ENVS) % BLOCK_ENVS : 0;
            const int pad = min(need, tot[0] - used);
            L.pad_lo[k] = used; L.pad_n[k] = pad; L.pad_pos[k] = P; used += pad; P += pad;
        }
        L.tail_lo = used; L.tail_pos = P;
    }
}
// position of the r-th env of run k (k >= 1; `start` = L.start[k] selected by the caller)
__device__ __forceinline__ int order_posk(const OrderLayout &L, int start, int r) {
    const int t = div_by(r, L.c, L.mc);
    return start + BLOCK_ENVS * t + (r - t * L.c);
}
__device__ __forceinline__ int order_pos0(const OrderLayout &L, int r) {       // position of the r-th key-0 env
    if (!L.chunked) {
        int pos = L.tail_pos + (r - L.tail_lo);
#pragma unroll
        for (int k = 1; k < 7; ++k)
            if (r >= L.pad_lo[k] && r < L.pad_lo[k] + L.pad_n[k]) pos = L.pad_pos[k] + (r - L.pad_lo[k]);
        return pos;
    }
    int pos = L.U * BLOCK_ENVS + (r - L.Ftot);          // behind all runs
    int st = 0, cn = 0, nk = 0, f0 = 0;
    bool in_run = false;
#pragma unroll
    for (int k = 1; k < 7; ++k) {
        const int fills = L.cnt[k] * BLOCK_ENVS - L.n[k];
        if (r >= L.F[k] && r < L.F[k] + fills) { in_run = true; st = L.start[k]; cn = L.cnt[k]; nk = L.n[k]; f0 = L.F[k]; }
    }
    if (in_run) {
        const int rp = r - f0, nfull = cn - 1;
        if (L.g > 0 && rp < nfull * L.g) {
            const int t = div_by(rp, L.g, L.mg);
            pos = st + BLOCK_ENVS * t + L.c + (rp - t * L.g);
        } else {
            pos = st + BLOCK_ENVS * nfull + (nk - nfull * L.c) + (rp - nfull * L.g);
        }
    }
    return pos;
}

// Four waves per row of 256 envs (wave wv owns envs 64 wv .. 64 wv + 63 of the row), two dependent memory round trips
// and one workgroup barrier in all:
//   commit: gather each env's result line from its position in the current order (one 64-byte read) and write
//           the caller's SoA arrays (state, outputs, qcache) with full-line stores;
//   sort  : place the row in the stable counting-sort order of the next step (7 keys), from the per-row key
//           counts of all rows: offset(key k, row) = (envs with a smaller key) + (key-k envs of earlier rows)
//           (+ key-k envs of the row's earlier waves, exchanged through LDS together with the waves' shares of the
//           count table).
// Every thread of the workgroup must call this (it holds the barrier); waves >= 4 only pass through it.
// Load order matters: position first, then the count table, then the record, so that the table's latency hides
// under the record's and the prefix sums run while the record is in flight. One wave per row (four envs per lane)
// took 8.3 us of dependent work after the launch floor; see DESIGN §10.
__device__ __forceinline__ void commit_and_place_row(const ReduceArgs &R, int row, int wv, int lane, int (*s_x)[24]) {
    const bool act = wv < 4 && row < R.nrow;
    const int e = row * 256 + wv * 64 + lane;
    const bool ok = act && e < R.n;
    const int pos_old = ok ? R.invperm[e] : 0;
    int tot[7], pre[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) { tot[k] = 0; pre[k] = 0; }
    if (act && R.sort) {
        for (int r = wv * 64 + lane; r < R.nrow; r += 256) {       // this wave's quarter of the count table
#pragma unroll
            for (int k = 0; k < 7; ++k) {
                const int h = R.hist[r * 8 + k];
                tot[k] += h;
                if (r < row) pre[k] += h;
            }
        }
    }
    float4 ra = make_float4(0.0f, 0.0f, 0.0f, 0.0f), rb = ra, rq = ra;
    float q4 = 0.0f;
    if (ok) {
        const float4 *r = R.outrec + (size_t)pos_old * 4;
        ra = r[0]; rb = r[1]; rq = r[2]; q4 = r[3].x;
    }
    int key = -1;
    uint64_t km[7];
    if (act) {
        if (R.sort) {
#pragma unroll
            for (int k = 0; k < 7; ++k) {                   // integer sums: any order
#pragma unroll
                for (int m = 1; m < 64; m <<= 1) { tot[k] += __shfl_xor(tot[k], m, 64); pre[k] += __shfl_xor(pre[k], m, 64); }
            }
        }
        const unsigned bits = __float_as_uint(rb.y);
        if (ok) {
            key = (int)((bits >> 16) & 255u);
            R.x[e] = ra.x; R.y[e] = ra.y; R.vx[e] = ra.z; R.vy[e] = ra.w;
            R.reward[e] = rb.x; R.action[e] = (uint8_t)(bits & 255u); R.done[e] = (uint8_t)((bits >> 8) & 255u);
            R.option_id_out[e] = key; R.opt_steps[e] = __float_as_int(rb.z); R.ep_steps[e] = __float_as_int(rb.w);
            if (R.qcache) {
                const size_t n = (size_t)R.n;
                R.qcache[e] = rq.x; R.qcache[n + e] = rq.y; R.qcache[2 * n + e] = rq.z;
                R.qcache[3 * n + e] = rq.w; R.qcache[4 * n + e] = q4;
            }
        }
        if (R.sort) {
#pragma unroll
            for (int k = 0; k < 7; ++k) km[k] = __ballot(key == k);
            if (lane < 21) {                                // [0..6] table totals, [7..13] rows before this one, [14..20] this wave's keys
                int v = 0;
#pragma unroll
                for (int k = 0; k < 7; ++k) {
                    if (lane == k) v = tot[k];
                    if (lane == 7 + k) v = pre[k];
                    if (lane == 14 + k) v = __popcll(km[k]);
                }
                s_x[wv][lane] = v;
            }
        }
    }
    if (R.c_rows && act) {                                  // the announced trigger's examples of this wave's envs (SPEC §7)
        bool in;
        int v = collect_v(e, R.n, R.c_events, R.c_prev, R.c_bits, R.c_evlen, R.c_ring_len, R.c_L, in);
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m, 64);
        if (lane == 0) s_x[wv][21] = v;
    }
    if (!R.sort && !R.c_rows) return;                       // workgroup-uniform
    __syncthreads();
    if (R.c_rows && wv == 0 && lane == 0 && row < R.nrow) {
        R.c_rows[row] = s_x[0][21] + s_x[1][21] + s_x[2][21] + s_x[3][21];
        if (row == 0) R.c_rows[R.nrow] = *R.c_count;        // the buffer's fill level
    }
    if (!R.sort || !act) return;
    int off[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        tot[k] = s_x[0][k] + s_x[1][k] + s_x[2][k] + s_x[3][k];
        off[k] = s_x[0][7 + k] + s_x[1][7 + k] + s_x[2][7 + k] + s_x[3][7 + k];
#pragma unroll
        for (int w = 0; w < 3; ++w) off[k] += w < wv ? s_x[w][14 + k] : 0;      // rank of this wave's first key-k env within its run
    }
    OrderLayout L;
    order_layout(tot, R.n, L);
    int rk = -1, st = 0;
#pragma unroll
    for (int k = 0; k < 7; ++k)
        if (key == k) { rk = off[k] + __popcll(km[k] & ((1ull << lane) - 1ull)); st = L.start[k]; }
    if (rk >= 0) {
        const int pos = key == 0 ? order_pos0(L, rk) : order_posk(L, st, rk);
        R.perm[pos] = e; R.invperm[e] = pos;
    }
    if (wv == 0 && lane < 8) R.hist_zero[row * 8 + lane] = 0;
}

// grid (column chunks, n_vf [+ rows of the env order]). A workgroup owns 64 float4 columns of one value function;
// its 16 waves each sum one segment's slabs, T_s = ((P_16s + P_16s+1) + ...) over the non-empty blocks with all
// 16 loads in flight, park T_s in LDS, and wave 0 adds the non-empty segments in order, G = ((T_0 + T_1) + ...)
// — SPEC §5's two levels in one launch.
__global__ __launch_bounds__(RED_THREADS) void reduce_kernel(const ReduceArgs R) {
    __shared__ float4 s_T[RED_WAVES * RED_SPW][64];
    __shared__ int s_cnt[RED_WAVES * RED_SPW];
    __shared__ int s_x[4][24];         // the commit rows' exchange area
    // trailing workgroups (blockIdx.y >= n_vf): one env row each — commit + next order
    const int k = (int)blockIdx.y < R.n_vf ? (int)blockIdx.y : -1;
    const int rowy = (int)blockIdx.y - R.n_vf;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (k < 0) {
        const int row = rowy * (int)gridDim.x + blockIdx.x;                 // one row per workgroup (waves 0..3): a row
        commit_and_place_row(R, row, wave, lane, s_x);                      // moves ~25 KB, so spread them over the CUs
        return;
    }
    const int i4 = blockIdx.x * 64 + lane;
    const bool live = i4 < RED_COLS;
    // slab addresses = wave-uniform base (block, value function: SGPRs) + this lane's column offset (one VGPR): sixteen
    // 64-bit per-lane pointers would not fit beside the sixteen float4 in flight (the kernel ran at the 128-VGPR cap
    // with 8 spilled registers and a vmcnt(0) in front of the first slab load)
    const size_t slab_stride = (size_t)R.n_vf * RED_COLS * sizeof(float4);
    const char *slab_k = reinterpret_cast<const char *>(R.slabs) + (size_t)k * RED_COLS * sizeof(float4);
    const unsigned col_off = (unsigned)(live ? i4 : 0) * (unsigned)sizeof(float4);
    const int nseg = (R.nblk + SEG - 1) / SEG;
    float4 S = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    int nk = 0;
    // wave 0 applies the update at the end: fetch its W and scale columns now, under the slab loads
    float4 *wp = reinterpret_cast<float4 *>(R.W) + (size_t)k * RED_COLS + (live ? i4 : 0);
    float4 w_old = make_float4(0.0f, 0.0f, 0.0f, 0.0f), sc = w_old;
    if (wave == 0 && R.apply) {
        w_old = *wp;
        sc = *reinterpret_cast<const float4 *>(R.scale + ((live ? i4 : 0) * 4) % NF);     // NF % 4 == 0: no row straddling
    }
    // A round = RED_SPW segments per wave (32 segments = 512 blocks in all at RED_SPW = 2: the bench size in ONE round): the
    // counts of all of a wave's segments are read first, then segment after segment its <= 16 slabs with all loads in flight,
    // and one barrier pair per round (round 2: a round was one segment per wave — two dependent count -> slab round trips and
    // two barrier pairs at the bench size)
    for (int sg0 = 0; sg0 < nseg; sg0 += RED_WAVES * RED_SPW) {
        int cs[RED_SPW];
#pragma unroll
        for (int j = 0; j < RED_SPW; ++j) {
            const int bl = (sg0 + j * RED_WAVES + wave) * SEG + lane;
            cs[j] = (lane < SEG && bl < R.nblk) ? R.cnts[(size_t)bl * R.n_vf + k] : 0;
        }
#pragma unroll
        for (int j = 0; j < RED_SPW; ++j) {
            const int b0 = (sg0 + j * RED_WAVES + wave) * SEG;
            int c = cs[j];
            const unsigned mask = (unsigned)__ballot(c > 0);     // wave-uniform: which of the segment's blocks hold a slab
#pragma unroll
            for (int m = 1; m < SEG; m <<= 1) c += __shfl_xor(c, m, 64);
            float4 T = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (mask) {
                // buffer loads: descriptor = the segment's first slab of this value function (SGPRs), scalar offset = slab u,
                // vector offset = the lane's column
                const __amdgpu_buffer_rsrc_t seg = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<char *>(slab_k + (size_t)b0 * slab_stride), 0, 0x7fffffff, 0x00020000);
                u4v v[SEG];
#pragma unroll
                for (int u = 0; u < SEG; ++u) {
                    v[u] = (u4v){0u, 0u, 0u, 0u};
                    if ((mask >> u) & 1u) v[u] = __builtin_amdgcn_raw_buffer_load_b128(seg, (int)col_off, (int)(u * (unsigned)slab_stride), 0);
                }
#pragma unroll
                for (int u = 0; u < SEG; ++u) {
                    if ((mask >> u) & 1u) {
                        T.x = T.x + __uint_as_float(v[u][0]); T.y = T.y + __uint_as_float(v[u][1]);
                        T.z = T.z + __uint_as_float(v[u][2]); T.w = T.w + __uint_as_float(v[u][3]);
                    }
                }
            }
            s_T[j * RED_WAVES + wave][lane] = T;
            if (lane == 0) s_cnt[j * RED_WAVES + wave] = c;
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int u = 0; u < RED_WAVES * RED_SPW; ++u) {      // segment order sg0 + u (SPEC §5)
                const int cu = s_cnt[u];
                if (cu > 0) {
                    const float4 t = s_T[u][lane];
                    S.x = S.x + t.x; S.y = S.y + t.y; S.z = S.z + t.z; S.w = S.w + t.w;
                    nk += cu;
                }
            }
        }
        __syncthreads();
    }
    if (wave != 0) return;
    if (blockIdx.x == 0 && lane == 0) {
        R.n_k[k] = nk;
        if (R.nk_f) R.nk_f[k] = (float)nk;               // exact: counts stay far below 2^24
    }
    if (!live) return;
    reinterpret_cast<float4 *>(R.G)[(size_t)k * RED_COLS + i4] = S;
    if (R.apply && nk > 0) {
        const float step = R.alpha / (float)nk;
        float4 w = w_old;
        w.x = fmaf(step * sc.x, S.x, w.x); w.y = fmaf(step * sc.y, S.y, w.y);
        w.z = fmaf(step * sc.z, S.z, w.z); w.w = fmaf(step * sc.w, S.w, w.w);
        *wp = w;
    }
}

// acting-only steps have no reduce launch: the commit alone, one workgroup of four waves per row of 256 envs
__global__ __launch_bounds__(256) void commit_kernel(const ReduceArgs R) {
    __shared__ int s_x[4][24];
    commit_and_place_row(R, blockIdx.x, threadIdx.x >> 6, threadIdx.x & 63, s_x);
}

__global__ __launch_bounds__(256) void apply_kernel(float *W, const float *G, const int32_t *n_k, const float *nk_f,
                                                    const float *scale, float alpha) {
    const int k = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= NACT * NF) return;
    const int nk = n_k ? n_k[k] : (int)(nk_f[k] + 0.5f);     // packed operand: counts summed as floats (exact)
    if (nk <= 0) return;
    const float step = alpha / (float)nk;
    const int f = i % NF;
    float *w = W + (size_t)k * NACT * NF + i;
    *w = fmaf(step * scale[f], G[(size_t)k * NACT * NF + i], *w);
}

// ------------------------------------------------------------------------------------------------
// SPEC §5 env order: stable counting sort of the envs by option_id (6 keys), two tiny kernels per step.
// Option-homogeneous workgroups turn five sparse option passes per workgroup into about one dense one.
__global__ __launch_bounds__(256) void sort_hist_kernel(const int32_t *option_id, int n, int n_vf, int32_t *hist) {
    __shared__ int s_c[4][8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int e = blockIdx.x * 256 + tid;
    int o = e < n ? option_id[e] : -1;
    if (o < 0 || o >= n_vf) o = e < n ? n_vf : -1;          // out-of-range ids sort last (key n_vf <= 6)
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        const uint64_t m = __ballot(o == k);
        if (lane == 0) s_c[wave][k] = __popcll(m);
    }
    __syncthreads();
    if (tid < 7) hist[blockIdx.x * 8 + tid] = s_c[0][tid] + s_c[1][tid] + s_c[2][tid] + s_c[3][tid];
}

__global__ __launch_bounds__(256) void sort_scatter_kernel(const int32_t *option_id, int n, int n_vf, int nblk,
                                                           const int32_t *hist, int32_t *perm, int32_t *invperm) {
    __shared__ int s_c[4][8];
    __shared__ int s_off[8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.x;
    // offset of (key k, block b) in the sorted order = (all envs with a smaller key) + (key-k envs of earlier blocks)
    __shared__ int s_tot[8], s_pre[8], s_part[4][16];
    int tot[7], pre[7];
#pragma unroll
    for (int kk = 0; kk < 7; ++kk) { tot[kk] = 0; pre[kk] = 0; }
    for (int bb0 = 0; bb0 < nblk; bb0 += 256) {
        const int bb = bb0 + tid;
        if (bb < nblk) {
#pragma unroll
            for (int kk = 0; kk < 7; ++kk) {
                const int h = hist[bb * 8 + kk];
                tot[kk] += h;
                if (bb < b) pre[kk] += h;
            }
        }
    }
#pragma unroll
    for (int kk = 0; kk < 7; ++kk) {                       // integer sums: any order
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) { tot[kk] += __shfl_xor(tot[kk], m, 64); pre[kk] += __shfl_xor(pre[kk], m, 64); }
        if (lane == 0) { s_part[wave][kk] = tot[kk]; s_part[wave][8 + kk] = pre[kk]; }
    }
    __syncthreads();
    if (tid < 7) {
        s_tot[tid] = s_part[0][tid] + s_part[1][tid] + s_part[2][tid] + s_part[3][tid];
        s_pre[tid] = s_part[0][8 + tid] + s_part[1][8 + tid] + s_part[2][8 + tid] + s_part[3][8 + tid];
    }
    __syncthreads();
    int tt[7];
#pragma unroll
    for (int kk = 0; kk < 7; ++kk) tt[kk] = s_tot[kk];
    OrderLayout L;
    order_layout(tt, n, L);
    if (tid < 7) s_off[tid] = s_pre[tid];               // rank of the row's first key-k env within its run
    const int e = b * 256 + tid;
    int o = e < n ? option_id[e] : -1;
    if (o < 0 || o >= n_vf) o = e < n ? n_vf : -1;
    int rank = 0;
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        const uint64_t m = __ballot(o == k);
        if (lane == 0) s_c[wave][k] = __popcll(m);
        if (o == k) rank = __popcll(m & ((1ull << lane) - 1ull));
    }
    __syncthreads();
    if (o >= 0) {
        int rk = s_off[o] + rank;
        for (int w = 0; w < wave; ++w) rk += s_c[w][o];
        int st = 0;
#pragma unroll
        for (int k = 1; k < 7; ++k) if (o == k) st = L.start[k];
        const int pos = o == 0 ? order_pos0(L, rk) : order_posk(L, st, rk);
        perm[pos] = e;
        invperm[e] = pos;
    }
}

// ------------------------------------------------------------------------------------------------
// SPEC §7: examples for an initiation-set fit, gathered from the trajectory ring (one thread per example)
__global__ __launch_bounds__(256) void harvest_kernel(int n_sel, const int32_t *sel_env, const float *ring_x,
                                                      const float *ring_y, int ring_len, int n, const int32_t *ev_len,
                                                      int l_pos, int l_neg, float *out_xy, uint8_t *out_label) {
    const int L = l_pos + l_neg;
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long long)n_sel * L) return;
    const int si = (int)(t / L), j = (int)(t - (long long)si * L);       // j = age: 0 = most recent recorded state
    const int e = sel_env[si];
    const int idx = ev_len[e] - 1 - j;
    const bool ok = idx >= 0 && j < ring_len;
    float x = 0.0f, y = 0.0f;
    if (ok) {
        const size_t row = (size_t)(idx & (ring_len - 1)) * n + e;
        x = ring_x[row]; y = ring_y[row];
    }
    out_xy[2 * t] = x; out_xy[2 * t + 1] = y;
    out_label[t] = ok ? (j < l_pos ? 1 : 0) : 255;
}

// SPEC §7 device-side trigger + harvest (no host round trip per step), two small launches over rows of COL_ROW envs — or one,
// when the trigger was announced with scg_arm_collect: the commit rows of the step's own last launch then leave the row totals.
// An env is selected when (events & bits) != 0 — with `prev_in` given, only on the step it ENTERS that state (prev_in is
// updated). A selected env contributes its v = min(L, ev_len, ring_len) most recent ring states (age j < l_pos: label 1,
// else 0), appended behind the *count examples the buffer already holds, in env order, ages ascending; what does not fit
// into `cap` is dropped.
//   collect_count_kernel    row totals of v (integer sums: order-free) -> rowsum[row]; the buffer's fill level -> rowsum[nrows]
//   collect_scatter_kernel  offset of a row = fill level + totals of the rows before it; inside a row ballots + popcounts
//                           per wave and a 16-entry scan across the waves; a selected env's examples are gathered by the
//                           lanes of its wave together (lane j = age j), not one after another by the env's own lane
// Deterministic: every position is a prefix sum of integers in env order. (Round 2 walked the envs with ONE workgroup,
// 1024 at a time behind three barriers each: 64 dependent memory round trips per step-batch at the bench size.)
constexpr int COL_ROW = 256;                   // = the env rows of the commit workgroups, which can stand in for collect_count_kernel

__device__ __forceinline__ int collect_v(int e, int n, const uint8_t *events, const uint8_t *prev_in, uint32_t bits,
                                         const int32_t *ev_len, int ring_len, int L, bool &in_out) {
    in_out = false;
    if (e >= n) return 0;
    const bool in = (events[e] & bits) != 0;
    in_out = in;
    const bool hit = prev_in ? (in && !prev_in[e]) : in;
    return hit ? min(min(L, ev_len[e]), ring_len) : 0;
}

__global__ __launch_bounds__(COL_ROW) void collect_count_kernel(int n, const uint8_t *events, const uint8_t *prev_in,
                                                                uint32_t bits, const int32_t *ev_len, int ring_len, int L,
                                                                int32_t *rowsum, int nrows, const int32_t *count) {
    __shared__ int s_w[COL_ROW / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    bool in;
    int v = collect_v(blockIdx.x * COL_ROW + tid, n, events, prev_in, bits, ev_len, ring_len, L, in);
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m, 64);
    if (lane == 0) s_w[wave] = v;
    __syncthreads();
    if (tid == 0) {
        int t = 0;
#pragma unroll
        for (int w2 = 0; w2 < COL_ROW / 64; ++w2) t += s_w[w2];
        rowsum[blockIdx.x] = t;
        if (blockIdx.x == 0) rowsum[nrows] = *count;
    }
}

__global__ __launch_bounds__(COL_ROW) void collect_scatter_kernel(int n, const uint8_t *events, uint8_t *prev_in, uint32_t bits,
                                                                  const float *ring_x, const float *ring_y, int ring_len,
                                                                  const int32_t *ev_len, int l_pos, int l_neg, float *ex_xy,
                                                                  uint8_t *ex_label, int32_t *count, int cap,
                                                                  const int32_t *rowsum, int nrows) {
    __shared__ int s_w[COL_ROW / 64], s_pre[COL_ROW / 64], s_tot[COL_ROW / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, row = blockIdx.x;
    const int L = l_pos + l_neg;
    const int e = row * COL_ROW + tid;
    bool in;
    const int v = collect_v(e, n, events, prev_in, bits, ev_len, ring_len, L, in);
    const int evl = v > 0 ? ev_len[e] : 0;
    if (prev_in && e < n) prev_in[e] = in ? 1 : 0;        // only this thread reads or writes this byte in this launch
    // totals of the rows before this one (and of all rows, for the new fill level)
    int before = 0, all = 0;
    for (int r = tid; r < nrows; r += COL_ROW) { const int t = rowsum[r]; all += t; if (r < row) before += t; }
    int incl = v;                                          // inclusive prefix of v inside the wave
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
        const int t = __shfl_up(incl, m, 64);
        if (lane >= m) incl += t;
    }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) { before += __shfl_xor(before, m, 64); all += __shfl_xor(all, m, 64); }
    if (lane == 63) s_w[wave] = incl;
    if (lane == 0) { s_pre[wave] = before; s_tot[wave] = all; }
    __syncthreads();
    int base = rowsum[nrows], woff = 0, total = 0;
#pragma unroll
    for (int w2 = 0; w2 < COL_ROW / 64; ++w2) {
        base += s_pre[w2]; total += s_tot[w2];
        if (w2 < wave) woff += s_w[w2];
    }
    if (row == 0 && tid == 0) *count = min(rowsum[nrows] + total, cap);
    const int pos0 = base + woff + incl - v;
    // the wave's selected envs one after another, the examples of one env on as many lanes
    uint64_t hits = __ballot(v > 0);
    while (hits) {
        const int src = (int)__builtin_ctzll(hits);
        hits &= hits - 1;
        const int he = __shfl(e, src, 64), hv = __shfl(v, src, 64), hp = __shfl(pos0, src, 64), hl = __shfl(evl, src, 64);
        for (int j = lane; j < hv; j += 64) {
            const int pos = hp + j;
            if (pos < cap) {
                const size_t rrow = (size_t)((hl - 1 - j) & (ring_len - 1)) * n + he;
                ex_xy[2 * (size_t)pos] = ring_x[rrow]; ex_xy[2 * (size_t)pos + 1] = ring_y[rrow];
                ex_label[pos] = j < l_pos ? 1 : 0;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// un-fused kernels
__global__ __launch_bounds__(256) void pinball_kernel(int n, float *x, float *y, float *vx, float *vy,
                                                      const uint8_t *action, float *reward, uint8_t *goal,
                                                      const float *edges, const uint64_t *cellmask, MapScalars ms) {
    // the fused step's physics, wave by wave (pinball_wave_*: free flight in place, (env, edge) pairs on the wave's own lanes)
    __shared__ __attribute__((aligned(16))) float s_edges[MAX_EDGES * 8];
    __shared__ uint32_t s_items[4][PITEMS];
    __shared__ float s_xs[4][4 * 64];
    __shared__ uint8_t s_g[4][64];
    for (int i = threadIdx.x; i < ms.n_edges * 8; i += 256) s_edges[i] = edges[i];
    __syncthreads();
    const int e = blockIdx.x * 256 + threadIdx.x, wv = threadIdx.x >> 6;
    const bool valid = e < n;
    float sx = 0.5f, sy = 0.5f, svx = 0.0f, svy = 0.0f;
    int a = NACT - 1;
    if (valid) { sx = x[e]; sy = y[e]; svx = vx[e]; svy = vy[e]; a = action[e]; }
    bool g, par;
    const int groups = pinball_wave_prepare_any(s_edges, cellmask, ms, valid, sx, sy, svx, svy, a, g, par, s_items[wv], s_xs[wv], 64);
    wave_lds_sync();
    for (int q = 0; q < groups; ++q) pinball_wave_group(s_edges, ms, s_items[wv] + 64 * q, s_xs[wv], 64, s_g[wv]);
    wave_lds_sync();
    const float r = pinball_wave_finish(par, sx, sy, svx, svy, a, g, s_xs[wv], 64, s_g[wv]);
    if (valid) {
        x[e] = sx; y[e] = sy; vx[e] = svx; vy[e] = svy;
        reward[e] = r; goal[e] = g ? 1 : 0;
    }
}

// one wavefront per env: materialises phi[n][1296] (the fused path never does this)
__global__ __launch_bounds__(64) void features_kernel(int n, const float *x, const float *y, const float *vx,
                                                      const float *vy, float *phi) {
    __shared__ float2 s_pw[20];
    __shared__ float2 s_abcd[72];
    const int lane = threadIdx.x;
    for (int e = blockIdx.x; e < n; e += gridDim.x) {
        if (lane == 0) state_powers(x[e], y[e], vx[e], vy[e], s_pw);
        wave_lds_sync();
        for (int p = lane; p < 72; p += 64) {
            const int q = p % 36, d0 = p < 36 ? 0 : 2;
            float2 v = pow_at(s_pw, d0 + 1, q % 6);                       // row 0; row c = row c - 1 times Z_d0^1
            for (int c = 1; c <= q / 6; ++c) v = cmul(v, pow_at(s_pw, d0, 1));
            s_abcd[p] = v;
        }
        wave_lds_sync();
        for (int f = lane; f < NF; f += 64) {
            const float2 ab = s_abcd[f / 36], cd = s_abcd[36 + f % 36];
            phi[(size_t)e * NF + f] = fmaf(-ab.y, cd.y, ab.x * cd.x);
        }
        wave_lds_sync();
    }
}

__global__ __launch_bounds__(256) void predict_kernel(int n, const float *x, const float *y, const float *w8,
                                                      uint8_t *out) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e < n) out[e] = clf_z(w8, x[e], y[e]) > 0.0f ? 1 : 0;
}

// SPEC §6: FIT_G workgroups of FIT_T threads per option. Thread gamma = j FIT_T + tau owns examples i = gamma (mod
// FIT_G FIT_T) and keeps the first FIT_EPT of them in registers for all iterations (65 536 examples per option; more are
// re-read from memory). Per iteration: per-thread fma chains -> butterfly inside each wave -> the workgroup's 16 waves in
// order -> the option's FIT_G workgroup partials in order, exchanged through global memory behind a counter barrier
// (the partials are double-buffered by iteration parity; FIT_G x n_fit <= 64 workgroups are co-resident by construction,
// and every spin is bounded). One 256-thread workgroup per option took 12.6 ms for 40 000 examples x 400 iterations.
constexpr int FIT_G = 8, FIT_T = 1024, FIT_EPT = 8, FIT_BATCH = 8;
constexpr int FIT_STRIDE = FIT_G * FIT_T;

__global__ __launch_bounds__(FIT_T) void fit_kernel(const float *xy, const uint8_t *label, const int32_t *offsets,
                                                    float *w, int iters, float lr, float l2, int q0,
                                                    unsigned long long *part, unsigned long long timeout_ticks,
                                                    uint32_t *async_word) {
    __shared__ float sw[8];
    __shared__ float swave[FIT_T / 64][6];
    __shared__ int s_abort;
    const int ql = blockIdx.y, q = q0 + ql, j = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i0 = offsets[q], M = offsets[q + 1] - offsets[q];
    if (M <= 0) return;                                   // the option's FIT_G workgroups all take this exit
    if (tid < 8) sw[tid] = w[CLF_STRIDE * q + tid];
    if (tid == 0) s_abort = 0;
    const int gamma = j * FIT_T + tid;
    float cu[FIT_EPT], cv[FIT_EPT], cl[FIT_EPT];
#pragma unroll
    for (int e = 0; e < FIT_EPT; ++e) {
        const int i = gamma + FIT_STRIDE * e;
        cu[e] = 0.0f; cv[e] = 0.0f; cl[e] = 0.0f;
        if (i < M) {
            cu[e] = fmaf(xy[2 * (size_t)(i0 + i)], 2.0f, -1.0f);
            cv[e] = fmaf(xy[2 * (size_t)(i0 + i) + 1], 2.0f, -1.0f);
            cl[e] = (float)label[i0 + i];
        }
    }
    const float invM = 1.0f / (float)M;
    unsigned long long *my_part = part + (size_t)ql * 2 * FIT_G * 8;
    for (int it = 0; it < iters; ++it) {
        __syncthreads();
        float wl[6];
#pragma unroll
        for (int jj = 0; jj < 6; ++jj) wl[jj] = sw[jj];
        float g[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        auto one = [&](float u, float v, float lbl) {
            const float psi[6] = {1.0f, u, v, u * u, u * v, v * v};
            float z = wl[0];
            z = fmaf(wl[1], u, z); z = fmaf(wl[2], v, z);
            z = fmaf(wl[3], psi[3], z); z = fmaf(wl[4], psi[4], z); z = fmaf(wl[5], psi[5], z);
            const float e = sigmoid_spec(z) - lbl;
#pragma unroll
            for (int jj = 0; jj < 6; ++jj) g[jj] = fmaf(e, psi[jj], g[jj]);
        };
#pragma unroll
        for (int e = 0; e < FIT_EPT; ++e)
            if (gamma + FIT_STRIDE * e < M) one(cu[e], cv[e], cl[e]);
        for (int i = gamma + FIT_STRIDE * FIT_EPT; i < M; i += FIT_STRIDE)          // beyond the register-resident part
            one(fmaf(xy[2 * (size_t)(i0 + i)], 2.0f, -1.0f), fmaf(xy[2 * (size_t)(i0 + i) + 1], 2.0f, -1.0f), (float)label[i0 + i]);
#pragma unroll
        for (int jj = 0; jj < 6; ++jj) {
            g[jj] = wave_sum(g[jj]);
            if (lane == 0) swave[wave][jj] = g[jj];
        }
        __syncthreads();
        // exchange of the workgroup partials: every value travels as ONE 64-bit word {iteration tag, float bits}, stored
        // and polled with 64-bit relaxed agent-scope atomics — a value that carries the awaited tag is valid by itself, so
        // the exchange costs one store and one (polled) load round trip; buffers alternate by iteration parity (a fast
        // workgroup writes iteration it + 1 while a slow one still reads iteration it)
        unsigned long long *buf = my_part + (it & 1) * FIT_G * 8;
        const unsigned long long tag = (unsigned long long)(unsigned)(it + 1) << 32;
        if (tid < 6) {
            float ps = swave[0][tid];
#pragma unroll
            for (int wv = 1; wv < FIT_T / 64; ++wv) ps = ps + swave[wv][tid];
            __hip_atomic_store(&buf[j * 8 + tid], tag | __float_as_uint(ps), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (wave == 0) {
            unsigned long long v = tag;
            if (lane < 6 * FIT_G) {                          // lane -> (workgroup lane / 6, component lane % 6)
                const unsigned long long *src = &buf[(lane / 6) * 8 + lane % 6];
                // The option's FIT_G workgroups must all be running for this to complete. A plain launch (and a
                // cooperative one: MI355X_MICROARCH.md, residency) promises that only on an otherwise idle card: another
                // stream or process may hold CUs. A late partner is waited for on the 100 MHz wall clock — seconds,
                // not a spin count — and a partner that never shows up ABORTS the fit: weights left as they were,
                // SCG_ASYNC_FIT_TIMEOUT raised in the ctx's host-visible status word (scg_async_status).
                unsigned long long t0 = 0;
                int spins = 0;
                while (((v = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 32) != (tag >> 32)) {
                    __builtin_amdgcn_s_sleep(1);
                    if ((++spins & 255) == 0) {
                        const unsigned long long now = __builtin_amdgcn_s_memrealtime();
                        if (t0 == 0) t0 = now;
                        else if (now - t0 > timeout_ticks) { s_abort = 1; break; }
                    }
                }
            }
            const float val = __uint_as_float((unsigned)v);
            const int c = lane < 6 ? lane : 0;
            float gs = __shfl(val, c, 64);                   // the FIT_G group sums in order
#pragma unroll
            for (int jw = 1; jw < FIT_G; ++jw) gs = gs + __shfl(val, jw * 6 + c, 64);
            if (lane < 6) {
                const float reg = (lane > 0) ? l2 * sw[lane] : 0.0f;
                sw[lane] = sw[lane] - lr * ((gs * invM) + reg);
            }
        }
        __syncthreads();
        if (s_abort) break;
    }
    __syncthreads();
    if (s_abort) {                                        // no silent NaN row: w keeps its old value, the host is told
        if (tid == 0 && async_word)
            __hip_atomic_fetch_or(async_word, SCG_ASYNC_FIT_TIMEOUT | (0x100u << (q & 15)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
    }
    if (j == 0 && tid < 6) w[CLF_STRIDE * q + tid] = sw[tid];
}

// ------------------------------------------------------------------------------------------------
// host side: the C-ABI
struct scg_ctx {
    scg_config cfg;
    int n_vf;
    int nblk;
    bool have_map;
    MapScalars ms;
    float *d_edges, *d_starts, *d_scale;
    uint64_t *d_cellmask;
    int32_t *d_perm, *d_hist;      // SPEC §5 env order of the current step (d_hist: scratch of the stand-alone sort)
    int32_t *d_collect_rows;       // scg_collect_examples: per-row totals [rows of COL_ROW envs] + the buffer's fill level
    uint32_t arm_bits;             // scg_arm_collect: the announced trigger (0 = none) ...
    const uint8_t *arm_prev;
    const int32_t *arm_count;
    int32_t arm_L;
    bool arm_rows_ready;           // ... and whether the last scg_step left its row totals in d_collect_rows
    unsigned long long *d_fit_part;   // fit_kernel: tagged workgroup partials [FIT_BATCH][2][FIT_G][8]
    uint32_t *h_async;             // pinned, device-visible status word: kernels that give up OR their reason into it
    uint32_t *d_async;             // ... its device address
    double fit_timeout_s;          // how long fit_kernel waits for a workgroup that is not running yet
    float4 *d_outrec;              // [nblk * BLOCK_ENVS][4] per-position step results (td_kernel -> commit_row)
    int32_t *d_invperm;            // [n_envs] position of each env in d_perm
    int32_t *d_hist2[2];           // per-row counts of the option ids a learning step leaves (double-buffered)
    int hist_parity;
    bool hist_dirty;               // a failed call may have left counts behind: clear both before the next use
    bool order_valid;              // d_perm already holds the order of the ids in order_ids (made by the last learning step)
    const int32_t *order_ids;
    uint32_t parents;              // packed option targets (default: the chain k -> k-1)
    uint32_t gest;                 // SPEC §4.4 options in gestation
    int32_t *gest_succ;            // caller-owned device counters [n_vf] (NULL = none)
    float *ring_x, *ring_y;        // SPEC §7 caller-owned trace buffers (NULL = off)
    uint8_t *events;
    int32_t *ev_len;
    int32_t ring_len;
    unsigned long long *d_stamps;   // diagnostic build only (NULL otherwise)
    float *d_slabs;
    int32_t *d_cnts;
    float *d_G;
    int32_t *d_nk;
    float *G_out;          // where reduce leaves G / n_k (ctx-owned by default)
    int32_t *nk_out;
    float *nkf_out;        // packed operand: float copy of the counts right after G (null = off)
    bool prof_on;          // measurement hook: event pairs round the fused kernel
    int prof_every;        // ... of every prof_every-th launch (events cost a few us of queue bubble each)
    long long prof_seen;
    std::vector<hipEvent_t> *prof_ev;
    size_t prof_used;
    char err[256];
};

static thread_local char g_err[256] = "";

#define SCG_HIP(ctx, call)                                                                     \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess) {                                                                \
            snprintf((ctx) ? (ctx)->err : g_err, 256, "%s failed: %s", #call, hipGetErrorString(e_)); \
            return SCG_ERR_HIP;                                                                \
        }                                                                                      \
    } while (0)

static int fail(scg_ctx *ctx, int code, const char *msg) {
    snprintf(ctx ? ctx->err : g_err, 256, "%s", msg);
    return code;
}

// Sticky device-side failures (include/scg_abi.h, "asynchronous failures"): checked before every launch.
static int decode_async(uint32_t word, char *buf, size_t n) {
    if (word == 0) { if (buf && n) buf[0] = 0; return SCG_OK; }
    if (buf && n) {
        if (word & SCG_ASYNC_FIT_TIMEOUT)
            snprintf(buf, n, "an earlier scg_fit_initiation gave up (problem mask 0x%x): its workgroups did not become "
                     "co-resident within the fit timeout (card shared with other work?); the affected classifier rows were "
                     "left unchanged. scg_clear_async_error() re-arms the context", (word >> 8) & 0xffffu);
        else
            snprintf(buf, n, "unknown asynchronous device status 0x%x", word);
    }
    return SCG_ERR_ASYNC;
}
static int async_pending(scg_ctx *c);
#define SCG_CHECK_ASYNC(c) do { if (async_pending(c)) return SCG_ERR_ASYNC; } while (0)

// Every entry point that touches the device runs with the ctx's device current and leaves the caller's current
// device as it found it (a multi-GPU caller that forgot torch.cuda.set_device would otherwise launch on the
// wrong card, against buffers owned by another GPU).
struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev) {
        int cur = -1;
        if (hipGetDevice(&cur) != hipSuccess) { ok = false; return; }
        if (cur != dev) {
            if (hipSetDevice(dev) != hipSuccess) { ok = false; return; }
            prev = cur;
        }
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
};
#define SCG_ON_DEVICE(c, what)                                                                   \
    DeviceGuard dev_guard_((c)->cfg.device);                                                     \
    if (!dev_guard_.ok) return fail((c), SCG_ERR_HIP, what ": cannot make the context's device current")

static int async_pending(scg_ctx *c) {
    if (!c || !c->h_async) return 0;
    const uint32_t w = *reinterpret_cast<volatile uint32_t *>(c->h_async);
    if (w == 0) return 0;
    decode_async(w, c->err, sizeof(c->err));
    return 1;
}

extern "C" {

int scg_abi_version(void) { return SCG_ABI_VERSION; }

int scg_decode_async_word(uint32_t word, char *buf, int32_t buf_len) {
    return decode_async(word, buf, buf_len > 0 ? (size_t)buf_len : 0);
}

int scg_async_status(scg_ctx *c, void *stream, int32_t synchronize, uint32_t *word_out) {
    if (!c) return fail(nullptr, SCG_ERR_INVALID, "scg_async_status: null ctx");
    if (synchronize) {
        SCG_ON_DEVICE(c, "scg_async_status");
        SCG_HIP(c, hipStreamSynchronize(reinterpret_cast<hipStream_t>(stream)));
    }
    const uint32_t w = c->h_async ? *reinterpret_cast<volatile uint32_t *>(c->h_async) : 0u;
    if (word_out) *word_out = w;
    return decode_async(w, c->err, sizeof(c->err));
}

int scg_clear_async_error(scg_ctx *c) {
    if (!c) return fail(nullptr, SCG_ERR_INVALID, "scg_clear_async_error: null ctx");
    if (c->h_async) *reinterpret_cast<volatile uint32_t *>(c->h_async) = 0u;
    return SCG_OK;
}

int scg_set_fit_timeout(scg_ctx *c, double seconds) {
    if (!c) return fail(nullptr, SCG_ERR_INVALID, "scg_set_fit_timeout: null ctx");
    if (!(seconds >= 0.0) || seconds > 3600.0) return fail(c, SCG_ERR_INVALID, "scg_set_fit_timeout: seconds must be in [0, 3600]");
    c->fit_timeout_s = seconds;
    return SCG_OK;
}

int scg_debug_raise_async(scg_ctx *c, uint32_t word) {
    if (!c || !c->h_async) return fail(c, SCG_ERR_INVALID, "scg_debug_raise_async: null ctx");
    *reinterpret_cast<volatile uint32_t *>(c->h_async) |= word;
    return SCG_OK;
}

int scg_block_envs(void) { return BLOCK_ENVS; }

const char *scg_strerror(int status) {
    switch (status) {
        case SCG_OK: return "ok";
        case SCG_ERR_INVALID: return "invalid argument";
        case SCG_ERR_NO_DEVICE: return "no usable HIP device";
        case SCG_ERR_HIP: return "HIP runtime error";
        case SCG_ERR_STATE: return "call order / state error";
        case SCG_ERR_ASYNC: return "an earlier launch failed on the device";
        default: return "unknown status";
    }
}

const char *scg_last_error(const scg_ctx *ctx) { return ctx ? ctx->err : g_err; }

int scg_create(scg_ctx **out, const scg_config *cfg) {
    if (!out || !cfg) return fail(nullptr, SCG_ERR_INVALID, "scg_create: null argument");
    *out = nullptr;
    if (cfg->n_envs < 1) return fail(nullptr, SCG_ERR_INVALID, "scg_create: n_envs must be >= 1");
    if (cfg->n_options < 0 || cfg->n_options > SCG_MAX_OPTIONS)
        return fail(nullptr, SCG_ERR_INVALID, "scg_create: n_options out of range [0,5]");
    if (cfg->fourier_order != SCG_FOURIER_ORDER)
        return fail(nullptr, SCG_ERR_INVALID, "scg_create: only Fourier order 5 is built");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(nullptr, SCG_ERR_NO_DEVICE, "scg_create: no HIP device visible");
    if (cfg->device < 0 || cfg->device >= ndev)
        return fail(nullptr, SCG_ERR_INVALID, "scg_create: device ordinal out of range");
    scg_ctx *c = new (std::nothrow) scg_ctx();
    if (!c) return fail(nullptr, SCG_ERR_INVALID, "scg_create: out of host memory");
    memset(c, 0, sizeof(*c));
    c->cfg = *cfg;
    c->n_vf = cfg->n_options + 1;
    c->nblk = (cfg->n_envs + BLOCK_ENVS - 1) / BLOCK_ENVS;
    int st = SCG_OK;
    DeviceGuard dev_guard_(cfg->device);
    do {
        if (!dev_guard_.ok) { st = SCG_ERR_HIP; break; }
        const size_t slab_bytes = (size_t)c->nblk * c->n_vf * NACT * NF * sizeof(float);
        if (hipMalloc(&c->d_slabs, slab_bytes) != hipSuccess) { st = SCG_ERR_HIP; break; }
        if (hipMalloc(&c->d_cnts, (size_t)c->nblk * c->n_vf * sizeof(int32_t)) != hipSuccess) { st = SCG_ERR_HIP; break; }
        if (hipMalloc(&c->d_G, (size_t)c->n_vf * NACT * NF * sizeof(float)) != hipSuccess) { st = SCG_ERR_HIP; break; }
        if (hipMalloc(&c->d_nk, MAX_VF * sizeof(int32_t)) != hipSuccess) { st = SCG_ERR_HIP; break; }
        if (hipMalloc(&c->d_edges, MAX_EDGES * 8 * sizeof(float)) != hipSuccess) { st = SCG_ERR_HIP; break; }
        if (hipMalloc(&c->d_scale, NF * sizeof(float)) != hipSuccess) { st = SCG_ERR_HIP; break; }
        if (hipMalloc(&c->d_perm, (size_t)c->nblk * BLOCK_ENVS * sizeof(int32_t)) != hipSuccess) { st = SCG_ERR_HIP; break; }
        if (hipMalloc(&c->d_hist, (size_t)c->nblk * 8 * sizeof(int32_t)) != hipSuccess) { st = SCG_ERR_HIP; break; }
        if (hipMalloc(&c->d_collect_rows, (size_t)((cfg->n_envs + COL_ROW - 1) / COL_ROW + 1) * sizeof(int32_t)) != hipSuccess) { st = SCG_ERR_HIP; break; }
        if (hipMalloc(&c->d_fit_part, (size_t)FIT_BATCH * 2 * FIT_G * 8 * sizeof(unsigned long long)) != hipSuccess) { st = SCG_ERR_HIP; break; }
        if (hipHostMalloc(reinterpret_cast<void **>(&c->h_async), 64, hipHostMallocMapped) != hipSuccess) { st = SCG_ERR_HIP; break; }
        *c->h_async = 0u;
        if (hipHostGetDevicePointer(reinterpret_cast<void **>(&c->d_async), c->h_async, 0) != hipSuccess) { st = SCG_ERR_HIP; break; }
        if (hipMalloc(&c->d_outrec, (size_t)c->nblk * BLOCK_ENVS * 4 * sizeof(float4)) != hipSuccess) { st = SCG_ERR_HIP; break; }
        if (hipMalloc(&c->d_invperm, (size_t)c->nblk * BLOCK_ENVS * sizeof(int32_t)) != hipSuccess) { st = SCG_ERR_HIP; break; }
        {
            const size_t hb = (size_t)((c->cfg.n_envs + 255) / 256) * 8 * sizeof(int32_t);
            if (hipMalloc(&c->d_hist2[0], hb) != hipSuccess || hipMalloc(&c->d_hist2[1], hb) != hipSuccess) { st = SCG_ERR_HIP; break; }
            if (hipMemset(c->d_hist2[0], 0, hb) != hipSuccess || hipMemset(c->d_hist2[1], 0, hb) != hipSuccess) { st = SCG_ERR_HIP; break; }
        }
        if (hipMalloc(&c->d_cellmask, (size_t)CELL_G * CELL_G * 4 * sizeof(uint64_t)) != hipSuccess) { st = SCG_ERR_HIP; break; }
        if (hipMemset(c->d_cnts, 0, (size_t)c->nblk * c->n_vf * sizeof(int32_t)) != hipSuccess) { st = SCG_ERR_HIP; break; }
        if (hipMemset(c->d_G, 0, (size_t)c->n_vf * NACT * NF * sizeof(float)) != hipSuccess) { st = SCG_ERR_HIP; break; }
        if (hipMemset(c->d_nk, 0, MAX_VF * sizeof(int32_t)) != hipSuccess) { st = SCG_ERR_HIP; break; }
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&td_kernel<MODE_FUSED>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess) { st = SCG_ERR_HIP; break; }
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&td_kernel<MODE_TRANS>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess) { st = SCG_ERR_HIP; break; }
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&td_kernel<MODE_QVAL>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess) { st = SCG_ERR_HIP; break; }
    } while (0);
    if (st != SCG_OK) {
        snprintf(g_err, 256, "scg_create: device allocation/setup failed: %s", hipGetErrorString(hipGetLastError()));
        scg_destroy(c);
        return st;
    }
#ifdef SCG_STAMPS
    if (hipMalloc(&c->d_stamps, (size_t)c->nblk * 32 * sizeof(unsigned long long)) == hipSuccess)
        (void)hipMemset(c->d_stamps, 0, (size_t)c->nblk * 32 * sizeof(unsigned long long));
#endif
    c->parents = 0;
    for (int k = 1; k < MAX_VF; ++k) c->parents |= (uint32_t)(k - 1) << (3 * k);      // chain: 1 -> goal, k -> k-1
    c->G_out = c->d_G; c->nk_out = c->d_nk;
    c->fit_timeout_s = 2.0;
    *out = c;
    return SCG_OK;
}

int scg_destroy(scg_ctx *c) {
    if (!c) return SCG_OK;
    (void)hipFree(c->d_slabs); (void)hipFree(c->d_cnts); (void)hipFree(c->d_G); (void)hipFree(c->d_nk);
    (void)hipFree(c->d_hist2[0]); (void)hipFree(c->d_hist2[1]); (void)hipFree(c->d_outrec); (void)hipFree(c->d_invperm);
    (void)hipFree(c->d_edges); (void)hipFree(c->d_starts); (void)hipFree(c->d_scale); (void)hipFree(c->d_cellmask); (void)hipFree(c->d_perm); (void)hipFree(c->d_hist);
    (void)hipFree(c->d_fit_part); (void)hipFree(c->d_collect_rows);
    if (c->h_async) (void)hipHostFree(c->h_async);
    if (c->prof_ev) {
        for (hipEvent_t e : *c->prof_ev) (void)hipEventDestroy(e);
        delete c->prof_ev;
    }
    delete c;
    return SCG_OK;
}

int scg_set_hparams(scg_ctx *c, float gamma, float alpha, float epsilon, float r_option_success,
                    int32_t max_episode_steps, int32_t max_option_steps) {
    if (!c) return fail(nullptr, SCG_ERR_INVALID, "scg_set_hparams: null ctx");
    c->cfg.gamma = gamma; c->cfg.alpha = alpha; c->cfg.epsilon = epsilon;
    c->cfg.r_option_success = r_option_success;
    c->cfg.max_episode_steps = max_episode_steps; c->cfg.max_option_steps = max_option_steps;
    return SCG_OK;
}

int scg_set_map(scg_ctx *c, const float *edges, int32_t n_edges, const float *starts, int32_t n_starts,
                const float map_scalars[6], const float *scale) {
    if (!c) return fail(nullptr, SCG_ERR_INVALID, "scg_set_map: null ctx");
    if (!edges || !starts || !map_scalars || !scale) return fail(c, SCG_ERR_INVALID, "scg_set_map: null argument");
    if (n_edges < 0 || n_edges > MAX_EDGES) return fail(c, SCG_ERR_INVALID, "scg_set_map: n_edges out of range [0,256]");
    if (n_starts < 1) return fail(c, SCG_ERR_INVALID, "scg_set_map: need at least one start position");
    SCG_ON_DEVICE(c, "scg_set_map");
    if (c->d_starts) { (void)hipFree(c->d_starts); c->d_starts = nullptr; }
    SCG_HIP(c, hipMalloc(&c->d_starts, (size_t)n_starts * 2 * sizeof(float)));
    if (n_edges > 0) SCG_HIP(c, hipMemcpy(c->d_edges, edges, (size_t)n_edges * 8 * sizeof(float), hipMemcpyHostToDevice));
    SCG_HIP(c, hipMemcpy(c->d_starts, starts, (size_t)n_starts * 2 * sizeof(float), hipMemcpyHostToDevice));
    SCG_HIP(c, hipMemcpy(c->d_scale, scale, NF * sizeof(float), hipMemcpyHostToDevice));
    const float R = map_scalars[0];
    c->ms.hstep = map_scalars[1]; c->ms.R2 = map_scalars[2];
    c->ms.TX = map_scalars[3]; c->ms.TY = map_scalars[4]; c->ms.TR2 = map_scalars[5];
    c->ms.R = R; c->ms.TR = (float)(std::sqrt((double)map_scalars[5]) * 1.0001);
    // Candidate masks (an internal acceleration table, not part of the arithmetic contract): cell (cx,cy)
    // lists every edge within  R(1.02 + 1.10*|v|max) + half a cell diagonal  of the cell centre, so the
    // mask of the ball's cell is a superset of the edges the per-step bound of pinball_step() can admit.
    {
        std::vector<uint64_t> cm((size_t)CELL_G * CELL_G * 4, 0);
        const double vmax = 2.0 * std::sqrt(2.0) * 1.001;
        const double reach = (double)R * (1.02 + 1.10 * vmax) * 1.01 + 0.5 * std::sqrt(2.0) / CELL_G + 1e-6;
        for (int cy = 0; cy < CELL_G; ++cy)
            for (int cx = 0; cx < CELL_G; ++cx) {
                const double px = (cx + 0.5) / CELL_G, py = (cy + 0.5) / CELL_G;
                for (int j = 0; j < n_edges; ++j) {
                    const float *E = edges + 8 * j;
                    const double dx = px - E[0], dy = py - E[1];
                    double t = (dx * E[2] + dy * E[3]) * E[4];
                    t = t < 0 ? 0 : (t > 1 ? 1 : t);
                    const double qx = E[0] + E[2] * t - px, qy = E[1] + E[3] * t - py;
                    if (std::sqrt(qx * qx + qy * qy) <= reach)
                        cm[((size_t)cy * CELL_G + cx) * 4 + (j >> 6)] |= (1ull << (j & 63));
                }
            }
        SCG_HIP(c, hipMemcpy(c->d_cellmask, cm.data(), cm.size() * sizeof(uint64_t), hipMemcpyHostToDevice));
    }
    c->ms.n_edges = n_edges; c->ms.n_starts = n_starts;
    c->have_map = true;
    return SCG_OK;
}

static void fill_common(const scg_ctx *c, StepArgs &A) {
    memset(&A, 0, sizeof(A));
    A.n_vf = c->n_vf;
    A.seed = c->cfg.seed; A.env_base = c->cfg.env_id_base;
    A.gamma = c->cfg.gamma; A.epsilon = c->cfg.epsilon; A.r_succ = c->cfg.r_option_success;
    A.max_ep = c->cfg.max_episode_steps; A.max_opt = c->cfg.max_option_steps;
    A.ms = c->ms;
    A.edges = c->d_edges; A.starts = c->d_starts; A.cellmask = c->d_cellmask;
    A.slabs = c->d_slabs; A.cnts = c->d_cnts;
    A.parents = c->parents;
    A.gest = c->gest; A.gest_succ = c->gest_succ;
    A.ring_x = c->ring_x; A.ring_y = c->ring_y; A.events = c->events; A.ev_len = c->ev_len;
    A.ring_mask = c->ring_len > 0 ? c->ring_len - 1 : 0;
    A.stamps = c->d_stamps;
}

// The reduce launch; for the fused step (`st` given) its extra workgroups also commit the step's per-position
// results to the caller's arrays and, with `sort`, place every row in the next step's env order.
static int launch_reduce(scg_ctx *c, float *W, uint32_t apply, int nblk, hipStream_t s,
                         const StepArgs *st = nullptr, bool sort = false, bool reduce = true) {
    ReduceArgs R;
    memset(&R, 0, sizeof(R));
    R.slabs = c->d_slabs; R.cnts = c->d_cnts; R.G = c->G_out; R.n_k = c->nk_out; R.nk_f = c->nkf_out; R.W = W; R.scale = c->d_scale;
    R.nblk = nblk; R.n_vf = c->n_vf; R.alpha = c->cfg.alpha; R.apply = apply;
    const int nrow = st ? (c->cfg.n_envs + 255) / 256 : 0;
    R.n = c->cfg.n_envs; R.nrow = nrow;
    if (st) {
        R.outrec = c->d_outrec; R.invperm = c->d_invperm; R.perm = c->d_perm; R.sort = sort ? 1 : 0;
        R.hist = c->d_hist2[c->hist_parity]; R.hist_zero = c->d_hist2[c->hist_parity ^ 1];
        R.x = st->x; R.y = st->y; R.vx = st->vx; R.vy = st->vy; R.reward = st->reward;
        R.option_id_out = st->option_id; R.opt_steps = st->opt_steps; R.ep_steps = st->ep_steps;
        R.action = st->action; R.done = st->done;
        R.qcache = st->k_hi >= 0 ? st->qcache : nullptr;
        if (c->arm_bits && c->events && c->ring_x) {
            R.c_events = c->events; R.c_prev = c->arm_prev; R.c_evlen = c->ev_len; R.c_count = c->arm_count;
            R.c_rows = c->d_collect_rows; R.c_bits = c->arm_bits; R.c_L = c->arm_L; R.c_ring_len = c->ring_len;
        }
    }
    if (!reduce) {                                       // acting-only step: the commit alone
        hipLaunchKernelGGL(commit_kernel, dim3(nrow), dim3(256), 0, s, R);
        SCG_HIP(c, hipGetLastError());
        return SCG_OK;
    }
    const int sy = (nrow + RED_NCOL - 1) / RED_NCOL;
    hipLaunchKernelGGL(reduce_kernel, dim3(RED_NCOL, c->n_vf + sy), dim3(RED_THREADS), 0, s, R);
    SCG_HIP(c, hipGetLastError());
    return SCG_OK;
}

int scg_step(scg_ctx *c, float *x, float *y, float *vx, float *vy, int32_t *option_id, int32_t *opt_steps,
             int32_t *ep_steps, float *qcache, uint8_t *action, float *reward, uint8_t *done, float *W,
             const float *clf, uint32_t enabled_mask, uint64_t t, uint32_t flags, void *stream) {
    if (!c) return fail(nullptr, SCG_ERR_INVALID, "scg_step: null ctx");
    if (!c->have_map) return fail(c, SCG_ERR_STATE, "scg_step: scg_set_map has not been called");
    if (!x || !y || !vx || !vy || !option_id || !opt_steps || !ep_steps || !qcache || !action || !reward ||
        !done || !W || !clf)
        return fail(c, SCG_ERR_INVALID, "scg_step: null array argument");
    SCG_CHECK_ASYNC(c);
    SCG_ON_DEVICE(c, "scg_step");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    StepArgs A;
    fill_common(c, A);
    A.x = x; A.y = y; A.vx = vx; A.vy = vy;
    A.option_id = option_id; A.opt_steps = opt_steps; A.ep_steps = ep_steps; A.qcache = qcache;
    A.action = action; A.reward = reward; A.done = done;
    A.W = W; A.clf = clf;
    A.n = c->cfg.n_envs; A.k_lo = 0; A.k_hi = c->n_vf - 1;
    A.enabled = enabled_mask; A.learn = (flags & SCG_STEP_LEARN) ? 1u : 0u; A.t = t;
    if (flags & 0x100u) A.k_hi = -1;     // diagnostic only (bench.py --diag-no-td): skip the TD passes
    // env order of this step (SPEC §5): counting sort by the option ids the previous step left
    // A learning step computes the NEXT step's order inside its reduce launches; the stand-alone sort runs only
    // when that order is missing or was invalidated (first step, other array, scg_invalidate_order).
    if (!(c->order_valid && c->order_ids == option_id)) {
        const int nrow = (c->cfg.n_envs + 255) / 256;    // the sort works on rows of 256 envs whatever the workgroup size
        hipLaunchKernelGGL(sort_hist_kernel, dim3(nrow), dim3(256), 0, s, option_id, c->cfg.n_envs, c->n_vf, c->d_hist);
        hipLaunchKernelGGL(sort_scatter_kernel, dim3(nrow), dim3(256), 0, s, option_id, c->cfg.n_envs, c->n_vf, nrow,
                           c->d_hist, c->d_perm, c->d_invperm);
        SCG_HIP(c, hipGetLastError());
    }
    c->order_valid = false;
    if (c->hist_dirty) {
        const size_t hb = (size_t)((c->cfg.n_envs + 255) / 256) * 8 * sizeof(int32_t);
        SCG_HIP(c, hipMemsetAsync(c->d_hist2[0], 0, hb, s));
        SCG_HIP(c, hipMemsetAsync(c->d_hist2[1], 0, hb, s));
        c->hist_dirty = false;
    }
    const bool fold = (flags & SCG_STEP_LEARN) && !(flags & 0x200u);      // 0x200: diagnostic, sort afresh every step
    A.hist_next = fold ? c->d_hist2[c->hist_parity] : nullptr;
    A.perm = c->d_perm; A.outrec = c->d_outrec;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    if (c->prof_on && (c->prof_seen++ % c->prof_every) == c->prof_every / 2) {     // (not the first launch into an idle queue)
        if (!c->prof_ev) c->prof_ev = new std::vector<hipEvent_t>();
        while (c->prof_ev->size() < c->prof_used + 2) {
            hipEvent_t e;
            SCG_HIP(c, hipEventCreate(&e));
            c->prof_ev->push_back(e);
        }
        ev0 = (*c->prof_ev)[c->prof_used]; ev1 = (*c->prof_ev)[c->prof_used + 1];
        c->prof_used += 2;
        SCG_HIP(c, hipEventRecord(ev0, s));
    }
    hipLaunchKernelGGL(td_kernel<MODE_FUSED>, dim3(c->nblk), dim3(THREADS), LDS_BYTES, s, A);
    SCG_HIP(c, hipGetLastError());
    if (ev1) SCG_HIP(c, hipEventRecord(ev1, s));
    // results reach the caller's arrays through the commit workgroups of the reduce launch (or a commit launch)
    c->arm_rows_ready = c->arm_bits && c->events && c->ring_x;      // ... which also leave an announced trigger's row totals
    if (!(flags & SCG_STEP_LEARN)) return launch_reduce(c, W, 0u, c->nblk, s, &A, false, false);
    if (!fold) return launch_reduce(c, W, (flags & SCG_STEP_APPLY) ? 1u : 0u, c->nblk, s, &A, false);
    c->hist_dirty = true;                          // until the reduce launch has consumed and re-armed the counts
    const int rc = launch_reduce(c, W, (flags & SCG_STEP_APPLY) ? 1u : 0u, c->nblk, s, &A, true);
    if (rc != SCG_OK) return rc;
    c->hist_dirty = false;
    c->hist_parity ^= 1;
    c->order_valid = true; c->order_ids = option_id;
    return SCG_OK;
}

int scg_invalidate_order(scg_ctx *c) {
    if (!c) return fail(nullptr, SCG_ERR_INVALID, "scg_invalidate_order: null ctx");
    c->order_valid = false;
    return SCG_OK;
}

#ifdef SCG_STAMPS
extern "C" int scg_diag_stamps(scg_ctx *c, unsigned long long *host_out /*[nblk][16]*/, int32_t reset) {
    if (!c || !c->d_stamps) return SCG_ERR_STATE;
    if (host_out && hipMemcpy(host_out, c->d_stamps, (size_t)c->nblk * 32 * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return SCG_ERR_HIP;
    if (reset) (void)hipMemset(c->d_stamps, 0, (size_t)c->nblk * 32 * sizeof(unsigned long long));
    return SCG_OK;
}
#endif

int scg_set_option_parents(scg_ctx *c, const int32_t *parents) {
    if (!c || !parents) return fail(c, SCG_ERR_INVALID, "scg_set_option_parents: null argument");
    uint32_t packed = 0;
    for (int k = 1; k <= c->cfg.n_options; ++k) {
        if (parents[k] < 0 || parents[k] > c->cfg.n_options || parents[k] == k)
            return fail(c, SCG_ERR_INVALID, "scg_set_option_parents: parent must be 0 (goal) or another option");
        packed |= (uint32_t)parents[k] << (3 * k);
    }
    for (int k = 1; k <= c->cfg.n_options; ++k) {          // no cycles: following parents must reach the goal
        int p = k, hops = 0;
        while (p != 0 && hops <= SCG_MAX_OPTIONS) { p = parents[p]; ++hops; }
        if (p != 0) return fail(c, SCG_ERR_INVALID, "scg_set_option_parents: the option graph has a cycle");
    }
    c->parents = packed;
    return SCG_OK;
}

int scg_set_gestation(scg_ctx *c, uint32_t gest_mask, int32_t *succ_counts) {
    if (!c) return fail(nullptr, SCG_ERR_INVALID, "scg_set_gestation: null ctx");
    if (gest_mask & ~(((1u << c->n_vf) - 1u) & ~1u)) return fail(c, SCG_ERR_INVALID, "scg_set_gestation: mask names no option of this context");
    c->gest = gest_mask; c->gest_succ = succ_counts;
    return SCG_OK;
}

int scg_collect_examples(scg_ctx *c, uint32_t event_bits, uint8_t *prev_in, int32_t l_pos, int32_t l_neg, float *ex_xy,
                         uint8_t *ex_label, int32_t *count, int32_t cap, void *stream) {
    if (!c) return fail(nullptr, SCG_ERR_INVALID, "scg_collect_examples: null ctx");
    if (!c->ring_x || !c->events) return fail(c, SCG_ERR_STATE, "scg_collect_examples: trace buffers are not attached");
    if (!event_bits || l_pos < 0 || l_neg < 0 || l_pos + l_neg < 1 || !ex_xy || !ex_label || !count || cap < 0)
        return fail(c, SCG_ERR_INVALID, "scg_collect_examples: bad argument");
    SCG_CHECK_ASYNC(c);
    SCG_ON_DEVICE(c, "scg_collect_examples");
    const int nrows = (c->cfg.n_envs + COL_ROW - 1) / COL_ROW;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const bool have_rows = c->arm_rows_ready && c->arm_bits == event_bits && c->arm_prev == prev_in && c->arm_count == count &&
                           c->arm_L == l_pos + l_neg;
    c->arm_rows_ready = false;                               // prev_in changes below: the totals are used up
    if (!have_rows)
    hipLaunchKernelGGL(collect_count_kernel, dim3(nrows), dim3(COL_ROW), 0, s, c->cfg.n_envs, c->events, prev_in, event_bits,
                       c->ev_len, c->ring_len, l_pos + l_neg, c->d_collect_rows, nrows, count);
    hipLaunchKernelGGL(collect_scatter_kernel, dim3(nrows), dim3(COL_ROW), 0, s, c->cfg.n_envs, c->events, prev_in, event_bits,
                       c->ring_x, c->ring_y, c->ring_len, c->ev_len, l_pos, l_neg, ex_xy, ex_label, count, cap,
                       c->d_collect_rows, nrows);
    SCG_HIP(c, hipGetLastError());
    return SCG_OK;
}

int scg_arm_collect(scg_ctx *c, uint32_t event_bits, const uint8_t *prev_in, int32_t l_pos, int32_t l_neg, const int32_t *count) {
    if (!c) return fail(nullptr, SCG_ERR_INVALID, "scg_arm_collect: null ctx");
    c->arm_rows_ready = false;
    if (event_bits == 0) { c->arm_bits = 0; return SCG_OK; }
    if (l_pos < 0 || l_neg < 0 || l_pos + l_neg < 1 || !count) return fail(c, SCG_ERR_INVALID, "scg_arm_collect: bad argument");
    if (!c->ring_x || !c->events) return fail(c, SCG_ERR_STATE, "scg_arm_collect: trace buffers are not attached");
    c->arm_bits = event_bits; c->arm_prev = prev_in; c->arm_count = count; c->arm_L = l_pos + l_neg;
    return SCG_OK;
}

int scg_set_trace_buffers(scg_ctx *c, float *ring_x, float *ring_y, int32_t ring_len, uint8_t *events,
                          int32_t *ev_len) {
    if (!c) return fail(nullptr, SCG_ERR_INVALID, "scg_set_trace_buffers: null ctx");
    c->arm_bits = 0; c->arm_rows_ready = false;             // an announced trigger refers to the old buffers
    if ((ring_x == nullptr) != (ring_y == nullptr)) return fail(c, SCG_ERR_INVALID, "scg_set_trace_buffers: ring_x and ring_y go together");
    if (ring_x && (ring_len < 1 || (ring_len & (ring_len - 1)))) return fail(c, SCG_ERR_INVALID, "scg_set_trace_buffers: ring_len must be a power of two");
    if ((events == nullptr) != (ev_len == nullptr)) return fail(c, SCG_ERR_INVALID, "scg_set_trace_buffers: events and ev_len go together");
    c->ring_x = ring_x; c->ring_y = ring_y; c->ring_len = ring_x ? ring_len : 0; c->events = events; c->ev_len = ev_len;
    return SCG_OK;
}

int scg_harvest(scg_ctx *c, int32_t n_sel, const int32_t *sel_env, const float *ring_x, const float *ring_y,
                int32_t ring_len, const int32_t *ev_len, int32_t l_pos, int32_t l_neg, float *out_xy,
                uint8_t *out_label, void *stream) {
    if (!c) return fail(nullptr, SCG_ERR_INVALID, "scg_harvest: null ctx");
    if (n_sel < 0 || l_pos < 0 || l_neg < 0 || l_pos + l_neg < 1 || ring_len < 1 || (ring_len & (ring_len - 1)) ||
        !sel_env || !ring_x || !ring_y || !ev_len || !out_xy || !out_label)
        return fail(c, SCG_ERR_INVALID, "scg_harvest: bad argument");
    SCG_CHECK_ASYNC(c);
    SCG_ON_DEVICE(c, "scg_harvest");
    if (n_sel == 0) return SCG_OK;
    const long long total = (long long)n_sel * (l_pos + l_neg);
    hipLaunchKernelGGL(harvest_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), n_sel, sel_env, ring_x, ring_y, ring_len, c->cfg.n_envs,
                       ev_len, l_pos, l_neg, out_xy, out_label);
    SCG_HIP(c, hipGetLastError());
    return SCG_OK;
}

int scg_profile_reset(scg_ctx *c, int32_t enable) {
    if (!c) return fail(nullptr, SCG_ERR_INVALID, "scg_profile_reset: null ctx");
    c->prof_on = enable > 0;
    c->prof_every = enable > 0 ? enable : 1;
    c->prof_seen = 0;
    c->prof_used = 0;
    if (enable > 0) {                       // a pool of events up front: creating them lazily cost the first sampled launches of a
        SCG_ON_DEVICE(c, "scg_profile_reset");      // timed region tens of microseconds of host time in front of an idle queue
        if (!c->prof_ev) c->prof_ev = new std::vector<hipEvent_t>();
        while (c->prof_ev->size() < 64) {
            hipEvent_t e;
            SCG_HIP(c, hipEventCreate(&e));
            c->prof_ev->push_back(e);
        }
    }
    return SCG_OK;
}

int scg_profile_read(scg_ctx *c, double *kernel_ms_sum, int64_t *launches) {
    if (!c || !kernel_ms_sum || !launches) return fail(c, SCG_ERR_INVALID, "scg_profile_read: null argument");
    double sum = 0.0;
    for (size_t i = 0; i + 1 < c->prof_used; i += 2) {
        SCG_HIP(c, hipEventSynchronize((*c->prof_ev)[i + 1]));
        float ms = 0.0f;
        SCG_HIP(c, hipEventElapsedTime(&ms, (*c->prof_ev)[i], (*c->prof_ev)[i + 1]));
        sum += ms;
    }
    *kernel_ms_sum = sum;
    *launches = (int64_t)(c->prof_used / 2);
    return SCG_OK;
}

int scg_grad_buffers(scg_ctx *c, float **G, int32_t **n_k) {
    if (!c || !G || !n_k) return fail(c, SCG_ERR_INVALID, "scg_grad_buffers: null argument");
    *G = c->G_out; *n_k = c->nk_out;
    return SCG_OK;
}

int scg_set_grad_buffers(scg_ctx *c, float *G, int32_t *n_k) {
    if (!c) return fail(nullptr, SCG_ERR_INVALID, "scg_set_grad_buffers: null ctx");
    if ((G == nullptr) != (n_k == nullptr))
        return fail(c, SCG_ERR_INVALID, "scg_set_grad_buffers: pass both buffers or neither");
    c->G_out = G ? G : c->d_G;
    c->nk_out = n_k ? n_k : c->d_nk;
    c->nkf_out = nullptr;
    return SCG_OK;
}

int scg_set_grad_buffer_packed(scg_ctx *c, float *G_packed) {
    if (!c) return fail(nullptr, SCG_ERR_INVALID, "scg_set_grad_buffer_packed: null ctx");
    c->G_out = G_packed ? G_packed : c->d_G;
    c->nk_out = c->d_nk;
    c->nkf_out = G_packed ? G_packed + (size_t)c->n_vf * NACT * NF : nullptr;
    return SCG_OK;
}

int scg_apply_update_packed(scg_ctx *c, float *W, const float *G_packed, void *stream) {
    if (!c || !W || !G_packed) return fail(c, SCG_ERR_INVALID, "scg_apply_update_packed: null argument");
    if (!c->have_map) return fail(c, SCG_ERR_STATE, "scg_apply_update_packed: scg_set_map has not been called (scale table)");
    SCG_CHECK_ASYNC(c);
    SCG_ON_DEVICE(c, "scg_apply_update_packed");
    dim3 grid((NACT * NF + 255) / 256, c->n_vf);
    hipLaunchKernelGGL(apply_kernel, grid, dim3(256), 0, reinterpret_cast<hipStream_t>(stream), W, G_packed,
                       (const int32_t *)nullptr, G_packed + (size_t)c->n_vf * NACT * NF, c->d_scale, c->cfg.alpha);
    SCG_HIP(c, hipGetLastError());
    return SCG_OK;
}

int scg_apply_update(scg_ctx *c, float *W, const float *G, const int32_t *n_k, void *stream) {
    if (!c || !W || !G || !n_k) return fail(c, SCG_ERR_INVALID, "scg_apply_update: null argument");
    if (!c->have_map) return fail(c, SCG_ERR_STATE, "scg_apply_update: scg_set_map has not been called (scale table)");
    SCG_CHECK_ASYNC(c);
    SCG_ON_DEVICE(c, "scg_apply_update");
    dim3 grid((NACT * NF + 255) / 256, c->n_vf);
    hipLaunchKernelGGL(apply_kernel, grid, dim3(256), 0, reinterpret_cast<hipStream_t>(stream), W, G, n_k,
                       (const float *)nullptr, c->d_scale, c->cfg.alpha);
    SCG_HIP(c, hipGetLastError());
    return SCG_OK;
}

int scg_pinball_step(scg_ctx *c, int32_t n, float *x, float *y, float *vx, float *vy, const uint8_t *action,
                     float *reward, uint8_t *goal, void *stream) {
    if (!c) return fail(nullptr, SCG_ERR_INVALID, "scg_pinball_step: null ctx");
    if (!c->have_map) return fail(c, SCG_ERR_STATE, "scg_pinball_step: scg_set_map has not been called");
    if (n < 0 || !x || !y || !vx || !vy || !action || !reward || !goal)
        return fail(c, SCG_ERR_INVALID, "scg_pinball_step: bad argument");
    SCG_CHECK_ASYNC(c);
    SCG_ON_DEVICE(c, "scg_pinball_step");
    if (n == 0) return SCG_OK;
    hipLaunchKernelGGL(pinball_kernel, dim3((n + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       n, x, y, vx, vy, action, reward, goal, c->d_edges, c->d_cellmask, c->ms);
    SCG_HIP(c, hipGetLastError());
    return SCG_OK;
}

int scg_fourier_features(scg_ctx *c, int32_t n, const float *x, const float *y, const float *vx,
                         const float *vy, float *phi, void *stream) {
    if (!c) return fail(nullptr, SCG_ERR_INVALID, "scg_fourier_features: null ctx");
    if (n < 0 || !x || !y || !vx || !vy || !phi) return fail(c, SCG_ERR_INVALID, "scg_fourier_features: bad argument");
    SCG_CHECK_ASYNC(c);
    SCG_ON_DEVICE(c, "scg_fourier_features");
    if (n == 0) return SCG_OK;
    const int grid = n < 8192 ? n : 8192;
    hipLaunchKernelGGL(features_kernel, dim3(grid), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), n, x, y,
                       vx, vy, phi);
    SCG_HIP(c, hipGetLastError());
    return SCG_OK;
}

int scg_q_values(scg_ctx *c, int32_t n, const float *x, const float *y, const float *vx, const float *vy,
                 const float *Wk, float *q, void *stream) {
    if (!c) return fail(nullptr, SCG_ERR_INVALID, "scg_q_values: null ctx");
    if (n < 0 || !x || !y || !vx || !vy || !Wk || !q) return fail(c, SCG_ERR_INVALID, "scg_q_values: bad argument");
    SCG_CHECK_ASYNC(c);
    SCG_ON_DEVICE(c, "scg_q_values");
    if (n == 0) return SCG_OK;
    StepArgs A;
    fill_common(c, A);
    A.x = const_cast<float *>(x); A.y = const_cast<float *>(y);
    A.vx = const_cast<float *>(vx); A.vy = const_cast<float *>(vy);
    A.qcache = q; A.W = Wk; A.n = n; A.k_lo = 0; A.k_hi = 0; A.cnts = nullptr; A.learn = 0;
    hipLaunchKernelGGL(td_kernel<MODE_QVAL>, dim3((n + BLOCK_ENVS - 1) / BLOCK_ENVS), dim3(THREADS), LDS_BYTES,
                       reinterpret_cast<hipStream_t>(stream), A);
    SCG_HIP(c, hipGetLastError());
    return SCG_OK;
}

int scg_q_update(scg_ctx *c, int32_t n, int32_t k, const float *x, const float *y, const float *vx,
                 const float *vy, const uint8_t *action, const float *r, const float *cont, const float *xn,
                 const float *yn, const float *vxn, const float *vyn, float *W, uint32_t flags, void *stream) {
    if (!c) return fail(nullptr, SCG_ERR_INVALID, "scg_q_update: null ctx");
    if (!c->have_map) return fail(c, SCG_ERR_STATE, "scg_q_update: scg_set_map has not been called (scale table)");
    if (n < 0 || n > c->cfg.n_envs) return fail(c, SCG_ERR_INVALID, "scg_q_update: n must be in [0, n_envs]");
    if (k < 0 || k >= c->n_vf) return fail(c, SCG_ERR_INVALID, "scg_q_update: VF index out of range");
    if (!x || !y || !vx || !vy || !action || !r || !cont || !xn || !yn || !vxn || !vyn || !W)
        return fail(c, SCG_ERR_INVALID, "scg_q_update: null array argument");
    SCG_CHECK_ASYNC(c);
    SCG_ON_DEVICE(c, "scg_q_update");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int nblk = (n + BLOCK_ENVS - 1) / BLOCK_ENVS;
    SCG_HIP(c, hipMemsetAsync(c->d_cnts, 0, (size_t)c->nblk * c->n_vf * sizeof(int32_t), s));
    if (n > 0) {
        StepArgs A;
        fill_common(c, A);
        A.x = const_cast<float *>(x); A.y = const_cast<float *>(y);
        A.vx = const_cast<float *>(vx); A.vy = const_cast<float *>(vy);
        A.action = const_cast<uint8_t *>(action); A.reward = const_cast<float *>(r); A.cont_in = cont;
        A.xn = xn; A.yn = yn; A.vxn = vxn; A.vyn = vyn;
        A.W = W; A.n = n; A.k_lo = k; A.k_hi = k; A.learn = 1;
        hipLaunchKernelGGL(td_kernel<MODE_TRANS>, dim3(nblk), dim3(THREADS), LDS_BYTES, s, A);
        SCG_HIP(c, hipGetLastError());
    }
    return launch_reduce(c, W, (flags & SCG_STEP_APPLY) ? 1u : 0u, nblk, s);
}

int scg_classifier_predict(scg_ctx *c, int32_t n, const float *x, const float *y, const float *w8, uint8_t *out,
                           void *stream) {
    if (!c) return fail(nullptr, SCG_ERR_INVALID, "scg_classifier_predict: null ctx");
    if (n < 0 || !x || !y || !w8 || !out) return fail(c, SCG_ERR_INVALID, "scg_classifier_predict: bad argument");
    SCG_CHECK_ASYNC(c);
    SCG_ON_DEVICE(c, "scg_classifier_predict");
    if (n == 0) return SCG_OK;
    hipLaunchKernelGGL(predict_kernel, dim3((n + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       n, x, y, w8, out);
    SCG_HIP(c, hipGetLastError());
    return SCG_OK;
}

int scg_fit_initiation(scg_ctx *c, int32_t n_fit, const float *xy, const uint8_t *label, const int32_t *offsets,
                       float *w, int32_t iters, float lr, float l2, void *stream) {
    if (!c) return fail(nullptr, SCG_ERR_INVALID, "scg_fit_initiation: null ctx");
    if (n_fit < 0 || iters < 0 || !xy || !label || !offsets || !w)
        return fail(c, SCG_ERR_INVALID, "scg_fit_initiation: bad argument");
    SCG_CHECK_ASYNC(c);
    SCG_ON_DEVICE(c, "scg_fit_initiation");
    if (n_fit == 0 || iters == 0) return SCG_OK;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    for (int q0 = 0; q0 < n_fit; q0 += FIT_BATCH) {        // FIT_G workgroups per option, at most 64 in flight: co-resident
        const int nb = n_fit - q0 < FIT_BATCH ? n_fit - q0 : FIT_BATCH;
        SCG_HIP(c, hipMemsetAsync(c->d_fit_part, 0, (size_t)FIT_BATCH * 2 * FIT_G * 8 * sizeof(unsigned long long), s));   // tags of a past call
        hipLaunchKernelGGL(fit_kernel, dim3(FIT_G, nb), dim3(FIT_T), 0, s, xy, label, offsets, w, iters, lr, l2, q0,
                           c->d_fit_part, (unsigned long long)(c->fit_timeout_s * 1e8), c->d_async);
        SCG_HIP(c, hipGetLastError());
    }
    return SCG_OK;
}

}  // extern "C"
