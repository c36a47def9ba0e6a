// scg_kernels.hip — gfx950 kernels + the C-ABI of include/scg_abi.h.
//
// One step-batch (SPEC §5) = td_kernel<FUSED> -> reduce_kernel (+ sort_hist / sort_scatter when no env order is
// prepared). td_kernel (scg_step_kernel.hpp): a workgroup = 16 wavefronts owns 256 consecutive positions of the
// option-sorted env order and the whole LDS of its CU; <= 128 VGPRs, four waves per SIMD; the root value function and the
// block's option run as ONE merged pass on shared tables of Fourier factors.
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

template <typename T>
__device__ __forceinline__ void gstore(T *p, T v) { *p = v; }

// 16-byte WRITE-THROUGH store (sc1) for data that only the NEXT launch reads — the slabs: 27 MB per step-batch. A plain store
// leaves the lines dirty in the XCD's L2 and the kernel boundary then waits for their write-back (MI355X_MICROARCH.md, "boundary":
// + B / 6 TB/s); written through, they are in the memory-side cache by then, where the reduce launch (other XCDs) reads them
// anyway: +1.1 % env-steps/s. (`nt` stores, round 2, bypassed that cache too and cost the reduce launch 5.7 us.)
// The trailing s_nop is part of the instruction's contract here: gfx9 reads the data VGPRs of a store wider than 8 bytes a
// few cycles AFTER issue, and a vector write to one of them in the next two wait states corrupts the stored value. hipcc's
// hazard recognizer pads its own stores; inline asm is opaque to it (round 4: the compiler reused v[26:27] of the data for the
// next address right behind the store and 4 x 8 entries of a tile came out as address bits — tools/debug_u2.py found it).
__device__ __forceinline__ void store_wt(float *p, f4v v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 2" :: "v"(p), "v"(v) : "memory");
}

constexpr int OREC = 4;        // float4 per result line (64 bytes): [0] state', [1] {reward, bits, counters}, [2..3] Q(s', .) of the VF acting next

struct StepArgs {
    // env state (FUSED: in/out; TRANS/QVAL: in)
    float *x, *y, *vx, *vy;
    int32_t *option_id, *opt_steps, *ep_steps;
    int32_t *hist_next;        // [rows of 256 envs][HSTRIDE] counts of the SORT KEYS (sort_key) this step leaves (null = off)
    float4 *qalt;              // FUSED: [positions][2] SPEC §4.2: the root's Q(s', .) of an env about to enter an option (read by commit_row if the env is declined)
    float4 *outrec;            // FUSED: [positions][OREC] per-env results in env-ORDER position (one 64-byte line:
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
    uint32_t *async_word;          // host-visible sticky status word (a hand-off poll that runs out is reported there)
    int32_t *fail_flag;            // ... and its device-side twin: the reduce launch of the same step reads it (no apply, no commit)
    int32_t n, n_vf, k_lo, k_hi;
    uint32_t enabled, learn;       // learn: 1 = learning step
    uint32_t gest;                 // SPEC §4.4: options in gestation (classifier known, not selectable, learning off-policy)
    int32_t *gest_succ;            // [n_vf] successes seen from inside a gestating option's initiation set (atomic counts)
    uint32_t parents;              // 3 bits per option k at [3k, 3k+3): target option of k (0 = the task goal)
    uint64_t t, seed;
    int64_t env_base;
    float gamma, epsilon, r_succ;
    int32_t max_ep, max_opt;
    uint32_t reoffer_mask;         // SPEC §4.2: reoffer_period - 1 (0: an option is offered every step)
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

#include "scg_step_kernel.hpp"

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
    int32_t nk_floor;              // SPEC §5 apply: divisor max(n_k, nk_floor)
    // next step's env order (SPEC §5) as extra workgroups (option_id null = off): the fused kernel has counted
    // the new option ids per row of 256 envs into `hist`; `hist_zero` is the other buffer, cleared for the next step
    const int32_t *option_id;
    int32_t *hist, *hist_zero, *perm;
    int32_t n, nrow;
    // commit of the fused kernel's per-position results to the caller's arrays (outrec null = off), one row of
    // 256 envs per wave; with `sort` the same wave then places its row in the next env order
    const float4 *outrec;
    const float4 *qalt;            // the root's Q(s', .) of the envs whose result line carries the declined mark (SPEC §4.2)
    int32_t *invperm;              // [n] position of env e in the current order (in: this step's, out: the next's)
    float *x, *y, *vx, *vy, *reward;
    int32_t *option_id_out, *opt_steps, *ep_steps;
    uint8_t *action, *done;
    float *qcache;                 // [5][n], null = the step ran no TD pass (diagnostic): leave it alone
    int32_t sort;
    const int32_t *fail_flag;      // set by a workgroup of the step kernel that gave up: the step is void (no apply, no commit)
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
// SPEC §5 sort key of an env from its signed option id: k in [1, n_vf) (running option k) -> k; 0 and -k (no option in sight / inside
// option k's initiation set but staying out of it, §4.2: either way the env runs the root) -> 0; anything else -> the last key n_vf.
// NKEY keys, count tables with HSTRIDE ints per row.
constexpr int NKEY = 7, HSTRIDE = 8;
__device__ __forceinline__ int sort_key(int o, int n_vf) {
    if (o <= 0) return o > -n_vf ? 0 : n_vf;
    return o < n_vf ? o : n_vf;
}
struct OrderLayout {
    int chunked, c, g, U, Ftot;
    uint32_t mc, mg;               // ceil(2^32 / c), ceil(2^32 / g): exact division of ranks (< 2^24) by mul-high
    int start[NKEY];               // position of run k's first env
    int cnt[NKEY], n[NKEY], F[NKEY];   // chunked: workgroups of run k, its size, key-0 fill slots before it
    int pad_lo[NKEY], pad_n[NKEY], pad_pos[NKEY], tail_lo, tail_pos;     // padded layout
};
__device__ __forceinline__ void order_layout(const int tot[NKEY], int n_envs, OrderLayout &L) {
    int S = 0, Rn = 0;
#pragma unroll
    for (int k = 1; k < NKEY; ++k) { S += tot[k]; Rn += tot[k] > 0 ? 1 : 0; L.n[k] = tot[k]; }
    L.n[0] = tot[0];
    const int Bf = n_envs / BLOCK_ENVS;
    int c = BLOCK_ENVS;
    if (Bf > Rn && S > 0) c = min(BLOCK_ENVS, (S + (Bf - Rn) - 1) / (Bf - Rn));
    int U = 0, F = 0;
    L.start[0] = 0; L.cnt[0] = 0; L.F[0] = 0;
    L.mc = div_magic((uint32_t)c);
    L.mg = div_magic((uint32_t)max(BLOCK_ENVS - c, 1));
#pragma unroll
    for (int k = 1; k < NKEY; ++k) {
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
        for (int k = 1; k < NKEY; ++k) {
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
        for (int k = 1; k < NKEY; ++k)
            if (r >= L.pad_lo[k] && r < L.pad_lo[k] + L.pad_n[k]) pos = L.pad_pos[k] + (r - L.pad_lo[k]);
        return pos;
    }
    int pos = L.U * BLOCK_ENVS + (r - L.Ftot);          // behind all runs
    int st = 0, cn = 0, nk = 0, f0 = 0;
    bool in_run = false;
#pragma unroll
    for (int k = 1; k < NKEY; ++k) {
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
__device__ __forceinline__ void commit_and_place_row(const ReduceArgs &R, int row, int wv, int lane, int (*s_x)[40]) {
    // a workgroup of this step gave up (uniform): the caller's arrays and the env order keep what the previous step left. The flag is
    // FETCHED here and looked at where the first result would be written: a test up front put one more dependent round trip in front
    // of everything the row does (+1.9 us per step-batch)
    // (read through a per-lane zero offset: as a wave-uniform load the compiler turns it into a scalar at once — global_load, s_waitcnt
    //  vmcnt(0), v_readfirstlane — which is the up-front test again)
    int zoff = 0;
    asm volatile("" : "+v"(zoff));
    const int step_failed = R.fail_flag[zoff];              // (never null: the context's flag word)
    const bool act = wv < 4 && row < R.nrow;
    const int e = row * 256 + wv * 64 + lane;
    const bool ok = act && e < R.n;
    const int pos_old = ok ? R.invperm[e] : 0;
    int tot[NKEY], pre[NKEY];
#pragma unroll
    for (int k = 0; k < NKEY; ++k) { tot[k] = 0; pre[k] = 0; }
    if (act && R.sort) {
        for (int r = wv * 64 + lane; r < R.nrow; r += 256) {       // this wave's quarter of the count table
#pragma unroll
            for (int k = 0; k < NKEY; ++k) {
                const int h = R.hist[r * HSTRIDE + k];
                tot[k] += h;
                if (r < row) pre[k] += h;
            }
        }
    }
    float4 ra = make_float4(0.0f, 0.0f, 0.0f, 0.0f), rb = ra, rq = ra;
    float q4 = 0.0f;
    if (ok) {
        const float4 *r = R.outrec + (size_t)pos_old * OREC;
        const float4 r3 = r[3];
        ra = r[0]; rb = r[1]; rq = r[2]; q4 = r3.x;
        if (__float_as_uint(r3.y) == OREC_DECLINED) {       // SPEC §4.2: the option promised less than the root — the env stays with the root
            rq = R.qalt[(size_t)pos_old * 2]; q4 = R.qalt[(size_t)pos_old * 2 + 1].x;
            rb.y = __uint_as_float(__float_as_uint(rb.y) | 0x01000000u);      // declined: bit 24 of the record's bits
        }
    }
    int key = -1;
    uint64_t km[NKEY];
    if (act) {
        if (R.sort) {
#pragma unroll
            for (int k = 0; k < NKEY; ++k) {                   // integer sums: any order
#pragma unroll
                for (int m = 1; m < 64; m <<= 1) { tot[k] += __shfl_xor(tot[k], m, 64); pre[k] += __shfl_xor(pre[k], m, 64); }
            }
        }
        const unsigned bits = __float_as_uint(rb.y);
        if (ok && !step_failed) {
            const int on = (int)((bits >> 16) & 255u);
            const bool declined = (bits >> 24) & 1u;            // set above from the result line's mark (SPEC §4.2): declined in this step
            // the caller sees -k: inside option k's initiation set, staying out of it (bits 28..30: it has been since an earlier step)
            const int oid = declined ? -on : (on ? on : -(int)((bits >> 28) & 7u));
            key = sort_key(oid, R.n_vf);
            R.x[e] = ra.x; R.y[e] = ra.y; R.vx[e] = ra.z; R.vy[e] = ra.w;
            R.reward[e] = rb.x; R.action[e] = (uint8_t)(bits & 255u); R.done[e] = (uint8_t)((bits >> 8) & 255u);
            R.option_id_out[e] = oid; R.opt_steps[e] = __float_as_int(rb.z); R.ep_steps[e] = __float_as_int(rb.w);
            if (R.qcache) {
                const size_t n = (size_t)R.n;
                R.qcache[e] = rq.x; R.qcache[n + e] = rq.y; R.qcache[2 * n + e] = rq.z;
                R.qcache[3 * n + e] = rq.w; R.qcache[4 * n + e] = q4;
            }
        }
        if (R.sort) {
#pragma unroll
            for (int k = 0; k < NKEY; ++k) km[k] = __ballot(key == k);
            if (lane < 3 * NKEY) {                          // [0..NKEY) table totals, [NKEY..2 NKEY) rows before this one, [2 NKEY..3 NKEY) this wave's keys
                int v = 0;
#pragma unroll
                for (int k = 0; k < NKEY; ++k) {
                    if (lane == k) v = tot[k];
                    if (lane == NKEY + k) v = pre[k];
                    if (lane == 2 * NKEY + k) v = __popcll(km[k]);
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
        if (lane == 0) s_x[wv][3 * NKEY] = v;
    }
    if (!R.sort && !R.c_rows) return;                       // workgroup-uniform
    __syncthreads();
    if (R.c_rows && wv == 0 && lane == 0 && row < R.nrow) {
        R.c_rows[row] = s_x[0][3 * NKEY] + s_x[1][3 * NKEY] + s_x[2][3 * NKEY] + s_x[3][3 * NKEY];
        if (row == 0) R.c_rows[R.nrow] = *R.c_count;        // the buffer's fill level
    }
    if (!R.sort || !act) return;
    int off[NKEY];
#pragma unroll
    for (int k = 0; k < NKEY; ++k) {
        tot[k] = s_x[0][k] + s_x[1][k] + s_x[2][k] + s_x[3][k];
        off[k] = s_x[0][NKEY + k] + s_x[1][NKEY + k] + s_x[2][NKEY + k] + s_x[3][NKEY + k];
#pragma unroll
        for (int w = 0; w < 3; ++w) off[k] += w < wv ? s_x[w][2 * NKEY + k] : 0;      // rank of this wave's first key-k env within its run
    }
    OrderLayout L;
    order_layout(tot, R.n, L);
    int rk = -1, st = 0;
#pragma unroll
    for (int k = 0; k < NKEY; ++k)
        if (key == k) { rk = off[k] + __popcll(km[k] & ((1ull << lane) - 1ull)); st = L.start[k]; }
    if (rk >= 0 && !step_failed) {
        const int pos = key == 0 ? order_pos0(L, rk) : order_posk(L, st, rk);
        R.perm[pos] = e; R.invperm[e] = pos;
    }
    if (wv == 0 && lane < HSTRIDE) R.hist_zero[row * HSTRIDE + lane] = 0;
}

// grid (column chunks, n_vf [+ rows of the env order]). A workgroup owns 64 float4 columns of one value function;
// its 16 waves each sum one segment's slabs, T_s = ((P_16s + P_16s+1) + ...) over the non-empty blocks with all
// 16 loads in flight, park T_s in LDS, and wave 0 adds the non-empty segments in order, G = ((T_0 + T_1) + ...)
// — SPEC §5's two levels in one launch.
// BATCH = slab loads in flight per wave: 16 (101 VGPRs, one 16-wave workgroup per CU: one memory round trip per segment — for launches
// whose workgroups fit the chip in one round anyway) or 8 (64 VGPRs, two workgroups per CU: the bench size's 448 workgroups are resident
// together instead of in 1.75 rounds). The sum runs in block order either way: same bits.
template <int BATCH>
__global__ __launch_bounds__(RED_THREADS, BATCH == 16 ? 4 : 8) void reduce_kernel(const ReduceArgs R) {
    __shared__ float4 s_T[RED_WAVES * RED_SPW][64];
    __shared__ int s_cnt[RED_WAVES * RED_SPW];
    __shared__ int s_x[4][40];         // the commit rows' exchange area
    // leading workgroups (blockIdx.y < gridDim.y - n_vf): one env row each — commit + next order
    const int sy_rows = (int)gridDim.y - R.n_vf;             // the commit rows come FIRST in dispatch order (theirs is the longer chain)
    const int k = (int)blockIdx.y >= sy_rows ? (int)blockIdx.y - sy_rows : -1;
    const int rowy = (int)blockIdx.y;
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
    // wave 0 applies the update at the end: its W and scale columns are fetched under the last round's barrier
    float4 w_old = make_float4(0.0f, 0.0f, 0.0f, 0.0f), sc = w_old;
    int step_failed = 0;
    // A round = RED_SPW segments per wave (32 segments = 512 blocks in all at RED_SPW = 2: the bench size in ONE round): the
    // counts of all of a wave's segments are read first, then segment after segment its <= 16 slabs with all loads in flight,
    // and one barrier pair per round (round 2: a round was one segment per wave — two dependent count -> slab round trips and
    // two barrier pairs at the bench size)
    for (int sg0 = 0; sg0 < nseg; sg0 += RED_WAVES * RED_SPW) {
        int cs[RED_SPW];
#pragma unroll
        for (int j = 0; j < RED_SPW; ++j) {
            const int bl = (sg0 + j * RED_WAVES + wave) * SEG + lane;
            cs[j] = (lane < SEG && bl < R.nblk) ? R.cnts[(unsigned)(bl * R.n_vf + k)] : 0;      // (32-bit index: nblk * n_vf is small; the 64-bit form was hoisted and spilled)
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
#pragma unroll
                for (int h = 0; h < SEG; h += BATCH) {
                    if (!((mask >> h) & ((1u << BATCH) - 1u))) continue;
                    u4v v[BATCH];
#pragma unroll
                    for (int u = 0; u < BATCH; ++u) {
                        v[u] = (u4v){0u, 0u, 0u, 0u};
                        if ((mask >> (h + u)) & 1u) v[u] = __builtin_amdgcn_raw_buffer_load_b128(seg, (int)col_off, (int)((h + u) * (unsigned)slab_stride), 0);
                    }
#pragma unroll
                    for (int u = 0; u < BATCH; ++u) {
                        if ((mask >> (h + u)) & 1u) {
                            T.x = T.x + __uint_as_float(v[u][0]); T.y = T.y + __uint_as_float(v[u][1]);
                            T.z = T.z + __uint_as_float(v[u][2]); T.w = T.w + __uint_as_float(v[u][3]);
                        }
                    }
                }
            }
            s_T[j * RED_WAVES + wave][lane] = T;
            if (lane == 0) s_cnt[j * RED_WAVES + wave] = c;
        }
        if (wave == 0 && R.apply && sg0 + RED_WAVES * RED_SPW >= nseg) {      // last round: under the barrier and the second-level sum
            step_failed = *R.fail_flag;                         // (fetched with the weights; looked at where they would be written)
            int col = live ? i4 : 0;
            asm volatile("" : "+v"(col));                      // (addresses made HERE: hoisted to the top of the kernel they are spilled too)
            w_old = reinterpret_cast<const float4 *>(R.W)[(size_t)k * RED_COLS + col];      // (not at the top of the kernel: held across the slab loads they were
            sc = *reinterpret_cast<const float4 *>(R.scale + (col * 4) % NF);     //  eight more registers — spilled at 64 VGPRs; NF % 4 == 0: no row straddling)
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
    if (R.apply && nk > 0 && !step_failed) {       // (a step in which a workgroup gave up leaves W as it was)
        const float step = R.alpha / (float)max(nk, R.nk_floor);
        float4 w = w_old;
        w.x = fmaf(step * sc.x, S.x, w.x); w.y = fmaf(step * sc.y, S.y, w.y);
        w.z = fmaf(step * sc.z, S.z, w.z); w.w = fmaf(step * sc.w, S.w, w.w);
        int col = i4;
        asm volatile("" : "+v"(col));
        reinterpret_cast<float4 *>(R.W)[(size_t)k * RED_COLS + col] = w;
    }
}

// acting-only steps have no reduce launch: the commit alone, one workgroup of four waves per row of 256 envs
__global__ __launch_bounds__(256) void commit_kernel(const ReduceArgs R) {
    __shared__ int s_x[4][40];
    commit_and_place_row(R, blockIdx.x, threadIdx.x >> 6, threadIdx.x & 63, s_x);
}

__global__ __launch_bounds__(256) void apply_kernel(float *W, const float *G, const int32_t *n_k, const float *nk_f,
                                                    const float *scale, float alpha, int nk_floor) {
    const int k = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= NACT * NF) return;
    const int nk = n_k ? n_k[k] : (int)(nk_f[k] + 0.5f);     // packed operand: counts summed as floats (exact)
    if (nk <= 0) return;
    const float step = alpha / (float)max(nk, nk_floor);
    const int f = i % NF;
    float *w = W + (size_t)k * NACT * NF + i;
    *w = fmaf(step * scale[f], G[(size_t)k * NACT * NF + i], *w);
}

// The order-pinned multi-rank form (SPEC §5): G and the counts are the sums of the ranks' packed operands in slot order.
__global__ __launch_bounds__(256) void apply_slots_kernel(float *W, const float *slots, int n_slots, long stride, int n_vf,
                                                          const float *scale, float alpha, int nk_floor) {
    const int k = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= NACT * NF) return;
    const size_t at = (size_t)k * NACT * NF + i, cnt_at = (size_t)n_vf * NACT * NF + k;
    float g = slots[at], nkf = slots[cnt_at];
    for (int r = 1; r < n_slots; ++r) {
        g = g + slots[(size_t)r * stride + at];
        nkf = nkf + slots[(size_t)r * stride + cnt_at];          // counts as floats: exact (far below 2^24)
    }
    const int nk = (int)(nkf + 0.5f);
    if (nk <= 0) return;
    const float step = alpha / (float)max(nk, nk_floor);
    float *w = W + at;
    *w = fmaf(step * scale[i % NF], g, *w);
}

// ------------------------------------------------------------------------------------------------
// SPEC §5 env order: stable counting sort of the envs by option_id (6 keys), two tiny kernels per step.
// Option-homogeneous workgroups turn five sparse option passes per workgroup into about one dense one.
__global__ __launch_bounds__(256) void sort_hist_kernel(const int32_t *option_id, int n, int n_vf, int32_t *hist) {
    __shared__ int s_c[4][HSTRIDE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int e = blockIdx.x * 256 + tid;
    const int o = e < n ? sort_key(option_id[e], n_vf) : -1;          // (out-of-range ids sort last)
#pragma unroll
    for (int k = 0; k < NKEY; ++k) {
        const uint64_t m = __ballot(o == k);
        if (lane == 0) s_c[wave][k] = __popcll(m);
    }
    __syncthreads();
    if (tid < NKEY) hist[blockIdx.x * HSTRIDE + tid] = s_c[0][tid] + s_c[1][tid] + s_c[2][tid] + s_c[3][tid];
}

__global__ __launch_bounds__(256) void sort_scatter_kernel(const int32_t *option_id, int n, int n_vf, int nblk,
                                                           const int32_t *hist, int32_t *perm, int32_t *invperm) {
    __shared__ int s_c[4][HSTRIDE];
    __shared__ int s_off[HSTRIDE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.x;
    // offset of (key k, block b) in the sorted order = (all envs with a smaller key) + (key-k envs of earlier blocks)
    __shared__ int s_tot[HSTRIDE], s_pre[HSTRIDE], s_part[4][2 * HSTRIDE];
    int tot[NKEY], pre[NKEY];
#pragma unroll
    for (int kk = 0; kk < NKEY; ++kk) { tot[kk] = 0; pre[kk] = 0; }
    for (int bb0 = 0; bb0 < nblk; bb0 += 256) {
        const int bb = bb0 + tid;
        if (bb < nblk) {
#pragma unroll
            for (int kk = 0; kk < NKEY; ++kk) {
                const int h = hist[bb * HSTRIDE + kk];
                tot[kk] += h;
                if (bb < b) pre[kk] += h;
            }
        }
    }
#pragma unroll
    for (int kk = 0; kk < NKEY; ++kk) {                       // integer sums: any order
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) { tot[kk] += __shfl_xor(tot[kk], m, 64); pre[kk] += __shfl_xor(pre[kk], m, 64); }
        if (lane == 0) { s_part[wave][kk] = tot[kk]; s_part[wave][HSTRIDE + kk] = pre[kk]; }
    }
    __syncthreads();
    if (tid < NKEY) {
        s_tot[tid] = s_part[0][tid] + s_part[1][tid] + s_part[2][tid] + s_part[3][tid];
        s_pre[tid] = s_part[0][HSTRIDE + tid] + s_part[1][HSTRIDE + tid] + s_part[2][HSTRIDE + tid] + s_part[3][HSTRIDE + tid];
    }
    __syncthreads();
    int tt[NKEY];
#pragma unroll
    for (int kk = 0; kk < NKEY; ++kk) tt[kk] = s_tot[kk];
    OrderLayout L;
    order_layout(tt, n, L);
    if (tid < NKEY) s_off[tid] = s_pre[tid];               // rank of the row's first key-k env within its run
    const int e = b * 256 + tid;
    const int o = e < n ? sort_key(option_id[e], n_vf) : -1;
    int rank = 0;
#pragma unroll
    for (int k = 0; k < NKEY; ++k) {
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
        for (int k = 1; k < NKEY; ++k) if (o == k) st = L.start[k];
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
                // SCG_ASYNC_FIT_TIMEOUT raised in the ctx's host-visible status word (scg_async_status) by the
                // problem's workgroup 0, the only one that writes the row.
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
    if (s_abort) {                                        // no silent NaN row: w keeps its old value, the host is told.
        // Workgroup 0 of the problem alone decides: it is the one that writes the row, so "status bit raised" and "row left
        // untouched" are the same event. A partner that gives up merely exits (workgroup 0 then either holds everything it
        // needs — the partner had published its last partial — and finishes exactly, or runs out of patience itself).
        if (j == 0 && tid == 0 && async_word)
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
    int n_cu;                      // compute units of the device (picks the reduce launch's form)
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
    int32_t *d_fail;               // device-side twin of the step's give-up bit (read by the reduce launch of the same step)
    double fit_timeout_s;          // how long fit_kernel waits for a workgroup that is not running yet
    float4 *d_outrec;              // [nblk * BLOCK_ENVS][OREC] per-position step results (td_kernel -> commit_row)
    float4 *d_qalt;                // [nblk * BLOCK_ENVS][2] the root's Q(s', .) of envs about to enter an option (SPEC §4.2)
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
        if (word & SCG_ASYNC_STEP_HANDOFF)
            snprintf(buf, n, "an earlier scg_step gave up inside a workgroup: a bounded hand-off poll between its wavefront subsets ran "
                     "out (a logic error or a hung wavefront); the block's partial gradients were dropped and the step's outputs for "
                     "its envs are unspecified — restore the state. scg_clear_async_error() re-arms the context");
        else if (word & SCG_ASYNC_FIT_TIMEOUT)
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
    // a step that gave up was voided on the device (no apply, no commit): whatever it left half-made is dropped here
    if (c->d_fail) {
        DeviceGuard g(c->cfg.device);
        if (g.ok) (void)hipMemset(c->d_fail, 0, sizeof(int32_t));
    }
    c->order_valid = false; c->hist_dirty = true; c->arm_rows_ready = false;
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
    if (cfg->update_count_floor < 0) return fail(nullptr, SCG_ERR_INVALID, "scg_create: update_count_floor must be >= 0");
    if (cfg->reoffer_period < 0 || (cfg->reoffer_period & (cfg->reoffer_period - 1)))
        return fail(nullptr, SCG_ERR_INVALID, "scg_create: reoffer_period must be a power of two (or 0)");
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
    if (hipDeviceGetAttribute(&c->n_cu, hipDeviceAttributeMultiprocessorCount, cfg->device) != hipSuccess || c->n_cu <= 0) c->n_cu = 256;
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
        if (hipMalloc(&c->d_hist, (size_t)((c->cfg.n_envs + 255) / 256) * HSTRIDE * sizeof(int32_t)) != hipSuccess) { st = SCG_ERR_HIP; break; }
        if (hipMalloc(&c->d_collect_rows, (size_t)((cfg->n_envs + COL_ROW - 1) / COL_ROW + 1) * sizeof(int32_t)) != hipSuccess) { st = SCG_ERR_HIP; break; }
        if (hipMalloc(&c->d_fit_part, (size_t)FIT_BATCH * 2 * FIT_G * 8 * sizeof(unsigned long long)) != hipSuccess) { st = SCG_ERR_HIP; break; }
        if (hipHostMalloc(reinterpret_cast<void **>(&c->h_async), 64, hipHostMallocMapped) != hipSuccess) { st = SCG_ERR_HIP; break; }
        *c->h_async = 0u;
        if (hipHostGetDevicePointer(reinterpret_cast<void **>(&c->d_async), c->h_async, 0) != hipSuccess) { st = SCG_ERR_HIP; break; }
        if (hipMalloc(&c->d_fail, sizeof(int32_t)) != hipSuccess || hipMemset(c->d_fail, 0, sizeof(int32_t)) != hipSuccess) { st = SCG_ERR_HIP; break; }
        if (hipMalloc(&c->d_outrec, (size_t)c->nblk * BLOCK_ENVS * OREC * sizeof(float4)) != hipSuccess) { st = SCG_ERR_HIP; break; }
        if (hipMalloc(&c->d_qalt, (size_t)c->nblk * BLOCK_ENVS * 2 * sizeof(float4)) != hipSuccess) { st = SCG_ERR_HIP; break; }
        if (hipMalloc(&c->d_invperm, (size_t)c->nblk * BLOCK_ENVS * sizeof(int32_t)) != hipSuccess) { st = SCG_ERR_HIP; break; }
        {
            const size_t hb = (size_t)((c->cfg.n_envs + 255) / 256) * HSTRIDE * sizeof(int32_t);
            if (hipMalloc(&c->d_hist2[0], hb) != hipSuccess || hipMalloc(&c->d_hist2[1], hb) != hipSuccess) { st = SCG_ERR_HIP; break; }
            if (hipMemset(c->d_hist2[0], 0, hb) != hipSuccess || hipMemset(c->d_hist2[1], 0, hb) != hipSuccess) { st = SCG_ERR_HIP; break; }
        }
        if (hipMalloc(&c->d_cellmask, (size_t)CELL_G * CELL_G * 4 * sizeof(uint64_t)) != hipSuccess) { st = SCG_ERR_HIP; break; }
        if (hipMemset(c->d_cnts, 0, (size_t)c->nblk * c->n_vf * sizeof(int32_t)) != hipSuccess) { st = SCG_ERR_HIP; break; }
        if (hipMemset(c->d_G, 0, (size_t)c->n_vf * NACT * NF * sizeof(float)) != hipSuccess) { st = SCG_ERR_HIP; break; }
        if (hipMemset(c->d_nk, 0, MAX_VF * sizeof(int32_t)) != hipSuccess) { st = SCG_ERR_HIP; break; }
        // (the step kernel's LDS — 160 KB, one workgroup per CU — is static: no dynamic-LDS attribute to raise)
    } while (0);
    if (st != SCG_OK) {
        snprintf(g_err, 256, "scg_create: device allocation/setup failed: %s", hipGetErrorString(hipGetLastError()));
        scg_destroy(c);
        return st;
    }
#if defined(SCG_STAMPS) || defined(SCG_STAMPS_LITE)
    if (hipMalloc(&c->d_stamps, (size_t)c->nblk * STAMP_SLOTS * sizeof(unsigned long long)) == hipSuccess)
        (void)hipMemset(c->d_stamps, 0, (size_t)c->nblk * STAMP_SLOTS * sizeof(unsigned long long));
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
    (void)hipFree(c->d_hist2[0]); (void)hipFree(c->d_hist2[1]); (void)hipFree(c->d_outrec); (void)hipFree(c->d_qalt); (void)hipFree(c->d_invperm);
    (void)hipFree(c->d_edges); (void)hipFree(c->d_starts); (void)hipFree(c->d_scale); (void)hipFree(c->d_cellmask); (void)hipFree(c->d_perm); (void)hipFree(c->d_hist);
    (void)hipFree(c->d_fit_part); (void)hipFree(c->d_collect_rows); (void)hipFree(c->d_fail);
    if (c->h_async) (void)hipHostFree(c->h_async);
    if (c->prof_ev) {
        for (hipEvent_t e : *c->prof_ev) (void)hipEventDestroy(e);
        delete c->prof_ev;
    }
    delete c;
    return SCG_OK;
}

int scg_set_hparams(scg_ctx *c, float gamma, float alpha, float epsilon, float r_option_success,
                    int32_t max_episode_steps, int32_t max_option_steps, int32_t update_count_floor, int32_t reoffer_period) {
    if (!c) return fail(nullptr, SCG_ERR_INVALID, "scg_set_hparams: null ctx");
    c->cfg.gamma = gamma; c->cfg.alpha = alpha; c->cfg.epsilon = epsilon;
    c->cfg.r_option_success = r_option_success;
    c->cfg.max_episode_steps = max_episode_steps; c->cfg.max_option_steps = max_option_steps;
    c->cfg.update_count_floor = update_count_floor < 0 ? 0 : update_count_floor;
    if (reoffer_period < 0 || (reoffer_period & (reoffer_period - 1))) return fail(c, SCG_ERR_INVALID, "scg_set_hparams: reoffer_period must be a power of two (or 0)");
    c->cfg.reoffer_period = reoffer_period;
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
    A.reoffer_mask = c->cfg.reoffer_period > 1 ? (uint32_t)(c->cfg.reoffer_period - 1) : 0u;
    A.ms = c->ms;
    A.edges = c->d_edges; A.starts = c->d_starts; A.cellmask = c->d_cellmask;
    A.slabs = c->d_slabs; A.cnts = c->d_cnts;
    A.parents = c->parents;
    A.gest = c->gest; A.gest_succ = c->gest_succ;
    A.ring_x = c->ring_x; A.ring_y = c->ring_y; A.events = c->events; A.ev_len = c->ev_len;
    A.ring_mask = c->ring_len > 0 ? c->ring_len - 1 : 0;
    A.stamps = c->d_stamps;
    A.async_word = c->d_async; A.fail_flag = c->d_fail;
}

// The reduce launch; for the fused step (`st` given) its extra workgroups also commit the step's per-position
// results to the caller's arrays and, with `sort`, place every row in the next step's env order.
static int launch_reduce(scg_ctx *c, float *W, uint32_t apply, int nblk, hipStream_t s,
                         const StepArgs *st = nullptr, bool sort = false, bool reduce = true) {
    ReduceArgs R;
    memset(&R, 0, sizeof(R));
    R.slabs = c->d_slabs; R.cnts = c->d_cnts; R.G = c->G_out; R.n_k = c->nk_out; R.nk_f = c->nkf_out; R.W = W; R.scale = c->d_scale;
    R.nblk = nblk; R.n_vf = c->n_vf; R.alpha = c->cfg.alpha; R.apply = apply; R.fail_flag = c->d_fail; R.nk_floor = c->cfg.update_count_floor;
    const int nrow = st ? (c->cfg.n_envs + 255) / 256 : 0;
    R.n = c->cfg.n_envs; R.nrow = nrow;
    if (st) {
        R.outrec = c->d_outrec; R.qalt = c->d_qalt; R.invperm = c->d_invperm; R.perm = c->d_perm; R.sort = sort ? 1 : 0;
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
    if (RED_NCOL * (c->n_vf + sy) <= c->n_cu)           // fits the chip at one workgroup per CU: all sixteen slabs of a segment in flight
        hipLaunchKernelGGL(reduce_kernel<16>, dim3(RED_NCOL, c->n_vf + sy), dim3(RED_THREADS), 0, s, R);
    else
        hipLaunchKernelGGL(reduce_kernel<8>, dim3(RED_NCOL, c->n_vf + sy), dim3(RED_THREADS), 0, s, R);
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
    if (c->arm_bits) {
        // An announced trigger (scg_arm_collect) makes this step's commit rows read the caller's prev_in / count buffers:
        // refuse to launch if they are no longer device allocations (freed since the announcement) instead of faulting
        // the GPU. Two host-side attribute queries per step, only while a trigger is armed.
        const void *ptrs[2] = {c->arm_count, c->arm_prev};
        for (const void *p : ptrs) {
            if (!p) continue;
            hipPointerAttribute_t at;
            if (hipPointerGetAttributes(&at, p) != hipSuccess || at.type != hipMemoryTypeDevice) {
                (void)hipGetLastError();
                c->arm_bits = 0; c->arm_rows_ready = false;
                return fail(c, SCG_ERR_STATE, "scg_step: the buffers announced with scg_arm_collect are no longer device memory "
                                              "(freed before scg_arm_collect(0)?); the trigger has been disarmed");
            }
        }
    }
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
        const size_t hb = (size_t)((c->cfg.n_envs + 255) / 256) * HSTRIDE * sizeof(int32_t);
        SCG_HIP(c, hipMemsetAsync(c->d_hist2[0], 0, hb, s));
        SCG_HIP(c, hipMemsetAsync(c->d_hist2[1], 0, hb, s));
        c->hist_dirty = false;
    }
    const bool fold = (flags & SCG_STEP_LEARN) && !(flags & 0x200u);      // 0x200: diagnostic, sort afresh every step
    A.hist_next = fold ? c->d_hist2[c->hist_parity] : nullptr;
    A.perm = c->d_perm; A.outrec = c->d_outrec; A.qalt = c->d_qalt;
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
    hipLaunchKernelGGL(td_kernel<MODE_FUSED>, dim3(c->nblk), dim3(THREADS), 0, s, A);
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

#if defined(SCG_STAMPS) || defined(SCG_STAMPS_LITE)
extern "C" int scg_diag_stamps(scg_ctx *c, unsigned long long *host_out /*[nblk][16]*/, int32_t reset) {
    if (!c || !c->d_stamps) return SCG_ERR_STATE;
    if (host_out && hipMemcpy(host_out, c->d_stamps, (size_t)c->nblk * STAMP_SLOTS * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return SCG_ERR_HIP;
    if (reset) (void)hipMemset(c->d_stamps, 0, (size_t)c->nblk * STAMP_SLOTS * sizeof(unsigned long long));
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
                       (const int32_t *)nullptr, G_packed + (size_t)c->n_vf * NACT * NF, c->d_scale, c->cfg.alpha, c->cfg.update_count_floor);
    SCG_HIP(c, hipGetLastError());
    return SCG_OK;
}

int scg_apply_update_slots(scg_ctx *c, float *W, const float *slots, int32_t n_slots, int64_t slot_stride, void *stream) {
    if (!c || !W || !slots) return fail(c, SCG_ERR_INVALID, "scg_apply_update_slots: null argument");
    if (n_slots < 1 || n_slots > 4096 || slot_stride < (int64_t)c->n_vf * NACT * NF + c->n_vf)
        return fail(c, SCG_ERR_INVALID, "scg_apply_update_slots: n_slots must be in [1, 4096] and slot_stride >= n_vf * 6480 + n_vf floats");
    if (!c->have_map) return fail(c, SCG_ERR_STATE, "scg_apply_update_slots: scg_set_map has not been called (scale table)");
    SCG_CHECK_ASYNC(c);
    SCG_ON_DEVICE(c, "scg_apply_update_slots");
    dim3 grid((NACT * NF + 255) / 256, c->n_vf);
    hipLaunchKernelGGL(apply_slots_kernel, grid, dim3(256), 0, reinterpret_cast<hipStream_t>(stream), W, slots, (int)n_slots,
                       (long)slot_stride, c->n_vf, c->d_scale, c->cfg.alpha, c->cfg.update_count_floor);
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
                       (const float *)nullptr, c->d_scale, c->cfg.alpha, c->cfg.update_count_floor);
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
    hipLaunchKernelGGL(td_kernel<MODE_QVAL>, dim3((n + BLOCK_ENVS - 1) / BLOCK_ENVS), dim3(THREADS), 0,
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
        hipLaunchKernelGGL(td_kernel<MODE_TRANS>, dim3(nblk), dim3(THREADS), 0, s, A);
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
