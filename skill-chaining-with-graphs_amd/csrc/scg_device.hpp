// scg_device.hpp — gfx950 device primitives for the skill-chaining hot path.
// Every function implements the SPEC.md section it names, operation for operation (binary32,
// fusion only where fmaf() is written; the translation unit is built with -ffp-contract=off).
// There is no upstream code to cite: /root/reference is README.md:1-2 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace scg {

constexpr int NACT = 5;
constexpr int NF = 1296;
#ifndef SCG_BLOCK_ENVS
#define SCG_BLOCK_ENVS 256     // (build parameter for the small-batch experiment of DESIGN §10: 64 / 128; the oracle follows by SCO_BLOCK_ENVS)
#endif
constexpr int BLOCK_ENVS = SCG_BLOCK_ENVS;    // SPEC §5 geometry: envs per block = per workgroup
constexpr int WAVES = 16;                     // wavefronts per workgroup (one workgroup per CU: four waves per SIMD)
constexpr int LIST_WAVES = BLOCK_ENVS / 64;   // waves that ballot the workgroup's env flags
constexpr int THREADS = WAVES * 64;
constexpr int MAX_EDGES = 256;
constexpr int CLF_STRIDE = 8;
constexpr int MAX_VF = 6;

// ------------------------------------------------------------------ SPEC §2
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t (&out)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// ------------------------------------------------------------------ SPEC §3
__device__ __forceinline__ float2 sincospi_cs(float t) {   // returns (cos, sin)
    const float S0 = 0x1.921fb6p+1f, S1 = -0x1.4abbcep+2f, S2 = 0x1.466bc6p+1f, S3 = -0x1.32d2ccp-1f,
                S4 = 0x1.507834p-4f;
    const float C0 = -0x1.3bd3ccp+2f, C1 = 0x1.03c1f0p+2f, C2 = -0x1.55d3c8p+0f, C3 = 0x1.e1f506p-3f,
                C4 = -0x1.a6d1f2p-6f;
    const float n = rintf(t + t);
    const float r = fmaf(n, -0.5f, t);
    const float z = r * r;
    float sp = fmaf(z, S4, S3); sp = fmaf(z, sp, S2); sp = fmaf(z, sp, S1); sp = fmaf(z, sp, S0); sp = sp * r;
    float cp = fmaf(z, C4, C3); cp = fmaf(z, cp, C2); cp = fmaf(z, cp, C1); cp = fmaf(z, cp, C0);
    cp = fmaf(z, cp, 1.0f);
    const int q = (int)n & 3;
    float c = cp, s = sp;
    if (q == 1) { c = -sp; s = cp; }
    else if (q == 2) { c = -cp; s = -sp; }
    else if (q == 3) { c = sp; s = -cp; }
    return make_float2(c, s);
}

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(fmaf(-a.y, b.y, a.x * b.x), fmaf(a.x, b.y, a.y * b.x));
}

// powers Z^1..Z^5 of the four normalised state variables -> dst[d*5 + (k-1)]
__device__ __forceinline__ void state_powers(float x, float y, float vx, float vy, float2 *dst) {
    const float sh[4] = {x, y, fmaf(vx, 0.25f, 0.5f), fmaf(vy, 0.25f, 0.5f)};
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const float2 z1 = sincospi_cs(sh[d]);
        float2 z = z1;
        dst[d * 5] = z;
#pragma unroll
        for (int k = 2; k <= 5; ++k) { z = cmul(z, z1); dst[d * 5 + k - 1] = z; }
    }
}

__device__ __forceinline__ float2 pow_at(const float2 *pw, int d, int k) {   // Z_d^k, k in 0..5
    const float2 v = pw[d * 5 + (k > 0 ? k - 1 : 0)];
    return k == 0 ? make_float2(1.0f, 0.0f) : v;
}

// ------------------------------------------------------------------ 64-lane butterfly (SPEC §6)
// for m in (1,2,4,8,16,32): v_l = v_l + v_(l xor m). Stages 1..8 are DPP adds inside a 16-lane row
// (after stages 1,2 a quad is uniform, so row_half_mirror == xor 4; after stage 4, row_mirror == xor 8);
// stages 16 and 32 use gfx950's v_permlane16_swap / v_permlane32_swap: with both operands = v they return
// the two partner rows / halves, whose (commutative) sum is the same bits in both partners. No LDS.
__device__ __forceinline__ float wave_sum(float v) {
    v = v + __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
    v = v + __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
    v = v + __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x141, 0xF, 0xF, true));  // row_half_mirror
    v = v + __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x140, 0xF, 0xF, true));  // row_mirror
    const auto r16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(r16[0]) + __uint_as_float(r16[1]);
    const auto r32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(r32[0]) + __uint_as_float(r32[1]);
    return v;
}

__device__ __forceinline__ float swz_xor4(float v) { return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x101F)); }

// ------------------------------------------------------------------ SPEC §1.3
struct MapScalars {
    float hstep, R2, TX, TY, TR2, R, TR;
    int n_edges, n_starts;
};

__device__ __forceinline__ bool intercept(const float4 ea, const float inv_len2, float R2, float x,
                                          float y, float vx, float vy) {
    const float KAPPA2 = 0x1.0553bep-14f;
    const float dx = x - ea.x, dy = y - ea.y;
    float t = fmaf(dy, ea.w, dx * ea.z) * inv_len2;
    t = fminf(fmaxf(t, 0.0f), 1.0f);
    const float cx = fmaf(ea.z, t, ea.x), cy = fmaf(ea.w, t, ea.y);
    const float bx = cx - x, by = cy - y;
    const float d2 = fmaf(by, by, bx * bx);
    if (d2 > R2) return false;
    const float dot = fmaf(by, vy, bx * vx);
    if (dot >= 0.0f) return true;
    const float vv = fmaf(vy, vy, vx * vx);
    return dot * dot <= (KAPPA2 * d2) * vv;
}

// squared distance from (x,y) to edge (prefilter only; same formula as intercept's d2)
__device__ __forceinline__ float edge_d2(const float4 ea, const float inv_len2, float x, float y) {
    const float dx = x - ea.x, dy = y - ea.y;
    float t = fmaf(dy, ea.w, dx * ea.z) * inv_len2;
    t = fminf(fmaxf(t, 0.0f), 1.0f);
    const float cx = fmaf(ea.z, t, ea.x), cy = fmaf(ea.w, t, ea.y);
    const float bx = cx - x, by = cy - y;
    return fmaf(by, by, bx * bx);
}

constexpr int CELL_G = 32;         // candidate-mask grid: 32 x 32 cells over the unit square

// `edges` = LDS table [n_edges][8]; `cellmask` = global table [32*32][4] of 64-bit edge masks (edges that can come within reach
// of any point of the cell at any legal speed — built on the host by scg_set_map). NW = number of 64-bit mask words in use
// (n_edges <= 64 NW).
// ------------------------------------------------------------------ SPEC §1.3, a whole wavefront of envs at once
// A per-lane form (rounds 1-2: one env per lane, three candidate edges in registers + an overflow loop) walks every candidate
// edge of its env one after another, 20 times, and a wave is as slow as its env with the most candidates: on the bench workload half the envs have no edge within reach at all, a third have two
// or more, and every 32-lane wave held some of each (3 register slots + the overflow loop = ~110 vector instructions per
// sub-step for everybody). Here the wave first settles the envs without candidates (free flight: 2 fmas + the goal test per
// sub-step), then deals the (env, candidate edge) PAIRS of the others to its lanes — one intercept per lane and sub-step —
// and combines the hits of an env's lanes (a run of <= PCAP consecutive lanes) with one ballot: count = popcount, first hit
// = lowest set bit, i.e. the lowest edge index, as SPEC §1.3 asks. Same candidate set, same tests, same order of
// operations per env as that form. Envs with more than PCAP candidates take a per-lane loop over their candidate mask.
constexpr int PCAP = 8;                        // candidate edges per env the pair form handles
constexpr int PITEMS = 64 * PCAP + 64;         // pair slots per wave (a run never straddles a group of 64: up to 7 pad slots per group)

// intercept() without its early exits (the same operations, every lane computes all of them): in the pair loop the exits were
// exec-mask branches that cost more scalar instructions than the arithmetic they skipped
__device__ __forceinline__ bool intercept_flat(const float4 ea, const float inv_len2, float R2, float x, float y, float vx, float vy) {
    const float KAPPA2 = 0x1.0553bep-14f;
    const float dx = x - ea.x, dy = y - ea.y;
    float t = fmaf(dy, ea.w, dx * ea.z) * inv_len2;
    t = fminf(fmaxf(t, 0.0f), 1.0f);
    const float cx = fmaf(ea.z, t, ea.x), cy = fmaf(ea.w, t, ea.y);
    const float bx = cx - x, by = cy - y;
    const float d2 = fmaf(by, by, bx * bx);
    const float dot = fmaf(by, vy, bx * vx);
    const float vv = fmaf(vy, vy, vx * vx);
    return !(d2 > R2) & ((dot >= 0.0f) | (dot * dot <= (KAPPA2 * d2) * vv));
}

// Part 1, by the wave that owns the envs (lane = env): impulse, candidate refinement, free flight for the envs without a
// candidate, the per-lane loop for envs with more than PCAP, and the pair list of the others in `items` (runs of <= PCAP
// consecutive slots, none straddling a group of 64). Returns the number of 64-slot groups to process (wave-uniform);
// `par` = this lane's env is in the list (its state sits in xs, to be read back by pinball_wave_finish).
template <int NW>
__device__ __forceinline__ int pinball_wave_prepare(const float *edges, const uint64_t *cellmask, const MapScalars &ms,
                                                    bool valid, float &x, float &y, float &vx, float &vy, int a,
                                                    bool &goal, bool &par, uint32_t *items, float *xs, int stride) {
    const float DV = 0x1.99999ap-3f, VMAX = 2.0f;
    const float4 *E4 = reinterpret_cast<const float4 *>(edges);
    const int lane = threadIdx.x & 63;
    {   // (as selects: the if / else-if chain over two by-reference floats became a dynamically indexed private array — the
        //  velocities went through scratch memory, three dependent round trips at the head of the physics)
        const float vxp = vx + DV, vxm = vx - DV, vyp = vy + DV, vym = vy - DV;
        vx = a == 0 ? vxp : (a == 2 ? vxm : vx);
        vy = a == 1 ? vyp : (a == 3 ? vym : vy);
    }
    vx = fminf(fmaxf(vx, -VMAX), VMAX);
    vy = fminf(fmaxf(vy, -VMAX), VMAX);
    // candidate set (SPEC §1.3, last paragraph; same bound and same refinement as pinball_step)
    const float spd = __builtin_sqrtf(fmaf(vy, vy, vx * vx));
    const float rr = fmaf(1.10f, spd, 1.02f);
    const float reach2 = ms.R2 * rr * rr;
    const int cxi = min(max((int)(x * (float)CELL_G), 0), CELL_G - 1);
    const int cyi = min(max((int)(y * (float)CELL_G), 0), CELL_G - 1);
    const uint64_t *cm = cellmask + (size_t)(cyi * CELL_G + cxi) * 4;
    uint64_t mask[NW];
    int nc = 0;
#pragma unroll
    for (int g = 0; g < NW; ++g) {
        uint64_t m = valid ? cm[g] : 0ull, keep = 0;
        while (m) {
            const int b = __builtin_ctzll(m);
            m &= m - 1;
            const int j = g * 64 + b;
            if (edge_d2(E4[2 * j], edges[8 * j + 4], x, y) <= reach2) { keep |= 1ull << b; ++nc; }
        }
        mask[g] = keep;
    }
    const float gx0 = x - ms.TX, gy0 = y - ms.TY;
    const float gr = ms.TR + (rr - 1.0f) * ms.R;
    const bool near_goal = valid && fmaf(gy0, gy0, gx0 * gx0) <= gr * gr;
    const float h = ms.hstep;
    goal = false;
    // ---- envs without a candidate: free flight
    if (__ballot(valid && nc == 0)) {
        const bool wave_goal = __ballot(near_goal && nc == 0) != 0;
        if (valid && nc == 0) {
            for (int i = 0; i < 20; ++i) {
                x = fmaf(vx, h, x); y = fmaf(vy, h, y);
                if (wave_goal) {
                    const float gx = x - ms.TX, gy = y - ms.TY;
                    if (fmaf(gy, gy, gx * gx) < ms.TR2) { goal = true; break; }
                }
            }
        }
    }
    // ---- envs with more candidates than the pair form takes: the per-lane loop over their mask
    if (__ballot(nc > PCAP)) {
        if (nc > PCAP) {
            for (int i = 0; i < 20; ++i) {
                x = fmaf(vx, h, x); y = fmaf(vy, h, y);
                int nhit = 0, first = -1;
#pragma unroll
                for (int g = 0; g < NW; ++g) {
                    uint64_t m = mask[g];
                    while (m) {
                        const int j = g * 64 + __builtin_ctzll(m);
                        m &= m - 1;
                        if (intercept(E4[2 * j], edges[8 * j + 4], ms.R2, x, y, vx, vy)) { if (nhit == 0) first = j; ++nhit; }
                    }
                }
                if (nhit == 1) {
                    const float ux = edges[8 * first + 5], uy = edges[8 * first + 6];
                    const float pr = fmaf(vy, uy, vx * ux);
                    const float tp = pr + pr;
                    const float nvx = fmaf(tp, ux, -vx), nvy = fmaf(tp, uy, -vy);
                    vx = nvx; vy = nvy;
                    if (i == 19) { x = fmaf(vx, h, x); y = fmaf(vy, h, y); }
                } else if (nhit > 1) {
                    vx = -vx; vy = -vy;
                }
                const float gx = x - ms.TX, gy = y - ms.TY;
                if (fmaf(gy, gy, gx * gx) < ms.TR2) { goal = true; break; }
            }
        }
    }
    // ---- the others: one slot per (env, candidate edge)
    par = nc >= 1 && nc <= PCAP;
    if (!__ballot(par)) return 0;
    const int cnt = par ? nc : 0;
    int p = cnt;                                           // exclusive prefix of cnt over the lanes
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
        const int t = __shfl_up(p, m, 64);
        if (lane >= m) p += t;
    }
    int total = __shfl(p, 63, 64);
    p -= cnt;
    for (int bnd = 64; bnd < total; bnd += 64) {           // no run may straddle a group of 64 slots: push it to the next group
        const uint64_t cross = __ballot(cnt > 0 && p < bnd && p + cnt > bnd);
        if (cross) {
            const int src = (int)__builtin_ctzll(cross);
            const int shift = bnd - __shfl(p, src, 64);
            if (lane >= src) p += shift;
            total += shift;
        }
    }
    const int groups = (total + 63) >> 6;
    for (int q = lane; q < 64 * groups; q += 64) items[q] = 0xffffffffu;      // pad slots stay empty
    if (par) {
        xs[lane] = x; xs[stride + lane] = y; xs[2 * stride + lane] = vx; xs[3 * stride + lane] = vy;
        int c = 0;
#pragma unroll
        for (int g = 0; g < NW; ++g) {
            uint64_t m = mask[g];
            while (m) {
                const int j = g * 64 + __builtin_ctzll(m);
                m &= m - 1;
                items[p + c] = (unsigned)lane | ((unsigned)j << 8) | ((unsigned)c << 16) | ((unsigned)cnt << 20) | (near_goal ? 1u << 24 : 0u);
                ++c;
            }
        }
    }
    return groups;
}

// Part 2, by ANY wave: the 20 sub-steps of one group of 64 pair slots. The envs' states are read from and written back to
// xs (slots of the wave that owns them), their goal flags to gflag.
__device__ __forceinline__ void pinball_wave_group(const float *edges, const MapScalars &ms, const uint32_t *items64,
                                                   float *xs, int stride, uint8_t *gflag) {
    const float4 *E4 = reinterpret_cast<const float4 *>(edges);
    const int lane = threadIdx.x & 63;
    const float h = ms.hstep;
    const unsigned it = items64[lane];
    const bool act = it != 0xffffffffu;
    const int src = act ? (int)(it & 63u) : 0, j = act ? (int)((it >> 8) & 255u) : 0;
    const int c = (int)((it >> 16) & 15u), n = (int)((it >> 20) & 15u);
    const uint64_t seg = act ? (((1ull << n) - 1ull) << (lane - c)) : 0ull;
    const bool wave_goal = __ballot(act && ((it >> 24) & 1u)) != 0;
    float px = xs[src], py = xs[stride + src], pvx = xs[2 * stride + src], pvy = xs[3 * stride + src];
    const float4 ea = E4[2 * j];
    const float4 eb = E4[2 * j + 1];                       // inv_len2, ux, uy, pad
    bool pg = false;
    for (int i = 0; i < 20; ++i) {
        if (!pg) { px = fmaf(pvx, h, px); py = fmaf(pvy, h, py); }
        const bool hit = act & !pg & intercept_flat(ea, eb.x, ms.R2, px, py, pvx, pvy);
        const uint64_t hits = __ballot(hit);
        if (hits) {                                         // wave-uniform
            const uint64_t mine = hits & seg;
            const int nhit = __popcll(mine);
            const int fl = mine ? (int)__builtin_ctzll(mine) : lane;
            const float ux = __shfl(eb.y, fl, 64), uy = __shfl(eb.z, fl, 64);      // the lowest edge index among the hits
            if (nhit == 1) {
                const float pr = fmaf(pvy, uy, pvx * ux);
                const float tp = pr + pr;
                const float nvx = fmaf(tp, ux, -pvx), nvy = fmaf(tp, uy, -pvy);
                pvx = nvx; pvy = nvy;
                if (i == 19) { px = fmaf(pvx, h, px); py = fmaf(pvy, h, py); }
            } else if (nhit > 1) {
                pvx = -pvx; pvy = -pvy;
            }
        }
        if (wave_goal) {
            const float gx = px - ms.TX, gy = py - ms.TY;
            if (act && !pg && fmaf(gy, gy, gx * gx) < ms.TR2) pg = true;
            if (!__ballot(act && !pg)) break;               // every env of this group is in the goal
        }
    }
    if (act && c == 0) {
        xs[src] = px; xs[stride + src] = py; xs[2 * stride + src] = pvx; xs[3 * stride + src] = pvy;
        gflag[src] = pg ? 1 : 0;
    }
}

// Part 3, by the owning wave again: read the pair form's result back, then drag / clamp / reward (SPEC §1.3)
__device__ __forceinline__ float pinball_wave_finish(bool par, float &x, float &y, float &vx, float &vy, int a, bool &goal,
                                                     const float *xs, int stride, const uint8_t *gflag) {
    const float DRAG = 0x1.fd70a4p-1f;
    const int lane = threadIdx.x & 63;
    if (par) { x = xs[lane]; y = xs[stride + lane]; vx = xs[2 * stride + lane]; vy = xs[3 * stride + lane]; goal = gflag[lane] != 0; }
    float reward;
    if (goal) {
        reward = 10000.0f;
    } else {
        vx = vx * DRAG; vy = vy * DRAG;
        x = fminf(fmaxf(x, 0.0f), 1.0f); y = fminf(fmaxf(y, 0.0f), 1.0f);
        reward = (a == 4) ? -1.0f : -5.0f;
    }
    return reward;
}

// dispatch on the number of mask words the map needs (wave-uniform)
__device__ __forceinline__ int pinball_wave_prepare_any(const float *edges, const uint64_t *cellmask, const MapScalars &ms,
                                                        bool valid, float &x, float &y, float &vx, float &vy, int a, bool &goal,
                                                        bool &par, uint32_t *items, float *xs, int stride) {
    if (ms.n_edges <= 64) return pinball_wave_prepare<1>(edges, cellmask, ms, valid, x, y, vx, vy, a, goal, par, items, xs, stride);
    if (ms.n_edges <= 128) return pinball_wave_prepare<2>(edges, cellmask, ms, valid, x, y, vx, vy, a, goal, par, items, xs, stride);
    return pinball_wave_prepare<4>(edges, cellmask, ms, valid, x, y, vx, vy, a, goal, par, items, xs, stride);
}

// ------------------------------------------------------------------ SPEC §4.1
__device__ __forceinline__ float clf_z(const float *w, float x, float y) {
    const float u = fmaf(x, 2.0f, -1.0f), v = fmaf(y, 2.0f, -1.0f);
    float z = w[0];
    z = fmaf(w[1], u, z); z = fmaf(w[2], v, z);
    z = fmaf(w[3], u * u, z); z = fmaf(w[4], u * v, z); z = fmaf(w[5], v * v, z);
    return z;
}

// ------------------------------------------------------------------ SPEC §6
__device__ __forceinline__ float sigmoid_spec(float z) {
    const float LOG2E = 0x1.715476p+0f, LN2HI = 0x1.63p-1f, LN2LO = -0x1.bd0106p-13f;
    const float E2 = 0x1p-1f, E3 = 0x1.555556p-3f, E4 = 0x1.555556p-5f, E5 = 0x1.111112p-7f,
                E6 = 0x1.6c16c2p-10f, E7 = 0x1.a01a02p-13f;
    float a = -fabsf(z);
    a = fmaxf(a, -87.0f);
    const float n = rintf(a * LOG2E);
    float r = fmaf(n, -LN2HI, a);
    r = fmaf(n, -LN2LO, r);
    float p = E7;
    p = fmaf(p, r, E6); p = fmaf(p, r, E5); p = fmaf(p, r, E4); p = fmaf(p, r, E3);
    p = fmaf(p, r, E2); p = fmaf(p, r, 1.0f); p = fmaf(p, r, 1.0f);
    const float sc = __uint_as_float((uint32_t)((int)n + 127) << 23);
    const float e = p * sc;
    return z >= 0.0f ? 1.0f / (1.0f + e) : e / (1.0f + e);
}

}  // namespace scg
