// scg_step_kernel.hpp — td_kernel: the step-batch kernel of SPEC §1.3–§5 (included by scg_kernels.hip).
//
// Round 4: ONE 16-wavefront workgroup per CU owns a block of 256 consecutive positions of the option-sorted env order
// (SPEC §5, B = 256) and all 160 KB of the CU's LDS. The root value function and the block's option — the option whose
// envs are the position PREFIX of the block, which is what the chunked env order produces — are staged TOGETHER
// (W_0 and W_k, 54 KB) and run as ONE merged pass: every table of Fourier factors is built once per state and
// serves both value functions:
//   phase P  waves 0..3 one lane per env (act, Pinball physics, bookkeeping, option logic), the physics' (env, edge) pair
//            groups pooled over waves 0..7; MEANWHILE waves 8..15 stage W_0 and W_k, take Z_d^1 of the entry states,
//            build the root's update list and run U1 of BOTH value functions (Q_0(s, a_t) and Q_k(s, a_t) from one set of
//            tables per 8-item column block)
//   phase Z  Z_d^1 of s_next
//   E        per 8-env position group one table build of s_next, then 108 MFMAs per value function that needs the group
//            (T = W (180 x 36, A operands streamed from LDS) x [Re CD | Im CD]); the few envs ENTERING the block's
//            option from outside the prefix are compacted into extra column blocks
//   U2       chunks of 144 padded slots of the root's update list (sorted by action, position): one build per slot gives
//            CDT (shared), PT_0 = delta_0 ABsel and, for the prefix slots, PT_k = delta_k ABsel; wave w < 15 owns
//            the three output tiles (mi = w % 3, ni = 0..2) of action w / 3 of BOTH value functions, in registers for the pass
//   Because an option's items are a prefix of every action run of the root's list, its groups of four are the root's
//   groups of four: the accumulation order of SPEC §5 is reproduced exactly (null items masked to +0 on both operands).
// Value functions with update items that are NOT that prefix (gestating options, a second option in the padded env
// order) take a single-VF pass of their own afterwards (compacted lists, as in rounds 2-3); value functions that only
// have envs entering them are evaluated as extra E units of pass 0 (operands from memory). MODE_TRANS / MODE_QVAL are single passes.
// Every sum has the pinned order of SPEC §3.1 / §5 (no atomics on data): the CPU oracle reproduces every bit.
#pragma once

constexpr int P_WAVES = BLOCK_ENVS / 64;      // phase P: one lane per env on full waves
constexpr int LIST0 = P_WAVES;                // waves LIST0 .. LIST0 + LIST_WAVES - 1 build the passes' flags and lists
#ifndef SCG_HELPER0
#define SCG_HELPER0 7          // (7 pool + 9 helper waves measured +1.2 % over 8 + 8, 6 + 10 +0.6 %)
#endif
constexpr int HELPER0 = SCG_HELPER0;          // waves HELPER0.. work under phase P (learning steps)
constexpr int N_HELP = WAVES - HELPER0;
constexpr int P_POOL = HELPER0;               // waves 0..P_POOL-1 share the physics' (env, edge) pair groups
static_assert(HELPER0 >= P_WAVES && N_HELP >= 8 && N_HELP * 64 * 3 >= 12 * 2 * 64, "stage_w by the helper waves: at most three float4 per thread");
// LDS map of the step kernel (bytes): one workgroup per CU, the whole 160 KB.
constexpr int OFF_RC = 0;                                      // float r0,c0,ro,co (per env), rk,ck (per env, single passes) [B]
constexpr int OFF_INT = OFF_RC + 6 * BLOCK_ENVS * 4;           // uint8 a, ot, on, gs, ia, ev [B]
constexpr int OFF_Z1 = OFF_INT + 6 * BLOCK_ENVS;               // float2 z1[B][2][4]: Z_d^1 of s and s_next
// region R, used by one phase at a time:
//   P, Z  : s[4][B], sn[4][B] (the envs' states), the edge table [256][8] and the pair lists, inside the table area of waves 0..7
//   E, U1 : W_A | W_B staged in A-operand order (12 row tiles x 9 k-blocks x 64 lanes each) + per wave CDk[36][16] + ABq[16][AS]
//   U2    : PT_A[36][US], PT_B[36][US], CDT[36][US] (one chunk of U2_CH padded slots = 2 U2_CH K-steps)
constexpr int W_FLOATS = 12 * 9 * 64;
constexpr int W_TAIL = 12 * 2 * 64 * 4;                        // k-block 8 of every tile sits behind the two float4 groups
constexpr int AS = 40;                                         // row stride of ABq (floats): the fold's ds_read_b128 conflict-free
constexpr int E_TAB_FLOATS = 36 * 16 + 16 * AS;
constexpr int U2_CH = 9 * WAVES;                               // U2 chunk: padded slots, nine per builder wave (54 of 64 lanes busy)
constexpr int US = 2 * U2_CH + 4;                              // row stride of the chunk tables (floats): 292 = 36 mod 64, operand reads conflict-free
constexpr int R_TAB = 2 * W_FLOATS;                            // private tables start behind the staged W_A, W_B
constexpr int R_S = R_TAB, R_EDGES = R_S + 8 * BLOCK_ENVS, R_PITEMS = R_EDGES + MAX_EDGES * 8;
constexpr int cmax3(int a, int b, int c) { return a > b ? (a > c ? a : c) : (b > c ? b : c); }
constexpr int R_FLOATS = cmax3(R_TAB + WAVES * E_TAB_FLOATS, 3 * 36 * US, R_PITEMS + P_WAVES * PITEMS);
constexpr int OFF_R = OFF_Z1 + BLOCK_ENVS * 2 * 4 * 8;
constexpr int OFF_ELIST = OFF_R + R_FLOATS * 4;                // uint16 compacted eval list [2 B] (stragglers / single passes)
constexpr int OFF_ULIST = OFF_ELIST + 2 * BLOCK_ENVS * 2;      // uint16 update list [B] (5 action runs)
constexpr int OFF_MAXQ = OFF_ULIST + BLOCK_ENVS * 2;           // float maxq[2][B] (per value function of the pass, per env)
constexpr int OFF_QSA = OFF_MAXQ + 2 * BLOCK_ENVS * 4;         // float qsa[2][B] (per update-list position)
constexpr int OFF_EFLAG = OFF_QSA + 2 * BLOCK_ENVS * 4;        // uint8 eflag[B / 8]: per position group, bit v = value function v needs it
constexpr int OFF_CLF = OFF_EFLAG + 64;                        // float clf[6][8]
constexpr int OFF_MISC = OFF_CLF + MAX_VF * CLF_STRIDE * 4;    // int misc[128]
#ifdef SCG_STAMPS
constexpr int STAMP_SLOTS = 48;                                // 32 sections of wave 0 + the E phase of every wave
constexpr int OFF_STAMP = OFF_MISC + 512;                      // unsigned stamp[STAMP_SLOTS] (diagnostic build)
constexpr int LDS_BYTES = OFF_STAMP + STAMP_SLOTS * 4;
#else
constexpr int LDS_BYTES = OFF_MISC + 512;
#endif
#ifdef SCG_STAMPS_LITE
// The light timing build (round 5): the full stamps build runs 7 % slower than the product and a change that took 3 % off ITS kernel
// took nothing off the product's — it measures another kernel. Here four waves (an env wave, a pool wave, the first and the last
// helper) read the clock at eight phase boundaries into scalar registers and write them out once, at the end: no LDS, no branches
// on the way. Slots per block: [wave role 0..3][8 boundaries] cycles since the wave's kernel entry, [32] start and [33] end on the
// 100 MHz wall clock (launch ramp across the chip).
constexpr int STAMP_SLOTS = 48;
#if SCG_STAMPS_LITE == 3      // variant: waves 12, 13 (two of the no-op's), 0 and 3 stamp INSIDE U2 of pass 0 (tools/lite_report.py --u2)
#define SCG_LITE(I) do { if (MODE == MODE_FUSED && lite_role >= 0 && ((I) == 0 || (I) == 7)) lt[(I)] = __builtin_amdgcn_s_memtime(); } while (0)
#define SCG_LITEP(I) do { } while (0)
#define SCG_LITEU(I) do { if (MODE == MODE_FUSED && lite_role >= 0 && pass == 0) lt[(I)] = __builtin_amdgcn_s_memtime(); } while (0)
#elif SCG_STAMPS_LITE == 2    // variant: the env wave's eight boundaries lie INSIDE phase P (the other roles keep theirs)
#define SCG_LITE(I) do { if (MODE == MODE_FUSED && (lite_role >= 1 || (I) == 0)) lt[(I)] = __builtin_amdgcn_s_memtime(); } while (0)
#define SCG_LITEP(I) do { if (MODE == MODE_FUSED && lite_role == 0) lt[(I)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define SCG_LITE(I) do { if (MODE == MODE_FUSED && lite_role >= 0) lt[(I)] = __builtin_amdgcn_s_memtime(); } while (0)
#define SCG_LITEP(I) do { } while (0)
#endif
#else
#define SCG_LITE(I) do { } while (0)
#define SCG_LITEP(I) do { } while (0)
#endif
#ifndef SCG_LITEU
#define SCG_LITEU(I) do { } while (0)
#endif
static_assert(BLOCK_ENVS / 8 <= 64, "eflag area");
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget: one workgroup per CU");
static_assert(OFF_Z1 % 16 == 0 && OFF_R % 16 == 0 && (R_TAB * 4) % 16 == 0 && (R_EDGES * 4) % 16 == 0 && (E_TAB_FLOATS * 4) % 16 == 0, "LDS alignment");
static_assert(R_PITEMS + P_WAVES * PITEMS <= R_TAB + HELPER0 * E_TAB_FLOATS, "states + edges + the physics pair lists fit the table area of the waves below the helpers");
// s_misc (ints)
constexpr int M_CNT = 0;          // [LIST_WAVES][16]: per list wave {compacted-eval count, five action-run counts, five counts of B's share of the runs}
constexpr int M_GROUPS = 64;      // [P_WAVES] pair groups of each env wave
constexpr int M_C_PREP = 68;      // hand-off counters: env waves have listed their pair groups,
constexpr int M_C_POOL = 69;      //   pool waves are through with them,
constexpr int M_C_HELP = 70;      //   helper waves have staged W / Z(s) / the list,
constexpr int M_C_PUB = 71;       //   envs whose state and action are published
constexpr int M_PRESENT = 72;     // bit k: some env here has an item for VF k
constexpr int M_UPD = 73;         // bit k: ... an UPDATE item
constexpr int M_FAIL = 74;        // a bounded hand-off poll ran out (reported through the async status word)
constexpr int M_C_PUBS = 77;      //   envs whose entry state and option id are published (the actions follow: M_C_PUB)
constexpr int M_KB = 75, M_MB = 76; // the block's option (value function B of the merged pass) and its prefix length, published by wave HELPER0
constexpr int M_U1CTR = 78;        // next U1 column block to take (helper waves, dynamic dealing)
constexpr int M_ECTR = 79;         // next E unit to take (dynamic dealing)
constexpr int M_EO = 80;           // [12] evaluation-only lists: [k] items of value function k (k = 1..5), [6 + k] their start in s_elist, [6] units in all
constexpr int M_C_TOP = 92;         // waves without envs have put the edge and classifier tables into LDS
constexpr int M_INTS = 128;
static_assert(LIST_WAVES * 16 <= M_GROUPS && P_WAVES <= 4, "s_misc layout");

enum { MODE_FUSED = 0, MODE_TRANS = 1, MODE_QVAL = 2 };
constexpr unsigned IA_ENTERING = 0x40u;     // s_ia bit 6: the env is about to ENTER option s_on (SPEC §4.2 value-gated entry)
constexpr unsigned OREC_DECLINED = 0x7fc0deadu;   // result line word [3].y: the env stays with the root after all (commit_row: option 0, the root's Q values)
// Build-time knobs for tools/ab_bench.py (the variants measured and dropped in round 4 — evaluation-only value functions on the
// vector pipe or behind the passes, static deals of E's and U1's units, E rebalancing — are kept as
// profiles/r04_dropped_kernel_variants.diff with their numbers in profiles/r04_td_kernel_ab_log.txt).
#ifndef SCG_E_TG
#define SCG_E_TG 4            // row tiles per operand group of the LDS-fed contraction (12 % SCG_E_TG == 0)
#endif
#ifndef SCG_EO_TG
#define SCG_EO_TG 3           // row tiles whose operands contract_g fetches together (register budget: 9 per tile)
#endif
#ifndef SCG_PRIO_P
#define SCG_PRIO_P 2          // env waves during phase P
#endif
#ifndef SCG_PRIO_HELP
#define SCG_PRIO_HELP 0       // helper waves during phase P
#endif
#ifndef SCG_PRIO_LIST
#define SCG_PRIO_LIST 1       // list waves (0..3) after phase P
#endif

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

template <int M>
__device__ __forceinline__ int sel5(const int (&v)[M], int a) {          // v[a], a < 5, without a dynamically indexed array
    return a == 0 ? v[0] : a == 1 ? v[1] : a == 2 ? v[2] : a == 3 ? v[3] : v[4];
}

// Lane roles. As an MFMA operand lane (16x16x4): n16 = lane & 15 is the tile row (A) / column (B, C, D), g = lane >> 4
// the k index (A, B) / the row group (C, D: rows 4 g + v). As a table builder: bi = lane & 7 is the item of an
// 8-item column block, cp = lane >> 3 the second index (c2 / c4; lanes with cp >= 6 idle).
// Columns of an 8-item block: item j = 4 h + i  (h = 0, 1; i = 0..3) has its real-part column at 8 h + i and its
// imaginary-part column at 8 h + 4 + i.
// Every phase derives them afresh from an opaque copy of the lane id: kept live from the top of the kernel, these dozen
// values (and the per-lane LDS addresses made from them) were what the register allocator spilled to scratch at every
// phase boundary — 1024 threads x 256 workgroups going to memory together, on the critical path (round 4 stamps).
#define SCG_LANE_ROLES()                                                                                          \
    int lane_r = lane;                                                                                            \
    asm volatile("" : "+v"(lane_r));                                                                              \
    const int n16 = lane_r & 15, g = lane_r >> 4;                                                                 \
    const int bi = lane_r & 7, cp = lane_r >> 3;                                                                  \
    const int bcol = 8 * (bi >> 2) + (bi & 3);               /* builder: real-part column of item bi */          \
    const int ocol_item = 4 * (n16 >> 3) + (n16 & 3);        /* operand lane: item of column n16 within the block */ \
    const bool out_lane = (g == 0) && !(n16 & 4);            /* lanes that hold an item's finished sums */       \
    float *cdk = s_R + R_TAB + wave * E_TAB_FLOATS, *abq = cdk + 36 * 16;     /* this wave's private tables */  \
    const float *ab_lane = abq + n16 * AS + 4 * g;                                                                \
    const f4v *w4 = reinterpret_cast<const f4v *>(s_W) + lane_r;                                                  \
    const float *w8 = s_W + W_TAIL + lane_r;                                                                      \
    (void)bi; (void)cp; (void)bcol; (void)ocol_item; (void)out_lane; (void)ab_lane; (void)w4; (void)w8; (void)n16; (void)g

// A zero accumulator made on the spot from an opaque scalar: written as the constant (f4v){0, 0, 0, 0}, LLVM hoists ONE 128-bit zero out of the
// pass loop, keeps it live across the whole pass in a register tuple and — at 128 VGPRs — spills and reloads that tuple a dozen times per
// pass (16 MB of scratch writes per launch, round 5) instead of re-making it with four moves.
__device__ __forceinline__ f4v zero4() {
    float z = 0.0f;
    asm volatile("" : "+v"(z));
    return (f4v){z, z, z, z};
}

// ... and the same for the integer zeros the list offsets start from: two of them, paired, were hoisted out of the pass loop, spilled before
// E and reloaded once per unit inside E's unit loop (2 MB of scratch writes per launch)
__device__ __forceinline__ int zero_i() {
    int z = 0;
    asm volatile("" : "+v"(z));
    return z;
}

// W_k -> LDS in A-operand order, OUT OF LINE, for the in-pass staging (single-VF passes, steps without helper waves): that path is cold,
// and inlined into the pass loop its sixteen-byte temporaries took part in the register allocation of the loop's hot stretches — 28 spill /
// reload sites at pass level (16 MB of scratch writes per launch) instead of 2 (round 5).
__device__ __attribute__((noinline)) void stage_w_cold(const float *Wk, float *s_dst, int tid) {
    // (all THREADS threads; every thread's loads are issued before its first LDS store: one memory round trip)
    struct __attribute__((packed, aligned(4))) F4U { float x, y, z, w; };
    f4v *dst4 = reinterpret_cast<f4v *>(s_dst);
    constexpr int N4 = 12 * 2 * 64, N1 = 12 * 64;
    static_assert(2 * THREADS >= N4 && THREADS >= N1, "two float4 and one float per thread");
    f4v v[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int d = tid + i * THREADS;
        v[i] = (f4v){0.0f, 0.0f, 0.0f, 0.0f};
        if (d < N4) {
            const int t2h = d >> 6, ln = d & 63, row = 16 * (t2h >> 1) + (ln & 15), col = 9 * (ln >> 4) + 4 * (t2h & 1);
            if (row < NACT * 36) {
                const F4U w = *reinterpret_cast<const F4U *>(Wk + row * 36 + col);
                v[i] = (f4v){w.x, w.y, w.z, w.w};
            }
        }
    }
    float tl = 0.0f;
    if (tid < N1) {
        const int ln = tid & 63, row = 16 * (tid >> 6) + (ln & 15);
        if (row < NACT * 36) tl = Wk[row * 36 + 9 * (ln >> 4) + 8];
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) { const int d = tid + i * THREADS; if (d < N4) dst4[d] = v[i]; }
    if (tid < N1) s_dst[W_TAIL + tid] = tl;
}

template <int MODE>
__global__ __launch_bounds__(THREADS, 4) void td_kernel(const StepArgs A) {
    // STATIC LDS: its addresses are compile-time literals. As dynamic LDS (extern __shared__[]) every address was base symbol + offset
    // — an s_add the compiler hoisted out of the pass loop and parked in spilled SGPRs (109 of them, 172 SGPR spills; 126 now): +0.65 %
    __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];
    float *s_R = reinterpret_cast<float *>(smem + OFF_R);
    float *s_s = s_R + R_S;                                             // [8][B]: s then sn   (region R, phases P and Z)
    float *s_edges = s_R + R_EDGES;                                     // [n_edges][8]        (region R, phase P)
    uint32_t *s_pitems = reinterpret_cast<uint32_t *>(s_R) + R_PITEMS;  // [P_WAVES][PITEMS] (env, edge) pairs of the physics
    float *s_r0 = reinterpret_cast<float *>(smem + OFF_RC);
    float *s_c0 = s_r0 + BLOCK_ENVS, *s_ro = s_c0 + BLOCK_ENVS, *s_co = s_ro + BLOCK_ENVS;
    float *s_rk = s_co + BLOCK_ENVS, *s_ck = s_rk + BLOCK_ENVS;       // reward / continuation of a single pass's value function
    uint8_t *s_a = reinterpret_cast<uint8_t *>(smem + OFF_INT);
    uint8_t *s_ot = s_a + BLOCK_ENVS, *s_on = s_ot + BLOCK_ENVS;
    uint8_t *s_gs = s_on + BLOCK_ENVS;      // bit k: gestating option k holds s in its initiation set (off-policy item)
    uint8_t *s_ia = s_gs + BLOCK_ENVS;      // bit 0: goal; bit k: in_k(s')
    uint8_t *s_ev = s_ia + BLOCK_ENVS;      // bit v: the env needs Q(s_next, .) of value function v of the current pass
    float2 *s_z1 = reinterpret_cast<float2 *>(smem + OFF_Z1);
    uint16_t *s_elist = reinterpret_cast<uint16_t *>(smem + OFF_ELIST);
    uint16_t *s_ulist = reinterpret_cast<uint16_t *>(smem + OFF_ULIST);
    float *s_maxq = reinterpret_cast<float *>(smem + OFF_MAXQ);
    float *s_qsa = reinterpret_cast<float *>(smem + OFF_QSA);
    uint8_t *s_eflag = reinterpret_cast<uint8_t *>(smem + OFF_EFLAG);
    float *s_W = s_R;                                                   // region W = the head of region R: W_A then W_B
    float *s_clf = reinterpret_cast<float *>(smem + OFF_CLF);
    int *s_misc = reinterpret_cast<int *>(smem + OFF_MISC);

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef SCG_DIAG_ROTATE          // (diagnostic: workgroup p works on block (p + SCG_DIAG_ROTATE) % grid — same results; do the slow blocks follow the DATA or the PLACE? tools/straggler_report.py)
    const int b = (int)((blockIdx.x + SCG_DIAG_ROTATE) % gridDim.x);
#else
    const int b = blockIdx.x;
#endif
    const int e0 = b * BLOCK_ENVS;
    const int nb = min(BLOCK_ENVS, A.n - e0);
    const int N = A.n;
#ifdef SCG_STAMPS_LITE
#if SCG_STAMPS_LITE == 3
    const int lite_role = wave == 12 ? 0 : wave == 13 ? 1 : wave == 0 ? 2 : wave == 3 ? 3 : -1;
#else
    const int lite_role = wave == 0 ? 0 : wave == P_WAVES ? 1 : wave == HELPER0 ? 2 : wave == WAVES - 1 ? 3 : -1;
#endif
    unsigned long long lt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long lite_r0 = __builtin_amdgcn_s_memrealtime();
    SCG_LITE(0);
#endif
#ifdef SCG_STAMPS
    unsigned *s_stamp = reinterpret_cast<unsigned *>(smem + OFF_STAMP);
    if (tid < STAMP_SLOTS) s_stamp[tid] = 0;
    unsigned long long stamp_prev = __builtin_amdgcn_s_memtime();
#endif

    // private tables of one 8-item column block from the builder lane's item `it` (already clamped by the caller),
    // state sg (0 = s, 1 = s_next): CDk[c34][col], ABq[col][c12] with ABsel = (Re AB | -Im AB)
    auto build_tables = [&](int it, int sg, int cp, int bcol, float *cdk, float *abq) {
        if (cp < 6) {
            float2 ab[6], cd[6];
            item_entries(s_z1 + (it * 2 + sg) * 4, cp, ab, cd);
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                abq[bcol * AS + 6 * c + cp] = ab[c].x; abq[(bcol + 4) * AS + 6 * c + cp] = -ab[c].y;
                cdk[(6 * c + cp) * 16 + bcol] = cd[c].x; cdk[(6 * c + cp) * 16 + bcol + 4] = cd[c].y;
            }
        }
    };

    // A operands come from the staged W (value function v of the pass at s_W + v W_FLOATS), per row tile two ds_read_b128
    // (k-blocks 0..3, 4..7) and one ds_read_b32 (k-block 8): w4 / w8 of SCG_LANE_ROLES
    // The pass's update list (s_ulist: items of value function A sorted by action, block order inside a run) and its geometry;
    // nBa[a] = how many items at the head of run a also update value function B (0 when the pass has none)
    int run_len[NACT], run_off[NACT], nBa[NACT];
#pragma unroll
    for (int a = 0; a < NACT; ++a) { run_len[a] = 0; run_off[a] = 0; nBa[a] = 0; }

    // U1: Q(s, a_t) of the update items, per action run, 8 items per wave-iteration: the contraction of E on the 3 row
    // tiles that hold action a's rows, for value function A and — while the column block still holds B items — for B on
    // the SAME tables -> s_qsa[v][list position]. The column blocks of all runs are dealt to `nw` waves.
    auto u1_block = [&](int a, int cb) {
        SCG_LANE_ROLES();
        {
            const int t0 = (36 * a) >> 4;                        // first of the 3 row tiles holding action a's rows
            const int cnt = sel5(run_len, a), cntB = sel5(nBa, a), ro = sel5(run_off, a);
            const uint16_t *lst = s_ulist + ro;
            {
                build_tables(lst[8 * cb + min(bi, cnt - 8 * cb - 1)], 0, cp, bcol, cdk, abq);
                wave_lds_sync();
                float B[9];
#pragma unroll
                for (int kb = 0; kb < 9; ++kb) B[kb] = cdk[(9 * g + kb) * 16 + n16];
                // both value functions' three row tiles as independent MFMA chains, interleaved (a column block that still holds
                // B items runs 6 chains x 9 MFMAs back to back instead of two times 3 dependent ones)
                auto tiles = [&](auto nv_c) {
                    constexpr int NV = decltype(nv_c)::value;
                    f4v c[NV][3];
#pragma unroll
                    for (int v = 0; v < NV; ++v) {
#pragma unroll
                        for (int tt = 0; tt < 3; ++tt) c[v][tt] = zero4();
                    }
#pragma unroll
                    for (int hk = 0; hk < 2; ++hk) {
                        f4v aop[NV][3];
#pragma unroll
                        for (int v = 0; v < NV; ++v) {
#pragma unroll
                            for (int tt = 0; tt < 3; ++tt) aop[v][tt] = w4[v * (W_FLOATS / 4) + ((t0 + tt) * 2 + hk) * 64];
                        }
#pragma unroll
                        for (int kb = 0; kb < 4; ++kb) {
#pragma unroll
                            for (int v = 0; v < NV; ++v) {
#pragma unroll
                                for (int tt = 0; tt < 3; ++tt)
                                    c[v][tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(aop[v][tt][kb], B[4 * hk + kb], c[v][tt], 0, 0, 0);
                            }
                        }
                    }
#pragma unroll
                    for (int v = 0; v < NV; ++v) {
#pragma unroll
                        for (int tt = 0; tt < 3; ++tt)
                            c[v][tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w8[v * W_FLOATS + (t0 + tt) * 64], B[8], c[v][tt], 0, 0, 0);
                    }
                    float qo[NV];
#pragma unroll
                    for (int v = 0; v < NV; ++v) {
                        float qs = 0.0f;
#pragma unroll
                        for (int tt = 0; tt < 3; ++tt) {
                            const int r0 = 16 * (t0 + tt) + 4 * g - 36 * a;     // c12 of the lane's first row, if in [0, 36)
                            const bool in = r0 >= 0 && r0 < 36;
                            const f4v ab4 = *reinterpret_cast<const f4v *>(abq + n16 * AS + (in ? r0 : 0));
                            float xq = qs;
#pragma unroll
                            for (int vv = 0; vv < 4; ++vv) xq = fmaf(c[v][tt][vv], ab4[vv], xq);
                            qs = in ? xq : qs;
                        }
                        qo[v] = qs;
                    }
                    item_tree_sum<NV>(qo);
#pragma unroll
                    for (int v = 0; v < NV; ++v) {
                        const int cv = v ? cntB : cnt;
                        if (out_lane && 8 * cb + ocol_item < cv) s_qsa[v * BLOCK_ENVS + ro + 8 * cb + ocol_item] = qo[v];
                    }
                };
                if (8 * cb < cntB) tiles(std::integral_constant<int, 2>{});
                else tiles(std::integral_constant<int, 1>{});
                wave_lds_sync();
            }
        }
    };
    // ... the column blocks of all runs dealt to `nw` waves in advance (single passes: behind E's units) ...
    auto run_u1 = [&](int wv, int nw, int base) {
#pragma unroll 1
        for (int a = 0; a < NACT; ++a) {
            const int cnt = sel5(run_len, a);
            for (int cb = (((wv - base) % nw) + nw) % nw; 8 * cb < cnt; cb += nw) u1_block(a, cb);
            base += (cnt + 7) >> 3;
        }
    };
    // ... or taken from a counter in LDS by whichever helper wave is free (under phase P the waves run at very different
    // rates beside the env waves of their SIMDs, and a block with option items costs twice one without)
    auto run_u1_dyn = [&]() {
        int nbk[NACT], tot = 0;
#pragma unroll
        for (int a = 0; a < NACT; ++a) { nbk[a] = (run_len[a] + 7) >> 3; tot += nbk[a]; }
        for (;;) {
            int j = 0;
            if (lane == 0) j = atomicAdd(&s_misc[M_U1CTR], 1);
            j = __builtin_amdgcn_readfirstlane(j);
            if (j >= tot) break;
            int a = 0;
#pragma unroll
            for (int aa = 0; aa < NACT - 1; ++aa) { if (a == aa && j >= nbk[aa]) { j -= nbk[aa]; a = aa + 1; } }
            u1_block(a, j);
        }
    };
    // Q(sigma, .) of the 8 items whose tables sit in this wave's private area, for the value function staged at float offset
    // `wofs` of region W (SPEC §3.1): T = W (180 x 36) x [Re CD | Im CD] on the matrix pipe — rows 16 t + 4 g + v -> action
    // rho / 36, c12 = rho % 36; a lane's four rows never straddle actions — then the fold with the AB factors and the butterfly.
    // The twelve row tiles are taken in two halves of six (MFMAs, then the fold of those six accumulators into the per-action
    // chains, tile order kept): 24 accumulator registers instead of 48 at the kernel's register peak. The finished sums are in
    // the lanes with out_lane.
    auto contract_with = [&](auto tg_c, auto load_a, const float (&B)[9], float (&qo)[NACT], int g, const float *ab_lane) {
        constexpr int TG = decltype(tg_c)::value;             // row tiles per group (operands of a group are fetched together)
        float q[NACT + 1] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        // the operands of group hh + 1 are fetched before the products of group hh are issued (two operand sets in registers)
        f4v a0[2][TG], a1[2][TG];
        float a8[2][TG];
#pragma unroll
        for (int tt = 0; tt < TG; ++tt) load_a(tt, a0[0][tt], a1[0][tt], a8[0][tt]);
#pragma unroll
        for (int hh = 0; hh < 12 / TG; ++hh) {
            if (hh + 1 < 12 / TG) {
#pragma unroll
                for (int tt = 0; tt < TG; ++tt) load_a(TG * (hh + 1) + tt, a0[(hh + 1) & 1][tt], a1[(hh + 1) & 1][tt], a8[(hh + 1) & 1][tt]);
            }
            f4v acc[TG];
#pragma unroll
            for (int tt = 0; tt < TG; ++tt) {
                f4v c = (f4v){0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[hh & 1][tt][kb], B[kb], c, 0, 0, 0);
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[hh & 1][tt][kb], B[4 + kb], c, 0, 0, 0);
                acc[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a8[hh & 1][tt], B[8], c, 0, 0, 0);
            }
#pragma unroll
            for (int tt = 0; tt < TG; ++tt) {
                const int t = TG * hh + tt;
                const int Ct = (16 * t) % 36, At = (16 * t) / 36;
                if (Ct + 12 < 36) {                          // the tile's 16 rows belong to one action
                    const f4v ab4 = *reinterpret_cast<const f4v *>(ab_lane + Ct);
#pragma unroll
                    for (int vv = 0; vv < 4; ++vv) q[At] = fmaf(acc[tt][vv], ab4[vv], q[At]);
                } else {                                     // row groups g >= (36 - Ct) / 4 belong to the next action
                    const bool wrap = 4 * g >= 36 - Ct;
                    const f4v ab4 = *reinterpret_cast<const f4v *>(ab_lane + (wrap ? Ct - 36 : Ct));
                    float xq = wrap ? q[At + 1] : q[At];
#pragma unroll
                    for (int vv = 0; vv < 4; ++vv) xq = fmaf(acc[tt][vv], ab4[vv], xq);
                    q[At] = wrap ? q[At] : xq;
                    q[At + 1] = wrap ? xq : q[At + 1];
                }
            }
        }
#pragma unroll
        for (int a = 0; a < NACT; ++a) qo[a] = q[a];
        item_tree_sum<NACT>(qo);
    };
    // ... A operands from the value function staged at float offset `wofs` of region W (two ds_read_b128 + one ds_read_b32 per tile)
    auto contract = [&](int wofs, const float (&B)[9], float (&qo)[NACT], int n16, int g, const f4v *w4, const float *w8, const float *ab_lane) {
        contract_with(std::integral_constant<int, SCG_E_TG>{}, [&](int t, f4v &a0, f4v &a1, float &a8) {
            a0 = w4[wofs / 4 + (t * 2) * 64]; a1 = w4[wofs / 4 + (t * 2 + 1) * 64]; a8 = w8[wofs + t * 64];
        }, B, qo, g, ab_lane);
    };
    // ... A operands straight from the caller's W_k[5][36][36] in memory: lane (n16, g) of tile t owns the nine consecutive floats
    // W[16 t + n16][9 g .. 9 g + 8] (two 16-byte loads at 4-byte-aligned addresses and one float; rows >= 180: zeros). For the few
    // envs that ENTER an option nobody in the block runs: its weights are not staged anywhere in this workgroup.
    auto contract_g = [&](const float *Wk, const float (&B)[9], float (&qo)[NACT], int n16, int g, const float *ab_lane) {
        struct __attribute__((packed, aligned(4))) F4U { float x, y, z, w; };
        contract_with(std::integral_constant<int, SCG_EO_TG>{}, [&](int t, f4v &a0, f4v &a1, float &a8) {
            const int row = 16 * t + n16;
            a0 = (f4v){0.0f, 0.0f, 0.0f, 0.0f}; a1 = a0; a8 = 0.0f;
            if (row < NACT * 36) {
                const float *pw = Wk + row * 36 + 9 * g;
                const F4U u0 = *reinterpret_cast<const F4U *>(pw), u1 = *reinterpret_cast<const F4U *>(pw + 4);
                a0 = (f4v){u0.x, u0.y, u0.z, u0.w}; a1 = (f4v){u1.x, u1.y, u1.z, u1.w}; a8 = pw[8];
            }
        }, B, qo, g, ab_lane);
    };
    // W_k -> region W (+ dstf floats) in A-operand order (12 row tiles of the 180 x 36 matrix; entry (tile t, k-block kb,
    // lane (n16, g)) = W[16 t + n16][9 g + kb], rows >= 180 zero; per tile and lane the k-blocks 0..3 and 4..7 form two
    // float4 — one ds_read_b128 feeds four MFMAs — and k-block 8 sits apart), by `nth` threads with index `ht`: one thread
    // per DESTINATION float4, a 16-byte load at a 4-byte-aligned source address, one linear ds_write_b128.
    struct WStage { f4v v[3]; float tl[2]; };
    auto stage_w_load = [&](const float *Wk, int ht, int nth, WStage &st) {
        // (all of a thread's loads are issued before its first LDS store: one memory round trip per staging, not one per
        //  loop iteration — the helper waves' W_0 took 10k cycles as three dependent load -> store rounds; nth >= 512)
        struct __attribute__((packed, aligned(4))) F4U { float x, y, z, w; };
        constexpr int N4 = 12 * 2 * 64, N1 = 12 * 64, MAXI = 3, MAXT = 2;
#pragma unroll
        for (int i = 0; i < MAXI; ++i) {
            const int d = ht + i * nth;
            st.v[i] = (f4v){0.0f, 0.0f, 0.0f, 0.0f};
            if (d < N4) {
                const int t2h = d >> 6, ln = d & 63, row = 16 * (t2h >> 1) + (ln & 15), col = 9 * (ln >> 4) + 4 * (t2h & 1);
                if (row < NACT * 36) {
                    const F4U w = *reinterpret_cast<const F4U *>(Wk + row * 36 + col);
                    st.v[i] = (f4v){w.x, w.y, w.z, w.w};
                }
            }
        }
#pragma unroll
        for (int i = 0; i < MAXT; ++i) {                                          // k-block 8 of every tile
            const int z = ht + i * nth;
            st.tl[i] = 0.0f;
            if (z < N1) {
                const int ln = z & 63, row = 16 * (z >> 6) + (ln & 15);
                if (row < NACT * 36) st.tl[i] = Wk[row * 36 + 9 * (ln >> 4) + 8];
            }
        }
    };
    auto stage_w_store = [&](int dstf, int ht, int nth, const WStage &st) {
        f4v *dst4 = reinterpret_cast<f4v *>(s_W + dstf);
        constexpr int N4 = 12 * 2 * 64, N1 = 12 * 64, MAXI = 3, MAXT = 2;
#pragma unroll
        for (int i = 0; i < MAXI; ++i) { const int d = ht + i * nth; if (d < N4) dst4[d] = st.v[i]; }
#pragma unroll
        for (int i = 0; i < MAXT; ++i) { const int z = ht + i * nth; if (z < N1) s_W[dstf + W_TAIL + z] = st.tl[i]; }
    };
    auto stage_w = [&](const float *Wk, int dstf, int ht, int nth) {
        WStage st;
        stage_w_load(Wk, ht, nth, st);
        stage_w_store(dstf, ht, nth, st);
    };
    // counters in LDS for hand-offs between SUBSETS of the workgroup's waves (s_barrier takes all sixteen): a producer
    // publishes with lds_arrive, a consumer polls with lds_await. Every awaited count is reached by waves that never
    // wait on the waiter, so the polls terminate; the bound only guards the GPU against a logic error — a poll that runs
    // out raises M_FAIL: the block then writes no slab and the step reports SCG_ASYNC_STEP_HANDOFF (never a silent SCG_OK).
    auto lds_arrive = [&](int *ctr, int amount) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == __builtin_ctzll(__ballot(true))) atomicAdd(ctr, amount);
    };
    auto lds_await = [&](int *ctr, int want) {
        int spins = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < want) {
            if (++spins > (1 << 20)) { s_misc[M_FAIL] = 1; break; }
            __builtin_amdgcn_s_sleep(2);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    };
    // The block's option (value function B of the merged pass): the option k >= 1 whose envs are exactly the positions
    // [0, m) of the block — what the chunked env order of SPEC §5 produces. Its update items are then a prefix of every
    // action run of the root's list. A gestating option never qualifies (its items are not its own envs). Every wave
    // derives the same (kB, mB) from s_ot.
    int kB = -1, mB = 0;
    auto decide_b = [&]() {
        kB = -1; mB = 0;
        if (MODE != MODE_FUSED || nb <= 0) return;
        const int k0 = s_ot[0];
        if (k0 < 1 || k0 >= A.n_vf || ((A.gest >> k0) & 1u)) return;
        int m = 0;
        bool prefix = true;
#pragma unroll
        for (int h = 0; h < BLOCK_ENVS / 64; ++h) {
            const int i = 64 * h + lane;
            const uint64_t mk = __ballot(i < nb && s_ot[i] == k0);
            const int c = __popcll(mk);
            if (mk != (c == 64 ? ~0ull : ((1ull << c) - 1ull))) prefix = false;       // the chunk's k0 envs are its first c positions
            if (c > 0 && m != 64 * h) prefix = false;                                 // ... and every earlier chunk is full
            m += c;
        }
        if (prefix) { kB = k0; mB = m; }
    };
    // the root's update list geometry from the published actions (ballots; every wave that runs U1 under phase P derives it itself)
    auto u1_geometry = [&](uint64_t (&mk)[P_WAVES][NACT], int (&at)[P_WAVES]) {
#pragma unroll
        for (int h = 0; h < P_WAVES; ++h) at[h] = 64 * h + lane < nb ? (int)s_a[64 * h + lane] : -1;
        int off = 0;
#pragma unroll
        for (int a = 0; a < NACT; ++a) {
            int rl = 0, nbq = 0;
#pragma unroll
            for (int h = 0; h < P_WAVES; ++h) {
                mk[h][a] = __ballot(at[h] == a);
                rl += __popcll(mk[h][a]);
                const int lim = mB - 64 * h;                                      // positions below mB belong to option kB
                const uint64_t pre = lim >= 64 ? ~0ull : (lim > 0 ? ((1ull << lim) - 1ull) : 0ull);
                nbq += __popcll(mk[h][a] & pre);
            }
            run_len[a] = rl; run_off[a] = off; nBa[a] = nbq; off += rl;
        }
    };
    // With learning on, waves 8..15 ("helpers") have nothing to do in phase P: they stage W_0 (and W_kB), take Z_d^1 of
    // the entry states, build the root's update list and run U1 of both value functions under it.
    const bool helpers = MODE == MODE_FUSED && A.learn && A.k_hi >= 0;

    // ------------------------------------------------------------------ kernel start (round 5)
    // The first round trips of a launch are cold ones (4-6k cycles each). The old start — edge table -> barrier -> perm -> state
    // gathers on the env waves, barrier -> W_0 on the helpers — paid three of them in a row on the env waves' chain and two on the
    // helpers'. Now the ONE barrier of the start stands in front of every global access (it only covers the counters' reset), every
    // wave issues its first loads right behind it, and the edge / classifier tables — loaded by the waves without envs — are
    // handed over through an LDS counter where they are first needed (the physics), not through a barrier.
    if (tid < M_INTS) s_misc[tid] = (tid == M_PRESENT || tid == M_UPD) ? 1 : 0;
    block_lds_sync();
    SCG_LITE(1);                                            // behind the start barrier
    int e_pre = e0 + tid;                                   // env waves: this position's env
    if (MODE == MODE_FUSED) {
        if (wave < P_WAVES) {
            if (tid < nb && A.perm) e_pre = A.perm[e0 + tid];
        } else {
            // helper waves: W_0 goes to its place in region W (free during phase P) first. (Holding it in registers across a
            // wait instead — 14 per thread — sent the register allocator of the whole kernel over the edge: 27 spill / reload
            // sites in E and U2 instead of 2; so did the edge table as two predicated loads per thread.)
            if (helpers && wave >= HELPER0) stage_w(A.W, 0, tid - HELPER0 * 64, N_HELP * 64);
            for (int i = tid - P_WAVES * 64; i < A.ms.n_edges * 8; i += THREADS - P_WAVES * 64) s_edges[i] = A.edges[i];
            if (tid - P_WAVES * 64 < A.n_vf * CLF_STRIDE) s_clf[tid - P_WAVES * 64] = A.clf[tid - P_WAVES * 64];
            lds_arrive(&s_misc[M_C_TOP], 1);
        }
    }

    // ------------------------------------------------------------------ phase P
    if (helpers) { if (wave < P_WAVES) __builtin_amdgcn_s_setprio(SCG_PRIO_P); else if (wave >= HELPER0) __builtin_amdgcn_s_setprio(SCG_PRIO_HELP); }
    if (wave < P_WAVES) {                             // one lane per env on P_WAVES full waves
        const int i = tid;
        const bool valid = i < nb;
        const int e = e_pre;
        if (MODE == MODE_FUSED) {
            uint32_t u[4] = {0u, 0u, 0u, 0u};
            int a = NACT - 1, ep0 = 0, o = 0, o_in = 0, osteps = 0;
            float sx = 0.5f, sy = 0.5f, svx = 0.0f, svy = 0.0f;
            float qc[NACT] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
            if (valid) {
                // entry state first: the helper waves can start on it (Z_d^1, the block's option, W staging) while the action is drawn
                // (all of the env's gathers are issued together: one memory round trip behind the perm lookup)
                sx = A.x[e]; sy = A.y[e]; svx = A.vx[e]; svy = A.vy[e];
#pragma unroll
                for (int aa = 0; aa < NACT; ++aa) qc[aa] = A.qcache[(size_t)aa * N + e];
                ep0 = A.ep_steps[e]; o = A.option_id[e]; osteps = A.opt_steps[e];
                o_in = o;                              // (-k: inside option k's initiation set, staying out of it — SPEC §4.2)
                o = max(o, 0);
                s_s[0 * BLOCK_ENVS + i] = sx; s_s[1 * BLOCK_ENVS + i] = sy;
                s_s[2 * BLOCK_ENVS + i] = svx; s_s[3 * BLOCK_ENVS + i] = svy;
                s_ot[i] = (uint8_t)o;
            } else {
                s_ot[i] = 255;
            }
            lds_arrive(&s_misc[M_C_PUBS], 64);
            SCG_LITEP(1);                                         // entry state gathered and published
            SCG_STAMP(17);                                        // P: perm + state gathers
            if (valid) {
                // act (SPEC §2, §4.3)
                const uint64_t gid = (uint64_t)(A.env_base + e);
                philox4x32_10((uint32_t)gid, (uint32_t)(A.t & 0xffffffffu), (uint32_t)(A.t >> 32), 0u,
                              (uint32_t)(A.seed & 0xffffffffu), (uint32_t)(A.seed >> 32), u);
                const bool explore = (float)(u[0] >> 8) * 0x1p-24f < A.epsilon;
                const int a_rand = (int)__umulhi(u[1], 5u);
                int a_greedy = 0;
                float best = qc[0];
#pragma unroll
                for (int aa = 1; aa < NACT; ++aa) {
                    if (qc[aa] > best) { best = qc[aa]; a_greedy = aa; }
                }
                a = explore ? a_rand : a_greedy;
                s_a[i] = (uint8_t)a;
            } else {
                s_a[i] = 0;
            }
            SCG_STAMP(16);                                        // P: qcache gathers, Philox, action
            lds_arrive(&s_misc[M_C_PUB], 64);                                     // state, action and option id of this wave's envs are out
            SCG_LITEP(2);
            // physics (SPEC §1.3), the whole wave together
            bool goal;
            // the envs' own wave settles free flight and lists the (env, candidate edge) pairs of the others in groups of 64;
            // the groups of ALL env waves are then dealt to waves 0..P_POOL-1
            bool par;
            float *xs_mine = s_s + 4 * BLOCK_ENVS + wave * 64;
            lds_await(&s_misc[M_C_TOP], WAVES - P_WAVES);                         // the edge and classifier tables are in LDS
#ifdef SCG_DIAG_NO_PHYSICS         // (diagnostic, WRONG results: prices a step kernel WITHOUT phase P's physics at its head — VERDICT r4 item 1c)
            goal = false; par = false;                  // free flight through everything (20 fmas; keeps the batch moving so that the
            {                                           // option / action mix of the timed workload stays what it is with physics)
                const float DV = 0x1.99999ap-3f;
                svx = fminf(fmaxf(a == 0 ? svx + DV : (a == 2 ? svx - DV : svx), -2.0f), 2.0f);
                svy = fminf(fmaxf(a == 1 ? svy + DV : (a == 3 ? svy - DV : svy), -2.0f), 2.0f);
                for (int q = 0; q < 20; ++q) { sx = fmaf(svx, A.ms.hstep, sx); sy = fmaf(svy, A.ms.hstep, sy); }
                if (sx < 0.0f || sx > 1.0f) svx = -svx;
                if (sy < 0.0f || sy > 1.0f) svy = -svy;
                svx *= 0x1.fd70a4p-1f; svy *= 0x1.fd70a4p-1f;
                sx = fminf(fmaxf(sx, 0.0f), 1.0f); sy = fminf(fmaxf(sy, 0.0f), 1.0f);
            }
            const float rew = a == 4 ? -1.0f : -5.0f;
#else
            const int groups = pinball_wave_prepare_any(s_edges, A.cellmask, A.ms, valid, sx, sy, svx, svy, a, goal, par,
                                                        s_pitems + wave * PITEMS, xs_mine, BLOCK_ENVS);
            if (lane == 0) s_misc[M_GROUPS + wave] = groups;
            lds_arrive(&s_misc[M_C_PREP], 1);
            SCG_LITEP(3);                                         // own part of the physics done, pair groups listed
            SCG_STAMP(18);                                        // P: physics, own part (refinement, free flight, pair lists)
            {
                lds_await(&s_misc[M_C_PREP], P_WAVES);
                int gsum[P_WAVES + 1];
                gsum[0] = 0;
#pragma unroll
                for (int w2 = 0; w2 < P_WAVES; ++w2) gsum[w2 + 1] = gsum[w2] + s_misc[M_GROUPS + w2];
#ifdef SCG_STAMPS_LITE
                if (A.stamps && wave == 0 && lane == 0) A.stamps[(size_t)blockIdx.x * STAMP_SLOTS + 42] = (unsigned long long)gsum[P_WAVES];      // slot 42: the block's pair groups (last launch)
#endif
                // (a block has 5-7 groups as a rule and 8 in 3 % of the cases: a second round on wave 0 — those blocks are 72 % of the launches'
                //  slowest workgroups; handing the groups beyond the first round to two helper waves was built, bit-exact, and did not pay:
                //  profiles/r05_stragglers.txt, profiles/r05_pair_group_overflow.diff)
                for (int q = wave; q < gsum[P_WAVES]; q += P_POOL) {
                    int owner = 0, first = 0;
#pragma unroll
                    for (int w2 = 1; w2 < P_WAVES; ++w2) if (q >= gsum[w2]) { owner = w2; first = gsum[w2]; }
                    pinball_wave_group(s_edges, A.ms, s_pitems + owner * PITEMS + 64 * (q - first),
                                       s_s + 4 * BLOCK_ENVS + owner * 64, BLOCK_ENVS, s_ia + owner * 64);
                }
                lds_arrive(&s_misc[M_C_POOL], 1);
                lds_await(&s_misc[M_C_POOL], P_POOL);
            }
            const float rew = pinball_wave_finish(par, sx, sy, svx, svy, a, goal, xs_mine, BLOCK_ENVS, s_ia + wave * 64);
#endif
            SCG_LITEP(4);                                         // pooled pair groups done (everybody's), results read back
            SCG_STAMP(2);                                         // P: physics, the pooled pair groups + hand-offs
            int hkey = -1;
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
                    const unsigned par2 = (A.parents >> (3 * (o & 7))) & 7u;       // SPEC §4.2: target of option o
                    const bool succ = (par2 == 0) ? goal : ((inA >> par2) & 1u);
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
                    const unsigned par2 = (A.parents >> (3 * k)) & 7u;
                    if (par2 != 0 && ((inB >> par2) & 1u)) tgtB |= 1u << k;
                }
                const unsigned sel = inB & ~tgtB & A.enabled;         // a gestating option is never selected
                const int cand = keep ? o : (sel ? __builtin_ctz(sel) : 0);
                // SPEC §4.2: an env that stayed out of option k (option_id = -k) and still has k as its candidate is offered k again only
                // every reoffer_period-th step (staggered by env id; a new episode is a new offer) — in between it stays out without
                // a new comparison, i.e. without the option's value function being evaluated for it
                const int stay = (!keep && cand >= 1 && dn == 0 && o_in == -cand && (((uint32_t)A.t + (uint32_t)(A.env_base + e)) & A.reoffer_mask) != 0u) ? cand : 0;
                const int on = stay ? 0 : cand;
                s_on[i] = (uint8_t)on;
                s_gs[i] = (uint8_t)inS; s_ia[i] = (uint8_t)((inA & 0x3Eu) | (goal ? 1u : 0u) | ((!keep && on >= 1) ? IA_ENTERING : 0u));
                {   // bit masks of the VFs with items / update items here: OR over the wave's lanes first, then one LDS atomic per wave
                    const unsigned pm = (1u << (o & 31)) | (1u << (on & 31)) | inS, um = (1u << (o & 31)) | inS;
                    unsigned pw = 0, uw = 0;
#pragma unroll
                    for (int k = 0; k < MAX_VF + 1; ++k) {
                        if (__ballot((pm >> k) & 1u)) pw |= 1u << k;
                        if (__ballot((um >> k) & 1u)) uw |= 1u << k;
                    }
                    if (lane == __builtin_ctzll(__ballot(true))) {
                        atomicOr(reinterpret_cast<unsigned *>(&s_misc[M_PRESENT]), pw);
                        atomicOr(reinterpret_cast<unsigned *>(&s_misc[M_UPD]), uw);
                    }
                }
                if (inS && A.gest_succ) {                             // SPEC §4.4: gestation successes (integer counts: order-free)
#pragma unroll
                    for (int k = 1; k < MAX_VF; ++k) {
                        const unsigned par2 = (A.parents >> (3 * k)) & 7u;
                        if (((inS >> k) & 1u) && ((par2 == 0) ? goal : (bool)((inA >> par2) & 1u))) atomicAdd(&A.gest_succ[k], 1);
                    }
                }
                s_r0[i] = rew; s_c0[i] = dn ? 0.0f : A.gamma; s_ro[i] = ro; s_co[i] = co;
                // results -> staging record at this env's POSITION (full-line stores); commit_row scatters them
                // to the caller's arrays in env order
                {
                    const int osn = keep ? osteps + 1 : 0, epn = dn ? 0 : eps1;
                    float4 *orec = A.outrec + (size_t)(e0 + i) * OREC;
                    orec[0] = make_float4(nx, ny, nvx, nvy);
                    orec[1] = make_float4(rew, __uint_as_float((unsigned)a | ((unsigned)dn << 8) | ((unsigned)on << 16) | ((unsigned)stay << 28)),
                                          __int_as_float(osn), __int_as_float(epn));
                    reinterpret_cast<float *>(orec + 3)[1] = 0.0f;        // not declined (this lane alone writes the word: here and in gate())
                }
                SCG_LITEP(5);
                SCG_STAMP(19);                                        // P: bookkeeping, option logic, result line
                if (A.ring_x) {                                       // SPEC §7: trajectory ring + events
                    const size_t row = (size_t)(ep0 & A.ring_mask) * N + e;
                    A.ring_x[row] = s_s[0 * BLOCK_ENVS + i]; A.ring_y[row] = s_s[1 * BLOCK_ENVS + i];
                }
                if (A.events) { A.events[e] = (uint8_t)((goal ? 1u : 0u) | (inA & 0x3Eu)); A.ev_len[e] = eps1; }
                hkey = (e >> 8) * 8 + on;                             // next step's counting sort: (row of 256 envs, option id; a declined entry is moved to 0 by gate())
            } else {
                s_on[i] = 255; s_gs[i] = 0; s_ia[i] = 0;
            }
            if (A.hist_next) {
                // one atomic per distinct (row, option id) of the wave instead of one per env: neighbours in the env order are
                // neighbours in env id, so a wave holds about ten distinct keys — six times fewer atomics on 2048 hot counters, whose
                // acknowledgements every later s_waitcnt vmcnt of this wave has to sit out
                uint64_t rem = __ballot(hkey >= 0);
                while (rem) {
                    const int src = (int)__builtin_ctzll(rem);
                    const int k0 = __shfl(hkey, src, 64);
                    const uint64_t m = __ballot(hkey == k0);
                    if (lane == src) atomicAdd(&A.hist_next[k0], (int)__popcll(m));
                    rem &= ~m;
                }
            }
            SCG_LITEP(6);                                         // trace, events, histogram issued
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
#ifndef SCG_DIAG_NO_PHYSICS
        lds_await(&s_misc[M_C_PREP], P_WAVES);
        int gsum[P_WAVES + 1];
        gsum[0] = 0;
#pragma unroll
        for (int w2 = 0; w2 < P_WAVES; ++w2) gsum[w2 + 1] = gsum[w2] + s_misc[M_GROUPS + w2];
        for (int q = wave; q < gsum[P_WAVES]; q += P_POOL) {
            int owner = 0, first = 0;
#pragma unroll
            for (int w2 = 1; w2 < P_WAVES; ++w2) if (q >= gsum[w2]) { owner = w2; first = gsum[w2]; }
            pinball_wave_group(s_edges, A.ms, s_pitems + owner * PITEMS + 64 * (q - first),
                               s_s + 4 * BLOCK_ENVS + owner * 64, BLOCK_ENVS, s_ia + owner * 64);
        }
        lds_arrive(&s_misc[M_C_POOL], 1);
#endif
    } else if (helpers && wave >= HELPER0) {
        const int ht = tid - HELPER0 * 64, hw = wave - HELPER0;       // helper thread / wave index
        constexpr int NHT = N_HELP * 64;
#ifdef SCG_STAMPS
#define SCG_HSTAMP(SEC) do { if (ht == 0 && A.stamps) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); s_stamp[(SEC)] += (unsigned)(t_ - hprev); hprev = t_; } } while (0)
        unsigned long long hprev = stamp_prev;
#else
#define SCG_HSTAMP(SEC) do { } while (0)
#endif
        SCG_HSTAMP(10);
        lds_await(&s_misc[M_C_PUBS], 64 * P_WAVES);                                    // the P waves have published s and the option ids
        SCG_HSTAMP(11);
        decide_b();
        if (hw == 0 && lane == 0) { s_misc[M_KB] = kB; s_misc[M_MB] = mB; }
        if (kB >= 1) stage_w(A.W + (size_t)kB * NACT * NF, W_FLOATS, ht, NHT);
        // (round 5, measured and dropped: staging W_kB on a guess — the option of position 0 — before the states are published, and
        //  splitting the helper waves so that Z(s) + the list run beside that staging: both shorten the helpers' way to U1 by 3-6k
        //  cycles in the timing builds and neither moves the end of phase P — profiles/r05_td_kernel_ab_log.txt)
        for (int u = ht; u < BLOCK_ENVS * 2; u += NHT) {                               // Z_d^1 of the entry states, two variables per thread
            const int i = u & (BLOCK_ENVS - 1), h = u / BLOCK_ENVS;
            if (i < nb) {
                const float v0 = s_s[(2 * h) * BLOCK_ENVS + i], v1 = s_s[(2 * h + 1) * BLOCK_ENVS + i];
                const float2 za = sincospi_cs(h ? fmaf(v0, 0.25f, 0.5f) : v0), zb = sincospi_cs(h ? fmaf(v1, 0.25f, 0.5f) : v1);
                *reinterpret_cast<float4 *>(s_z1 + (i * 2 + 0) * 4 + 2 * h) = make_float4(za.x, za.y, zb.x, zb.y);
            }
        }
        // the root's update list (every env, one run per action, block order inside a run): each helper wave derives the
        // run geometry itself from ballots; helper wave 0 writes the list
        SCG_HSTAMP(12);
        lds_await(&s_misc[M_C_PUB], 64 * P_WAVES);                                     // ... and the actions
        SCG_HSTAMP(13);
        {
            uint64_t mk[P_WAVES][NACT];
            int at[P_WAVES];
            u1_geometry(mk, at);
            if (hw == 0) {
                const uint64_t below = (1ull << lane) - 1ull;
#pragma unroll
                for (int a = 0; a < NACT; ++a) {
                    int before = 0;
#pragma unroll
                    for (int h = 0; h < P_WAVES; ++h) {
                        if (at[h] == a) s_ulist[run_off[a] + before + __popcll(mk[h][a] & below)] = (uint16_t)(64 * h + lane);
                        before += __popcll(mk[h][a]);
                    }
                }
            }
        }
#ifdef SCG_TEST_DROP_ARRIVE      // (fault-injection build, tests/test_gpu_robustness.py: the REAL give-up path, once) helper wave 0 of block 1 never arrives at step t = 0x7e57
        if (!(hw == 0 && blockIdx.x == 1 && A.t == 0x7e57ull))
#endif
        lds_arrive(&s_misc[M_C_HELP], 1);
        lds_await(&s_misc[M_C_HELP], N_HELP);                                  // W_0, W_kB, Z(s) and the list are complete
        SCG_HSTAMP(14);
#ifdef SCG_DIAG_SIMD0_FREE      // (diagnostic, correct results) the helper waves that share SIMD 0 with env wave 0 leave U1 to the others
        if ((wave & 3) != 0)
#endif
        run_u1_dyn();
#ifdef SCG_STAMPS
        if (ht == 0 && A.stamps) s_stamp[28] += (unsigned)(__builtin_amdgcn_s_memtime() - stamp_prev);   // helper wave 0: start -> done
#endif
    }
    else if (MODE == MODE_FUSED && wave == HELPER0) {      // acting-only steps: this wave still settles the block's option
        lds_await(&s_misc[M_C_PUBS], 64 * P_WAVES);
        decide_b();
        if (lane == 0) { s_misc[M_KB] = kB; s_misc[M_MB] = mB; }
    }
#ifdef SCG_DIAG_NO_PHYSICS
    if (helpers && wave < HELPER0) {                    // (diagnostic) with no physics to run, waves 0..HELPER0-1 join the helpers' U1
        lds_await(&s_misc[M_C_HELP], N_HELP);
        mB = s_misc[M_MB];
        uint64_t mk[P_WAVES][NACT];
        int at[P_WAVES];
        u1_geometry(mk, at);
        run_u1_dyn();
    }
#endif
    if (helpers) { if ((unsigned)(wave - LIST0) < (unsigned)LIST_WAVES) __builtin_amdgcn_s_setprio(SCG_PRIO_LIST); else __builtin_amdgcn_s_setprio(0); }
    SCG_LITE(2);                                            // this wave's own phase-P work is done
    block_lds_sync();
    SCG_LITE(3);                                            // ... everybody's
    SCG_LITEP(7);

    SCG_STAMP(0);   // phase P
    // ------------------------------------------------------------------ phase Z (SPEC §3): Z_d^1 of s_next (and of s where no helper did it)
    for (int u = tid; u < BLOCK_ENVS * 8; u += THREADS) {     // thread -> (state sg, env i, variable d): four consecutive lanes write one env's 32 bytes
        const int i = (u >> 2) & (BLOCK_ENVS - 1), d = u & 3, sg = u / (4 * BLOCK_ENVS);      // (one lane per env and variable 64 bytes apart was a 32-way bank conflict)
        if (i < nb && (MODE != MODE_QVAL || sg == 1) && !(helpers && sg == 0)) {
            const float sv = s_s[(4 * sg + d) * BLOCK_ENVS + i];
            s_z1[(i * 2 + sg) * 4 + d] = sincospi_cs(d < 2 ? sv : fmaf(sv, 0.25f, 0.5f));
        }
    }
    if (MODE == MODE_FUSED && A.k_hi >= 0) {                // the block's option, settled by wave HELPER0 before the barrier
        kB = __builtin_amdgcn_readfirstlane(s_misc[M_KB]); mB = __builtin_amdgcn_readfirstlane(s_misc[M_MB]);
    } else { kB = -1; mB = 0; }

    // ------------------------------------------------------------------ phase TD (SPEC §3.1, §5) on the matrix pipe
    const unsigned present = (MODE == MODE_FUSED) ? (unsigned)__builtin_amdgcn_readfirstlane(s_misc[M_PRESENT]) : ~0u;
    const unsigned updm = (MODE == MODE_FUSED) ? (A.learn ? (unsigned)__builtin_amdgcn_readfirstlane(s_misc[M_UPD]) : 0u) : ~0u;
    // Passes: pass 0 = the merged pass (value function A = the root — or the caller's VF in the un-fused modes — and B = the
    // block's option, if any); then one single-VF pass per other value function with update items here. Value functions
    // that only have envs ENTERING them (no update item) skip the pass machinery: see the tail of the kernel.
    unsigned single = (MODE == MODE_FUSED && A.k_hi >= 0) ? (updm & present & ~1u & ~(kB >= 1 ? 1u << kB : 0u)) : 0u;
    const unsigned eval_only = (MODE == MODE_FUSED && A.k_hi >= 0) ? (present & ~updm & ~1u & ~(kB >= 1 ? 1u << kB : 0u)) : 0u;
    if (MODE == MODE_FUSED && tid < A.n_vf && A.cnts && tid != 0 && tid != kB && !((single >> tid) & 1u))
        A.cnts[(size_t)b * A.n_vf + tid] = 0;               // value functions without a pass here leave no slab
    const int n_pass0 = A.k_hi >= A.k_lo ? 1 : 0;
    const unsigned single0 = single;
    // SPEC §4.2 value-gated entry. An env about to enter option k (s_on, IA_ENTERING) does so only if the option promises at least
    // what the root does from s_next: max_a Q_k(s_next, a) >= max_a Q_0(s_next, a), both as E left them in LDS. kdec < 0: the
    // candidates evaluated in pass 0 (the block's option, value functions without a pass of their own); kdec = k: option k's own
    // single pass has just run. A declined env gets option 0 in the block's flags, the mark in its result line (commit_row then takes
    // the root's Q values the root's unit left beside the line) and its share of the next order's histogram moved to key 0.
    auto gate = [&](int kdec) {
        if (MODE != MODE_FUSED) return;
        int ln;                                             // (the thread id is re-made from the hardware lane counter: the kernel's own `tid`, kept
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));      //  live across the pass loop for this, is what the
        const int ti = (wave << 6) | ln;                    //  register allocator spills — see the top of the pass loop)
        if (ti >= nb || !(s_ia[ti] & IA_ENTERING)) return;
        const int cand = s_on[ti];
        const bool mine = kdec < 0 ? (cand >= 1 && cand < MAX_VF && !((single0 >> cand) & 1u)) : cand == kdec;
        if (!mine) return;
        const float cm = kdec < 0 ? s_maxq[BLOCK_ENVS + ti] : s_maxq[ti];
        const float rm = (kdec >= 0 || single0 != 0u) ? s_qsa[BLOCK_ENVS + ti] : s_maxq[ti];       // (single passes reuse s_maxq's first half: the first of them saved the root's column)
        if (cm >= rm) return;
        s_on[ti] = 0;
        reinterpret_cast<float *>(A.outrec + (size_t)(e0 + ti) * OREC + 3)[1] = __uint_as_float(OREC_DECLINED);
        if (A.hist_next) {
            const int e = A.perm ? A.perm[e0 + ti] : e0 + ti;
            atomicAdd(&A.hist_next[(e >> 8) * 8 + cand], -1);
            atomicAdd(&A.hist_next[(e >> 8) * 8], 1);
        }
    };
    // A hand-off poll of phase P that ran out leaves LDS data unpublished (lists, tables, the physics' results): the block goes inert —
    // no pass, no slab, counts 0 — instead of computing on them; the step as a whole is voided by the reduce launch
    const bool blk_fail = MODE == MODE_FUSED && __builtin_amdgcn_readfirstlane(s_misc[M_FAIL]) != 0;
    if (blk_fail && tid < A.n_vf && A.cnts) A.cnts[(size_t)b * A.n_vf + tid] = 0;
    for (int pass = 0; pass < n_pass0 + MAX_VF && !blk_fail; ++pass) {
        // Opaque copies of the thread and lane ids for the pass: every per-lane address of the pass body is loop-invariant, and
        // hoisted out of the pass loop they all stay live across it — the register allocator then spills them to scratch.
        // (Re-made from the hardware lane counter rather than copied from the kernel's tid: kept live across the loop the copy
        //  itself was spilled, and the reload's s_waitcnt vmcnt — in order — made the env waves wait here for every global
        //  store and atomic of phase P: 6-13k cycles in front of the pass.)
        int lane_p;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_p));
        const int tid_p = (wave << 6) | lane_p;
        const int tid = tid_p, lane = lane_p;
        int kA, kBp;                                        // value functions of this pass (kBp < 0: none)
        bool dense;                                         // E over position groups (pass 0) or over the compacted eval list
        if (pass == 0) {
            if (!n_pass0) continue;
            kA = MODE == MODE_FUSED ? 0 : A.k_lo; kBp = kB; dense = true;
        } else {
            if (!single) break;
            kA = __builtin_ctz(single); single &= single - 1; kBp = -1; dense = false;
        }
        const bool u1_done = helpers && pass == 0;          // the helper waves ran U1 and built the list under phase P
        if (pass > 0) block_lds_sync();                     // (pass 0 starts behind the barrier that ends phase P)
        if (MODE == MODE_FUSED && pass > 0 && kA == __builtin_ctz(single0)) {      // first single pass: save the root's max_a Q_0(s_next, a) of pass 0
            if (tid < nb) s_qsa[BLOCK_ENVS + tid] = s_maxq[tid];                 // before this pass's E reuses s_maxq (s_qsa's second half: B's Q(s, a), pass 0 only)
        }
        SCG_STAMP(pass == 0 ? 5 : 12);   // (diagnostic) wait at the pass's first barrier
        // ---- per-env flags of the pass (SPEC §5): ev bit v = the env needs Q_v(s_next, .) (bootstrap target and/or next
        // action); update items of A (all of them in pass 0 of a fused step: the root updates on every env) with their action
        // The flags and lists are made by waves LIST0 .. LIST0 + 3 (thread ft <-> position ft), not by the env waves 0..3: those still
        // have the global stores and atomics of phase P in flight, and any s_waitcnt vmcnt the compiler places in this stretch
        // (a register reload is enough) would make them — and, at the next barrier, everybody — wait for all of that to drain.
        const int ft = tid - LIST0 * 64, lw = wave - LIST0;
        bool evA = false, evB = false, up = false;
        int at = -1;
        if ((unsigned)ft < (unsigned)nb) {
            const int ot = s_ot[ft], on = s_on[ft];
            const bool own = (MODE == MODE_FUSED && kA == 0) || ot == kA;
            const bool gst = MODE == MODE_FUSED && !own && ((s_gs[ft] >> kA) & 1);       // SPEC §4.4 off-policy item
            up = (MODE != MODE_QVAL) && A.learn && (own || gst);
            float rk = s_r0[ft], cont = s_c0[ft];
            if (MODE == MODE_FUSED && kA != 0) {
                if (ot == kA) { rk = s_ro[ft]; cont = s_co[ft]; }
                else {                                    // as if the env ran option kA: no time-out, no selection
                    const unsigned ia = s_ia[ft], par2 = (A.parents >> (3 * kA)) & 7u;
                    const bool succ = (par2 == 0) ? (ia & 1u) : ((ia >> par2) & 1u);
                    const bool fail = !succ && !((ia >> kA) & 1u);
                    rk = rk + (succ ? A.r_succ : 0.0f);
                    cont = (cont == 0.0f || succ || fail) ? 0.0f : A.gamma;
                }
                s_rk[ft] = rk; s_ck[ft] = cont;
            }
            evA = (on == kA) || (up && cont > 0.0f) || (MODE == MODE_FUSED && kA == 0 && (s_ia[ft] & IA_ENTERING));      // (SPEC §4.2: the gate compares with the root)
            if (kBp >= 1) evB = (on == kBp) || (ot == kBp && A.learn && s_co[ft] > 0.0f);
            at = s_a[ft];
            s_ev[ft] = (uint8_t)((evA ? 1 : 0) | (evB ? 2 : 0));
        }
        // compacted list: single passes -> the eval items of A; pass 0 -> the envs that need Q_B but lie outside B's prefix groups
        const int pg_b = (mB + 7) >> 3;                     // position groups that hold B's prefix
        const bool cmp = dense ? (evB && (ft >> 3) >= pg_b) : evA;
        const bool eo_pass = pass == 0 && MODE == MODE_FUSED && eval_only != 0;
        const int on_me = (unsigned)ft < (unsigned)nb ? (int)s_on[ft] : 0;
        const bool eo = eo_pass && (unsigned)ft < (unsigned)nb && ((eval_only >> (on_me & 7)) & 1u);       // this env enters an evaluation-only value function
        uint64_t me[MAX_VF];
        if (ft == 0) s_misc[M_ECTR] = 0;
        uint64_t mb[1 + NACT];
        if ((unsigned)lw < (unsigned)LIST_WAVES) {
            mb[0] = __ballot(cmp);
#pragma unroll
            for (int k = 1; k < MAX_VF; ++k) me[k] = eo_pass ? __ballot(eo && on_me == k) : 0ull;
            if (lane >= 11 && lane < 10 + MAX_VF) {                                       // counts per value function: slots 11..15
                const int k = lane - 10;
                s_misc[M_CNT + lw * 16 + lane] = __popcll(k == 1 ? me[1] : k == 2 ? me[2] : k == 3 ? me[3] : k == 4 ? me[4] : me[5]);
            }
#pragma unroll
            for (int a = 0; a < NACT; ++a) mb[1 + a] = __ballot(up && at == a);
            const int lim = (kBp >= 1 ? mB : 0) - 64 * lw;                          // positions below mB belong to option kB
            const uint64_t pre = lim >= 64 ? ~0ull : (lim > 0 ? ((1ull << lim) - 1ull) : 0ull);
            if (lane < 1 + 2 * NACT) {
                const int sl = lane <= NACT ? lane : lane - NACT;
                const uint64_t mine = sl == 0 ? mb[0] : sl == 1 ? mb[1] : sl == 2 ? mb[2] : sl == 3 ? mb[3] : sl == 4 ? mb[4] : mb[5];
                s_misc[M_CNT + lw * 16 + lane] = __popcll(lane <= NACT ? mine : (mine & pre));
            }
            if (dense) {                                    // per position group: which value functions need it
                const uint64_t ma = __ballot(evA), mbb = __ballot(evB);
                if (lane < 8) {
                    const int pgrp = lw * 8 + lane;
                    s_eflag[pgrp] = (uint8_t)((((ma >> (8 * lane)) & 0xffull) ? 1 : 0) | ((((mbb >> (8 * lane)) & 0xffull) && pgrp < pg_b) ? 2 : 0));
                }
            }
        }
        SCG_STAMP(23);                                       // (diagnostic) flags + ballots
        if (!u1_done) {                                      // (the helper waves staged W_0 / W_kB under phase P)
            // (kA / kBp are wave-uniform: readfirstlane keeps the pointer arithmetic on the scalar unit)
            const float *Wa = A.W + (MODE == MODE_QVAL ? 0 : (size_t)__builtin_amdgcn_readfirstlane(kA) * NACT * NF);
            stage_w_cold(Wa, s_W, tid);
            if (kBp >= 1) stage_w_cold(A.W + (size_t)__builtin_amdgcn_readfirstlane(kBp) * NACT * NF, s_W + W_FLOATS, tid);
        }
        SCG_STAMP(24);                                       // (diagnostic) W staging
        block_lds_sync();
        SCG_STAMP(25);                                       // (diagnostic) wait at the barrier behind the staging
        int n_cmp = 0, nupdB = 0;
        {
            int se = 0;
#pragma unroll
            for (int w2 = 0; w2 < LIST_WAVES; ++w2) se += s_misc[M_CNT + w2 * 16];
            n_cmp = __builtin_amdgcn_readfirstlane(se);
            int off = 0;
#pragma unroll
            for (int a = 0; a < NACT; ++a) {      // wave-uniform: keep them in SGPRs (the helper waves hold the same values already)
                int sr = 0, sb = 0;
#pragma unroll
                for (int w2 = 0; w2 < LIST_WAVES; ++w2) { sr += s_misc[M_CNT + w2 * 16 + 1 + a]; sb += s_misc[M_CNT + w2 * 16 + 1 + NACT + a]; }
                run_len[a] = __builtin_amdgcn_readfirstlane(sr);
                nBa[a] = __builtin_amdgcn_readfirstlane(sb);
                run_off[a] = off;
                off += run_len[a];
                nupdB += nBa[a];
            }
        }
        const int nupd = run_off[NACT - 1] + run_len[NACT - 1];
        // compacted eval lists in s_elist: [0, n_cmp) the pass's own compacted items, then (pass 0) one list per evaluation-only
        // value function, every list starting at a multiple of 8 (a unit never mixes value functions)
        // (their geometry is worked out by the list waves and published through s_misc: ten more wave-uniform values held
        //  by all sixteen waves across the pass were ten more scalar registers spilled)
        int eo_cnt[MAX_VF], eo_base[MAX_VF], eo_units_l = 0;
#pragma unroll
        for (int k = 0; k < MAX_VF; ++k) { eo_cnt[k] = 0; eo_base[k] = 0; }
        if (eo_pass && (unsigned)lw < (unsigned)LIST_WAVES) {
            int base = (n_cmp + 7) & ~7;
#pragma unroll
            for (int k = 1; k < MAX_VF; ++k) {
                int c = 0;
#pragma unroll
                for (int w2 = 0; w2 < LIST_WAVES; ++w2) c += s_misc[M_CNT + w2 * 16 + 10 + k];
                eo_cnt[k] = __builtin_amdgcn_readfirstlane(c);
                eo_base[k] = base;
                base += (eo_cnt[k] + 7) & ~7;
                eo_units_l += (eo_cnt[k] + 7) >> 3;
            }
            if (lw == 0 && lane < MAX_VF) {
                const int k = lane;
                s_misc[M_EO + k] = k == 0 ? 0 : k == 1 ? eo_cnt[1] : k == 2 ? eo_cnt[2] : k == 3 ? eo_cnt[3] : k == 4 ? eo_cnt[4] : eo_cnt[5];
                s_misc[M_EO + 6 + k] = k == 0 ? eo_units_l : k == 1 ? eo_base[1] : k == 2 ? eo_base[2] : k == 3 ? eo_base[3] : k == 4 ? eo_base[4] : eo_base[5];
            }
        }
        if ((unsigned)lw < (unsigned)LIST_WAVES) {
            const uint64_t below = (1ull << lane) - 1ull;
            if (eo) {
                int off = zero_i();
                for (int w2 = 0; w2 < lw; ++w2) off += s_misc[M_CNT + w2 * 16 + 10 + on_me];
                const uint64_t mine = on_me == 1 ? me[1] : on_me == 2 ? me[2] : on_me == 3 ? me[3] : on_me == 4 ? me[4] : me[5];
                const int eb = on_me == 1 ? eo_base[1] : on_me == 2 ? eo_base[2] : on_me == 3 ? eo_base[3] : on_me == 4 ? eo_base[4] : eo_base[5];
                s_elist[eb + off + __popcll(mine & below)] = (uint16_t)ft;
            }
            if (cmp) {
                int off = zero_i();
                for (int w2 = 0; w2 < lw; ++w2) off += s_misc[M_CNT + w2 * 16];
                s_elist[off + __popcll(mb[0] & below)] = (uint16_t)ft;
            }
            if (up && !u1_done) {
                int off = zero_i();
                const uint64_t mine = at == 0 ? mb[1] : at == 1 ? mb[2] : at == 2 ? mb[3] : at == 3 ? mb[4] : mb[5];
                for (int w2 = 0; w2 < lw; ++w2) off += s_misc[M_CNT + w2 * 16 + 1 + at];
                s_ulist[sel5(run_off, at) + off + __popcll(mine & below)] = (uint16_t)ft;
            }
        }
        block_lds_sync();
        if (tid == 0 && A.cnts) {
            A.cnts[(size_t)b * A.n_vf + kA] = nupd;
            if (kBp >= 1) A.cnts[(size_t)b * A.n_vf + kBp] = nupdB;
        }
        SCG_STAMP(pass == 0 ? 1 : 8);    // phase Z (first pass only) + list build + W staging
        if (pass == 0) SCG_LITE(4);                         // E starts
        // ---- E: Q_v(s_next, .), one 8-item column block per wave-iteration (SPEC §3.1); the tables of a block are built
        // once and serve both value functions. Units [0, npg) are position groups (dense pass), the rest 8-item blocks of
        // the compacted list.
        const int npg = dense ? (nb + 7) >> 3 : 0;
        const int n_own = (n_cmp + 7) >> 3;                 // units of the pass's own compacted list
        const int eo_units = eo_pass ? __builtin_amdgcn_readfirstlane(s_misc[M_EO + 6]) : 0;
        const int n_units = npg + n_own + eo_units;
        if (n_units + nupd == 0) continue;
        {
#ifdef SCG_STAMPS
        const unsigned long long e_t0 = __builtin_amdgcn_s_memtime();
#endif
        // Dealing: units are taken from a counter in LDS by whichever wave is free (a static deal left the four top waves —
        // the youngest of their SIMDs — 15k cycles behind the others). Order: the compacted units first (the pass's own list,
        // then the evaluation-only value functions', which wait on memory for their operands and want LDS-fed units beside
        // them), then the position groups in order — the option's prefix (two evaluations each) before the groups behind it
        // (one): the big items first, the small ones fill the end. (Letting the position groups start before the lists are
        // written — an LDS counter instead of the barrier above — measured 750.6 against 749.6 M env-steps/s: not kept.)
        const int n_cu = n_own + eo_units;
        for (;;) {
            int it = 0;
            if (lane == 0) it = atomicAdd(&s_misc[M_ECTR], 1);
            it = __builtin_amdgcn_readfirstlane(it);
            if (it >= n_cu + npg) break;
            const bool du = it >= n_cu;
            const int u = du ? it - n_cu : it;
            int base, cnt, kg = -1;                          // kg >= 1: unit of evaluation-only value function kg (operands from memory)
            unsigned fl;
            if (du) { base = 8 * u; cnt = min(8, nb - base); fl = (unsigned)__builtin_amdgcn_readfirstlane((int)s_eflag[u]); }
            else if (u < n_own) { base = 8 * u; cnt = min(8, n_cmp - base); fl = dense ? 2u : 1u; }
            else {
                int c = u - n_own;
                base = 0; cnt = 0; fl = 1u;
#pragma unroll
                for (int k = 1; k < MAX_VF; ++k) {
                    const int ck = __builtin_amdgcn_readfirstlane(s_misc[M_EO + k]), uk = (ck + 7) >> 3;
                    if (kg < 0 && c < uk) { kg = k; base = __builtin_amdgcn_readfirstlane(s_misc[M_EO + 6 + k]) + 8 * c; cnt = min(8, ck - 8 * c); }
                    if (kg < 0) c -= uk;
                }
            }
            if (!fl) continue;
            SCG_LANE_ROLES();                               // (per unit: nothing of it lives across the loop)
            {
                const int j = min(bi, cnt - 1);
                build_tables(du ? base + j : (int)s_elist[base + j], 1, cp, bcol, cdk, abq);
            }
            wave_lds_sync();
            float B[9];
#pragma unroll
            for (int kb = 0; kb < 9; ++kb) B[kb] = cdk[(9 * g + kb) * 16 + n16];
            const bool have = ocol_item < cnt;
            const int il = du ? base + min(ocol_item, cnt - 1) : (int)s_elist[base + min(ocol_item, cnt - 1)];
            if (kg >= 1) {                                   // an evaluation-only value function: only the next action's values are needed
                float qo[NACT];
                int n16o = n16, go = g;                      // (opaque copies: the twelve per-lane row addresses are loop-invariant and
                asm volatile("" : "+v"(n16o), "+v"(go));     //  would otherwise be hoisted out of the unit loop and spilled)
                contract_g(A.W + (size_t)kg * NACT * NF, B, qo, n16o, go, ab_lane);
                if (out_lane && have) {
                    float4 *orec = A.outrec + (size_t)(e0 + il) * OREC;
                    orec[2] = make_float4(qo[0], qo[1], qo[2], qo[3]);
                    orec[3].x = qo[4];
                    float mx = qo[0];
#pragma unroll
                    for (int a = 1; a < NACT; ++a) mx = fmaxf(mx, qo[a]);
                    s_maxq[BLOCK_ENVS + il] = mx;              // SPEC §4.2: what the option promises (slot 1: the env has no item of B that reads it)
                }
                wave_lds_sync();
                continue;
            }
#pragma unroll 1
            for (int v = 0; v < 2; ++v) {
                if (!((fl >> v) & 1u)) continue;
                float qo[NACT];
                contract(v * W_FLOATS, B, qo, n16, g, w4, w8, ab_lane);
                if (out_lane && have && ((s_ev[il] >> v) & 1)) {
                    const int kv = v ? kBp : kA;
                    if (MODE == MODE_FUSED) {             // into the env's result line; commit_row writes qcache
                        if (kv == 0 && (s_ia[il] & IA_ENTERING)) {      // SPEC §4.2: the root's values, should the env stay with the root after all
                            float4 *qa = A.qalt + (size_t)(e0 + il) * 2;
                            qa[0] = make_float4(qo[0], qo[1], qo[2], qo[3]);
                            qa[1].x = qo[4];
                        }
                        if (s_on[il] == kv) {
                            float4 *orec = A.outrec + (size_t)(e0 + il) * OREC;
                            orec[2] = make_float4(qo[0], qo[1], qo[2], qo[3]);
                            orec[3].x = qo[4];
                        }
                    } else if (s_on[il] == kv) {
#pragma unroll
                        for (int a = 0; a < NACT; ++a) gstore(&A.qcache[(size_t)a * N + e0 + il], qo[a]);
                    }
                    float mx = qo[0];
#pragma unroll
                    for (int a = 1; a < NACT; ++a) mx = fmaxf(mx, qo[a]);
                    s_maxq[v * BLOCK_ENVS + il] = mx;
                }
            }
            wave_lds_sync();
        }
#ifdef SCG_STAMPS
        if (pass == 0 && MODE == MODE_FUSED && A.stamps && lane == 0) s_stamp[32 + wave] += (unsigned)(__builtin_amdgcn_s_memtime() - e_t0);
#endif
        }
        SCG_STAMP(pass == 0 ? 3 : 10);   // E (wave 0's share)
        // ---- U1 (pass 0 of a learning step: ran under phase P on the helper waves)
        if (MODE != MODE_QVAL && nupd > 0 && !u1_done) run_u1(wave, WAVES, n_units);   // dealt on behind E's blocks
        if (MODE == MODE_QVAL || nupd == 0) continue;
        SCG_STAMP(pass == 0 ? 4 : 11);   // U1 (wave 0's share)
        if (pass == 0) SCG_LITE(5);                         // this wave's E is done
        block_lds_sync();                                   // s_maxq, s_qsa cross waves; the staging area changes hands
        if (pass == 0) SCG_LITE(6);                         // ... everybody's: U2 starts
        SCG_STAMP(pass == 0 ? 7 : 14);   // wait for the other waves

        // ---- U2: the block partials (SPEC §5). Every action run of A's list is padded with null items to a multiple of 4
        // (groups of four items); run a has Ga groups, the first GBa of which also carry items of B. The groups are laid out in
        // chunks of <= U2_CH = 144 slots: one chunk if everything fits, else two, chunk 0 holding the first ceil(Ga / 2) groups of
        // EVERY run and chunk 1 the rest (sum_a ceil(Ga / 2) <= 36 groups = 144 slots for the <= 271 padded slots of 256 envs) — so
        // every wave has half of its products in either chunk (a chunk of consecutive slots held whole runs: the waves of two or
        // three actions multiplied while the others waited). The chains of SPEC §5 run per action in group order: unchanged.
        //   build  CDT[c34][kap] = CD, PT_A[c12][kap] = delta_A ABsel and, for B's groups, PT_B[c12][kap] = delta_B ABsel
        //          (kap = 2 slot + part; null items: +0); wave w owns slots 9 w .. 9 w + 8 of the chunk
        //   MFMA   G_v[a] += PT_v x CDT^T per group: one MFMA over the four real parts, one over the imaginary parts.
        //          Wave w < 15 owns the three output tiles (mi = w % 3, ni = 0..2) of action a = w / 3 of BOTH value functions:
        //          per group one P operand per value function and three C operands feed six (twelve) independent MFMAs;
        //          accumulators stay in registers for the whole pass and go straight to the block's slabs
        int lane_u = lane;
        asm volatile("" : "+v"(lane_u));                    // (lane roles derived afresh: see SCG_LANE_ROLES)
        const int n16 = lane_u & 15, g = lane_u >> 4;
        int Ga[NACT], GBa[NACT];
        int Gtot = 0;
#pragma unroll
        for (int a = 0; a < NACT; ++a) { Ga[a] = (run_len[a] + 3) >> 2; GBa[a] = (nBa[a] + 3) >> 2; Gtot += Ga[a]; }
        // chunk geometry: UCH slots per chunk, nch chunks, chunk c holding groups [c Ga / nch, (c + 1) Ga / nch) of EVERY run
        // (a double-buffered build — 72-slot chunks in two buffers, built by the wave halves in turn beside the products of the
        //  chunk before — measured 678 against 731 M env-steps/s: five chunks, six barriers, half-empty builder waves;
        //  profiles/r04_u2_double_buffer.diff)
        constexpr int UCH = U2_CH, NBUILD = WAVES;
        const int nch = 4 * Gtot <= UCH ? 1 : 2;
        constexpr int USX = 2 * UCH + 4;                    // row stride of the chunk tables (floats): 292 / 148 = 36 / 20 mod 64, operand reads conflict-free
        static_assert(3 * 36 * USX <= R_FLOATS && NBUILD * 9 == UCH, "chunk tables fit region R");
        int wave_u = wave;
        asm volatile("" : "+s"(wave_u));                    // keeps the per-wave tile geometry inside the pass
        const bool haveB = nupdB > 0;
        const bool worker = wave_u < 3 * NACT;              // waves 0..14 multiply, wave 15 only builds
        const int ja = (wave_u * 11) >> 5, jm = wave_u - 3 * ja;              // this wave's action (w / 3) and row tile (w % 3)
        const int j_GB = sel5(GBa, ja), j_nB = sel5(nBa, ja);
        f4v accU[2][3];
#pragma unroll
        for (int v = 0; v < 2; ++v) {
#pragma unroll
            for (int s = 0; s < 3; ++s) accU[v][s] = zero4();
        }
        const float *rA = (MODE == MODE_FUSED && kA != 0) ? s_rk : s_r0, *cA = (MODE == MODE_FUSED && kA != 0) ? s_ck : s_c0;
        const int bi9 = lane_u % 9, cp9 = lane_u / 9;       // builder lanes of a chunk: (slot 9 w + bi9, second index cp9 < 6)
        // this chunk's share of every run: groups [gb[a], gb[a] + gc[a]) at slots [co[a], co[a + 1])
        auto chunk_geo = [&](int ch, int (&gb)[NACT], int (&gc)[NACT], int (&co)[NACT + 1]) {
            co[0] = 0;
#pragma unroll
            for (int a = 0; a < NACT; ++a) {
                const int half = Ga[a] >> 1;                            // chunk 0 of two: the first floor(Ga / 2) groups, chunk 1 the rest
                const int lo = ch == 0 ? 0 : half, hi = (nch == 1 || ch == 1) ? Ga[a] : half;
                gb[a] = lo; gc[a] = hi - lo;
                co[a + 1] = co[a] + 4 * gc[a];
            }
        };
        auto u2_build = [&](int ch, float *tab, int bw /* this wave's index among the chunk's builders, < 0: none */) {
            int gb[NACT], gc[NACT], co[NACT + 1];
            chunk_geo(ch, gb, gc, co);
            float *ptabA = tab, *ptabB = tab + 36 * USX, *ctab = tab + 2 * 36 * USX;
            const int slot = 9 * bw + bi9;
            if (bw >= 0 && cp9 < 6 && slot < co[NACT]) {
                int a_ = 0;
#pragma unroll
                for (int a = 1; a < NACT; ++a) a_ += slot >= co[a] ? 1 : 0;
                const int rl = sel5(run_len, a_), ro = sel5(run_off, a_), nbq = sel5(nBa, a_), l4b = 4 * sel5(GBa, a_);
                const int j = 4 * sel5(gb, a_) + slot - sel5(co, a_);     // slot of the padded run
                float *pdA = ptabA + cp9 * USX + 2 * slot, *pdB = ptabB + cp9 * USX + 2 * slot, *cdst = ctab + cp9 * USX + 2 * slot;
                if (j < rl) {
                    const int li = ro + j, il = s_ulist[li];
                    const float rr = rA[il], cont = cA[il];
                    // SPEC §4.2 exit rule (single passes of an option): the option ended (continuation 0) but the episode goes on -> the
                    // root's value of where the env goes next (kept in s_qsa's second half by the first single pass)
                    const bool boot = MODE == MODE_FUSED && kA != 0 && cont == 0.0f && s_c0[il] > 0.0f;
                    const float target = boot ? fmaf(A.gamma, s_qsa[BLOCK_ENVS + il], rr) : (cont > 0.0f ? fmaf(cont, s_maxq[il], rr) : rr);
                    const float d = target - s_qsa[li];
                    float2 ab[6], cd[6];
                    item_entries(s_z1 + (il * 2 + 0) * 4, cp9, ab, cd);
#pragma unroll
                    for (int c = 0; c < 6; ++c) {
                        *reinterpret_cast<float2 *>(pdA + 6 * c * USX) = make_float2(d * ab[c].x, d * (-ab[c].y));
                        *reinterpret_cast<float2 *>(cdst + 6 * c * USX) = make_float2(cd[c].x, cd[c].y);
                    }
                    if (j < nbq) {                           // the item also updates value function B
                        const float rb = s_ro[il], cb2 = s_co[il];
                        const float tb = (cb2 == 0.0f && s_c0[il] > 0.0f) ? fmaf(A.gamma, s_maxq[il], rb)          // (exit rule: s_maxq[il] is the root's max in pass 0)
                                                                         : (cb2 > 0.0f ? fmaf(cb2, s_maxq[BLOCK_ENVS + il], rb) : rb);
                        const float db = tb - s_qsa[BLOCK_ENVS + li];
#pragma unroll
                        for (int c = 0; c < 6; ++c)
                            *reinterpret_cast<float2 *>(pdB + 6 * c * USX) = make_float2(db * ab[c].x, db * (-ab[c].y));
                    } else if (j < l4b) {                    // null item of B's run (its C operand is masked in the MFMA loop)
#pragma unroll
                        for (int c = 0; c < 6; ++c) *reinterpret_cast<float2 *>(pdB + 6 * c * USX) = make_float2(0.0f, 0.0f);
                    }
                } else {                                     // null item padding a run to a multiple of 4
#pragma unroll
                    for (int c = 0; c < 6; ++c) {
                        *reinterpret_cast<float2 *>(pdA + 6 * c * USX) = make_float2(0.0f, 0.0f);
                        *reinterpret_cast<float2 *>(cdst + 6 * c * USX) = make_float2(0.0f, 0.0f);
                    }
                    if (j < l4b) {
#pragma unroll
                        for (int c = 0; c < 6; ++c) *reinterpret_cast<float2 *>(pdB + 6 * c * USX) = make_float2(0.0f, 0.0f);
                    }
                }
            }
        };
        auto u2_mfma = [&](int ch, const float *tab) {
            if (!worker) return;
#ifdef SCG_DIAG_FREE_BORDER         // (diagnostic, WRONG results: prices U2's five border tiles at zero — the ceiling of "border tiles on the vector pipe")
            if (jm == 2) return;
#endif
            int gb[NACT], gc[NACT], co[NACT + 1];
            chunk_geo(ch, gb, gc, co);
            const float *ptabA = tab, *ctab = tab + 2 * 36 * USX;                // (PT_B lies between them)
            const int ngrp = sel5(gc, ja), g0 = sel5(gb, ja), so = sel5(co, ja);      // run ja's groups in this chunk, from group g0, at slot so
            if (ngrp <= 0) return;
            const int ngrpB = haveB ? min(max(j_GB - g0, 0), ngrp) : 0;          // ... of which B's
            const int prow = min(16 * jm + n16, 35) * USX + 2 * g + 2 * so;
            const float *paA = ptabA + prow;                                     // (B's P operand: the same row of ptabB = + 36 USX floats)
            const float *pb0 = ctab + n16 * USX + 2 * g + 2 * so,                // (the second C operand: + 16 USX floats)
                        *pb2 = ctab + min(32 + n16, 35) * USX + 2 * g + 2 * so;
            const int nvalid = j_nB - 4 * g0;                                    // B items of the run still real from group g0 on
            // The no-op is 70-80 % of the items: its three waves walk ~24 groups per chunk, alone on their SIMDs (the other waves are
            // through in a tenth of the time and wait at the barrier), so nothing hides the operands' LDS round trip — 110 of 300 ticks per
            // group (tools/u2_loop_probe.hip). Hence: straight-line loops (the groups that carry items of B — the one group that can hold
            // null items peeled off —, then the rest) on two operand sets, the NEXT group's operands fetched before this group's products
            // are issued; a fetch past the run's end reads inside the chunk tables.
            // The whole walk is ONE asm statement on fixed operand registers (v108..v127), accumulators tied (vdst = srcC). As builtins the
            // register allocator kept two copies of every accumulator round the loops: per iteration of 12 products `s_nop 6` (the copies
            // read MFMA results: the pipe drains), 10-18 v_mov_b64 on the pipe the products use and two taken branches — the 1.35-1.8x
            // between this loop in the kernel and in the probe (profiles/r05_u2_anatomy.txt item 5; found by reading the ISA). The compiler's
            // hazard recognizer does not see inside asm: dependent products are three apart on one opcode and one vdst (what the compiler
            // emits itself), operands are re-fetched only behind the products that read them (an LDS return is > 64 cycles away, sources are
            // read at issue), and the wait states a VALU / VMEM read of a result needs are issued once, at the end.
#define U2_LDS(P) ((unsigned)(size_t)(const __attribute__((address_space(3))) void *)(P))
            unsigned aP = U2_LDS(paA), a0 = U2_LDS(pb0), a2_ = U2_LDS(pb2);          // B's P operand (paB) and the second C operand (pb1) lie at fixed offsets from these
            f4v x0 = accU[0][0], x1 = accU[0][1], x2 = accU[0][2], y0 = accU[1][0], y1 = accU[1][1], y2 = accU[1][2];
            const int nBfull = ngrpB > 0 && nvalid < 4 * ngrpB ? ngrpB - 1 : ngrpB;      // B's groups of four REAL items (only its last group can hold null items)
            const int nAonly = ngrp - ngrpB;
            const unsigned long long nullmask = __ballot(4 * nBfull + g >= nvalid);     // lanes whose item of B's last group is a null item: C operand +0 (SPEC §5)
            unsigned cnt;
            // operand set 1: P v[100:101], C0 v[102:103], C1 v[104:105], C2 v[106:107], B's P v[108:109]; set 2: v[110:111] .. v[118:119]
#define MF(ACC, A_, B_) "v_mfma_f32_16x16x4_f32 %[" ACC "], " A_ ", " B_ ", %[" ACC "]\n\t"
#define SX1 MF("x0", "v102", "v100") MF("x1", "v104", "v100") MF("x2", "v106", "v100") MF("x0", "v103", "v101") MF("x1", "v105", "v101") MF("x2", "v107", "v101")
#define SY1 MF("y0", "v102", "v108") MF("y1", "v104", "v108") MF("y2", "v106", "v108") MF("y0", "v103", "v109") MF("y1", "v105", "v109") MF("y2", "v107", "v109")
#define SX2 MF("x0", "v112", "v110") MF("x1", "v114", "v110") MF("x2", "v116", "v110") MF("x0", "v113", "v111") MF("x1", "v115", "v111") MF("x2", "v117", "v111")
#define SY2 MF("y0", "v112", "v118") MF("y1", "v114", "v118") MF("y2", "v116", "v118") MF("y0", "v113", "v119") MF("y1", "v115", "v119") MF("y2", "v117", "v119")
#define SYP MF("y0", "v112", "v108") MF("y1", "v114", "v108") MF("y2", "v116", "v108") MF("y0", "v113", "v109") MF("y1", "v115", "v109") MF("y2", "v117", "v109")
#define F1A(O) "ds_read_b64 v[100:101], %[aP] offset:" O "\n\tds_read_b64 v[102:103], %[a0] offset:" O "\n\tds_read_b64 v[104:105], %[a0] offset:%[o1]+" O "\n\tds_read_b64 v[106:107], %[a2] offset:" O "\n\t"
#define F2A(O) "ds_read_b64 v[110:111], %[aP] offset:" O "\n\tds_read_b64 v[112:113], %[a0] offset:" O "\n\tds_read_b64 v[114:115], %[a0] offset:%[o1]+" O "\n\tds_read_b64 v[116:117], %[a2] offset:" O "\n\t"
#define F1Q(O) "ds_read_b64 v[108:109], %[aP] offset:%[oq]+" O "\n\t"
#define F2Q(O) "ds_read_b64 v[118:119], %[aP] offset:%[oq]+" O "\n\t"
#define LW "s_waitcnt lgkmcnt(0)\n\t"
#define BUMP(N) "v_add_u32_e32 %[aP], " N ", %[aP]\n\tv_add_u32_e32 %[a0], " N ", %[a0]\n\tv_add_u32_e32 %[a2], " N ", %[a2]\n\t"
#define ZC(D, S) "v_cndmask_b32_e64 " D ", " S ", 0, %[nm]\n\t"
            asm volatile(
                "s_cmp_eq_u32 %[nBall], 0\n\ts_cbranch_scc1 10f\n\t"
                // ---- groups with items of both value functions: invariant — set 1 holds the landed operands of the next group
                F1A("0") F1Q("0") LW
                "s_lshr_b32 %[cnt], %[nB], 2\n\ts_cmp_eq_u32 %[cnt], 0\n\ts_cbranch_scc1 2f\n"
                "1:\n\t" F2A("32") F2Q("32") SX1 SY1 LW F1A("64") F1Q("64") SX2 SY2 LW F2A("96") F2Q("96") SX1 SY1 LW F1A("128") F1Q("128") SX2 SY2 LW BUMP("0x80")
                "s_sub_u32 %[cnt], %[cnt], 1\n\ts_cmp_lg_u32 %[cnt], 0\n\ts_cbranch_scc1 1b\n"
                "2:\n\ts_bitcmp1_b32 %[nB], 1\n\ts_cbranch_scc0 3f\n\t" F2A("32") F2Q("32") SX1 SY1 LW F1A("64") F1Q("64") SX2 SY2 LW BUMP("64")
                "\n3:\n\ts_bitcmp1_b32 %[nB], 0\n\ts_cbranch_scc0 4f\n\t" SX1 SY1 F1A("32") F1Q("32") LW BUMP("32")
                // B's last group with null items: the C operands of the null lanes are +0 for B's products only
                "\n4:\n\ts_cmp_eq_u32 %[nBall], %[nB]\n\ts_cbranch_scc1 11f\n\t" SX1
                ZC("v112", "v102") ZC("v113", "v103") ZC("v114", "v104") ZC("v115", "v105") ZC("v116", "v106") ZC("v117", "v107") SYP F1A("32") LW BUMP("32")
                "s_branch 11f\n"
                "10:\n\ts_cmp_eq_u32 %[nA], 0\n\ts_cbranch_scc1 20f\n\t" F1A("0") LW
                // ---- the rest of the run: the block's own value function only
                "\n11:\n\ts_lshr_b32 %[cnt], %[nA], 2\n\ts_cmp_eq_u32 %[cnt], 0\n\ts_cbranch_scc1 13f\n"
                "12:\n\t" F2A("32") SX1 LW F1A("64") SX2 LW F2A("96") SX1 LW F1A("128") SX2 LW BUMP("0x80")
                "s_sub_u32 %[cnt], %[cnt], 1\n\ts_cmp_lg_u32 %[cnt], 0\n\ts_cbranch_scc1 12b\n"
                "13:\n\ts_bitcmp1_b32 %[nA], 1\n\ts_cbranch_scc0 14f\n\t" F2A("32") SX1 LW F1A("64") SX2 LW BUMP("64")
                "\n14:\n\ts_bitcmp1_b32 %[nA], 0\n\ts_cbranch_scc0 20f\n\t" SX1
                "\n20:\n\ts_nop 15\n\ts_nop 7"
                : [x0] "+v"(x0), [x1] "+v"(x1), [x2] "+v"(x2), [y0] "+v"(y0), [y1] "+v"(y1), [y2] "+v"(y2),
                  [aP] "+v"(aP), [a0] "+v"(a0), [a2] "+v"(a2_), [cnt] "=&s"(cnt)
                : [nB] "s"(nBfull), [nBall] "s"(ngrpB), [nA] "s"(nAonly), [nm] "s"(nullmask), [oq] "i"(36 * USX * 4), [o1] "i"(16 * USX * 4)
                : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116",
                  "v117", "v118", "v119", "scc");
#undef MF
#undef SX1
#undef SY1
#undef SX2
#undef SY2
#undef SYP
#undef F1A
#undef F2A
#undef F1Q
#undef F2Q
#undef LW
#undef BUMP
#undef ZC
#undef U2_LDS
            accU[0][0] = x0; accU[0][1] = x1; accU[0][2] = x2; accU[1][0] = y0; accU[1][1] = y1; accU[1][2] = y2;
        };
        SCG_LITEU(1);                                            // (light stamps, variant 3) U2 starts
        for (int ch = 0; ch < nch; ++ch) {
            if (ch > 0) { block_lds_sync(); SCG_LITEU(4); }                   // previous chunk's operands consumed
            SCG_STAMP(20);                                                    // (diagnostic) U2: MFMAs of the previous chunk + wait
            u2_build(ch, s_R, wave);
            SCG_STAMP(21);                                       // (diagnostic) U2: build
            block_lds_sync();                                    // operands visible
            SCG_STAMP(22);                                       // (diagnostic) U2: wait for the other waves' build
            if (ch == 0) SCG_LITEU(2); else SCG_LITEU(5);
            u2_mfma(ch, s_R);
            if (ch == 0) SCG_LITEU(3); else SCG_LITEU(6);
            SCG_STAMP(9);                                        // (diagnostic) U2: this wave's own products of the chunk
        }
        SCG_STAMP(pass == 0 ? 6 : 13);   // U2
        // the block partials straight from the accumulators (zeros for an empty run). The tiles were accumulated
        // TRANSPOSED (A operand = CDT rows, B operand = PT rows; fma(a, b, c) = fma(b, a, c)), so register v of lane (n16, g)
        // of tile (mi, ni) is G[a][c12 = 16 mi + n16][c34 = 16 ni + 4 g + v]: one 16-byte store per lane and tile
        if (!s_misc[M_FAIL]) {
            if (worker) {
#pragma unroll
                for (int v = 0; v < 2; ++v) {
                    if (v == 1 && !haveB) break;
                    const int kv = v ? kBp : kA;
                    float *slab_lane = A.slabs + ((size_t)b * A.n_vf + kv) * NACT * NF + ja * NF + (16 * jm + n16) * 36 + 4 * g;
#pragma unroll
                    for (int ni = 0; ni < 3; ++ni) {
                        const bool okl = (jm < 2 || n16 < 4) && (ni < 2 || g == 0);
                        if (okl) store_wt(slab_lane + 16 * ni, accU[v][ni]);
                    }
                }
            }
        } else if (tid == 0 && A.cnts) {
            A.cnts[(size_t)b * A.n_vf + kA] = 0;
            if (kBp >= 1) A.cnts[(size_t)b * A.n_vf + kBp] = 0;
        }
        SCG_STAMP(15);                // slab stores issued
        if (pass == 0) SCG_LITE(7);                         // pass 0 done (slab stores issued)
        if (pass > 0) gate(kA);
    }
    if (MODE == MODE_FUSED && n_pass0 && !blk_fail) {       // the candidates evaluated in pass 0 (their maxima are in s_maxq's second half, which
        if (!A.learn) block_lds_sync();                     // no later pass touches; acting-only steps: pass 0 ended behind E, without a barrier)
        gate(-1);
    }
    if (MODE == MODE_FUSED && A.async_word) {               // a hand-off poll ran out somewhere in this block: tell the host (sticky)
        block_lds_sync();
        if (tid == 0 && s_misc[M_FAIL]) {
            __hip_atomic_fetch_or(A.async_word, SCG_ASYNC_STEP_HANDOFF, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (A.fail_flag) __hip_atomic_store(A.fail_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
#ifdef SCG_STAMPS_LITE
    if (MODE == MODE_FUSED && A.stamps && lite_role >= 0 && lane == 0) {
        unsigned long long *o = A.stamps + (size_t)blockIdx.x * STAMP_SLOTS;
#pragma unroll
        for (int i = 1; i < 8; ++i) o[lite_role * 8 + i] += lt[i] ? lt[i] - lt[0] : 0ull;
        o[lite_role * 8] += __builtin_amdgcn_s_memtime() - lt[0];                 // [0]: the wave's whole kernel
        if (lite_role == 0) { o[32] = lite_r0; o[33] = __builtin_amdgcn_s_memrealtime(); }
    }
    if (MODE == MODE_FUSED && A.stamps && lane == 0) {      // slots 34..41: HW_REG_HW_ID of each of the 16 waves (last launch): where the dispatcher put wave w (tools/wave_placement.py)
        unsigned hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        reinterpret_cast<unsigned *>(A.stamps + (size_t)blockIdx.x * STAMP_SLOTS + 34)[wave] = hwid;
    }
#endif
#ifdef SCG_STAMPS
    if (MODE == MODE_FUSED && A.stamps) {
        __syncthreads();
        if (tid < STAMP_SLOTS) A.stamps[(size_t)blockIdx.x * STAMP_SLOTS + tid] += s_stamp[tid];
    }
#endif
}
