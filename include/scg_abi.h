/*
 * scg_abi.h — C-ABI of libscg_hip.so, the MI355X (gfx950) hot path of vectorized skill chaining.
 *
 * Reference interface replaced: BASELINE.json's north_star names PinballDomain.step,
 * FourierBasis.features, Option.{policy,beta,initiation_classifier} and SkillChainingAgent.q_update,
 * but the reference at /root/reference is README.md:1-2 only (title + one sentence naming Konidaris &
 * Barto 2009) — it has no code, hence no FFI to bind and no file:line to cite for any entry point
 * (SURVEY.md §0, §8a/b). The signatures below are therefore this build's own; each comment names the
 * north_star symbol the entry point stands in for and the SPEC.md section that defines its arithmetic.
 *
 * Conventions
 *  - plain C, no torch types. All array arguments are DEVICE pointers owned by the caller (e.g.
 *    torch.Tensor.data_ptr()) unless marked HOST. `stream` is a hipStream_t passed as void*
 *    (NULL = default stream). Launches are asynchronous on `stream`; nothing here synchronises.
 *  - every function returns 0 on success or a negative scg_status; scg_last_error(ctx) gives text.
 *    No exceptions cross the boundary.
 *  - one ctx per (device, stream of use); calls on one ctx are not re-entrant. The library owns only
 *    the ctx: map tables, the per-block partial-gradient slabs and the reduced gradient.
 *  - state is SoA: x[N], y[N], vx[N], vy[N] float32.
 */
#ifndef SCG_ABI_H
#define SCG_ABI_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SCG_ABI_VERSION 5          /* 2: SCG_ASYNC_STEP_HANDOFF, announced-trigger pointer check, 256-env blocks; 3: scg_apply_update_slots;
                                      4: a second update rule (SCG_STEP_CACHED_QSA + a baseline cache); 5: that rule and its two entry
                                      points are gone again (its definition depended on the launch geometry and it was never the
                                      default), SPEC §4.2's exit rule (an option that ends bootstraps from the root) */
#define SCG_NUM_ACTIONS 5
#define SCG_FOURIER_ORDER 5
#define SCG_NUM_FEATURES 1296      /* (order+1)^4 */
#define SCG_MAX_OPTIONS 5
#define SCG_MAX_EDGES 256
#define SCG_CLF_STRIDE 8           /* floats per classifier row (6 used) */

typedef enum {
    SCG_OK = 0,
    SCG_ERR_INVALID = -1,      /* bad argument / unsupported configuration */
    SCG_ERR_NO_DEVICE = -2,    /* no HIP device / wrong architecture */
    SCG_ERR_HIP = -3,          /* a HIP runtime call failed */
    SCG_ERR_STATE = -4,        /* call order (e.g. step before set_map) */
    SCG_ERR_ASYNC = -5         /* a kernel launched earlier gave up on the device (sticky; see "asynchronous failures") */
} scg_status;

typedef struct scg_ctx scg_ctx;

typedef struct {
    int32_t n_envs;              /* envs on this rank */
    int32_t n_options;           /* chained options 1..n_options (<= SCG_MAX_OPTIONS); VF 0 = root */
    int32_t fourier_order;       /* must be SCG_FOURIER_ORDER */
    int32_t device;              /* HIP device ordinal */
    int64_t env_id_base;         /* global id of local env 0 (SPEC §2) */
    uint64_t seed;
    float gamma, alpha, epsilon, r_option_success;
    int32_t max_episode_steps, max_option_steps;
    int32_t update_count_floor;   /* SPEC §5 apply: the divisor of a value function's summed update is max(n_k, floor). 0 = plain n_k. A value
                                   * function with a handful of update items otherwise takes full-size steps on their average and runs away
                                   * (profiles/r05_oracle_chain_curves_gated.txt: NaN weights in an option few envs run); n_envs / 16 is a safe choice */
    int32_t reoffer_period;       /* SPEC §4.2: an env that stays out of option k although inside its initiation set (option_id = -k) is offered k
                                   * again when (t + global env id) % reoffer_period == 0. A power of two; 1 (or 0) = every step */
} scg_config;

/* flags for scg_step */
#define SCG_STEP_LEARN 1u        /* accumulate the TD gradient (otherwise act + physics + qcache only) */
#define SCG_STEP_APPLY 2u        /* apply it to W in the same call (single-rank path) */

int scg_abi_version(void);
int scg_block_envs(void);            /* SPEC §5 block size this library was built with (one 16-wavefront workgroup per block): 256 for libscg_hip.so;
                                        libscg_hip_b128.so / _b64.so are the same source built for 128 / 64 (small env counts). The block size
                                        orders the partial sums of G: a sharded run uses ONE geometry on all ranks */
const char *scg_strerror(int status);
const char *scg_last_error(const scg_ctx *ctx);

int scg_create(scg_ctx **out, const scg_config *cfg);
int scg_destroy(scg_ctx *ctx);
/* change hyper-parameters between steps (n_envs / n_options / device are fixed at create) */
int scg_set_hparams(scg_ctx *ctx, float gamma, float alpha, float epsilon, float r_option_success,
                    int32_t max_episode_steps, int32_t max_option_steps, int32_t update_count_floor, int32_t reoffer_period);

/* SPEC §1.1. All HOST pointers, copied. edges[n_edges][8], starts[n_starts][2], scale[1296].
 * map_scalars = {R, hstep, R2, TX, TY, TR2}. */
int scg_set_map(scg_ctx *ctx, const float *edges, int32_t n_edges, const float *starts, int32_t n_starts,
                const float map_scalars[6], const float *scale);

/* The fused step-batch: Option.policy (act from qcache) -> PinballDomain.step -> reset/bookkeeping ->
 * Option.beta / option selection -> FourierBasis.features -> Q -> SkillChainingAgent.q_update
 * (SPEC §1.3-§5). In/out: x,y,vx,vy, option_id, opt_steps, ep_steps, qcache[5][N].
 * Out: action[N] u8, reward[N] f32, done[N] u8 (0 live, 1 goal, 2 time-limit).
 * W[n_vf][5][1296] (updated iff SCG_STEP_APPLY), clf[n_vf][8] (row 0 unused), enabled bit k = option k.
 * t = step counter (RNG key). */
int scg_step(scg_ctx *ctx, float *x, float *y, float *vx, float *vy, int32_t *option_id,
             int32_t *opt_steps, int32_t *ep_steps, float *qcache, uint8_t *action, float *reward,
             uint8_t *done, float *W, const float *clf, uint32_t enabled_mask, uint64_t t,
             uint32_t flags, void *stream);

/* Device pointers of the ctx-owned reduced gradient G[n_vf][5][1296] and counts n_k[n_vf] (int32)
 * left by the last scg_step(LEARN) — the buffers a multi-rank caller all-reduces (SPEC §5). */
int scg_grad_buffers(scg_ctx *ctx, float **G, int32_t **n_k);
/* Make scg_step / scg_q_update leave G and n_k in caller-owned device buffers instead
 * (G[n_vf][5][1296] float32, n_k[n_vf] int32); NULL, NULL restores the ctx-owned ones. */
int scg_set_grad_buffers(scg_ctx *ctx, float *G, int32_t *n_k);
/* Apply (possibly all-reduced) G / n_k to W: W_k += alpha/n_k * scale * G_k (SPEC §5). */
int scg_apply_update(scg_ctx *ctx, float *W, const float *G, const int32_t *n_k, void *stream);
/* The same pair as ONE all-reduce operand: G_packed holds n_vf*6480 floats of G followed by n_vf floats that
 * receive the update counts as floats (exact: they stay far below 2^24), so a sharded run with shared weights
 * sums (G, n_k) over the ranks with a single latency-bound collective per step. NULL restores the ctx-owned
 * buffers. scg_apply_update_packed reads the counts back from the tail. */
int scg_set_grad_buffer_packed(scg_ctx *ctx, float *G_packed);
int scg_apply_update_packed(scg_ctx *ctx, float *W, const float *G_packed, void *stream);
/* The ORDER-PINNED multi-rank sum (ABI 3): `slots` holds the packed operands of n_slots ranks, slot r at
 * slots + r * slot_stride floats (slot_stride >= n_vf*6480 + n_vf) — what an all-gather of every rank's G_packed leaves —
 * and the update is applied from their sum taken IN SLOT ORDER, ((G_0 + G_1) + G_2) + ..., element by element
 * (SPEC §5): the weights are then identical on every rank of a run and reproducible by the oracle for any number of ranks (the rank count, like the block size and the seed, is part of the run's identity).
 * (An all-reduce's order of additions is the library's business: exact for two ranks, to rounding beyond.) */
int scg_apply_update_slots(scg_ctx *ctx, float *W, const float *slots, int32_t n_slots, int64_t slot_stride, void *stream);

/* ---- un-fused entry points (same arithmetic; used by the API facade and the parity tests) ---- */

/* PinballDomain.step (SPEC §1.3) with caller-given actions; no reset. goal[n] u8. */
int scg_pinball_step(scg_ctx *ctx, int32_t n, float *x, float *y, float *vx, float *vy,
                     const uint8_t *action, float *reward, uint8_t *goal, void *stream);
/* FourierBasis.features (SPEC §3): phi[n][1296]. */
int scg_fourier_features(scg_ctx *ctx, int32_t n, const float *x, const float *y, const float *vx,
                         const float *vy, float *phi, void *stream);
/* Option.policy value part (SPEC §3.1): q[5][n] = Q_k(s, .) for one VF Wk[5][1296]. */
int scg_q_values(scg_ctx *ctx, int32_t n, const float *x, const float *y, const float *vx,
                 const float *vy, const float *Wk, float *q, void *stream);
/* SkillChainingAgent.q_update on explicit transitions for one VF (SPEC §5): accumulates G_k and n_k
 * into the ctx gradient buffers at VF index k (other VFs zeroed), optionally applies to Wk_all. */
int scg_q_update(scg_ctx *ctx, int32_t n, int32_t k, const float *x, const float *y, const float *vx,
                 const float *vy, const uint8_t *action, const float *r, const float *cont,
                 const float *xn, const float *yn, const float *vxn, const float *vyn, float *W,
                 uint32_t flags, void *stream);
/* Option.initiation_classifier.predict (SPEC §4.1): out[n] u8 = w.psi(x,y) > 0 for one row w8[8]. */
int scg_classifier_predict(scg_ctx *ctx, int32_t n, const float *x, const float *y, const float *w8,
                           uint8_t *out, void *stream);
/* Option.initiation_classifier.fit (SPEC §6), batched over n_fit options: xy[M_total][2],
 * label[M_total] u8, offsets[n_fit+1] int32 (device), w[n_fit][8] in/out. */
int scg_fit_initiation(scg_ctx *ctx, int32_t n_fit, const float *xy, const uint8_t *label,
                       const int32_t *offsets, float *w, int32_t iters, float lr, float l2, void *stream);

/* Option graph (SPEC §4.2): parents[k] for k = 1..n_options (HOST int32[n_options+1], entry 0
 * ignored) = the option whose initiation set option k targets, 0 = the task goal. Default: the chain
 * k -> k-1. Must be acyclic (every option reaches the goal). */
int scg_set_option_parents(scg_ctx *ctx, const int32_t *parents);

/* A learning scg_step also prepares the env order (SPEC §5) of the next step from the option ids it leaves.
 * Call this after writing `option_id` by any other means (a reset, a restored checkpoint): the next scg_step
 * then sorts afresh. Without it the step is still correct for the ids it finds, but groups and sums them in
 * the stale order (slower, and not the canonical rounding). A different `option_id` pointer is noticed
 * automatically. */
int scg_invalidate_order(scg_ctx *ctx);

/* ---- outer-loop support (SPEC §7): device-resident trajectory ring + per-step events,
 * so that the host skill-discovery loop never has to read env state every step.
 * scg_set_trace_buffers: caller-owned device buffers filled by every following scg_step (NULLs disable):
 *   ring_x, ring_y  f32[ring_len][N]  position of s_t at row (ep_steps at entry) & (ring_len-1); ring_len = 2^m
 *   events          u8[N]   bit 0 = reached the goal this step; bit k (1..5) = s' lies in initiation set k
 *   ev_len          i32[N]  states recorded so far in the env's episode (ep_steps at entry + 1)
 * scg_harvest: for each listed env (device int32 list, caller-sorted) emit L_pos + L_neg examples taken
 * backwards from the most recent recorded state: out_xy[n_sel][L][2], out_label[n_sel][L] u8 with
 * 1 = one of the last L_pos states, 0 = older, 255 = no such state (episode too short / ring overwritten). */
int scg_set_trace_buffers(scg_ctx *ctx, float *ring_x, float *ring_y, int32_t ring_len, uint8_t *events,
                          int32_t *ev_len);
int scg_harvest(scg_ctx *ctx, int32_t n_sel, const int32_t *sel_env, const float *ring_x, const float *ring_y,
                int32_t ring_len, const int32_t *ev_len, int32_t l_pos, int32_t l_neg, float *out_xy,
                uint8_t *out_label, void *stream);

/* scg_collect_examples: the device-side creation trigger (SPEC §7), one launch, nothing returns to the host: every env
 * whose `events` byte has one of `event_bits` set — with prev_in (u8[N], in/out) only on the step the bit goes up —
 * appends its min(L, ev_len, ring_len) most recent ring states behind the *count (device int32, in/out) examples
 * already in ex_xy[cap][2] / ex_label[cap] (1 = one of the last l_pos states, 0 = older), in env order; what does not
 * fit is dropped. The trace buffers of scg_set_trace_buffers are read. */
int scg_collect_examples(scg_ctx *ctx, uint32_t event_bits, uint8_t *prev_in, int32_t l_pos, int32_t l_neg,
                         float *ex_xy, uint8_t *ex_label, int32_t *count, int32_t cap, void *stream);
/* Optional: announce the trigger the NEXT scg_collect_examples will be called with (same event_bits, prev_in, l_pos + l_neg and
 * count). Every following scg_step then leaves the per-row example totals behind while it commits its results (the rows are
 * the same), and a matching scg_collect_examples right after it needs one launch instead of two. Results are identical either
 * way; a call that does not match the announcement, or comes without a step in between, takes the two-launch path.
 * event_bits = 0 withdraws the announcement. prev_in and count are READ by every following scg_step until then (device
 * pointers kept in the ctx): keep them alive, or withdraw the announcement before freeing them. scg_step checks them best-effort
 * (hipPointerGetAttributes: memory handed back to the DRIVER is noticed and refused with SCG_ERR_STATE; memory a caching allocator
 * such as torch's has merely recycled still reads as a device allocation and is NOT noticed) — the rule above is the contract. */
int scg_arm_collect(scg_ctx *ctx, uint32_t event_bits, const uint8_t *prev_in, int32_t l_pos, int32_t l_neg, const int32_t *count);

/* Gestation (SPEC §4.4; Konidaris & Barto 2009: a new option learns off-policy before it may run). Bit k of gest_mask:
 * option k's classifier is in use (initiation / target tests, event bits) but the option is never selected; every env
 * whose state lies in its initiation set contributes an off-policy TD item to VF k; succ_counts[k] (device int32[n_vf],
 * caller-owned, may be NULL) counts the transitions that reached option k's target from inside its initiation set.
 * The caller moves the bit from gest_mask to scg_step's enabled_mask when the count is high enough. */
int scg_set_gestation(scg_ctx *ctx, uint32_t gest_mask, int32_t *succ_counts);

/* ---- asynchronous failures ----
 * Launches are asynchronous, so a kernel that has to give up cannot return a status. It leaves its outputs untouched
 * and ORs a reason into a host-visible status word owned by the ctx; from then on EVERY entry point that would launch
 * work returns SCG_ERR_ASYNC (scg_last_error names the reason) until scg_clear_async_error is called. Today one
 * kernel can do this: scg_fit_initiation's — the eight workgroups of a fit problem exchange partial sums through
 * memory every iteration and need to be running at the same time; if the card is shared (another stream, another
 * process) a workgroup may be kept off it. A missing partner is waited for scg_set_fit_timeout seconds of wall clock
 * (default 2 s), then the fit of that problem is abandoned: its row of `w` keeps the values it had, never a partial
 * result or a NaN. Word layout: SCG_ASYNC_FIT_TIMEOUT | 0x100 << (problem index & 15).
 * The step kernel is the second: the wavefront subsets of a workgroup hand work to each other through counters in LDS and
 * poll them with a bound (2^20 rounds; every awaited count is produced by wavefronts that never wait on the waiter, so
 * the bound is only reached through a logic error or a hung wavefront). A poll that runs out raises
 * SCG_ASYNC_STEP_HANDOFF: that block's partial gradients are dropped (its slab counts read 0) and the step's other
 * outputs for the block's envs are unspecified — the state must be restored from a checkpoint.
 *   scg_async_status      the word (optional out) and its status; `synchronize` != 0 waits for `stream` first, which
 *                         makes the answer final for everything launched on it so far
 *   scg_decode_async_word the same mapping word -> status + text without a ctx (pure host code) */
#define SCG_ASYNC_FIT_TIMEOUT 0x1u
#define SCG_ASYNC_STEP_HANDOFF 0x2u
int scg_async_status(scg_ctx *ctx, void *stream, int32_t synchronize, uint32_t *word_out);
int scg_clear_async_error(scg_ctx *ctx);
int scg_set_fit_timeout(scg_ctx *ctx, double seconds);
int scg_decode_async_word(uint32_t word, char *buf, int32_t buf_len);
/* test hook: raise bits of the status word from the host, as a kernel would */
int scg_debug_raise_async(scg_ctx *ctx, uint32_t word);

/* ---- measurement hooks (bench.py's roofline leg) ----
 * scg_profile_reset(ctx, p) with p >= 1 makes every p-th following scg_step record a HIP event pair round
 * its fused kernel on the launch stream (each pair costs a few microseconds of queue bubble, so sampling
 * perturbs the timed region less); scg_profile_read synchronises those events and returns the summed
 * kernel time (ms) and the number of launches measured; scg_profile_reset(ctx, 0) stops recording. */
int scg_profile_reset(scg_ctx *ctx, int32_t enable);
int scg_profile_read(scg_ctx *ctx, double *kernel_ms_sum, int64_t *launches);

#ifdef __cplusplus
}
#endif
#endif
