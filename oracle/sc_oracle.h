/*
 * sc_oracle.h — scalar CPU restatement of SPEC.md (TEST INFRASTRUCTURE, NOT PRODUCT).
 *
 * PARITY UNPINNED: the upstream reference (/root/reference) holds only README.md:1-2 (a title and one
 * sentence naming Konidaris & Barto 2009); it has no code, tests, fixtures or golden vectors, so
 * nothing here can be checked against it. This oracle follows this repo's own
 * SPEC.md and is pinned only by the hand-computed known-answer tests in tests/ and the published
 * Philox4x32-10 vectors. It may be imported/linked only by tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg. The product (skill-chaining-with-graphs_amd/) never touches it.
 */
#ifndef SC_ORACLE_H
#define SC_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SCO_NACT 5
#define SCO_NF 1296
#ifndef SCO_BLOCK_ENVS
#define SCO_BLOCK_ENVS 256          /* SPEC §5: envs per block (64 / 128: the small-batch experiment of DESIGN §10) */
#endif
#define SCO_CLF_STRIDE 8

typedef struct {
    int32_t n_envs;
    int32_t n_options;          /* chained options 1..n_options; VF 0 = root */
    int64_t env_id_base;
    uint64_t seed;
    float gamma, alpha, epsilon, r_option_success;
    int32_t max_episode_steps, max_option_steps;
    uint32_t enabled_mask;      /* bit k = option k usable */
    int32_t n_threads;          /* OpenMP threads for the block loop (1 = scalar) */
    /* map (SPEC §1.1) */
    int32_t n_edges;
    int32_t n_starts;
    const float *edges;         /* [n_edges][8] x0,y0,ex,ey,inv_len2,ux,uy,pad */
    const float *starts;        /* [n_starts][2] */
    float radius, hstep, r2, tx, ty, tr2;
    const float *scale;         /* [1296] 1/||c|| */
    /* SPEC §7 trace buffers (NULL = off) */
    float *ring_x, *ring_y;     /* [ring_len][n_envs] */
    uint8_t *events;            /* [n_envs] */
    int32_t *ev_len;            /* [n_envs] */
    int32_t ring_len;           /* power of two */
    int32_t parents[8];         /* SPEC §4.2 option graph: target option of k (0 = goal); [0] unused */
    uint32_t gest_mask;         /* SPEC §4.4: options in gestation */
    int32_t *gest_succ;         /* [n_options + 1] success counters of gestating options (NULL = off) */
    /* SPEC §4.2 exit rule of an option's own (and gestating, §4.4) items: what an option that ENDS without the episode ending
     * bootstraps from. 0: nothing (continuation 0 — rounds 1-4); 1: fail / time-out bootstrap from the root's value of where
     * the env goes next, gamma * max_a Q_0(s_next, a), success ends with r + r_option_success; 2: success bootstraps too;
     * 3 (experiment): as 0 with a pure subgoal reward — the option's reward is r_option_success on success and 0 otherwise (no step costs) */
    int32_t exit_rule;
    /* SPEC §4.2 option selection: 0 = forced (an env inside an initiation set must run the option); 1 = value-gated entry */
    int32_t select_rule;
    /* SPEC §5 apply: the divisor of a value function's summed update is max(n_k, nk_floor) (0 = plain n_k) */
    int32_t nk_floor;
    /* SPEC §4.2: an env staying out of option k (option_id = -k) is offered k again when (t + global env id) % reoffer_period == 0
     * (<= 1: every step) */
    int32_t reoffer_period;
} sco_params;

void sco_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
void sco_sincospi(float t, float *c, float *s);
float sco_sigmoid(float z);
int sco_block_envs(void);            /* the SPEC §5 block size this build of the checker was compiled for */

/* SPEC §1.3 for n independent envs with given actions; no reset, no bookkeeping. */
void sco_pinball_step(const sco_params *p, int n, float *x, float *y, float *vx, float *vy,
                      const uint8_t *action, float *reward, uint8_t *goal);
/* SPEC §3: phi[n][1296] */
void sco_features(int n, const float *x, const float *y, const float *vx, const float *vy, float *phi);
/* SPEC §3.1: q[5][n] for one VF (W_k = [5][1296]) */
void sco_q_values(int n, const float *x, const float *y, const float *vx, const float *vy,
                  const float *Wk, float *q);
/* SPEC §4.1 */
void sco_classifier_predict(int n, const float *x, const float *y, const float *w8, uint8_t *out);
/* SPEC §5 for one VF on explicit transitions (all items upd, none cache); G[5][1296], returns n_k. */
int sco_q_update_grad(const sco_params *p, int n, const float *s4[4], const uint8_t *action,
                      const float *r, const float *cont, const float *sn4[4], const float *Wk, float *G);
/* SPEC §5 apply for n_vf value functions. */
void sco_apply(const sco_params *p, int n_vf, float *W, const float *G, const int32_t *n_k);
/* SPEC §1.4, §2, §4, §5: one step-batch. G[n_vf][5][1296] and n_k[n_vf] are outputs; W is not modified. */
void sco_step(const sco_params *p, float *x, float *y, float *vx, float *vy, int32_t *option_id,
              int32_t *opt_steps, int32_t *ep_steps, float *qcache, uint8_t *action, float *reward,
              uint8_t *done, const float *W, const float *clf, uint64_t t, float *G, int32_t *n_k);
/* SPEC §7: examples from the trajectory ring for the listed envs: out_xy[n_sel][L][2], out_label[n_sel][L] */
void sco_harvest(int n_sel, const int32_t *sel_env, const float *ring_x, const float *ring_y, int ring_len,
                 int n_envs, const int32_t *ev_len, int l_pos, int l_neg, float *out_xy, uint8_t *out_label);
/* SPEC §7 device-side trigger: envs (in env order) whose events byte has one of `bits` set — with prev_in only on the step
 * the bit goes up (prev_in is updated) — append their min(L, ev_len, ring_len) most recent ring states behind the *count
 * examples already held; what does not fit into cap is dropped. */
void sco_collect_examples(int n_envs, const uint8_t *events, uint8_t *prev_in, uint32_t bits, const float *ring_x,
                          const float *ring_y, int ring_len, const int32_t *ev_len, int l_pos, int l_neg,
                          float *ex_xy, uint8_t *ex_label, int32_t *count, int cap);
/* SPEC §6: n_fit problems; offsets[n_fit+1] index xy/label; w[n_fit][8] in/out. */
void sco_fit_initiation(int n_fit, const float *xy, const uint8_t *label, const int32_t *offsets,
                        float *w, int iters, float lr, float l2);

#ifdef __cplusplus
}
#endif
#endif
