/*
 * sc_oracle.c — scalar CPU restatement of SPEC.md. TEST INFRASTRUCTURE, NOT PRODUCT; PARITY UNPINNED
 * (see sc_oracle.h: the upstream reference is README.md:1-2 only, there is nothing to follow or cite
 * beyond the paper title; every function below cites the SPEC.md section it implements).
 *
 * Build: gcc -O2 -std=c11 -ffp-contract=off -mfma -fopenmp -fPIC -shared (oracle/Makefile).
 * -ffp-contract=off is REQUIRED: fusion happens only where fmaf() is written.
 */
#include "sc_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define NACT SCO_NACT
/* SPEC §5 geometry: envs per block (a block's update items share one accumulation chain per weight) */
static const int g_block_envs = SCO_BLOCK_ENVS;
#define NF SCO_NF

/* ------------------------------------------------------------------ SPEC §2: Philox4x32-10 */
void sco_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int round = 0; round < 10; ++round) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

static inline uint32_t mulhi32(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }

/* ------------------------------------------------------------------ SPEC §3: sincospi, cmul */
void sco_sincospi(float t, float *c, float *s) {
    const float S0 = 0x1.921fb6p+1f, S1 = -0x1.4abbcep+2f, S2 = 0x1.466bc6p+1f, S3 = -0x1.32d2ccp-1f,
                S4 = 0x1.507834p-4f;
    const float C0 = -0x1.3bd3ccp+2f, C1 = 0x1.03c1f0p+2f, C2 = -0x1.55d3c8p+0f, C3 = 0x1.e1f506p-3f,
                C4 = -0x1.a6d1f2p-6f;
    float n = rintf(t + t);
    float r = fmaf(n, -0.5f, t);
    float z = r * r;
    float sp = fmaf(z, S4, S3); sp = fmaf(z, sp, S2); sp = fmaf(z, sp, S1); sp = fmaf(z, sp, S0); sp = sp * r;
    float cp = fmaf(z, C4, C3); cp = fmaf(z, cp, C2); cp = fmaf(z, cp, C1); cp = fmaf(z, cp, C0);
    cp = fmaf(z, cp, 1.0f);
    int q = (int)n & 3;
    switch (q) {
        case 0: *c = cp; *s = sp; break;
        case 1: *c = -sp; *s = cp; break;
        case 2: *c = -cp; *s = -sp; break;
        default: *c = sp; *s = -cp; break;
    }
}

typedef struct { float re, im; } cplx;
static inline cplx cmul(cplx a, cplx b) {
    cplx o;
    o.re = fmaf(-a.im, b.im, a.re * b.re);
    o.im = fmaf(a.re, b.im, a.im * b.re);
    return o;
}

/* AB[36], CD[36] of one state (SPEC §3) */
static void state_tables(float x, float y, float vx, float vy, cplx AB[36], cplx CD[36]) {
    float sh[4] = {x, y, fmaf(vx, 0.25f, 0.5f), fmaf(vy, 0.25f, 0.5f)};
    cplx Z[4][6];
    for (int d = 0; d < 4; ++d) {
        Z[d][0].re = 1.0f; Z[d][0].im = 0.0f;
        sco_sincospi(sh[d], &Z[d][1].re, &Z[d][1].im);
        for (int k = 2; k < 6; ++k) Z[d][k] = cmul(Z[d][k - 1], Z[d][1]);
    }
    for (int b = 0; b < 6; ++b) {                 /* row 0 is the power itself, row c the row above times Z^1 */
        AB[b] = Z[1][b];
        CD[b] = Z[3][b];
        for (int a = 1; a < 6; ++a) {
            AB[a * 6 + b] = cmul(AB[(a - 1) * 6 + b], Z[0][1]);
            CD[a * 6 + b] = cmul(CD[(a - 1) * 6 + b], Z[2][1]);
        }
    }
}

static void state_features(float x, float y, float vx, float vy, float *phi) {
    cplx AB[36], CD[36];
    state_tables(x, y, vx, vy, AB, CD);
    for (int c12 = 0; c12 < 36; ++c12)
        for (int c34 = 0; c34 < 36; ++c34)
            phi[c12 * 36 + c34] = fmaf(-AB[c12].im, CD[c34].im, AB[c12].re * CD[c34].re);
}

/* ------------------------------------------------------------------ SPEC §3.1: Q_k(sigma, a) */
typedef struct { cplx AB[36], CD[36]; } st_tab;

/* Wa = W_k[a] as [36][36] (c12 major). Two contractions: T[c12][part] over c34 in the order c34 = 9 g + kb
 * (kb = 0..8 outer, g = 0..3 inner), then four partial chains over c12 (one per row group) and a fixed tree. */
static float q_value(const float *Wa, int a, const st_tab *tb) {
    float q[4][2] = {{0.0f, 0.0f}, {0.0f, 0.0f}, {0.0f, 0.0f}, {0.0f, 0.0f}};
    for (int c12 = 0; c12 < 36; ++c12) {
        float tre = 0.0f, tim = 0.0f;
        for (int kb = 0; kb < 9; ++kb)
            for (int g = 0; g < 4; ++g) {
                const int c34 = 9 * g + kb;
                const float w = Wa[c12 * 36 + c34];
                tre = fmaf(w, tb->CD[c34].re, tre);
                tim = fmaf(w, tb->CD[c34].im, tim);
            }
        const int grp = ((36 * a + c12) % 16) / 4;
        q[grp][0] = fmaf(tre, tb->AB[c12].re, q[grp][0]);
        q[grp][1] = fmaf(tim, -tb->AB[c12].im, q[grp][1]);
    }
    float u[4];
    for (int g = 0; g < 4; ++g) u[g] = q[g][0] + q[g][1];
    return (u[0] + u[1]) + (u[2] + u[3]);
}

/* ------------------------------------------------------------------ SPEC §1.3: physics */
static int intercept(const float *E, float R2, float x, float y, float vx, float vy) {
    const float KAPPA2 = 0x1.0553bep-14f;
    float x0 = E[0], y0 = E[1], ex = E[2], ey = E[3], inv = E[4];
    float dx = x - x0, dy = y - y0;
    float t = fmaf(dy, ey, dx * ex) * inv;
    t = fminf(fmaxf(t, 0.0f), 1.0f);
    float cx = fmaf(ex, t, x0), cy = fmaf(ey, t, y0);
    float bx = cx - x, by = cy - y;
    float d2 = fmaf(by, by, bx * bx);
    if (d2 > R2) return 0;
    float dot = fmaf(by, vy, bx * vx);
    if (dot >= 0.0f) return 1;
    float vv = fmaf(vy, vy, vx * vx);
    return dot * dot <= (KAPPA2 * d2) * vv;
}

static void pinball_step1(const sco_params *p, float *px, float *py, float *pvx, float *pvy, int a,
                          float *reward, int *goal_out) {
    const float DV = 0x1.99999ap-3f, VMAX = 2.0f, DRAG = 0x1.fd70a4p-1f;
    float x = *px, y = *py, vx = *pvx, vy = *pvy;
    const float h = p->hstep;
    if (a == 0) vx = vx + DV;
    else if (a == 2) vx = vx - DV;
    else if (a == 1) vy = vy + DV;
    else if (a == 3) vy = vy - DV;
    vx = fminf(fmaxf(vx, -VMAX), VMAX);
    vy = fminf(fmaxf(vy, -VMAX), VMAX);
    int goal = 0;
    for (int i = 0; i < 20; ++i) {
        x = fmaf(vx, h, x); y = fmaf(vy, h, y);
        int nhit = 0, first = -1;
        for (int j = 0; j < p->n_edges; ++j)
            if (intercept(p->edges + 8 * j, p->r2, x, y, vx, vy)) {
                if (nhit == 0) first = j;
                ++nhit;
            }
        if (nhit == 1) {
            const float *E = p->edges + 8 * first;
            float ux = E[5], uy = E[6];
            float pr = fmaf(vy, uy, vx * ux);
            float tp = pr + pr;
            float nvx = fmaf(tp, ux, -vx), nvy = fmaf(tp, uy, -vy);
            vx = nvx; vy = nvy;
            if (i == 19) { x = fmaf(vx, h, x); y = fmaf(vy, h, y); }
        } else if (nhit > 1) {
            vx = -vx; vy = -vy;
        }
        float gx = x - p->tx, gy = y - p->ty;
        if (fmaf(gy, gy, gx * gx) < p->tr2) { goal = 1; break; }
    }
    if (goal) {
        *reward = 10000.0f;
    } else {
        vx = vx * DRAG; vy = vy * DRAG;
        x = fminf(fmaxf(x, 0.0f), 1.0f); y = fminf(fmaxf(y, 0.0f), 1.0f);
        *reward = (a == 4) ? -1.0f : -5.0f;
    }
    *px = x; *py = y; *pvx = vx; *pvy = vy; *goal_out = goal;
}

void sco_pinball_step(const sco_params *p, int n, float *x, float *y, float *vx, float *vy,
                      const uint8_t *action, float *reward, uint8_t *goal) {
    for (int e = 0; e < n; ++e) {
        int g;
        pinball_step1(p, &x[e], &y[e], &vx[e], &vy[e], action[e], &reward[e], &g);
        goal[e] = (uint8_t)g;
    }
}

void sco_features(int n, const float *x, const float *y, const float *vx, const float *vy, float *phi) {
    for (int e = 0; e < n; ++e) state_features(x[e], y[e], vx[e], vy[e], phi + (size_t)e * NF);
}

void sco_q_values(int n, const float *x, const float *y, const float *vx, const float *vy,
                  const float *Wk, float *q) {
    for (int e = 0; e < n; ++e) {
        st_tab tb;
        state_tables(x[e], y[e], vx[e], vy[e], tb.AB, tb.CD);
        for (int a = 0; a < NACT; ++a) q[(size_t)a * n + e] = q_value(Wk + a * NF, a, &tb);
    }
}

/* ------------------------------------------------------------------ SPEC §4.1: classifier */
static float clf_z(const float *w, float x, float y) {
    float u = fmaf(x, 2.0f, -1.0f), v = fmaf(y, 2.0f, -1.0f);
    float z = w[0];
    z = fmaf(w[1], u, z); z = fmaf(w[2], v, z);
    z = fmaf(w[3], u * u, z); z = fmaf(w[4], u * v, z); z = fmaf(w[5], v * v, z);
    return z;
}

void sco_classifier_predict(int n, const float *x, const float *y, const float *w8, uint8_t *out) {
    for (int e = 0; e < n; ++e) out[e] = clf_z(w8, x[e], y[e]) > 0.0f;
}

/* membership test of a KNOWN option (enabled or gestating, SPEC §4.4) */
static int in_set(const sco_params *p, const float *clf, int k, float x, float y) {
    if (k < 1 || k > p->n_options) return 0;
    if (!(((p->enabled_mask | p->gest_mask) >> k) & 1u)) return 0;
    return clf_z(clf + SCO_CLF_STRIDE * k, x, y) > 0.0f;
}

/* ------------------------------------------------------------------ SPEC §5: block TD machinery */
typedef struct {
    float s[4], sn[4];
    int a;
    /* per VF flags for this env, filled by the caller */
} env_rec;

/* One (block, VF) pass. items: block-local env indices in block order with flags; tab_s / tab_n: the AB / CD
 * tables of s and s_next of every env of the block. Writes the block partial P_b,k into Pout[5][1296]. */
typedef struct {
    int env;      /* block-local env index */
    int upd, tgt, cache;
    float r, cont;
    int want;     /* SPEC §4.2 value-gated entry: evaluate Q(s_next, .) although the item neither bootstraps nor caches */
    int boot;     /* SPEC §4.2 exit rule: the target is r + gamma * boot_v (the root's max_a Q_0(s_next, a)) */
    float boot_v, boot_g;
} td_item;

static void block_vf_pass(const float *Wk, int n_items, const td_item *items, const env_rec *rec,
                          const st_tab *tab_s, const st_tab *tab_n, float *qcache, int qstride,
                          const int *env_of, float *Pout, int *n_upd, float *maxq_env, int maxq_cache_only, float (*qn_env)[NACT]) {
    /* evaluations (order-free: each item's Q values depend on nothing else) */
    float *maxq = (float *)calloc((size_t)(n_items > 0 ? n_items : 1), sizeof(float));
    for (int i = 0; i < n_items; ++i) {
        const td_item *it = &items[i];
        if (it->tgt || it->cache || it->want) {
            float qn[NACT];
            for (int a = 0; a < NACT; ++a) qn[a] = q_value(Wk + a * NF, a, &tab_n[it->env]);
            if (it->cache)
                for (int a = 0; a < NACT; ++a) qcache[(size_t)a * qstride + env_of[it->env]] = qn[a];
            float m = qn[0];
            for (int a = 1; a < NACT; ++a) m = fmaxf(m, qn[a]);
            maxq[i] = m;
            if (maxq_env && (!maxq_cache_only || it->cache)) maxq_env[it->env] = m;
            if (qn_env) for (int a = 0; a < NACT; ++a) qn_env[it->env][a] = qn[a];
        }
    }
    /* SPEC §5 accumulation: per action, the run of update items in block order, in groups of four (the last
     * group padded with null items whose operands are +0); inside a group first the real parts of the four
     * items, then the imaginary parts: G = fma(P, C, G) with P = delta * ABsel, C = CD. */
    memset(Pout, 0, sizeof(float) * NACT * NF);
    int cnt = 0;
    int *run = (int *)malloc(sizeof(int) * (size_t)(n_items > 0 ? n_items : 1));
    float *dl = (float *)malloc(sizeof(float) * (size_t)(n_items > 0 ? n_items : 1));
    for (int act = 0; act < NACT; ++act) {
        int m = 0;
        for (int i = 0; i < n_items; ++i) {
            const td_item *it = &items[i];
            if (!it->upd || rec[it->env].a != act) continue;
            float qsa = q_value(Wk + act * NF, act, &tab_s[it->env]);
            float target = it->boot ? fmaf(it->boot_g, it->boot_v, it->r) : (it->tgt ? fmaf(it->cont, maxq[i], it->r) : it->r);
            dl[m] = target - qsa;
            run[m++] = it->env;
        }
        cnt += m;
        float *Ga = Pout + (size_t)act * NF;
        for (int g0 = 0; g0 < m; g0 += 4)
            for (int part = 0; part < 2; ++part)
                for (int j = 0; j < 4; ++j) {
                    const int null = g0 + j >= m;
                    const st_tab *tb = null ? NULL : &tab_s[run[g0 + j]];
                    const float d = null ? 0.0f : dl[g0 + j];
                    float cc[36];
                    for (int c34 = 0; c34 < 36; ++c34) cc[c34] = null ? 0.0f : (part ? tb->CD[c34].im : tb->CD[c34].re);
                    for (int c12 = 0; c12 < 36; ++c12) {
                        const float pp = null ? 0.0f : (part ? d * (-tb->AB[c12].im) : d * tb->AB[c12].re);
                        float *row = Ga + c12 * 36;
                        for (int c34 = 0; c34 < 36; ++c34) row[c34] = fmaf(pp, cc[c34], row[c34]);
                    }
                }
    }
    free(run); free(dl); free(maxq);
    *n_upd = cnt;
}


/* SPEC §5: G = ((T_0 + T_1) + ...), T_s = ((P_16s + P_16s+1) + ...) over the blocks of segment s */
#define SCO_SEG 16
static float seg_sum(const float *P, size_t stride, size_t idx, int nblk) {
    float g = 0.0f;
    int first = 1;
    for (int s0 = 0; s0 < nblk; s0 += SCO_SEG) {
        float t = P[(size_t)s0 * stride + idx];
        int s1 = s0 + SCO_SEG < nblk ? s0 + SCO_SEG : nblk;
        for (int b = s0 + 1; b < s1; ++b) t = t + P[(size_t)b * stride + idx];
        g = first ? t : g + t;
        first = 0;
    }
    return g;
}

int sco_q_update_grad(const sco_params *p, int n, const float *s4[4], const uint8_t *action,
                      const float *r, const float *cont, const float *sn4[4], const float *Wk, float *G) {
    int nblk = (n + g_block_envs - 1) / g_block_envs;
    float *P = (float *)malloc(sizeof(float) * (size_t)(nblk > 0 ? nblk : 1) * NACT * NF);
    int *cnts = (int *)calloc((size_t)(nblk > 0 ? nblk : 1), sizeof(int));
    int nthreads = p && p->n_threads > 0 ? p->n_threads : 1;
    (void)nthreads;
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 1)
    for (int b = 0; b < nblk; ++b) {
        int e0 = b * g_block_envs;
        int nb = n - e0 < g_block_envs ? n - e0 : g_block_envs;
        env_rec *rec = (env_rec *)malloc(sizeof(env_rec) * nb);
        td_item *items = (td_item *)malloc(sizeof(td_item) * nb);
        st_tab *tab_s = (st_tab *)malloc(sizeof(st_tab) * nb);
        st_tab *tab_n = (st_tab *)malloc(sizeof(st_tab) * nb);
        float dummy_q[NACT];
        for (int i = 0; i < nb; ++i) {
            int e = e0 + i;
            for (int d = 0; d < 4; ++d) { rec[i].s[d] = s4[d][e]; rec[i].sn[d] = sn4[d][e]; }
            rec[i].a = action[e];
            items[i].env = i; items[i].upd = 1; items[i].tgt = cont[e] > 0.0f; items[i].cache = 0;
            items[i].want = 0; items[i].boot = 0; items[i].boot_v = 0.0f; items[i].boot_g = 0.0f;
            items[i].r = r[e]; items[i].cont = cont[e];
            state_tables(rec[i].s[0], rec[i].s[1], rec[i].s[2], rec[i].s[3], tab_s[i].AB, tab_s[i].CD);
            if (items[i].tgt)
                state_tables(rec[i].sn[0], rec[i].sn[1], rec[i].sn[2], rec[i].sn[3], tab_n[i].AB, tab_n[i].CD);
        }
        block_vf_pass(Wk, nb, items, rec, tab_s, tab_n, dummy_q, 0, NULL, P + (size_t)b * NACT * NF, &cnts[b], NULL, 0, NULL);
        free(rec); free(items); free(tab_s); free(tab_n);
    }
    int total = 0;
    for (size_t i = 0; i < (size_t)NACT * NF; ++i) G[i] = seg_sum(P, (size_t)NACT * NF, i, nblk);
    for (int b = 0; b < nblk; ++b) total += cnts[b];
    free(P); free(cnts);
    return total;
}

void sco_apply(const sco_params *p, int n_vf, float *W, const float *G, const int32_t *n_k) {
    for (int k = 0; k < n_vf; ++k) {
        if (n_k[k] <= 0) continue;
        float step = p->alpha / (float)(n_k[k] > p->nk_floor ? n_k[k] : p->nk_floor);
        for (int a = 0; a < NACT; ++a)
            for (int f = 0; f < NF; ++f) {
                size_t i = ((size_t)k * NACT + a) * NF + f;
                W[i] = fmaf(step * p->scale[f], G[i], W[i]);
            }
    }
}

/* ------------------------------------------------------------------ SPEC §1.4, §2, §4, §5: step-batch */
void sco_step(const sco_params *p, float *x, float *y, float *vx, float *vy, int32_t *option_id,
              int32_t *opt_steps, int32_t *ep_steps, float *qcache, uint8_t *action, float *reward,
              uint8_t *done, const float *W, const float *clf, uint64_t t, float *G, int32_t *n_k) {
    const int N = p->n_envs;
    const int n_vf = p->n_options + 1;
    const int nblk = (N + g_block_envs - 1) / g_block_envs;
    /* SPEC §5: envs are taken in the order of (sort key at entry, env index) — a stable counting sort. The key comes from the
     * signed option id: k in [1, n_vf) -> k (running option k); 0 and -k (no option in sight / inside option k's initiation set
     * but staying out of it, §4.2: either way the env runs the root) -> 0; anything else -> the last key n_vf. */
    int *perm = (int *)malloc(sizeof(int) * (size_t)(N > 0 ? N : 1));
    {
        enum { NK = 7 };
        int tot[NK];
        for (int k = 0; k < NK; ++k) tot[k] = 0;
#define SCO_KEY(o) ((o) <= 0 ? ((o) > -n_vf ? 0 : n_vf) : ((o) < n_vf ? (o) : n_vf))
        for (int e = 0; e < N; ++e) tot[SCO_KEY(option_id[e])]++;
        /* lists of envs per key, env order */
        int *lst[NK], fillpos = 0;
        for (int k = 0; k < NK; ++k) lst[k] = (int *)malloc(sizeof(int) * (size_t)(tot[k] > 0 ? tot[k] : 1));
        {
            int cntk[NK];
            for (int k = 0; k < NK; ++k) cntk[k] = 0;
            for (int e = 0; e < N; ++e) { int k = SCO_KEY(option_id[e]); lst[k][cntk[k]++] = e; }
        }
        int S = 0, Rn = 0;
        for (int k = 1; k < NK; ++k) { S += tot[k]; Rn += tot[k] > 0; }
        const int B = g_block_envs, Bf = N / B;
        int c = B;
        if (Bf > Rn && S > 0) { c = (S + (Bf - Rn) - 1) / (Bf - Rn); if (c > B) c = B; }
        int U = 0;
        for (int k = 1; k < NK; ++k) U += (tot[k] + c - 1) / c;
        int pos = 0;
        if ((long)U * B <= N) {
            /* chunked layout: every block takes at most c envs of one run, key-0 envs fill it up */
            for (int k = 1; k < NK; ++k)
                for (int r = 0; r < tot[k]; r += c) {
                    const int m = tot[k] - r < c ? tot[k] - r : c;
                    for (int i = 0; i < m; ++i) perm[pos++] = lst[k][r + i];
                    for (int i = m; i < B; ++i) perm[pos++] = lst[0][fillpos++];
                }
        } else {
            /* padded layout: runs back to back, each padded to the next block boundary while key-0 envs are left */
            for (int k = 1; k < NK; ++k) {
                for (int r = 0; r < tot[k]; ++r) perm[pos++] = lst[k][r];
                if (tot[k] > 0)
                    while (pos % B != 0 && fillpos < tot[0]) perm[pos++] = lst[0][fillpos++];
            }
        }
        while (fillpos < tot[0]) perm[pos++] = lst[0][fillpos++];
        for (int k = 0; k < NK; ++k) free(lst[k]);
#undef SCO_KEY
    }
    float *P = (float *)malloc(sizeof(float) * (size_t)(nblk > 0 ? nblk : 1) * n_vf * NACT * NF);
    int *cnts = (int *)calloc((size_t)(nblk > 0 ? nblk : 1) * n_vf, sizeof(int));
    int nthreads = p->n_threads > 0 ? p->n_threads : 1;
    (void)nthreads;
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 1)
    for (int b = 0; b < nblk; ++b) {
        int e0 = b * g_block_envs;
        int nb = N - e0 < g_block_envs ? N - e0 : g_block_envs;
        env_rec rec[SCO_BLOCK_ENVS];
        int o_t[SCO_BLOCK_ENVS], o_n[SCO_BLOCK_ENVS];
        float r0[SCO_BLOCK_ENVS], c0[SCO_BLOCK_ENVS], ro[SCO_BLOCK_ENVS], co[SCO_BLOCK_ENVS];
        unsigned gs[SCO_BLOCK_ENVS];
        float rg[SCO_BLOCK_ENVS][8], cg[SCO_BLOCK_ENVS][8];
        float rootmax[SCO_BLOCK_ENVS];           /* max_a Q_0(s_next, a) of every env whose episode goes on (root pass) */
        float candmax[SCO_BLOCK_ENVS];           /* SPEC §4.2 value-gated entry: max_a Q_cand(s_next, a) of an env about to ENTER option cand */
        float rootq[SCO_BLOCK_ENVS][NACT];
        int entering[SCO_BLOCK_ENVS];
        unsigned xo[SCO_BLOCK_ENVS], xg[SCO_BLOCK_ENVS];   /* exit-rule flags: bit 0 own option ended (episode goes on), bit 1 by success; xg: per gestating k, 2 bits each */
        st_tab *tab_s = (st_tab *)malloc(sizeof(st_tab) * nb);
        st_tab *tab_n = (st_tab *)malloc(sizeof(st_tab) * nb);
        td_item *items = (td_item *)malloc(sizeof(td_item) * nb);
        /* ---- phase P: act, physics, bookkeeping, option logic (per env) */
        int env_of[SCO_BLOCK_ENVS];
        for (int i = 0; i < nb; ++i) {
            int e = perm[e0 + i];
            env_of[i] = e;
            uint64_t g = (uint64_t)(p->env_id_base + e);
            uint32_t ctr[4] = {(uint32_t)g, (uint32_t)(t & 0xffffffffu), (uint32_t)(t >> 32), 0u};
            uint32_t key[2] = {(uint32_t)(p->seed & 0xffffffffu), (uint32_t)(p->seed >> 32)};
            uint32_t u[4];
            sco_philox4x32_10(ctr, key, u);
            int explore = (float)(u[0] >> 8) * 0x1p-24f < p->epsilon;
            int a_rand = (int)mulhi32(u[1], 5u);
            int a_greedy = 0;
            float best = qcache[e];
            for (int a = 1; a < NACT; ++a) {
                float q = qcache[(size_t)a * N + e];
                if (q > best) { best = q; a_greedy = a; }
            }
            int a = explore ? a_rand : a_greedy;
            float sx = x[e], sy = y[e], svx = vx[e], svy = vy[e];
            rec[i].s[0] = sx; rec[i].s[1] = sy; rec[i].s[2] = svx; rec[i].s[3] = svy;
            rec[i].a = a;
            float rew; int goal;
            pinball_step1(p, &sx, &sy, &svx, &svy, a, &rew, &goal);
            int eps1 = ep_steps[e] + 1;
            int timeout = !goal && eps1 >= p->max_episode_steps;
            int dn = goal ? 1 : (timeout ? 2 : 0);
            float nx = sx, ny = sy, nvx = svx, nvy = svy;
            if (dn) {
                uint32_t si = mulhi32(u[2], (uint32_t)p->n_starts);
                nx = p->starts[2 * si]; ny = p->starts[2 * si + 1]; nvx = 0.0f; nvy = 0.0f;
            }
            rec[i].sn[0] = nx; rec[i].sn[1] = ny; rec[i].sn[2] = nvx; rec[i].sn[3] = nvy;
            int o = option_id[e] > 0 ? option_id[e] : 0;        /* (-k: inside option k's initiation set, staying with the root) */
            int keep = 0;
            ro[i] = 0.0f; co[i] = 0.0f; xo[i] = 0; xg[i] = 0; rootmax[i] = 0.0f;
            if (o >= 1) {
                int par = p->parents[o & 7];
                int succ = (par == 0) ? goal : in_set(p, clf, par, sx, sy);
                int fail = !succ && !in_set(p, clf, o, sx, sy);
                int otime = opt_steps[e] + 1 >= p->max_option_steps;
                int term = (dn != 0) || succ || fail || otime;
                ro[i] = (p->exit_rule == 3 ? 0.0f : rew) + (succ ? p->r_option_success : 0.0f);
                co[i] = term ? 0.0f : p->gamma;
                keep = !term;
                if (term && dn == 0) xo[i] = 1u | (succ ? 2u : 0u);
            }
            int on = 0;
            if (keep) on = o;
            else
                for (int k = 1; k <= p->n_options; ++k) {
                    if (!((p->enabled_mask >> k) & 1u)) continue;              /* a gestating option is never selected */
                    if (!in_set(p, clf, k, nx, ny)) continue;
                    if (p->parents[k] != 0 && in_set(p, clf, p->parents[k], nx, ny)) continue;
                    on = k; break;
                }
            int was_declined = (dn == 0 && option_id[e] < 0) ? -option_id[e] : 0;   /* SPEC §4.2: the option this env stays out of although inside its initiation set (a new episode is a new visit) */
            /* SPEC §4.2: an env that stays out of option k is offered k again only every reoffer_period-th step (staggered by env id) */
            int sticky = p->select_rule == 1 && !keep && on >= 1 && on == was_declined && p->reoffer_period > 1 &&
                         ((t + g) % (uint64_t)p->reoffer_period) != 0;
            o_t[i] = o; o_n[i] = sticky ? 0 : on;
            entering[i] = p->select_rule == 1 && !keep && on >= 1 && !sticky; candmax[i] = 0.0f;
            r0[i] = rew; c0[i] = dn ? 0.0f : p->gamma;
            /* SPEC §4.4: gestating options that hold the ENTRY state in their initiation set learn off-policy from this
             * transition, as if the env had been running them (no time-out); their successes are counted */
            gs[i] = 0;
            for (int k = 1; k <= p->n_options; ++k) {
                if (!((p->gest_mask >> k) & 1u) || !in_set(p, clf, k, rec[i].s[0], rec[i].s[1])) continue;
                int par = p->parents[k];
                int succ = (par == 0) ? goal : in_set(p, clf, par, sx, sy);
                int fail = !succ && !in_set(p, clf, k, sx, sy);
                gs[i] |= 1u << k;
                rg[i][k] = (p->exit_rule == 3 ? 0.0f : rew) + (succ ? p->r_option_success : 0.0f);
                cg[i][k] = (dn != 0 || succ || fail) ? 0.0f : p->gamma;
                if (dn == 0 && (succ || fail)) xg[i] |= (1u | (succ ? 2u : 0u)) << (2 * k);
                if (succ && p->gest_succ) {
#pragma omp atomic
                    p->gest_succ[k] += 1;
                }
            }
            /* SPEC §7: trajectory ring (position of s_t) and per-step events */
            if (p->ring_x) {
                size_t row = (size_t)(ep_steps[e] & (p->ring_len - 1)) * N + e;
                p->ring_x[row] = rec[i].s[0]; p->ring_y[row] = rec[i].s[1];
            }
            if (p->events) {
                unsigned inA = 0;
                for (int k = 1; k <= p->n_options; ++k)
                    if (in_set(p, clf, k, sx, sy)) inA |= 1u << k;
                p->events[e] = (uint8_t)((goal ? 1u : 0u) | (inA & 0x3Eu));
                p->ev_len[e] = eps1;
            }
            /* outputs */
            action[e] = (uint8_t)a; reward[e] = rew; done[e] = (uint8_t)dn;
            x[e] = nx; y[e] = ny; vx[e] = nvx; vy[e] = nvy;
            option_id[e] = sticky ? -on : on;
            opt_steps[e] = keep ? opt_steps[e] + 1 : 0;
            ep_steps[e] = dn ? 0 : eps1;
            state_tables(rec[i].s[0], rec[i].s[1], rec[i].s[2], rec[i].s[3], tab_s[i].AB, tab_s[i].CD);
            state_tables(nx, ny, nvx, nvy, tab_n[i].AB, tab_n[i].CD);
        }
        /* ---- TD passes, VF by VF */
        for (int k = 0; k < n_vf; ++k) {
            int m = 0;
            for (int i = 0; i < nb; ++i) {
                int own = (k == 0) || (o_t[i] == k);
                int gst = !own && ((gs[i] >> k) & 1u);
                int upd = own || gst;
                int cache = (o_n[i] == k);
                int want = (k == 0) && entering[i];                             /* the root's value of s_next, to compare with the option's */
                if (!upd && !cache && !want) continue;
                float cont = (k == 0) ? c0[i] : (gst ? cg[i][k] : co[i]);
                items[m].env = i; items[m].upd = upd; items[m].cache = cache; items[m].want = want;
                items[m].tgt = upd && cont > 0.0f;
                items[m].r = (k == 0) ? r0[i] : (gst ? rg[i][k] : ro[i]);
                items[m].cont = cont;
                {   /* SPEC §4.2 exit rule */
                    const unsigned xf = (k == 0) ? 0u : (gst ? (xg[i] >> (2 * k)) & 3u : (own ? xo[i] : 0u));
                    items[m].boot = upd && (xf & 1u) && (p->exit_rule == 2 || (p->exit_rule == 1 && !(xf & 2u)));
                    items[m].boot_v = rootmax[i]; items[m].boot_g = p->gamma;
                }
                ++m;
            }
            block_vf_pass(W + (size_t)k * NACT * NF, m, items, rec, tab_s, tab_n, qcache, N, env_of,
                          P + ((size_t)b * n_vf + k) * NACT * NF, &cnts[(size_t)b * n_vf + k], k == 0 ? rootmax : candmax, k != 0, k == 0 ? rootq : NULL);
        }
        /* SPEC §4.2 value-gated entry: an env about to enter option k does so only if the option promises at least what the root
         * does from there, max_a Q_k(s_next, a) >= max_a Q_0(s_next, a) (both as evaluated above, under this step's weights);
         * otherwise it stays with the root, whose values of s_next become its qcache */
        for (int i = 0; i < nb; ++i) {
            if (!entering[i] || candmax[i] >= rootmax[i]) continue;
            int e = env_of[i];
            option_id[e] = -o_n[i];
            for (int a = 0; a < NACT; ++a) qcache[(size_t)a * N + e] = rootq[i][a];
        }
        free(tab_s); free(tab_n); free(items);
    }
    for (int k = 0; k < n_vf; ++k) {
        for (size_t i = 0; i < (size_t)NACT * NF; ++i)
            G[(size_t)k * NACT * NF + i] = seg_sum(P, (size_t)n_vf * NACT * NF, (size_t)k * NACT * NF + i, nblk);
        int tot = 0;
        for (int b = 0; b < nblk; ++b) tot += cnts[(size_t)b * n_vf + k];
        n_k[k] = tot;
    }
    free(P); free(cnts); free(perm);
}

/* ------------------------------------------------------------------ SPEC §7: example harvest */
void sco_harvest(int n_sel, const int32_t *sel_env, const float *ring_x, const float *ring_y, int ring_len,
                 int n_envs, const int32_t *ev_len, int l_pos, int l_neg, float *out_xy, uint8_t *out_label) {
    const int L = l_pos + l_neg;
    for (int si = 0; si < n_sel; ++si) {
        int e = sel_env[si];
        for (int j = 0; j < L; ++j) {
            size_t t = (size_t)si * L + j;
            int idx = ev_len[e] - 1 - j;
            int ok = idx >= 0 && j < ring_len;
            float x = 0.0f, y = 0.0f;
            if (ok) {
                size_t row = (size_t)(idx & (ring_len - 1)) * n_envs + e;
                x = ring_x[row]; y = ring_y[row];
            }
            out_xy[2 * t] = x; out_xy[2 * t + 1] = y;
            out_label[t] = ok ? (j < l_pos ? 1 : 0) : 255;
        }
    }
}

void sco_collect_examples(int n_envs, const uint8_t *events, uint8_t *prev_in, uint32_t bits, const float *ring_x,
                          const float *ring_y, int ring_len, const int32_t *ev_len, int l_pos, int l_neg,
                          float *ex_xy, uint8_t *ex_label, int32_t *count, int cap) {
    const int L = l_pos + l_neg;
    int pos = *count;
    for (int e = 0; e < n_envs; ++e) {
        int in = (events[e] & bits) != 0, hit = in;
        if (prev_in) { hit = in && !prev_in[e]; prev_in[e] = (uint8_t)in; }
        if (!hit) continue;
        int v = L < ev_len[e] ? L : ev_len[e];
        if (ring_len < v) v = ring_len;
        for (int j = 0; j < v; ++j, ++pos) {
            if (pos >= cap) continue;
            size_t row = (size_t)((ev_len[e] - 1 - j) & (ring_len - 1)) * n_envs + e;
            ex_xy[2 * (size_t)pos] = ring_x[row]; ex_xy[2 * (size_t)pos + 1] = ring_y[row];
            ex_label[pos] = j < l_pos ? 1 : 0;
        }
    }
    *count = pos < cap ? pos : cap;
}

/* ------------------------------------------------------------------ SPEC §6: logistic regression */
int sco_block_envs(void) { return g_block_envs; }

float sco_sigmoid(float z) {
    const float LOG2E = 0x1.715476p+0f, LN2HI = 0x1.63p-1f, LN2LO = -0x1.bd0106p-13f;
    const float E2 = 0x1p-1f, E3 = 0x1.555556p-3f, E4 = 0x1.555556p-5f, E5 = 0x1.111112p-7f,
                E6 = 0x1.6c16c2p-10f, E7 = 0x1.a01a02p-13f;
    float a = -fabsf(z);
    a = fmaxf(a, -87.0f);
    float n = rintf(a * LOG2E);
    float r = fmaf(n, -LN2HI, a);
    r = fmaf(n, -LN2LO, r);
    float pl = E7;
    pl = fmaf(pl, r, E6); pl = fmaf(pl, r, E5); pl = fmaf(pl, r, E4); pl = fmaf(pl, r, E3);
    pl = fmaf(pl, r, E2); pl = fmaf(pl, r, 1.0f); pl = fmaf(pl, r, 1.0f);
    union { uint32_t u; float f; } sc;
    sc.u = (uint32_t)((int)n + 127) << 23;
    float e = pl * sc.f;
    return z >= 0.0f ? 1.0f / (1.0f + e) : e / (1.0f + e);
}

void sco_fit_initiation(int n_fit, const float *xy, const uint8_t *label, const int32_t *offsets,
                        float *w, int iters, float lr, float l2) {
    /* SPEC §6 summation order: G = 8 groups of T = 1024 chains; chain gamma = j T + tau owns examples
     * i = gamma (mod G T) in increasing i; 64 consecutive chains form a butterfly; a group's 16 butterfly results are
     * added in order; the 8 group sums are added in order. */
    enum { G = 8, T = 1024, GT = G * T };
    float (*part)[6] = (float (*)[6])malloc(sizeof(float) * GT * 6);
    for (int q = 0; q < n_fit; ++q) {
        float *wq = w + SCO_CLF_STRIDE * q;
        int i0 = offsets[q], M = offsets[q + 1] - offsets[q];
        if (M <= 0) continue;
        float invM = 1.0f / (float)M;
        for (int it = 0; it < iters; ++it) {
            for (int gamma = 0; gamma < GT; ++gamma) {
                float g[6] = {0, 0, 0, 0, 0, 0};
                for (int i = gamma; i < M; i += GT) {
                    float xx = xy[2 * (size_t)(i0 + i)], yy = xy[2 * (size_t)(i0 + i) + 1];
                    float u = fmaf(xx, 2.0f, -1.0f), v = fmaf(yy, 2.0f, -1.0f);
                    float psi[6] = {1.0f, u, v, u * u, u * v, v * v};
                    float z = clf_z(wq, xx, yy);
                    float e = sco_sigmoid(z) - (float)label[i0 + i];
                    for (int j = 0; j < 6; ++j) g[j] = fmaf(e, psi[j], g[j]);
                }
                for (int j = 0; j < 6; ++j) part[gamma][j] = g[j];
            }
            for (int j = 0; j < 6; ++j) {
                float gs = 0.0f;
                for (int grp = 0; grp < G; ++grp) {
                    float ps = 0.0f;
                    for (int wv = 0; wv < T / 64; ++wv) {
                        float pbuf[64], qbuf[64];
                        for (int l = 0; l < 64; ++l) pbuf[l] = part[grp * T + wv * 64 + l][j];
                        for (int m = 1; m < 64; m <<= 1) {
                            for (int l = 0; l < 64; ++l) qbuf[l] = pbuf[l] + pbuf[l ^ m];
                            memcpy(pbuf, qbuf, sizeof pbuf);
                        }
                        ps = wv == 0 ? pbuf[0] : ps + pbuf[0];
                    }
                    gs = grp == 0 ? ps : gs + ps;
                }
                float reg = (j > 0) ? l2 * wq[j] : 0.0f;
                wq[j] = wq[j] - lr * ((gs * invM) + reg);
            }
        }
    }
    free(part);
}
