/* TEST INFRASTRUCTURE ONLY -- see rl_oracle.h.  Sequential CPU restatement of the ReinLife hot path.
 * Every function cites the reference lines it follows (paths relative to /root/reference/ReinLife). */
#include "rl_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define MAXC 4096 /* grid cells supported (64x64) */

typedef struct {
    int i, j, it, jt;
    int health, age, max_age, gene, brain, uid;
    int dead, reproduced, killed, ate, inter, intra;
    int action;
    double fitness, reward;
    int done;
} agent_t;

typedef struct {
    const rlo_config* cfg;
    int W, H, C, cap, w;
    uint8_t* type; /* [C] view into state */
    int occ[MAXC]; /* agent index at cell or -1 */
    agent_t* ag;   /* [cap] */
    int n;
    uint32_t tick, epoch;
    const rlo_tape* tape;
} world_t;

/* ------------------------------------------------------------------------------------------------------------ */
/* Philox4x32-10 (Salmon et al. 2011) */
void rlo_philox(uint64_t seed, uint32_t epoch, uint32_t world, uint32_t tick, uint32_t site, uint32_t index,
                uint32_t out[4])
{
    uint32_t c0 = index, c1 = site, c2 = tick, c3 = world;
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32) ^ (epoch * 0x9E3779B9u);
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
static inline double u24(uint32_t x) { return (double)(x >> 8) * (1.0 / 16777216.0); }
static inline uint32_t mulhi32(uint32_t x, uint32_t n) { return (uint32_t)(((uint64_t)x * n) >> 32); }

static void draw(const world_t* wd, uint32_t site, uint32_t index, uint32_t out[4])
{
    rlo_philox(wd->cfg->seed, wd->epoch, (uint32_t)(wd->cfg->world_base + wd->w), wd->tick, site, index, out);
}

/* ------------------------------------------------------------------------------------------------------------ */
static inline int wrap(int x, int n) { return x < 0 ? x + n : (x >= n ? x - n : x); }

/* neighbour of (i,j) in direction d: up 0 (i-1), right 1 (j+1), down 2 (i+1), left 3 (j-1); toroidal.
 * utils.py:4-17, environment.py:601-623 / 664-689 */
static inline void neighbour(const world_t* wd, int i, int j, int d, int* ni, int* nj)
{
    *ni = i; *nj = j;
    if (d == 0) *ni = (i == 0) ? wd->H - 1 : i - 1;
    else if (d == 1) *nj = (j == wd->W - 1) ? 0 : j + 1;
    else if (d == 2) *ni = (i == wd->H - 1) ? 0 : i + 1;
    else *nj = (j == 0) ? wd->W - 1 : j - 1;
}

static void load_world(world_t* wd, const rlo_config* cfg, rlo_state* st, int w, const rlo_tape* tape)
{
    wd->cfg = cfg; wd->W = cfg->width; wd->H = cfg->height; wd->C = wd->W * wd->H; wd->cap = cfg->slot_cap; wd->w = w;
    wd->type = st->cell_type + (size_t)w * wd->C;
    wd->n = st->n_agents[w];
    wd->tick = (uint32_t)st->tick[w]; wd->epoch = (uint32_t)st->epoch[w];
    wd->tape = (tape && tape->food_k) ? tape : NULL;
    wd->ag = (agent_t*)calloc((size_t)wd->cap, sizeof(agent_t));
    for (int c = 0; c < wd->C; ++c) wd->occ[c] = -1;
    size_t b = (size_t)w * wd->cap;
    for (int k = 0; k < wd->n; ++k) {
        agent_t* a = &wd->ag[k];
        a->i = st->a_i[b + k]; a->j = st->a_j[b + k]; a->it = a->i; a->jt = a->j;
        a->health = st->a_health[b + k]; a->age = st->a_age[b + k]; a->max_age = st->a_max_age[b + k];
        a->gene = st->a_gene[b + k]; a->brain = st->a_brain[b + k]; a->uid = st->a_uid[b + k];
        int f = st->a_flags[b + k];
        a->dead = !!(f & RLO_F_DEAD); a->reproduced = !!(f & RLO_F_REPRODUCED); a->killed = !!(f & RLO_F_KILLED);
        a->ate = !!(f & RLO_F_ATE_SUPER); a->inter = !!(f & RLO_F_INTER); a->intra = !!(f & RLO_F_INTRA);
        a->action = st->a_action[b + k]; a->fitness = st->a_fitness[b + k];
        wd->occ[a->i * wd->W + a->j] = k;
    }
}

static int flags_of(const agent_t* a)
{
    return (a->dead ? RLO_F_DEAD : 0) | (a->reproduced ? RLO_F_REPRODUCED : 0) | (a->killed ? RLO_F_KILLED : 0) |
           (a->ate ? RLO_F_ATE_SUPER : 0) | (a->inter ? RLO_F_INTER : 0) | (a->intra ? RLO_F_INTRA : 0);
}

/* Grid.get_entities(agent): on-grid agents in row-major order (grid.py:60-67).  Fills list[] with indices into
 * wd->ag, returns the count. */
static int grid_agents(const world_t* wd, int* list)
{
    int n = 0;
    for (int c = 0; c < wd->C; ++c)
        if (wd->type[c] == RLO_AGENT) list[n++] = wd->occ[c];
    return n;
}

/* write the agents list[0..n) back as the world's new row-major list */
static void store_world(world_t* wd, rlo_state* st, const int* list, int n)
{
    size_t b = (size_t)wd->w * wd->cap;
    for (int k = 0; k < n; ++k) {
        const agent_t* a = &wd->ag[list[k]];
        st->a_i[b + k] = (uint8_t)a->i; st->a_j[b + k] = (uint8_t)a->j;
        st->a_health[b + k] = a->health; st->a_age[b + k] = a->age; st->a_max_age[b + k] = a->max_age;
        st->a_gene[b + k] = a->gene; st->a_brain[b + k] = a->brain; st->a_uid[b + k] = a->uid;
        st->a_flags[b + k] = (uint8_t)flags_of(a); st->a_action[b + k] = (int8_t)a->action;
        st->a_fitness[b + k] = a->fitness;
    }
    st->n_agents[wd->w] = n;
}

/* Grid.set_random (grid.py:69-83): list the empty cells row-major, draw an index, draw a coin, place if coin < p.
 * Returns the cell or -1.  `slot` selects the tape entry / Philox index of this call within its site. */
static int set_random(world_t* wd, uint32_t site, int slot, double p, int new_type)
{
    int n_empty = 0;
    for (int c = 0; c < wd->C; ++c) n_empty += (wd->type[c] == RLO_EMPTY);
    if (n_empty == 0) return -2; /* np.random.randint(0, 0) raises ValueError -> None, no draws (grid.py:82-83) */
    int k; double u;
    if (wd->tape) {
        size_t w = (size_t)wd->w;
        if (site == RLO_SITE_FOOD) { k = wd->tape->food_k[w * RLO_FOOD_TRIES + slot]; u = wd->tape->food_u[w * RLO_FOOD_TRIES + slot]; }
        else { k = wd->tape->birth_k[w * (wd->cap + 1) + slot]; u = 0.0; }
    } else {
        uint32_t r[4]; draw(wd, site, (uint32_t)slot, r);
        k = (int)mulhi32(r[0], (uint32_t)n_empty); u = u24(r[1]);
    }
    if (k < 0 || k >= n_empty) return -3; /* tape inconsistent with the world */
    if (!(u < p)) return -1;
    int seen = 0;
    for (int c = 0; c < wd->C; ++c)
        if (wd->type[c] == RLO_EMPTY) { if (seen == k) { wd->type[c] = (uint8_t)new_type; return c; } ++seen; }
    return -3;
}

/* ------------------------------------------------------------------------------------------------------------ */
/* _get_observations (environment.py:313-375) with _prepare_observations (:377-404), _get_food (:432-446),
 * _get_genes (:448-456), _extract_gene_observation (:406-430), Grid.fov (grid.py:90-117).
 * list[] = grid_agents() order.  obs: [n][153] float (the reference builds float64 and casts to float32 at the
 * net input, PERD3QN.py:88 / DQN.py:133 / PPO.py:165; we compute in double and cast once). */
static void observe(const world_t* wd, const int* list, int n, float* obs)
{
    double food_map[MAXC], health_map[MAXC];
    long gene_map[MAXC];
    const int W = wd->W, H = wd->H;
    /* np.vectorize infers the output dtype from the first cell: int64 unless cell (0,0) holds an agent
     * (environment.py:396-398: `obj.health / obj.max_health ... else -1`) */
    const int float_mode = (wd->type[0] == RLO_AGENT);
    for (int c = 0; c < wd->C; ++c) {
        int t = wd->type[c];
        const agent_t* a = (t == RLO_AGENT) ? &wd->ag[wd->occ[c]] : NULL;
        food_map[c] = (t == RLO_FOOD) ? 0.5 : (t == RLO_SUPER) ? 1.0 : (t == RLO_POISON) ? -1.0
                      : (a && a->health < 0) ? 1.0 : 0.0;
        if (a) {
            double v = (double)a->health / 200.0;
            health_map[c] = float_mode ? v : (double)(long)v; /* astype(int64): truncation toward zero */
        } else health_map[c] = -1.0;
        gene_map[c] = (a && a->dead) ? a->gene : -2; /* live agents report -2 (environment.py:450-456) */
    }
    for (int k = 0; k < n; ++k) {
        const agent_t* a = &wd->ag[list[k]];
        float* o = obs + (size_t)k * RLO_OBS_DIM;
        for (int di = -3; di <= 3; ++di)
            for (int dj = -3; dj <= 3; ++dj) {
                int c = wrap(a->i + di, H) * W + wrap(a->j + dj, W);
                int idx = (di + 3) * 7 + (dj + 3);
                o[idx] = (float)food_map[c];
                o[49 + idx] = (float)health_map[c];
                long g = gene_map[c];
                if (g > -1 && g != a->gene) g = -1;
                if (g == a->gene) g = 1;
                if (g == -2) g = 0;
                o[98 + idx] = (float)g;
            }
        int same = 0;
        for (int m = 0; m < n; ++m) same += (wd->ag[list[m]].gene == a->gene);
        o[147] = (float)((double)a->health / 200.0);
        o[148] = a->reproduced ? 1.0f : 0.0f;
        o[149] = (float)((double)same / (double)n);
        o[150] = (float)((double)n / (double)wd->cfg->max_agents);
        o[151] = (float)a->killed;
        o[152] = a->ate ? 1.0f : -1.0f; /* ate_super_food starts at -1 (entities.py:158), set to 1. on eating */
    }
}

/* ------------------------------------------------------------------------------------------------------------ */
/* numpy's pairwise float64 summation (np.add.reduce on a contiguous array, numpy/core/src/umath/loops_utils.h
 * pairwise_sum: < 8 sequential; <= 128 eight accumulators + tail; else split at a multiple of 8) -- np.mean of the
 * tracker's per-agent lists is sum/len with exactly this sum. */
static double np_pairwise_sum(const double* a, int n)
{
    if (n < 8) { double r = 0.0; for (int i = 0; i < n; ++i) r += a[i]; return r; }
    if (n <= 128) {
        double r[8]; int i;
        for (int k = 0; k < 8; ++k) r[k] = a[k];
        for (i = 8; i < n - (n % 8); i += 8) for (int k = 0; k < 8; ++k) r[k] += a[i + k];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    }
    int n2 = n / 2; n2 -= n2 % 8;
    return np_pairwise_sum(a, n2) + np_pairwise_sum(a + n2, n - n2);
}

/* Tracker._track_results over the post-step env.agents (tracker.py:178-266; called from update_env, environment.py:206) */
static void track_world(const world_t* wd, const int* l1, int n1, rlo_step_out* out)
{
    const rlo_config* cfg = wd->cfg;
    const int G = cfg->static_families ? cfg->n_brains : 1;
    double* tick = out->trk_tick + (size_t)wd->w * G * RLO_TRK_VARS;
    double* sum = out->trk_sum + (size_t)wd->w * G * RLO_TRK_VARS;
    int32_t* cnt = out->trk_cnt + (size_t)wd->w * G * RLO_TRK_VARS;
    double* pop = out->trk_pop + (size_t)wd->w * 3;
    double* rew = (double*)malloc(sizeof(double) * (size_t)(n1 + 1));
    int n_distinct = 0;
    for (int k = 0; k < n1; ++k) {
        int seen = 0;
        for (int m = 0; m < k; ++m) seen |= (wd->ag[l1[m]].gene == wd->ag[l1[k]].gene);
        n_distinct += !seen;
    }
    for (int g = 0; g < G; ++g) {
        double v[RLO_TRK_VARS];
        int m = 0, sum_age = 0, best = 0, attacks = 0, kills = 0;
        for (int k = 0; k < n1; ++k) {
            const agent_t* a = &wd->ag[l1[k]];
            if (cfg->static_families && a->gene != g) continue;
            rew[m++] = a->reward; sum_age += a->age; best = a->age > best ? a->age : best;
            attacks += a->action >= 4; kills += a->killed;
        }
        if (n1 == 0) { for (int i = 0; i < RLO_TRK_VARS; ++i) v[i] = -1.0; }  /* tracker.py:191-196 */
        else {
            if (m == 0) { v[0] = v[1] = v[2] = v[3] = v[4] = -1.0; }
            else {
                v[0] = cfg->static_families ? (double)m : (double)n1 / (double)n_distinct; /* np.mean(counts of unique genes) */
                v[1] = (double)sum_age / (double)m;
                v[2] = np_pairwise_sum(rew, m) / (double)m;  /* mean of agent.reward, not fitness (tracker.py:221) */
                v[3] = (double)best;
                v[4] = (double)attacks / (double)m;
            }
            v[5] = (double)kills;                 /* appended even for an empty group (tracker.py:255) */
            v[6] = kills != 0 ? 1.0 : 0.0;        /* intra_killed is computed exactly like killed (tracker.py:252-260) */
        }
        for (int i = 0; i < RLO_TRK_VARS; ++i) {
            tick[g * RLO_TRK_VARS + i] = v[i];
            if (v[i] > -1.0) { sum[g * RLO_TRK_VARS + i] += v[i]; cnt[g * RLO_TRK_VARS + i] += 1; }  /* _aggregate, tracker.py:279-282 */
        }
    }
    pop[0] = n1 == 0 ? -1.0 : (double)n_distinct;
    if (pop[0] > -1.0) { pop[1] += pop[0]; pop[2] += 1.0; }
    free(rew);
}

/* Environment.step (environment.py:160-186) */
static int step_world(world_t* wd, rlo_state* st, const int8_t* actions, rlo_step_out* out)
{
    const int W = wd->W, n0 = wd->n, w = wd->w, cap = wd->cap;
    agent_t* ag = wd->ag;
    /* _act prologue (environment.py:267-271) */
    for (int k = 0; k < n0; ++k) {
        agent_t* a = &ag[k];
        a->action = actions[(size_t)w * cap + k];
        a->health = a->health - 10 < 200 ? a->health - 10 : 200;
        a->age = a->age + 1 < a->max_age ? a->age + 1 : a->max_age;
        a->killed = a->inter = a->intra = 0;
    }
    /* _attack (environment.py:652-699): sequential in list order */
    for (int k = 0; k < n0; ++k) {
        agent_t* a = &ag[k];
        if (a->dead || a->action < 4 || a->action > 7) continue;
        int ti, tj; neighbour(wd, a->i, a->j, a->action - 4, &ti, &tj);
        if (wd->type[ti * W + tj] == RLO_AGENT) {
            agent_t* t = &ag[wd->occ[ti * W + tj]];
            t->health = 0;                                              /* is_attacked, entities.py:183-185 */
            a->health = a->health + 100 < 200 ? a->health + 100 : 200;  /* execute_attack, entities.py:178-181 */
            a->killed = 1;
            if (t->gene == a->gene) a->inter = 1; else a->intra = 1;
        }
    }
    /* _prepare_movement (environment.py:591-625); actions outside 0..7 are a no-op here (the reference raises) */
    for (int k = 0; k < n0; ++k) {
        agent_t* a = &ag[k];
        a->it = a->i; a->jt = a->j;
        if (a->action >= 0 && a->action <= 3 && !a->dead) neighbour(wd, a->i, a->j, a->action, &a->it, &a->jt);
    }
    /* _execute_movement (environment.py:627-650) + _get_impossible_coordinates (:717-726) */
    {
        int count[MAXC];
        int any;
        do {
            memset(count, 0, sizeof(int) * (size_t)wd->C);
            for (int k = 0; k < n0; ++k) count[ag[k].it * W + ag[k].jt]++;
            any = 0;
            for (int c = 0; c < wd->C; ++c) any |= (count[c] > 1);
            for (int k = 0; k < n0; ++k)
                if (count[ag[k].it * W + ag[k].jt] > 1) { ag[k].it = ag[k].i; ag[k].jt = ag[k].j; }
        } while (any);
    }
    for (int k = 0; k < n0; ++k) {
        agent_t* a = &ag[k];
        if (a->action < 0 || a->action > 3) continue;
        int tc = a->it * W + a->jt, oc = a->i * W + a->j;
        /* _eat (environment.py:701-715) */
        if (wd->type[tc] == RLO_FOOD) a->health = a->health + 40 < 200 ? a->health + 40 : 200;
        else if (wd->type[tc] == RLO_POISON) a->health = a->health - 40 < 200 ? a->health - 40 : 200;
        else if (wd->type[tc] == RLO_SUPER) {
            a->health = a->health + 40 < 200 ? a->health + 40 : 200;
            a->max_age = (int)((double)a->max_age * 1.2);
            a->ate = 1;
        }
        /* _update_agent_position (environment.py:778-782): sequential overwrite => "vanish" rule */
        wd->type[oc] = RLO_EMPTY; wd->occ[oc] = -1;
        wd->type[tc] = RLO_AGENT; wd->occ[tc] = k;
        a->i = a->it; a->j = a->jt;
    }
    /* _update_death_status (environment.py:789-793) */
    for (int k = 0; k < n0; ++k)
        if (ag[k].health <= 0 || ag[k].age == ag[k].max_age) ag[k].dead = 1;
    /* _get_rewards (environment.py:277-311) over the _act list (vanished agents included) */
    for (int k = 0; k < n0; ++k) {
        agent_t* a = &ag[k];
        int kin = 0, alive = 0;
        for (int m = 0; m < n0; ++m) { kin += (!ag[m].dead && ag[m].gene == a->gene); alive += !ag[m].dead; }
        kin = kin - 1 > 0 ? kin - 1 : 0;
        double r; a->done = 0;
        if (a->dead) { r = (double)(-alive + kin); a->done = 1; }
        else if (alive == 1) r = 0.0;
        else r = (double)kin / (double)alive;
        if (a->killed && wd->cfg->incentivize_killing) r += 0.2;
        a->reward = r; a->fitness += r; /* update_rl_stats, entities.py:187-192 */
    }
    /* best_agents hold *references* (environment.py:739): their fitness keeps growing while they live */
    if (!wd->cfg->static_families)
        for (int b = 0; b < RLO_N_BEST; ++b)
            for (int k = 0; k < n0; ++k)
                if (st->best_uid[(size_t)w * RLO_N_BEST + b] == ag[k].uid && ag[k].uid >= 0)
                    st->best_fit[(size_t)w * RLO_N_BEST + b] = ag[k].fitness;
    /* _add_food (environment.py:763-776) */
    {
        int nf = 0, np_ = 0, ns = 0, rc;
        for (int c = 0; c < wd->C; ++c) nf += (wd->type[c] == RLO_FOOD);
        if ((double)nf <= (double)wd->C / 10.0)
            for (int t = 0; t < 3; ++t) { rc = set_random(wd, RLO_SITE_FOOD, t, 0.2, RLO_FOOD); if (rc == -3) return -3; }
        for (int c = 0; c < wd->C; ++c) np_ += (wd->type[c] == RLO_POISON);
        if ((double)np_ <= (double)wd->C / 20.0)
            for (int t = 0; t < 3; ++t) { rc = set_random(wd, RLO_SITE_FOOD, 3 + t, 0.2, RLO_POISON); if (rc == -3) return -3; }
        for (int c = 0; c < wd->C; ++c) ns += (wd->type[c] == RLO_SUPER);
        if (ns == 0) { rc = set_random(wd, RLO_SITE_FOOD, 6, 1.0, RLO_SUPER); if (rc == -3) return -3; }
    }
    /* _get_observations (environment.py:186,313-375) -> env.agents becomes the post-move grid list */
    int* l1 = (int*)malloc(sizeof(int) * (size_t)cap);
    int n1 = grid_agents(wd, l1);
    size_t b = (size_t)w * cap;
    if (out) {
        if (out->n_acted) out->n_acted[w] = n0;
        if (out->obs) observe(wd, l1, n1, out->obs + b * RLO_OBS_DIM);
        for (int k = 0; k < n1; ++k) {
            if (out->reward) out->reward[b + k] = (float)ag[l1[k]].reward;
            if (out->done) out->done[b + k] = (uint8_t)ag[l1[k]].done;
            if (out->src) out->src[b + k] = (int16_t)l1[k];
        }
        if (out->trk_tick) track_world(wd, l1, n1, out);
        for (int k = 0; k < n0; ++k) {
            if (out->l0_health) out->l0_health[b + k] = ag[k].health;
            if (out->l0_flags) out->l0_flags[b + k] = (uint8_t)flags_of(&ag[k]);
            if (out->l0_reward) out->l0_reward[b + k] = ag[k].reward;
            if (out->l0_i) out->l0_i[b + k] = (uint8_t)ag[k].i;
            if (out->l0_j) out->l0_j[b + k] = (uint8_t)ag[k].j;
        }
    }
    store_world(wd, st, l1, n1);
    free(l1);
    return 0;
}

/* ------------------------------------------------------------------------------------------------------------ */
static void new_agent(world_t* wd, rlo_state* st, int idx, int cell, int gene, int brain)
{
    agent_t* a = &wd->ag[idx];
    memset(a, 0, sizeof(*a));
    a->i = cell / wd->W; a->j = cell % wd->W; a->it = a->i; a->jt = a->j;
    a->health = 200; a->age = 0; a->max_age = 50; a->gene = gene; a->brain = brain; a->action = -1; /* entities.py:145-159 */
    a->uid = st->next_uid[wd->w]++;
    wd->occ[cell] = idx;
}

/* Environment.update_env (environment.py:188-215), tracker excluded */
static int update_world(world_t* wd, rlo_state* st, rlo_update_out* out)
{
    const int w = wd->w, cap = wd->cap, W = wd->W;
    const rlo_config* cfg = wd->cfg;
    agent_t* ag = wd->ag;
    const int n1 = wd->n; /* the stored list IS grid.get_entities(agent) (environment.py:210) */
    int n_all = n1;       /* next free index in ag[] */
    int32_t* buid = st->best_uid + (size_t)w * RLO_N_BEST;
    double* bfit = st->best_fit + (size_t)w * RLO_N_BEST;
    int32_t* bbr = st->best_brain + (size_t)w * RLO_N_BEST;
    /* _update_best_agents (environment.py:728-739) */
    if (!cfg->static_families) {
        int mi = 0;
        for (int b = 1; b < RLO_N_BEST; ++b) if (bfit[b] < bfit[mi]) mi = b; /* np.argmin: first minimum */
        if (n1 > 0) {
            int xi = 0;
            for (int k = 1; k < n1; ++k) if (ag[k].fitness > ag[xi].fitness) xi = k; /* np.argmax: first maximum */
            int present = 0;
            for (int b = 0; b < RLO_N_BEST; ++b) present |= (buid[b] == ag[xi].uid);
            if (!present && ag[xi].fitness > bfit[mi]) { buid[mi] = ag[xi].uid; bfit[mi] = ag[xi].fitness; bbr[mi] = ag[xi].brain; }
        }
    }
    /* _reproduce (environment.py:488-519); _get_empty_within_fov (:549-589) compares Entity objects with 0 and is
     * therefore always [] -> offspring always go through Grid.set_random(Agent, p=1) */
    int n_elig = 0, n_birth = 0;
    for (int k = 0; k < n1; ++k) {
        agent_t* a = &ag[k];
        if (!(!a->dead && !a->reproduced && a->age > 5)) continue; /* can_reproduce, entities.py:244-248 */
        if (!(n1 <= cfg->max_agents)) continue;
        double u;
        if (wd->tape) u = wd->tape->repro_u[(size_t)w * cap + n_elig];
        else { uint32_t r[4]; draw(wd, RLO_SITE_REPRO, (uint32_t)n_elig, r); u = u24(r[0]); }
        ++n_elig;
        if (!(u > 0.95)) continue;
        int brain = cfg->static_families ? a->gene : a->brain;
        int cell = set_random(wd, RLO_SITE_BIRTH, n_birth, 1.0, RLO_AGENT);
        if (cell == -3) return -3;
        if (cell != -2) ++n_birth;
        if (cell >= 0) { if (n_all >= cap) return -4; new_agent(wd, st, n_all++, cell, a->gene, brain); }
        if (cfg->limit_reproduction) a->reproduced = 1;
    }
    /* _produce (environment.py:521-547) */
    if (n1 <= cfg->max_agents) {
        double u; uint32_t r[4] = {0, 0, 0, 0};
        if (wd->tape) u = wd->tape->produce_u[w];
        else { draw(wd, RLO_SITE_PRODUCE, 0, r); u = u24(r[0]); }
        if (u > 0.95) {
            int gene, brain;
            if (cfg->static_families) {
                if (wd->tape) gene = wd->tape->produce_choice[w];
                else {
                    int cnt = 0, pick = -1;
                    for (int g = 0; g < cfg->n_brains; ++g) {
                        int present = 0;
                        for (int k = 0; k < n1; ++k) present |= (ag[k].gene == g);
                        cnt += !present;
                    }
                    if (cnt > 0) {
                        int want = (int)mulhi32(r[1], (uint32_t)cnt), seen = 0;
                        for (int g = 0; g < cfg->n_brains && pick < 0; ++g) {
                            int present = 0;
                            for (int k = 0; k < n1; ++k) present |= (ag[k].gene == g);
                            if (!present) { if (seen == want) pick = g; ++seen; }
                        }
                        gene = pick;
                    } else gene = (int)mulhi32(r[1], (uint32_t)cfg->n_brains);
                }
                if (gene < 0 || gene >= cfg->n_brains) return -3;
                brain = gene;
            } else {
                st->max_gene[w] += 1; /* incremented even when the placement fails (environment.py:543) */
                int c = wd->tape ? wd->tape->produce_choice[w] : (int)mulhi32(r[1], RLO_N_BEST);
                if (c < 0 || c >= RLO_N_BEST) return -3;
                gene = st->max_gene[w]; brain = bbr[c];
            }
            int cell = set_random(wd, RLO_SITE_BIRTH, n_birth, 1.0, RLO_AGENT);
            if (cell == -3) return -3;
            if (cell != -2) ++n_birth;
            if (cell >= 0) { if (n_all >= cap) return -4; new_agent(wd, st, n_all++, cell, gene, brain); }
        }
    }
    /* _remove_dead_agents (environment.py:795-799): corpses become Food */
    for (int k = 0; k < n1; ++k)
        if (ag[k].dead) { int c = ag[k].i * W + ag[k].j; wd->type[c] = RLO_FOOD; wd->occ[c] = -1; }
    /* _get_observations + _update_agents_state (environment.py:214-215) */
    int* l2 = (int*)malloc(sizeof(int) * (size_t)cap);
    int n2 = grid_agents(wd, l2);
    size_t b = (size_t)w * cap;
    if (out) {
        if (out->obs) observe(wd, l2, n2, out->obs + b * RLO_OBS_DIM);
        if (out->src) for (int k = 0; k < n2; ++k) out->src[b + k] = (int16_t)(l2[k] < n1 ? l2[k] : -1);
    }
    store_world(wd, st, l2, n2);
    st->tick[w] += 1;
    free(l2);
    return 0;
}

int rlo_step(const rlo_config* cfg, rlo_state* st, const int8_t* actions, const rlo_tape* tape, rlo_step_out* out,
             int w0, int w1)
{
    if (cfg->width * cfg->height > MAXC) return -1;
    for (int w = w0; w < w1; ++w) {
        world_t wd; load_world(&wd, cfg, st, w, tape);
        int rc = step_world(&wd, st, actions, out);
        free(wd.ag);
        if (rc) return rc;
    }
    return 0;
}

int rlo_update(const rlo_config* cfg, rlo_state* st, const rlo_tape* tape, rlo_update_out* out, int w0, int w1)
{
    if (cfg->width * cfg->height > MAXC) return -1;
    for (int w = w0; w < w1; ++w) {
        world_t wd; load_world(&wd, cfg, st, w, tape);
        int rc = update_world(&wd, st, out);
        free(wd.ag);
        if (rc) return rc;
    }
    return 0;
}

int rlo_observe(const rlo_config* cfg, const rlo_state* st, float* obs, int w0, int w1)
{
    if (cfg->width * cfg->height > MAXC) return -1;
    for (int w = w0; w < w1; ++w) {
        world_t wd; load_world(&wd, cfg, (rlo_state*)st, w, NULL);
        int* l = (int*)malloc(sizeof(int) * (size_t)wd.cap);
        int n = grid_agents(&wd, l);
        observe(&wd, l, n, obs + (size_t)w * wd.cap * RLO_OBS_DIM);
        free(l); free(wd.ag);
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------------------------ */
/* Synthetic world generator (the build's own rule, identical on CPU and GPU, parallel by design):
 *   every cell c draws r = Philox(seed, epoch, world, tick 0, site RESET_CELL, index c);
 *   cells are ranked by the unique key (r.x & ~0xFFF) | c  (a uniformly random permutation of the cells);
 *   n_food = #{c : u24(r.z) < 0.1}, n_poison = #{c : u24(r.w) < 0.05}  (the Binomial(H*W, p) counts that
 *   Environment._init_food's H*W coin flips produce, environment.py:741-761);
 *   rank < n_agents -> Agent (gene = brain = mulhi(r.y, n_brains), health 200, age 0), the next n_food ranks Food,
 *   the next n_poison ranks Poison, the next one SuperFood.  Agents are listed row-major; uid = list index.
 * families != 0: Environment.reset()'s population instead (environment.py:147-149: one agent per brain, gene = brain = its index,
 *   each at a uniformly random cell): n_agents = n_brains and the agent of key rank p gets gene p. */
typedef struct { uint32_t key; int cell; } keyed_t;
static int cmp_keyed(const void* a, const void* b)
{
    uint32_t x = ((const keyed_t*)a)->key, y = ((const keyed_t*)b)->key;
    return x < y ? -1 : (x > y ? 1 : 0);
}

static void reset_world(const rlo_config* cfg, rlo_state* st, int w, int n_agents, float* obs, int families)
{
    world_t wd; memset(&wd, 0, sizeof(wd));
    wd.cfg = cfg; wd.W = cfg->width; wd.H = cfg->height; wd.C = wd.W * wd.H; wd.cap = cfg->slot_cap; wd.w = w;
    wd.type = st->cell_type + (size_t)w * wd.C;
    wd.tick = 0; wd.epoch = (uint32_t)st->epoch[w];
    wd.ag = (agent_t*)calloc((size_t)wd.cap, sizeof(agent_t));
    memset(wd.type, 0, (size_t)wd.C);
    for (int c = 0; c < wd.C; ++c) wd.occ[c] = -1;
    st->next_uid[w] = 0; st->max_gene[w] = cfg->n_brains; st->tick[w] = 0;
    for (int b = 0; b < RLO_N_BEST; ++b) {
        st->best_uid[(size_t)w * RLO_N_BEST + b] = -1; st->best_fit[(size_t)w * RLO_N_BEST + b] = 0.0;
        st->best_brain[(size_t)w * RLO_N_BEST + b] = 0;
    }
    keyed_t* ks = (keyed_t*)malloc(sizeof(keyed_t) * (size_t)wd.C);
    uint32_t* gdraw = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)wd.C);
    int nf = 0, np_ = 0;
    for (int c = 0; c < wd.C; ++c) {
        uint32_t r[4]; draw(&wd, RLO_SITE_RESET_AGENT, (uint32_t)c, r);
        ks[c].key = (r[0] & ~0xFFFu) | (uint32_t)c; ks[c].cell = c; gdraw[c] = r[1];
        nf += (u24(r[2]) < 0.1); np_ += (u24(r[3]) < 0.05);
    }
    qsort(ks, (size_t)wd.C, sizeof(keyed_t), cmp_keyed);
    const int na = n_agents < wd.C ? n_agents : wd.C;
    for (int p = 0; p < wd.C; ++p) {
        int c = ks[p].cell;
        if (families && p < na) gdraw[c] = (uint32_t)p;   /* the gene itself */
        wd.type[c] = p < na ? RLO_AGENT : p < na + nf ? RLO_FOOD : p < na + nf + np_ ? RLO_POISON : p == na + nf + np_ ? RLO_SUPER : RLO_EMPTY;
    }
    int idx = 0;
    for (int c = 0; c < wd.C; ++c)
        if (wd.type[c] == RLO_AGENT) { int g = families ? (int)gdraw[c] : (int)mulhi32(gdraw[c], (uint32_t)cfg->n_brains); new_agent(&wd, st, idx++, c, g, g); }
    free(ks); free(gdraw);
    int* l = (int*)malloc(sizeof(int) * (size_t)wd.cap);
    int n = grid_agents(&wd, l);
    if (obs) observe(&wd, l, n, obs + (size_t)w * wd.cap * RLO_OBS_DIM);
    store_world(&wd, st, l, n);
    free(l); free(wd.ag);
}

int rlo_reset_synthetic(const rlo_config* cfg, rlo_state* st, int n_agents, float* obs, int w0, int w1)
{
    if (cfg->width * cfg->height > MAXC || n_agents > cfg->slot_cap) return -1;
    for (int w = w0; w < w1; ++w) reset_world(cfg, st, w, n_agents, obs, 0);
    return 0;
}

int rlo_reset_families(const rlo_config* cfg, rlo_state* st, float* obs, int w0, int w1)
{
    if (cfg->width * cfg->height > MAXC || cfg->n_brains > cfg->slot_cap || cfg->n_brains > cfg->width * cfg->height) return -1;
    for (int w = w0; w < w1; ++w) reset_world(cfg, st, w, cfg->n_brains, obs, 1);
    return 0;
}

int rlo_refill(const rlo_config* cfg, rlo_state* st, int threshold, int n_agents, float* obs, int w0, int w1)
{
    int cnt = 0;
    if (cfg->width * cfg->height > MAXC || n_agents > cfg->slot_cap) return -1;
    for (int w = w0; w < w1; ++w)
        if (st->n_agents[w] < threshold) { st->epoch[w] += 1; reset_world(cfg, st, w, n_agents, obs, 0); ++cnt; }
    return cnt;
}

/* ------------------------------------------------------------------------------------------------------------ */
/* policy forward */
static void linear(const float* w, const float* b, const float* x, int n_in, int n_out, float* y, int relu)
{
    for (int o = 0; o < n_out; ++o) {
        float acc = b[o];
        for (int k = 0; k < n_in; ++k) acc = fmaf(x[k], w[(size_t)o * n_in + k], acc);
        y[o] = relu && acc < 0.0f ? 0.0f : acc;
    }
}

int64_t rlo_policy_n_params(int kind)
{
    if (kind == RLO_DQN) return 153 * 128 + 128 + 128 * 64 + 64 + 64 * 8 + 8;
    if (kind == RLO_D3QN || kind == RLO_PERD3QN) return 153 * 128 + 128 + 2 * (128 * 128 + 128) + 128 * 8 + 8 + 128 + 1;
    if (kind == RLO_PPO) return 153 * 256 + 256 + 256 * 256 + 256 + 256 * 8 + 8 + 256 + 1;
    return -1;
}

int rlo_policy_forward(int kind, const float* wts, const float* obs, int n_rows, float* out)
{
    float h1[256], h2[256], h3[256];
    for (int r = 0; r < n_rows; ++r) {
        const float* x = obs + (size_t)r * RLO_OBS_DIM;
        float* y = out + (size_t)r * 8;
        const float* p = wts;
        if (kind == RLO_DQN) { /* Qnet.forward, DQN.py:126-130 */
            linear(p, p + 153 * 128, x, 153, 128, h1, 1); p += 153 * 128 + 128;
            linear(p, p + 128 * 64, h1, 128, 64, h2, 1); p += 128 * 64 + 64;
            linear(p, p + 64 * 8, h2, 64, 8, y, 0);
        } else if (kind == RLO_D3QN || kind == RLO_PERD3QN) { /* forward, D3QN.py:161-165 / PERD3QN.py:198-202 */
            float adv[8], val[1];
            linear(p, p + 153 * 128, x, 153, 128, h1, 1); p += 153 * 128 + 128; /* relu(feature) feeds both branches */
            linear(p, p + 128 * 128, h1, 128, 128, h2, 1); p += 128 * 128 + 128;
            linear(p, p + 128 * 8, h2, 128, 8, adv, 0); p += 128 * 8 + 8;
            linear(p, p + 128 * 128, h1, 128, 128, h3, 1); p += 128 * 128 + 128;
            linear(p, p + 128, h3, 128, 1, val, 0);
            float mean = 0.0f; /* advantage.mean() over the [1,8] tensor == per-row mean */
            for (int a = 0; a < 8; ++a) mean += adv[a];
            mean *= 0.125f;
            for (int a = 0; a < 8; ++a) y[a] = adv[a] + val[0] - mean;
        } else if (kind == RLO_PPO) { /* PPO.pi, PPO.py:101-106 */
            float lg[8];
            linear(p, p + 153 * 256, x, 153, 256, h1, 1); p += 153 * 256 + 256;
            linear(p, p + 256 * 256, h1, 256, 256, h2, 1); p += 256 * 256 + 256;
            linear(p, p + 256 * 8, h2, 256, 8, lg, 0);
            float m = lg[0], s = 0.0f;
            for (int a = 1; a < 8; ++a) m = lg[a] > m ? lg[a] : m;
            for (int a = 0; a < 8; ++a) { y[a] = expf(lg[a] - m); s += y[a]; }
            for (int a = 0; a < 8; ++a) y[a] /= s;
        } else return -1;
    }
    return 0;
}

int rlo_select_actions(const rlo_config* cfg, int kind, const float* out, int n_rows, const int32_t* world_of_row,
                       const int32_t* index_in_world, const int32_t* tick_of_world, const int32_t* epoch_of_world,
                       float eps, int8_t* actions)
{
    for (int r = 0; r < n_rows; ++r) {
        const float* y = out + (size_t)r * 8;
        uint32_t rn[4];
        int w = world_of_row[r];
        rlo_philox(cfg->seed, (uint32_t)epoch_of_world[w], (uint32_t)(cfg->world_base + w), (uint32_t)tick_of_world[w], RLO_SITE_ACT, (uint32_t)index_in_world[r], rn);
        float u = (float)u24(rn[0]);
        int a = 0;
        if (kind == RLO_PPO) { /* Categorical(prob).sample(), PPO.py:166-167, as inverse CDF */
            float cum = 0.0f; a = 7;
            for (int k = 0; k < 8; ++k) { cum += y[k]; if (u < cum) { a = k; break; } }
        } else if (u < eps) a = (int)(rn[1] >> 29); /* random action, DQN.py:136-137 / PERD3QN.py:208-209 */
        else for (int k = 1; k < 8; ++k) if (y[k] > y[a]) a = k; /* argmax, first maximum */
        actions[r] = (int8_t)a;
    }
    return 0;
}
