/* reinlife_hip.h -- C ABI of libreinlife_hip.so: the MI355X (gfx950) implementation of ReinLife's hot path.
 *
 * The reference (MaartenGr/ReinLife v1.0.1) is pure Python and has no FFI; this ABI is what a binding for its
 * per-tick path would call.  Each entry point names the reference code it replaces (paths under ReinLife/):
 *
 *   rl_step            Environment.step()                         World/environment.py:160-186
 *   rl_update          Environment.update_env() minus the Tracker World/environment.py:188-215
 *   rl_tick            step() + update_env() of one trainer-loop iteration, fused (Helpers/trainer.py:92,99)
 *   rl_observe         Environment._get_observations()            World/environment.py:313-375
 *   rl_reset_synthetic Environment.reset()-style world generator  World/environment.py:133-158, 741-761
 *   rl_reset_families  Environment.reset() itself (one agent per brain) World/environment.py:133-158
 *   rl_policy_act      Agent.get_action() over all agents         World/entities.py:215-222 ->
 *                      DQN.py:126-139, D3QN.py:161-173, PERD3QN.py:198-210, PPO.py:101-106,164-169
 *   rl_policy_forward  the bare network forward of one brain on a dense batch of observation rows
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; no torch / C++ types.  `stream` is a hipStream_t passed as void*.
 *   - The CALLER owns every device buffer (e.g. torch tensors) and the stream; the library allocates no device
 *     memory, starts no threads and never synchronises the device.  A handle is not thread-safe; distinct handles
 *     are independent.
 *   - Every function returns 0 on success or a negative rl_status; rl_last_error() gives the message (thread-local).
 *   - World state is struct-of-arrays over `n_worlds` independent worlds.  A world's agent list is ALWAYS its
 *     on-grid agents in row-major cell order (= Grid.get_entities order, World/grid.py:60-67), so index k is index
 *     k of the reference's env.agents.
 *   - Randomness: either a recorded tape of the reference's draws (parity mode) or Philox4x32-10 keyed
 *     (seed, epoch, world, tick, site, index) generated in-kernel (performance mode).  See DESIGN.md.
 */
#ifndef REINLIFE_HIP_H
#define REINLIFE_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* libreinlife_hip.so is built with -fvisibility=hidden: the declarations between this push and the pop at the end of the file ARE its
 * export list (tests/test_abi_cpu.py compares the library's dynamic symbol table with them). */
#pragma GCC visibility push(default)

#define RL_OBS_DIM 153   /* Environment.observation_space, environment.py:119 */
#define RL_N_ACTIONS 8   /* Environment.action_space, environment.py:118; World/utils.py:4-17 */
#define RL_N_BEST 10     /* len(best_agents), environment.py:149 */
#define RL_FOOD_TRIES 7  /* 3 Food + 3 Poison + 1 SuperFood set_random calls, environment.py:763-776 */
#define RL_MAX_CELLS 4096
#define RL_MAX_BRAINS 64
#define RL_TRK_VARS 7

typedef enum {
    RL_OK = 0, RL_E_INVALID = -1, RL_E_UNBOUND = -2, RL_E_LAUNCH = -3, RL_E_UNSUPPORTED = -4
} rl_status;

/* cell codes: World/utils.py:20-33 (kin = 4 is never placed on the grid) */
enum { RL_EMPTY = 0, RL_FOOD = 1, RL_POISON = 2, RL_AGENT = 3, RL_KIN = 4, RL_SUPER_FOOD = 5 };
/* a_flags bits (Agent booleans, World/entities.py:150-159) */
enum { RL_F_DEAD = 1, RL_F_REPRODUCED = 2, RL_F_KILLED = 4, RL_F_ATE_SUPER = 8, RL_F_INTER_KILLED = 16,
       RL_F_INTRA_KILLED = 32 };
/* brain kinds (BasicBrain.method, Models/utils.py:1-14) */
enum { RL_DQN = 0, RL_D3QN = 1, RL_PERD3QN = 2, RL_PPO = 3 };
/* Philox draw sites */
enum { RL_SITE_FOOD = 1, RL_SITE_REPRO = 2, RL_SITE_BIRTH = 3, RL_SITE_PRODUCE = 4, RL_SITE_ACT = 5,
       RL_SITE_RESET_AGENT = 6, RL_SITE_RESET_FOOD = 7, RL_SITE_RESET_POISON = 8, RL_SITE_RESET_SUPER = 9 };

/* keyword arguments of Environment(...) that matter on the path (environment.py:74-89) */
typedef struct {
    int32_t width, height;
    int32_t max_agents;
    int32_t n_brains;            /* len(brains) */
    int32_t slot_cap;            /* capacity of the per-world agent arrays; multiple of 64, >= 2*max_agents+2, <= 4096 */
    int32_t n_worlds;            /* independent replicas on this GPU, < 2^19 */
    int32_t static_families, limit_reproduction, incentivize_killing;
    int32_t world_base;          /* global id of world 0 (replica sharding across GPUs): Philox uses world_base + w */
    uint64_t seed;               /* Philox key */
} rl_config;

/* Device pointers, all caller-owned.  Shapes: R = n_worlds, C = width*height, cap = slot_cap. */
typedef struct {
    uint8_t* cell_type;   /* [R][C]   Grid.get_numpy() */
    int32_t* n_agents;    /* [R]      len(env.agents) */
    uint8_t* a_i;         /* [R][cap] Agent.i */
    uint8_t* a_j;         /* [R][cap] Agent.j */
    int32_t* a_health;    /* [R][cap] */
    int32_t* a_age;
    int32_t* a_max_age;
    int32_t* a_gene;
    int32_t* a_brain;     /* index into the brains list the agent's brain descends from */
    int32_t* a_uid;       /* creation-order id within the world (object identity in the reference) */
    uint8_t* a_flags;
    int8_t* a_action;     /* last action taken, -1 for newborns (entities.py:153) */
    double* a_fitness;    /* Agent.fitness (float64 sum of rewards, entities.py:189) */
    int32_t* max_gene;    /* [R] Environment.max_gene */
    int32_t* next_uid;    /* [R] */
    int32_t* tick;        /* [R] ticks since the last reset (Philox counter) */
    int32_t* epoch;       /* [R] resets so far (Philox key) */
    int32_t* best_uid;    /* [R][10] Environment.best_agents (non-static families) */
    double* best_fit;     /* [R][10] */
    int32_t* best_brain;  /* [R][10] */
} rl_state;

/* One tick of recorded reference draws (device pointers).  food_k == NULL => Philox. */
typedef struct {
    const int32_t* food_k;         /* [R][7]     np.random.randint results of _add_food's set_random calls */
    const double* food_u;          /* [R][7]     their np.random.random coins */
    const double* repro_u;         /* [R][cap]   random.random() gate of the k-th eligible agent (env.py:501) */
    const int32_t* birth_k;        /* [R][cap+1] np.random.randint result of the b-th birth placement */
    const double* produce_u;       /* [R]        random.random() gate of _produce (env.py:528) */
    const int32_t* produce_choice; /* [R]        static: chosen gene (env.py:536/538); else best_agents index (:544) */
} rl_tape;

/* Outputs of a step, indexed in the POST-step env.agents order.  Any pointer may be NULL. */
typedef struct {
    int32_t* n_acted;   /* [R]       agents that acted = agent-steps of this tick */
    float* reward;      /* [R][cap]  Agent.reward */
    uint8_t* done;      /* [R][cap]  Agent.done */
    int16_t* src;       /* [R][cap]  index of the agent in the PRE-step list (its state/action live there) */
    float* obs;         /* [R][cap][153] Agent.state_prime */
    unsigned long long* acted_total; /* [1] running sum of n_acted over worlds and calls (metric counter) */
    /* Tracker accumulators (Helpers/tracker.py:178-282), all or none.  G = n_brains (static families) or 1 groups x
     * RL_TRK_VARS variables [size, age, fitness, best age, attacks, kills, intra kills] over the post-step env.agents;
     * a per-tick value enters sum/cnt iff it is > -1, exactly like Tracker._aggregate. */
    double* trk_tick;   /* [R][G][7] this tick's values (-1 = no result) */
    double* trk_sum;    /* [R][G][7] running sum of valid values (caller zeroes it at interval boundaries) */
    int32_t* trk_cnt;   /* [R][G][7] number of valid ticks */
    double* trk_pop;    /* [R][3]    "Avg Number of Populations": this tick, running sum, running count */
    /* post-step list attributes that a fused rl_tick would otherwise lose (needed by rl_capture_transitions) */
    int32_t* n_post;    /* [R]       length of the post-step list (len(env.agents) after step()) */
    int32_t* age;       /* [R][cap]  Agent.age after the step */
    int32_t* brain;     /* [R][cap]  brains-list index of the agent's brain */
} rl_step_out;

/* Outputs of an update, indexed in the POST-update env.agents order. */
typedef struct {
    int16_t* src;       /* [R][cap]  index in the post-step list, -1 for newborns */
    float* obs;         /* [R][cap][153] Agent.state (what the policy reads next tick) */
} rl_update_out;

/* Replay ring of ONE brain (caller-owned device buffers): what Agent.learn hands to brain.learn each tick
 * (World/entities.py:194-208 -> DQN.py:73-84, D3QN.py:94-96, PERD3QN.py:91-92,117-125, PPO.py:69-76), stored in
 * slot (count % capacity) in Agent.learn call order per world. */
typedef struct {
    float* state;                /* [capacity][153] observation the policy read before the step */
    float* state_prime;          /* [capacity][153] post-step observation */
    int8_t* action;              /* [capacity] */
    float* reward;               /* [capacity] raw Agent.reward (the PPO brain divides by 100 itself, PPO.py:71) */
    uint8_t* done;               /* [capacity] */
    float* prob;                 /* [capacity] policy output of the taken action (PPO: prob[action]) if outputs are given */
    int32_t* age;                /* [capacity] post-step age (train_freq logic, e.g. DQN.py:86) */
    unsigned long long* count;   /* [1] transitions stored so far */
    int64_t capacity;
} rl_replay;
#define RL_MAX_CAPTURE_BRAINS 16

/* One brain of the brains list. */
typedef struct {
    int32_t kind;            /* RL_DQN .. RL_PPO */
    float epsilon;           /* exploration rate (0 = greedy, training=False); ignored for PPO (always samples) */
    const float* packed;     /* device: weights in the layout produced by rl_policy_pack_weights */
} rl_brain;

typedef struct rl_world rl_world;

const char* rl_last_error(void);
const char* rl_version(void);

int rl_create(const rl_config* cfg, rl_world** out);
void rl_destroy(rl_world* h);
int rl_bind_state(rl_world* h, const rl_state* device_ptrs);
/* optional device int32[4] that kernels set on inconsistencies: [0] code, [1] world, [2..3] detail */
int rl_bind_error_flag(rl_world* h, int32_t* device_flag);

/* tuning aid: device int64[32] receiving shader-clock stamps at the phase boundaries of world `world` (NULL = off) */
int rl_bind_phase_profile(rl_world* h, long long* device_stamps, int world);

int rl_reset_synthetic(rl_world* h, int n_agents, float* obs, void* stream);
/* Environment.reset() for every world (environment.py:133-158): one agent per brain -- gene = brain = its index, each at a
 * uniformly random cell (:147-149) -- and _init_food's Binomial food / poison counts + one super food (:741-761), from the
 * Philox streams (epoch as found in the state); obs as in rl_reset_synthetic */
int rl_reset_families(rl_world* h, float* obs, void* stream);
/* worlds with n_agents < threshold are re-generated (epoch+1); refill_count: optional device int32 accumulator */
int rl_refill(rl_world* h, int threshold, int n_agents, float* obs, int32_t* refill_count, void* stream);
int rl_observe(rl_world* h, float* obs, void* stream);
int rl_step(rl_world* h, const int8_t* actions, const rl_tape* tape, const rl_step_out* out, void* stream);
/* Environment.step() in two launches, for callers that draw _add_food's random numbers themselves from the reference's
 * own np.random stream (their count and arguments depend on the post-movement grid, environment.py:763-776):
 *   rl_step_split  everything up to and including _get_rewards; pre_counts [R][4] = food, poison, super food and empty
 *                  cells after movement; the outputs except `obs` are produced here
 *   rl_step_food   _add_food driven by tape->food_k / food_u, then the observation pass into `obs` (state_prime) */
int rl_step_split(rl_world* h, const int8_t* actions, const rl_step_out* out, int32_t* pre_counts, void* stream);
int rl_step_food(rl_world* h, const rl_tape* tape, float* obs, void* stream);
int rl_update(rl_world* h, const rl_tape* tape, const rl_update_out* out, void* stream);
int rl_tick(rl_world* h, const int8_t* actions, const rl_tape* tape, const rl_step_out* sout,
            const rl_update_out* uout, void* stream);
/* rl_tick (Philox draws) followed, in the same launch, by rl_refill(threshold, n_agents) of every world */
int rl_tick_refill(rl_world* h, const int8_t* actions, const rl_step_out* sout, const rl_update_out* uout,
                   int threshold, int n_agents, int32_t* refill_count, void* stream);

/* n_ticks iterations of the inference loop of Helpers/trainer.py:85-99 (minus learn) in ONE launch:
 *     for agent in env.agents: agent.get_action(n_epi)   (= rl_policy_act)      trainer.py:88-89
 *     env.step(); env.update_env(n_epi)                  (= rl_tick_refill)     trainer.py:92,99
 * Every world stays in its workgroup's LDS for the whole launch; the results are those of n_ticks calls of rl_policy_act +
 * rl_tick_refill (same Philox streams), and every per-tick output is written every tick, so after the call the buffers hold
 * the LAST tick's values:
 *   actions     [R][cap]      the actions chosen in the last tick (its pre-step list order)
 *   sout        reward / done / src / obs (state_prime) / n_acted / acted_total and the Tracker accumulators as in rl_tick (n_post / age /
 *               brain, the inputs of rl_capture_transitions, are not written here: rl_run_ex captures inside the launch instead)
 *   obs[2]      Agent.state ping-pong pair: tick i READS obs[(first_obs + i) & 1] (the policy's input; for i = 0 it must hold
 *               the current Agent.state rows) and WRITES the other one; after the call the current rows are in
 *               obs[(first_obs + n_ticks) & 1] and the rows the policy read for the last tick in the other buffer
 *   update_src  [R][cap] or NULL: rl_update_out.src of the last tick
 *   threshold   < 0: no re-generation; else worlds below `threshold` agents are re-generated with n_agents (rl_refill)
 * Supported (rl_run_supported() != 0) in THIS library (libreinlife_hip.so): n_brains <= 8, slot_cap <= 512 (the workgroup: 512 threads,
 * one world per workgroup), brains of any kinds, and n_worlds <= 768 -- or any n_worlds with the "run_always" option set (several
 * workgroups per CU then take turns; the two-launch loop is faster there, which is why it is not the default).
 * Not in this library: the 256- and 1024-thread instantiations ("world_block" = 256 / 1024, dueling kinds only) exist in the tuning
 * build alone (libreinlife_hip_tune.so, RL_TUNE=1 python reinlife_amd/build.py); here they answer 0 / RL_E_UNSUPPORTED.
 * Otherwise RL_E_UNSUPPORTED: loop over rl_policy_act + rl_tick_refill. */
/* rl_run_ex: rl_run with the options a TRAINING loop needs (Helpers/trainer.py:85-99 with training=True):
 *   eps_schedule  device [n_ticks][n_brains] or NULL: the brains' exploration rate in every tick of the launch -- the reference's brains
 *                 change epsilon from episode to episode (D3QN.py:84-89 / PERD3QN.py:82-86: x 0.99 per new n_epi; DQN.py:67-69);
 *                 NULL = brains[b].epsilon throughout
 *   sout->trk_*   the Tracker accumulators of rl_step_out are maintained every tick (environment.py:206-207 -> tracker.py:107-121):
 *                 trk_tick holds the last tick's values, trk_sum / trk_cnt / trk_pop[1..2] are read at the start of the launch and
 *                 written back at its end (running sums over as many launches as the caller likes; it zeroes them at interval ends) */
#define RL_EPS_INLINE_MAX 256
typedef struct {
    int32_t threshold, n_agents;     /* refill rule as in rl_run (threshold < 0: none) */
    int32_t* refill_count;
    const float* eps_schedule;
    int32_t trk_skip_ticks;          /* the first trk_skip_ticks ticks of the launch write trk_tick but stay out of the running sums:
                                      * episode 0 of a training run never reaches an aggregate (tracker.py:279-282 keeps the last
                                      * update_interval entries of update_interval + 1) */
    int32_t eps_schedule_on_host;    /* != 0: eps_schedule is a HOST pointer and n_ticks * n_brains <= RL_EPS_INLINE_MAX: the table travels
                                      * inside the kernel arguments (copied during the call) -- a short launch then waits for no upload */
    const rl_replay* replays;        /* host array [n_brains] or NULL: every tick's transitions are appended to the brains' replay rings
                                      * inside the launch -- what rl_capture_transitions does after a stand-alone tick (trainer.py:95-96,
                                      * entities.py:194-208); the rings' order is Agent.learn call order per world, worlds interleaved */
    float* policy_out;               /* device [R][cap][8] or NULL: the policy's outputs (Q values / PPO probabilities) of the LAST tick,
                                      * indexed like `actions`; with `replays`, rl_replay.prob gets the taken action's entry every tick */
} rl_run_opts;
int rl_run_supported(const rl_world* h, const rl_brain* brains, int n_brains);
int rl_run(rl_world* h, const rl_brain* brains, int n_brains, int n_ticks, int8_t* actions, const rl_step_out* sout,
           float* const obs[2], int first_obs, int16_t* update_src, int threshold, int n_agents, int32_t* refill_count,
           void* stream);
int rl_run_ex(rl_world* h, const rl_brain* brains, int n_brains, int n_ticks, int8_t* actions, const rl_step_out* sout,
              float* const obs[2], int first_obs, int16_t* update_src, const rl_run_opts* opts, void* stream);

/* trainer.py:95-96 for every world: agents of the post-step list with age > 1 append (state, action, reward,
 * state_prime, done[, prob]) to the replay ring of their brain.
 *   state      [R][cap][153] the observation buffer the policy read for this tick (keep it: ping-pong rl_update_out.obs)
 *   actions    [R][cap]      this tick's actions (pre-step list order)
 *   policy_out [R][cap][8]   optional rl_policy_act outputs of this tick
 *   step       outputs of this tick's rl_step / rl_tick; needs reward, done, src, obs, n_post, age, brain
 *   replays    host array of n_brains rings (n_brains <= RL_MAX_CAPTURE_BRAINS) */
int rl_capture_transitions(rl_world* h, const float* state, const int8_t* actions, const float* policy_out,
                           const rl_step_out* step, const rl_replay* replays, int n_brains, void* stream);

/* ---- policy ------------------------------------------------------------------------------------------------- */
/* number of floats in a brain's state dict (flat, registration order) / in its packed MFMA layout */
int64_t rl_policy_n_params(int kind);
int64_t rl_policy_packed_floats(int kind);
/* host -> host: state-dict order  (DQN: fc1.w fc1.b fc2.w fc2.b fc3.w fc3.b;  D3QN/PERD3QN: fc.w fc.b adv_fc1.w
 * adv_fc1.b adv_fc2.w adv_fc2.b value_fc1.w value_fc1.b value_fc2.w value_fc2.b;  PPO: fc1.w fc1.b fc2.w fc2.b
 * fc_pi.w fc_pi.b fc_v.w fc_v.b)  ->  MFMA-fragment-major layout read by the kernels */
int rl_policy_pack_weights(int kind, const float* state_dict_flat, float* packed);
/* dense batch: obs [n_rows][153] (device) -> out [n_rows][8]: Q values (per-row dueling mean) or PPO probabilities */
int rl_policy_forward(int kind, const float* packed, const float* obs, int64_t n_rows, float* out, void* stream);
/* all agents of all worlds: row (w,k) uses brains[a_brain[w][k]].
 *   obs     [R][cap][153]   actions [R][cap] (written for k < n_agents[w])   out_q [R][cap][8] or NULL
 *   work    device scratch of rl_policy_work_bytes(h) bytes, zero-initialised ONCE by the caller before the first
 *           call and owned by this handle afterwards (it carries double-buffered counters between calls)
 *   tape_actions: optional [R][cap] recorded actions (parity mode) that override the selected ones */
size_t rl_policy_work_bytes(const rl_world* h);
/* optional: hand the work buffer to the handle so that rl_tick / rl_update / rl_reset_synthetic / rl_refill also emit
 * the per-brain row lists rl_policy_act needs (saves its bucket launch).  All launches of a handle must be issued in
 * stream order.  rl_bind_state (call it again after rewriting state buffers yourself) invalidates the lists. */
int rl_bind_policy_work(rl_world* h, void* work);
int rl_policy_act(rl_world* h, const rl_brain* brains, int n_brains, const float* obs, int8_t* actions, float* out_q,
                  void* work, void* stream);

/* ---- options -------------------------------------------------------------------------------------------------- */
/* Tuning / test switches (nothing like them in the reference).  Process-level values start from the environment, read ONCE
 * (RL_WORLD_BLOCK, RL_WORLD_GENERIC, RL_POLICY_VARIANT, RL_RUN_ALWAYS), and change only through rl_set_option;
 * rl_create copies them into the handle, so a handle's kernels never change under it and no launch reads the environment.
 *   "world_block"      "0" (by world count) | "256" | "512" | "1024": workgroup size of the world kernels and of rl_run
 *   "world_generic"    "1": the generic world code also for the default 30x30 / 100-agent shape
 *   "policy_variant"   "auto" (default: the tiles of rl_run's policy half -- ONE arithmetic on every path: `pair`, or `dense` for
 *                      dueling brains from 1,536 tiles on, bit-identical to each other) | "pair" | "dense" | "wave" (one wave per
 *                      tile, dueling kinds; the same bits).  Every variant computes the same Q values bit for bit.
 *   "run_always"       "1": callers that choose between rl_run and the two-launch loop by world count take rl_run
 * value NULL restores what the environment says.  rl_get_option: the handle's snapshot (h != NULL) or the process level (h == NULL);
 * policy_variant as 0 auto, 2 wave, 3 dense, 4 pair; -1 for an unknown name. */
int rl_set_option(const char* name, const char* value);
int rl_get_option(const rl_world* h, const char* name);

/* Philox4x32-10 exactly as the kernels use it (host helper for tests / tools) */
void rl_philox(uint64_t seed, uint32_t epoch, uint32_t world, uint32_t tick, uint32_t site, uint32_t index,
               uint32_t out[4]);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif
