// rl_capi.hip -- extern "C" surface of libreinlife_hip.so (see include/reinlife_hip.h for the contract).
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <new>

#include "rl_common.h"

// implemented in rl_world.hip / rl_policy.hip
size_t rl_world_smem_bytes(int cpad, int cap, int hash, int plane_stride, int height);
int rl_world_block();
int rl_world_launch_step(rl_world*, const int8_t*, const rl_tape*, const rl_step_out*, hipStream_t);
int rl_world_launch_step_split(rl_world*, const int8_t*, const rl_step_out*, int32_t*, hipStream_t);
int rl_world_launch_step_food(rl_world*, const rl_tape*, float*, hipStream_t);
int rl_world_launch_update(rl_world*, const rl_tape*, const rl_update_out*, hipStream_t);
int rl_world_launch_tick(rl_world*, const int8_t*, const rl_tape*, const rl_step_out*, const rl_update_out*, int, int, int32_t*, hipStream_t);
int rl_world_launch_observe(const rl_world*, float*, hipStream_t);
int rl_world_launch_reset(rl_world*, int, int, float*, int32_t*, int, hipStream_t);
int rl_world_launch_capture(rl_world*, const float*, const int8_t*, const float*, const rl_step_out*, const rl_replay*, int, hipStream_t);
int rl_world_run_supported(const rl_world*, const rl_brain*, int);
int rl_world_launch_run(rl_world*, const rl_brain*, int, int, int8_t*, const rl_step_out*, float* const*, int, int16_t*, int, int, int32_t*, const float*, int, int, const rl_replay*, float*, hipStream_t);
int64_t rl_policy_n_params_impl(int);
int64_t rl_policy_packed_floats_impl(int);
int rl_policy_pack_impl(int, const float*, float*);
int rl_policy_forward_impl(int, const float*, const float*, int64_t, float*, hipStream_t);
size_t rl_policy_work_bytes_impl(const rl_world*);
int rl_policy_act_impl(rl_world*, const rl_brain*, int, const float*, int8_t*, float*, void*, hipStream_t);

static thread_local char g_err[512] = "";

// ---- options: environment read once, rl_set_option afterwards -------------------------------------------------------------------
#include <stdlib.h>
static int parse_variant(const char* v)
{
    if (!v || !*v || !strcmp(v, "auto")) return RL_PV_AUTO;
    if (!strcmp(v, "wave")) return RL_PV_WAVE;
    if (!strcmp(v, "dense")) return RL_PV_DENSE;
    if (!strcmp(v, "pair")) return RL_PV_PAIR;
    return -1;
}
static int parse_block(const char* v)
{
    const int b = (v && *v) ? atoi(v) : 0;
    return (b == 0 || b == 256 || b == 512 || b == 1024) ? b : -1;
}
static rl_options options_from_env()
{
    rl_options o{};
    const int b = parse_block(getenv("RL_WORLD_BLOCK"));
    o.world_block = b < 0 ? 0 : b;
    o.world_generic = getenv("RL_WORLD_GENERIC") ? 1 : 0;
    const int v = parse_variant(getenv("RL_POLICY_VARIANT"));
    o.policy_variant = v < 0 ? RL_PV_AUTO : v;
    o.run_always = getenv("RL_RUN_ALWAYS") ? 1 : 0;
    return o;
}
static rl_options& options_mut()
{
    static rl_options o = options_from_env();   // (thread-safe one-time initialisation)
    return o;
}
const rl_options& rl_options_current() { return options_mut(); }

// The device a handle's buffers live on (taken from the state pointers in rl_bind_state).  Every launching entry point
// makes it current for the duration of the call, so a caller whose current device is another GPU (several handles on
// several GPUs in one process) still launches in the right context.
static int device_of_pointer(const void* p)
{
    hipPointerAttribute_t a;
    if (!p || hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return -1; }
    return a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged ? a.device : -1;
}
struct DeviceGuard {
    int prev = -1;
    explicit DeviceGuard(int dev)
    {
        int cur = -1;
        if (dev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != dev && hipSetDevice(dev) == hipSuccess) prev = cur;
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

void rl_set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" {

int rl_set_option(const char* name, const char* value)
{
    if (!name) { rl_set_error("rl_set_option: null name"); return RL_E_INVALID; }
    rl_options& o = options_mut();
    const rl_options env = options_from_env();   // value == NULL: back to what the environment says
    if (!strcmp(name, "world_block")) {
        const int b = value ? parse_block(value) : env.world_block;
        if (b < 0) { rl_set_error("rl_set_option: world_block must be 0 (auto), 256, 512 or 1024"); return RL_E_INVALID; }
        o.world_block = b;
    } else if (!strcmp(name, "world_generic")) o.world_generic = value ? (atoi(value) != 0) : env.world_generic;
    else if (!strcmp(name, "policy_variant")) {
        const int v = value ? parse_variant(value) : env.policy_variant;
        if (v < 0) { rl_set_error("rl_set_option: policy_variant must be auto, pair, dense or wave"); return RL_E_INVALID; }
        o.policy_variant = v;
    } else if (!strcmp(name, "run_always")) o.run_always = value ? (atoi(value) != 0) : env.run_always;
    else { rl_set_error("rl_set_option: unknown option '%s'", name); return RL_E_INVALID; }
    return RL_OK;
}

int rl_get_option(const rl_world* h, const char* name)
{
    const rl_options& o = h ? h->opt : rl_options_current();
    if (!name) return -1;
    if (!strcmp(name, "world_block")) return o.world_block;
    if (!strcmp(name, "world_generic")) return o.world_generic;
    if (!strcmp(name, "policy_variant")) return o.policy_variant;
    if (!strcmp(name, "run_always")) return o.run_always;
    return -1;
}

const char* rl_last_error(void) { return g_err; }
const char* rl_version(void) { return "reinlife_hip 0.2 (gfx950)"; }

int rl_create(const rl_config* cfg, rl_world** out)
{
    if (!cfg || !out) { rl_set_error("rl_create: null argument"); return RL_E_INVALID; }
    *out = nullptr;
    if (cfg->width < 3 || cfg->height < 3 || cfg->width > 255 || cfg->height > 255) {  // Grid asserts >= 3 (grid.py:23-24)
        rl_set_error("rl_create: width/height must be in [3,255] (got %dx%d): the reference's Grid needs >= 3 (grid.py:23-24) and has no upper bound; "
                     "this library keeps a world in ONE workgroup's LDS with 8-bit coordinates", cfg->width, cfg->height);
        return RL_E_INVALID;
    }
    const int cells = cfg->width * cfg->height;
    if (cells > RL_MAX_CELLS) {
        rl_set_error("rl_create: %dx%d = %d cells, this library supports width*height <= %d (a world lives in one workgroup's 160 KB of LDS); "
                     "the reference's Grid(width, height) is unbounded (grid.py:22-33) -- a LIMIT of this build, not a rule of the reference",
                     cfg->width, cfg->height, cells, RL_MAX_CELLS);
        return RL_E_UNSUPPORTED;
    }
    if (cfg->n_brains < 1 || cfg->n_brains > RL_MAX_BRAINS) { rl_set_error("rl_create: n_brains must be in [1,%d]", RL_MAX_BRAINS); return RL_E_INVALID; }
    if (cfg->n_worlds < 1 || cfg->n_worlds >= (1 << 19)) { rl_set_error("rl_create: n_worlds must be in [1, 2^19)"); return RL_E_INVALID; }
    if (cfg->max_agents < 1) { rl_set_error("rl_create: max_agents must be >= 1"); return RL_E_INVALID; }
    if (cfg->slot_cap < 64 || (cfg->slot_cap & 63) || cfg->slot_cap > 4096 || cfg->slot_cap < 2 * cfg->max_agents + 2) {
        rl_set_error("rl_create: slot_cap must be a multiple of 64 in [max(64, 2*max_agents+2), 4096] (got %d)", cfg->slot_cap);
        return RL_E_INVALID;
    }
    rl_world* h = new (std::nothrow) rl_world();
    if (!h) { rl_set_error("rl_create: out of memory"); return RL_E_INVALID; }
    h->cfg = *cfg;
    h->opt = rl_options_current();   // the handle's kernels are chosen from this snapshot: nothing reads the environment at a launch
    h->cells = cells;
    h->cpad = (cells + 63) & ~63;
    int hs = 64;
    while (hs < 2 * cfg->slot_cap) hs <<= 1;
    h->hash_size = hs;
    h->smem_bytes = rl_world_smem_bytes(h->cpad, cfg->slot_cap, hs, cfg->width, cfg->height);
    h->block = rl_world_block();
    if (h->smem_bytes > 160 * 1024) {
        rl_set_error("rl_create: world needs %zu bytes of LDS (> 160 KB)", h->smem_bytes);
        delete h;
        return RL_E_UNSUPPORTED;
    }
    *out = h;  // no device call here: handles can be created (and configs validated) without a GPU
    return RL_OK;
}

void rl_destroy(rl_world* h) { delete h; }

int rl_bind_state(rl_world* h, const rl_state* s)
{
    if (!h || !s) { rl_set_error("rl_bind_state: null argument"); return RL_E_INVALID; }
    const void* const* p = (const void* const*)s;
    for (size_t i = 0; i < sizeof(rl_state) / sizeof(void*); ++i)
        if (!p[i]) { rl_set_error("rl_bind_state: state pointer #%zu is null", i); return RL_E_INVALID; }
    h->st = *s;
    h->device = device_of_pointer(s->cell_type);  // -1 (no guard) when it cannot be determined
    h->bound = 1;
    h->lists_valid = 0;  // the caller may have rewritten the state
    return RL_OK;
}

int rl_bind_error_flag(rl_world* h, int32_t* flag)
{
    if (!h) { rl_set_error("rl_bind_error_flag: null handle"); return RL_E_INVALID; }
    h->err_flag = flag;
    return RL_OK;
}

int rl_bind_phase_profile(rl_world* h, long long* device_stamps, int world)
{
    if (!h) { rl_set_error("rl_bind_phase_profile: null handle"); return RL_E_INVALID; }
    h->prof = device_stamps; h->prof_world = world;
    return RL_OK;
}

int rl_bind_policy_work(rl_world* h, void* work)
{
    if (!h) { rl_set_error("rl_bind_policy_work: null handle"); return RL_E_INVALID; }
    h->work = work; h->lists_valid = 0;
    return RL_OK;
}

#define RL_CHECK_BOUND(fn)                                                              \
    if (!h) { rl_set_error(fn ": null handle"); return RL_E_INVALID; }                 \
    if (!h->bound) { rl_set_error(fn ": rl_bind_state was not called"); return RL_E_UNBOUND; }  \
    DeviceGuard rl_guard_(h->device);

int rl_reset_synthetic(rl_world* h, int n_agents, float* obs, void* stream)
{
    RL_CHECK_BOUND("rl_reset_synthetic")
    if (n_agents < 0 || n_agents > h->cfg.slot_cap || n_agents > h->cells) { rl_set_error("rl_reset_synthetic: bad n_agents %d", n_agents); return RL_E_INVALID; }
    return rl_world_launch_reset(h, n_agents, -1, obs, nullptr, 0, (hipStream_t)stream);
}

int rl_reset_families(rl_world* h, float* obs, void* stream)
{
    RL_CHECK_BOUND("rl_reset_families")
    if (h->cfg.n_brains > h->cells) { rl_set_error("rl_reset_families: %d brains on %d cells", h->cfg.n_brains, h->cells); return RL_E_INVALID; }
    return rl_world_launch_reset(h, h->cfg.n_brains, -1, obs, nullptr, 1, (hipStream_t)stream);
}

int rl_refill(rl_world* h, int threshold, int n_agents, float* obs, int32_t* refill_count, void* stream)
{
    RL_CHECK_BOUND("rl_refill")
    if (threshold < 0 || n_agents < 0 || n_agents > h->cfg.slot_cap || n_agents > h->cells) { rl_set_error("rl_refill: bad arguments"); return RL_E_INVALID; }
    return rl_world_launch_reset(h, n_agents, threshold, obs, refill_count, 0, (hipStream_t)stream);
}

int rl_observe(rl_world* h, float* obs, void* stream)
{
    RL_CHECK_BOUND("rl_observe")
    if (!obs) { rl_set_error("rl_observe: null obs"); return RL_E_INVALID; }
    return rl_world_launch_observe(h, obs, (hipStream_t)stream);
}

static int check_tape(const rl_tape* t, const char* fn)
{
    if (!t || !t->food_k) return RL_OK;
    if (!t->food_u || !t->repro_u || !t->birth_k || !t->produce_u || !t->produce_choice) {
        rl_set_error("%s: a tape must provide all six arrays", fn);
        return RL_E_INVALID;
    }
    return RL_OK;
}

static int check_step_out(const rl_step_out* o, const char* fn)
{
    if (!o) return RL_OK;
    const int n = (o->trk_tick != nullptr) + (o->trk_sum != nullptr) + (o->trk_cnt != nullptr) + (o->trk_pop != nullptr);
    if (n != 0 && n != 4) { rl_set_error("%s: the four tracker pointers must be given together", fn); return RL_E_INVALID; }
    return RL_OK;
}

int rl_step(rl_world* h, const int8_t* actions, const rl_tape* tape, const rl_step_out* out, void* stream)
{
    RL_CHECK_BOUND("rl_step")
    if (int rc = check_step_out(out, "rl_step")) return rc;
    if (!actions) { rl_set_error("rl_step: null actions"); return RL_E_INVALID; }
    if (int rc = check_tape(tape, "rl_step")) return rc;
    return rl_world_launch_step(h, actions, tape, out, (hipStream_t)stream);
}

int rl_step_split(rl_world* h, const int8_t* actions, const rl_step_out* out, int32_t* pre_counts, void* stream)
{
    RL_CHECK_BOUND("rl_step_split")
    if (!actions || !pre_counts) { rl_set_error("rl_step_split: null argument"); return RL_E_INVALID; }
    if (int rc = check_step_out(out, "rl_step_split")) return rc;
    return rl_world_launch_step_split(h, actions, out, pre_counts, (hipStream_t)stream);
}

int rl_step_food(rl_world* h, const rl_tape* tape, float* obs, void* stream)
{
    RL_CHECK_BOUND("rl_step_food")
    if (!tape || !tape->food_k || !tape->food_u) { rl_set_error("rl_step_food: a tape with food_k / food_u is required"); return RL_E_INVALID; }
    return rl_world_launch_step_food(h, tape, obs, (hipStream_t)stream);
}

int rl_update(rl_world* h, const rl_tape* tape, const rl_update_out* out, void* stream)
{
    RL_CHECK_BOUND("rl_update")
    if (int rc = check_tape(tape, "rl_update")) return rc;
    return rl_world_launch_update(h, tape, out, (hipStream_t)stream);
}

int rl_tick(rl_world* h, const int8_t* actions, const rl_tape* tape, const rl_step_out* sout, const rl_update_out* uout, void* stream)
{
    RL_CHECK_BOUND("rl_tick")
    if (!actions) { rl_set_error("rl_tick: null actions"); return RL_E_INVALID; }
    if (int rc = check_tape(tape, "rl_tick")) return rc;
    if (int rc = check_step_out(sout, "rl_tick")) return rc;
    return rl_world_launch_tick(h, actions, tape, sout, uout, -1, 0, nullptr, (hipStream_t)stream);
}

int rl_tick_refill(rl_world* h, const int8_t* actions, const rl_step_out* sout, const rl_update_out* uout, int threshold,
                   int n_agents, int32_t* refill_count, void* stream)
{
    RL_CHECK_BOUND("rl_tick_refill")
    if (!actions) { rl_set_error("rl_tick_refill: null actions"); return RL_E_INVALID; }
    if (threshold < 0 || n_agents < 0 || n_agents > h->cfg.slot_cap || n_agents > h->cells) { rl_set_error("rl_tick_refill: bad arguments"); return RL_E_INVALID; }
    if (int rc = check_step_out(sout, "rl_tick_refill")) return rc;
    return rl_world_launch_tick(h, actions, nullptr, sout, uout, threshold, n_agents, refill_count, (hipStream_t)stream);
}

int rl_run_supported(const rl_world* h, const rl_brain* brains, int n_brains)
{
    if (!h || !brains) return 0;
    return rl_world_run_supported(h, brains, n_brains);
}

int rl_run_ex(rl_world* h, const rl_brain* brains, int n_brains, int n_ticks, int8_t* actions, const rl_step_out* sout,
              float* const obs[2], int first_obs, int16_t* update_src, const rl_run_opts* opts, void* stream)
{
    RL_CHECK_BOUND("rl_run")
    const rl_run_opts none = {-1, 0, nullptr, nullptr, 0, 0, nullptr, nullptr};
    if (!opts) opts = &none;
    if (!brains || !actions || !obs || !obs[0] || !obs[1]) { rl_set_error("rl_run: null argument"); return RL_E_INVALID; }
    if (n_brains != h->cfg.n_brains) { rl_set_error("rl_run: n_brains %d != config %d", n_brains, h->cfg.n_brains); return RL_E_INVALID; }
    if (n_ticks < 0 || (first_obs != 0 && first_obs != 1)) { rl_set_error("rl_run: bad n_ticks / first_obs"); return RL_E_INVALID; }
    if (opts->threshold >= 0 && (opts->n_agents < 0 || opts->n_agents > h->cfg.slot_cap || opts->n_agents > h->cells)) { rl_set_error("rl_run: bad refill arguments"); return RL_E_INVALID; }
    if (int rc = check_step_out(sout, "rl_run")) return rc;
    if (opts->replays) {
        for (int i = 0; i < n_brains; ++i) {
            const rl_replay& r = opts->replays[i];
            if (!r.state || !r.state_prime || !r.action || !r.reward || !r.done || !r.age || !r.count || r.capacity < 1) { rl_set_error("rl_run: replay %d incomplete", i); return RL_E_INVALID; }
        }
    }
    for (int b = 0; b < n_brains; ++b)
        if (brains[b].kind < RL_DQN || brains[b].kind > RL_PPO || !brains[b].packed) { rl_set_error("rl_run: brain %d invalid", b); return RL_E_INVALID; }
    if (n_ticks == 0) return RL_OK;
    if (opts->eps_schedule_on_host && (!opts->eps_schedule || (int64_t)n_ticks * n_brains > RL_EPS_INLINE_MAX)) {
        rl_set_error("rl_run: eps_schedule_on_host needs a table of at most %d floats (got %lld)", RL_EPS_INLINE_MAX, (long long)n_ticks * n_brains);
        return RL_E_INVALID;
    }
    return rl_world_launch_run(h, brains, n_brains, n_ticks, actions, sout, obs, first_obs, update_src, opts->threshold, opts->n_agents,
                               opts->refill_count, opts->eps_schedule, opts->eps_schedule_on_host, opts->trk_skip_ticks, opts->replays, opts->policy_out, (hipStream_t)stream);
}

int rl_run(rl_world* h, const rl_brain* brains, int n_brains, int n_ticks, int8_t* actions, const rl_step_out* sout,
           float* const obs[2], int first_obs, int16_t* update_src, int threshold, int n_agents, int32_t* refill_count, void* stream)
{
    const rl_run_opts o = {threshold, n_agents, refill_count, nullptr, 0, 0, nullptr, nullptr};
    return rl_run_ex(h, brains, n_brains, n_ticks, actions, sout, obs, first_obs, update_src, &o, stream);
}

int rl_capture_transitions(rl_world* h, const float* state, const int8_t* actions, const float* policy_out, const rl_step_out* step,
                           const rl_replay* replays, int n_brains, void* stream)
{
    RL_CHECK_BOUND("rl_capture_transitions")
    if (!state || !actions || !step || !replays) { rl_set_error("rl_capture_transitions: null argument"); return RL_E_INVALID; }
    if (n_brains != h->cfg.n_brains || n_brains > RL_MAX_CAPTURE_BRAINS) { rl_set_error("rl_capture_transitions: n_brains %d (config %d, max %d)", n_brains, h->cfg.n_brains, RL_MAX_CAPTURE_BRAINS); return RL_E_INVALID; }
    if (!step->reward || !step->done || !step->src || !step->obs || !step->n_post || !step->age || !step->brain) { rl_set_error("rl_capture_transitions: step outputs reward/done/src/obs/n_post/age/brain are required"); return RL_E_INVALID; }
    if (h->cfg.slot_cap > 4096) { rl_set_error("rl_capture_transitions: slot_cap > 4096"); return RL_E_UNSUPPORTED; }
    for (int i = 0; i < n_brains; ++i)
        if (!replays[i].state || !replays[i].state_prime || !replays[i].action || !replays[i].reward || !replays[i].done || !replays[i].age || !replays[i].count || replays[i].capacity < 1) { rl_set_error("rl_capture_transitions: replay %d incomplete", i); return RL_E_INVALID; }
    return rl_world_launch_capture(h, state, actions, policy_out, step, replays, n_brains, (hipStream_t)stream);
}

int64_t rl_policy_n_params(int kind) { return rl_policy_n_params_impl(kind); }
int64_t rl_policy_packed_floats(int kind) { return rl_policy_packed_floats_impl(kind); }

int rl_policy_pack_weights(int kind, const float* sd, float* packed)
{
    if (!sd || !packed) { rl_set_error("rl_policy_pack_weights: null argument"); return RL_E_INVALID; }
    return rl_policy_pack_impl(kind, sd, packed);
}

int rl_policy_forward(int kind, const float* packed, const float* obs, int64_t n_rows, float* out, void* stream)
{
    if (!packed || !obs || !out || n_rows < 0) { rl_set_error("rl_policy_forward: bad argument"); return RL_E_INVALID; }
    if (kind < RL_DQN || kind > RL_PPO) { rl_set_error("rl_policy_forward: unknown brain kind %d", kind); return RL_E_INVALID; }
    if (n_rows == 0) return RL_OK;
    DeviceGuard guard(device_of_pointer(obs));
    return rl_policy_forward_impl(kind, packed, obs, n_rows, out, (hipStream_t)stream);
}

size_t rl_policy_work_bytes(const rl_world* h) { return h ? rl_policy_work_bytes_impl(h) : 0; }

int rl_policy_act(rl_world* h, const rl_brain* brains, int n_brains, const float* obs, int8_t* actions, float* out_q, void* work, void* stream)
{
    RL_CHECK_BOUND("rl_policy_act")
    if (!brains || !obs || !actions || !work) { rl_set_error("rl_policy_act: null argument"); return RL_E_INVALID; }
    if (n_brains != h->cfg.n_brains) { rl_set_error("rl_policy_act: n_brains %d != config %d", n_brains, h->cfg.n_brains); return RL_E_INVALID; }
    for (int b = 0; b < n_brains; ++b)
        if (brains[b].kind < RL_DQN || brains[b].kind > RL_PPO || !brains[b].packed) { rl_set_error("rl_policy_act: brain %d invalid", b); return RL_E_INVALID; }
    return rl_policy_act_impl(h, brains, n_brains, obs, actions, out_q, work, (hipStream_t)stream);
}

void rl_philox(uint64_t seed, uint32_t epoch, uint32_t world, uint32_t tick, uint32_t site, uint32_t index, uint32_t out[4])
{
    const rl_u4 r = rl_philox4x32(seed, epoch, world, tick, site, index);
    out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w;
}

}  // extern "C"
