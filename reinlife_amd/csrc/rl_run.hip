// rl_run.hip -- the multi-tick kernel of the ReinLife hot path on MI355X (gfx950): k_run and its host launcher.
#include <mutex>
#include "rl_world_dev.h"
// What the input layer may assume about the Agent.state rows (XM of the tiles, rl_policy_dev.h in_chunk_class): measured per kernel family
// (tools/xf_ab.sh).  The dueling-only kernel is SLOWER with it (21.8 -> 22.3 - 22.5 us per tick at configs[3]: its tiles are
// bound by dependent latency, not by the matrix pipe, and fewer MFMAs leave the splits and the weight fetches nothing to hide behind), the
// mixed-kind kernel faster (configs[4]: 26.7 -> 25.7 us): there a PPO tile shares its SIMD with a lighter one and every MFMA not issued is room.
// RL_XM_DUELING_KERNEL != 0 is an A/B switch only (tools/xf_ab.sh): run_policy1 hands it to the tile WITHOUT the per-world oflags gate
// of run_policy_all, i.e. it assumes row scale 2^10 and zero lo halves unconditionally -- true for states the world's own rules produce,
// WRONG for a caller-loaded state with |health| >= 400.  Such a build must say that it knows.
#ifndef RL_XM_DUELING_KERNEL
#define RL_XM_DUELING_KERNEL 0
#endif
#if RL_XM_DUELING_KERNEL != 0 && !defined(RL_XM_SELF_GENERATED_STATES_ONLY)
#error "RL_XM_DUELING_KERNEL != 0 is valid for self-generated states only: add -DRL_XM_SELF_GENERATED_STATES_ONLY to an A/B build that knows"
#endif
#ifndef RL_XM_ALL_DUELING_TILE   /* the dueling tile INSIDE the mixed-kind kernel (A/B: RL_XM_ALL_KERNEL, or 0 = one copy of it) */
#define RL_XM_ALL_DUELING_TILE RL_XM_ALL_KERNEL
#endif
#ifndef RL_XM_ALL_KERNEL      /* 0: never; 2: tiles of worlds whose rows are known to be scaled by 2^10 with an integer health plane */
#define RL_XM_ALL_KERNEL 2
#endif

// Measurement switch of the multi-tick launch (bench.py's per-half timings, tools/): 1 = skip the policy half, 2 = skip the tick half,
// 4 / 8 / 16 = at most 2 / 1 / 3 policy tiles, 32 = no staggered start.  Results are then WRONG, so it exists in the TUNING builds only
// (-DRL_TUNING: lib/libreinlife_hip_tune.so, and the stamped _prof build): the product library neither exports the switch nor compiles
// the branches (RL_RUN_DBG is the constant 0 there).
#ifdef RL_TUNING
#if !(defined(RL_RUN_UNIT) && RL_RUN_UNIT == 1)   /* (the host side lives in unit 0) */
static int g_run_debug = 0;
extern "C" __attribute__((visibility("default"))) void rl_debug_set_run_mask(int mask) { g_run_debug = mask; }
extern "C" __attribute__((visibility("default"))) int rl_debug_get_run_mask(void) { return g_run_debug; }
#endif
#define RL_RUN_DBG(ka) (*(const int __attribute__((address_space(4)))*)&(ka)->ra.debug)
#else
#define RL_RUN_DBG(ka) 0
#endif

namespace {

// ---------------------------------------------------------------------------------------------------------------
// k_run: n_ticks iterations of the inference loop (Helpers/trainer.py:85-99 minus learn) in ONE launch.
//
//   for every tick:  Agent.get_action for the world's agents (policy_tile1s<PAIR> / policy_pair2, rl_policy_dev.h)  ->  Environment.step  ->
//                    update_env  ->  optional re-generation below the refill threshold
//
// A world never leaves its workgroup: the state is loaded once, every tick runs out of LDS (the same phase functions as
// k_world), the agent list is re-packed in LDS between ticks (recycle_world = what store_world + load_world do through
// HBM), and the state is stored once at the end.  Every API-visible per-tick output is still written every tick
// (state_prime / state rows, reward, done, both permutations, actions), so a tick moves the same algorithmic bytes as
// rl_policy_act + rl_tick_refill; what disappears is two launches, the world's load / store, the row lists, and the trip of
// the observation rows through the fabric to another CU (the policy reads its own world's rows back from L2, sc1).
// The policy tiles run on the waves of the world's workgroup -- two waves per 32-row tile in the product's 512-thread kernels (run_policy1 /
// run_policy_all; one or four in the tuning-only 256- / 1024-thread instantiations) -- and every wave executes the same number of
// workgroup barriers.
// ---------------------------------------------------------------------------------------------------------------
#include "rl_policy_dev.h"

constexpr int kRunMaxBrains = 8;
// k_run's KIND parameter: RL_PERD3QN = every brain is of a dueling kind (D3QN / PERD3QN: the kernel bench.py times, nothing but their tile
// in it); kKindAll = any mix of DQN / D3QN / PERD3QN / PPO brains, the tile code picked per tile at run time (512-thread workgroups).
constexpr int kKindAll = 4;
__host__ __device__ constexpr int kind_widest(int KIND) { return KIND == kKindAll ? RL_PPO : KIND; }

#ifndef RL_RUN_COHERENT
#define RL_RUN_COHERENT true
#endif
struct RunArgs {
    const float* packed[kRunMaxBrains];   // device: packed weights per brains-list entry
    float eps[kRunMaxBrains];
    int kind[kRunMaxBrains];              // RL_DQN .. RL_PPO per brains-list entry (kKindAll kernels)
    float* obs[2];                        // Agent.state ping-pong: tick i reads obs[(first + i) & 1], writes the other
    int first;
    int n_ticks;
    int8_t* actions;                      // [R][cap] chosen actions of the LAST tick (API-visible)
    const float* eps_sched;               // optional device [n_ticks][n_brains]: the brains' exploration rates tick by tick (else eps[])
    int eps_inline_on;                    // the schedule is eps_inline[] (a short launch: no upload to wait for), not eps_sched
    float eps_inline[RL_EPS_INLINE_MAX];
    int trk_skip;                         // the Tracker's running sums leave out the first trk_skip ticks of the launch
    int capture;                          // TRAIN launches: append every tick's transitions to the brains' replay rings (rp[])
    float* policy_out;                    // [R][cap][8] or null: the policy's outputs of the tick (PPO: probabilities) for rl_replay.prob
    rl_replay rp[kRunMaxBrains];
    int debug;                            // measurement only (rl_debug_set_run_mask): 1 = skip the policy half, 2 = skip the tick half (results WRONG)
};

struct PolSmem {
    short* prow;        // [cap] per list entry: brain << 10 | position in the brain's list
    int* bstart;        // [64] first entry of brain b in prow
    int* bcnt;          // [64]
    int* tstart;        // [64] first tile of brain b
    short* trow;        // [kMaxTiles][32] list index of tile row j (0x8000: padding -- reads list entry 0, not valid)
    int* tbrain;        // [kMaxTiles] brain of tile t
    int* meta;          // [16] [0] number of tiles; loop state of k_run: [1] list length, [2] Agent.state parity, [3] ticks done,
                        //      [4] the LDS mirror holds the current Agent.state rows, [5] / [6] the schedule of kKindAll (below),
                        //      [8..15] spare
    float* pairv;       // [4 tiles][32 row values | 2 x 64 partial row maxima] of the two-waves-per-tile policy (T = 512), or null
    float* cconst;      // [n_brains][3][256] epilogue constants of the brains' three 128-wide layers for policy_tile1s (T = 512), or null
    float* xmirror;     // [xrows][kXStride] Agent.state rows of this world for the one-wave policy tile (T <= 512), or null
    int xrows;
    int xbytes;         // bytes reserved for the mirror: xrows rows, or more where the tiles' exchange slices (which alias it) need more
    double* trk_scr;    // [cap] scratch of the Tracker pass (track_world_wave0)
    TrkLds trk;         // the Tracker's running sums for the length of the launch
    // kKindAll kernels: the policy half's schedule, written by wave 0 next to the row lists (policy_schedule_wave0).  meta[5]: 0 = every
    // tile gets a wave pair in ONE round (<= 4 tiles whose exchange buffers fit into the mirror), 1 = rounds of meta[6] tiles
    int* wtask;         // [8]  per wave: tile | role << 8 | brains-list index << 12 | kind << 20, or -1
    int* texoff;        // [4]  per tile: byte offset of its exchange buffer inside the mirror
    int* bkind;         // [8]  the brains' kinds (filled once per launch)
    int* oflags;        // [4]  [0] what is known about the CURRENT Agent.state rows (RL_XF_*, rl_policy_dev.h): the input layer skips the row-maximum
                        //      pass and the exactly-zero partial products; [1] sticky for the launch: the loaded state held an agent with
                        //      |health| >= 400 (never produced by the world's own rules: entities.py:145-159, environment.py:701-715)
};
// With mirror_budget > 0 the Agent.state rows are mirrored in LDS (as many rows as fit below the budget, at most cap); the two- and
// four-wave tiles' exchange buffers alias the mirror.
constexpr int kMaxTiles = 32;                // 32-row tiles of one brain per world: <= cap / 32 + n_brains
constexpr int kPairFloats = 32 + 2 * 64;     // per tile pair: row values, partial row maxima of the two roles
constexpr int kPairExBytes = 8 * kPlanes * 64 * 16;   // per tile pair: the split activations of the input layer (aliases the Agent.state mirror)
constexpr int kPairFloatsAll = kPairValFloats + 2 * 64;   // kKindAll: role 1's head partials (4 per lane), then the partial row maxima
template <int KIND> __host__ __device__ constexpr int run_const_floats() { return KIND == kKindAll ? kTileConstMax : kTileConstFloats; }
template <int KIND> __host__ __device__ constexpr int run_pair_floats() { return KIND == kKindAll ? kPairFloatsAll : kPairFloats; }
template <int KIND>
__host__ __device__ inline size_t carve_policy(PolSmem& ps, char* base, size_t o, int cap, size_t mirror_budget = 0, int n_cbrains = 0,
                                               int pair_floats = run_pair_floats<KIND>(),   // per tile: the tile waves' small exchanges (kQuadFloats for the four-wave tile)
                                               size_t min_region = 0)                       // the four-wave tiles' exchange slices: the mirror's region is at least this large
{
    o = align16(o);
    ps.xmirror = nullptr; ps.xrows = 0; ps.xbytes = 0;
    // what follows the mirror: row lists and tile descriptors (2 * cap + ~3.5 KB), the Tracker's scratch and sums (8 * cap + ~0.8 KB), ...
    const size_t tail = 10 * (size_t)cap + 5120 + sizeof(float) * 4 * (size_t)pair_floats + sizeof(float) * run_const_floats<KIND>() * (size_t)n_cbrains;
    if (mirror_budget > o + tail) {
        const size_t rows = (mirror_budget - o - tail) / (sizeof(float) * kXStride);
        ps.xrows = (int)(rows < (size_t)cap ? rows : (size_t)cap);
        if (ps.xrows >= 32) {
            size_t bytes = sizeof(float) * kXStride * (size_t)ps.xrows;
            if (bytes < min_region && o + tail + min_region <= mirror_budget) bytes = min_region;
            ps.xmirror = (float*)(base + o); ps.xbytes = (int)bytes; o = align16(o + bytes);
        } else ps.xrows = 0;
    }
    ps.prow = (short*)(base + o); o = align16(o + sizeof(short) * (size_t)cap);
    ps.bstart = (int*)(base + o); o = align16(o + sizeof(int) * 64);
    ps.bcnt = (int*)(base + o); o = align16(o + sizeof(int) * 64);
    ps.tstart = (int*)(base + o); o = align16(o + sizeof(int) * 64);
    ps.trow = (short*)(base + o); o = align16(o + sizeof(short) * kMaxTiles * 32);
    ps.tbrain = (int*)(base + o); o = align16(o + sizeof(int) * kMaxTiles);
    ps.meta = (int*)(base + o); o = align16(o + sizeof(int) * 16);
    ps.cconst = nullptr;
    if (n_cbrains > 0) { ps.cconst = (float*)(base + o); o = align16(o + sizeof(float) * run_const_floats<KIND>() * (size_t)n_cbrains); }
    ps.pairv = nullptr;
    if (n_cbrains > 0) { ps.pairv = (float*)(base + o); o = align16(o + sizeof(float) * 4 * (size_t)pair_floats); }
    ps.wtask = (int*)(base + o); o = align16(o + sizeof(int) * 8);
    ps.texoff = (int*)(base + o); o = align16(o + sizeof(int) * 4);
    ps.bkind = (int*)(base + o); o = align16(o + sizeof(int) * kRunMaxBrains);
    if (KIND == kKindAll) { ps.oflags = (int*)(base + o); o = align16(o + sizeof(int) * 4); }   // (the dueling-only kernel makes no use of it: see RL_XM_DUELING_KERNEL)
    else ps.oflags = nullptr;
    ps.trk_scr = (double*)(base + o); o = align16(o + sizeof(double) * (size_t)cap);
    ps.trk.sum = (double*)(base + o); o = align16(o + sizeof(double) * kRunMaxBrains * RL_TRK_VARS);
    ps.trk.pop = (double*)(base + o); o = align16(o + sizeof(double) * 2);
    ps.trk.cnt = (int*)(base + o); o = align16(o + sizeof(int) * kRunMaxBrains * RL_TRK_VARS);
    return o;
}
// Agent.get_action for the n agents of this world (slot k == list index k): actions into s.action[] and the global
// `actions` buffer.  Must be called by the whole workgroup; leaves with a barrier behind the last action store.
// (Per-brain arguments are fetched from the kernel-argument block with a uniform index -- scalar loads; a by-value copy of the
// argument struct indexed at run time would live in scratch, and at 1024 threads x 256 worlds every dword of scratch per thread
// is 1 MB of memory traffic per tick.)
struct RunParams;
typedef const RunParams __attribute__((address_space(4))) RunParamsC;

// Between two ticks of k_run: the post-update list becomes slots 0..n-1 (slot == list index, what load_world establishes),
// and every per-tick scratch is reset to what load_world leaves behind.  slot_cap <= T: one agent per thread.
// (the reads are issued BEFORE the last observation pass, which leaves the agents alone: their dependent LDS round trips then
// overlap that pass instead of forming an interval of their own)
struct RecycleRegs {
    unsigned short pos;
    int h, age, ma, g, b, u;
    uint8_t fl;
    signed char act;
    double f;
};
__device__ inline RecycleRegs recycle_read(Smem& s, int n)
{
    const int tid = rl_tidx();
    const int a = tid < n ? s.order[tid] : 0;
    RecycleRegs r;
    r.pos = s.pos[a];
    r.h = s.health[a]; r.age = s.age[a]; r.ma = s.max_age[a]; r.g = s.gene[a]; r.b = s.brain[a]; r.u = s.uid[a];
    r.fl = s.flags[a];
    r.act = s.action[a];
    r.f = s.fitness[a];
    return r;
}
// RL_SEAM_OPEN (round 6): between two ticks of ONE launch the closing barrier is left out (`open`): nothing the policy half touches is
// written in here any more -- its Philox keys (s.scal[S_TICK] / [S_EPOCH]) are stored before the first barrier by the caller and only
// re-written with the same values here, its action bytes land in
// s.action[] slots whose carried-over value nobody reads before they are overwritten (so the carry-over itself is left to the launch's
// last tick, where the barrier stays because store_world follows) -- and every wave passes the policy half's own barriers before
// Environment.step reads what is written here.  Kept closed when rows have to be drained to L2 first (tiles reading them from memory).
#ifndef RL_SEAM_OPEN
#define RL_SEAM_OPEN 1
#endif
template <int T, bool SPEC>
__device__ __forceinline__ void recycle_world(const KParams& p, Smem& s, int n, int tick, int epoch, int next_uid, int max_gene, bool drain_stores, const RecycleRegs& rr,
                                              const int* also_drain = nullptr,   // optional LDS flag, valid after the first barrier in here
                                              bool open = false)                 // uniform: a policy half of the same launch follows
{
    const int tid = rl_tidx();
    const bool mine = tid < n;
    const unsigned short r_pos = rr.pos;
    const int r_h = rr.h, r_age = rr.age, r_ma = rr.ma, r_g = rr.g, r_b = rr.b, r_u = rr.u;
    const uint8_t r_fl = rr.fl;
    const signed char r_act = rr.act;
    const double r_f = rr.f;
    lds_barrier();   // every field is in registers; the planes were last read before the barrier that precedes this call
    if (mine) {
        s.pos[tid] = r_pos; s.health[tid] = r_h; s.age[tid] = r_age; s.max_age[tid] = r_ma; s.gene[tid] = r_g; s.brain[tid] = r_b;
        s.uid[tid] = r_u; s.flags[tid] = r_fl; s.fitness[tid] = r_f;
        if (!open) s.action[tid] = r_act;   // (open: the tiles of the next policy half write s.action[k] for every k < n, possibly before this line runs)
        s.aux[tid] = 0; s.src[tid] = (short)tid; s.order[tid] = (short)tid; s.newidx[tid] = (short)tid;
        // the occupancy grid names the agents by slot: every live agent renames its own cell.  No other cell holds a slot: a cell an
        // agent leaves, a corpse's cell and the target of an erased mover are set to -1 where that happens (phase_step, reproduce_wave0),
        // a refill rewrites the whole grid -- so there is no sweep over the cells here, and no barrier between one and the renaming
        // (round 3: clear all, barrier, rename)
        s.occ[(r_pos & 255) * p.W + (r_pos >> 8)] = (short)tid;
    }
    for (int c = tid; c < p.Cp; c += T) ((unsigned*)s.foodv)[c] = 0u;   // (the target counts of the next step live in the food plane)
    for (int i = tid; i < p.hash_size; i += T) { s.hkey[i] = -1; s.hcnt[i] = 0u; }
    if (tid < RL_MAX_BRAINS) s.present[tid] = 0;
    if (SPEC) for (int i = tid; i < 2 * p.cap; i += T) ((unsigned*)s.reward)[i] = 0u;
    if (tid < S_COUNT)
        s.scal[tid] = tid == S_NSLOTS ? n : tid == S_TICK ? tick : tid == S_EPOCH ? epoch : tid == S_NEXT_UID ? next_uid : tid == S_MAX_GENE ? max_gene : 0;
    const bool drain = drain_stores || (also_drain && __builtin_amdgcn_readfirstlane(*also_drain) != 0);   // (uniform)
    if (drain) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this tick's observation rows are in L2 before any wave reads them back
    if (!open || drain) lds_barrier();
}

// The kernel's only parameter.  The two halves of a tick are real (noinline) function calls, each with a register allocation
// of its own (inlined into one body the tile code's ~126 VGPRs and the tick's hoisted loop invariants spill into each other's
// loops), and the kernel body keeps NOTHING alive across them: the loop state lives in LDS (PolSmem::meta), because whatever a
// caller holds in registers across a call of a 128-VGPR callee goes through scratch.
struct RunParams {
    KParams p;
    RunArgs ra;
};
// Exploration rate of brains-list entry b in this tick: TRAIN launches may carry a schedule (one row per tick of the launch: the
// reference's brains decay epsilon from episode to episode, D3QN.py:84-89, DQN.py:67-69), else the brain's constant one.  Uniform: scalar loads.
template <int TRAIN>
__device__ inline float run_eps(RunParamsC* ka, int b, int n_brains, const PolSmem& ps)
{
    typedef const float __attribute__((address_space(4))) cfloat;
    if (TRAIN) {
        if (*(const int __attribute__((address_space(4)))*)&ka->ra.eps_inline_on)
            return ((cfloat*)ka->ra.eps_inline)[__builtin_amdgcn_readfirstlane(ps.meta[3]) * n_brains + b];
        const float* es = *(const float* const __attribute__((address_space(4)))*)&ka->ra.eps_sched;
        if (es) return ((cfloat*)es)[__builtin_amdgcn_readfirstlane(ps.meta[3]) * n_brains + b];
    }
    return ((cfloat*)ka->ra.eps)[b];
}

// Row stride of the observation planes in this kernel: padded (plane_stride: the window gather hits every LDS bank once) with ONE world
// per CU, i.e. workgroups of >= 512 threads; row-major for 256-thread workgroups, which share a CU's LDS (rl_world_dev.h plane_stride).
__host__ __device__ constexpr int run_plane_stride(int T, int W, int H) { return T >= 512 ? plane_stride(W, H) : W; }
template <bool FIXED, int T>
__device__ inline KParams run_params(RunParamsC* ka)
{
    // A struct copy out of the CONSTANT address space: every field that is used becomes a scalar load of the kernel-argument
    // block.  Only the device pass can express it (for the host pass the implicit copy constructor cannot bind an
    // address-space-qualified reference).
#if defined(__HIP_DEVICE_COMPILE__)
    KParams p = *(const KParams __attribute__((address_space(4)))*)&ka->p;
#else
    KParams p{};
#endif
    if (FIXED) {
        p.W = kFixW; p.H = kFixH; p.C = kFixC; p.Cp = kFixCp; p.nW = kFixCp / 64; p.PS = run_plane_stride(T, kFixW, kFixH); p.invW = kFixInvW;
        p.cap = kFixCap; p.hash_size = kFixHash; p.hash_mask = kFixHash - 1;
    }
    return p;
}
// LDS of the multi-tick kernel per workgroup size: T >= 512 (one world per CU) mirrors the Agent.state rows in what is left of the CU's
// 160 KB -- the two-wave (T = 512) and four-wave (T = 1024) tiles read their rows there, and their exchange buffers alias it; T = 256
// (several worlds per CU) runs the one-wave tile, which reads its rows back from L2.
constexpr size_t kRunLdsBudget = 160 * 1024;
__host__ __device__ constexpr size_t run_mirror_budget(int T) { return T >= 512 ? kRunLdsBudget : 0; }
__host__ __device__ constexpr int run_cbrains(int T, int n_brains) { return T >= 512 ? n_brains : 0; }   // the hand-scheduled tiles keep their epilogue constants in LDS
template <int KIND> __host__ __device__ constexpr int run_tile_floats(int T) { return T == 1024 ? kQuadFloats : run_pair_floats<KIND>(); }
__host__ __device__ constexpr size_t run_min_region(int T) { return T == 1024 ? 4 * (size_t)kQuadExBytes : 0; }
template <bool FIXED, int KIND>
__device__ inline void run_carve(const KParams& p, Smem& s, PolSmem& ps, char* smem_raw, int T)
{
    const size_t o0 = FIXED ? carve(s, smem_raw, kFixCp, kFixCap, kFixHash, plane_words(run_plane_stride(T, kFixW, kFixH), kFixH, kFixCp))
                            : carve(s, smem_raw, p.Cp, p.cap, p.hash_size, plane_words(p.PS, p.H, p.Cp));
    carve_policy<KIND>(ps, smem_raw, o0, p.cap, run_mirror_budget(T), run_cbrains(T, p.n_brains), run_tile_floats<KIND>(T), run_min_region(T));
}

// Rows of a world grouped by brain, 32-row tiles per brain, by ONE wave: trow / tbrain / tstart / bcnt / meta[0].  `brain_of(k)`:
// brains-list index of list entry k.  Every entry takes a position in its brain's list with a RETURNING LDS atomic -- any order will do:
// the rows of a tile are independent of each other (per-row scales, per-row outputs addressed by list index, Philox keyed by it), so the
// results do not depend on which tile row an agent sits in -- and parks brain | position in prow; once the brains' tile ranges are known
// it writes its list index straight into its tile row.  Padding rows read list entry 0 and are marked invalid.  (The ballot version --
// per chunk and brain a ballot / popcount step, twice, then a gather for the descriptors -- was ~200 instructions of a lone wave.)
template <typename F>
__device__ inline void policy_lists_wave0(const KParams& p, PolSmem& ps, int n, int lane, F brain_of)
{
    if (lane < p.n_brains) ps.bcnt[lane] = 0;
    for (int base = 0; base < n; base += 64) {
        const int k = base + lane;
        if (k < n) {
            const int b = brain_of(k);
            const int pos = __hip_atomic_fetch_add(&ps.bcnt[b], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            ps.prow[k] = (short)(unsigned short)(pos | (b << 10));   // (pos < 1024 entries per world, b < 64)
        }
    }
    const int mine = lane < p.n_brains ? ps.bcnt[lane] : 0;
    const int tiles = (mine + 31) >> 5;
    const int tincl = wave_incl_scan(tiles);
    const int first_tile = tincl - tiles;
    if (lane < p.n_brains) ps.tstart[lane] = first_tile;
    if (lane == 63) ps.meta[0] = tincl;
    const int tt = min(read_lane(tincl, 63), kMaxTiles);
    for (int e = lane; e < tt * 32; e += 64) ps.trow[e] = (short)0x8000;
    if (lane < tt) {
        int b = 0;
        for (int bb = 1; bb < p.n_brains; ++bb) if (read_lane(first_tile, bb) <= lane && read_lane(mine, bb) > 0) b = bb;
        ps.tbrain[lane] = b;
    }
    for (int base = 0; base < n; base += 64) {
        const int k = base + lane;
        if (k < n) {
            const int e = (unsigned short)ps.prow[k], pos = e & 1023, b = e >> 10;
            const int t = ps.tstart[b] + (pos >> 5);
            if (t < kMaxTiles) ps.trow[t * 32 + (pos & 31)] = (short)k;
        }
    }
}

// kKindAll kernels: the policy half's schedule for the tiles policy_lists_wave0 just described (same wave, program order).  Every tile
// is handled by TWO waves (policy_tile1s<PAIR> / policy_pair2) when the world has at most four tiles and their exchange buffers fit into
// the Agent.state mirror; the two roles of a tile may sit on any two wave slots (only LDS and the workgroup barriers connect them), so
// the tiles are dealt heaviest first, each role onto the least loaded SIMD with a free slot (waves v and v + 4 share SIMD v): a PPO tile
// (672 MFMAs) next to a dueling one (360) loads every SIMD with 516 instead of 672 / 360.  Otherwise: the same tiles in several rounds.
// RL_XF_* of the Agent.state rows of a world whose state is in LDS (n rows, built by build_planes / write_observations from THIS state):
// see in_lo_zero (rl_policy_dev.h).  bad_health: some agent of the launch's initial state had |health| >= 400.
__device__ inline int run_obs_flags(const KParams& p, const Smem& s, int n, int bad_health)
{
    if (bad_health || n >= 2 * p.max_agents) return 0;
    return RL_XF_SCALE | (s.type[0] == RL_AGENT ? 0 : RL_XF_INT_HEALTH);
}

__device__ inline void policy_schedule_wave0(const KParams& p, PolSmem& ps, RunParamsC* ka, int lane, bool first_call = false)
{
    // Lane-parallel and in closed form (as serial single-lane code with a memory load per tile this cost 5.5k cycles on the wave that is
    // the longest of its interval): lane t < 4 owns tile t.  With at most four tiles the balanced deal has a fixed shape -- the heaviest
    // tile's roles on SIMDs 0 and 1, the second's on SIMDs 2 and 3, the lightest joins the heaviest, the third the second -- so a wave
    // only needs the tile of the cost RANK its slot stands for.
    const int nt = __builtin_amdgcn_readfirstlane(ps.meta[0]);
    const int t = lane & 3;
    const int brain = ps.tbrain[t] & (kRunMaxBrains - 1);
    const int kind = ps.bkind[brain];
    const int cost = t < nt ? (kind == RL_PPO ? 672 : kind == RL_DQN ? 180 : 360) : 0;   // MFMAs per tile
    const int ex = t < nt ? pair_ex_bytes(kind) : 0;
    int rank = 0, off = 0, total_ex = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int cu = __builtin_amdgcn_readlane(cost, u), eu = __builtin_amdgcn_readlane(ex, u);
        rank += (cu > cost || (cu == cost && u < t)) ? 1 : 0;
        off += u < t ? eu : 0;
        total_ex += eu;
    }
    const size_t mirror_bytes = (size_t)ps.xrows * kXStride * sizeof(float);
    const bool pair_ok = nt <= 4 && ps.xmirror != nullptr && ps.pairv != nullptr && (size_t)total_ex <= mirror_bytes;
    const int fit = (int)(mirror_bytes / (size_t)pair_ex_bytes(RL_PPO));
    // wave v = slot (v >> 2) of SIMD v & 3: ranks 0 0 1 1 / 3 3 2 2 (heaviest with lightest, second with third), role = v & 1
    const int want = (0x22331100 >> (4 * (lane & 7))) & 15;
    int tile = -1, tb = 0, tk = 0;   // (brain and kind ride along: the tile wave then needs neither tbrain[] nor the kind table)
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int bu = __builtin_amdgcn_readlane(brain, u), ku = __builtin_amdgcn_readlane(kind, u);
        if (u < nt && __builtin_amdgcn_readlane(rank, u) == want) { tile = u; tb = bu; tk = ku; }
    }
    if (lane < 4) ps.texoff[lane] = off;
    if (lane < 8) ps.wtask[lane] = (pair_ok && tile >= 0) ? (tile | ((lane & 1) << 8) | (tb << 12) | (tk << 20)) : -1;
    // (meta[6], the tiles per round when the tiles take several rounds, does not change during a launch: written once, by run_load_call
    // -- written here it was a loop invariant of the tick loop, spilled, and reloaded every tick behind an s_waitcnt vmcnt(0) on the wave
    // that is the long pole of its interval)
    if (lane == 0) {
        ps.meta[5] = pair_ok ? 0 : 1;
        if (first_call) ps.meta[6] = fit < 1 ? 1 : (fit > 4 ? 4 : fit);
    }
}

// The policy half of the dueling-kind kernels.  T = 512 (256 VGPRs per wave), up to four tiles: TWO waves per tile on one SIMD
// (policy_tile1s<PAIR>); five to eight tiles: one hand-scheduled tile per wave (policy_tile1s); T = 256: one policy_tile1 per
// wave (no LDS, no barrier inside a tile), wave i takes tiles i, i + 4, ...; T = 1024 (128 VGPRs per wave): FOUR waves per tile on one SIMD
// (policy_quad), rounds of four tiles.  Tile rows, validity and brain come from the descriptors wave 0 wrote next to the
// row lists (policy_lists_wave0).
template <int T, int KIND, int TRAIN>
__device__ __forceinline__ void run_policy1(const KParams& p, Smem& s, PolSmem& ps, RunParamsC* ka, int w, int n, const float* obs_rows, char* smem_base, int wave)
{
    // (`wave`: the wave's index in the workgroup, uniform -- computed once per launch and kept in an SGPR)
    const int lane = rl_lane_fresh(), j = lane & 31;
    [[maybe_unused]] const int tid = wave * 64 + lane;   // (the stamped build)
#ifdef RL_PHASE_PROFILE
    const long long t_entry = (long long)clock64();
#endif
    // (the per-brain row lists were built by wave 0 while the other waves wrote the previous tick's Agent.state rows: policy_lists_wave0)
    int ntiles = __builtin_amdgcn_readfirstlane(ps.meta[0]);
    {   // measurement only (run mask & 4 / & 8): run at most 2 / 1 tiles (results WRONG)
        const int dbg = RL_RUN_DBG(ka);
        if (dbg & 4) ntiles = min(ntiles, 2);
        if (dbg & 8) ntiles = min(ntiles, 1);
        if (dbg & 16) ntiles = min(ntiles, 3);
    }
    const bool mirrored = ps.xmirror != nullptr && __builtin_amdgcn_readfirstlane(ps.meta[4]) != 0;
    auto tile_io = [&](int ti, TileIO& io, int j) {   // j: tile row of the lane
        const int b = __builtin_amdgcn_readfirstlane(ps.tbrain[ti]);
        const int e = (unsigned short)ps.trow[ti * 32 + j], k = e & 0x7fff;
        io.packed = (gfloat*)((const float* const __attribute__((address_space(4)))*)ka->ra.packed)[b];
        io.obs = obs_rows;
        io.row = (int64_t)w * p.cap + k;
        io.valid = !(e & 0x8000);
        io.eps = run_eps<TRAIN>(ka, b, p.n_brains, ps);
        io.out = TRAIN == 2 ? *(float* const __attribute__((address_space(4)))*)&ka->ra.policy_out : nullptr;
        io.actions = *(int8_t* const __attribute__((address_space(4)))*)&ka->ra.actions;
        io.seed = p.seed;
        io.key_world = (uint32_t)(p.world_base + w); io.key_tick = (uint32_t)s.scal[S_TICK]; io.key_epoch = (uint32_t)s.scal[S_EPOCH];
        io.key_index = (uint32_t)k;
        io.lds_actions_off = (int)((char*)s.action - smem_base); io.lds_slot = k;
        io.x_lds_off = (mirrored && k < ps.xrows) ? (int)((char*)(ps.xmirror + k * kXStride) - smem_base) : -1;
        io.c_lds_off = ps.cconst ? (int)((char*)(ps.cconst + run_const_floats<KIND>() * b) - smem_base) : -1;
#ifdef RL_PHASE_PROFILE
        io.prof = (p.prof && (int)blockIdx.x == p.prof_world && wave == 0) ? p.prof : nullptr;
        if (io.prof && lane == 0) { io.prof[100] = t_entry; io.prof[110] = (long long)clock64(); }
#endif
    };
    if constexpr (T == 1024) {
        // FOUR waves per tile, all on one SIMD (waves t, t + 4, t + 8, t + 12): policy_quad.  More than four tiles (> 128 agents or an uneven
        // brain split): rounds of four, rows from memory (the exchange slices alias the mirror; recycle_world drained the rows: meta[5]).
        const int q = __builtin_amdgcn_readfirstlane(wave >> 2), slot = wave & 3;
        for (int t0 = 0; t0 < ntiles; t0 += 4) {
            // (a fresh lane index per round: nothing per-lane can be hoisted out of this loop and kept alive -- spilled -- across the tile)
            const int lane = rl_lane_fresh(), j = lane & 31;
            const int ti = t0 + slot;
            const bool have = ti < ntiles;
            TileIO io;
            Tile1Part part;
            QuadLds ql;
            ql.pmax = ps.pairv + kQuadFloats * slot; ql.val = ql.pmax + 3 * 32 * 4;
            ql.ex = (f32x4*)((char*)ps.xmirror + (size_t)kQuadExBytes * slot);
#ifdef RL_PHASE_PROFILE
            ql.prof = (p.prof && (int)blockIdx.x == p.prof_world && slot == 0) ? p.prof : nullptr;
#endif
            if (have) {
                tile_io(ti, io, j);
                if (ntiles > 4) io.x_lds_off = -1;
                policy_quad<KIND, RL_RUN_COHERENT>(io, lane, q, &ql, &part);
            } else {
#pragma unroll
                for (int i = 0; i < kQuadBarriers; ++i) lds_barrier();
            }
            lds_barrier();
            if (have && q == 0) tile1_finish<KIND>(io, lane, part.head, ql.val[j], part.draw, *(const f32x4*)((const float*)(smem_base + io.c_lds_off) + 768 + 8 + 4 * (lane >> 5)));
        }
    } else
    if (T == 512 && ps.pairv != nullptr && ntiles <= 4 && (size_t)ps.xrows * kXStride * sizeof(float) >= 4 * (size_t)kPairExBytes) {
        // TWO waves per tile, on the same SIMD (waves i and i + 4): policy_tile1s<PAIR>
        const int role = __builtin_amdgcn_readfirstlane(wave >> 2), slot = wave & 3;
        const bool have = slot < ntiles;
        TileIO io;
        Tile1Part part;
        PairLds pl;
        pl.val = ps.pairv + kPairFloats * slot; pl.pmax = pl.val + 32;
        pl.ex = (f32x4*)((char*)ps.xmirror + (size_t)kPairExBytes * slot);
        if (have) {
            tile_io(slot, io, j);
            policy_tile1s<KIND, RL_RUN_COHERENT, true, RL_XM_DUELING_KERNEL>(io, lane, role, &pl, &part);
        } else { lds_barrier(); lds_barrier(); }   // (the two exchanges inside the tile)
#ifdef RL_PHASE_PROFILE
        if (p.prof && (int)blockIdx.x == p.prof_world && lane == 0) p.prof[116 + wave] = (long long)clock64();   // (128 slots)
#endif
        lds_barrier();
        if (have && role == 0) {   // (a fresh lane index: `lane` / `j` from above the tile lived through it -- spilled in the TRAIN instantiations, a reload behind vmcnt(0) right here)
            const int fl = rl_lane_fresh();
            tile1_finish<KIND>(io, fl, part.head, pl.val[fl & 31], part.draw, *(const f32x4*)((const float*)(smem_base + io.c_lds_off) + 768 + 8 + 4 * (fl >> 5)));
        }
    } else
    for (int ti = wave; ti < ntiles; ti += T / 64) {
        TileIO io;
        tile_io(ti, io, j);
        if (T == 512) policy_tile1s<KIND, RL_RUN_COHERENT, false, RL_XM_DUELING_KERNEL>(io, lane);
        else policy_tile1<KIND, RL_RUN_COHERENT, true>(io, lane);
    }
#ifdef RL_PHASE_PROFILE
    if (p.prof && (int)blockIdx.x == p.prof_world && tid == 0) p.prof[111] = (long long)clock64();
#endif
    lds_barrier();
#ifdef RL_PHASE_PROFILE
    if (p.prof && (int)blockIdx.x == p.prof_world && tid == 0) p.prof[112] = (long long)clock64();
#endif
}

// RL_XM_COLD_CALL (round 6): the GENERAL copies of the tiles (nothing assumed about the rows: one tick in ten) as ONE
// out-of-line function with a register allocation of its own, so that the kernel body's allocation is that of the hot copies alone -- what
// the forced build `f3` had (5 spilled VGPRs against 12-14 with both copies inlined).
#ifndef RL_XM_COLD_CALL
#define RL_XM_COLD_CALL 0
#endif
// RL_XM_LIKELY: the branch between the two copies of a tile carries its frequency (nine ticks in ten take the copy that knows its rows), so
// that the register allocator's spill weights and the block layout favour that copy (A/B: DESIGN.md 5.12)
#ifndef RL_XM_LIKELY
#define RL_XM_LIKELY 1
#endif
#if RL_XM_LIKELY
#define RL_XM_EXPECT(x) __builtin_expect_with_probability(!!(x), 1, 0.9)
#else
#define RL_XM_EXPECT(x) (x)
#endif
__device__ __attribute__((noinline, cold)) Tile1Part run_tile_general(TileIO io, PairLds pl, int role, int kind)
{
    Tile1Part part;
    const int lane = rl_lane_fresh();
    if (kind == RL_DQN) policy_pair2<RL_DQN, RL_RUN_COHERENT>(io, lane, role, &pl, &part);
    else if (kind == RL_PPO) policy_pair2<RL_PPO, RL_RUN_COHERENT>(io, lane, role, &pl, &part);
    else policy_tile1s<RL_PERD3QN, RL_RUN_COHERENT, true>(io, lane, role, &pl, &part);
    return part;
}

// The policy half of the kKindAll kernels (512-thread workgroups): per tile, the code of its brain's kind.
template <int T, int TRAIN>
__device__ __forceinline__ void run_policy_all(const KParams& p, Smem& s, PolSmem& ps, RunParamsC* ka, int w, int n, const float* obs_rows, char* smem_base, int wave)
{
    static_assert(T == 512, "kKindAll: 512-thread workgroups");
    typedef const int __attribute__((address_space(4))) cint;
    const int lane = rl_lane_fresh(), j = lane & 31;
#ifdef RL_PHASE_PROFILE
    if (p.prof && (int)blockIdx.x == p.prof_world && wave == 0 && lane == 0) p.prof[100] = (long long)clock64();
#endif
    const int ntiles = __builtin_amdgcn_readfirstlane(ps.meta[0]);
    const bool mirrored = ps.xmirror != nullptr && __builtin_amdgcn_readfirstlane(ps.meta[4]) != 0;
    const bool fallback = __builtin_amdgcn_readfirstlane(ps.meta[5]) != 0;
    // (read with the schedule, in ONE batch of LDS loads: not on the way into every tile)
    constexpr int xneed = RL_XM_ALL_KERNEL == 2 ? (RL_XF_SCALE | RL_XF_INT_HEALTH) : RL_XF_SCALE;
#ifdef RL_XM_ASSUME_ALWAYS   /* negative control of tests/test_hip_round5.py (results WRONG where the rows are not what is assumed) */
    const bool rows_known = true;
#else
    const bool rows_known = RL_XM_ALL_KERNEL != 0 && (__builtin_amdgcn_readfirstlane(ps.oflags[0]) & xneed) == xneed;
#endif
    auto tile_io = [&](int ti, TileIO& io, bool from_mirror, int task) {   // task >= 0: the wave's descriptor names brain and kind
        const int b = task >= 0 ? ((task >> 12) & 255) : __builtin_amdgcn_readfirstlane(ps.tbrain[ti]);
        const int e = (unsigned short)ps.trow[ti * 32 + j], k = e & 0x7fff;
        io.packed = (gfloat*)((const float* const __attribute__((address_space(4)))*)ka->ra.packed)[b];
        io.obs = obs_rows;
        io.row = (int64_t)w * p.cap + k;
        io.valid = !(e & 0x8000);
        io.eps = run_eps<TRAIN>(ka, b, p.n_brains, ps);
        io.out = TRAIN == 2 ? *(float* const __attribute__((address_space(4)))*)&ka->ra.policy_out : nullptr;
        io.actions = *(int8_t* const __attribute__((address_space(4)))*)&ka->ra.actions;
        io.seed = p.seed;
        io.key_world = (uint32_t)(p.world_base + w); io.key_tick = (uint32_t)s.scal[S_TICK]; io.key_epoch = (uint32_t)s.scal[S_EPOCH];
        io.key_index = (uint32_t)k;
        io.lds_actions_off = (int)((char*)s.action - smem_base); io.lds_slot = k;
        io.x_lds_off = (from_mirror && mirrored && k < ps.xrows) ? (int)((char*)(ps.xmirror + k * kXStride) - smem_base) : -1;
        io.c_lds_off = (int)((char*)(ps.cconst + kTileConstMax * b) - smem_base);
#ifdef RL_PHASE_PROFILE
        io.prof = (p.prof && (int)blockIdx.x == p.prof_world && wave == ((p.ablate >> 20) & 7)) ? p.prof : nullptr;   // (tuning: the stamped wave = bits 20-22 of the ablate mask)
#endif
        return task >= 0 ? ((task >> 20) & 7) : ((cint*)ka->ra.kind)[b];
    };
    // One tile on a pair of waves: policy_tile1s<PAIR> (dueling kinds) / policy_pair2 (DQN, PPO); every wave meets the same barriers.
    auto pair_round = [&](bool have, int ti, int role, int slot, int ex_off, bool from_mirror, int task) {
        TileIO io;
        Tile1Part part;
        PairLds pl;
        int kind = -1;
        if (have) {
            kind = __builtin_amdgcn_readfirstlane(tile_io(ti, io, from_mirror, task));
            pl.val = ps.pairv + kPairFloatsAll * slot; pl.pmax = pl.val + kPairValFloats;
            pl.ex = (f32x4*)((char*)ps.xmirror + ex_off);
            // two copies of every tile: the rows of THIS world and tick are known to be scaled by 2^10 with exactly-zero lo halves in the
            // plane chunks (oflags[0], run_obs_flags: ~9 in 10 ticks) -- or nothing is assumed.  Bit-identical either way.
            const bool plain = kind == RL_DQN || kind == RL_PPO;
            if (!plain && RL_XM_ALL_DUELING_TILE == 0) policy_tile1s<RL_PERD3QN, RL_RUN_COHERENT, true>(io, lane, role, &pl, &part);   // (ONE copy of this tile)
            else if (RL_XM_EXPECT(rows_known)) {
                if (kind == RL_DQN) policy_pair2<RL_DQN, RL_RUN_COHERENT, RL_XM_ALL_KERNEL>(io, lane, role, &pl, &part);
                else if (kind == RL_PPO) policy_pair2<RL_PPO, RL_RUN_COHERENT, RL_XM_ALL_KERNEL>(io, lane, role, &pl, &part);
                else policy_tile1s<RL_PERD3QN, RL_RUN_COHERENT, true, RL_XM_ALL_DUELING_TILE>(io, lane, role, &pl, &part);
            } else if (RL_XM_COLD_CALL) part = run_tile_general(io, pl, role, kind);
            else {
                if (kind == RL_DQN) policy_pair2<RL_DQN, RL_RUN_COHERENT>(io, lane, role, &pl, &part);
                else if (kind == RL_PPO) policy_pair2<RL_PPO, RL_RUN_COHERENT>(io, lane, role, &pl, &part);
                else policy_tile1s<RL_PERD3QN, RL_RUN_COHERENT, true>(io, lane, role, &pl, &part);
            }
        } else { lds_barrier(); lds_barrier(); }   // (the two exchanges inside a tile)
        lds_barrier();
        if (have && role == 0) {
            const int fl = rl_lane_fresh();   // (not the lane index from above the tile: see run_policy1)
            if (kind == RL_DQN) pair_finish<RL_DQN>(io, fl, part, &pl);
            else if (kind == RL_PPO) pair_finish<RL_PPO>(io, fl, part, &pl);
            else tile1_finish<RL_PERD3QN>(io, fl, part.head, pl.val[fl & 31], part.draw, *(const f32x4*)((const float*)(smem_base + io.c_lds_off) + 768 + 8 + 4 * (fl >> 5)));
        }
    };
    // meta[5] == 0: every tile in ONE round, the waves dealt over the SIMDs by cost (policy_schedule_wave0), rows from the mirror.
    // Otherwise (crowded worlds, many brains): rounds of meta[6] tiles on fixed wave pairs (waves s and s + 4), a 32 KB exchange buffer
    // per slot; round 1's buffers overwrite the mirror, so every tile reads its rows from memory (recycle_world drained the stores).
    // The same tiles, hence the same bits, either way.  (ONE call site: the tile code exists once in the kernel.)
    const int pr = __builtin_amdgcn_readfirstlane(ps.meta[6]);
    const int task = __builtin_amdgcn_readfirstlane(ps.wtask[wave]);
    const int n_rounds = fallback ? (ntiles + pr - 1) / pr : 1;
    for (int r = 0; r < n_rounds; ++r) {
        const int slot = fallback ? (wave & 3) : (task & 255);
        const int ti = fallback ? r * pr + slot : slot;
        const bool have = fallback ? (slot < pr && ti < ntiles) : task >= 0;
        const int role = __builtin_amdgcn_readfirstlane(fallback ? (wave >> 2) : ((task >> 8) & 1));
        const int ex_off = fallback ? slot * pair_ex_bytes(RL_PPO) : (have ? ps.texoff[slot] : 0);
        pair_round(have, ti, role, slot, ex_off, !fallback, fallback ? -1 : task);
        if (fallback) lds_barrier();   // (the finish read this round's partials; the next round overwrites them)
    }
#ifdef RL_PHASE_PROFILE
    if (p.prof && (int)blockIdx.x == p.prof_world && lane == 0) p.prof[116 + wave] = (long long)clock64();   // (128 slots)
    if (p.prof && (int)blockIdx.x == p.prof_world && wave == 0 && lane == 0) p.prof[111] = (long long)clock64();
#endif
    lds_barrier();
#ifdef RL_PHASE_PROFILE
    if (p.prof && (int)blockIdx.x == p.prof_world && wave == 0 && lane == 0) p.prof[112] = (long long)clock64();
#endif
}

// First half of a tick: the policy.  Reads the list length and the Agent.state parity from LDS.
template <int T, bool FIXED, int KIND, int TRAIN>
__device__ __forceinline__ void run_policy_half(RunParamsC* ka, int wave)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const KParams p = run_params<FIXED, T>(ka);
    Smem s;
    PolSmem ps;
    run_carve<FIXED, KIND>(p, s, ps, smem_raw, T);
    const int w = blockIdx.x;
    const int n = __builtin_amdgcn_readfirstlane(ps.meta[1]), cur = __builtin_amdgcn_readfirstlane(ps.meta[2]);
    const float* obs_in = ((float* const __attribute__((address_space(4)))*)ka->ra.obs)[cur];
    if constexpr (KIND == kKindAll) run_policy_all<T, TRAIN>(p, s, ps, ka, w, n, obs_in, smem_raw, wave);
    else run_policy1<T, KIND, TRAIN>(p, s, ps, ka, w, n, obs_in, smem_raw, wave);
}

// The observation planes of the post-UPDATE grid are the post-step planes with two kinds of cells changed: a corpse's cell holds Food now
// (environment.py:795-799) and a newborn's cell holds an agent of health 200 that is not dead (h = 1 in either dtype mode, f = 0, no gene
// entry).  Patching those few cells replaces a sweep over the whole grid.  One exception: np.vectorize infers the health plane's dtype from
// cell (0,0) (build_planes), so a birth or a death THERE changes every agent cell: the caller then rebuilds the planes (S_PLANES_DIRTY).
// Runs after the barrier behind reproduce_wave0 (order[] still lists the post-step agents; the newborns occupy slots first_new .. end_new-1).
template <int T>
__device__ inline void patch_planes_after_update(const KParams& p, Smem& s, int n1, int first_new, int end_new)
{
    const int tid = rl_tidx();
    bool cell0 = false;
    for (int k = tid; k < n1; k += T) {
        const int a = s.order[k];
        const int fl = s.flags[a], ps = s.pos[a];  // one batch
        if (fl & RL_F_DEAD) {
            const int c = (ps & 255) * p.PS + (ps >> 8);   // (plane word; cell (0,0) is word 0)
            s.foodv[c] = 0.5f; s.healthv[c] = -1.f; s.genev[c] = -2;
            cell0 |= c == 0;
        }
    }
    for (int i = first_new + tid; i < end_new; i += T) {
        const int c = cell_to_plane(p, s.tgt[i]);
        s.foodv[c] = 0.f; s.healthv[c] = 1.f; s.genev[c] = -2;
        cell0 |= c == 0;
    }
    if (cell0) s.scal[S_PLANES_DIRTY] = 1;
}

// trainer.py:95-96 + entities.py:194-208 inside the multi-tick launch (TRAIN, ra.capture): what rl_capture_transitions does after a
// stand-alone tick.  Two parts.  capture_reserve (ONE wave, next to _reproduce on wave 0): the post-step agents with age > 1 are counted
// per brain with ballots, one 64-bit atomic per brain reserves their ring slots, slot[k] (or -1) goes to LDS.  capture_rows (the whole
// workgroup, an interval of its own before the planes are patched for the post-update grid): every captured agent's state_prime row is
// produced once more from the post-step planes straight into its ring slot, its state row -- the observation the policy read, still in
// the Agent.state buffer of this tick: slot a of the pre-step list -- is copied from memory (sc1: written by this workgroup a tick ago,
// or by the previous launch), and the scalars follow.  Order within a brain's ring: Agent.learn call order per world, worlds interleaved
// by the atomics (as with rl_capture_transitions).
__device__ inline rl_replay load_replay(RunParamsC* ka, int b)   // (a struct copy out of the constant address space: device pass only, like run_params)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return *(const rl_replay __attribute__((address_space(4)))*)&ka->ra.rp[b];
#else
    return rl_replay{};
#endif
}
__device__ inline void capture_reserve(const KParams& p, Smem& s, RunParamsC* ka, int n1, int* slot)
{
    const int lane = lane_id();
    int cnt = 0;
    for (int base = 0; base < n1; base += 64) {
        const int k = base + lane;
        const int a = k < n1 ? s.order[k] : 0;
        const int br = (k < n1 && s.age[a] > 1) ? s.brain[a] : -1;   // Agent.learn: `if self.age > 1` (entities.py:196)
        for (int bb = 0; bb < p.n_brains; ++bb) { const int c = __popcll(__ballot(br == bb)); if (lane == bb) cnt += c; }
    }
    unsigned long long pos = 0, capacity = 1;
    if (lane < p.n_brains) {
        const rl_replay R = load_replay(ka, lane);
        capacity = (unsigned long long)R.capacity;
        if (cnt) pos = atomicAdd(R.count, (unsigned long long)cnt);
    }
    for (int base = 0; base < n1; base += 64) {
        const int k = base + lane;
        const int a = k < n1 ? s.order[k] : 0;
        const int br = (k < n1 && s.age[a] > 1) ? s.brain[a] : -1;
        int mine = -1;
        for (int bb = 0; bb < p.n_brains; ++bb) {
            const unsigned long long m = __ballot(br == bb);
            const unsigned long long start = read_lane_u64(pos, bb), capb = read_lane_u64(capacity, bb);
            if (br == bb) mine = (int)((start + (unsigned long long)__popcll(m & lowmask(lane))) % capb);
            if (lane == bb) pos += (unsigned long long)__popcll(m);
        }
        if (k < n1) slot[k] = mine;
    }
}
template <int T>
__device__ __forceinline__ void capture_rows(const KParams& p, Smem& s, RunParamsC* ka, int w, int n1, const int* slot, const float* state_rows)
{
    const int t = rl_tidx();
    constexpr int G = T / 49;
    const int g0 = t / 49, idx = t - g0 * 49;
    const int dr = idx / 7 - 3, dc = idx - (idx / 7) * 7 - 3;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)state_rows, 0, 0x7fffffff, 0x00027000);
    auto ld = [&](int64_t float_index) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)(float_index * 4), 0, 16 /* sc1 */)); };
    const int64_t wbase = (int64_t)w * p.cap;
    if (g0 < G)
        for (int k = g0; k < n1; k += G) {
            const int sl = slot[k];
            if (sl < 0) continue;
            const int a = s.order[k];
            const rl_replay R = load_replay(ka, s.brain[a]);
            const int pa = s.pos[a];
            int ci = (pa & 255) + dr, cj = (pa >> 8) + dc;
            ci += ci < 0 ? p.H : 0; ci -= ci >= p.H ? p.H : 0;
            cj += cj < 0 ? p.W : 0; cj -= cj >= p.W ? p.W : 0;
            const int c = ci * p.PS + cj;
            const int g = s.genev[c];
            float* d1 = R.state_prime + (size_t)sl * RL_OBS_DIM + idx;
            d1[0] = s.foodv[c]; d1[49] = s.healthv[c]; d1[98] = g == -2 ? 0.f : (g == s.gene[a] ? 1.f : -1.f);   // (write_observations' values)
            const int64_t src = (wbase + a) * RL_OBS_DIM + idx;   // (slot a of the tick == index a of the pre-step list: the row the policy read)
            float* d0 = R.state + (size_t)sl * RL_OBS_DIM + idx;
            d0[0] = ld(src); d0[49] = ld(src + 49); d0[98] = ld(src + 98);
        }
    float* const pout = *(float* const __attribute__((address_space(4)))*)&ka->ra.policy_out;
    for (int k = t; k < n1; k += T) {
        const int sl = slot[k];
        if (sl < 0) continue;
        const int a = s.order[k];
        const rl_replay R = load_replay(ka, s.brain[a]);
        const int same = (int)(s.hcnt[s.hslot[a]] >> 16);
        float* d1 = R.state_prime + (size_t)sl * RL_OBS_DIM + 147;
        d1[0] = (float)((double)s.health[a] * rl_one_200th());
        d1[1] = (s.flags[a] & RL_F_REPRODUCED) ? 1.f : 0.f;
        d1[2] = (float)((double)same / (double)n1);
        d1[3] = (float)((double)n1 / (double)p.max_agents);
        d1[4] = (s.flags[a] & RL_F_KILLED) ? 1.f : 0.f;
        d1[5] = (s.flags[a] & RL_F_ATE_SUPER) ? 1.f : -1.f;
        const int64_t src = (wbase + a) * RL_OBS_DIM + 147;
        float* d0 = R.state + (size_t)sl * RL_OBS_DIM + 147;
#pragma unroll
        for (int e = 0; e < 6; ++e) d0[e] = ld(src + e);
        const int act = s.action[a];
        R.action[sl] = (int8_t)act;
        R.reward[sl] = (float)s.reward[a];
        R.done[sl] = (s.flags[a] & RL_F_DEAD) ? 1 : 0;
        R.age[sl] = s.age[a];
        if (pout && R.prob && act >= 0 && act < 8) {
            const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc((void*)pout, 0, 0x7fffffff, 0x00027000);
            R.prob[sl] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rq, (int)(((wbase + a) * 8 + act) * 4), 0, 16));
        }
    }
}

// Second half: Environment.step + update_env (+ re-generation) out of LDS, then recycle_world.  Same sequence as
// k_world<T, MODE_TICK, LEAN>; writes Agent.state into ra.obs[cur ^ 1] and advances the loop state in LDS.
template <int T, bool FIXED, int KIND, int TRAIN>
__device__ __forceinline__ void run_tick_body(RunParamsC* ka)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const KParams p = run_params<FIXED, T>(ka);
    Smem s;
    PolSmem ps;
    run_carve<FIXED, KIND>(p, s, ps, smem_raw, T);
    const int w = blockIdx.x;
    const int n0 = __builtin_amdgcn_readfirstlane(ps.meta[1]), cur = __builtin_amdgcn_readfirstlane(ps.meta[2]);
    const int ticks_done = __builtin_amdgcn_readfirstlane(ps.meta[3]);
    float* const obs_out = ((float* const __attribute__((address_space(4)))*)ka->ra.obs)[cur ^ 1];
    const int tid = rl_tidx();
    constexpr bool kSpec = T == 1024;
    int nslots = n0;
    constexpr bool kPlanesEarly = T >= 256;
    RL_MARK(60);
    phase_step<T, true, kPlanesEarly, kSpec>(p, s, w, n0);
    RL_MARK(61);
    assign_order<T>(p, s, nslots);
    if (kPlanesEarly) patch_placed_planes(p, s);
    const int n1 = s.scal[S_N1];
    lds_barrier();
    RL_MARK(62);
    const bool overlapped = !p.limit_reproduction;
    const size_t b = (size_t)w * p.cap;
    auto step_outputs = [&](int t, int nt) {
        for (int k = t; k < n1; k += nt) {
            const int a = s.order[k];
            if (p.so.reward) p.so.reward[b + k] = (float)s.reward[a];
            if (p.so.done) p.so.done[b + k] = (s.flags[a] & RL_F_DEAD) ? 1 : 0;
            if (p.so.src) p.so.src[b + k] = (short)a;
        }
        if (t == 0 && p.so.n_acted) p.so.n_acted[w] = n0;
        if (t == 0 && p.so.acted_total && n0) atomicAdd(p.so.acted_total, (unsigned long long)n0);
    };
    // TRAIN launches with Tracker outputs: Tracker._track_results over the post-step list (environment.py:206-207 -> tracker.py:178-266)
    // runs on wave 1 ALONE, next to wave 0's serial section (_reproduce) and the other waves' state_prime rows: as one more job of a
    // row-writing wave it made that wave the longest of the interval (+4 us per tick).  It reads only what _reproduce leaves alone.
    const bool trk = TRAIN && T >= 256 && p.so.trk_tick != nullptr;   // uniform
    const bool cap = TRAIN == 2 && T >= 256 && *(const int __attribute__((address_space(4)))*)&ka->ra.capture != 0;   // transition capture (uniform)
    int* const cap_slot = (int*)ps.trk_scr;   // (free once the Tracker pass of the same wave is through with it)
    auto track = [&]() {
        if (trk) track_world_wave0(p, s, w, n1, ps.trk_scr, &ps.trk, ticks_done >= *(const int __attribute__((address_space(4)))*)&ka->ra.trk_skip);
        if (cap) capture_reserve(p, s, ka, n1, cap_slot);
    };
    if (overlapped) {
        if (tid < 64) {
            if (!p.static_families) best_agents_wave(s, n1);
            reproduce_wave0<T, true>(p, s, w, n1, nslots);
        } else if (TRAIN && (trk || cap)) {
            if (tid < 128) track();
            else {
                write_observations<(T >= 256 ? T - 128 : 64)>(p, s, w, n1, p.so.obs, tid - 128);
                step_outputs(tid - 128, T - 128);
            }
        } else {
            write_observations<(T > 64 ? T - 64 : 64)>(p, s, w, n1, p.so.obs, tid - 64);
            step_outputs(tid - 64, T - 64);
        }
    } else if (TRAIN && (trk || cap)) {
        if (tid >= 64 && tid < 128) track();
        else {
            const int t = tid < 64 ? tid : tid - 64;
            write_observations<(T >= 256 ? T - 64 : 64)>(p, s, w, n1, p.so.obs, t);
            step_outputs(t, T - 64);
        }
    } else {
        write_observations<T>(p, s, w, n1, p.so.obs);
        step_outputs(tid, T);
    }
    lds_barrier();
    RL_MARK(63);
    if (TRAIN == 2 && cap) {   // the tick's transitions into the replay rings, while the planes still show the post-step grid
        capture_rows<T>(p, s, ka, w, n1, cap_slot, ((float* const __attribute__((address_space(4)))*)ka->ra.obs)[cur]);
        lds_barrier();
    }
    for (int a = tid; a < nslots; a += T) s.src[a] = s.newidx[a];
    if (!overlapped) lds_barrier();
    else { const int first_new = nslots; nslots = s.scal[S_NSLOTS]; patch_planes_after_update<T>(p, s, n1, first_new, nslots); }
    if (!overlapped) phase_update<T, true>(p, s, w, n1, nslots, true);
    // The agent bitmap of the post-update grid AND its prefix (scan_order_wave) by the LAST wave alone, lane l = bitmap word l: all words'
    // loads in one trip, a ballot each, one DPP scan -- a barrier interval less than "every wave its cells, barrier, one wave scans"
    // (the other waves clear the gene table meanwhile).
    if (tid >= T - 64) {
        const int l = tid - (T - 64);
        unsigned long long mine = 0ull;
        for (int wd = 0; wd < p.nW; ++wd) {
            const unsigned long long m = __ballot(s.type[wd * 64 + l] == RL_AGENT);
            if (l == wd) mine = m;
        }
        const int cntw = __popcll(mine);
        const int incl = wave_incl_scan(cntw);
        s.agbits[l] = mine;
        s.wordbase[l] = incl - cntw;
        if (l == 63) s.scal[S_N2] = incl;
    }
    static_assert(T > 64, "k_run: at least two waves");
    if (tid < T - 64)
        for (int i = tid; i < p.hash_size; i += T - 64) { s.hkey[i] = -1; s.hcnt[i] = 0u; }
    lds_barrier();
    RL_MARK(64);
    RL_MARK(65);
    int n2 = s.scal[S_N2];
    int tick_next = s.scal[S_TICK] + 1, epoch_next = s.scal[S_EPOCH];
    int next_uid = s.scal[S_NEXT_UID], max_gene = s.scal[S_MAX_GENE];
    const bool refill = p.refill_threshold >= 0 && n2 < p.refill_threshold;
    if (refill) {
        const bool prepared = kSpec && s.scal[S_SPEC_DONE] != 0;
        const uint32_t new_epoch = (uint32_t)epoch_next + 1u;
        if (prepared) {
            n2 = apply_spec_refill<T>(p, s, w, new_epoch);
            const int np2 = (n2 + 63) & ~63;
            for (int k = tid; k < np2; k += T) hash_insert_wave(s, p.hash_mask, k < n2, k, k < n2 ? s.gene[k] : 0, 1u << 16);
        } else {
            n2 = reset_world_lds<T>(p, s, w, new_epoch);
            rebuild_gene_counts<T>(p, s, n2);
        }
        tick_next = 0; epoch_next = (int)new_epoch; next_uid = n2; max_gene = p.n_brains;
    } else {
        assign_order<T>(p, s, nslots);
        const int nsp = (nslots + 63) & ~63;
        for (int a = tid; a < nsp; a += T) {
            const int aa = a < nslots ? a : 0;
            const int ps_ = s.pos[aa], ge = s.gene[aa];
            const bool on = a < nslots && s.occ[(ps_ & 255) * p.W + (ps_ >> 8)] == a;
            hash_insert_wave(s, p.hash_mask, on, a, on ? ge : 0, 1u << 16);
        }
    }
    RL_MARK(66);
    if (refill || !overlapped || s.scal[S_PLANES_DIRTY]) build_planes<T>(p, s);   // (otherwise patched: patch_planes_after_update)
    lds_barrier();
    RL_MARK(67);
    if (p.uo.src)
        for (int k = tid; k < n2; k += T) p.uo.src[b + k] = refill ? (short)-1 : s.src[s.order[k]];
    const RecycleRegs rr = recycle_read(s, n2);
    {   // wave 0 prepares the next tick's policy (rows grouped by brain) while the others write the Agent.state rows
        if (tid < 64) {
            policy_lists_wave0(p, ps, n2, tid, [&](int k) { return s.brain[s.order[k]]; });
            if (T == 1024 && tid == 0) ps.meta[5] = ps.meta[0] > 4;   // rounds of four tiles read their rows from memory: drain them
            RL_MARK_T(96, 0);
            if constexpr (KIND == kKindAll) policy_schedule_wave0(p, ps, ka, tid);
            RL_MARK_T(97, 0);
        } else {
            // (what the next tick's input layer may assume about these rows -- s.type[0] is what build_planes saw; on a lane of wave 1:
            // wave 0 is the long pole of this interval)
            if constexpr (KIND == kKindAll) { if (tid == 64) ps.oflags[0] = run_obs_flags(p, s, n2, ps.oflags[1]); }
            write_observations<(T > 64 ? T - 64 : 64)>(p, s, w, n2, obs_out, tid - 64, ps.xmirror, ps.xrows);
            RL_MARK_T(98, 64); RL_MARK_T(99, T - 64);
        }
    }
    RL_MARK(68);
    if (tid == 0) { ps.meta[1] = n2; ps.meta[2] = cur ^ 1; ps.meta[3] = ticks_done + 1; }
    // (the mirror holds the rows from the launch's first tick on: written in that tick only -- written every tick, the loop-invariant 0 / 1 was
    // hoisted out of the tick loop as a VGPR, spilled, and reloaded here behind an s_waitcnt vmcnt(0) on wave 0: round 6, found in the ISA)
    if (ticks_done == 0 && tid == 0) ps.meta[4] = ps.xmirror != nullptr;
    // (RL_SEAM_OPEN: the next policy half's Philox keys are in place BEFORE recycle_world's first barrier -- nobody reads the two words in this
    // interval --, and recycle_world writes the same values again)
    if (RL_SEAM_OPEN && tid == 64) { s.scal[S_TICK] = tick_next; s.scal[S_EPOCH] = epoch_next; }
    const bool seam_open = RL_SEAM_OPEN && T == 512 && ticks_done + 1 < *(const int __attribute__((address_space(4)))*)&ka->ra.n_ticks;   // (uniform; the launch's last tick: store_world follows)
    // (rows the policy will read back from memory must have reached L2 first; with every row mirrored in LDS the stores just drain)
    // (capture copies the rows the policy read back from memory a tick later: they must have arrived)
    recycle_world<T, kSpec>(p, s, n2, tick_next, epoch_next, next_uid, max_gene, ps.xmirror == nullptr || n2 > ps.xrows || (TRAIN == 2 && cap), rr,
                            (KIND == kKindAll || T == 1024) ? &ps.meta[5] : nullptr,   // (tiles that take several rounds read their rows from memory)
                            seam_open);
    RL_MARK(69);
}

template <int T, int KIND, int TRAIN>
__device__ __forceinline__ void run_tick_call_fixed(RunParamsC* ka) { run_tick_body<T, true, KIND, TRAIN>(ka); }
template <int T, int KIND, int TRAIN>
__device__ __forceinline__ void run_tick_call_generic(RunParamsC* ka) { run_tick_body<T, false, KIND, TRAIN>(ka); }

template <int T, bool FIXED, int KIND, int TRAIN>
__device__ __forceinline__ void run_load_call(RunParamsC* ka)   // (inlined: load_world reads the kernel-argument segment through the intrinsic)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    typedef const int __attribute__((address_space(4))) cint;
    const KParams p = run_params<FIXED, T>(ka);
    Smem s;
    PolSmem ps;
    run_carve<FIXED, KIND>(p, s, ps, smem_raw, T);
    int n0;
    RL_MARK(90);
    // Everything this call needs from memory is REQUESTED before load_world waits for its own loads -- the first tick's policy input (the
    // Agent.state rows written by the previous launch / the reset, copied into the LDS mirror: fetched row by row by the tile waves
    // they cost the first tick 12.2k cycles against 3.5k from the mirror) and the brains' epilogue constants -- so a launch pays ONE trip
    // to memory, not three in a row (stamps: load_world 7.2k, rows 7.4k, constants 2.9k cycles of a launch's fixed 23k).  Wave 0
    // builds the first tick's row lists meanwhile (4.3k cycles) and stays out of the copying.
    const int tid = rl_tidx();
    const int first = *(cint*)&ka->ra.first;
    const int n_pre = ((cint*)ka->p.st.n_agents)[blockIdx.x];   // (the scalar load load_world makes as well)
    const bool preload = ps.xmirror != nullptr && n_pre <= ps.xrows;
    constexpr int NT = T > 64 ? T - 64 : T;            // copying threads: everybody but wave 0
    const int pt = T > 64 ? tid - 64 : tid;
    typedef const f32x4 __attribute__((address_space(1))) gvec;
    gvec* rows4 = (gvec*)(((float* const __attribute__((address_space(4)))*)ka->ra.obs)[first] + (size_t)blockIdx.x * p.cap * RL_OBS_DIM);
    // (16-byte loads over the world's contiguous rows: a world's block of cap rows starts 16-byte aligned; the last vector may reach into
    // row n0 -- inside the world's block, or the buffer's padding row)
    const int total = n_pre * RL_OBS_DIM, nvec = (total + 3) >> 2;
    constexpr int U = 8, UC = 4;
    f32x4 rv[U];
    float cv[UC];
    if (preload && pt >= 0 && nvec > 0) {
#pragma unroll
        for (int u = 0; u < U; ++u) rv[u] = rows4[min(u * NT + pt, nvec - 1)];
    }
    constexpr int CF = run_const_floats<KIND>();   // per-brain block of epilogue / head constants in LDS (layout: tile_const_src)
    const int ctotal = ps.cconst ? CF * p.n_brains : 0;
    auto const_src = [&](int i) -> gfloat* {
        const int b = i / CF, j = i - CF * b;
        gfloat* pk = (gfloat*)((const float* const __attribute__((address_space(4)))*)ka->ra.packed)[b];
        if constexpr (KIND == kKindAll) {
            const int kind = ((const int __attribute__((address_space(4)))*)ka->ra.kind)[b];
            return pk + tile_const_src(kind, min(j, tile_const_floats(kind) - 1));
        } else
            return pk + tile_const_src(KIND, j);
    };
    if (ctotal > 0) {
#pragma unroll
        for (int u = 0; u < UC; ++u) cv[u] = *const_src(min(u * T + tid, ctotal - 1));
    }
    load_world<T, (T == 1024)>(p, s, (int)blockIdx.x, n0);
    RL_MARK(91);
    auto put_rows = [&](int q, const f32x4& v) {
        if (q < nvec) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int i = 4 * q + e, r = i / RL_OBS_DIM;
                if (i < total) ps.xmirror[r * kXStride + (i - r * RL_OBS_DIM)] = v[e];
            }
        }
    };
    if (preload && pt >= 0 && nvec > 0) {
#pragma unroll
        for (int u = 0; u < U; ++u) put_rows(u * NT + pt, rv[u]);
        for (int base = NT * U; base < nvec; base += NT * U) {   // (worlds above ~90 agents: further batches)
#pragma unroll
            for (int u = 0; u < U; ++u) rv[u] = rows4[min(base + u * NT + pt, nvec - 1)];
#pragma unroll
            for (int u = 0; u < U; ++u) put_rows(base + u * NT + pt, rv[u]);
        }
    }
    if (preload && pt >= 0)   // (the rows' zero padding: write_observations)
        for (int r = pt; r < n_pre; r += NT) {
            float* m = ps.xmirror + r * kXStride + 153;
            m[0] = 0.0f; m[1] = 0.0f; m[2] = 0.0f; *(float4*)(m + 3) = float4{0.0f, 0.0f, 0.0f, 0.0f};
        }
    RL_MARK(92);
    if constexpr (KIND == kKindAll) {   // what is known about the rows this launch starts from (written by an earlier launch from THIS state): the loaded agents' health
        // decides whether any row can exceed [1, 2) -- a world's own rules keep |health| <= 300, a state written by the caller may not
        int bad = 0;
        for (int a = tid; a < n0; a += T) bad |= (s.health[a] >= 400 || s.health[a] <= -400) ? 1 : 0;
        bad = __any(bad);
        if (tid == 0) ps.oflags[1] = 0;
        lds_barrier();
        if (bad && (tid & 63) == 0) ps.oflags[1] = 1;
        lds_barrier();
    }
    if (tid == 0) {
        ps.meta[1] = n0; ps.meta[2] = first; ps.meta[3] = 0; ps.meta[4] = preload ? 1 : 0;
        if constexpr (KIND == kKindAll) ps.oflags[0] = run_obs_flags(p, s, n0, ps.oflags[1]);
    }
    if (TRAIN && p.so.trk_tick) {   // the Tracker's running sums live in LDS for the length of the launch
        const int G = p.static_families ? p.n_brains : 1, t = tid;
        const size_t o = (size_t)blockIdx.x * G * RL_TRK_VARS;
        if (t < G * RL_TRK_VARS) { ps.trk.sum[t] = p.so.trk_sum[o + t]; ps.trk.cnt[t] = p.so.trk_cnt[o + t]; }
        if (t < 2) ps.trk.pop[t] = p.so.trk_pop[(size_t)blockIdx.x * 3 + 1 + t];
    }
    if (tid < 64) {
        if constexpr (KIND == kKindAll) { if (tid < kRunMaxBrains) ps.bkind[tid] = tid < p.n_brains ? ((const int __attribute__((address_space(4)))*)ka->ra.kind)[tid] : RL_D3QN; }
        policy_lists_wave0(p, ps, n0, tid, [&](int k) { return s.brain[k]; });   // (slot == list index after load_world)
        if (T == 1024 && tid == 0) ps.meta[5] = ps.meta[0] > 4;
        if constexpr (KIND == kKindAll) policy_schedule_wave0(p, ps, ka, tid, true);
    }
    RL_MARK(93);
    if (ctotal > 0) {   // the brains' epilogue constants (three 128-wide layers x 256 floats + the heads') for policy_tile1s
#pragma unroll
        for (int u = 0; u < UC; ++u) { const int i = u * T + tid; if (i < ctotal) ps.cconst[i] = cv[u]; }
        for (int base = T * UC; base < ctotal; base += T * UC) {   // (more than two brains)
#pragma unroll
            for (int u = 0; u < UC; ++u) cv[u] = *const_src(min(base + u * T + tid, ctotal - 1));
#pragma unroll
            for (int u = 0; u < UC; ++u) { const int i = base + u * T + tid; if (i < ctotal) ps.cconst[i] = cv[u]; }
        }
    }
    RL_MARK(94);
    lds_barrier();
    RL_MARK(95);
}
template <int T, bool FIXED, int KIND, int TRAIN>
__device__ __forceinline__ void run_store_call(RunParamsC* ka)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const KParams p = run_params<FIXED, T>(ka);
    Smem s;
    PolSmem ps;
    run_carve<FIXED, KIND>(p, s, ps, smem_raw, T);
    store_world<T>(p, s, (int)blockIdx.x, ps.meta[1]);
    if (TRAIN && p.so.trk_tick) {
        const int G = p.static_families ? p.n_brains : 1, t = rl_tidx();
        const size_t o = (size_t)blockIdx.x * G * RL_TRK_VARS;
        if (t < G * RL_TRK_VARS) { p.so.trk_sum[o + t] = ps.trk.sum[t]; p.so.trk_cnt[o + t] = ps.trk.cnt[t]; }
        if (t < 2) p.so.trk_pop[(size_t)blockIdx.x * 3 + 1 + t] = ps.trk.pop[t];
    }
    if (rl_tidx() == 0) { p.st.tick[blockIdx.x] = s.scal[S_TICK]; p.st.epoch[blockIdx.x] = s.scal[S_EPOCH]; p.st.next_uid[blockIdx.x] = s.scal[S_NEXT_UID]; p.st.max_gene[blockIdx.x] = s.scal[S_MAX_GENE]; }
}

// Kernel body = the policy half (inlined: a kernel saves no registers, and the tile code gets the 128-VGPR budget of a
// 1024-thread workgroup to itself); the tick half, the initial load and the final store are callees.
// TRAIN: the launch also serves a training loop (rl_run_ex) -- 1: per-tick exploration rates and the Tracker's statistics; 2: also the
// policy's outputs and the transition capture into the replay rings.  The plain inference launch (0, what bench.py times) carries none of
// that code, and a training loop that does not capture (trainer(): learn() is a no-op in this build) not the capture's.
template <int T, bool FIXED, int KIND, int TRAIN>
__global__ __launch_bounds__(T) void k_run(const RunParams rp)
{
    RunParamsC* ka = (RunParamsC*)__builtin_amdgcn_kernarg_segment_ptr();
    typedef const int __attribute__((address_space(4))) cint;
    const int n_ticks = *(cint*)&ka->ra.n_ticks;
    const int dbg = RL_RUN_DBG(ka);
    // Workgroups that start in exact lockstep stay in lockstep for tens of ticks (every world does the same work at the same moment:
    // 256 CUs ask L2 for the same weight lines, then all write their observation rows), and such ticks are ~3 us slower than those of
    // drifted-apart worlds: kernel time of a 20-tick launch 545 -> 518 us with the starts spread over 3.75 us (16 steps of 0.25 us; 0.5 / 1 us
    // steps, 32 or 64 groups give the same), 100- and 500-tick launches unchanged.  run mask & 32 switches it off (measurements).
#ifdef RL_PHASE_PROFILE
#define RL_KMARK(i) do { long long* pf_ = *(long long* const __attribute__((address_space(4)))*)&ka->p.prof; \
        if (pf_ && (int)blockIdx.x == *(cint*)&ka->p.prof_world && threadIdx.x == 0) RL_G(pf_)[i] = (long long)clock64(); } while (0)
#else
#define RL_KMARK(i) do { } while (0)
#endif
    RL_KMARK(70);
#ifndef RL_STAGGER_BEFORE_LOAD
    // (the delay runs next to the world's load: a workgroup whose load takes longer than its delay does not wait at all)
    const long long t_entry = (long long)clock64();
    run_load_call<T, FIXED, KIND, TRAIN>(ka);
    RL_KMARK(71);
    if (!(dbg & 32)) {
        const long long until = t_entry + (long long)(blockIdx.x & 15) * 500;
        while ((long long)clock64() < until) __builtin_amdgcn_s_sleep(2);
    }
#else
    if (!(dbg & 32)) {
        const long long until = (long long)clock64() + (long long)(blockIdx.x & 15) * 500;
        while ((long long)clock64() < until) __builtin_amdgcn_s_sleep(2);
    }
    RL_KMARK(71);
    run_load_call<T, FIXED, KIND, TRAIN>(ka);
#endif
    RL_KMARK(72);
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    for (int it = 0; it < n_ticks; ++it) {   // (`it` and the bounds are uniform: SGPRs, which a callee preserves)
        if (!(dbg & 1)) run_policy_half<T, FIXED, KIND, TRAIN>(ka, wave);
        if (it == 0) RL_KMARK(73);
        if (!(dbg & 2)) {
            if constexpr (FIXED) run_tick_call_fixed<T, KIND, TRAIN>(ka);
            else run_tick_call_generic<T, KIND, TRAIN>(ka);
        }
        if (it == 0) RL_KMARK(74);
        if (it == 1) RL_KMARK(75);
    }
    RL_KMARK(76);
    run_store_call<T, FIXED, KIND, TRAIN>(ka);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    RL_KMARK(77);
}

}  // namespace

// Two translation units from this one file (build.py; the kernels of the two families compile side by side -- the mixed-kind family alone
// takes as long as everything else in the library): RL_RUN_UNIT 1 holds the kKindAll instantiations and hands their addresses out through
// rl_run_fn_all(); RL_RUN_UNIT 0 holds the dueling-kind instantiations and the host side.  Undefined (tuning one-offs): everything in one unit.
#if defined(RL_RUN_UNIT) && RL_RUN_UNIT == 1
const void* rl_run_fn_all(int fixed, int train)
{
#define RL_RUN_ALL(FX, TR) if (fixed == (FX ? 1 : 0) && train == TR) return (const void*)k_run<512, FX, kKindAll, TR>;
    RL_RUN_ALL(true, 0) RL_RUN_ALL(true, 1) RL_RUN_ALL(true, 2) RL_RUN_ALL(false, 0) RL_RUN_ALL(false, 1) RL_RUN_ALL(false, 2)
#undef RL_RUN_ALL
    return nullptr;
}
#else
#if defined(RL_RUN_UNIT)
const void* rl_run_fn_all(int fixed, int train);
#endif

// rl_run: which kernel serves these brains -- RL_PERD3QN (all of a dueling kind), kKindAll (any mix), or -1
static int run_kind_of(const rl_brain* brains, int n_brains)
{
    if (n_brains < 1 || n_brains > kRunMaxBrains) return -1;
    bool duel = true;
    for (int b = 0; b < n_brains; ++b) {
        if (brains[b].kind < RL_DQN || brains[b].kind > RL_PPO) return -1;
        duel = duel && (brains[b].kind == RL_D3QN || brains[b].kind == RL_PERD3QN);
    }
    return duel ? RL_PERD3QN : kKindAll;   // (D3QN and PERD3QN are the same network: PERD3QN.py:186-202, D3QN.py:149-165)
}
// LDS of a launch with planes of row stride `stride`; host_plane_stride picks the stride: padded where it fits, else row-major
template <int KIND>
static size_t run_smem_bytes(const rl_world* h, int T, int stride, int* xrows = nullptr, size_t* xbytes = nullptr)
{
    PolSmem ps;
    const size_t world = rl_world_smem_bytes(h->cpad, h->cfg.slot_cap, h->hash_size, stride, h->cfg.height);
    const size_t b = carve_policy<KIND>(ps, nullptr, world, h->cfg.slot_cap, run_mirror_budget(T), run_cbrains(T, h->cfg.n_brains), run_tile_floats<KIND>(T), run_min_region(T));
    if (xrows) *xrows = ps.xrows;
    if (xbytes) *xbytes = (size_t)ps.xbytes;
    return b;
}
template <int KIND>
static int host_plane_stride(const rl_world* h, int T)
{
    const int padded = run_plane_stride(T, h->cfg.width, h->cfg.height);
    return (padded != h->cfg.width && run_smem_bytes<KIND>(h, T, padded) > 160 * 1024) ? h->cfg.width : padded;
}
// Workgroup size of the multi-tick kernel: 512 threads for few worlds (the one-wave policy tile needs the 256-VGPR budget; the
// tick half alone would prefer 1024: 10.6 vs 13.1 us at 256 worlds), 256 when there are many worlds (several per CU).
// The world_block option (rl_set_option / RL_WORLD_BLOCK at rl_create) overrides it like for the other world kernels.
static int run_block(const rl_world* h)
{
    const int forced = h->opt.world_block;
    if (forced == 256 || forced == 512 || forced == 1024) return forced;
    if (h->cfg.n_worlds <= 768) return 512;
#ifdef RL_RUN_256
    return 256;
#else
    // the product library holds no k_run<256> (slower than the two-launch loop wherever it would be chosen): beyond 768 worlds the answer is
    // "not supported" -- callers loop over rl_policy_act + rl_tick_refill -- unless the run_always option asks for the single launch, which
    // then runs the 512-thread kernel with several workgroups per CU taking turns (8.9e8 against 9.7e8 agent-steps/s at 1,024 worlds)
    return h->opt.run_always ? 512 : 256;
#endif
}
int rl_world_run_supported(const rl_world* h, const rl_brain* brains, int n_brains)
{
    const int kind = run_kind_of(brains, n_brains);
    if (kind < 0) return 0;
    const int T = run_block(h);
    if (h->cfg.slot_cap > T) return 0;
#ifndef RL_RUN_1024
    // k_run<1024> (sixteen waves per world, four-wave policy tiles: policy_quad) is bit-identical to the 512-thread kernel and 4.4 us per
    // tick slower: it is compiled into the tuning builds only (-DRL_RUN_1024); callers fall back to the two-launch loop
    if (T == 1024) { rl_set_error("rl_run: 1024-thread workgroups are a tuning-build instantiation (-DRL_RUN_1024), not in this library"); return 0; }
#endif
#ifndef RL_RUN_256
    // k_run<256> (several worlds per CU) loses to the two-launch loop wherever it would be chosen (6.5e8 against 9.7e8 agent-steps/s at 1,024
    // worlds): tuning builds only (-DRL_RUN_256)
    if (T == 256) { rl_set_error("rl_run: 256-thread workgroups are a tuning-build instantiation (-DRL_RUN_256), not in this library"); return 0; }
#endif
    if (kind == kKindAll) {
        // the mixed-kind kernel exists for 512-thread workgroups; the tiles' exchange buffers (32 KB for a PPO tile) lie in the mirror
        if (T != 512) return 0;
        int xrows = 0;
        if (run_smem_bytes<kKindAll>(h, T, host_plane_stride<kKindAll>(h, T), &xrows) > 160 * 1024) return 0;
        return (size_t)xrows * kXStride * sizeof(float) >= (size_t)pair_ex_bytes(RL_PPO);   // (at least one tile per round)
    }
    int xrows = 0;
    size_t xbytes = 0;
    if (run_smem_bytes<RL_PERD3QN>(h, T, host_plane_stride<RL_PERD3QN>(h, T), &xrows, &xbytes) > 160 * 1024) return 0;
    // 1024 threads: the four-wave tiles' exchange slices (24 KB each) lie in the mirror's region
    if (T == 1024 && xbytes < 4 * (size_t)kQuadExBytes) {
        rl_set_error("rl_run: 1024-thread workgroups need %zu bytes of LDS for the tiles' exchange slices, this world leaves %zu (%d mirror rows)",
                     4 * (size_t)kQuadExBytes, xbytes, xrows);
        return 0;
    }
    return 1;
}
int rl_world_launch_run(rl_world* h, const rl_brain* brains, int n_brains, int n_ticks, int8_t* actions, const rl_step_out* so,
                        float* const obs[2], int first, int16_t* upd_src, int refill_threshold, int refill_n_agents,
                        int32_t* refill_count, const float* eps_sched, int eps_on_host, int trk_skip, const rl_replay* replays, float* policy_out, hipStream_t st)
{
    if (!rl_world_run_supported(h, brains, n_brains)) { rl_set_error("rl_run: unsupported configuration (brain kinds / slot_cap / LDS)"); return RL_E_UNSUPPORTED; }
    KParams p = make_params(h);
    set_list_production(h, p, false);   // the row lists describe the state BEFORE this launch
    p.actions = actions;
    if (so) p.so = *so;
    p.uo.src = upd_src; p.uo.obs = obs[first ^ 1];
    p.refill_threshold = refill_threshold; p.reset_n_agents = refill_n_agents; p.refill_count = refill_count;
    RunParams rp{};
    rp.p = p;
    RunArgs& ra = rp.ra;
    for (int b = 0; b < n_brains; ++b) { ra.packed[b] = brains[b].packed; ra.eps[b] = brains[b].epsilon; ra.kind[b] = brains[b].kind; }
    ra.obs[0] = obs[0]; ra.obs[1] = obs[1]; ra.first = first; ra.n_ticks = n_ticks; ra.actions = actions; ra.trk_skip = trk_skip;
    if (eps_on_host) {   // (checked by the caller: a host table of at most RL_EPS_INLINE_MAX floats) -- it rides in the kernel arguments
        ra.eps_sched = nullptr; ra.eps_inline_on = 1;
        for (int i = 0; i < n_ticks * n_brains; ++i) ra.eps_inline[i] = eps_sched[i];
    } else { ra.eps_sched = eps_sched; ra.eps_inline_on = 0; }
    ra.capture = replays != nullptr; ra.policy_out = policy_out;
    if (replays) for (int b = 0; b < n_brains; ++b) ra.rp[b] = replays[b];
#ifdef RL_TUNING
    ra.debug = g_run_debug;
    if (g_run_debug) rl_set_error("rl_run: measurement mask %d is set (rl_debug_set_run_mask): the results of this launch are not valid", g_run_debug);
#endif
    const int T = run_block(h);
    const int kind = run_kind_of(brains, n_brains);
    rp.p.PS = kind == kKindAll ? host_plane_stride<kKindAll>(h, T) : host_plane_stride<RL_PERD3QN>(h, T);
    const size_t bytes = kind == kKindAll ? run_smem_bytes<kKindAll>(h, T, rp.p.PS) : run_smem_bytes<RL_PERD3QN>(h, T, rp.p.PS);
    const bool fixed = p.W == kFixW && p.H == kFixH && rp.p.PS == run_plane_stride(T, kFixW, kFixH) && p.cap == kFixCap && p.hash_size == kFixHash && !h->opt.world_generic;
    const int train = (replays != nullptr || policy_out != nullptr) ? 2 : (eps_sched != nullptr || p.so.trk_tick != nullptr) ? 1 : 0;   // (eps_sched: device table or host table, either way a schedule)
    const void* fn = nullptr;
#define RL_RUN_PICK(TT, FX, KD, TR) if (T == TT && fixed == FX && kind == KD && train == TR) fn = (const void*)k_run<TT, FX, KD, TR>;
#ifdef RL_RUN_DEV_BUILD   /* tuning builds: only the instantiations bench.py times (compile time) */
    RL_RUN_PICK(512, true, RL_PERD3QN, 0)
#ifdef RL_RUN_DEV_ALL
    RL_RUN_PICK(512, true, kKindAll, 0)
#endif
#else
#define RL_RUN_PICK3(TT, FX, KD) RL_RUN_PICK(TT, FX, KD, 0) RL_RUN_PICK(TT, FX, KD, 1) RL_RUN_PICK(TT, FX, KD, 2)
#ifdef RL_RUN_1024
    RL_RUN_PICK3(1024, true, RL_PERD3QN) RL_RUN_PICK3(1024, false, RL_PERD3QN)
#endif
    RL_RUN_PICK3(512, true, RL_PERD3QN) RL_RUN_PICK3(512, false, RL_PERD3QN)
#ifdef RL_RUN_256
    RL_RUN_PICK3(256, true, RL_PERD3QN) RL_RUN_PICK3(256, false, RL_PERD3QN)
#endif
#if defined(RL_RUN_UNIT)
    if (T == 512 && kind == kKindAll) fn = rl_run_fn_all(fixed ? 1 : 0, train);   // (the other translation unit)
#else
    RL_RUN_PICK3(512, true, kKindAll) RL_RUN_PICK3(512, false, kKindAll)
#endif
#undef RL_RUN_PICK3
#endif
#undef RL_RUN_PICK
    if (!fn) { rl_set_error("rl_run: no kernel instantiation for T=%d fixed=%d kind=%d train=%d in this build", T, (int)fixed, kind, train); return RL_E_UNSUPPORTED; }
    if (bytes > 64 * 1024) {   // opt in to the large dynamic-LDS window: once per (kernel, device, size) -- the attribute belongs to the device's copy of the kernel
        // (a process may alternate several instantiations -- bench.py's TRAIN 0 line and trainer()'s TRAIN 1: each keeps its grant)
        // (process-wide table behind a mutex: handles are independent of each other, include/reinlife_hip.h, and may launch from several threads)
        struct Grant { const void* fn; size_t bytes; };
        static Grant granted[64][8];
        static std::mutex granted_mu;
        const std::lock_guard<std::mutex> hold(granted_mu);
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
        Grant* g = nullptr;
        for (int i = 0; i < 8 && !g; ++i) if (granted[dev][i].fn == fn || granted[dev][i].fn == nullptr) g = &granted[dev][i];
        if (!g) { g = &granted[dev][0]; g->fn = nullptr; g->bytes = 0; }   // (more than eight instantiations in one process: reuse a slot)
        if (g->fn != fn || g->bytes < bytes) {
            const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
            if (e != hipSuccess) { rl_set_error("hipFuncSetAttribute(%zu bytes of LDS) failed: %s", bytes, hipGetErrorString(e)); return RL_E_LAUNCH; }
            g->fn = fn; g->bytes = bytes;
        }
    }
    void* kargs[] = {(void*)&rp};
    const hipError_t le = hipLaunchKernel(fn, dim3(h->cfg.n_worlds), dim3(T), kargs, bytes, st);
    if (le != hipSuccess) { rl_set_error("run kernel launch failed: %s", hipGetErrorString(le)); return RL_E_LAUNCH; }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { rl_set_error("run kernel launch failed: %s", hipGetErrorString(e)); return RL_E_LAUNCH; }
    return RL_OK;
}
#endif   // RL_RUN_UNIT != 1
