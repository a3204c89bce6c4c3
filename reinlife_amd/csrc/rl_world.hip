// rl_world.hip -- the ReinLife world tick on MI355X (gfx950), hand-written HIP.
//
// One workgroup per world.  The world (H*W type grid + row-major agent list, ~6 KB) is staged into LDS once, the
// whole tick runs out of LDS, and the new row-major list + both observation passes are streamed back to HBM.
//
// Reference semantics (paths under /root/reference/ReinLife; the sequential restatement is oracle/rl_oracle.c):
//   step()        World/environment.py:160-186   _act :258-275, _attack :652-699, _prepare_movement :591-625,
//                 _execute_movement :627-650, _eat :701-715, _update_agent_position :778-782,
//                 _update_death_status :789-793, _get_rewards :277-311, _add_food :763-776
//   update_env()  World/environment.py:188-215   _update_best_agents :728-739, _reproduce :488-519, _produce :521-547,
//                 _remove_dead_agents :795-799
//   observation   World/environment.py:313-456 + Grid.fov World/grid.py:90-117
//
// The reference resolves everything sequentially in agent (row-major cell) order.  Here each phase is a closed form
// evaluated by one lane per agent, ordered only by comparing cell indices:
//   attack    final health from the set of successful attackers among the 4 neighbours and own success
//   movement  Jacobi fixed point on an LDS target-count grid (one __syncthreads_or per iteration)
//   vanish    a mover entering the cell of a later-ordered mover is erased (sequential grid overwrite)
//   ordering  Grid.get_entities == rank of the agent's cell in a 64-bit-per-wave ballot bitmap (popcount prefix)
//   set_random  k-th empty cell == select on the ballot bitmap of occupied cells, held in wave 0's registers; all births of
//               a tick at once (ranks in the original list of empty cells by a recurrence, cells selected in parallel)
//
// Kernel variants (k_world<T, MODE, LEAN, FIXED>): LEAN = the fused inference tick (wave 0 runs the update's serial section
// next to the other waves' observation pass, Philox draws precomputed by idle waves, a refill prepared ahead of time on
// idle waves); FIXED = LEAN with the reference's default world shape (30x30, 100 agents) folded into constants.  A launch
// lasts as long as its slowest world, so the tail of the per-world time matters as much as its mean.
#include <stdlib.h>

#include "rl_common.h"

// The thread index as the multi-tick kernel sees it: opaque, so that nothing derived from it is loop-invariant.  k_run runs
// policy and tick back to back inside a tick loop; with the plain builtin every per-thread constant of the tick phases (window
// offsets, row bases, ...) was hoisted out of that loop and kept alive -- i.e. spilled -- across the 126-VGPR tile code.
__device__ inline int rl_tidx() { int t = (int)threadIdx.x; asm volatile("" : "+v"(t)); return t; }
// The lane index WITHOUT the thread index: at the head of the policy half the thread index has been spilled (the tile code takes all 256
// VGPRs), and its reload is a memory round trip that also waits for the wave's observation-row stores.  Opaque for the same reason as above.
__device__ inline int rl_lane_fresh()
{
    int l = 0;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
#endif
    return l;
}

int rl_world_prepare_bytes(size_t bytes);
size_t rl_world_smem_bytes(int cpad, int cap, int hash);
static int g_ablate = 0;  // tuning only
extern "C" void rl_debug_set_ablate(int mask) { g_ablate = mask; }

namespace {

constexpr int kSuper = RL_SUPER_FOOD;
constexpr uint8_t kPadCell = 0xFF;  // grid padding up to a multiple of 64 cells: neither empty nor anything else

// scalar slots in LDS
enum { S_ALIVE = 0, S_NFOOD, S_NPOISON, S_NSUPER, S_NSLOTS, S_N1, S_N2, S_NPARENTS, S_NELIG, S_BESTK, S_ERR, S_TICK, S_EPOCH,
       S_NEXT_UID, S_MAX_GENE, S_ANYFLAG0, S_ANYFLAG1, S_NPLACED, S_SPEC_NF, S_SPEC_NP, S_SPEC_DONE, S_PLANES_DIRTY, S_COUNT = 24 };

struct KParams {
    int W, H, C, Cp, nW;
    int cap, max_agents, n_brains, hash_size, hash_mask, world_base;
    int static_families, limit_reproduction, incentivize_killing;
    uint64_t seed;
    rl_state st;
    const int8_t* actions;
    rl_tape tape;
    rl_step_out so;
    rl_update_out uo;
    float* obs_only;
    int32_t* err;
    int reset_n_agents, refill_threshold;
    int32_t* refill_count;
    int split_food;        // MODE_STEP only: stop before _add_food and report the cell counts its draws depend on
    int32_t* pre_counts;   // [R][4] food, poison, super food, empty cells after movement (split_food)
    int* lists_counts;       // optional: per-brain row-list counters of this launch's parity (policy work buffer)
    int* lists_counts_zero;  //           the other parity, cleared by block 0 for the next producer
    int* lists;              //           row ids (world*cap + k), [n_brains][list_stride]
    long long list_stride;
    int ablate;       // tuning only (env RL_ABLATE): bit mask of sections to skip -- results are then WRONG
    long long* prof;  // optional: shader-clock stamps of world prof_world's phases (debug / tuning)
    int prof_world;
};

struct Smem {
    unsigned long long* occbits;  // [64] non-empty cells
    unsigned long long* agbits;   // [64] agent cells
    int* wordbase;                // [64] exclusive prefix of popc(agbits)
    int* scal;                    // [S_COUNT]
    int* best_uid;                // [16]
    int* best_brain;              // [16]
    double* best_fit;             // [16]
    double* wred_f;               // [16] per-wave argmax
    int* wred_k;                  // [16]
    int* present;                 // [RL_MAX_BRAINS]
    int* hkey;                    // [hash]
    unsigned* hcnt;               // [hash] low 16: alive, high 16: on grid
    float* foodv;                 // [Cp]  (aliased: unsigned target counts during movement)
    float* healthv;               // [Cp]
    int* genev;                   // [Cp]
    short* occ;                   // [Cp]
    uint8_t* type;                // [Cp]
    int *health, *age, *max_age, *gene, *brain, *uid;  // [cap]
    double *fitness, *reward, *trk_rew;                // [cap]
    unsigned short *pos, *tgt, *hslot;                 // [cap]
    short *newidx, *order, *src, *plist;               // [cap]
    unsigned short* spec;         // [Cp]  a refill generated ahead of time: cell type | gene << 8 (spec_refill_stage)
    unsigned long long* spec_agbits;  // [64] its agent cells ...
    int* spec_wordbase;               // [64] ... and their exclusive prefix per bitmap word
    uint8_t *flags, *aux;                              // [cap]
    signed char* action;                               // [cap]
};

enum { AUX_VANISH = 1, AUX_PARENT = 2 };

#ifdef RL_PHASE_PROFILE
#define RL_ABL(bit) (p.ablate & (bit))  /* tuning build only: skip a section (results are then WRONG) */
#define RL_MARK(i) do { if (p.prof && (int)blockIdx.x == p.prof_world && rl_tidx() == 0) RL_G(p.prof)[i] = (long long)__builtin_readcyclecounter(); } while (0) /* global (not flat) store: stays off lgkmcnt */
#define RL_MARK_T(i, t) do { if (p.prof && (int)blockIdx.x == p.prof_world && rl_tidx() == (t)) RL_G(p.prof)[i] = (long long)__builtin_readcyclecounter(); } while (0)
#define RL_MARK_W(base) do { if (p.prof && (int)blockIdx.x == p.prof_world && (rl_tidx() & 63) == 0) RL_G(p.prof)[(base) + (rl_tidx() >> 6)] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define RL_MARK_W(base) do { } while (0)
#define RL_ABL(bit) 0
#define RL_MARK(i) do { } while (0)
#define RL_MARK_T(i, t) do { } while (0)
#endif

__host__ __device__ inline size_t align16(size_t x) { return (x + 15) & ~(size_t)15; }

__host__ __device__ inline size_t carve(Smem& s, char* base, int Cp, int cap, int hash)
{
    size_t o = 0;
#define CARVE(field, type, count) s.field = (type*)(base + o); o = align16(o + sizeof(type) * (size_t)(count));
    CARVE(occbits, unsigned long long, 64)
    CARVE(agbits, unsigned long long, 64)
    CARVE(spec_agbits, unsigned long long, 64)
    CARVE(best_fit, double, 16)
    CARVE(wred_f, double, 16)
    CARVE(fitness, double, cap)
    CARVE(reward, double, cap)
    CARVE(trk_rew, double, cap)
    CARVE(wordbase, int, 64)
    CARVE(spec_wordbase, int, 64)
    CARVE(scal, int, S_COUNT)
    CARVE(best_uid, int, 16)
    CARVE(best_brain, int, 16)
    CARVE(wred_k, int, 16)
    CARVE(present, int, RL_MAX_BRAINS)
    CARVE(hkey, int, hash)
    CARVE(hcnt, unsigned, hash)
    CARVE(foodv, float, Cp)
    CARVE(healthv, float, Cp)
    CARVE(genev, int, Cp)
    CARVE(health, int, cap)
    CARVE(age, int, cap)
    CARVE(max_age, int, cap)
    CARVE(gene, int, cap)
    CARVE(brain, int, cap)
    CARVE(uid, int, cap)
    CARVE(occ, short, Cp)
    CARVE(pos, unsigned short, cap)
    CARVE(tgt, unsigned short, cap)
    CARVE(hslot, unsigned short, cap)
    CARVE(newidx, short, cap)
    CARVE(order, short, cap)
    CARVE(src, short, cap)
    CARVE(plist, short, cap)
    CARVE(spec, unsigned short, Cp)
    CARVE(type, uint8_t, Cp)
    CARVE(flags, uint8_t, cap)
    CARVE(aux, uint8_t, cap)
    CARVE(action, signed char, cap)
#undef CARVE
    return o;
}

// ---------------------------------------------------------------------------------------------------------------
// small device helpers
// ---------------------------------------------------------------------------------------------------------------
__device__ inline int lane_id() { return rl_tidx() & 63; }

// The kernel argument block (KParams is the only kernel parameter, so it starts at offset 0 of the kernarg segment), made
// opaque so that every use site re-reads the few pointers it needs with s_load instead of keeping all ~45 pointers alive
// from kernel entry to the final store (which spilled >150 SGPRs into VGPR lanes).
// The block is read through the CONSTANT address space (scalar loads, also inside divergent code) and the pointers found
// in it are used through the GLOBAL address space (RL_G): as generic pointers they became flat_load / flat_store, which
// are counted on lgkmcnt as well -- every LDS wait and every lds_barrier() then also waited for HBM traffic.
struct KParams;
typedef const KParams __attribute__((address_space(4))) KParamsC;
__device__ inline KParamsC* kernargs()
{
    KParamsC* q = (KParamsC*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(q));
    return q;
}
template <typename P> struct rl_global_ptr;
template <typename E> struct rl_global_ptr<E*> { typedef E __attribute__((address_space(1)))* type; };
#define RL_G(ptr) ((typename rl_global_ptr<decltype(ptr)>::type)(ptr))

// Workgroup barrier for LDS-only communication.  lds_barrier() is a full workgroup fence: it emits
// s_waitcnt vmcnt(0), which on gfx950 also waits for every outstanding global STORE (observation rows, outputs) -- an
// HBM write round trip (~1 us) at each of the ~40 barriers of a tick.  Threads of these kernels only ever exchange data
// through LDS, so waiting for the LDS queue is sufficient; global stores drain in the background.
#define RL_HAVE_LDS_BARRIER 1
#ifdef RL_FULL_FENCE
__device__ inline void lds_barrier() { __syncthreads(); }
#else
__device__ inline void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
#endif

// block-wide OR over LDS flags (replaces __syncthreads_or, which carries the same full fence); `flags` is int[2] in LDS,
// zero-initialised, `phase` a per-thread register toggled identically by all threads
__device__ inline bool block_any(int* flags, int& phase, bool pred)
{
    if (pred) flags[phase] = 1;
    lds_barrier();
    const bool r = flags[phase] != 0;
    phase ^= 1;
    if (rl_tidx() == 0) flags[phase] = 0;  // next use of this slot is after at least one more barrier
    return r;
}
__device__ inline unsigned long long shfl_u64(unsigned long long v, int src)
{
    unsigned lo = __shfl((unsigned)v, src), hi = __shfl((unsigned)(v >> 32), src);
    return ((unsigned long long)hi << 32) | lo;
}
__device__ inline double shfl_f64(double v, int src) { return __longlong_as_double((long long)shfl_u64((unsigned long long)__double_as_longlong(v), src)); }
__device__ inline double shfl_xor_f64(double v, int m)
{
    unsigned long long u = (unsigned long long)__double_as_longlong(v);
    unsigned lo = __shfl_xor((unsigned)u, m), hi = __shfl_xor((unsigned)(u >> 32), m);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
// inclusive prefix sum over the 64 lanes with DPP (no LDS traffic: __shfl_up lowers to ds_bpermute, ~100 cycles each):
// Kogge-Stone inside each row of 16 lanes, then row_bcast:15 / row_bcast:31 carry the row totals across rows.
__device__ inline int wave_incl_scan(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);  // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);  // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true);  // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true);  // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, true);  // row_bcast:15 -> rows 1,3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, true);  // row_bcast:31 -> rows 2,3
    return v;
}
__device__ inline int read_lane(int v, int l) { return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(l)); }
__device__ inline unsigned long long read_lane_u64(unsigned long long v, int l)
{
    const int sl = __builtin_amdgcn_readfirstlane(l);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, sl);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), sl);
    return ((unsigned long long)hi << 32) | lo;
}
__device__ inline unsigned long long lowmask(int b) { return b ? (~0ull >> (64 - b)) : 0ull; }

// toroidal neighbour of cell (i,j): up 0 (i-1), right 1 (j+1), down 2 (i+1), left 3 (j-1)   World/utils.py:4-17
__device__ inline int neighbour_cell(int i, int j, int d, int W, int H)
{
    int ni = i, nj = j;
    if (d == 0) ni = (i == 0) ? H - 1 : i - 1;
    else if (d == 1) nj = (j == W - 1) ? 0 : j + 1;
    else if (d == 2) ni = (i == H - 1) ? 0 : i + 1;
    else nj = (j == 0) ? W - 1 : j - 1;
    return ni * W + nj;
}

// k-th empty cell of Grid.set_random (World/grid.py:69-83) as a select on wave 0's occupancy bitmap.
// The per-lane inclusive prefix of empty-cell counts is computed once (DPP scan) and then maintained incrementally:
// a placement in word L just decrements the prefix of lanes >= L.
struct Placer {
    unsigned long long word;  // lane l: cells 64l..64l+63, bit set = not empty
    int incl;                 // empty cells in words 0..l
    int n_empty;
};
__device__ inline void placer_init(Placer& P, unsigned long long word)
{
    P.word = word;
    P.incl = wave_incl_scan(__popcll(~word));
    P.n_empty = __builtin_amdgcn_readlane(P.incl, 63);
}
__device__ inline int placer_take(Placer& P, int k)
{
    const int l = lane_id();
    const unsigned long long m = __ballot(k < P.incl);
    const int L = __ffsll((long long)m) - 1;
    const unsigned long long z = ~read_lane_u64(P.word, L);
    const int kk = k - (read_lane(P.incl, L) - __popcll(z));
    const bool set = (z >> l) & 1ull;
    const int rank = __popcll(z & lowmask(l));
    const unsigned long long hit = __ballot(set && rank == kk);
    const int bit = __ffsll((long long)hit) - 1;
    if (l == L) P.word |= 1ull << bit;
    if (l >= L) P.incl -= 1;
    P.n_empty -= 1;
    return L * 64 + bit;
}

__device__ inline void flag_error(const KParams& p, Smem& s, int code, int w, int d0, int d1)
{
    if (p.err && atomicCAS(p.err, 0, code) == 0) { p.err[1] = w; p.err[2] = d0; p.err[3] = d1; }
    s.scal[S_ERR] = code;
}

// ---------------------------------------------------------------------------------------------------------------
// Speculative refill.  A launch lasts as long as its slowest world, and with the refill-below-threshold rule a dozen of
// the 256 worlds regenerate themselves in every tick: ~8,000 cycles of generator (one Philox block per cell + a counting
// sort, nine barrier intervals) on top of a full tick.  But the new world depends only on (seed, epoch + 1, world), not on
// the state -- so a world whose population is close to the threshold lets its IDLE waves (2..15; the agents live on waves 0
// and 1) run the generator -- the Philox blocks and the counting sort in five existing barrier intervals of the step phase --
// into scratch that is free there
// (keys = the gene plane, sorted keys = the health plane, bucket counters = the reward array, cleared by load_world), and
// parks the result in s.spec[cell] = type | gene << 8.  If the refill then happens, apply_spec_refill only unpacks it
// (two barrier intervals, no Philox); if the world was not prepared, the in-line generator runs as before.
// Same rule, same arithmetic as reset_world_lds: both are checked against the oracle.
// ---------------------------------------------------------------------------------------------------------------
constexpr int kSpecFirst = 64;    // first thread that takes part: 960 threads, one cell each for a 30x30 grid (wave 0 runs the serial sections)
constexpr int kSpecMargin = 20;   // worlds with fewer than threshold + margin agents prepare a refill
__device__ inline bool spec_refill_wanted(const KParams& p, int n0, int& lg)
{
    lg = 6;
    while ((2 << lg) <= p.Cp) ++lg;  // NB = largest power of two <= Cp (reset_world_lds)
    return p.refill_threshold >= 0 && n0 < p.refill_threshold + kSpecMargin && n0 <= 128 && (1 << lg) <= 2 * p.cap;
}
struct SpecState {
    bool on;     // this world prepares a refill in this tick (uniform per workgroup)
    int lg;      // log2 of the bucket count
    int nf, np;  // this WAVE's food / poison coins (spec_refill_keys)
};
// The Philox block of every cell (key, coins, gene).
template <int T>
__device__ inline void spec_refill_keys(const KParams& p, Smem& s, int w, uint32_t epoch, SpecState& st)
{
    const int tid = rl_tidx();
    st.nf = st.np = 0;
    if (tid < kSpecFirst) return;
    const int sp = tid - kSpecFirst;
    constexpr int NS = T - kSpecFirst;
    unsigned* keys = (unsigned*)s.genev;
    for (int c = sp; c < p.Cp; c += NS) {   // (whole waves per iteration: NS and Cp are multiples of 64)
        bool f = false, q = false;
        if (c < p.C) {
            const rl_u4 r = rl_philox4x32(p.seed, epoch, (uint32_t)(p.world_base + w), 0u, RL_SITE_RESET_AGENT, (uint32_t)c);
            keys[c] = (r.x & ~0xFFFu) | (unsigned)c;
            f = rl_u24(r.z) < 0.1; q = rl_u24(r.w) < 0.05;
            s.spec[c] = (unsigned short)(rl_mulhi(r.y, (unsigned)p.n_brains) << 8);  // the gene, should the cell get an agent
        }
        st.nf += __popcll(__ballot(f)); st.np += __popcll(__ballot(q));
    }
}
// stage 0: histogram, 1: scan (one wave), 2: scatter, 3: rank + classify, 4: agent bitmap + prefix (one wave).  One barrier
// between stages.
template <int T>
__device__ inline void spec_refill_stage(const KParams& p, Smem& s, int w, const SpecState& st, int stage)
{
    const int tid = rl_tidx();
    const int lg = st.lg;
    if (tid < kSpecFirst) return;
    const int sp = tid - kSpecFirst;
    constexpr int NS = T - kSpecFirst;
    unsigned* keys = (unsigned*)s.genev;
    unsigned* cum = (unsigned*)s.reward;
    unsigned* sorted = (unsigned*)s.healthv;
    const int NB = 1 << lg, sh = 32 - lg;
    if (stage == 0) {   // histogram of the key prefixes + the coin counts of spec_refill_keys
        for (int c = sp; c < p.C; c += NS) atomicAdd(&cum[keys[c] >> sh], 1u);
        if (lane_id() == 0) {
            if (st.nf) atomicAdd(&s.scal[S_SPEC_NF], st.nf);
            if (st.np) atomicAdd(&s.scal[S_SPEC_NP], st.np);
        }
    } else if (stage == 1) {
        if (sp < 64) {  // exclusive scan of the NB bucket counts by one wave: lane l owns NB/64 consecutive buckets
            const int per = NB >> 6, b0 = sp * per;
            unsigned local = 0;
            for (int i = 0; i < per; ++i) local += cum[b0 + i];
            unsigned base = (unsigned)(wave_incl_scan((int)local) - (int)local);
            for (int i = 0; i < per; ++i) { const unsigned c = cum[b0 + i]; cum[b0 + i] = base; base += c; }
        }
    } else if (stage == 2) {
        for (int c = sp; c < p.C; c += NS) {
            const unsigned key = keys[c];
            sorted[atomicAdd(&cum[key >> sh], 1u)] = key;  // afterwards cum[b] = END of bucket b
        }
    } else if (stage == 3) {
        const int na = min(p.reset_n_agents, p.C);
        const int k1 = na, k2 = na + s.scal[S_SPEC_NF], k3 = k2 + s.scal[S_SPEC_NP];
        for (int c = sp; c < p.Cp; c += NS) {
            unsigned v = kPadCell;
            if (c < p.C) {
                const unsigned key = keys[c];
                const unsigned b = key >> sh;
                const int start = b ? (int)cum[b - 1] : 0, end = (int)cum[b];
                int rank = start;
                for (int j = start; j < end; ++j) rank += sorted[j] < key;
                const unsigned t = rank < k1 ? RL_AGENT : rank < k2 ? RL_FOOD : rank < k3 ? RL_POISON : rank == k3 ? (unsigned)kSuper : RL_EMPTY;
                v = t | (s.spec[c] & 0xFF00u);
            }
            s.spec[c] = (unsigned short)v;
        }
    } else if (stage == 4) {   // agent bitmap of the prepared world, scanned by the wave that holds it
        if (sp < 64) {
            unsigned long long mine = 0ull;
            for (int wd = 0; wd < p.nW; ++wd) {
                const unsigned long long m = __ballot((s.spec[wd * 64 + sp] & 0xFFu) == RL_AGENT);
                if (sp == wd) mine = m;
            }
            const int cntw = __popcll(mine);
            s.spec_agbits[sp] = mine;
            s.spec_wordbase[sp] = wave_incl_scan(cntw) - cntw;
            if (sp == 0) s.scal[S_SPEC_DONE] = 1;
        }
    }
}

__device__ inline void init_newborn(Smem& s, int idx, int cell, int W, int gene, int brain, int uid);

// The prepared world replaces the old one: what reset_world_lds leaves behind, from s.spec.  Returns the agent count.
template <int T>
__device__ __forceinline__ int apply_spec_refill(const KParams& p, Smem& s, int w, uint32_t epoch)
{
    const int tid = rl_tidx();
    lds_barrier();
    if (tid < S_COUNT) s.scal[tid] = 0;
    // one pass: the bitmap and its prefix were prepared too, and a cell's occ / type are written by its own thread only
    for (int c = tid; c < p.Cp; c += T) {
        const unsigned v = s.spec[c];
        const unsigned t = v & 0xFFu;
        s.type[c] = (uint8_t)t;
        s.occ[c] = -1;
        if (t == RL_AGENT) {
            const int idx = s.spec_wordbase[c >> 6] + __popcll(s.spec_agbits[c >> 6] & lowmask(c & 63));
            const int gene = (int)(v >> 8);
            init_newborn(s, idx, c, p.W, gene, gene, idx);
            s.order[idx] = (short)idx; s.newidx[idx] = (short)idx;
        }
    }
    if (tid < 64) { s.agbits[tid] = s.spec_agbits[tid]; s.wordbase[tid] = s.spec_wordbase[tid]; }
    for (int i = tid; i < p.hash_size; i += T) { s.hkey[i] = -1; s.hcnt[i] = 0u; }  // the gene table of the new world starts empty
    const int na = min(p.reset_n_agents, p.C);
    if (tid < RL_N_BEST) {
        s.best_uid[tid] = -1; s.best_fit[tid] = 0.0; s.best_brain[tid] = 0;
        p.st.best_uid[(size_t)w * RL_N_BEST + tid] = -1;
        p.st.best_fit[(size_t)w * RL_N_BEST + tid] = 0.0;
        p.st.best_brain[(size_t)w * RL_N_BEST + tid] = 0;
    }
    if (tid == 0) {
        p.st.next_uid[w] = na; p.st.max_gene[w] = p.n_brains; p.st.tick[w] = 0; p.st.epoch[w] = (int)epoch;
        if (p.refill_count) atomicAdd(p.refill_count, 1);
    }
    lds_barrier();
    return na;
}

// ---------------------------------------------------------------------------------------------------------------
// phases
// ---------------------------------------------------------------------------------------------------------------
template <int T, bool SPEC = false>
__device__ __forceinline__ void load_world(const KParams& p, Smem& s, int w, int& n0)
{
    const int tid = rl_tidx();
    KParamsC* q = kernargs();
    const size_t b = (size_t)w * p.cap;
    // ---- every pointer first, in uniform code (one batch of scalar loads) ...
    const auto gt = RL_G(q->st.cell_type) + (size_t)w * p.C;
    const auto g_i = RL_G(q->st.a_i) + b, g_j = RL_G(q->st.a_j) + b, g_fl = RL_G(q->st.a_flags) + b;
    const auto g_h = RL_G(q->st.a_health) + b, g_age = RL_G(q->st.a_age) + b, g_ma = RL_G(q->st.a_max_age) + b;
    const auto g_g = RL_G(q->st.a_gene) + b, g_b = RL_G(q->st.a_brain) + b, g_u = RL_G(q->st.a_uid) + b;
    const auto g_f = RL_G(q->st.a_fitness) + b;
    const auto g_act = (q->actions ? RL_G(q->actions) : RL_G((const int8_t*)q->st.a_action)) + b;
    const auto g_bu = RL_G(q->st.best_uid) + (size_t)w * RL_N_BEST, g_bb = RL_G(q->st.best_brain) + (size_t)w * RL_N_BEST;
    const auto g_bf = RL_G(q->st.best_fit) + (size_t)w * RL_N_BEST;
    // ---- ... the per-world scalars through the scalar cache (uniform addresses: no lane select of POINTERS, no readfirstlane) ...
    typedef const int32_t __attribute__((address_space(4))) cint;
    n0 = ((cint*)q->st.n_agents)[w];
    const int v_tick = ((cint*)q->st.tick)[w], v_epoch = ((cint*)q->st.epoch)[w];
    const int v_uid = ((cint*)q->st.next_uid)[w], v_mg = ((cint*)q->st.max_gene)[w];
    // ---- ... then EVERY vector load (one HBM round trip): nothing below depends on a loaded value until the LDS writes.
    // Slot `tid` of the agent arrays is read unconditionally (inside the allocation; ignored beyond n_agents).
    int bu = -1, bb = 0; double bf = 0.0;
    if (tid < RL_N_BEST) { bu = g_bu[tid]; bf = g_bf[tid]; bb = g_bb[tid]; }
    uint8_t ty[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int c = tid + u * T; ty[u] = c < p.C ? gt[c] : kPadCell; }
    // unconditional (index clamped into the allocation, not predicated): a predicated block attracts the first uses of
    // the loaded values -- and with them a wait -- into itself
    const bool ha = tid < p.cap;
    const int ti = ha ? tid : p.cap - 1;
    // (waves whose slots all lie beyond the allocation skip the agent loads -- a wave-uniform branch: their eleven load
    // instructions would only queue in front of the useful ones of the last waves)
    uint8_t r_i = 0, r_j = 0, r_fl = 0;
    signed char r_act = 0;
    int r_h = 0, r_age = 0, r_ma = 0, r_g = 0, r_b = 0, r_u = 0;
    double r_f = 0.0;
    if (__builtin_amdgcn_readfirstlane(tid) < p.cap) {
        r_i = g_i[ti]; r_j = g_j[ti]; r_fl = g_fl[ti];
        r_act = g_act[ti];
        r_h = g_h[ti]; r_age = g_age[ti]; r_ma = g_ma[ti]; r_g = g_g[ti]; r_b = g_b[ti]; r_u = g_u[ti];
        r_f = g_f[ti];
    }
    const int sc_val = tid == S_TICK ? v_tick : tid == S_EPOCH ? v_epoch : tid == S_NEXT_UID ? v_uid : tid == S_MAX_GENE ? v_mg : 0;
    __builtin_amdgcn_sched_barrier(0);  // keep every use of a loaded value below the LDS initialisation (the compiler hoisted
                                        // a shift of r_j up here, i.e. a wait for the loads right after issuing them)
    RL_MARK(30);
    // ---- LDS initialisation that needs no loaded value
    for (int c = tid; c < p.Cp; c += T) { s.occ[c] = -1; ((unsigned*)s.foodv)[c] = 0u; }
    for (int i = tid; i < p.hash_size; i += T) { s.hkey[i] = -1; s.hcnt[i] = 0u; }
    if (tid < RL_MAX_BRAINS) s.present[tid] = 0;
    if (SPEC) for (int i = tid; i < 2 * p.cap; i += T) ((unsigned*)s.reward)[i] = 0u;  // bucket counters of a speculative refill
    __builtin_amdgcn_sched_barrier(0);
    RL_MARK(31);
    // ---- consume the loads
    if (tid < S_COUNT) s.scal[tid] = tid == S_NSLOTS ? n0 : sc_val;
    if (tid < RL_N_BEST) { s.best_uid[tid] = bu; s.best_fit[tid] = bf; s.best_brain[tid] = bb; }
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int c = tid + u * T; if (c < p.Cp) s.type[c] = ty[u]; }
    for (int c = tid + 4 * T; c < p.Cp; c += T) s.type[c] = c < p.C ? gt[c] : kPadCell;
    RL_MARK(32);
    auto put = [&](int a, int pi, int pj, int h, int age, int ma, int g, int br, int u, int fl, int act, double f) {
        s.pos[a] = (unsigned short)(pi | (pj << 8));
        s.health[a] = h; s.age[a] = age; s.max_age[a] = ma; s.gene[a] = g; s.brain[a] = br; s.uid[a] = u;
        s.flags[a] = (uint8_t)fl; s.action[a] = (signed char)act; s.fitness[a] = f;
        s.aux[a] = 0; s.src[a] = (short)a; s.order[a] = (short)a; s.newidx[a] = (short)a;
    };
    if (ha && tid < n0) put(tid, r_i, r_j, r_h, r_age, r_ma, r_g, r_b, r_u, r_fl, r_act, r_f);
    for (int a = tid + T; a < n0; a += T)
        put(a, g_i[a], g_j[a], g_h[a], g_age[a], g_ma[a], g_g[a], g_b[a], g_u[a], g_fl[a], g_act[a], g_f[a]);
    RL_MARK(33);
    lds_barrier();
    RL_MARK(34);
    for (int a = tid; a < n0; a += T) s.occ[(s.pos[a] & 255) * p.W + (s.pos[a] >> 8)] = (short)a;
    lds_barrier();
}

// gene -> (alive count, on-grid count) open-addressing table; every agent remembers its slot.  Lanes of a wave that
// share a gene are aggregated first (one CAS + one add per distinct gene per wave instead of per agent): with a
// handful of families every agent would otherwise hammer the same two LDS words.
// Must be called by all 64 lanes of the wave; `active` lanes contribute.
__device__ inline void hash_insert_wave(Smem& s, int mask, bool active, int a, int gene, unsigned add)
{
    unsigned long long pending = __ballot(active);
    while (pending) {
        const int leader = __builtin_amdgcn_readfirstlane(__ffsll((long long)pending) - 1);
        const int g = __builtin_amdgcn_readlane(gene, leader);
        const bool mine = active && gene == g;
        const unsigned long long m = __ballot(mine);
        const unsigned lo = (unsigned)__popcll(__ballot(mine && (add & 1u)));
        const unsigned hi = (unsigned)__popcll(__ballot(mine && (add >> 16)));
        int h = 0;
        if (lane_id() == leader) {
            unsigned hh = ((unsigned)g * 2654435761u) & (unsigned)mask;
            for (;;) {
                const int old = atomicCAS(&s.hkey[hh], -1, g);
                if (old == -1 || old == g) break;
                hh = (hh + 1) & (unsigned)mask;
            }
            if (lo | hi) atomicAdd(&s.hcnt[hh], lo | (hi << 16));
            h = (int)hh;
        }
        h = __builtin_amdgcn_readlane(h, leader);
        if (mine) s.hslot[a] = (unsigned short)h;
        pending &= ~m;
    }
}

// Grid.get_entities order of the current grid: newidx[a] / order[k] for all slots, returns count via scal[slot]
template <int T>
__device__ __forceinline__ void build_order(const KParams& p, Smem& s, int nslots, int out_slot)
{
    const int tid = rl_tidx();
    for (int c = tid; c < p.Cp; c += T) {
        const unsigned long long m = __ballot(s.type[c] == RL_AGENT);
        if (lane_id() == 0) s.agbits[c >> 6] = m;
    }
    lds_barrier();
    if (tid < 64) {
        const int cntw = tid < p.nW ? __popcll(s.agbits[tid]) : 0;
        const int incl = wave_incl_scan(cntw);
        s.wordbase[tid] = incl - cntw;
        if (tid == 63) s.scal[out_slot] = incl;
    }
    lds_barrier();
    for (int a = tid; a < nslots; a += T) {
        const int cell = (s.pos[a] & 255) * p.W + (s.pos[a] >> 8);
        const int oc = s.occ[cell], wb = s.wordbase[cell >> 6];  // one batch (the list is a handful of agents per lane:
        const unsigned long long ab = s.agbits[cell >> 6];       // the chain's latency is what counts)
        short ni = -1;
        if (oc == a) {
            ni = (short)(wb + __popcll(ab & lowmask(cell & 63)));
            s.order[ni] = (short)a;
        }
        s.newidx[a] = ni;
    }
    lds_barrier();
}

// Third part of build_order alone (agbits / wordbase / scal[slot] already there); no trailing barrier
template <int T>
__device__ __forceinline__ void assign_order(const KParams& p, Smem& s, int nslots)
{
    for (int a = rl_tidx(); a < nslots; a += T) {
        const int cell = (s.pos[a] & 255) * p.W + (s.pos[a] >> 8);
        const int oc = s.occ[cell], wb = s.wordbase[cell >> 6];  // one batch (the list is a handful of agents per lane:
        const unsigned long long ab = s.agbits[cell >> 6];       // the chain's latency is what counts)
        short ni = -1;
        if (oc == a) {
            ni = (short)(wb + __popcll(ab & lowmask(cell & 63)));
            s.order[ni] = (short)a;
        }
        s.newidx[a] = ni;
    }
}
// Second part of build_order by ONE wave (lane l = bitmap word l): exclusive prefix of the agent counts
__device__ inline void scan_order_wave(const KParams& p, Smem& s, int lane, int out_slot)
{
    const int cntw = lane < p.nW ? __popcll(s.agbits[lane]) : 0;
    const int incl = wave_incl_scan(cntw);
    s.wordbase[lane] = incl - cntw;
    if (lane == 63) s.scal[out_slot] = incl;
}

// _prepare_observations (environment.py:377-404) into LDS planes
template <int T>
__device__ __forceinline__ void build_planes(const KParams& p, Smem& s, int t0 = rl_tidx(), int nt = T)
{
    if (RL_ABL(2)) return;
    const bool float_mode = s.type[0] == RL_AGENT;  // np.vectorize dtype inference from cell (0,0)
    for (int c = t0; c < p.C; c += nt) {
        const int t = s.type[c];
        float f = 0.f, h = -1.f;
        int g = -2;
        if (t == RL_FOOD) f = 0.5f;
        else if (t == kSuper) f = 1.f;
        else if (t == RL_POISON) f = -1.f;
        else if (t == RL_AGENT) {
            const int a = s.occ[c];
            const int hp = s.health[a];
            if (hp < 0) f = 1.f;                                   // _get_food, environment.py:440-444
            // hp / 200.0 as one f64 multiply: (float)(hp * 0.005) == (float)(hp / 200.0) and trunc() of both agree for every
            // integer |hp| <= 1e5 (checked exhaustively; health stays within [-300, 200]); an f64 division is ~20 instructions
            const double v = (double)hp * 0.005;
            h = float_mode ? (float)v : (float)(double)(long long)v;  // astype(int64) truncates toward zero
            if (s.flags[a] & RL_F_DEAD) g = s.gene[a];             // _get_genes, environment.py:448-456
        }
        s.foodv[c] = f; s.healthv[c] = h; s.genev[c] = g;
    }
}

// _get_observations (environment.py:349-375): n agents in order[] -> obs rows (coalesced 49-float runs).  Executed by NT
// threads, `t` = this thread's index among them (the fused tick lets wave 0 run _reproduce meanwhile).
// `mirror` (optional, LDS): the same rows as float32 [xrows][kXStride] for the in-workgroup policy of k_run (rows >= xrows only
// go to HBM).
constexpr int kXStride = 164;   // floats per mirrored row (656 B: 16-byte aligned rows, spread over the banks)
template <int NT>
__device__ __forceinline__ void write_observations(const KParams& p, Smem& s, int w, int n, float* obs, int t, float* mirror = nullptr, int xrows = 0)
{
    if (!obs || RL_ABL(1)) return;
    float* base = obs + (size_t)w * p.cap * RL_OBS_DIM;
    // thread t owns window cell idx = t % 49 for agents t / 49, t / 49 + NT / 49, ...: the divisions and the window offsets
    // are computed once per thread, not once per (agent, cell) item
    constexpr int G = NT / 49;
    const int g0 = t / 49, idx = t - g0 * 49;
    const int dr = idx / 7 - 3, dc = idx - (idx / 7) * 7 - 3;
    if (g0 < G) {
        // U agents per trip through the dependent LDS chain order -> (position, gene) -> the three planes.  Small workgroups
        // (256 threads: 3 agents per pass, ~28 passes) gain from it (4096 worlds: 147 -> 141 us); with 960 writer threads a
        // pass covers 19 agents and one agent at a time is faster (256 worlds: 20.55 vs 20.85 us), so U = 1 there.
        constexpr int U = NT >= 900 ? 1 : 5;
        float* o = base + (size_t)g0 * RL_OBS_DIM + idx;
        if (U == 1) {
            for (int k = g0; k < n; k += G, o += (size_t)G * RL_OBS_DIM) {
                const int a = s.order[k];
                const int pa = s.pos[a];
                int ci = (pa & 255) + dr, cj = (pa >> 8) + dc;
                ci += ci < 0 ? p.H : 0; ci -= ci >= p.H ? p.H : 0;
                cj += cj < 0 ? p.W : 0; cj -= cj >= p.W ? p.W : 0;
                const int c = ci * p.W + cj;
                const int g = s.genev[c];
                const float vf = s.foodv[c], vh = s.healthv[c], vg = g == -2 ? 0.f : (g == s.gene[a] ? 1.f : -1.f);  // _extract_gene_observation, environment.py:424-428
                o[0] = vf;
                o[49] = vh;
                o[98] = vg;
                if (mirror && k < xrows) { float* m = mirror + k * kXStride + idx; m[0] = vf; m[49] = vh; m[98] = vg; }
            }
        } else
        for (int k = g0; k < n; k += U * G, o += (size_t)(U * G) * RL_OBS_DIM) {
            int a[U], pa[U], ga[U], c[U], g[U];
            float f[U], h[U];
            bool ok[U];
#pragma unroll
            for (int u = 0; u < U; ++u) { ok[u] = k + u * G < n; a[u] = s.order[ok[u] ? k + u * G : k]; }
#pragma unroll
            for (int u = 0; u < U; ++u) { pa[u] = s.pos[a[u]]; ga[u] = s.gene[a[u]]; }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                int ci = (pa[u] & 255) + dr, cj = (pa[u] >> 8) + dc;
                ci += ci < 0 ? p.H : 0; ci -= ci >= p.H ? p.H : 0;
                cj += cj < 0 ? p.W : 0; cj -= cj >= p.W ? p.W : 0;
                c[u] = ci * p.W + cj;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) { g[u] = s.genev[c[u]]; f[u] = s.foodv[c[u]]; h[u] = s.healthv[c[u]]; }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (ok[u]) {
                    float* ou = o + (size_t)(u * G) * RL_OBS_DIM;
                    const float vg = g[u] == -2 ? 0.f : (g[u] == ga[u] ? 1.f : -1.f);  // _extract_gene_observation, environment.py:424-428
                    ou[0] = f[u];
                    ou[49] = h[u];
                    ou[98] = vg;
                    if (mirror && k + u * G < xrows) { float* m = mirror + (k + u * G) * kXStride + idx; m[0] = f[u]; m[49] = h[u]; m[98] = vg; }
                }
        }
    }
    for (int k = t; k < n; k += NT) {
        const int a = s.order[k];
        const int same = (int)(s.hcnt[s.hslot[a]] >> 16);
        float* o = base + (size_t)k * RL_OBS_DIM + 147;
        const float v0 = (float)((double)s.health[a] * 0.005);  // == (float)(health / 200.0), see build_planes
        const float v1 = (s.flags[a] & RL_F_REPRODUCED) ? 1.f : 0.f;
        const float v2 = (float)((double)same / (double)n);
        const float v3 = (float)((double)n / (double)p.max_agents);
        const float v4 = (s.flags[a] & RL_F_KILLED) ? 1.f : 0.f;
        const float v5 = (s.flags[a] & RL_F_ATE_SUPER) ? 1.f : -1.f;
        o[0] = v0; o[1] = v1; o[2] = v2; o[3] = v3; o[4] = v4; o[5] = v5;
        if (mirror && k < xrows) { float* m = mirror + k * kXStride + 147; m[0] = v0; m[1] = v1; m[2] = v2; m[3] = v3; m[4] = v4; m[5] = v5; }
    }
}
template <int T>
__device__ inline void write_observations(const KParams& p, Smem& s, int w, int n, float* obs, float* mirror = nullptr, int xrows = 0)
{
    write_observations<T>(p, s, w, n, obs, rl_tidx(), mirror, xrows);
}

// Lean tick: every Philox draw of the tick depends only on (seed, epoch, world, tick, site, index), so the idle upper half of
// the workgroup computes them during the first agent phase and parks them in LDS (the tracker's scratch, unused in the lean
// kernel); the serial sections on wave 0 (_add_food, _reproduce, _produce) then just read them.
// layout (32-bit words): [0..6] food x, [8..14] food y, [16,17] produce x,y, [32..95] birth draws 0..63, [128..) gate draws
struct DrawCache {
    unsigned* w;
    int n_gate;  // gate draws cached for ranks < n_gate
};
__device__ inline DrawCache draw_cache(const KParams& p, Smem& s)
{
    DrawCache c;
    c.w = (unsigned*)s.trk_rew;
    c.n_gate = max(0, min(p.cap, 2 * p.cap - 128));
    return c;
}
template <int T>
__device__ inline void precompute_draws(const KParams& p, Smem& s, int w, int n0)
{
    // One Philox block per item and ONE call site: item -> (site, index) first.  (Separate calls per kind made a wave that
    // held items of three kinds run three blocks back to back -- the longest path of the first agent phase.)  Gate ranks are
    // positions among the eligible agents of the post-step list, so only ranks < n0 can be asked for.  Items are dealt
    // from the top thread down: the low waves carry the agents.
    const DrawCache c = draw_cache(p, s);
    const uint32_t epoch = (uint32_t)s.scal[S_EPOCH], tick = (uint32_t)s.scal[S_TICK], world = (uint32_t)(p.world_base + w);
    constexpr int kFirst = T > 128 ? 128 : T / 2;  // threads below stay out of it
    const int n_items = 128 + min(c.n_gate, n0);
    if (rl_tidx() < kFirst) return;
    for (int item = T - 1 - rl_tidx(); item < n_items; item += T - kFirst) {
        uint32_t site, idx;
        if (item < RL_FOOD_TRIES) { site = RL_SITE_FOOD; idx = (uint32_t)item; }
        else if (item == 16) { site = RL_SITE_PRODUCE; idx = 0u; }
        else if (item >= 32 && item < 96) { site = RL_SITE_BIRTH; idx = (uint32_t)(item - 32); }
        else if (item >= 128) { site = RL_SITE_REPRO; idx = (uint32_t)(item - 128); }
        else continue;
        const rl_u4 r = rl_philox4x32(p.seed, epoch, world, tick, site, idx);
        c.w[item] = r.x;
        if (item < RL_FOOD_TRIES) c.w[8 + item] = r.y;
        else if (item == 16) c.w[17] = r.y;
    }
}

// Environment.step up to (not including) the observation pass
template <int T, bool LEAN, bool PLANES_EARLY, bool SPEC = false>
__device__ __forceinline__ void phase_step(const KParams& p, Smem& s, int w, int n0)
{
    const int tid = rl_tidx();
    const int W = p.W, H = p.H;
    SpecState spec_state;
    spec_state.on = SPEC && spec_refill_wanted(p, n0, spec_state.lg);
    const bool spec = spec_state.on;  // uniform per workgroup
    unsigned* cnt = (unsigned*)s.foodv;
    // ---- _act prologue + _attack (closed form) + _prepare_movement ------------------------------------------------
    for (int a = tid; a < n0; a += T) {
        const int i = s.pos[a] & 255, j = s.pos[a] >> 8, cx = i * W + j;
        const int act = s.action[a];
        const int fl = s.flags[a] & ~(RL_F_KILLED | RL_F_INTER_KILLED | RL_F_INTRA_KILLED);
        const bool dead = fl & RL_F_DEAD;
        const int h0 = min(200, s.health[a] - 10);
        s.age[a] = min(s.max_age[a], s.age[a] + 1);
        // The four neighbours are looked up with UNCONDITIONAL loads in two batches (who stands there; then its flags,
        // action and gene): written with && chains the compiler keeps every load behind its guard -- twelve dependent LDS
        // round trips per agent instead of two (this phase: 2,640 -> 1,740 cycles).
        int nc[4], y[4], yfl[4], yact[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) nc[d] = neighbour_cell(i, j, d, W, H);
#pragma unroll
        for (int d = 0; d < 4; ++d) y[d] = s.occ[nc[d]];
        const int tdir = (act >= 4 && act <= 7) ? act - 4 : 0;
        const int t = tdir == 0 ? y[0] : tdir == 1 ? y[1] : tdir == 2 ? y[2] : y[3];
#pragma unroll
        for (int d = 0; d < 4; ++d) { const int yc = max(y[d], 0); yfl[d] = s.flags[yc]; yact[d] = s.action[yc]; }
        const int tgene = s.gene[max(t, 0)], mygene = s.gene[a];
        // own attack: succeeds iff the adjacent cell holds an agent (environment.py:692)
        bool own = false;
        int nfl = fl;
        if (!dead && act >= 4 && act <= 7 && t >= 0) {
            own = true;
            nfl |= RL_F_KILLED | (tgene == mygene ? RL_F_INTER_KILLED : RL_F_INTRA_KILLED);
        }
        // successful attackers of this agent: the neighbour in direction d attacking in direction d^2
        int zmax = -1;
#pragma unroll
        for (int d = 0; d < 4; ++d)
            if (y[d] >= 0 && !(yfl[d] & RL_F_DEAD) && yact[d] == 4 + (d ^ 2)) zmax = max(zmax, nc[d]);
        int h;
        if (own) h = zmax > cx ? 0 : (zmax >= 0 ? 100 : min(200, h0 + 100));
        else h = zmax >= 0 ? 0 : h0;
        s.health[a] = h;
        s.flags[a] = (uint8_t)nfl;
        const int tg = (!dead && act >= 0 && act <= 3) ? neighbour_cell(i, j, act, W, H) : cx;
        s.tgt[a] = (unsigned short)tg;
        atomicAdd(&cnt[tg], 1u);
    }
    // (the cells' Philox blocks run here, next to the first agent phase of waves 0 and 1, not in load_world's wait for HBM:
    // the world's loads come back from L2 / MALL in ~800 cycles, the blocks take ~1,300 -- measured 21.0 vs 21.2 us)
    if (SPEC && spec) { spec_refill_keys<T>(p, s, w, (uint32_t)s.scal[S_EPOCH] + 1u, spec_state); spec_refill_stage<T>(p, s, w, spec_state, 0); }
    lds_barrier();
    RL_MARK(2);
    // ---- _execute_movement: Jacobi fixed point (environment.py:637-644) --------------------------------------------
    int any_phase = 0;
    for (; !RL_ABL(16);) {
        int conflict = 0;
        for (int a = tid; a < n0; a += T) {
            const int cx = (s.pos[a] & 255) * W + (s.pos[a] >> 8);
            const int tg = s.tgt[a];
            const unsigned ct = cnt[tg];  // unconditional: behind `tg != cx &&` it would be a third dependent LDS trip
            const bool c = tg != cx && ct > 1u;
            s.aux[a] = c ? 0x80 : 0;
            conflict |= c;
        }
        if (!block_any(&s.scal[S_ANYFLAG0], any_phase, conflict != 0)) break;
        for (int a = tid; a < n0; a += T)
            if (s.aux[a] & 0x80) {
                const int cx = (s.pos[a] & 255) * W + (s.pos[a] >> 8);
                atomicSub(&cnt[s.tgt[a]], 1u);
                atomicAdd(&cnt[cx], 1u);
                s.tgt[a] = (unsigned short)cx;
            }
        lds_barrier();
    }
    RL_MARK(3);
    // ---- _eat + vanish rule (reads the pre-move grid) ----------------------------------------------------------------
    for (int a = tid; a < n0; a += T) {
        const int cx = (s.pos[a] & 255) * W + (s.pos[a] >> 8);
        const int tg = s.tgt[a], act = s.action[a];
        // everything the rule may need, in one batch (guarded loads would each be a dependent LDS trip)
        const int tt = s.type[tg], oc = s.occ[tg], hp = s.health[a], ma = s.max_age[a], fl = s.flags[a];
        uint8_t ax = 0;
        if (act >= 0 && act <= 3) {
            if (tt == RL_FOOD) s.health[a] = min(200, hp + 40);
            else if (tt == RL_POISON) s.health[a] = min(200, hp - 40);
            else if (tt == kSuper) {
                s.health[a] = min(200, hp + 40);
                s.max_age[a] = (int)((double)ma * 1.2);
                s.flags[a] = (uint8_t)(fl | RL_F_ATE_SUPER);
            }
            // entering the cell of a later-ordered agent that is itself leaving: erased by its grid[old]=Empty
            if (tg != cx && oc >= 0 && tg > cx) ax = AUX_VANISH;
        }
        s.aux[a] = ax;
    }
    // (the tick's draws: first needed by _add_food below.  Not in the first agent phase, where a preparing world's idle
    // waves are busy with their cells' Philox blocks)
    if (LEAN) precompute_draws<T>(p, s, w, n0);
    if (SPEC && spec) spec_refill_stage<T>(p, s, w, spec_state, 1);
    lds_barrier();
    RL_MARK(35);
    for (int a = tid; a < n0; a += T) {
        const int cx = (s.pos[a] & 255) * W + (s.pos[a] >> 8);
        if (s.tgt[a] != cx) { s.type[cx] = RL_EMPTY; s.occ[cx] = -1; }
    }
    if (SPEC && spec) spec_refill_stage<T>(p, s, w, spec_state, 2);
    lds_barrier();
    RL_MARK(36);
    // (per-wave counts are taken with ballots and added by one lane: a per-lane atomicAdd on one LDS word is turned by the
    // compiler into a scalar loop over the active lanes, ~60 cycles per lane -- 2 us for a full wave)
    int alive_wave = 0;
    const int n0p = (n0 + 63) & ~63;  // whole waves take part in the gene aggregation
    for (int a = tid; a < n0p; a += T) {
        const bool valid = a < n0;
        unsigned alive = 0u, ongrid = 0u;
        int gene = 0;
        if (valid) {
            const int cx = (s.pos[a] & 255) * W + (s.pos[a] >> 8);
            const int tg = s.tgt[a];
            if (tg != cx) {
                if (!(s.aux[a] & AUX_VANISH)) { s.type[tg] = RL_AGENT; s.occ[tg] = (short)a; }
                const int ti = tg / W;
                s.pos[a] = (unsigned short)(ti | ((tg - ti * W) << 8));
            }
            // _update_death_status (environment.py:789-793)
            int fl = s.flags[a];
            const int hp = s.health[a], ag = s.age[a], ma = s.max_age[a];  // one batch
            if (hp <= 0 || ag == ma) fl |= RL_F_DEAD;
            s.flags[a] = (uint8_t)fl;
            alive = (fl & RL_F_DEAD) ? 0u : 1u;
            ongrid = (s.aux[a] & AUX_VANISH) ? 0u : 1u;
            gene = s.gene[a];
        }
        alive_wave += __popcll(__ballot(alive != 0u));
        hash_insert_wave(s, p.hash_mask, valid, a, gene, alive | (ongrid << 16));
    }
    RL_MARK(37);
    if (alive_wave && lane_id() == 0) atomicAdd(&s.scal[S_ALIVE], alive_wave);
    if (SPEC && spec) spec_refill_stage<T>(p, s, w, spec_state, 3);
    lds_barrier();
    RL_MARK(4);
    // ---- _get_rewards over the _act list incl. vanished agents (environment.py:291-311) ------------------------------
    const int alive = s.scal[S_ALIVE];
    for (int a = tid; a < n0; a += T) {
        const int fl = s.flags[a];
        const double fit = s.fitness[a];
        const int kin = max(0, (int)(s.hcnt[s.hslot[a]] & 0xFFFFu) - 1);
        double r;
        if (fl & RL_F_DEAD) r = (double)(kin - alive);
        else if (alive == 1) r = 0.0;
        else r = (double)kin / (double)alive;
        if ((fl & RL_F_KILLED) && p.incentivize_killing) r += 0.2;
        s.reward[a] = r;
        s.fitness[a] = fit + r;
        if (!p.static_families && s.uid[a] >= 0)  // best_agents are references: their fitness tracks the live agent
            for (int b = 0; b < RL_N_BEST; ++b)
                if (s.best_uid[b] == s.uid[a]) s.best_fit[b] = s.fitness[a];
    }
    RL_MARK(5);
    // ---- _add_food (environment.py:763-776) ---------------------------------------------------------------------------
    // (the agent bitmap of the post-step ordering is taken in the same sweep: food placement does not touch agent cells,
    // so the ordering's prefix scan can run on wave 1 next to the placement on wave 0)
    int nf = 0, np_ = 0, ns = 0;  // per wave (Cp is a multiple of 64: whole waves run each iteration)
    for (int c = tid; c < p.Cp; c += T) {
        const int t = s.type[c];
        nf += __popcll(__ballot(t == RL_FOOD)); np_ += __popcll(__ballot(t == RL_POISON)); ns += __popcll(__ballot(t == kSuper));
        const unsigned long long m = __ballot(t != RL_EMPTY);
        const unsigned long long ma = __ballot(t == RL_AGENT);
        if (lane_id() == 0) { s.occbits[c >> 6] = m; s.agbits[c >> 6] = ma; }
    }
    if (lane_id() == 0) {
        if (nf) atomicAdd(&s.scal[S_NFOOD], nf);
        if (np_) atomicAdd(&s.scal[S_NPOISON], np_);
        if (ns) atomicAdd(&s.scal[S_NSUPER], ns);
    }
    if (SPEC && spec) spec_refill_stage<T>(p, s, w, spec_state, 4);
    lds_barrier();
    RL_MARK(6);
    if (tid >= 64 && tid < 128) scan_order_wave(p, s, tid - 64, S_N1);
    if (!LEAN && p.split_food) {  // seed-compatible stepping: the host needs these counts to draw exactly like _add_food
        if (tid < 64) {
            int ne = tid < p.nW ? __popcll(~s.occbits[tid]) : 0;
            ne = __builtin_amdgcn_readlane(wave_incl_scan(ne), 63);
            if (tid == 0 && p.pre_counts) {
                int32_t* o = p.pre_counts + (size_t)w * 4;
                o[0] = s.scal[S_NFOOD]; o[1] = s.scal[S_NPOISON]; o[2] = s.scal[S_NSUPER]; o[3] = ne;
            }
        }
        lds_barrier();
        return;
    }
    if (tid == 0) s.scal[S_NPLACED] = 0;
    if (tid < 64 && !RL_ABL(4)) {
        int nplaced = 0;
        Placer P;
        placer_init(P, tid < p.nW ? s.occbits[tid] : ~0ull);
        const bool tape = !LEAN && p.tape.food_k != nullptr;
        unsigned xk = 0; double u = 2.0;
        if (tid < RL_FOOD_TRIES) {
            if (tape) { xk = (unsigned)p.tape.food_k[(size_t)w * RL_FOOD_TRIES + tid]; u = p.tape.food_u[(size_t)w * RL_FOOD_TRIES + tid]; }
            else if (LEAN) { const DrawCache c = draw_cache(p, s); xk = c.w[tid]; u = rl_u24(c.w[8 + tid]); }
            else {
                const rl_u4 r = rl_philox4x32(p.seed, (uint32_t)s.scal[S_EPOCH], (uint32_t)(p.world_base + w), (uint32_t)s.scal[S_TICK], RL_SITE_FOOD, (uint32_t)tid);
                xk = r.x; u = rl_u24(r.y);
            }
        }
        const bool en_food = (double)s.scal[S_NFOOD] <= (double)p.C / 10.0;
        const bool en_poison = (double)s.scal[S_NPOISON] <= (double)p.C / 20.0;
        const bool en_super = s.scal[S_NSUPER] == 0;
        // a try whose coin fails changes nothing (its two draws are simply consumed), so only the placing tries are
        // walked, in order; all coins are evaluated in parallel (lane t = try t)
        const bool en = tid < 3 ? en_food : (tid < 6 ? en_poison : en_super);
        const bool places = tid < RL_FOOD_TRIES && en && u < (tid < 6 ? 0.2 : 1.0);
        unsigned long long todo = __ballot(places);
        while (todo) {
            const int t = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            if (P.n_empty <= 0) break;  // full grid: randint raises, nothing is placed (grid.py:82-83)
            const unsigned x = (unsigned)read_lane((int)xk, t);
            const int k = tape ? (int)x : (int)rl_mulhi(x, (unsigned)P.n_empty);
            if (k < 0 || k >= P.n_empty) { if (tid == 0) flag_error(p, s, 1, w, t, k); continue; }
            const int cell = placer_take(P, k);
            if (tid == 0) {
                s.type[cell] = (uint8_t)(t < 3 ? RL_FOOD : (t < 6 ? RL_POISON : kSuper));
                s.plist[nplaced] = (short)cell;  // (plist is free until the update: remembered for the planes' patch)
            }
            ++nplaced;
        }
        if (tid < p.nW) s.occbits[tid] = P.word;
        if (tid == 0) s.scal[S_NPLACED] = nplaced;
    } else if (PLANES_EARLY && tid >= 128) {
        // waves 2.. build the observation planes of the post-step grid meanwhile, as if nothing were placed; the (at most
        // seven) placed cells are patched in the next interval (patch_placed_planes)
        build_planes<T>(p, s, tid - 128, T - 128);
    }
    lds_barrier();
}

// the planes of the cells _add_food just filled (they were built as empty cells next to the placement)
__device__ inline void patch_placed_planes(Smem& s)
{
    const int tid = rl_tidx();
    if (tid < s.scal[S_NPLACED]) {
        const int c = s.plist[tid];
        const int t = s.type[c];
        s.foodv[c] = t == RL_FOOD ? 0.5f : (t == kSuper ? 1.f : -1.f);
    }
}

__device__ inline void init_newborn(Smem& s, int idx, int cell, int W, int gene, int brain, int uid)
{
    const int i = cell / W;
    s.pos[idx] = (unsigned short)(i | ((cell - i * W) << 8));
    s.health[idx] = 200; s.age[idx] = 0; s.max_age[idx] = 50;  // entities.py:145-159
    s.gene[idx] = gene; s.brain[idx] = brain; s.uid[idx] = uid;
    s.flags[idx] = 0; s.action[idx] = -1; s.fitness[idx] = 0.0; s.reward[idx] = 0.0;
    s.aux[idx] = 0; s.src[idx] = -1; s.tgt[idx] = (unsigned short)cell;
    s.occ[cell] = (short)idx; s.type[cell] = RL_AGENT;
}

// _reproduce + _produce + _remove_dead_agents (environment.py:488-547, 795-799) by WAVE 0 alone, no workgroup barriers:
// gates for the eligible agents in list order, one draw each (rank among the eligible = draw index); births are placed
// sequentially on the occupancy bitmap, the newborns themselves are initialised in parallel afterwards.  Touches only
// the occupancy grid, the new slots and (limit_reproduction) the parents' flags, so a fused tick runs it next to the
// state_prime observation pass of the other waves.
template <int T, bool LEAN>
__device__ __forceinline__ void reproduce_wave0(const KParams& p, Smem& s, int w, int n1, int nslots)
{
    const int tid = rl_tidx();
    const bool room = n1 <= p.max_agents;
    const bool tape = !LEAN && p.tape.food_k != nullptr;
    const uint32_t epoch = (uint32_t)s.scal[S_EPOCH], tick = (uint32_t)s.scal[S_TICK];
    const DrawCache dc = draw_cache(p, s);
    if (RL_ABL(8)) return;
    const int lane = tid;
    int rank_base = 0, npar = 0;
    for (int base = 0; base < n1; base += 64) {
        const int k = base + lane;
        const bool act = k < n1;
        const int a = act ? s.order[k] : 0;
        const int fl = s.flags[a], age = s.age[a], ge = s.gene[a], ps_ = s.pos[a];  // one batch (slot 0 is always valid)
        const bool e = act && room && !(fl & (RL_F_DEAD | RL_F_REPRODUCED)) && age > 5;  // can_reproduce, entities.py:244
        // _remove_dead_agents (environment.py:795-799) rides in the same pass: corpses become Food.  The reference does it after the
        // placements, which must still see the corpses' cells as occupied -- they do: the placements below work on the occupancy BITMAP
        // taken before this pass, and nothing else in this function reads type[] / occ[] of a corpse's cell.
        if (act && (fl & RL_F_DEAD)) {
            const int cell = (ps_ & 255) * p.W + (ps_ >> 8);
            s.type[cell] = RL_FOOD; s.occ[cell] = -1;
        }
        if (act && p.static_families && ge >= 0 && ge < RL_MAX_BRAINS) s.present[ge] = 1;
        const unsigned long long em = __ballot(e);
        bool par = false;
        if (e) {
            const int rank = rank_base + __popcll(em & lowmask(lane));
            double u;
            if (tape) u = p.tape.repro_u[(size_t)w * p.cap + rank];
            else if (LEAN && rank < dc.n_gate) u = rl_u24(dc.w[128 + rank]);
            else u = rl_u24(rl_philox4x32(p.seed, epoch, (uint32_t)(p.world_base + w), tick, RL_SITE_REPRO, (uint32_t)rank).x);
            par = u > 0.95;
            if (par && p.limit_reproduction) s.flags[a] = (uint8_t)(fl | RL_F_REPRODUCED);
        }
        const unsigned long long pm = __ballot(par);
        if (par) s.plist[npar + __popcll(pm & lowmask(lane))] = (short)a;
        npar += __popcll(pm);
        rank_base += __popcll(em);
    }
    RL_MARK(14);
    Placer P;
    placer_init(P, tid < p.nW ? s.occbits[tid] : ~0ull);
    int next_uid = s.scal[S_NEXT_UID];
    int max_gene = s.scal[S_MAX_GENE];
    int n_birth = 0, slots = nslots;
    // draw b of this tick's birth placements lives in lane b%64 (fetched / generated 64 at a time, in parallel)
    unsigned bdraw = 0; int bdraw_base = -64;
    auto birth_draw = [&](int b) -> unsigned {
        if (b >= bdraw_base + 64 || b < bdraw_base) {
            bdraw_base = b & ~63;
            const int mine = bdraw_base + tid;
            if (tape) bdraw = mine <= p.cap ? (unsigned)p.tape.birth_k[(size_t)w * (p.cap + 1) + mine] : 0u;
            else if (LEAN && mine < 64) bdraw = dc.w[32 + mine];
            else bdraw = rl_philox4x32(p.seed, epoch, (uint32_t)(p.world_base + w), tick, RL_SITE_BIRTH, (uint32_t)mine).x;
        }
        return (unsigned)read_lane((int)bdraw, b & 63);
    };
    // sequential part: only the placement; (cell, gene, brain) of newborn i are parked in tgt/gene/brain of its slot
    auto place_birth = [&](int gene, int brain, int errtag) {
        const unsigned x = birth_draw(n_birth);  // draw indices advance only when a draw happens
        const int k = tape ? (int)x : (int)rl_mulhi(x, (unsigned)P.n_empty);
        ++n_birth;
        if (k < 0 || k >= P.n_empty) { if (tid == 0) flag_error(p, s, 2, w, errtag, k); return; }
        const int cell = placer_take(P, k);
        if (slots >= p.cap) { if (tid == 0) flag_error(p, s, 3, w, slots, 0); return; }
        if (tid == 0) { s.tgt[slots] = (unsigned short)cell; s.gene[slots] = gene; s.brain[slots] = brain; }
        ++slots;
    };
    // the parents' gene / brain are fetched by one lane each up front (two LDS trips in all, not two per birth on the
    // serial path); newborn slots lie behind every parent slot, so parking the newborns' data cannot alias them
    int par_gene = 0, par_brain = 0;
    if (tid < npar) { const int par = s.plist[tid]; par_gene = s.gene[par]; par_brain = s.brain[par]; }
    // _produce's decision (environment.py:519-547) does not depend on the births; its placement comes after them
    int prod_gene = -1, prod_brain = 0;
    bool prod_place = false;
    if (room) {
        double u; unsigned x1 = 0;
        if (tape) u = p.tape.produce_u[w];
        else if (LEAN) { u = rl_u24(dc.w[16]); x1 = dc.w[17]; }
        else { const rl_u4 r = rl_philox4x32(p.seed, epoch, (uint32_t)(p.world_base + w), tick, RL_SITE_PRODUCE, 0u); u = rl_u24(r.x); x1 = r.y; }
        if (u > 0.95) {
            int gene = -1, brain = 0;
            if (p.static_families) {
                if (tape) gene = p.tape.produce_choice[w];
                else {
                    const bool absent = tid < p.n_brains && !s.present[tid];
                    const unsigned long long m = __ballot(absent);
                    const int cntabs = __popcll(m);
                    if (cntabs > 0) {
                        const int want = (int)rl_mulhi(x1, (unsigned)cntabs);
                        const unsigned long long hit = __ballot(absent && __popcll(m & lowmask(tid)) == want);
                        gene = __ffsll((long long)hit) - 1;
                    } else gene = (int)rl_mulhi(x1, (unsigned)p.n_brains);
                }
                brain = gene;
            } else {
                max_gene += 1;  // incremented even if the placement fails (environment.py:543)
                const int c = tape ? p.tape.produce_choice[w] : (int)rl_mulhi(x1, RL_N_BEST);
                gene = max_gene;
                brain = (c >= 0 && c < RL_N_BEST) ? s.best_brain[c] : 0;
                if (c < 0 || c >= RL_N_BEST) { if (tid == 0) flag_error(p, s, 4, w, c, 0); }
            }
            prod_gene = gene; prod_brain = brain; prod_place = gene >= 0;
        }
    }
    // All placements of the tick AT ONCE (lean tick: in-kernel draws).  Placement b takes the k_b-th empty cell of the grid
    // left by placements 0..b-1, k_b = floor(u_b * (E - b)): sequential by definition, ~450 cycles each through the wave's
    // bitmap -- and a cohort that comes of age together gives one world 20 births in a tick, which then holds up the whole
    // launch.  But "k-th element of the complement of a sorted set C" is k + |{j : C_j - j <= k}|, so the ranks r_b in the
    // ORIGINAL list of empty cells follow from a short scalar recurrence (two ballots per placement on a sorted register
    // across the lanes), and all cells are then selected in parallel: lane b looks its word up in the prefix of the empty
    // counts and picks the bit with a six-step popcount search.
    const int n_empty0 = P.n_empty;
    const int nb_births = min(npar, n_empty0);  // a parent whose turn finds the grid full draws nothing (grid.py:82-83)
    const bool batch = LEAN && !tape && npar < 64 && nslots + npar + 1 <= p.cap;
    if (batch) {
        const bool prod_now = prod_place && n_empty0 - nb_births > 0;
        const int total = nb_births + (prod_now ? 1 : 0);
        const unsigned x = dc.w[32 + tid];                                 // birth draw `tid` (the cache holds 64 of them)
        const int k = tid < total ? (int)rl_mulhi(x, (unsigned)(n_empty0 - tid)) : 0;
        int chosen = 0x7fffffff, rsel = 0;                                    // lane j: j-th smallest rank chosen so far
        for (int b = 0; b < total; ++b) {
            const int kb = read_lane(k, b);
            const int r = kb + __popcll(__ballot(tid < b && chosen - tid <= kb));
            const int pos = __popcll(__ballot(tid < b && chosen < r));
            const int up = __builtin_amdgcn_update_dpp(chosen, chosen, 0x138, 0xF, 0xF, false);  // wave_shr:1 -- lane j takes lane j-1's
            chosen = tid > pos ? up : (tid == pos ? r : chosen);
            if (tid == b) rsel = r;
        }
        s.wordbase[tid] = P.incl;  // (free here: the orderings that use it are built before / after this section)
        int L = 0;
        for (int wd = 0; wd < p.nW; ++wd) L += s.wordbase[wd] <= rsel;
        L = min(L, p.nW - 1);
        unsigned long long zz = ~s.occbits[L];
        int kk = rsel - (L ? s.wordbase[L - 1] : 0), bit = 0;
#pragma unroll
        for (int sft = 32; sft; sft >>= 1) {
            const int c = __popcll(zz & ((1ull << sft) - 1ull));
            if (kk >= c) { kk -= c; zz >>= sft; bit += sft; }
        }
        if (tid < total) {
            const bool is_prod = tid >= nb_births;
            const int slot = nslots + tid;
            s.tgt[slot] = (unsigned short)(L * 64 + bit);
            s.gene[slot] = is_prod ? prod_gene : par_gene;
            s.brain[slot] = is_prod ? prod_brain : (p.static_families ? par_gene : par_brain);
        }
        slots = nslots + total; n_birth = total;
        RL_MARK(41);
    } else {
        for (int b = 0; b < npar; ++b) {
            if (P.n_empty <= 0) continue;  // full grid: randint raises, no draw, no offspring (grid.py:82-83)
            int g, br;
            if (b < 64) { g = read_lane(par_gene, b); br = read_lane(par_brain, b); }
            else { const int par = s.plist[b]; g = s.gene[par]; br = s.brain[par]; }
            place_birth(g, p.static_families ? g : br, b);
        }
        RL_MARK(41);
        if (prod_place && P.n_empty > 0) place_birth(prod_gene, prod_brain, -1);
    }
    RL_MARK(42);
    // newborns (entities.py:145-159), initialised in parallel: lane i -> slot nslots + i
    for (int i = nslots + lane; i < slots; i += 64) {
        const int cell = s.tgt[i];
        init_newborn(s, i, cell, p.W, s.gene[i], s.brain[i], next_uid + (i - nslots));
    }
    next_uid += slots - nslots;
    if (tid == 0) {
        s.scal[S_NSLOTS] = slots; s.scal[S_NEXT_UID] = next_uid; s.scal[S_MAX_GENE] = max_gene;
        p.st.next_uid[w] = next_uid; p.st.max_gene[w] = max_gene;
    }
}

// _update_best_agents (environment.py:728-739) by ONE wave (no workgroup barrier): used when the update's serial section
// runs on wave 0 next to the other waves' observation pass
__device__ inline void best_agents_wave(Smem& s, int n1)
{
    const int lane = lane_id();
    double bf = -1.0e300; int bk = 0x7fffffff;
    for (int k = lane; k < n1; k += 64) {
        const double f = s.fitness[s.order[k]];
        if (f > bf || (f == bf && k < bk)) { bf = f; bk = k; }
    }
#pragma unroll
    for (int m = 32; m; m >>= 1) {
        const double of = shfl_xor_f64(bf, m); const int ok = __shfl_xor(bk, m);
        if (of > bf || (of == bf && ok < bk)) { bf = of; bk = ok; }
    }
    if (lane == 0 && n1 > 0) {
        int mi = 0;
        for (int b = 1; b < RL_N_BEST; ++b) if (s.best_fit[b] < s.best_fit[mi]) mi = b;
        const int a = s.order[bk];
        bool present = false;
        for (int b = 0; b < RL_N_BEST; ++b) present |= s.best_uid[b] == s.uid[a];
        if (!present && bf > s.best_fit[mi]) { s.best_uid[mi] = s.uid[a]; s.best_fit[mi] = bf; s.best_brain[mi] = s.brain[a]; }
    }
}

// Environment.update_env up to (not including) the observation pass.  order[0..n1) is the grid list.
template <int T, bool LEAN>
__device__ __forceinline__ void phase_update(const KParams& p, Smem& s, int w, int n1, int& nslots, bool fresh_bitmap)
{
    const int tid = rl_tidx();
    // ---- _update_best_agents (environment.py:728-739) ----------------------------------------------------------------
    if (!p.static_families) {
        double bf = -1.0e300; int bk = 0x7fffffff;
        for (int k = tid; k < n1; k += T) {
            const double f = s.fitness[s.order[k]];
            if (f > bf || (f == bf && k < bk)) { bf = f; bk = k; }
        }
#pragma unroll
        for (int m = 32; m; m >>= 1) {
            const double of = shfl_xor_f64(bf, m); const int ok = __shfl_xor(bk, m);
            if (of > bf || (of == bf && ok < bk)) { bf = of; bk = ok; }
        }
        if (lane_id() == 0) { s.wred_f[tid >> 6] = bf; s.wred_k[tid >> 6] = bk; }
        lds_barrier();
        if (tid == 0 && n1 > 0) {
            for (int v = 1; v < T / 64; ++v)
                if (s.wred_f[v] > bf || (s.wred_f[v] == bf && s.wred_k[v] < bk)) { bf = s.wred_f[v]; bk = s.wred_k[v]; }
            int mi = 0;
            for (int b = 1; b < RL_N_BEST; ++b) if (s.best_fit[b] < s.best_fit[mi]) mi = b;
            const int a = s.order[bk];
            bool present = false;
            for (int b = 0; b < RL_N_BEST; ++b) present |= s.best_uid[b] == s.uid[a];
            if (!present && bf > s.best_fit[mi]) { s.best_uid[mi] = s.uid[a]; s.best_fit[mi] = bf; s.best_brain[mi] = s.brain[a]; }
        }
        lds_barrier();
    }
    RL_MARK(13);
    RL_MARK(13);
    if (!fresh_bitmap) {  // standalone update: rebuild the occupancy bitmap (a fused tick reuses the food phase's)
        for (int c = tid; c < p.Cp; c += T) {
            const unsigned long long m = __ballot(s.type[c] != RL_EMPTY);
            if (lane_id() == 0) s.occbits[c >> 6] = m;
        }
        lds_barrier();
    }
    if (tid < 64) reproduce_wave0<T, LEAN>(p, s, w, n1, nslots);
    lds_barrier();
    nslots = s.scal[S_NSLOTS];
    RL_MARK(15);
}

// on-grid gene counts for the observation's percent_genes (environment.py:357)
template <int T>
__device__ __forceinline__ void rebuild_gene_counts(const KParams& p, Smem& s, int n)
{
    for (int i = rl_tidx(); i < p.hash_size; i += T) { s.hkey[i] = -1; s.hcnt[i] = 0u; }
    lds_barrier();
    const int np2 = (n + 63) & ~63;
    for (int k = rl_tidx(); k < np2; k += T) {
        const bool act = k < n;
        const int a = act ? s.order[k] : 0;
        hash_insert_wave(s, p.hash_mask, act, a, act ? s.gene[a] : 0, 1u << 16);
    }
    lds_barrier();
}

// inclusive running maximum over the 64 lanes (values >= 0), same DPP pattern as wave_incl_scan
__device__ inline int wave_incl_scan_max(int v)
{
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, true));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, true));
    return v;
}

// numpy's pairwise float64 summation (np.add.reduce; what np.mean of the Tracker's per-agent lists does): blocks of
// <= 128 use eight accumulators + a sequential tail, longer arrays split at a multiple of 8.
__device__ __noinline__ double np_pairwise_block(const double* a, int n)
{
    if (n < 8) { double r = 0.0; for (int i = 0; i < n; ++i) r += a[i]; return r; }
    double r0 = a[0], r1 = a[1], r2 = a[2], r3 = a[3], r4 = a[4], r5 = a[5], r6 = a[6], r7 = a[7];
    int i = 8;
    for (; i < n - (n % 8); i += 8) { r0 += a[i]; r1 += a[i + 1]; r2 += a[i + 2]; r3 += a[i + 3]; r4 += a[i + 4]; r5 += a[i + 5]; r6 += a[i + 6]; r7 += a[i + 7]; }
    double res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
    for (; i < n; ++i) res += a[i];
    return res;
}
template <int DEPTH>
__device__ inline double np_pairwise_sum(const double* a, int n)
{
    if (n <= 128) return np_pairwise_block(a, n);
    int n2 = n / 2; n2 -= n2 % 8;
    return np_pairwise_sum<DEPTH - 1>(a, n2) + np_pairwise_sum<DEPTH - 1>(a + n2, n - n2);
}
template <>
__device__ inline double np_pairwise_sum<0>(const double* a, int n) { return np_pairwise_block(a, n); }

// Tracker._track_results over the post-step list (Helpers/tracker.py:178-266), executed by wave 0 without atomics:
// lane g owns group g (gene g with static families, everybody otherwise).
__device__ __forceinline__ void track_world_wave0(const KParams& p, Smem& s, int w, int n1)
{
    const int lane = lane_id();
    const int G = p.static_families ? p.n_brains : 1;
    int m = 0, sum_age = 0, best = 0, attacks = 0, kills = 0;  // lane g: statistics of group g
    // pass 1: per-group integer statistics, one ballot-aggregated step per distinct group of every 64-agent chunk
    for (int base = 0; base < n1; base += 64) {
        const int k = base + lane;
        const bool act = k < n1;
        const int a = act ? s.order[k] : 0;
        const int g = act ? (p.static_families ? s.gene[a] : 0) : -1;
        const bool in_range = act && g >= 0 && g < G;
        unsigned long long pending = __ballot(in_range);
        while (pending) {
            const int gg = read_lane(g, __ffsll((long long)pending) - 1);
            const bool mine = in_range && g == gg;
            const unsigned long long mm = __ballot(mine);
            const int age_sum = read_lane(wave_incl_scan(mine ? s.age[a] : 0), 63);
            const int age_max = read_lane(wave_incl_scan_max(mine ? s.age[a] : 0), 63);
            const int att = __popcll(__ballot(mine && s.action[a] >= 4));
            const int kil = __popcll(__ballot(mine && (s.flags[a] & RL_F_KILLED)));
            if (lane == gg) { m += __popcll(mm); sum_age += age_sum; best = max(best, age_max); attacks += att; kills += kil; }
            pending &= ~mm;
        }
    }
    // pass 2: rewards of each group contiguous and in list order (np.mean's summation order depends on it)
    const int incl = wave_incl_scan(lane < G ? m : 0);
    const int off = incl - (lane < G ? m : 0);
    int run = off;
    for (int base = 0; base < n1; base += 64) {
        const int k = base + lane;
        const bool act = k < n1;
        const int a = act ? s.order[k] : 0;
        const int g = act ? (p.static_families ? s.gene[a] : 0) : -1;
        const bool in_range = act && g >= 0 && g < G;
        unsigned long long pending = __ballot(in_range);
        while (pending) {
            const int gg = read_lane(g, __ffsll((long long)pending) - 1);
            const bool mine = in_range && g == gg;
            const unsigned long long mm = __ballot(mine);
            const int start = read_lane(run, gg);
            if (mine) s.trk_rew[start + __popcll(mm & lowmask(lane))] = s.reward[a];
            if (lane == gg) run += __popcll(mm);
            pending &= ~mm;
        }
    }
    // number of populations = distinct genes on the grid
    int n_distinct;
    if (p.static_families) n_distinct = __popcll(__ballot(lane < G && m > 0));
    else {
        int c = 0;
        for (int i = lane; i < p.hash_size; i += 64) c += (s.hkey[i] != -1 && (s.hcnt[i] >> 16) != 0);
        n_distinct = read_lane(wave_incl_scan(c), 63);
    }
    if (lane < G) {
        double v[RL_TRK_VARS];
        if (n1 == 0) {
#pragma unroll
            for (int i = 0; i < RL_TRK_VARS; ++i) v[i] = -1.0;
        } else {
            if (m == 0) { v[0] = v[1] = v[2] = v[3] = v[4] = -1.0; }
            else {
                v[0] = p.static_families ? (double)m : (double)n1 / (double)n_distinct;
                v[1] = (double)sum_age / (double)m;
                v[2] = np_pairwise_sum<5>(s.trk_rew + off, m) / (double)m;
                v[3] = (double)best;
                v[4] = (double)attacks / (double)m;
            }
            v[5] = (double)kills;
            v[6] = kills != 0 ? 1.0 : 0.0;
        }
        const size_t o = ((size_t)w * G + lane) * RL_TRK_VARS;
#pragma unroll
        for (int i = 0; i < RL_TRK_VARS; ++i) {
            p.so.trk_tick[o + i] = v[i];
            if (v[i] > -1.0) { p.so.trk_sum[o + i] += v[i]; p.so.trk_cnt[o + i] += 1; }
        }
    }
    if (lane == 0) {
        const double pv = n1 == 0 ? -1.0 : (double)n_distinct;
        p.so.trk_pop[(size_t)w * 3] = pv;
        if (pv > -1.0) { p.so.trk_pop[(size_t)w * 3 + 1] += pv; p.so.trk_pop[(size_t)w * 3 + 2] += 1.0; }
    }
}

// Per-brain row lists for the policy kernel (replaces a separate bucket launch): wave 0 counts the world's agents per
// brain with ballots (lane b keeps brain b's count), ONE atomic instruction reserves the ranges of all brains, a second
// pass scatters the row ids.  `brain_of(k)` reads the brain of list entry k (from LDS or from HBM).
template <typename F>
__device__ inline void emit_brain_lists_wave0(const KParams& p, int w, int n, F brain_of)
{
    const int lane = lane_id();
    if (blockIdx.x == 0) p.lists_counts_zero[lane] = 0;
    int cnt = 0;
    for (int base = 0; base < n; base += 64) {
        const int k = base + lane;
        const int b = k < n ? brain_of(k) : -1;
        for (int bb = 0; bb < p.n_brains; ++bb) {
            const int c = __popcll(__ballot(b == bb));
            if (lane == bb) cnt += c;
        }
    }
    int pos = (lane < p.n_brains && cnt) ? atomicAdd(&p.lists_counts[lane], cnt) : 0;
    for (int base = 0; base < n; base += 64) {
        const int k = base + lane;
        const int b = k < n ? brain_of(k) : -1;
        for (int bb = 0; bb < p.n_brains; ++bb) {
            const unsigned long long m = __ballot(b == bb);
            const int start = read_lane(pos, bb);
            if (b == bb) p.lists[bb * p.list_stride + start + __popcll(m & lowmask(lane))] = rl_list_entry(w, k);
            if (lane == bb) pos += __popcll(m);
        }
    }
}

template <int T>
__device__ __forceinline__ void store_world(const KParams& p, Smem& s, int w, int n)
{
    const int tid = rl_tidx();
    KParamsC* q = kernargs();
    if (RL_ABL(64)) return;
    auto gt = RL_G(q->st.cell_type) + (size_t)w * p.C;
    if (!RL_ABL(1024)) for (int c = tid; c < p.C; c += T) gt[c] = s.type[c];
    const size_t b = (size_t)w * p.cap;
    // (agents are dealt from the TOP thread down: in the fused tick the low waves go on to the row lists and the first
    // observation rows)
    if (!RL_ABL(2048))
    for (int k = T - 1 - tid; k < n; k += T) {
        const int a = s.order[k];
        if (!RL_ABL(4096)) {
        RL_G(q->st.a_i)[b + k] = (uint8_t)(s.pos[a] & 255);
        RL_G(q->st.a_j)[b + k] = (uint8_t)(s.pos[a] >> 8);
        RL_G(q->st.a_flags)[b + k] = s.flags[a];
        RL_G(q->st.a_action)[b + k] = s.action[a];
        }
        if (!RL_ABL(8192)) {
        RL_G(q->st.a_health)[b + k] = s.health[a];
        RL_G(q->st.a_age)[b + k] = s.age[a];
        RL_G(q->st.a_max_age)[b + k] = s.max_age[a];
        RL_G(q->st.a_gene)[b + k] = s.gene[a];
        RL_G(q->st.a_brain)[b + k] = s.brain[a];
        RL_G(q->st.a_uid)[b + k] = s.uid[a];
        }
        if (!RL_ABL(16384)) RL_G(q->st.a_fitness)[b + k] = s.fitness[a];
    }
    if (tid == 0) RL_G(q->st.n_agents)[w] = n;
    if (tid < RL_N_BEST && !p.static_families) {
        RL_G(q->st.best_uid)[(size_t)w * RL_N_BEST + tid] = s.best_uid[tid];
        RL_G(q->st.best_fit)[(size_t)w * RL_N_BEST + tid] = s.best_fit[tid];
        RL_G(q->st.best_brain)[(size_t)w * RL_N_BEST + tid] = s.best_brain[tid];
    }
}

template <int T>
__device__ __forceinline__ int reset_world_lds(const KParams& p, Smem& s, int w, uint32_t epoch);

enum { MODE_STEP = 0, MODE_UPDATE = 1, MODE_TICK = 2, MODE_OBSERVE = 3, MODE_FOOD = 4 };

// LEAN = performance path: no recorded tape, no tracker, no capture outputs (their pointers are known to be null), which
// lets the compiler drop those parameters and branches (SGPR pressure: the full kernel keeps ~45 pointers alive)
// FIXED: the kernel is specialised for ONE world shape -- the reference's default, trainer(width=30, height=30,
// max_agents=100) -- and the host picks it when the handle has exactly that shape (any other shape runs the generic code).
// Two things come from it:
//  * the LDS layout is carved from constants, so every array base is an immediate.  With run-time sizes the ~35 bases do
//    not fit in SGPRs next to everything else and the compiler RE-DERIVES them (an s_add/s_and chain of ~70 scalar
//    instructions) at the top of most barrier intervals -- a few hundred cycles, twenty-odd times per tick;
//  * width, height, the padded cell count and the slot capacity fold into the address arithmetic (no run-time division by
//    the width, single-trip cell loops, constant window wrap): another ~1,100 instructions and 28 VGPRs less.
constexpr int kFixW = 30, kFixH = 30, kFixMaxAgents = 100;
constexpr int kFixC = kFixW * kFixH, kFixCp = (kFixC + 63) & ~63;
constexpr int kFixCap = ((2 * kFixMaxAgents + 2 + 63) / 64) * 64;   // 256: births can overshoot max_agents up to 2n+1
constexpr int kFixHash = 512;                                       // rl_create: the power of two >= 2 * slot_cap
static_assert(kFixHash >= 2 * kFixCap && kFixHash / 2 < 2 * kFixCap, "hash size rule of rl_create");
template <int T, int MODE, bool LEAN, bool FIXED = false>
__global__ __launch_bounds__(T) void k_world(const KParams p_in)
{
    KParams p = p_in;
    if (FIXED) {
        p.W = kFixW; p.H = kFixH; p.C = kFixC; p.Cp = kFixCp; p.nW = kFixCp / 64;
        p.cap = kFixCap; p.hash_size = kFixHash; p.hash_mask = kFixHash - 1;
    }
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
#ifdef RL_PHASE_PROFILE
    unsigned long long t_entry;  // before the first kernel-argument access
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_entry));
#endif
    {   // Touch every 64-byte line of the kernel-argument block with ONE batch of scalar loads: the compiler fetches
        // arguments lazily, a few at a time, and each first touch of a line is a scalar-cache miss (~0.2 us) on the
        // critical path of load_world; afterwards they are hits.
        auto ka = __builtin_amdgcn_kernarg_segment_ptr();
        int t0, t1, t2, t3, t4, t5, t6, t7;
        asm volatile("s_load_dword %0, %8, 0x0\n\ts_load_dword %1, %8, 0x40\n\ts_load_dword %2, %8, 0x80\n\t"
                     "s_load_dword %3, %8, 0xc0\n\ts_load_dword %4, %8, 0x100\n\ts_load_dword %5, %8, 0x140\n\t"
                     "s_load_dword %6, %8, 0x180\n\ts_load_dword %7, %8, 0x1c0\n\ts_waitcnt lgkmcnt(0)"
                     : "=s"(t0), "=s"(t1), "=s"(t2), "=s"(t3), "=s"(t4), "=s"(t5), "=s"(t6), "=s"(t7)
                     : "s"(ka) : "memory");
        static_assert(sizeof(KParams) >= 0x1c0 + 4 && sizeof(KParams) <= 0x200, "the warm-up loads must cover the argument block");
    }
    Smem s;
    if (FIXED) carve(s, smem_raw, kFixCp, kFixCap, kFixHash);
    else carve(s, smem_raw, p.Cp, p.cap, p.hash_size);
    const int w = blockIdx.x;
    const int tid = rl_tidx();
    int n0;
    RL_MARK(0);
#ifdef RL_PHASE_PROFILE
    if (p.prof && (int)blockIdx.x == p.prof_world && rl_tidx() == 0) p.prof[23] = (long long)t_entry;
#endif
    if (RL_ABL(32768)) return;
    // (speculative refill: lean fused tick, 1024-thread workgroups -- the latency-bound regime of one world per CU)
    constexpr bool kSpec = LEAN && MODE == MODE_TICK && T == 1024;
    load_world<T, kSpec>(p, s, w, n0);
    RL_MARK(1);
    if (RL_ABL(65536)) return;
    int nslots = n0;
    int n_cur = n0;  // length of order[]
    bool overlapped = false;  // the update's wave-0 section already ran next to the state_prime pass

    if (MODE == MODE_FOOD) {
        // second half of a split step (_add_food with a host-drawn tape, then the observation pass, environment.py:185-186)
        int nf = 0, np_ = 0, ns = 0;  // per wave
        for (int c = tid; c < p.Cp; c += T) {
            const int t = s.type[c];
            nf += __popcll(__ballot(t == RL_FOOD)); np_ += __popcll(__ballot(t == RL_POISON)); ns += __popcll(__ballot(t == kSuper));
            const unsigned long long m = __ballot(t != RL_EMPTY);
            if (lane_id() == 0) s.occbits[c >> 6] = m;
        }
        if (lane_id() == 0) {
            if (nf) atomicAdd(&s.scal[S_NFOOD], nf);
            if (np_) atomicAdd(&s.scal[S_NPOISON], np_);
            if (ns) atomicAdd(&s.scal[S_NSUPER], ns);
        }
        lds_barrier();
        if (tid < 64) {
            Placer P;
            placer_init(P, tid < p.nW ? s.occbits[tid] : ~0ull);
            int xk = 0; double u = 2.0;
            if (tid < RL_FOOD_TRIES && p.tape.food_k) { xk = p.tape.food_k[(size_t)w * RL_FOOD_TRIES + tid]; u = p.tape.food_u[(size_t)w * RL_FOOD_TRIES + tid]; }
            const bool en = tid < 3 ? (double)s.scal[S_NFOOD] <= (double)p.C / 10.0 : (tid < 6 ? (double)s.scal[S_NPOISON] <= (double)p.C / 20.0 : s.scal[S_NSUPER] == 0);
            unsigned long long todo = __ballot(tid < RL_FOOD_TRIES && en && u < (tid < 6 ? 0.2 : 1.0));
            while (todo) {
                const int t = __ffsll((long long)todo) - 1;
                todo &= todo - 1;
                if (P.n_empty <= 0) break;
                const int k = read_lane(xk, t);
                if (k < 0 || k >= P.n_empty) { if (tid == 0) flag_error(p, s, 1, w, t, k); continue; }
                const int cell = placer_take(P, k);
                if (tid == 0) s.type[cell] = (uint8_t)(t < 3 ? RL_FOOD : (t < 6 ? RL_POISON : kSuper));
            }
        }
        lds_barrier();
        rebuild_gene_counts<T>(p, s, n0);
        build_planes<T>(p, s);
        lds_barrier();
        write_observations<T>(p, s, w, n0, p.so.obs);
        uint8_t* gt = p.st.cell_type + (size_t)w * p.C;
        for (int c = tid; c < p.C; c += T) gt[c] = s.type[c];
        return;
    }
    if (MODE == MODE_OBSERVE) {
        rebuild_gene_counts<T>(p, s, n0);
        build_planes<T>(p, s);
        lds_barrier();
        write_observations<T>(p, s, w, n0, p.obs_only);
        return;
    }
    if (MODE == MODE_STEP || MODE == MODE_TICK) {
        const bool split = !LEAN && MODE == MODE_STEP && p.split_food;  // observation pass comes with the food half
        constexpr bool kPlanesEarly = T >= 256 && MODE == MODE_TICK;  // (a split step returns before the placement interval)
        if (!RL_ABL(512)) {
            // leaves the agent bitmap, its prefix and scal[S_N1] of the new ordering -- and, kPlanesEarly, the planes
            phase_step<T, LEAN, kPlanesEarly, kSpec>(p, s, w, n0);
            RL_MARK(8);
            assign_order<T>(p, s, nslots);     // same barrier interval as the planes: they do not read the ordering
            if (kPlanesEarly) patch_placed_planes(s);
        } else build_order<T>(p, s, nslots, S_N1);
        RL_MARK(9);
        const int n1 = s.scal[S_N1];
        if (!split && !(kPlanesEarly && !RL_ABL(512))) build_planes<T>(p, s);
        lds_barrier();
        RL_MARK(10);
        // Lean fused tick: wave 0 runs _update_best_agents / _reproduce / _produce / _remove_dead_agents (serial work on
        // the occupancy bitmap and the new slots) WHILE the other waves write the state_prime rows and the step outputs.
        // (limit_reproduction sets a flag the observation pass reads: it takes the sequential path.)
        overlapped = LEAN && MODE == MODE_TICK && T > 64 && !p.limit_reproduction && !RL_ABL(256);
        const size_t b = (size_t)w * p.cap;
        auto step_outputs = [&](int t, int nt) {
            for (int k = t; k < n1; k += nt) {
                const int a = s.order[k];
                if (p.so.reward) p.so.reward[b + k] = (float)s.reward[a];
                if (p.so.done) p.so.done[b + k] = (s.flags[a] & RL_F_DEAD) ? 1 : 0;
                if (p.so.src) p.so.src[b + k] = (short)a;
                if (!LEAN && p.so.age) p.so.age[b + k] = s.age[a];
                if (!LEAN && p.so.brain) p.so.brain[b + k] = s.brain[a];
            }
            if (t == 0 && p.so.n_acted) p.so.n_acted[w] = n0;
            if (!LEAN && t == 0 && p.so.n_post) p.so.n_post[w] = n1;
            if (t == 0 && p.so.acted_total && n0) atomicAdd(p.so.acted_total, (unsigned long long)n0);
        };
        if (overlapped) {
            if (tid < 64) {
                if (!p.static_families) best_agents_wave(s, n1);  // _produce reads best_brain: same wave, program order
                reproduce_wave0<T, LEAN>(p, s, w, n1, nslots);
            } else {
                RL_MARK_T(43, 64); RL_MARK_T(48, 512); RL_MARK_T(51, 896);
                write_observations<(T > 64 ? T - 64 : 64)>(p, s, w, n1, p.so.obs, tid - 64);
                RL_MARK_T(44, 64); RL_MARK_T(49, 512); RL_MARK_T(52, 896);
                step_outputs(tid - 64, T - 64);
                RL_MARK_T(45, 64); RL_MARK_T(50, 512); RL_MARK_T(53, 896);
            }
        } else {
            if (!split) write_observations<T>(p, s, w, n1, p.so.obs);
            RL_MARK(11);
            step_outputs(tid, T);
        }
        if (!LEAN && p.so.trk_tick && tid < 64) track_world_wave0(p, s, w, n1);
        n_cur = n1;
        if (MODE == MODE_STEP) { store_world<T>(p, s, w, n1); return; }
        RL_MARK_W(64);   // (tuning build: stamps need a 128-entry buffer)
        lds_barrier();
        RL_MARK_W(80);
        // fused tick: agents keep their LDS slot; remember their post-step list index for uo.src (newborns carry -1)
        for (int a = tid; a < nslots; a += T) s.src[a] = s.newidx[a];
        if (!overlapped) lds_barrier();  // (overlapped: this loop shares the interval of the update's first sweep below)
        else nslots = s.scal[S_NSLOTS];
    }
    if (MODE == MODE_UPDATE || MODE == MODE_TICK) {
        const int n1 = n_cur;
        RL_MARK(12);
        if (!RL_ABL(256) && !overlapped) phase_update<T, LEAN>(p, s, w, n1, nslots, MODE == MODE_TICK);
        RL_MARK(17);
        // new ordering + on-grid gene counts + observation planes in three barrier intervals: (1) agent bitmap sweep and
        // gene table clear, (2) prefix scan by one wave, (3) order assignment, gene counting by slot and the planes
        for (int c = tid; c < p.Cp; c += T) {
            const unsigned long long m = __ballot(s.type[c] == RL_AGENT);
            if (lane_id() == 0) s.agbits[c >> 6] = m;
        }
        for (int i = tid; i < p.hash_size; i += T) { s.hkey[i] = -1; s.hcnt[i] = 0u; }
        lds_barrier();
        if (tid < 64) scan_order_wave(p, s, tid, S_N2);
        lds_barrier();
        RL_MARK(18);
        int n2 = s.scal[S_N2];
        // optional fused refill: a world whose population fell below the threshold is re-generated
        const bool refill = p.refill_threshold >= 0 && n2 < p.refill_threshold;  // uniform per workgroup
        RL_MARK(19);
        if (refill) {
            const bool prepared = kSpec && s.scal[S_SPEC_DONE] != 0;  // (read before the barrier inside either path clears the slots)
            const uint32_t new_epoch = (uint32_t)s.scal[S_EPOCH] + 1u;
            if (prepared) {  // unpack, then count the genes in the same interval as the planes below
                n2 = apply_spec_refill<T>(p, s, w, new_epoch);
                const int np2 = (n2 + 63) & ~63;
                for (int k = tid; k < np2; k += T) hash_insert_wave(s, p.hash_mask, k < n2, k, k < n2 ? s.gene[k] : 0, 1u << 16);  // order[k] == k
            } else {
                n2 = reset_world_lds<T>(p, s, w, new_epoch);
                rebuild_gene_counts<T>(p, s, n2);
            }
        } else {
            assign_order<T>(p, s, nslots);
            const int nsp = (nslots + 63) & ~63;  // whole waves take part in the gene aggregation
            for (int a = tid; a < nsp; a += T) {
                const int aa = a < nslots ? a : 0;
                const int ps = s.pos[aa], ge = s.gene[aa];  // one batch
                const bool on = a < nslots && s.occ[(ps & 255) * p.W + (ps >> 8)] == a;
                hash_insert_wave(s, p.hash_mask, on, a, on ? ge : 0, 1u << 16);
            }
        }
        build_planes<T>(p, s);
        lds_barrier();
        RL_MARK(20);
        // the world's state goes out FIRST (a few dependent LDS reads per thread, then stores): issued behind the observation
        // rows it would queue after ~27 KB of row stores per wave at the very end of the launch
        if (p.uo.src) {
            const size_t b = (size_t)w * p.cap;
            for (int k = tid; k < n2; k += T) p.uo.src[b + k] = s.src[s.order[k]];
        }
        store_world<T>(p, s, w, n2);
        RL_MARK(22);
        if (LEAN && MODE == MODE_TICK && T > 64 && p.lists && !RL_ABL(128)) {
            // wave 0 reserves and fills the per-brain row lists (an atomic round trip) while the others write the rows
            if (tid < 64) emit_brain_lists_wave0(p, w, n2, [&](int k) { return s.brain[s.order[k]]; });
            else {
                RL_MARK_T(46, 64);
                write_observations<(T > 64 ? T - 64 : 64)>(p, s, w, n2, p.uo.obs, tid - 64);
                RL_MARK_T(47, 64);
            }
        } else {
            write_observations<T>(p, s, w, n2, p.uo.obs);
            if (p.lists && tid < 64 && !RL_ABL(128)) emit_brain_lists_wave0(p, w, n2, [&](int k) { return s.brain[s.order[k]]; });
        }
        RL_MARK(21);
        if (tid == 0 && !refill) p.st.tick[w] = s.scal[S_TICK] + 1;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// synthetic world generator (same rule as oracle/rl_oracle.c reset_world), parallel by design:
// every cell draws one Philox block; cells are ranked by a unique random key; the first n_agents ranks become agents,
// then Binomial(C,.1) Food, Binomial(C,.05) Poison, one SuperFood (Environment._init_food's counts, environment.py:741-761)
// ---------------------------------------------------------------------------------------------------------------
// Leaves LDS holding the new world: type/occ, agents in slots 0..n-1 in row-major order, order[k] = k.  Returns n.
template <int T>
__device__ __forceinline__ int reset_world_lds(const KParams& p, Smem& s, int w, uint32_t epoch)
{
    const int tid = rl_tidx();
    // LDS scratch (the observation planes are rebuilt afterwards): key per cell, bucket counters, keys grouped by bucket
    unsigned* keys = (unsigned*)s.genev;
    unsigned* cum = (unsigned*)s.foodv;
    unsigned* sorted = (unsigned*)s.healthv;
    int lg = 6;
    while ((2 << lg) <= p.Cp) ++lg;          // NB = largest power of two <= Cp
    const int NB = 1 << lg, sh = 32 - lg;
    lds_barrier();
    if (tid < S_COUNT) s.scal[tid] = 0;
    for (int b = tid; b < NB; b += T) cum[b] = 0u;
    lds_barrier();
    // 1. one Philox block per cell: unique random key, food / poison coins; histogram of the key prefixes
    int nf = 0, np_ = 0;
    for (int c = tid; c < p.Cp; c += T) {
        s.occ[c] = -1;
        if (c < p.C) {
            const rl_u4 r = rl_philox4x32(p.seed, epoch, (uint32_t)(p.world_base + w), 0u, RL_SITE_RESET_AGENT, (uint32_t)c);
            const unsigned key = (r.x & ~0xFFFu) | (unsigned)c;
            keys[c] = key;
            atomicAdd(&cum[key >> sh], 1u);
            nf += rl_u24(r.z) < 0.1; np_ += rl_u24(r.w) < 0.05;
        }
    }
    if (nf) atomicAdd(&s.scal[S_NFOOD], nf);
    if (np_) atomicAdd(&s.scal[S_NPOISON], np_);
    lds_barrier();
    // 2. exclusive scan of the NB bucket counts, in place (each thread owns `per` consecutive buckets)
    {
        const int per = (NB + T - 1) / T;
        const int b0 = tid * per;
        unsigned local = 0;
        for (int i = 0; i < per; ++i) if (b0 + i < NB) local += cum[b0 + i];
        const int incl = wave_incl_scan((int)local);
        if (lane_id() == 63) s.wred_k[tid >> 6] = incl;
        lds_barrier();
        unsigned base = (unsigned)(incl - (int)local);
        for (int v = 0; v < (tid >> 6); ++v) base += (unsigned)s.wred_k[v];
        for (int i = 0; i < per; ++i)
            if (b0 + i < NB) { const unsigned c = cum[b0 + i]; cum[b0 + i] = base; base += c; }
    }
    lds_barrier();
    // 3. counting-sort scatter: afterwards cum[b] is the END of bucket b (= start of bucket b+1)
    for (int c = tid; c < p.C; c += T) {
        const unsigned key = keys[c];
        sorted[atomicAdd(&cum[key >> sh], 1u)] = key;
    }
    lds_barrier();
    // 4. exact rank = bucket start + smaller keys inside the (one- or two-element) bucket; classify the cell
    const int na = min(p.reset_n_agents, p.C);
    const int k1 = na, k2 = na + s.scal[S_NFOOD], k3 = k2 + s.scal[S_NPOISON];
    for (int c = tid; c < p.Cp; c += T) {
        uint8_t t = kPadCell;
        if (c < p.C) {
            const unsigned key = keys[c];
            const unsigned b = key >> sh;
            const int start = b ? (int)cum[b - 1] : 0, end = (int)cum[b];
            int rank = start;
            for (int j = start; j < end; ++j) rank += sorted[j] < key;
            t = (uint8_t)(rank < k1 ? RL_AGENT : rank < k2 ? RL_FOOD : rank < k3 ? RL_POISON : rank == k3 ? kSuper : RL_EMPTY);
        }
        s.type[c] = t;
    }
    lds_barrier();
    for (int c = tid; c < p.Cp; c += T) {
        const unsigned long long m = __ballot(s.type[c] == RL_AGENT);
        if (lane_id() == 0) s.agbits[c >> 6] = m;
    }
    lds_barrier();
    if (tid < 64) {
        const int cntw = tid < p.nW ? __popcll(s.agbits[tid]) : 0;
        const int incl = wave_incl_scan(cntw);
        s.wordbase[tid] = incl - cntw;
    }
    lds_barrier();
    for (int c = tid; c < p.C; c += T)
        if (s.type[c] == RL_AGENT) {
            const int idx = s.wordbase[c >> 6] + __popcll(s.agbits[c >> 6] & lowmask(c & 63));
            const rl_u4 r = rl_philox4x32(p.seed, epoch, (uint32_t)(p.world_base + w), 0u, RL_SITE_RESET_AGENT, (uint32_t)c);
            const int gene = (int)rl_mulhi(r.y, (unsigned)p.n_brains);
            init_newborn(s, idx, c, p.W, gene, gene, idx);
            s.order[idx] = (short)idx; s.newidx[idx] = (short)idx;
        }
    if (tid < RL_N_BEST) {
        s.best_uid[tid] = -1; s.best_fit[tid] = 0.0; s.best_brain[tid] = 0;
        p.st.best_uid[(size_t)w * RL_N_BEST + tid] = -1;
        p.st.best_fit[(size_t)w * RL_N_BEST + tid] = 0.0;
        p.st.best_brain[(size_t)w * RL_N_BEST + tid] = 0;
    }
    if (tid == 0) {
        p.st.next_uid[w] = na; p.st.max_gene[w] = p.n_brains; p.st.tick[w] = 0; p.st.epoch[w] = (int)epoch;
        if (p.refill_count) atomicAdd(p.refill_count, 1);
    }
    lds_barrier();
    return na;
}

template <int T>
__global__ __launch_bounds__(T) void k_reset(const KParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    Smem s;
    carve(s, smem_raw, p.Cp, p.cap, p.hash_size);
    const int w = blockIdx.x;
    if (p.refill_threshold >= 0 && p.st.n_agents[w] >= p.refill_threshold) {  // uniform per workgroup: nothing to re-generate
        if (p.lists && rl_tidx() < 64) {
            const int32_t* br = p.st.a_brain + (size_t)w * p.cap;
            emit_brain_lists_wave0(p, w, p.st.n_agents[w], [&](int k) { return br[k]; });
        }
        return;
    }
    const uint32_t epoch = (uint32_t)p.st.epoch[w] + (p.refill_threshold >= 0 ? 1u : 0u);
    const int n = reset_world_lds<T>(p, s, w, epoch);
    rebuild_gene_counts<T>(p, s, n);
    build_planes<T>(p, s);
    lds_barrier();
    write_observations<T>(p, s, w, n, p.obs_only);
    store_world<T>(p, s, w, n);
    if (p.lists && rl_tidx() < 64) emit_brain_lists_wave0(p, w, n, [&](int k) { return s.brain[s.order[k]]; });
}

// ---------------------------------------------------------------------------------------------------------------
// k_run: n_ticks iterations of the inference loop (Helpers/trainer.py:85-99 minus learn) in ONE launch.
//
//   for every tick:  Agent.get_action for the world's agents (policy_tile, rl_policy_dev.h)  ->  Environment.step  ->
//                    update_env  ->  optional re-generation below the refill threshold
//
// A world never leaves its workgroup: the state is loaded once, every tick runs out of LDS (the same phase functions as
// k_world), the agent list is re-packed in LDS between ticks (recycle_world = what store_world + load_world do through
// HBM), and the state is stored once at the end.  Every API-visible per-tick output is still written every tick
// (state_prime / state rows, reward, done, both permutations, actions), so a tick moves the same algorithmic bytes as
// rl_policy_act + rl_tick_refill; what disappears is two launches, the world's load / store, the row lists, and the trip of
// the observation rows through the fabric to another CU (the policy reads its own world's rows back from L2, sc1).
// The 4-wave tile code runs on groups of four waves of the world's workgroup (T / 256 tiles at a time); every group executes
// the same number of workgroup barriers per round.
// ---------------------------------------------------------------------------------------------------------------
#include "rl_policy_dev.h"

constexpr int kRunMaxBrains = 8;

#ifndef RL_RUN_COHERENT
#define RL_RUN_COHERENT true
#endif
struct RunArgs {
    const float* packed[kRunMaxBrains];   // device: packed weights per brains-list entry
    float eps[kRunMaxBrains];
    float* obs[2];                        // Agent.state ping-pong: tick i reads obs[(first + i) & 1], writes the other
    int first;
    int n_ticks;
    int8_t* actions;                      // [R][cap] chosen actions of the LAST tick (API-visible)
    int debug;                            // tuning only (env RL_RUN_DEBUG): 1 = skip the policy half, 2 = skip the tick half (results WRONG)
};

struct PolSmem {
    char* group0;       // per-group block: lds_h | lds_aux | lds_part, group g at group0 + g * group_bytes
    int group_bytes;
    short* prow;        // [cap] list indices grouped by brain
    int* bstart;        // [64] first entry of brain b in prow
    int* bcnt;          // [64]
    int* tstart;        // [64] first tile of brain b
    short* trow;        // [kMaxTiles][32] list index of tile row j (| 0x8000: padding, repeats the brain's last row)
    int* tbrain;        // [kMaxTiles] brain of tile t
    int* meta;          // [8]  [0] number of tiles; loop state of k_run: [1] list length, [2] Agent.state parity, [3] ticks done,
                        //      [4] the LDS mirror holds the current Agent.state rows
    float* pairv;       // [4 tiles][32 row values | 2 x 64 partial row maxima] of the two-waves-per-tile policy (T = 512), or null
    float* cconst;      // [n_brains][3][256] epilogue constants of the brains' three 128-wide layers for policy_tile1s (T = 512), or null
    float* xmirror;     // [xrows][kXStride] Agent.state rows of this world for the one-wave policy tile (T <= 512), or null
    int xrows;
};
template <int KIND>
__host__ __device__ constexpr int policy_group_bytes()
{
    return (int)(align16(sizeof(f32x4) * (size_t)policy_lds_units(KIND)) + align16(sizeof(float) * kAuxFloats) + align16(sizeof(float) * 4 * 32 * 9));
}
// groups > 0: `groups` blocks for the 4-wave tile (T = 1024).  groups == 0: the one-wave tile needs no LDS of its own; with
// mirror_budget > 0 the Agent.state rows are mirrored in LDS instead (as many rows as fit below the budget, at most cap).
constexpr int kMaxTiles = 32;                // 32-row tiles of one brain per world: <= cap / 32 + n_brains
constexpr int kPairFloats = 32 + 2 * 64;     // per tile pair: row values, partial row maxima of the two roles
constexpr int kPairExBytes = 8 * kPlanes * 64 * 16;   // per tile pair: the split activations of the input layer (aliases the Agent.state mirror)
template <int KIND>
__host__ __device__ inline size_t carve_policy(PolSmem& ps, char* base, size_t o, int cap, int groups, size_t mirror_budget = 0, int n_cbrains = 0)
{
    o = align16(o);
    ps.group0 = base + o; ps.group_bytes = policy_group_bytes<KIND>();
    o += (size_t)groups * policy_group_bytes<KIND>();
    ps.xmirror = nullptr; ps.xrows = 0;
    const size_t tail = 6144 + sizeof(float) * 4 * kPairFloats + sizeof(float) * kTileConstFloats * (size_t)n_cbrains;   // what follows the mirror
    if (groups == 0 && mirror_budget > o + tail) {
        const size_t rows = (mirror_budget - o - tail) / (sizeof(float) * kXStride);
        ps.xrows = (int)(rows < (size_t)cap ? rows : (size_t)cap);
        if (ps.xrows >= 32) { ps.xmirror = (float*)(base + o); o = align16(o + sizeof(float) * kXStride * (size_t)ps.xrows); }
        else ps.xrows = 0;
    }
    ps.prow = (short*)(base + o); o = align16(o + sizeof(short) * (size_t)cap);
    ps.bstart = (int*)(base + o); o = align16(o + sizeof(int) * 64);
    ps.bcnt = (int*)(base + o); o = align16(o + sizeof(int) * 64);
    ps.tstart = (int*)(base + o); o = align16(o + sizeof(int) * 64);
    ps.trow = (short*)(base + o); o = align16(o + sizeof(short) * kMaxTiles * 32);
    ps.tbrain = (int*)(base + o); o = align16(o + sizeof(int) * kMaxTiles);
    ps.meta = (int*)(base + o); o = align16(o + sizeof(int) * 8);
    ps.cconst = nullptr;
    if (n_cbrains > 0) { ps.cconst = (float*)(base + o); o = align16(o + sizeof(float) * kTileConstFloats * (size_t)n_cbrains); }
    ps.pairv = nullptr;
    if (n_cbrains > 0) { ps.pairv = (float*)(base + o); o = align16(o + sizeof(float) * 4 * kPairFloats); }
    return o;
}
template <int KIND> __device__ inline f32x4* pol_h(const PolSmem& ps, int g) { return (f32x4*)(ps.group0 + g * ps.group_bytes); }
template <int KIND> __device__ inline float* pol_aux(const PolSmem& ps, int g)
{
    return (float*)(ps.group0 + g * ps.group_bytes + align16(sizeof(f32x4) * (size_t)policy_lds_units(KIND)));
}
template <int KIND> __device__ inline float (*pol_part(const PolSmem& ps, int g))[32][9]
{
    return (float (*)[32][9])(ps.group0 + g * ps.group_bytes + align16(sizeof(f32x4) * (size_t)policy_lds_units(KIND)) + align16(sizeof(float) * kAuxFloats));
}

// Agent.get_action for the n agents of this world (slot k == list index k): actions into s.action[] and the global
// `actions` buffer.  Must be called by the whole workgroup; leaves with a barrier behind the last action store.
// (Per-brain arguments are fetched from the kernel-argument block with a uniform index -- scalar loads; a by-value copy of the
// argument struct indexed at run time would live in scratch, and at 1024 threads x 256 worlds every dword of scratch per thread
// is 1 MB of memory traffic per tick.)
struct RunParams;
typedef const RunParams __attribute__((address_space(4))) RunParamsC;
template <int T, int KIND>
__device__ __forceinline__ void run_policy(const KParams& p, Smem& s, PolSmem& ps, RunParamsC* ka, int w, int n, const float* obs_rows, char* smem_base);

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
template <int T, bool SPEC>
__device__ __forceinline__ void recycle_world(const KParams& p, Smem& s, int n, int tick, int epoch, int next_uid, int max_gene, bool drain_stores, const RecycleRegs& rr)
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
        s.uid[tid] = r_u; s.flags[tid] = r_fl; s.action[tid] = r_act; s.fitness[tid] = r_f;
        s.aux[tid] = 0; s.src[tid] = (short)tid; s.order[tid] = (short)tid; s.newidx[tid] = (short)tid;
    }
    for (int c = tid; c < p.Cp; c += T) { s.occ[c] = -1; ((unsigned*)s.foodv)[c] = 0u; }
    for (int i = tid; i < p.hash_size; i += T) { s.hkey[i] = -1; s.hcnt[i] = 0u; }
    if (tid < RL_MAX_BRAINS) s.present[tid] = 0;
    if (SPEC) for (int i = tid; i < 2 * p.cap; i += T) ((unsigned*)s.reward)[i] = 0u;
    if (tid < S_COUNT)
        s.scal[tid] = tid == S_NSLOTS ? n : tid == S_TICK ? tick : tid == S_EPOCH ? epoch : tid == S_NEXT_UID ? next_uid : tid == S_MAX_GENE ? max_gene : 0;
    lds_barrier();
    if (mine) s.occ[(r_pos & 255) * p.W + (r_pos >> 8)] = (short)tid;
    if (drain_stores) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this tick's observation rows are in L2 before any wave reads them back
    lds_barrier();
}

// The kernel's only parameter.  The two halves of a tick are real (noinline) function calls, each with a register allocation
// of its own (inlined into one body the tile code's ~126 VGPRs and the tick's hoisted loop invariants spill into each other's
// loops), and the kernel body keeps NOTHING alive across them: the loop state lives in LDS (PolSmem::meta), because whatever a
// caller holds in registers across a call of a 128-VGPR callee goes through scratch.
struct RunParams {
    KParams p;
    RunArgs ra;
};
template <bool FIXED>
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
        p.W = kFixW; p.H = kFixH; p.C = kFixC; p.Cp = kFixCp; p.nW = kFixCp / 64;
        p.cap = kFixCap; p.hash_size = kFixHash; p.hash_mask = kFixHash - 1;
    }
    return p;
}
// LDS of the multi-tick kernel per workgroup size: T = 1024 runs the 4-wave policy tile (four tile blocks); T = 512 the one-wave
// tile with the Agent.state rows mirrored in what is left of the CU's 160 KB; T = 256 (several worlds per CU) the one-wave tile
// reading its rows back from L2.
constexpr size_t kRunLdsBudget = 160 * 1024;
__host__ __device__ constexpr int run_groups(int T) { return T == 1024 ? 4 : 0; }
__host__ __device__ constexpr size_t run_mirror_budget(int T) { return T == 512 ? kRunLdsBudget : 0; }
__host__ __device__ constexpr int run_cbrains(int T, int n_brains) { return T == 512 ? n_brains : 0; }   // the hand-scheduled tile keeps its epilogue constants in LDS
template <bool FIXED, int KIND>
__device__ inline void run_carve(const KParams& p, Smem& s, PolSmem& ps, char* smem_raw, int T)
{
    const size_t o0 = FIXED ? carve(s, smem_raw, kFixCp, kFixCap, kFixHash) : carve(s, smem_raw, p.Cp, p.cap, p.hash_size);
    carve_policy<KIND>(ps, smem_raw, o0, p.cap, run_groups(T), run_mirror_budget(T), run_cbrains(T, p.n_brains));
}

template <int T, int KIND>
__device__ __forceinline__ void run_policy(const KParams& p, Smem& s, PolSmem& ps, RunParamsC* ka, int w, int n, const float* obs_rows, char* smem_base)
{
    constexpr int GROUPS = T / 256;
    const int tid = rl_tidx(), lane = tid & 63, wave = tid >> 6, grp = wave >> 2, v = wave & 3, j = lane & 31;
#ifdef RL_PHASE_PROFILE
    if (p.prof && (int)blockIdx.x == p.prof_world && rl_tidx() == 0) p.prof[100] = (long long)clock64();
#endif
    if (wave == 0) {   // rows grouped by brain (ballots; lane b keeps brain b's count), tiles of 32 rows per brain
        int cnt = 0;
        for (int base = 0; base < n; base += 64) {
            const int k = base + lane;
            const int b = k < n ? s.brain[k] : -1;
            for (int bb = 0; bb < p.n_brains; ++bb) { const int c = __popcll(__ballot(b == bb)); if (lane == bb) cnt += c; }
        }
        const int mine = lane < p.n_brains ? cnt : 0;
        const int incl = wave_incl_scan(mine);
        const int tiles = (mine + 31) >> 5;
        const int tincl = wave_incl_scan(tiles);
        if (lane < p.n_brains) { ps.bstart[lane] = incl - mine; ps.bcnt[lane] = mine; ps.tstart[lane] = tincl - tiles; }
        if (lane == 63) ps.meta[0] = tincl;
        int pos = incl - mine;
        for (int base = 0; base < n; base += 64) {
            const int k = base + lane;
            const int b = k < n ? s.brain[k] : -1;
            for (int bb = 0; bb < p.n_brains; ++bb) {
                const unsigned long long m = __ballot(b == bb);
                const int start = read_lane(pos, bb);
                if (b == bb) ps.prow[start + __popcll(m & lowmask(lane))] = (short)k;
                if (lane == bb) pos += __popcll(m);
            }
        }
    }
    lds_barrier();
    int ntiles = __builtin_amdgcn_readfirstlane(ps.meta[0]);
    {   // tuning only (RL_RUN_DEBUG & 4 / & 8): run at most 2 / 1 tiles (results WRONG)
        const int dbg = *(const int __attribute__((address_space(4)))*)&ka->ra.debug;
        if (dbg & 4) ntiles = min(ntiles, 2);
        if (dbg & 8) ntiles = min(ntiles, 1);
    }
    for (int t0 = 0; t0 < ntiles; t0 += GROUPS) {   // uniform trip count: every group runs the same barriers
        const int ti = t0 + grp;
        const bool have = ti < ntiles;
        int b = 0;
        if (have) for (int bb = 1; bb < p.n_brains; ++bb) if (ps.tstart[bb] <= ti && ps.bcnt[bb] > 0) b = bb;
        b = __builtin_amdgcn_readfirstlane(b);   // uniform per wave: the brain's arguments come by scalar loads
        const int cntb = have ? ps.bcnt[b] : 0;
        const int li = have ? (ti - ps.tstart[b]) * 32 + j : 0;
        const bool valid = have && li < cntb;
        const int k = (have && cntb > 0) ? ps.prow[ps.bstart[b] + min(li, cntb - 1)] : 0;
        TileIO io;
        io.packed = (gfloat*)((const float* const __attribute__((address_space(4)))*)ka->ra.packed)[b];
        io.obs = obs_rows;
        io.row = (int64_t)w * p.cap + k;
        io.valid = valid;
        io.eps = ((const float __attribute__((address_space(4)))*)ka->ra.eps)[b];
        io.out = nullptr;
        io.actions = *(int8_t* const __attribute__((address_space(4)))*)&ka->ra.actions;
        io.seed = p.seed;
        io.key_world = (uint32_t)(p.world_base + w); io.key_tick = (uint32_t)s.scal[S_TICK]; io.key_epoch = (uint32_t)s.scal[S_EPOCH];
        io.key_index = (uint32_t)k;
        io.lds_actions_off = (int)((char*)s.action - smem_base); io.lds_slot = k;
        io.x_lds_off = -1; io.c_lds_off = -1;
#ifdef RL_PHASE_PROFILE
        io.prof = (p.prof && (int)blockIdx.x == p.prof_world) ? p.prof : nullptr;
        if (io.prof && rl_tidx() == 0) io.prof[101] = (long long)clock64();
#endif
        policy_tile<KIND, false, RL_RUN_COHERENT>(io, pol_h<KIND>(ps, grp), pol_aux<KIND>(ps, grp), pol_part<KIND>(ps, grp), lane, v);
    }
}

// Rows of a world grouped by brain, 32-row tiles per brain, by ONE wave (ballots; lane b keeps brain b's count): prow / bstart / bcnt /
// tstart / meta[0].  `brain_of(k)`: brains-list index of list entry k.
template <typename F>
__device__ inline void policy_lists_wave0(const KParams& p, PolSmem& ps, int n, int lane, F brain_of)
{
    int cnt = 0;
    for (int base = 0; base < n; base += 64) {
        const int k = base + lane;
        const int b = k < n ? brain_of(k) : -1;
        for (int bb = 0; bb < p.n_brains; ++bb) { const int c = __popcll(__ballot(b == bb)); if (lane == bb) cnt += c; }
    }
    const int mine = lane < p.n_brains ? cnt : 0;
    const int incl = wave_incl_scan(mine);
    const int tiles = (mine + 31) >> 5;
    const int tincl = wave_incl_scan(tiles);
    if (lane < p.n_brains) { ps.bstart[lane] = incl - mine; ps.bcnt[lane] = mine; ps.tstart[lane] = tincl - tiles; }
    if (lane == 63) ps.meta[0] = tincl;
    int pos = incl - mine;
    for (int base = 0; base < n; base += 64) {
        const int k = base + lane;
        const int b = k < n ? brain_of(k) : -1;
        for (int bb = 0; bb < p.n_brains; ++bb) {
            const unsigned long long m = __ballot(b == bb);
            const int start = read_lane(pos, bb);
            if (b == bb) ps.prow[start + __popcll(m & lowmask(lane))] = (short)k;
            if (lane == bb) pos += __popcll(m);
        }
    }
    // tile descriptors: one LDS read per lane instead of a dependent chain of four at the start of every policy half
    const int tt = min(read_lane(tincl, 63), kMaxTiles);
    const int first_tile = tincl - tiles, first_row = incl - mine;
    for (int e = lane; e < tt * 32; e += 64) {
        const int t = e >> 5, j = e & 31;
        int b = 0;
        for (int bb = 1; bb < p.n_brains; ++bb) if (read_lane(first_tile, bb) <= t && read_lane(mine, bb) > 0) b = bb;
        const int cntb = __shfl(mine, b), li = (t - __shfl(first_tile, b)) * 32 + j;
        const int k = ps.prow[__shfl(first_row, b) + min(li, cntb - 1)];
        ps.trow[e] = (short)(k | (li < cntb ? 0 : 0x8000));
        if (j == 0) ps.tbrain[t] = b;
    }
}

// The policy half for workgroups of at most 512 threads (the tiles below need the 256-VGPR budget).  T = 512, up to four tiles: TWO waves per
// tile on one SIMD (policy_tile1s<PAIR>); five to eight tiles: one hand-scheduled tile per wave (policy_tile1s); T = 256: one
// policy_tile1 per wave (no LDS, no barrier inside a tile), wave i takes tiles i, i + 4, ...  Tile rows, validity and brain come from the
// descriptors wave 0 wrote next to the row lists (policy_lists_wave0).
template <int T, int KIND>
__device__ __forceinline__ void run_policy1(const KParams& p, Smem& s, PolSmem& ps, RunParamsC* ka, int w, int n, const float* obs_rows, char* smem_base, int wave)
{
    // (`wave`: the wave's index in the workgroup, uniform -- computed once per launch and kept in an SGPR)
    const int lane = rl_lane_fresh(), tid = wave * 64 + lane, j = lane & 31;
#ifdef RL_PHASE_PROFILE
    const long long t_entry = (long long)clock64();
#endif
    // (the per-brain row lists were built by wave 0 while the other waves wrote the previous tick's Agent.state rows: policy_lists_wave0)
    int ntiles = __builtin_amdgcn_readfirstlane(ps.meta[0]);
    {   // tuning only (RL_RUN_DEBUG & 4 / & 8): run at most 2 / 1 tiles (results WRONG)
        const int dbg = *(const int __attribute__((address_space(4)))*)&ka->ra.debug;
        if (dbg & 4) ntiles = min(ntiles, 2);
        if (dbg & 8) ntiles = min(ntiles, 1);
        if (dbg & 16) ntiles = min(ntiles, 3);
    }
    const bool mirrored = ps.xmirror != nullptr && __builtin_amdgcn_readfirstlane(ps.meta[4]) != 0;
    auto tile_io = [&](int ti, TileIO& io) {
        const int b = __builtin_amdgcn_readfirstlane(ps.tbrain[ti]);
        const int e = (unsigned short)ps.trow[ti * 32 + j], k = e & 0x7fff;
        io.packed = (gfloat*)((const float* const __attribute__((address_space(4)))*)ka->ra.packed)[b];
        io.obs = obs_rows;
        io.row = (int64_t)w * p.cap + k;
        io.valid = !(e & 0x8000);
        io.eps = ((const float __attribute__((address_space(4)))*)ka->ra.eps)[b];
        io.out = nullptr;
        io.actions = *(int8_t* const __attribute__((address_space(4)))*)&ka->ra.actions;
        io.seed = p.seed;
        io.key_world = (uint32_t)(p.world_base + w); io.key_tick = (uint32_t)s.scal[S_TICK]; io.key_epoch = (uint32_t)s.scal[S_EPOCH];
        io.key_index = (uint32_t)k;
        io.lds_actions_off = (int)((char*)s.action - smem_base); io.lds_slot = k;
        io.x_lds_off = (mirrored && k < ps.xrows) ? (int)((char*)(ps.xmirror + k * kXStride) - smem_base) : -1;
        io.c_lds_off = ps.cconst ? (int)((char*)(ps.cconst + kTileConstFloats * b) - smem_base) : -1;
#ifdef RL_PHASE_PROFILE
        io.prof = (p.prof && (int)blockIdx.x == p.prof_world && wave == 0) ? p.prof : nullptr;
        if (io.prof && lane == 0) { io.prof[100] = t_entry; io.prof[110] = (long long)clock64(); }
#endif
    };
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
            tile_io(slot, io);
            policy_tile1s<KIND, RL_RUN_COHERENT, true>(io, lane, role, &pl, &part);
        } else { lds_barrier(); lds_barrier(); }   // (the two exchanges inside the tile)
#ifdef RL_PHASE_PROFILE
        if (p.prof && (int)blockIdx.x == p.prof_world && lane == 0) p.prof[116 + wave] = (long long)clock64();   // (128 slots)
#endif
        lds_barrier();
        if (have && role == 0) tile1_finish<KIND>(io, lane, part.head, pl.val[j], part.draw, *(const f32x4*)((const float*)(smem_base + io.c_lds_off) + 768 + 8 + 4 * (lane >> 5)));
    } else
    for (int ti = wave; ti < ntiles; ti += T / 64) {
        TileIO io;
        tile_io(ti, io);
        if (T == 512) policy_tile1s<KIND, RL_RUN_COHERENT>(io, lane);
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

// First half of a tick: the policy.  Reads the list length and the Agent.state parity from LDS.
template <int T, bool FIXED, int KIND>
__device__ __forceinline__ void run_policy_half(RunParamsC* ka, int wave)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const KParams p = run_params<FIXED>(ka);
    Smem s;
    PolSmem ps;
    run_carve<FIXED, KIND>(p, s, ps, smem_raw, T);
    const int w = blockIdx.x;
    const int n = __builtin_amdgcn_readfirstlane(ps.meta[1]), cur = __builtin_amdgcn_readfirstlane(ps.meta[2]);
    const float* obs_in = ((float* const __attribute__((address_space(4)))*)ka->ra.obs)[cur];
    if (T <= 512) run_policy1<T, KIND>(p, s, ps, ka, w, n, obs_in, smem_raw, wave);
    else run_policy<T, KIND>(p, s, ps, ka, w, n, obs_in, smem_raw);
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
            const int c = (ps & 255) * p.W + (ps >> 8);
            s.foodv[c] = 0.5f; s.healthv[c] = -1.f; s.genev[c] = -2;
            cell0 |= c == 0;
        }
    }
    for (int i = first_new + tid; i < end_new; i += T) {
        const int c = s.tgt[i];
        s.foodv[c] = 0.f; s.healthv[c] = 1.f; s.genev[c] = -2;
        cell0 |= c == 0;
    }
    if (cell0) s.scal[S_PLANES_DIRTY] = 1;
}

// Second half: Environment.step + update_env (+ re-generation) out of LDS, then recycle_world.  Same sequence as
// k_world<T, MODE_TICK, LEAN>; writes Agent.state into ra.obs[cur ^ 1] and advances the loop state in LDS.
template <int T, bool FIXED, int KIND>
__device__ __forceinline__ void run_tick_body(RunParamsC* ka)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const KParams p = run_params<FIXED>(ka);
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
    if (kPlanesEarly) patch_placed_planes(s);
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
    if (overlapped) {
        if (tid < 64) {
            if (!p.static_families) best_agents_wave(s, n1);
            reproduce_wave0<T, true>(p, s, w, n1, nslots);
        } else {
            write_observations<(T > 64 ? T - 64 : 64)>(p, s, w, n1, p.so.obs, tid - 64);
            step_outputs(tid - 64, T - 64);
        }
    } else {
        write_observations<T>(p, s, w, n1, p.so.obs);
        step_outputs(tid, T);
    }
    lds_barrier();
    RL_MARK(63);
    for (int a = tid; a < nslots; a += T) s.src[a] = s.newidx[a];
    if (!overlapped) lds_barrier();
    else { const int first_new = nslots; nslots = s.scal[S_NSLOTS]; patch_planes_after_update<T>(p, s, n1, first_new, nslots); }
    if (!overlapped) phase_update<T, true>(p, s, w, n1, nslots, true);
    for (int c = tid; c < p.Cp; c += T) {
        const unsigned long long m = __ballot(s.type[c] == RL_AGENT);
        if (lane_id() == 0) s.agbits[c >> 6] = m;
    }
    for (int i = tid; i < p.hash_size; i += T) { s.hkey[i] = -1; s.hcnt[i] = 0u; }
    lds_barrier();
    RL_MARK(64);
    if (tid < 64) scan_order_wave(p, s, tid, S_N2);
    lds_barrier();
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
    if (T <= 512) {   // wave 0 prepares the next tick's policy (rows grouped by brain) while the others write the Agent.state rows
        if (tid < 64) policy_lists_wave0(p, ps, n2, tid, [&](int k) { return s.brain[s.order[k]]; });
        else write_observations<(T > 64 ? T - 64 : 64)>(p, s, w, n2, obs_out, tid - 64, ps.xmirror, ps.xrows);
    } else
        write_observations<T>(p, s, w, n2, obs_out, ps.xmirror, ps.xrows);
    RL_MARK(68);
    if (tid == 0) { ps.meta[1] = n2; ps.meta[2] = cur ^ 1; ps.meta[3] = ticks_done + 1; ps.meta[4] = ps.xmirror != nullptr; }
    // (rows the policy will read back from memory must have reached L2 first; with every row mirrored in LDS the stores just drain)
    recycle_world<T, kSpec>(p, s, n2, tick_next, epoch_next, next_uid, max_gene, ps.xmirror == nullptr || n2 > ps.xrows, rr);
    RL_MARK(69);
}

template <int T, int KIND>
__device__ __forceinline__ void run_tick_call_fixed(RunParamsC* ka) { run_tick_body<T, true, KIND>(ka); }
template <int T, int KIND>
__device__ __forceinline__ void run_tick_call_generic(RunParamsC* ka) { run_tick_body<T, false, KIND>(ka); }

template <int T, bool FIXED, int KIND>
__device__ __forceinline__ void run_load_call(RunParamsC* ka)   // (inlined: load_world reads the kernel-argument segment through the intrinsic)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    typedef const int __attribute__((address_space(4))) cint;
    const KParams p = run_params<FIXED>(ka);
    Smem s;
    PolSmem ps;
    run_carve<FIXED, KIND>(p, s, ps, smem_raw, T);
    int n0;
    load_world<T, (T == 1024)>(p, s, (int)blockIdx.x, n0);
    // The first tick's policy input: rows written by the previous launch (or the reset / observe call).  With the mirror they are copied
    // into LDS here, by the whole workgroup with coalesced loads, instead of being fetched row by row by the tile waves (the row phase of
    // a launch's first tick: 12.2k cycles against 3.5k from the mirror).
    const int first = *(cint*)&ka->ra.first;
    const bool preload = ps.xmirror != nullptr && n0 <= ps.xrows;
    if (preload) {
        const float* rows = ((float* const __attribute__((address_space(4)))*)ka->ra.obs)[first] + (size_t)blockIdx.x * p.cap * RL_OBS_DIM;
        for (int i = rl_tidx(); i < n0 * RL_OBS_DIM; i += T) {
            const int r = i / RL_OBS_DIM;
            ps.xmirror[r * kXStride + (i - r * RL_OBS_DIM)] = rows[i];
        }
    }
    if (rl_tidx() == 0) { ps.meta[1] = n0; ps.meta[2] = first; ps.meta[3] = 0; ps.meta[4] = preload ? 1 : 0; }
    if (T <= 512 && rl_tidx() < 64) policy_lists_wave0(p, ps, n0, rl_tidx(), [&](int k) { return s.brain[k]; });   // (slot == list index after load_world)
    if (ps.cconst) {   // the brains' epilogue constants (three 128-wide layers x 256 floats) for policy_tile1s
        const Layout L = layout_of(KIND);
        for (int i = rl_tidx(); i < kTileConstFloats * p.n_brains; i += T) {
            const int b = i / kTileConstFloats, j = i - kTileConstFloats * b, layer = j >> 8;
            gfloat* pk = (gfloat*)((const float* const __attribute__((address_space(4)))*)ka->ra.packed)[b];
            const int64_t off = layer == 0 ? L.l1 + frag_floats(kInChunks, 4) : layer == 1 ? L.l2a + frag_floats(8, 4) : layer == 2 ? L.l2b + frag_floats(8, 4)
                              : (j < 768 + 16 ? L.ha : L.hb) + head_consts_off(4) - (j < 768 + 16 ? 768 : 768 + 16);
            ps.cconst[i] = pk[off + (layer < 3 ? (j & 255) : j)];
        }
    }
    lds_barrier();
}
template <int T, bool FIXED, int KIND>
__device__ __forceinline__ void run_store_call(RunParamsC* ka)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const KParams p = run_params<FIXED>(ka);
    Smem s;
    PolSmem ps;
    run_carve<FIXED, KIND>(p, s, ps, smem_raw, T);
    store_world<T>(p, s, (int)blockIdx.x, ps.meta[1]);
    if (rl_tidx() == 0) { p.st.tick[blockIdx.x] = s.scal[S_TICK]; p.st.epoch[blockIdx.x] = s.scal[S_EPOCH]; p.st.next_uid[blockIdx.x] = s.scal[S_NEXT_UID]; p.st.max_gene[blockIdx.x] = s.scal[S_MAX_GENE]; }
}

// Kernel body = the policy half (inlined: a kernel saves no registers, and the tile code gets the 128-VGPR budget of a
// 1024-thread workgroup to itself); the tick half, the initial load and the final store are callees.
template <int T, bool FIXED, int KIND>
__global__ __launch_bounds__(T) void k_run(const RunParams rp)
{
    RunParamsC* ka = (RunParamsC*)__builtin_amdgcn_kernarg_segment_ptr();
    typedef const int __attribute__((address_space(4))) cint;
    const int n_ticks = *(cint*)&ka->ra.n_ticks;
    const int dbg = *(cint*)&ka->ra.debug;
    // Workgroups that start in exact lockstep stay in lockstep for tens of ticks (every world does the same work at the same moment:
    // 256 CUs ask L2 for the same weight lines, then all write their observation rows), and such ticks are ~3 us slower than those of
    // drifted-apart worlds: kernel time of a 20-tick launch 545 -> 518 us with the starts spread over 3.75 us (16 steps of 0.25 us; 0.5 / 1 us
    // steps, 32 or 64 groups give the same), 100- and 500-tick launches unchanged.  RL_RUN_DEBUG & 32 switches it off (measurements).
    if (!(dbg & 32)) {
        const long long until = (long long)clock64() + (long long)(blockIdx.x & 15) * 500;
        while ((long long)clock64() < until) __builtin_amdgcn_s_sleep(2);
    }
    run_load_call<T, FIXED, KIND>(ka);
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    for (int it = 0; it < n_ticks; ++it) {   // (`it` and the bounds are uniform: SGPRs, which a callee preserves)
        if (!(dbg & 1)) run_policy_half<T, FIXED, KIND>(ka, wave);
        if (!(dbg & 2)) {
            if constexpr (FIXED) run_tick_call_fixed<T, KIND>(ka);
            else run_tick_call_generic<T, KIND>(ka);
        }
    }
    run_store_call<T, FIXED, KIND>(ka);
}

// 1024 threads per world when there are few worlds (latency-bound: one world per CU), 256 when there are many
// (throughput-bound: several worlds per CU hide each other's barriers).
inline int pick_block(const rl_world* h)
{
    const char* env = getenv("RL_WORLD_BLOCK");  // read at every launch (like RL_WORLD_GENERIC): tests and A/B runs switch it
    const int forced = env ? atoi(env) : 0;
    if (forced == 256 || forced == 512 || forced == 1024) return forced;
    return h->cfg.n_worlds <= 768 ? 1024 : 256;
}

KParams make_params(const rl_world* h)
{
    KParams p{};
    p.W = h->cfg.width; p.H = h->cfg.height; p.C = h->cells; p.Cp = h->cpad; p.nW = h->cpad / 64;
    p.cap = h->cfg.slot_cap; p.max_agents = h->cfg.max_agents; p.n_brains = h->cfg.n_brains;
    p.hash_size = h->hash_size; p.hash_mask = h->hash_size - 1; p.world_base = h->cfg.world_base;
    p.static_families = h->cfg.static_families; p.limit_reproduction = h->cfg.limit_reproduction;
    p.incentivize_killing = h->cfg.incentivize_killing;
    p.seed = h->cfg.seed;
    p.st = h->st;
    p.err = h->err_flag;
    p.refill_threshold = -1;
    p.prof = h->prof; p.prof_world = h->prof_world;
    p.lists = nullptr; p.lists_counts = nullptr; p.lists_counts_zero = nullptr; p.list_stride = 0;
    p.ablate = g_ablate;
    return p;
}

template <int MODE, bool LEAN>
int launch_world_v(const rl_world* h, const KParams& p, hipStream_t stream)
{
    const int blk = pick_block(h);
    static const size_t fixed_bytes = rl_world_smem_bytes(kFixCp, kFixCap, kFixHash);  // inside the default 64 KB window
    const bool fixed = LEAN && MODE == MODE_TICK && p.W == kFixW && p.H == kFixH && p.cap == kFixCap && p.hash_size == kFixHash &&
                       fixed_bytes <= 64 * 1024 && !getenv("RL_WORLD_GENERIC");  // (env: run the generic code -- tests, A/B)
    const dim3 grid(h->cfg.n_worlds);
    if (fixed) {
        if (blk == 1024) hipLaunchKernelGGL((k_world<1024, MODE_TICK, true, true>), grid, dim3(1024), fixed_bytes, stream, p);
        else if (blk == 512) hipLaunchKernelGGL((k_world<512, MODE_TICK, true, true>), grid, dim3(512), fixed_bytes, stream, p);
        else hipLaunchKernelGGL((k_world<256, MODE_TICK, true, true>), grid, dim3(256), fixed_bytes, stream, p);
    } else if (blk == 1024)
        hipLaunchKernelGGL((k_world<1024, MODE, LEAN>), grid, dim3(1024), h->smem_bytes, stream, p);
    else if (blk == 512)
        hipLaunchKernelGGL((k_world<512, MODE, LEAN>), grid, dim3(512), h->smem_bytes, stream, p);
    else
        hipLaunchKernelGGL((k_world<256, MODE, LEAN>), grid, dim3(256), h->smem_bytes, stream, p);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { rl_set_error("world kernel launch failed: %s", hipGetErrorString(e)); return RL_E_LAUNCH; }
    return RL_OK;
}

template <int MODE>
int launch_world(const rl_world* h, const KParams& p, hipStream_t stream)
{
    if (int rc = rl_world_prepare_bytes(h->smem_bytes)) return rc;
    const bool lean = MODE == MODE_TICK && !p.tape.food_k && !p.so.trk_tick && !p.so.age && !p.so.brain && !p.so.n_post;
    return lean ? launch_world_v<MODE, true>(h, p, stream) : launch_world_v<MODE, false>(h, p, stream);
}

// trainer.py:95-96 + entities.py:194-208 for one world per workgroup: wave 0 reserves ring slots per brain (ballots, one
// atomic per brain), then the four waves copy the two 153-float rows of every transition.
struct CaptureArgs {
    rl_replay rp[RL_MAX_CAPTURE_BRAINS];
    int n_brains, cap;
    const float* state; const int8_t* actions; const float* policy_out;
    rl_step_out so;
};

__global__ __launch_bounds__(256) void k_capture(const CaptureArgs A)
{
    __shared__ int slot[4096];
    const int w = blockIdx.x, tid = rl_tidx(), lane = tid & 63;
    const int n1 = A.so.n_post[w];
    const size_t b = (size_t)w * A.cap;
    if (tid < 64) {
        int cnt = 0;
        for (int base = 0; base < n1; base += 64) {
            const int k = base + lane;
            const bool act = k < n1 && A.so.age[b + k] > 1;  // Agent.learn: `if self.age > 1` (entities.py:196)
            const int br = act ? A.so.brain[b + k] : -1;
            for (int bb = 0; bb < A.n_brains; ++bb) { const int c = __popcll(__ballot(br == bb)); if (lane == bb) cnt += c; }
        }
        unsigned long long pos = 0;
        if (lane < A.n_brains && cnt) pos = atomicAdd(A.rp[lane].count, (unsigned long long)cnt);
        for (int base = 0; base < n1; base += 64) {
            const int k = base + lane;
            const bool act = k < n1 && A.so.age[b + k] > 1;
            const int br = act ? A.so.brain[b + k] : -1;
            int myslot = -1;
            for (int bb = 0; bb < A.n_brains; ++bb) {
                const unsigned long long m = __ballot(br == bb);
                const unsigned long long start = read_lane_u64(pos, bb);
                if (br == bb) myslot = (int)((start + (unsigned long long)__popcll(m & lowmask(lane))) % (unsigned long long)A.rp[bb].capacity);
                if (lane == bb) pos += (unsigned long long)__popcll(m);
            }
            if (k < n1) slot[k] = myslot;
        }
    }
    lds_barrier();
    for (int k = tid >> 6; k < n1; k += 4) {  // one wave per transition
        const int sl = slot[k];
        if (sl < 0) continue;
        const rl_replay R = A.rp[A.so.brain[b + k]];
        const int src = A.so.src[b + k];
        const float* s0 = A.state + (b + src) * RL_OBS_DIM;
        const float* s1 = A.so.obs + (b + k) * RL_OBS_DIM;
        float* d0 = R.state + (size_t)sl * RL_OBS_DIM;
        float* d1 = R.state_prime + (size_t)sl * RL_OBS_DIM;
        for (int f = lane; f < RL_OBS_DIM; f += 64) { d0[f] = s0[f]; d1[f] = s1[f]; }
        if (lane == 0) {
            const int a = A.actions[b + src];
            R.action[sl] = (int8_t)a;
            R.reward[sl] = A.so.reward[b + k];
            R.done[sl] = A.so.done[b + k];
            R.age[sl] = A.so.age[b + k];
            if (A.policy_out && R.prob && a >= 0 && a < 8) R.prob[sl] = A.policy_out[(b + src) * 8 + a];
        }
    }
}

// A launch that leaves every world policy-ready (tick / update / reset / refill) also produces the per-brain row lists
// when a policy work buffer is bound; any other launch invalidates them.
void set_list_production(rl_world* h, KParams& p, bool produces)
{
    if (produces && h->work) {
        const int cur = h->parity_next & 1;
        int* counts = (int*)h->work;
        p.lists_counts = counts + 64 * cur;
        p.lists_counts_zero = counts + 64 * (cur ^ 1);
        p.lists = counts + 128;
        p.list_stride = (long long)h->cfg.n_worlds * h->cfg.slot_cap;
        h->lists_valid = 1; h->lists_parity = cur; h->parity_next ^= 1;
    } else {
        h->lists_valid = 0;
    }
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------
// host entry points used by rl_capi.hip
// ---------------------------------------------------------------------------------------------------------------
size_t rl_world_smem_bytes(int cpad, int cap, int hash)
{
    Smem s;
    return carve(s, nullptr, cpad, cap, hash);
}
int rl_world_block() { return 1024; }

int rl_world_prepare_bytes(size_t bytes)
{
    // worlds that need more than the default 64 KB dynamic-LDS window opt in (160 KB per CU on gfx950); the attribute
    // belongs to the device's copy of the kernel, so the grant is remembered per device
    constexpr int kMaxDev = 64;
    static size_t granted_dev[kMaxDev];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) dev = 0;
    size_t& granted = granted_dev[dev];
    if (granted < 64 * 1024) granted = 64 * 1024;
    if (bytes <= granted) return RL_OK;
    hipError_t e = hipSuccess;
#define RL_ATTR(K) e = e != hipSuccess ? e : hipFuncSetAttribute((const void*)(K), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
#define RL_ATTR_T(T) RL_ATTR((k_world<T, MODE_STEP, false>)) RL_ATTR((k_world<T, MODE_UPDATE, false>)) RL_ATTR((k_world<T, MODE_TICK, false>)) \
    RL_ATTR((k_world<T, MODE_TICK, true>)) RL_ATTR((k_world<T, MODE_OBSERVE, false>)) RL_ATTR((k_world<T, MODE_FOOD, false>)) RL_ATTR((k_reset<T>))
    RL_ATTR_T(256) RL_ATTR_T(512) RL_ATTR_T(1024)
#undef RL_ATTR_T
#undef RL_ATTR
    if (e != hipSuccess) { rl_set_error("hipFuncSetAttribute(%zu bytes of LDS) failed: %s", bytes, hipGetErrorString(e)); return RL_E_LAUNCH; }
    granted = bytes;
    return RL_OK;
}

int rl_world_launch_step(rl_world* h, const int8_t* actions, const rl_tape* tape, const rl_step_out* out, hipStream_t st)
{
    KParams p = make_params(h);
    set_list_production(h, p, false);
    p.actions = actions;
    if (tape) p.tape = *tape;
    if (out) p.so = *out;
    return launch_world<MODE_STEP>(h, p, st);
}
int rl_world_launch_step_split(rl_world* h, const int8_t* actions, const rl_step_out* out, int32_t* pre_counts, hipStream_t st)
{
    KParams p = make_params(h);
    set_list_production(h, p, false);
    p.actions = actions;
    if (out) p.so = *out;
    p.split_food = 1; p.pre_counts = pre_counts;
    return launch_world<MODE_STEP>(h, p, st);
}
int rl_world_launch_step_food(rl_world* h, const rl_tape* tape, float* obs, hipStream_t st)
{
    KParams p = make_params(h);
    set_list_production(h, p, false);
    if (tape) p.tape = *tape;
    p.so.obs = obs;
    return launch_world<MODE_FOOD>(h, p, st);
}
int rl_world_launch_update(rl_world* h, const rl_tape* tape, const rl_update_out* out, hipStream_t st)
{
    KParams p = make_params(h);
    set_list_production(h, p, true);
    if (tape) p.tape = *tape;
    if (out) p.uo = *out;
    return launch_world<MODE_UPDATE>(h, p, st);
}
int rl_world_launch_tick(rl_world* h, const int8_t* actions, const rl_tape* tape, const rl_step_out* so,
                         const rl_update_out* uo, int refill_threshold, int refill_n_agents, int32_t* refill_count, hipStream_t st)
{
    KParams p = make_params(h);
    set_list_production(h, p, true);
    p.actions = actions;
    if (tape) p.tape = *tape;
    if (so) p.so = *so;
    if (uo) p.uo = *uo;
    p.refill_threshold = refill_threshold; p.reset_n_agents = refill_n_agents; p.refill_count = refill_count;
    return launch_world<MODE_TICK>(h, p, st);
}
int rl_world_launch_observe(const rl_world* h, float* obs, hipStream_t st)
{
    KParams p = make_params(h);
    p.obs_only = obs;
    return launch_world<MODE_OBSERVE>(h, p, st);
}
int rl_world_launch_reset(rl_world* h, int n_agents, int threshold, float* obs, int32_t* refill_count, hipStream_t st)
{
    KParams p = make_params(h);
    set_list_production(h, p, true);
    p.reset_n_agents = n_agents; p.refill_threshold = threshold; p.obs_only = obs; p.refill_count = refill_count;
    if (int rc = rl_world_prepare_bytes(h->smem_bytes)) return rc;
    const int blk = pick_block(h);
    if (blk == 1024) hipLaunchKernelGGL((k_reset<1024>), dim3(h->cfg.n_worlds), dim3(1024), h->smem_bytes, st, p);
    else if (blk == 512) hipLaunchKernelGGL((k_reset<512>), dim3(h->cfg.n_worlds), dim3(512), h->smem_bytes, st, p);
    else hipLaunchKernelGGL((k_reset<256>), dim3(h->cfg.n_worlds), dim3(256), h->smem_bytes, st, p);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { rl_set_error("reset kernel launch failed: %s", hipGetErrorString(e)); return RL_E_LAUNCH; }
    return RL_OK;
}

// rl_run: which kernel serves this handle / these brains, or 0
static int run_kind_of(const rl_brain* brains, int n_brains)
{
    if (n_brains < 1 || n_brains > kRunMaxBrains) return -1;
    bool duel = true;
    for (int b = 0; b < n_brains; ++b) duel = duel && (brains[b].kind == RL_D3QN || brains[b].kind == RL_PERD3QN);
    return duel ? RL_PERD3QN : -1;   // (D3QN and PERD3QN are the same network: PERD3QN.py:186-202, D3QN.py:149-165)
}
template <int KIND>
static size_t run_smem_bytes(const rl_world* h, int T)
{
    PolSmem ps;
    return carve_policy<KIND>(ps, nullptr, h->smem_bytes, h->cfg.slot_cap, run_groups(T), run_mirror_budget(T), run_cbrains(T, h->cfg.n_brains));
}
// Workgroup size of the multi-tick kernel: 512 threads for few worlds (the one-wave policy tile needs the 256-VGPR budget; the
// tick half alone would prefer 1024: 10.6 vs 13.1 us at 256 worlds), 256 when there are many worlds (several per CU).
// RL_WORLD_BLOCK overrides it like for the other world kernels.
static int run_block(const rl_world* h)
{
    const char* env = getenv("RL_WORLD_BLOCK");
    const int forced = env ? atoi(env) : 0;
    if (forced == 256 || forced == 512 || forced == 1024) return forced;
    return h->cfg.n_worlds <= 768 ? 512 : 256;
}
int rl_world_run_supported(const rl_world* h, const rl_brain* brains, int n_brains)
{
    if (run_kind_of(brains, n_brains) < 0) return 0;
    const int T = run_block(h);
    if (h->cfg.slot_cap > T) return 0;
    return run_smem_bytes<RL_PERD3QN>(h, T) <= 160 * 1024;
}
int rl_world_launch_run(rl_world* h, const rl_brain* brains, int n_brains, int n_ticks, int8_t* actions, const rl_step_out* so,
                        float* const obs[2], int first, int16_t* upd_src, int refill_threshold, int refill_n_agents,
                        int32_t* refill_count, hipStream_t st)
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
    for (int b = 0; b < n_brains; ++b) { ra.packed[b] = brains[b].packed; ra.eps[b] = brains[b].epsilon; }
    ra.obs[0] = obs[0]; ra.obs[1] = obs[1]; ra.first = first; ra.n_ticks = n_ticks; ra.actions = actions;
    ra.debug = getenv("RL_RUN_DEBUG") ? atoi(getenv("RL_RUN_DEBUG")) : 0;
    const int T = run_block(h);
    const size_t bytes = run_smem_bytes<RL_PERD3QN>(h, T);
    const bool fixed = p.W == kFixW && p.H == kFixH && p.cap == kFixCap && p.hash_size == kFixHash && !getenv("RL_WORLD_GENERIC");
    const void* fn = T == 1024 ? (fixed ? (const void*)k_run<1024, true, RL_PERD3QN> : (const void*)k_run<1024, false, RL_PERD3QN>)
                   : T == 512 ? (fixed ? (const void*)k_run<512, true, RL_PERD3QN> : (const void*)k_run<512, false, RL_PERD3QN>)
                              : (fixed ? (const void*)k_run<256, true, RL_PERD3QN> : (const void*)k_run<256, false, RL_PERD3QN>);
    if (bytes > 64 * 1024) {   // opt in to the large dynamic-LDS window (per device copy of the kernel: cheap, done every launch)
        const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) { rl_set_error("hipFuncSetAttribute(%zu bytes of LDS) failed: %s", bytes, hipGetErrorString(e)); return RL_E_LAUNCH; }
    }
    const dim3 grid(h->cfg.n_worlds);
    if (T == 1024) {
        if (fixed) hipLaunchKernelGGL((k_run<1024, true, RL_PERD3QN>), grid, dim3(1024), bytes, st, rp);
        else hipLaunchKernelGGL((k_run<1024, false, RL_PERD3QN>), grid, dim3(1024), bytes, st, rp);
    } else if (T == 512) {
        if (fixed) hipLaunchKernelGGL((k_run<512, true, RL_PERD3QN>), grid, dim3(512), bytes, st, rp);
        else hipLaunchKernelGGL((k_run<512, false, RL_PERD3QN>), grid, dim3(512), bytes, st, rp);
    } else {
        if (fixed) hipLaunchKernelGGL((k_run<256, true, RL_PERD3QN>), grid, dim3(256), bytes, st, rp);
        else hipLaunchKernelGGL((k_run<256, false, RL_PERD3QN>), grid, dim3(256), bytes, st, rp);
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { rl_set_error("run kernel launch failed: %s", hipGetErrorString(e)); return RL_E_LAUNCH; }
    return RL_OK;
}

int rl_world_launch_capture(rl_world* h, const float* state, const int8_t* actions, const float* policy_out, const rl_step_out* so,
                            const rl_replay* replays, int n_brains, hipStream_t st)
{
    CaptureArgs a{};
    for (int i = 0; i < n_brains; ++i) a.rp[i] = replays[i];
    a.n_brains = n_brains; a.cap = h->cfg.slot_cap; a.state = state; a.actions = actions; a.policy_out = policy_out; a.so = *so;
    hipLaunchKernelGGL(k_capture, dim3(h->cfg.n_worlds), dim3(256), 0, st, a);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { rl_set_error("capture kernel launch failed: %s", hipGetErrorString(e)); return RL_E_LAUNCH; }
    return RL_OK;
}
