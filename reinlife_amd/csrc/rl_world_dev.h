// rl_world_dev.h -- device side of the ReinLife world tick on MI355X (gfx950), hand-written HIP: everything the world kernels
// (rl_world.hip: k_world / k_reset / k_capture) and the multi-tick kernel (rl_run.hip: k_run) share.
//
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
#pragma once
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
size_t rl_world_smem_bytes(int cpad, int cap, int hash, int plane_stride, int height);
#ifdef RL_PHASE_PROFILE
extern int g_rl_ablate;   // tuning only (rl_debug_set_ablate, rl_world.hip)
#endif

namespace {

constexpr int kSuper = RL_SUPER_FOOD;
constexpr uint8_t kPadCell = 0xFF;  // grid padding up to a multiple of 64 cells: neither empty nor anything else

// scalar slots in LDS
enum { S_ALIVE = 0, S_NFOOD, S_NPOISON, S_NSUPER, S_NSLOTS, S_N1, S_N2, S_NPARENTS, S_NELIG, S_BESTK, S_ERR, S_TICK, S_EPOCH,
       S_NEXT_UID, S_MAX_GENE, S_ANYFLAG0, S_ANYFLAG1, S_NPLACED, S_SPEC_NF, S_SPEC_NP, S_SPEC_DONE, S_PLANES_DIRTY, S_COUNT = 24 };

struct KParams {
    int W, H, C, Cp, nW;
    int PS;           // row stride of the observation planes in LDS: W, or plane_stride(W, H) in the multi-tick kernel
    unsigned invW;    // 2^32 / W rounded up: cell / W == umulhi(cell, invW) for every cell < 2^16 (cell_to_plane)
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
    int reset_families;    // k_reset only: Environment.reset()'s population -- n_brains agents, the one of key rank r gets gene r
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
    float* foodv;                 // [plane_words]: cell (i, j) at i * PS + j  (aliased: unsigned target counts per cell during movement)
    float* healthv;               // [plane_words]
    int* genev;                   // [plane_words]
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

// Row stride (32-bit words) of the three observation planes in LDS.  write_observations gathers the 7x7 window of ONE agent with 49
// consecutive lanes, cell (i + dr, j + dc) = word (i + dr) * S + j + dc, by ds_read_b32: two groups of 32 lanes, bank = word mod 32
// (MI355X_MICROARCH.md, LDS).  Lanes 0-31 are window rows 0-3 and four cells of row 4: with S = 7 (mod 32) the rows start at banks 0, 7,
// 14, 21, 28 and the 32 lanes hit the 32 banks once each; lanes 32-48 (rest of row 4, rows 5, 6) then take banks 0-16.  With S = W = 30
// the rows start at 0, 30, 28, 26, 24: every row collides with its neighbours (SQ_LDS_BANK_CONFLICT 29.5 % of the LDS-active cycles of
// k_run).  The padding is taken when it costs at most 16 KB over the three planes (30x30: 39 words, +3.2 KB) and only by
// the multi-tick kernel with ONE world per CU (workgroups of >= 512 threads: rl_run.hip run_plane_stride) -- the stand-alone world
// kernels keep S = W: with 256-thread workgroups four 30x30 worlds share a CU's 160 KB (39.9 KB each), and 3 KB more per world would
// make that three (1,024 worlds: 47.9 -> 62.9 us per tick launch).  KParams::PS carries the stride of the launch.
__host__ __device__ constexpr int plane_stride(int W, int H)
{
    int S = W;
    while ((S & 31) != 7) ++S;
    return (S - W) * H * 12 <= 16 * 1024 ? S : W;
}
__host__ __device__ constexpr int plane_words(int PS, int H, int Cp)   // (>= Cp: the planes double as per-cell scratch of other phases)
{
    const int n = (PS * H + 63) & ~63;
    return n > Cp ? n : Cp;
}
__host__ __device__ inline size_t carve(Smem& s, char* base, int Cp, int cap, int hash, int Pp)
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
    CARVE(foodv, float, Pp)
    CARVE(healthv, float, Pp)
    CARVE(genev, int, Pp)
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
// (from v_mbcnt, not from the work-item id: in the TRAIN instantiations of the multi-tick kernel the work-item id VGPR is spilled across the
// tile code, and every lane_id() of the one-wave sections -- six in the Tracker pass alone -- was a scratch reload + s_waitcnt vmcnt(0))
__device__ inline int lane_id() { return rl_lane_fresh(); }
// float64 constants that must NOT be hoisted out of k_run's tick loop: loop-invariant 64-bit values are kept alive across the policy half's
// 250-VGPR tile code, i.e. spilled, and every use becomes a scratch reload -- a memory round trip, on wave 0's serial sections at that
// (the ISA of round 3 reloaded -1e300 twice per tick in _update_best_agents and 0.005 in the planes).  Built from opaque halves at the use site.
__device__ inline double rl_opaque_f64(unsigned hi, unsigned lo)
{
    asm volatile("" : "+v"(hi), "+v"(lo));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ inline double rl_minus_huge() { return rl_opaque_f64(0xFE37E43Cu, 0x8800759Cu); }   // -1.0e300
__device__ inline double rl_one_200th() { return rl_opaque_f64(0x3F747AE1u, 0x47AE147Bu); }    // 0.005

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

// gene -> (alive count, on-grid count) open-addressing table; every agent remembers its slot (only ever used as an index).
// `active` lanes contribute.  (Rounds 1-3 aggregated the lanes of a wave that share a gene first -- one CAS + one add per distinct gene
// per wave -- on the assumption that two families' agents would otherwise hammer the same two LDS words; measured at the end of round 3,
// the hammering is the cheaper of the two.)
__device__ inline void hash_insert_wave(Smem& s, int mask, bool active, int a, int gene, unsigned add)
{
    // One CAS probe + one add per LANE.  The lanes of a gene serialise on its slot (~100 cycles per instruction for a full wave), which
    // is still cheaper than aggregating them first: the ballot version (per distinct gene of the wave a leader lane probes and adds the
    // popcounts, the others wait for its slot) cost 0.5 us per tick at configs[3] -- 23.57 -> 23.08 us -- although it issues two atomics
    // per gene instead of two per agent.
    if (active) {
        unsigned hh = ((unsigned)gene * 2654435761u) & (unsigned)mask;
        for (;;) {
            const int old = atomicCAS(&s.hkey[hh], -1, gene);
            if (old == -1 || old == gene) break;
            hh = (hh + 1) & (unsigned)mask;
        }
        if (add) atomicAdd(&s.hcnt[hh], add);
        s.hslot[a] = (unsigned short)hh;
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

// plane word of row-major cell c
__device__ __forceinline__ int cell_to_plane(const KParams& p, int c) { return c + (int)__umulhi((unsigned)c, p.invW) * (p.PS - p.W); }

// _prepare_observations (environment.py:377-404) into LDS planes
// `nslots` >= 0: the agents are slots 0 .. nslots-1 (on the grid iff occ[cell] names them) -- then in two disjoint parts: every
// NON-agent cell from its type alone (one LDS trip, no divergent chain), and every agent writes its own cell (position, health, flags,
// gene in one trip, the cell's occupant in a second).  The cell-driven version below pays type -> occupant -> (health, flags) -> gene,
// four dependent trips, in every wave that sees an agent cell, i.e. all of them, three times over for a 30x30 grid: it was the longest
// job of its interval in the multi-tick kernel (3.9k of the interval's 3.9k counts; without it 2.6k).
template <int T>
__device__ __forceinline__ void build_planes(const KParams& p, Smem& s, int t0 = rl_tidx(), int nt = T, int nslots = -1)
{
    if (RL_ABL(2)) return;
    const bool float_mode = s.type[0] == RL_AGENT;  // np.vectorize dtype inference from cell (0,0)
    if (nslots >= 0) {
        for (int c = t0; c < p.C; c += nt) {
            const int t = s.type[c];
            if (t == RL_AGENT) continue;
            const float f = t == RL_FOOD ? 0.5f : (t == kSuper ? 1.f : (t == RL_POISON ? -1.f : 0.f));
            const int pc = cell_to_plane(p, c);
            s.foodv[pc] = f; s.healthv[pc] = -1.f; s.genev[pc] = -2;
        }
        for (int a = t0; a < nslots; a += nt) {
            const int ps = s.pos[a], hp = s.health[a], fl = s.flags[a], ge = s.gene[a];  // one batch
            const int i = ps & 255, j = ps >> 8, c = i * p.W + j;
            if (s.type[c] != RL_AGENT || s.occ[c] != a) continue;   // (vanished: not on the grid)
            const double v = (double)hp * rl_one_200th();            // see below
            const int pc = i * p.PS + j;
            s.foodv[pc] = hp < 0 ? 1.f : 0.f;
            s.healthv[pc] = float_mode ? (float)v : (float)(double)(long long)v;
            s.genev[pc] = (fl & RL_F_DEAD) ? ge : -2;
        }
        return;
    }
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
            const double v = (double)hp * rl_one_200th();
            h = float_mode ? (float)v : (float)(double)(long long)v;  // astype(int64) truncates toward zero
            if (s.flags[a] & RL_F_DEAD) g = s.gene[a];             // _get_genes, environment.py:448-456
        }
        const int pc = cell_to_plane(p, c);
        s.foodv[pc] = f; s.healthv[pc] = h; s.genev[pc] = g;
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
                const int c = ci * p.PS + cj;
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
                c[u] = ci * p.PS + cj;
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
    // (Loop invariants of the multi-tick kernel's TICK loop must not outlive a tick: the compiler hoists (double)max_agents and the zero
    // vector of the mirror's padding to the kernel's preamble, the 256-VGPR tile code makes it spill them, and every reload here is a
    // scratch load + s_waitcnt vmcnt(0) -- which also waits for every observation-row store this wave has just issued.  Found in the ISA
    // of the mixed-kind kernel in round 6; both values are rebuilt from opaque operands where they are used: one v_cvt / v_mov each.)
    int max_agents_here = p.max_agents;
    asm volatile("" : "+s"(max_agents_here));
    float zero_here = 0.0f;
    asm volatile("" : "+v"(zero_here));
    for (int k = t; k < n; k += NT) {
        const int a = s.order[k];
        const int same = (int)(s.hcnt[s.hslot[a]] >> 16);
        float* o = base + (size_t)k * RL_OBS_DIM + 147;
        const float v0 = (float)((double)s.health[a] * rl_one_200th());  // == (float)(health / 200.0), see build_planes
        const float v1 = (s.flags[a] & RL_F_REPRODUCED) ? 1.f : 0.f;
        const float v2 = (float)((double)same / (double)n);
        const float v3 = (float)((double)n / (double)max_agents_here);
        const float v4 = (s.flags[a] & RL_F_KILLED) ? 1.f : 0.f;
        const float v5 = (s.flags[a] & RL_F_ATE_SUPER) ? 1.f : -1.f;
        o[0] = v0; o[1] = v1; o[2] = v2; o[3] = v3; o[4] = v4; o[5] = v5;
        if (mirror && k < xrows) {
            float* m = mirror + k * kXStride + 147;
            m[0] = v0; m[1] = v1; m[2] = v2; m[3] = v3; m[4] = v4; m[5] = v5;
            // floats 153 .. 159 of a mirror row are the zero padding of the input layer's last K-chunk (the four-wave tile reads the chunk
            // as it lies; the tiles' exchange buffers alias the mirror, so the padding is rewritten with the row)
            m[6] = zero_here; m[7] = zero_here; m[8] = zero_here; *(float4*)(m + 9) = float4{zero_here, zero_here, zero_here, zero_here};
        }
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
    // Every round of the loop also LOADS what _eat and the vanish rule need of the pre-move grid at the agent's current target (cell
    // type, occupant) and of the agent itself: in the round whose vote finds no conflict the targets are final, so what was loaded is
    // what _eat reads -- and the interval that follows only has to commit.  (Round 3: _eat, the clearing of the old cells and the
    // placement were three barrier intervals of their own behind the loop; they are ONE now, see below.)
    int any_phase = 0;
    // (parked in aux[a] next to the conflict flag: bits 0-2 the target's pre-move type, bit 3 = it held an agent, bit 4 = nobody targets
    // the agent's own cell, bit 7 = conflict)
    for (; !RL_ABL(16);) {
        int conflict = 0;
        for (int a = tid; a < n0; a += T) {
            const int cx = (s.pos[a] & 255) * W + (s.pos[a] >> 8);
            const int tg = s.tgt[a];
            const unsigned ct = cnt[tg];  // unconditional: behind `tg != cx &&` it would be a third dependent LDS trip
            const int e_tt = s.type[tg], e_oc = s.occ[tg];
            const unsigned e_cnt = cnt[cx];
            const bool c = tg != cx && ct > 1u;
            s.aux[a] = (uint8_t)((c ? 0x80 : 0) | (e_tt & 7) | (e_oc >= 0 ? 8 : 0) | (e_cnt == 0u ? 16 : 0));
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
    // ---- _eat + vanish rule + _update_agent_position + _update_death_status, ONE interval -------------------------------------------
    // The reference moves the agents one by one in list order: grid[old] = Empty; grid[target] = agent (environment.py:778-782).  As a
    // parallel rule with ONE writer per cell: a mover empties its old cell only if NOBODY targets it (the final target counts say so);
    // a cell that is entered is written by the agent entering it -- as an agent cell, or as an EMPTY cell when the entering agent is
    // erased by the later-ordered occupant's grid[old] = Empty (the vanish rule).  A cell's pre-move content is only read in the loop
    // above (before the last vote's barrier), so nothing here reads what another lane writes.
    // (the tick's draws: first needed by _add_food below.  Not in the first agent phase, where a preparing world's idle
    // waves are busy with their cells' Philox blocks)
    if (LEAN) precompute_draws<T>(p, s, w, n0);
    if (SPEC && spec) {   // the refill's counting sort keeps its barrier-separated stages (a preparing world only)
        spec_refill_stage<T>(p, s, w, spec_state, 1); lds_barrier();
        spec_refill_stage<T>(p, s, w, spec_state, 2); lds_barrier();
    }
    RL_MARK(35);
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
            const int tg = s.tgt[a], act = s.action[a];
            int hp = s.health[a], ma = s.max_age[a], fl = s.flags[a];
            const int ag = s.age[a], ex = s.aux[a];   // one batch
            gene = s.gene[a];
            const int e_tt = ex & 7;
            const bool e_occupied = ex & 8, old_free = ex & 16;
            bool vanish = false;
            if (act >= 0 && act <= 3) {   // _eat (environment.py:701-715): the pre-move type at the final target
                if (e_tt == RL_FOOD) hp = min(200, hp + 40);
                else if (e_tt == RL_POISON) hp = min(200, hp - 40);
                else if (e_tt == kSuper) { hp = min(200, hp + 40); ma = (int)((double)ma * 1.2); fl |= RL_F_ATE_SUPER; }
                // entering the cell of a later-ordered agent that is itself leaving: erased by its grid[old]=Empty
                vanish = tg != cx && e_occupied && tg > cx;
            }
            if (tg != cx) {
                if (old_free) { s.type[cx] = RL_EMPTY; s.occ[cx] = -1; }            // nobody enters the old cell
                if (vanish) { s.type[tg] = RL_EMPTY; s.occ[tg] = -1; }              // (the occupant left it; its own clear does not happen: its cell IS targeted)
                else { s.type[tg] = RL_AGENT; s.occ[tg] = (short)a; }
                const int ti = tg / W;
                s.pos[a] = (unsigned short)(ti | ((tg - ti * W) << 8));
            }
            // _update_death_status (environment.py:789-793)
            if (hp <= 0 || ag == ma) fl |= RL_F_DEAD;
            s.health[a] = hp; s.max_age[a] = ma; s.flags[a] = (uint8_t)fl;
            s.aux[a] = vanish ? AUX_VANISH : 0;
            alive = (fl & RL_F_DEAD) ? 0u : 1u;
            ongrid = vanish ? 0u : 1u;
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
    // (workgroups of >= 512 threads: the cells go to waves 2.., which have nothing else in this interval -- waves 0 and 1 carry the
    // agents' rewards above, and as the last to arrive at the sweep they made it the interval's longest job)
    constexpr int kSweep0 = T >= 512 ? 128 : 0;
    for (int c = tid - kSweep0; c >= 0 && c < p.Cp; c += T - kSweep0) {
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
        build_planes<T>(p, s, tid - 128, T - 128, n0);   // (the list of the step: slots 0 .. n0-1)
    }
    lds_barrier();
}

// the planes of the cells _add_food just filled (they were built as empty cells next to the placement)
__device__ inline void patch_placed_planes(const KParams& p, Smem& s)
{
    const int tid = rl_tidx();
    if (tid < s.scal[S_NPLACED]) {
        const int c = s.plist[tid];
        const int t = s.type[c];
        s.foodv[cell_to_plane(p, c)] = t == RL_FOOD ? 0.5f : (t == kSuper ? 1.f : -1.f);
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

// (fitness, list index) of the wave's fittest agent -- first on ties, like np.argmax -- in every lane.  DPP moves (row_shr 1 / 2 / 4 / 8, then
// the two row broadcasts: lane 63 ends up with the wave's best) instead of six xor-shuffle steps of three ds_bpermute each: a dependent
// LDS-crossbar trip apiece on the wave that is the longest of its interval.
__device__ inline void wave_argmax_f64(double& f, int& k)
{
#define RL_ARGMAX_STEP(CTRL, RMASK) do { \
        const long long bits_ = __double_as_longlong(f); \
        const int lo_ = (int)(unsigned)bits_, hi_ = (int)(unsigned)((unsigned long long)bits_ >> 32); \
        const unsigned olo_ = (unsigned)__builtin_amdgcn_update_dpp(lo_, lo_, CTRL, RMASK, 0xF, false); \
        const unsigned ohi_ = (unsigned)__builtin_amdgcn_update_dpp(hi_, hi_, CTRL, RMASK, 0xF, false); \
        const int ok_ = __builtin_amdgcn_update_dpp(k, k, CTRL, RMASK, 0xF, false); \
        const double of_ = __longlong_as_double((long long)(((unsigned long long)ohi_ << 32) | olo_)); \
        if (of_ > f || (of_ == f && ok_ < k)) { f = of_; k = ok_; } } while (0)   /* (a lane without a source gets its own value back) */
    RL_ARGMAX_STEP(0x111, 0xF); RL_ARGMAX_STEP(0x112, 0xF); RL_ARGMAX_STEP(0x114, 0xF); RL_ARGMAX_STEP(0x118, 0xF);
    RL_ARGMAX_STEP(0x142, 0xA); RL_ARGMAX_STEP(0x143, 0xC);
#undef RL_ARGMAX_STEP
    const long long bits = __double_as_longlong(f);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)bits, 63);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)((unsigned long long)bits >> 32), 63);
    f = __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
    k = __builtin_amdgcn_readlane(k, 63);
}
// _update_best_agents (environment.py:728-739) by ONE wave (no workgroup barrier): used when the update's serial section
// runs on wave 0 next to the other waves' observation pass
__device__ inline void best_agents_wave(Smem& s, int n1)
{
    const int lane = lane_id();
    double bf = rl_minus_huge(); int bk = 0x7fffffff;
    for (int k = lane; k < n1; k += 64) {
        const double f = s.fitness[s.order[k]];
        if (f > bf || (f == bf && k < bk)) { bf = f; bk = k; }
    }
    wave_argmax_f64(bf, bk);
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
        double bf = rl_minus_huge(); int bk = 0x7fffffff;
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
// `scr` [cap]: the rewards grouped by gene (the lean kernels park their Philox draws in s.trk_rew, so the multi-tick kernel hands in a
// scratch of its own); `acc` (optional, LDS): running sums kept in the workgroup for the length of a multi-tick launch -- loaded from /
// flushed to trk_sum / trk_cnt / trk_pop once per launch instead of a read-modify-write through L2 every tick.
struct TrkLds {
    double* sum;   // [G][RL_TRK_VARS]
    int* cnt;      // [G][RL_TRK_VARS]
    double* pop;   // [2] running sum, running count of "Avg Number of Populations"
};
// `accumulate` false: only trk_tick is written (episode 0 of a training run never reaches an aggregate, tracker.py:279-282).
__device__ __forceinline__ void track_world_wave0(const KParams& p, Smem& s, int w, int n1, double* scr = nullptr, const TrkLds* acc = nullptr,
                                                  bool accumulate = true)
{
    if (!scr) scr = s.trk_rew;
    const int lane = lane_id();
    const int G = p.static_families ? p.n_brains : 1;
    int m = 0, sum_age = 0, best = 0, attacks = 0, kills = 0;  // lane g: statistics of group g
    // pass 1: per-group integer statistics.  With LDS atomics on four words per group (count | attacks << 16, sum of the ages, oldest,
    // kills; integer sums and maxima do not depend on the order): the lanes of a group serialise on its words, ~100 cycles per
    // instruction -- the ballot-aggregated loop below (two wave scans per distinct group of every 64-agent chunk) took 1.4 us of a wave
    // that is the longest of its interval (tools/trk_parts.py).  The words live at the head of scr, which pass 2 overwrites afterwards.
    const bool by_atomics = 16 * G <= 8 * p.cap && n1 <= 4096;
    if (by_atomics) {
        unsigned* const acc4 = (unsigned*)scr;
        unsigned zero_here = 0u;   // (opaque: as a constant the 16-byte zero vector was hoisted out of the multi-tick kernel's tick loop, spilled, and reloaded
        asm volatile("" : "+v"(zero_here));   //  here behind an s_waitcnt vmcnt(0) -- see write_observations)
        if (lane < G) { acc4[4 * lane] = zero_here; acc4[4 * lane + 1] = zero_here; acc4[4 * lane + 2] = zero_here; acc4[4 * lane + 3] = zero_here; }
        for (int base = 0; base < n1; base += 64) {
            const int k = base + lane;
            if (k < n1) {
                const int a = s.order[k];
                const int g = p.static_families ? s.gene[a] : 0;
                if (g >= 0 && g < G) {
                    const int age = s.age[a];
                    const unsigned fl = s.flags[a];
                    __hip_atomic_fetch_add(&acc4[4 * g], 1u + (s.action[a] >= 4 ? 0x10000u : 0u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_add(&acc4[4 * g + 1], (unsigned)age, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_max(&acc4[4 * g + 2], (unsigned)max(age, 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (fl & RL_F_KILLED) __hip_atomic_fetch_add(&acc4[4 * g + 3], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        }
        if (lane < G) {
            const unsigned ca = acc4[4 * lane];
            m = (int)(ca & 0xffffu); attacks = (int)(ca >> 16); sum_age = (int)acc4[4 * lane + 1]; best = (int)acc4[4 * lane + 2]; kills = (int)acc4[4 * lane + 3];
        }
        asm volatile("" ::: "memory");   // (pass 2 stores doubles into the same bytes: keep it behind these reads)
    } else
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
            if (mine) scr[start + __popcll(mm & lowmask(lane))] = s.reward[a];
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
                v[2] = np_pairwise_sum<5>(scr + off, m) / (double)m;
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
            if (accumulate && v[i] > -1.0) {
                if (acc) { acc->sum[lane * RL_TRK_VARS + i] += v[i]; acc->cnt[lane * RL_TRK_VARS + i] += 1; }
                else { p.so.trk_sum[o + i] += v[i]; p.so.trk_cnt[o + i] += 1; }
            }
        }
    }
    if (lane == 0) {
        const double pv = n1 == 0 ? -1.0 : (double)n_distinct;
        p.so.trk_pop[(size_t)w * 3] = pv;
        if (accumulate && pv > -1.0) {
            if (acc) { acc->pop[0] += pv; acc->pop[1] += 1.0; }
            else { p.so.trk_pop[(size_t)w * 3 + 1] += pv; p.so.trk_pop[(size_t)w * 3 + 2] += 1.0; }
        }
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
constexpr int kFixPp = plane_words(kFixW, kFixH, kFixCp);   // (the stand-alone kernels: row-major planes)
constexpr unsigned kFixInvW = (unsigned)(((1ull << 32) + kFixW - 1) / kFixW);
constexpr int kFixCap = ((2 * kFixMaxAgents + 2 + 63) / 64) * 64;   // 256: births can overshoot max_agents up to 2n+1
constexpr int kFixHash = 512;                                       // rl_create: the power of two >= 2 * slot_cap
static_assert(kFixHash >= 2 * kFixCap && kFixHash / 2 < 2 * kFixCap, "hash size rule of rl_create");
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
            int gene = (int)rl_mulhi(r.y, (unsigned)p.n_brains);
            if (p.reset_families) {   // one agent per brain (environment.py:147-149): the gene is the cell's key rank (keys / buckets are still in LDS)
                const unsigned key = keys[c], b = key >> sh;
                const int start = b ? (int)cum[b - 1] : 0, end = (int)cum[b];
                gene = start;
                for (int j = start; j < end; ++j) gene += sorted[j] < key;
            }
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

// 1024 threads per world when there are few worlds (latency-bound: one world per CU), 256 when there are many
// (throughput-bound: several worlds per CU hide each other's barriers).
inline int pick_block(const rl_world* h)
{
    const int forced = h->opt.world_block;   // (rl_set_option / RL_WORLD_BLOCK at rl_create: tests and A/B runs)
    if (forced == 256 || forced == 512 || forced == 1024) return forced;
    return h->cfg.n_worlds <= 768 ? 1024 : 256;
}

inline KParams make_params(const rl_world* h)
{
    KParams p{};
    p.W = h->cfg.width; p.H = h->cfg.height; p.C = h->cells; p.Cp = h->cpad; p.nW = h->cpad / 64;
    p.PS = p.W; p.invW = (unsigned)(((1ull << 32) + p.W - 1) / p.W);
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
#ifdef RL_PHASE_PROFILE
    p.ablate = g_rl_ablate;
#endif
    return p;
}

// A launch that leaves every world policy-ready (tick / update / reset / refill) also produces the per-brain row lists
// when a policy work buffer is bound; any other launch invalidates them.
inline void set_list_production(rl_world* h, KParams& p, bool produces)
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
