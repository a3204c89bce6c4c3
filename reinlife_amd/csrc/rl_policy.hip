// rl_policy.hip -- batched policy inference for ReinLife's brains on MI355X (gfx950), hand-written HIP + MFMA.
//
// Reference (paths under /root/reference/ReinLife/Models):
//   DQN      Qnet.forward            DQN.py:126-130      153 -> 128 -> 64 -> 8
//   D3QN     dueling_ddqn.forward    D3QN.py:161-165     153 -> 128 -> (128 -> 8 || 128 -> 1), q = adv + val - mean(adv)
//   PERD3QN  DuelingDDQN.forward     PERD3QN.py:198-202  (same network)
//   PPO      PPO.pi                  PPO.py:101-106      153 -> 256 -> 256 -> 8 -> softmax
//   action selection                 DQN.py:132-139, D3QN.py:167-173, PERD3QN.py:204-210, PPO.py:164-169
// The reference runs one batch-1 forward per agent; here 32 agents of one brain form a tile computed by TWO waves on one SIMD (the tiles of
// the multi-tick kernel's policy half, rl_policy_dev.h): k_policy_pair = four tiles of one brain per 512-thread workgroup, k_policy_dense
// from 1,536 dueling tiles on; k_policy_wave (policy_variant "wave") = one wave per tile.  One arithmetic in all of them:
//
//   * f32-grade results from the f16 matrix pipe ("2 x f16, block-scaled"): every operand row is scaled by a power of two,
//     split into two f16 parts x = hi + lo, and a product is hi.hi + hi.lo + lo.hi -- three v_mfma_f32_32x32x16_f16 per 16 k
//     with f32 accumulation.  The scheme, its error bound and its history are described in rl_policy_dev.h.
//   * transposed formulation  H_out[feature][row] = W[feature][k] . H_in[k][row]: the WEIGHTS are the MFMA A operand
//     and the ACTIVATIONS the B operand.  The 32x32 f32 accumulator layout (lane = row, register r of half h =
//     feature (r&3) + 8(r>>2) + 4h) is dtype-independent, so a lane's registers 8c..8c+7 are exactly its 8 B-operand
//     values of K-chunk (tile, c) of the next layer when the next layer's weights are packed in that k order:
//     activations are never transposed; the weights are pre-packed so every A fragment is one coalesced
//     16-byte-per-lane load (rl_policy_pack_weights).
//   * the narrow heads (8 / 1 outputs) also run on the matrix pipe (head rows = A operand padded to 32, the wave's own
//     activation registers = B operand), followed by the dueling combine / softmax and the epsilon-greedy / categorical
//     draw (Philox) in the same kernel.
#include "rl_policy_dev.h"
#include <math.h>
#include <string.h>

#include <vector>

namespace {

constexpr int kMaxBrainsPerLaunch = 8;

struct BrainSlot {
    const float* packed;   // device, rl_policy_pack_weights layout
    const int* rowlist;    // row ids (world*cap + k) of the agents using this brain; nullptr = dense rows 0..n_rows-1
    const int* count_ptr;  // device count of rowlist entries (nullptr = n_rows)
    float eps;
    int kind;              // RL_DQN .. RL_PPO (read by the mixed-kind launch only)
};

struct PolicyArgs {
    BrainSlot b[kMaxBrainsPerLaunch];
    int nb;
    const float* obs;         // rows of 153 floats
    int64_t n_rows;           // dense mode only
    float* out;               // [row][8] or nullptr
    int8_t* actions;          // [row] or nullptr
    uint64_t seed;
    int cap, world_base;
    const int32_t* tick;      // per world
    const int32_t* epoch;
#ifdef RL_PHASE_PROFILE
    long long* prof;          // tuning build: shader-clock stamps of workgroup prof_block, wave 0 (slots 100.. of a 128-entry buffer)
    int prof_block;
#endif
};

// One wave per tile (policy_tile1, the dueling kinds): a 64-thread workgroup per (tile, brain), no LDS, no barrier.
template <int KIND>
__global__ __launch_bounds__(64, 2) void k_policy1(const PolicyArgs A)
{
    const int lane = threadIdx.x & 63, j = lane & 31;
    {
        auto ka = __builtin_amdgcn_kernarg_segment_ptr();
        int t0, t1, t2, t3, t4, t5;
        asm volatile("s_load_dword %0, %6, 0x0\n\ts_load_dword %1, %6, 0x40\n\ts_load_dword %2, %6, 0x80\n\t"
                     "s_load_dword %3, %6, 0xc0\n\ts_load_dword %4, %6, 0x100\n\ts_load_dword %5, %6, 0x140\n\ts_waitcnt lgkmcnt(0)"
                     : "=s"(t0), "=s"(t1), "=s"(t2), "=s"(t3), "=s"(t4), "=s"(t5) : "s"(ka) : "memory");
    }
    typedef const int __attribute__((address_space(4))) cint;
    const int bi = blockIdx.y, tile = blockIdx.x;
    const BrainSlot B = A.b[bi];
    const int li = tile * 32 + j;
    const int n = B.count_ptr ? ((cint*)B.count_ptr)[0] : (int)A.n_rows;
    if (tile * 32 >= n) return;   // (before the row list is touched: the grid is sized by the rows a brain COULD have)
    const int entry = B.rowlist ? B.rowlist[min(li, n - 1)] : 0;   // lanes past the end: the brain's last row, not stored (as k_policy_dense)
    const int e_w = rl_list_world(entry), e_k = rl_list_slot(entry);
    const int64_t listed = B.rowlist ? (int64_t)e_w * A.cap + e_k : (int64_t)li;
    TileIO io;
    io.packed = (gfloat*)B.packed;
    io.obs = A.obs;
    io.valid = li < n;
    io.row = (io.valid || B.rowlist) ? listed : (int64_t)tile * 32;
    io.eps = B.eps; io.out = A.out; io.actions = A.actions; io.seed = A.seed;
    io.key_world = io.key_tick = io.key_epoch = io.key_index = 0;
    io.lds_actions_off = -1; io.lds_slot = 0; io.x_lds_off = -1; io.c_lds_off = -1;
    if (A.actions && lane < 32) {
        io.key_world = (uint32_t)(A.world_base + e_w); io.key_index = (uint32_t)e_k;
        io.key_tick = (uint32_t)A.tick[e_w]; io.key_epoch = (uint32_t)A.epoch[e_w];
    }
#ifdef RL_PHASE_PROFILE
    io.prof = nullptr;
#endif
    policy_tile1<KIND, false>(io, lane);
}

// kDenseTiles tiles of one brain per workgroup, the weights through LDS (policy_tile1ds): grid = (groups of kDenseTiles tiles a brain can
// have at most, brains).
template <int KIND>
__global__ __launch_bounds__(64 * kDenseTiles, 2) void k_policy_dense(const PolicyArgs A)
{
    __shared__ __attribute__((aligned(16))) f32x4 lds_w[3 * kStageUnits];
    __shared__ __attribute__((aligned(16))) float lds_c[kTileConstFloats];
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, wave = tid >> 6;
    typedef const int __attribute__((address_space(4))) cint;
    const int bi = blockIdx.y;
    const BrainSlot B = A.b[bi];
    const int n = B.count_ptr ? ((cint*)B.count_ptr)[0] : (int)A.n_rows;
    if ((int64_t)blockIdx.x * kDenseTiles * 32 >= n) return;   // (uniform: the whole workgroup)
    const int tile = blockIdx.x * kDenseTiles + wave;
    const int li = tile * 32 + j;
    const int lic = min(li, n - 1);                  // waves / lanes past the end: the brain's last row, not stored
    const int entry = B.rowlist ? B.rowlist[lic] : 0;
    const int e_w = rl_list_world(entry), e_k = rl_list_slot(entry);
    TileIO io;
    io.packed = (gfloat*)B.packed;
    io.obs = A.obs;
    io.valid = li < n;
    io.row = B.rowlist ? (int64_t)e_w * A.cap + e_k : (int64_t)lic;
    io.eps = B.eps; io.out = A.out; io.actions = A.actions; io.seed = A.seed;
    io.key_world = io.key_tick = io.key_epoch = io.key_index = 0;
    io.lds_actions_off = -1; io.lds_slot = 0; io.x_lds_off = -1; io.c_lds_off = -1;
    if (A.actions && lane < 32) {
        io.key_world = (uint32_t)(A.world_base + e_w); io.key_index = (uint32_t)e_k;
        io.key_tick = (uint32_t)A.tick[e_w]; io.key_epoch = (uint32_t)A.epoch[e_w];
    }
#ifdef RL_PHASE_PROFILE
    io.prof = nullptr;
#endif
    {   // the brain's epilogue / head constants (published by the barrier in WStage2::start)
        const Layout L = layout_of(KIND);
        gfloat* pk = (gfloat*)B.packed;
        for (int i = tid; i < kTileConstFloats; i += 64 * kDenseTiles) {
            const int layer = i >> 8;
            const int64_t off = layer == 0 ? L.l1 + frag_floats(kInChunks, 4) : layer == 1 ? L.l2a + frag_floats(8, 4) : layer == 2 ? L.l2b + frag_floats(8, 4)
                              : (i < 768 + 16 ? L.ha : L.hb) + head_consts_off(4) - (i < 768 + 16 ? 768 : 768 + 16);
            lds_c[i] = pk[off + (layer < 3 ? (i & 255) : i)];
        }
    }
    WStage2 ws;
    ws.buf = lds_w; ws.src = (gf32x4*)B.packed + tid; ws.tid = tid; ws.lane = lane;
    policy_tile1ds<KIND>(io, lane, ws, lds_c);
}

// policy_variant "pair" (the default stand-alone launch): TWO waves per 32-row tile -- policy_tile1s<PAIR> for the dueling kinds,
// policy_pair2 for DQN / PPO: the tiles (and the arithmetic) of the multi-tick kernel's policy half, for brains of any kinds.
// A workgroup is what k_run's policy half is: kPairTiles tiles of ONE brain on 2 * kPairTiles waves, tile t on waves t and t + kPairTiles --
// for four tiles the two roles of a tile share a SIMD (waves i and i + 4 do: tools/ubench/simd_map.hip), which is what the pair was
// scheduled for: a role alone on its SIMD issues a VALU instruction every 8 cycles, two fill each other's waits.  With
// one tile per 128-thread workgroup (round 3) the roles sat on different SIMDs: 25.4 us per launch at 256 worlds, 4.6 us behind the
// 4-wave tile.  Grid = (groups of kPairTiles tiles a brain can have at most, brains).  Dynamic LDS:
// [the brain's constants kTileConstMax | per tile: partial row maxima 128, role 1's head partials 256 | per tile: exchange buffer].
#ifndef RL_PAIR_TILES
#define RL_PAIR_TILES 4
#endif
constexpr int kPairStageStride = 164, kPairStageBytes = 32 * kPairStageStride * 4;   // the tile's 32 staged observation rows (20.5 KB)
constexpr int kPairTiles = RL_PAIR_TILES;
__host__ __device__ constexpr int pair_kernel_lds(int ex_bytes) { return (kTileConstMax + kPairTiles * (128 + kPairValFloats)) * 4 + kPairTiles * ex_bytes; }
__global__ __launch_bounds__(128 * kPairTiles) void k_policy_pair(const PolicyArgs A, const int ex_bytes)
{
    extern __shared__ __attribute__((aligned(16))) char rl_dyn_lds[];
    float* lds_c = (float*)rl_dyn_lds;
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int slot = wave % kPairTiles, role = wave / kPairTiles;
    typedef const int __attribute__((address_space(4))) cint;
    const int bi = blockIdx.y, tile = blockIdx.x * kPairTiles + slot;
    const BrainSlot B = A.b[bi];
    const int n = B.count_ptr ? ((cint*)B.count_ptr)[0] : (int)A.n_rows;
    if ((int64_t)blockIdx.x * kPairTiles * 32 >= n) return;   // (uniform: the whole workgroup)
    const bool have = tile * 32 < n;                           // (uniform per wave; a wave without a tile still meets the barriers)
    const int li = tile * 32 + j, lic = min(li, n - 1);
    const int entry = B.rowlist ? B.rowlist[lic] : 0;
    const int e_w = rl_list_world(entry), e_k = rl_list_slot(entry);
    TileIO io;
    io.packed = (gfloat*)B.packed;
    io.obs = A.obs;
    io.valid = li < n;
    io.row = B.rowlist ? (int64_t)e_w * A.cap + e_k : (int64_t)lic;
    io.eps = B.eps; io.out = A.out; io.actions = A.actions; io.seed = A.seed;
    io.key_world = io.key_tick = io.key_epoch = io.key_index = 0;
    io.lds_actions_off = -1; io.lds_slot = 0; io.x_lds_off = -1; io.c_lds_off = 0;
    if (A.actions && lane < 32) {
        io.key_world = (uint32_t)(A.world_base + e_w); io.key_index = (uint32_t)e_k;
        io.key_tick = (uint32_t)A.tick[e_w]; io.key_epoch = (uint32_t)A.epoch[e_w];
    }
#ifdef RL_PHASE_PROFILE
    io.prof = nullptr;
#endif
    const int kind = B.kind;
    PairLds pl;
    pl.pmax = lds_c + kTileConstMax + slot * (128 + kPairValFloats); pl.val = pl.pmax + 128;
    pl.ex = (f32x4*)((char*)(lds_c + kTileConstMax + kPairTiles * (128 + kPairValFloats)) + (size_t)slot * ex_bytes);
    {
        // The tile's 32 observation rows, staged through LDS: wave `role` fetches rows 16 role .. 16 role + 15 with ONE coalesced 612-byte
        // read per row (lane m: floats 4m .. 4m+3; lane 38: the last float) -- read straight into B-operand order, as the tile does with
        // rows that are not mirrored, every 16-byte load instruction touches 32 different rows (64 cache lines), twice over for the two
        // roles.  The rows lie where the pair's exchange buffer will be (dead once both roles have read them: the first barrier inside
        // the tile), 164 floats apart like k_run's mirror.
        constexpr int kStride = kPairStageStride;   // (ex_bytes >= kPairStageBytes: the launcher sizes the slots)
        float* stage = (float*)pl.ex;
        const int lo = (int)(io.row & 0xffffffff), hi = (int)(io.row >> 32);
        const int off = lane < 38 ? 4 * lane : 149;   // (every lane loads: lanes >= 38 read floats 149..152, inside the row)
        f32x4 val[16];
        if (have) {
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) {
                const int jj = 16 * role + rr;
                const int64_t r = ((int64_t)__builtin_amdgcn_readlane(hi, jj) << 32) | (unsigned)__builtin_amdgcn_readlane(lo, jj);
                val[rr] = *(const f32x4u*)(A.obs + r * RL_OBS_DIM + off);
            }
        }
        gfloat* pk = (gfloat*)B.packed;   // (the brain's epilogue / head constants meanwhile)
        for (int i = tid; i < tile_const_floats(kind); i += 128 * kPairTiles) lds_c[i] = pk[tile_const_src(kind, i)];
        if (have) {
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) {
                float* dst = stage + (16 * role + rr) * kStride;
                if (lane < 38) *(f32x4*)(dst + 4 * lane) = val[rr];
                else if (lane == 38) dst[152] = val[rr].w;
            }
        }
        io.x_lds_off = (int)((char*)(stage + j * kStride) - rl_dyn_lds);
    }
    lds_barrier();
    Tile1Part part;
    if (have) {
        if (kind == RL_DQN) policy_pair2<RL_DQN, false>(io, lane, role, &pl, &part);
        else if (kind == RL_PPO) policy_pair2<RL_PPO, false>(io, lane, role, &pl, &part);
        else policy_tile1s<RL_PERD3QN, false, true>(io, lane, role, &pl, &part);
    } else { lds_barrier(); lds_barrier(); }   // (the two exchanges inside a tile)
    lds_barrier();
    if (have && role == 0) {
        if (kind == RL_DQN) pair_finish<RL_DQN>(io, lane, part, &pl);
        else if (kind == RL_PPO) pair_finish<RL_PPO>(io, lane, part, &pl);
        else tile1_finish<RL_PERD3QN>(io, lane, part.head, pl.val[j], part.draw, *(const f32x4*)(lds_c + 768 + 8 + 4 * (lane >> 5)));
    }
}

// Per-brain row lists: one wave per world.  Pass 1 counts the world's agents per brain with ballots (lane b keeps
// brain b's count), ONE atomic instruction reserves the ranges of all brains, pass 2 scatters the row ids.
// counts_zero is the other parity's counter block, cleared here for the next call (no memset node needed).
__global__ __launch_bounds__(256) void k_bucket(const int32_t* __restrict__ n_agents, const int32_t* __restrict__ a_brain,
                                                int n_worlds, int cap, int n_brains, int* __restrict__ counts,
                                                int* __restrict__ counts_zero, int* __restrict__ lists, int64_t list_stride)
{
    const int lane = threadIdx.x & 63;
    if (blockIdx.x == 0 && threadIdx.x < 64) counts_zero[threadIdx.x] = 0;
    const int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (w >= n_worlds) return;
    const int n = n_agents[w];
    int cnt = 0;
    for (int base = 0; base < n; base += 64) {
        const int k = base + lane;
        const int b = k < n ? a_brain[(size_t)w * cap + k] : -1;
        for (int bb = 0; bb < n_brains; ++bb) {
            const int c = __popcll(__ballot(b == bb));
            if (lane == bb) cnt += c;
        }
    }
    int pos = (lane < n_brains && cnt) ? atomicAdd(&counts[lane], cnt) : 0;
    for (int base = 0; base < n; base += 64) {
        const int k = base + lane;
        const int b = k < n ? a_brain[(size_t)w * cap + k] : -1;
        for (int bb = 0; bb < n_brains; ++bb) {
            const unsigned long long m = __ballot(b == bb);
            const int start = __shfl(pos, bb);
            if (b == bb) lists[bb * list_stride + start + __popcll(m & ((1ull << lane) - 1ull))] = rl_list_entry(w, k);
            if (lane == bb) pos += __popcll(m);
        }
    }
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------
int64_t rl_policy_n_params_impl(int kind)
{
    if (kind == RL_DQN) return 153 * 128 + 128 + 128 * 64 + 64 + 64 * 8 + 8;
    if (kind == RL_D3QN || kind == RL_PERD3QN) return 153 * 128 + 128 + 2 * (128 * 128 + 128) + 128 * 8 + 8 + 128 + 1;
    if (kind == RL_PPO) return 153 * 256 + 256 + 256 * 256 + 256 + 256 * 8 + 8 + 256 + 1;
    return -1;
}
int64_t rl_policy_packed_floats_impl(int kind)
{
    if (kind < RL_DQN || kind > RL_PPO) return -1;
    return layout_of(kind).total;
}

// ---- f16 split of a scaled weight (host mirror of split_pair): hi = x toward zero, lo = x - hi toward zero ----
static inline uint16_t f16_rtz(float f)  // |f| < 65504 (scaled weights are < 2^12), result exact toward zero
{
    uint32_t u;
    memcpy(&u, &f, 4);
    const uint32_t sign = (u >> 16) & 0x8000u;
    const int e = (int)((u >> 23) & 0xff) - 127;
    const uint32_t man = u & 0x7fffffu;
    if (e < -24) return (uint16_t)sign;                                       // below the smallest subnormal
    if (e < -14) return (uint16_t)(sign | ((man | 0x800000u) >> (13 + (-14 - e))));  // subnormal, truncated
    return (uint16_t)(sign | (uint32_t)((e + 15) << 10) | (man >> 13));
}
static inline float f16_to_float(uint16_t h)
{
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 0x1f, man = h & 0x3ffu;
    float f;
    if (e == 0) { f = (float)man * (1.0f / 16777216.0f); uint32_t u; memcpy(&u, &f, 4); u |= sign; memcpy(&f, &u, 4); return f; }
    const uint32_t u = sign | ((e - 15 + 127) << 23) | (man << 13);
    memcpy(&f, &u, 4);
    return f;
}
static inline void split2_host(float x, uint16_t (&out)[2])
{
    out[0] = f16_rtz(x);
    out[1] = f16_rtz(x - f16_to_float(out[0]));
}
// scale of output feature o of a row-major [n_out][n_in] matrix: 2^(kScaleExp - exponent(max |W[o][:]|))
static void feature_scales(const float* W, int n_out, int n_in, std::vector<float>& sc, std::vector<float>& un)
{
    sc.resize(n_out); un.resize(n_out);
    for (int o = 0; o < n_out; ++o) {
        float mx = 0.0f;
        for (int k = 0; k < n_in; ++k) mx = fmaxf(mx, fabsf(W[(size_t)o * n_in + k]));
        row_scale(mx, sc[o], un[o]);
    }
}
static void write_epilogue_consts(float* consts, int tout, const std::vector<float>& un, const float* b)
{
    // per output tile t2 and lane half h: [unscale 16 | bias 16] for feature 32 t2 + (r&3) + 8(r>>2) + 4h, r = 0..15
    for (int t2 = 0; t2 < tout; ++t2)
        for (int h = 0; h < 2; ++h)
            for (int r = 0; r < 16; ++r) {
                const int o = 32 * t2 + (r & 3) + 8 * (r >> 2) + 4 * h;
                consts[(t2 * 2 + h) * 32 + r] = un[o];
                consts[(t2 * 2 + h) * 32 + 16 + r] = b[o];
            }
}

static void pack_in_layer(const float* W, const float* b, int n_out, float* dst)
{
    // dst16[c][t][plane][lane][e] = part_plane(scale[o] * W[o = 32t + (lane&31)][k = 16c + 8(lane>>5) + e]), zero for k >= 153
    const int tout = n_out / 32;
    std::vector<float> sc, un;
    feature_scales(W, n_out, 153, sc, un);
    uint16_t* d16 = (uint16_t*)dst;
    for (int c = 0; c < kInChunks; ++c)
        for (int t = 0; t < tout; ++t)
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 8; ++e) {
                    const int o = 32 * t + (lane & 31), k = 16 * c + 8 * (lane >> 5) + e;
                    uint16_t parts[2];
                    split2_host(k < 153 ? W[(size_t)o * 153 + k] * sc[o] : 0.0f, parts);
                    for (int pl = 0; pl < kPlanes; ++pl) d16[(((((size_t)c * tout + t) * kPlanes + pl) * 64 + lane) * 8) + e] = parts[pl];
                }
    write_epilogue_consts(dst + frag_floats(kInChunks, tout), tout, un, b);
}
static void pack_hidden_layer(const float* W, const float* b, int n_in, int n_out, float* dst)
{
    // dst16[s = t*2+c][t2][plane][lane][e] = part_plane(scale[o] * W[o = 32 t2 + (lane&31)][32 t + (r&3) + 8(r>>2) + 4(lane>>5)]), r = 8c + e
    const int tin = n_in / 32, tout = n_out / 32;
    std::vector<float> sc, un;
    feature_scales(W, n_out, n_in, sc, un);
    uint16_t* d16 = (uint16_t*)dst;
    for (int t = 0; t < tin; ++t)
        for (int c = 0; c < 2; ++c)
            for (int t2 = 0; t2 < tout; ++t2)
                for (int lane = 0; lane < 64; ++lane)
                    for (int e = 0; e < 8; ++e) {
                        const int r = 8 * c + e;
                        const int o = 32 * t2 + (lane & 31), k = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                        uint16_t parts[2];
                        split2_host(W[(size_t)o * n_in + k] * sc[o], parts);
                        const size_t sidx = (size_t)t * 2 + c;
                        for (int pl = 0; pl < kPlanes; ++pl) d16[((((sidx * tout + t2) * kPlanes + pl) * 64 + lane) * 8) + e] = parts[pl];
                    }
    write_epilogue_consts(dst + frag_floats(2 * tin, tout), tout, un, b);
}
static void pack_head(const float* W, const float* b, int n_in, int n_out, float* dst)
{
    // dst16[t][c][plane][lane][e] = part_plane(scale[o] * W[o = lane&31][32 t + (r&3) + 8(r>>2) + 4(lane>>5)]), r = 8c + e, rows
    // >= n_out are zero; then unscale[8] (1 for unused outputs), then bias[8] (0 for unused outputs)
    const int tin = n_in / 32;
    std::vector<float> sc, un;
    feature_scales(W, n_out, n_in, sc, un);
    uint16_t* d16 = (uint16_t*)dst;
    for (int t = 0; t < tin; ++t)
        for (int c = 0; c < 2; ++c)
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 8; ++e) {
                    const int r = 8 * c + e, o = lane & 31, k = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    uint16_t parts[2];
                    split2_host(o < n_out ? W[(size_t)o * n_in + k] * sc[o] : 0.0f, parts);
                    for (int pl = 0; pl < kPlanes; ++pl) d16[(((((size_t)t * 2 + c) * kPlanes + pl) * 64 + lane) * 8) + e] = parts[pl];
                }
    float* consts = dst + head_consts_off(tin);
    for (int i = 0; i < 8; ++i) { consts[i] = i < n_out ? un[i] : 1.0f; consts[8 + i] = i < n_out ? b[i] : 0.0f; }
}

int rl_policy_pack_impl(int kind, const float* sd, float* packed)
{
    if (kind < RL_DQN || kind > RL_PPO) { rl_set_error("unknown brain kind %d", kind); return RL_E_INVALID; }
    const Layout L = layout_of(kind);
    const float* p = sd;
    if (kind == RL_DQN) {
        pack_in_layer(p, p + 153 * 128, 128, packed + L.l1); p += 153 * 128 + 128;
        pack_hidden_layer(p, p + 128 * 64, 128, 64, packed + L.l2a); p += 128 * 64 + 64;
        pack_head(p, p + 64 * 8, 64, 8, packed + L.ha);
    } else if (kind == RL_D3QN || kind == RL_PERD3QN) {
        pack_in_layer(p, p + 153 * 128, 128, packed + L.l1); p += 153 * 128 + 128;
        pack_hidden_layer(p, p + 128 * 128, 128, 128, packed + L.l2a); p += 128 * 128 + 128;
        pack_head(p, p + 128 * 8, 128, 8, packed + L.ha); p += 128 * 8 + 8;
        pack_hidden_layer(p, p + 128 * 128, 128, 128, packed + L.l2b); p += 128 * 128 + 128;
        pack_head(p, p + 128, 128, 1, packed + L.hb);
    } else {
        pack_in_layer(p, p + 153 * 256, 256, packed + L.l1); p += 153 * 256 + 256;
        pack_hidden_layer(p, p + 256 * 256, 256, 256, packed + L.l2a); p += 256 * 256 + 256;
        pack_head(p, p + 256 * 8, 256, 8, packed + L.ha);  // fc_v is not evaluated when acting (PPO.py:164-169)
    }
    return RL_OK;
}

static int policy_grid(int64_t max_rows)  // tiles one brain can have: one 4-wave workgroup per 32-row tile
{
    const int64_t blocks = (max_rows + 31) / 32;
    return (int)(blocks < 1 ? 1 : blocks);
}

// Which kernel serves a stand-alone policy launch.
//
// ONE ARITHMETIC: the default ("auto") runs the tiles of the multi-tick kernel's policy half -- per output accumulator the partial products
// hi.lo, hi.hi, lo.hi of every 16-k chunk in chunk order into ONE f32 accumulator, heads as three product chains joined at the end
// (policy_tile1 / policy_tile1s / policy_tile1ds for the dueling kinds, policy_pair2 for DQN and PPO) -- so rl_policy_act + rl_tick,
// rl_run, one GPU or eight, 256 worlds or 4096 give the same Q values, probabilities and actions BIT FOR BIT:
//   pair    k_policy_pair: two waves per 32-row tile, brains of any kinds in one launch (the default below 1,536 tiles, and for DQN / PPO)
//   dense   k_policy_dense: four one-wave tiles of one dueling brain per workgroup, weights through LDS -- the same bits as `pair` for
//           the dueling kinds; the default from 1,536 tiles on (2,048 / 4,096 / 10,880 tiles: 32.4 / 60.9 / 148 us against 38.6 / 72.7 /
//           188 for the 4-wave tile)
//   wave    k_policy1: one wave per tile, no LDS (dueling kinds; the same bits again) -- measurement only
// (The 4-wave N-split tile of rounds 1-2 -- "nsplit", ~1e-7 away from these -- was removed in round 5: it was nobody's arithmetic any more.)
// The variant comes from the handle's option snapshot (rl_set_option / RL_POLICY_VARIANT at rl_create), not from getenv at the launch.
static bool kind_is_dueling(int kind) { return kind == RL_D3QN || kind == RL_PERD3QN; }
static int resolve_variant(int variant, bool all_dueling, int64_t expected_rows)
{
    if (variant == RL_PV_AUTO) return (all_dueling && expected_rows / 32 >= 1536) ? RL_PV_DENSE : RL_PV_PAIR;
    if ((variant == RL_PV_DENSE || variant == RL_PV_WAVE) && !all_dueling) return RL_PV_PAIR;   // (those kernels exist for the dueling kinds)
    return variant;
}
static int launch_check(const char* what)
{
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { rl_set_error("%s launch failed: %s", what, hipGetErrorString(e)); return RL_E_LAUNCH; }
    return RL_OK;
}
// One launch for the brains in `a` (<= kMaxBrainsPerLaunch); `variant` already resolved.
static int launch_policy(int variant, int kind, const PolicyArgs& a, int64_t max_rows, int64_t expected_rows, hipStream_t st)
{
    // grid = (tiles a brain can have at most, brains): the bound is several times the real tile count (a brain COULD own every
    // agent), but the ~2,500 empty workgroups cost < 1 us (a dense launch of the same 680 tiles without them: 17.6 vs 18.4 us).
    // Brain-fastest order (all real tiles dispatched first) is SLOWER: 21.6 vs 18.4 us.
    const dim3 grid(policy_grid(max_rows), a.nb);
    if (variant == RL_PV_PAIR) {
        int ex_bytes = kPairStageBytes;   // per tile: the staged rows (20.5 KB), then the exchange buffer of the widest kind (16 KB; PPO: 32 KB)
        for (int b = 0; b < a.nb; ++b) ex_bytes = ex_bytes > pair_ex_bytes(a.b[b].kind) ? ex_bytes : pair_ex_bytes(a.b[b].kind);
        const int lds = pair_kernel_lds(ex_bytes);
        if (lds > 64 * 1024) {   // opt in to the large dynamic-LDS window, once per device and size
            static int granted[64];
            int dev = 0;
            if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
            if (granted[dev] < lds) {
                const hipError_t e = hipFuncSetAttribute((const void*)k_policy_pair, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
                if (e != hipSuccess) { rl_set_error("hipFuncSetAttribute(%d bytes of LDS) failed: %s", lds, hipGetErrorString(e)); return RL_E_LAUNCH; }
                granted[dev] = lds;
            }
        }
        const dim3 gridp((grid.x + kPairTiles - 1) / kPairTiles, a.nb);
        hipLaunchKernelGGL(k_policy_pair, gridp, dim3(128 * kPairTiles), lds, st, a, ex_bytes);
        return launch_check("policy kernel (pair)");
    }
    if (variant == RL_PV_DENSE) {
        const dim3 gridd((grid.x + kDenseTiles - 1) / kDenseTiles, a.nb);
        hipLaunchKernelGGL((k_policy_dense<RL_PERD3QN>), gridd, dim3(64 * kDenseTiles), 0, st, a);   // (D3QN and PERD3QN: one network)
        return launch_check("policy kernel (dense)");
    }
    if (variant == RL_PV_WAVE) {
        hipLaunchKernelGGL((k_policy1<RL_PERD3QN>), grid, dim3(64), 0, st, a);
        return launch_check("policy kernel (wave)");
    }
    rl_set_error("policy launch: unknown variant %d", variant);
    return RL_E_INVALID;
}

int rl_policy_forward_impl(int kind, const float* packed, const float* obs, int64_t n_rows, float* out, hipStream_t st)
{
    if (kind < RL_DQN || kind > RL_PPO) { rl_set_error("unknown brain kind %d", kind); return RL_E_INVALID; }
    PolicyArgs a{};
    a.nb = 1; a.b[0].packed = packed; a.b[0].kind = kind; a.obs = obs; a.n_rows = n_rows; a.out = out; a.cap = 1;
    return launch_policy(resolve_variant(rl_options_current().policy_variant, kind_is_dueling(kind), n_rows), kind, a, n_rows, n_rows, st);
}

// work layout: int counts[2][64] (parity double buffer, zero-initialised once by the caller), then
// int lists[n_brains][n_worlds*slot_cap]
size_t rl_policy_work_bytes_impl(const rl_world* h)
{
    return 128 * sizeof(int) + (size_t)h->cfg.n_brains * h->cfg.n_worlds * h->cfg.slot_cap * sizeof(int);
}

int rl_policy_act_impl(rl_world* h, const rl_brain* brains, int n_brains, const float* obs, int8_t* actions, float* out_q,
                       void* work, hipStream_t st)
{
    const int R = h->cfg.n_worlds, cap = h->cfg.slot_cap;
    const int64_t stride = (int64_t)R * cap;
    int* lists = (int*)work + 128;
    int* counts;
    if (h->lists_valid && h->work == work) {
        counts = (int*)work + 64 * (h->lists_parity & 1);  // produced by the last world launch: no bucket kernel needed
    } else {
        const int cur = h->parity_next & 1;
        h->parity_next ^= 1;
        counts = (int*)work + 64 * cur;
        int* counts_zero = (int*)work + 64 * (cur ^ 1);
        hipLaunchKernelGGL(k_bucket, dim3((R + 3) / 4), dim3(256), 0, st, h->st.n_agents, h->st.a_brain, R, cap, n_brains, counts,
                           counts_zero, lists, stride);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) { rl_set_error("bucket kernel launch failed: %s", hipGetErrorString(e)); return RL_E_LAUNCH; }
        if (h->work == work) { h->lists_valid = 1; h->lists_parity = cur; }
    }
    // One brain can own every slot of every world: the reference bounds a RUNNING population by 2*max_agents+1
    // (environment.py:501 snapshot rule), but rl_reset_synthetic / a state written by the caller may hold up to slot_cap agents
    // of one brain, and a tile beyond the grid would silently keep its stale action.  Workgroups beyond a brain's row count
    // exit at once (~0.1 us per thousand).
    const int64_t bound = (int64_t)R * (int64_t)cap;
    const int64_t expected = (int64_t)R * h->cfg.max_agents;  // populations hover around max_agents
    auto base_args = [&]() {
        PolicyArgs a{};
        a.obs = obs; a.out = out_q; a.actions = actions; a.seed = h->cfg.seed; a.cap = cap; a.world_base = h->cfg.world_base;
        a.tick = h->st.tick; a.epoch = h->st.epoch;
#ifdef RL_PHASE_PROFILE
        a.prof = h->prof; a.prof_block = h->prof_world;
#endif
        return a;
    };
    auto slot_of = [&](int b) {
        BrainSlot s{};
        s.packed = brains[b].packed; s.rowlist = lists + b * stride; s.count_ptr = counts + b; s.eps = brains[b].epsilon;
        s.kind = brains[b].kind;
        return s;
    };
    for (int b = 0; b < n_brains; ++b)
        if (brains[b].kind < RL_DQN || brains[b].kind > RL_PPO) { rl_set_error("unknown brain kind %d", brains[b].kind); return RL_E_INVALID; }
    bool all_dueling = true;
    for (int b = 0; b < n_brains; ++b) all_dueling = all_dueling && kind_is_dueling(brains[b].kind);
    const int variant = resolve_variant(h->opt.policy_variant, all_dueling, expected);
    // one arithmetic (see launch_policy): the tiles of rl_run's policy half, brains of any kinds side by side in a launch
    PolicyArgs a = base_args();
    for (int b = 0; b < n_brains; ++b) {
        a.b[a.nb++] = slot_of(b);
        if (a.nb == kMaxBrainsPerLaunch || b + 1 == n_brains) {
            if (int rc = launch_policy(variant, RL_PERD3QN, a, bound, expected, st)) return rc;
            a.nb = 0;
        }
    }
    return RL_OK;
}
