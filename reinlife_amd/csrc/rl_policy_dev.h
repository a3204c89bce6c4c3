// rl_policy_dev.h -- device side of the policy (shared by rl_policy.hip and the fused tick of rl_world.hip): batched policy inference for ReinLife's brains on MI355X (gfx950), hand-written HIP + MFMA.
//
// Reference (paths under /root/reference/ReinLife/Models):
//   DQN      Qnet.forward            DQN.py:126-130      153 -> 128 -> 64 -> 8
//   D3QN     dueling_ddqn.forward    D3QN.py:161-165     153 -> 128 -> (128 -> 8 || 128 -> 1), q = adv + val - mean(adv)
//   PERD3QN  DuelingDDQN.forward     PERD3QN.py:198-202  (same network)
//   PPO      PPO.pi                  PPO.py:101-106      153 -> 256 -> 256 -> 8 -> softmax
//   action selection                 DQN.py:132-139, D3QN.py:167-173, PERD3QN.py:204-210, PPO.py:164-169
// The reference runs one batch-1 forward per agent; here a 4-wave workgroup owns 32 agents (observation rows):
// every wave computes a quarter of each layer's output features for those 32 rows ("N-split": 4x shorter dependency
// chain per tile and 4x more waves than one-wave-per-tile, which is what matters at 256 worlds = ~700 tiles on
// 1024 SIMDs), activations cross waves through LDS once per layer:
//
//   * fp32 results from the bf16 matrix pipe ("3 x bf16"): every f32 operand is split into three bf16 parts
//     x = hi + mid + lo (8 + 8 + 8 mantissa bits, each part the round-to-nearest bf16 of the remaining residual, so the
//     three parts carry the whole 24-bit f32 mantissa) and a product is the sum of the six partial products of weight
//     2^-16 or more (hi.hi, hi.mid, mid.hi, mid.mid, hi.lo, lo.hi), each EXACT in the f32 accumulator of
//     v_mfma_f32_32x32x16_bf16.  The dropped terms are below 2^-23 relative -- the size of f32's own rounding -- so the
//     result is f32-grade (measured error vs torch ~1e-6, the bar is 1e-5) at 6 x 32 cycles per 16 k instead of
//     8 x 64 with v_mfma_f32_32x32x2_f32: 2.7x the f32 matrix rate.  Weights are split once when packed, activations
//     once where they are produced (observation staging / layer output), so the split costs a few VALU ops per value.
//   * transposed formulation  H_out[feature][row] = W[feature][k] . H_in[k][row]: the WEIGHTS are the MFMA A operand
//     and the ACTIVATIONS the B operand.  The 32x32 f32 accumulator layout (lane = row, register r of half h =
//     feature (r&3) + 8(r>>2) + 4h) is dtype-independent, so a lane's registers 8c..8c+7 are exactly its 8 B-operand
//     values of K-chunk (tile, c) of the next layer when the next layer's weights are packed in that k order:
//     activations are never transposed; the weights are pre-packed so every A fragment is one coalesced
//     16-byte-per-lane load (rl_policy_pack_weights).
//   * the narrow heads (8 / 1 outputs) run on the VALU (an MFMA tile would be 75-97 % padding), followed by the
//     dueling combine / softmax and the epsilon-greedy / categorical draw (Philox) in the same kernel.
#pragma once
#include "rl_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
// weight pointers come out of a runtime-indexed brain table, which makes them generic (flat_load, counted on lgkmcnt AND
// vmcnt); they always point to device global memory, so say so: global_load + a prefetch ring that survives LDS barriers
typedef const float __attribute__((address_space(1))) gfloat;
typedef const f32x4 __attribute__((address_space(1))) gf32x4;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int kInChunks = 10;    // input layer: K padded to 160 = 10 chunks of 16 (lanes 0-31: k 16c..16c+7, lanes 32-63: +8)
constexpr int kBiasK = 153;      // x[153] := 1, W[:,153] := bias

// packed sizes in 4-byte units: a fragment is 8 bf16 = 16 bytes per lane, three planes (hi, mid, lo) per fragment
__host__ __device__ constexpr int64_t in_layer_floats(int tiles) { return (int64_t)kInChunks * tiles * 3 * 64 * 4; }
__host__ __device__ constexpr int64_t hid_layer_floats(int tin, int tout) { return (int64_t)(2 * tin) * tout * 3 * 64 * 4 + (int64_t)tout * 32; }
__host__ __device__ constexpr int64_t head_floats(int tin, int nout) { return (int64_t)tin * 2 * 3 * 64 * 4 + nout; }  // fragments + f32 bias
__host__ __device__ constexpr int64_t head_bias_off(int tin) { return (int64_t)tin * 2 * 3 * 64 * 4; }

struct Layout {  // offsets (floats) into a brain's packed buffer
    int64_t l1, l2a, l2b, ha, hb, total;
};
__host__ __device__ inline Layout layout_of(int kind)
{
    Layout L{};
    int64_t o = 0;
    if (kind == RL_DQN) {
        L.l1 = o; o += in_layer_floats(4);
        L.l2a = o; o += hid_layer_floats(4, 2);
        L.ha = o; o += head_floats(2, 8);
    } else if (kind == RL_D3QN || kind == RL_PERD3QN) {
        L.l1 = o; o += in_layer_floats(4);
        L.l2a = o; o += hid_layer_floats(4, 4);
        L.ha = o; o += head_floats(4, 8);
        L.l2b = o; o += hid_layer_floats(4, 4);
        L.hb = o; o += head_floats(4, 1);
    } else {
        L.l1 = o; o += in_layer_floats(8);
        L.l2a = o; o += hid_layer_floats(8, 8);
        L.ha = o; o += head_floats(8, 8);
    }
    L.total = o;
    return L;
}

// ---------------------------------------------------------------------------------------------------------------
// device building blocks (everything fully unrolled: accumulators must stay in registers)
// ---------------------------------------------------------------------------------------------------------------
// LDS-only workgroup barrier: lds_barrier() would also wait vmcnt(0), i.e. drain the weight-prefetch ring at every
// layer boundary; waves only exchange activations / partial sums through LDS.
#ifdef RL_FULL_FENCE
__device__ inline void lds_barrier() { __syncthreads(); }
#else
__device__ inline void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
#endif

__device__ inline f32x16 mfma(const f32x4& a, const f32x4& b, f32x16 c)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// x = hi + mid + lo, each the bf16 nearest to what is left (the subtractions are exact in f32)
__device__ inline void split3(float x, __bf16& hi, __bf16& mid, __bf16& lo)
{
    hi = (__bf16)x;
    const float r1 = x - (float)hi;
    mid = (__bf16)r1;
    lo = (__bf16)(r1 - (float)mid);
}

// the six partial products of one K-chunk, three per accumulator chain (a / b: planes hi, mid, lo)
__device__ inline void mfma6(const f32x4 (&a)[3], const f32x4 (&b)[3], f32x16& acc, f32x16& acc2)
{
#ifdef RL_ABL_MFMA  // tuning experiment: no matrix instructions (results are WRONG)
    acc[0] += a[0][0] + a[1][0] + a[2][0] + b[0][0] + b[1][0] + b[2][0]; acc2[0] += 1.0f;
    return;
#endif
    acc2 = mfma(a[0], b[2], acc2);  // hi.lo
    acc = mfma(a[2], b[0], acc);    // lo.hi
    acc2 = mfma(a[1], b[1], acc2);  // mid.mid
    acc = mfma(a[1], b[0], acc);    // mid.hi
    acc2 = mfma(a[0], b[1], acc2);  // hi.mid
    acc = mfma(a[0], b[0], acc);    // hi.hi
}

// Observation tile in LDS, split and already in B-operand order: plane p (hi, mid, lo) holds, for K-chunk c and lane
// (row j = lane&31, half kh = lane>>5), the 8 bf16 x[j][16c + 8kh + 0..7] as one 16-byte unit at
// p*kXPlane + (2c + kh)*33 + j, with x[153] := 1 (bias input) and x[154..159] := 0.  Groups are 33 (not 32) units
// apart so that the staging writes of one row (40 lanes, one 8-byte half unit each) spread over the banks.
constexpr int kXGroup = 33;
constexpr int kXPlane = 2 * kInChunks * kXGroup;
constexpr int kXsUnits = 3 * kXPlane;

// Stage the 32 rows of a tile: wave v loads rows 8v..8v+7, ONE coalesced 612-byte read per row (lane m reads floats
// 4m..4m+3), instead of every wave gathering 16 bytes per lane from 32 different rows for each K-step (64 cache lines
// per load instruction, four times over): the input layer was request-bound on the vector L1.
// `row_of_lane`: observation row id of tile row (lane & 31).
__device__ inline void stage_x(f32x4* __restrict__ xs, const float* __restrict__ obs, int64_t row_of_lane, int lane, int v)
{
    const int lo = (int)(row_of_lane & 0xffffffff), hi = (int)(row_of_lane >> 32);
    f32x2* x2 = (f32x2*)xs;
    const int unit0 = (lane >> 1) * kXGroup + 8 * v, half = lane & 1;  // lane m: floats 4m..4m+3 = half (m&1) of group m>>1
    const int off = lane < 38 ? 4 * lane : 149;  // every lane loads (lanes >= 38 read floats 149..152, inside the row): not
                                                 // predicated -- a predicated load drags its first use, and a wait, up to itself
    {
        constexpr int r0 = 0;
        f32x4 val[8];  // all eight rows of this wave in flight: one round trip
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) {
            const int jj = 8 * v + r0 + rr;
            const int64_t r = ((int64_t)__builtin_amdgcn_readlane(hi, jj) << 32) | (unsigned)__builtin_amdgcn_readlane(lo, jj);
#ifdef RL_ABL_X  // tuning experiment: no observation reads (results are WRONG)
            val[rr] = f32x4{(float)r, 1.0f, 2.0f, 3.0f};
#else
            val[rr] = *(const f32x4u*)(obs + r * RL_OBS_DIM + off);
#endif
        }
        __builtin_amdgcn_sched_barrier(0);
        if (lane < 40) {
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) {
                f32x4 t = val[rr];
                if (lane == 38) t = f32x4{t.w, 1.0f, 0.0f, 0.0f};  // k = 152, the bias input, padding
                else if (lane == 39) t = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                bf16x4 ph, pm, pl;
#pragma unroll
                for (int e = 0; e < 4; ++e) { __bf16 a, b, c; split3(t[e], a, b, c); ph[e] = a; pm[e] = b; pl[e] = c; }
                const int u = (unit0 + r0 + rr) * 2 + half;
                x2[u] = __builtin_bit_cast(f32x2, ph);
                x2[kXPlane * 2 + u] = __builtin_bit_cast(f32x2, pm);
                x2[kXPlane * 4 + u] = __builtin_bit_cast(f32x2, pl);
            }
        }
    }
}

// The packed weights are STEP-major: [K-chunk][output tile][plane][lane][8 bf16], so the fragments of one chunk are
// 1 KiB apart (immediate offsets of one running pointer).  The pointer is made opaque at every step so that the
// compiler neither precomputes nor hoists hundreds of 64-bit addresses, and a sched_barrier per step bounds the
// prefetch distance to exactly one step.
//
// layer_in: this wave computes output tiles {t0, t0 + TSTRIDE, ...} (NT of them) of a layer with TOUT tiles; the
// B operand comes from the staged observation tile.  D = prefetch ring depth in K-chunks: a chunk is only 6*NT MFMAs
// (192*NT cycles), an L2 round trip under load is several times that, so D chunks of weights are kept in flight.
// Weight prefetch ring of one layer for one wave: D K-chunks of A fragments (3 planes each) in flight.  start() only
// needs the packed pointer, so it is issued BEFORE the wait that precedes the layer (observation staging, the LDS
// exchange of the previous layer, the VALU head): the first L2 round trip of every layer overlaps that wait.
template <int TOUT, int NT, int TSTRIDE, int D>
struct WRing {
    f32x4 a[D][NT][3];
    gf32x4* p;
    __device__ inline void start(gfloat* __restrict__ pw, int lane, int t0)
    {
#ifdef RL_ABL_W  // tuning experiment: every chunk re-reads the first 3 KiB of the layer (cache hits; results are WRONG)
        p = (gf32x4*)pw + lane;
#else
        p = (gf32x4*)pw + t0 * 3 * 64 + lane;
#endif
#pragma unroll
        for (int d = 0; d < D; ++d)
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) a[d][t][pl] = p[((d * TOUT + t * TSTRIDE) * 3 + pl) * 64];
    }
    // take chunk s out of the ring and refill its slot with chunk s + D (of NS)
    template <int NS>
    __device__ inline void next(int s, f32x4 (&ac)[NT][3])
    {
        const int cur = s % D;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) ac[t][pl] = a[cur][t][pl];
#ifndef RL_ABL_W
        p += TOUT * 3 * 64;
#endif
        asm volatile("" : "+v"(p));
        if (s + D < NS) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) a[cur][t][pl] = p[(((D - 1) * TOUT + t * TSTRIDE) * 3 + pl) * 64];
        }
    }
};

template <int TOUT, int NT, int TSTRIDE, int D>
__device__ inline void layer_in(WRing<TOUT, NT, TSTRIDE, D>& w, int lane, const f32x4* __restrict__ xs, f32x16 (&acc)[NT])
{
    f32x16 acc2[NT];  // second accumulator chain: consecutive MFMAs of one wave do not wait for each other's result
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[t][r] = 0.0f; acc2[t][r] = 0.0f; }
    f32x4 x[2][3];
    const int xb = (lane >> 5) * kXGroup + (lane & 31);
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) x[0][pl] = xs[pl * kXPlane + xb];
#pragma unroll
    for (int c = 0; c < kInChunks; ++c) {
        f32x4 ac[NT][3], xc[3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) xc[pl] = x[c & 1][pl];
        w.template next<kInChunks>(c, ac);
        if (c + 1 < kInChunks) {
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) x[(c + 1) & 1][pl] = xs[pl * kXPlane + xb + (c + 1) * 2 * kXGroup];
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) mfma6(ac[t], xc, acc[t], acc2[t]);
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] += acc2[t][r];
}

template <int NT>
__device__ inline void relu_inplace(f32x16 (&h)[NT])
{
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) h[t][r] = fmaxf(h[t][r], 0.0f);
}

// Publish this wave's activation tile `t` to the workgroup, split: plane p unit (t*2 + c)*64 + lane = registers
// 8c..8c+7, i.e. exactly the B-operand fragment of K-chunk (t, c) of the next layer for this lane.
__device__ inline void publish_tile(f32x4* lds, int plane_units, int t, int lane, const f32x16& h)
{
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        bf16x8 ph, pm, pl;
#pragma unroll
        for (int e = 0; e < 8; ++e) { __bf16 x, y, z; split3(h[8 * c + e], x, y, z); ph[e] = x; pm[e] = y; pl[e] = z; }
        const int u = (t * 2 + c) * 64 + lane;
        lds[u] = __builtin_bit_cast(f32x4, ph);
        lds[plane_units + u] = __builtin_bit_cast(f32x4, pm);
        lds[2 * plane_units + u] = __builtin_bit_cast(f32x4, pl);
    }
}

// layer_hidden: input = TIN published tiles in LDS (three planes of TIN*128 units); this wave computes output tiles
// {t0, t0 + TSTRIDE, ...}.  The accumulators start from the bias (packed in accumulator order, behind the fragments).
template <int TIN, int TOUT, int NT, int TSTRIDE, int D>
__device__ inline void layer_hidden(WRing<TOUT, NT, TSTRIDE, D>& w, gfloat* __restrict__ pw, int lane, int t0,
                                    const f32x4* __restrict__ hin, f32x16 (&acc)[NT])
{
    constexpr int NS = TIN * 2;        // chunk s = t*2 + c covers input features 32t + (r&3) + 8(r>>2) + 4h, r = 8c..8c+7
    constexpr int PS = TIN * 2 * 64;   // units per plane
    gf32x4* bias = (gf32x4*)(pw + (int64_t)NS * TOUT * 3 * 64 * 4) + (t0 * 2 + (lane >> 5)) * 4;
    f32x4 b[2][3];
    f32x16 acc2[NT];
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) b[0][pl] = hin[pl * PS + lane];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 bq = bias[t * TSTRIDE * 8 + q];
#pragma unroll
            for (int e = 0; e < 4; ++e) { acc[t][4 * q + e] = bq[e]; acc2[t][4 * q + e] = 0.0f; }
        }
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        f32x4 ac[NT][3], bc[3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) bc[pl] = b[s & 1][pl];
        w.template next<NS>(s, ac);
        if (s + 1 < NS) {
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) b[(s + 1) & 1][pl] = hin[pl * PS + (s + 1) * 64 + lane];
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) mfma6(ac[t], bc, acc[t], acc2[t]);
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] += acc2[t][r];
}

// Narrow heads (8 / 1 outputs) also run on the matrix pipe: the head's weight rows are the A operand (outputs padded
// to 32 rows with zeros), the B operand is this wave's own activation registers (a lane's registers 8c..8c+7 are its
// B fragment of chunk c -- no exchange needed), 12 MFMAs per 32 input features.  A VALU version (16 FMAs per output
// and input tile, weights fetched inside the loop) took 2-4 k cycles of mostly load latency per head.
// Result: out[r], r = 0..3 = this wave's partial sum of output 4*(lane>>5) + r for row lane&31.
template <int NT, int TSTRIDE>
struct HeadW {
    f32x4 a[NT][2][3];
    __device__ inline void start(gfloat* __restrict__ hw, int lane, int t0)
    {
        gf32x4* p = (gf32x4*)hw + lane;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) a[t][c][pl] = p[(((t0 + t * TSTRIDE) * 2 + c) * 3 + pl) * 64];
    }
};

template <int NT, int TSTRIDE>
__device__ inline void head_mfma(const HeadW<NT, TSTRIDE>& w, const f32x16 (&hin)[NT], float (&out)[4])
{
    f32x16 acc, acc2;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[r] = 0.0f; acc2[r] = 0.0f; }
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            bf16x8 ph, pm, pl;
#pragma unroll
            for (int e = 0; e < 8; ++e) { __bf16 x, y, z; split3(hin[t][8 * c + e], x, y, z); ph[e] = x; pm[e] = y; pl[e] = z; }
            const f32x4 b[3] = {__builtin_bit_cast(f32x4, ph), __builtin_bit_cast(f32x4, pm), __builtin_bit_cast(f32x4, pl)};
            mfma6(w.a[t][c], b, acc, acc2);
        }
#pragma unroll
    for (int r = 0; r < 4; ++r) out[r] = acc[r] + acc2[r];
}

// ---------------------------------------------------------------------------------------------------------------
// one 32-row tile, executed by 4 waves (v = 0..3) that share `lds_h` / `lds_part`
// ---------------------------------------------------------------------------------------------------------------
#ifndef RL_RING_D
#define RL_RING_D 3   // K-chunks of weights in flight per wave and layer (dueling brains)
#endif

struct TileIO {
    gfloat* packed;        // the brain's packed weights
    const float* obs;      // observation rows (153 floats each)
    int64_t row;           // observation row of tile row (lane & 31); a valid row even where `valid` is false
    bool valid;            // tile row (lane & 31) exists
    float eps;             // epsilon of the brain (ignored by PPO)
    float* out;            // [row][8] Q values / probabilities, or nullptr
    int8_t* actions;       // [row] selected action, or nullptr
    uint64_t seed;         // Philox key of this lane's row: (seed, epoch; world, tick, RL_SITE_ACT, index)
    uint32_t key_world, key_tick, key_epoch, key_index;
#ifdef RL_PHASE_PROFILE
    long long* prof;       // tuning build: shader-clock stamps (slots 48..), non-null in the profiled workgroup only
#endif
};

#ifdef RL_PHASE_PROFILE
#define RL_PMARK(i) do { if (io.prof && threadIdx.x == 0) io.prof[48 + (i)] = (long long)clock64(); } while (0)
#else
#define RL_PMARK(i) do { } while (0)
#endif

constexpr __host__ __device__ int policy_lds_units(int kind)  // f32x4 units of lds_h one tile needs
{
    return (3 * (kind == RL_PPO ? 8 : 4) * 2 * 64 > kXsUnits) ? 3 * (kind == RL_PPO ? 8 : 4) * 2 * 64 : kXsUnits;
}

// Every layer's weight ring and head fragments are requested BEFORE the wait that precedes the layer (observation
// staging, the LDS exchange, the previous head): ~154 VGPRs for the 128-wide brains, i.e. 3 waves per SIMD.  A variant
// that starts the rings at their layer (128 VGPRs, 4 waves per SIMD) and one with 64-row tiles were measured slower at
// 256 AND at 4096 worlds.
// GUARD: the tile may be inactive (a world kernel running fewer tiles than it has wave quads): every workgroup barrier
// is still executed, everything else is skipped.  The barriers are workgroup-wide, so all tiles of a workgroup must run
// the same KIND.
template <int KIND, bool GUARD>
__device__ inline void policy_tile(const TileIO& io, bool active, f32x4* __restrict__ lds_h, float (*__restrict__ lds_part)[32][9],
                                   int lane, int v)
{
    constexpr int HID_TILES = KIND == RL_PPO ? 8 : 4;
    constexpr int PS = HID_TILES * 2 * 64;  // units per plane of the published activations
    const int h = lane >> 5, j = lane & 31;
    const Layout L = layout_of(KIND);
    gfloat* __restrict__ packed = io.packed;
    const bool on = !GUARD || active;
    if (KIND == RL_DQN) {
        f32x16 h1[1], h2[1];
        WRing<4, 1, 1, 3> w1;
        WRing<2, 1, 1, 3> w2;
        HeadW<1, 1> wh;
        float q4[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (on) {
            w1.start(packed + L.l1, lane, v);
            stage_x(lds_h, io.obs, io.row, lane, v);
        }
        lds_barrier();
        if (on) {
            layer_in(w1, lane, lds_h, h1);
            if (v < 2) { w2.start(packed + L.l2a, lane, v); wh.start(packed + L.ha, lane, v); }
            relu_inplace<1>(h1);
        }
        lds_barrier();  // every wave is done with the observation tile: its LDS becomes the activation exchange
        if (on) publish_tile(lds_h, PS, v, lane, h1[0]);
        lds_barrier();
        if (on) {
            if (v < 2) {  // the second hidden layer has 2 output tiles: waves 0 and 1
                layer_hidden<4>(w2, packed + L.l2a, lane, v, lds_h, h2);
                relu_inplace<1>(h2);
                head_mfma(wh, h2, q4);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) lds_part[v][j][4 * h + r] = q4[r];
        }
    } else if (KIND == RL_D3QN || KIND == RL_PERD3QN) {
        f32x16 h1[1], h2[1];
        float adv[4], val[4];
        WRing<4, 1, 1, RL_RING_D> w1, w2;
        HeadW<1, 1> wh;
        if (on) {
            w1.start(packed + L.l1, lane, v);
            stage_x(lds_h, io.obs, io.row, lane, v);
        }
        lds_barrier();
        RL_PMARK(10);
        if (on) {
            layer_in(w1, lane, lds_h, h1);
            RL_PMARK(2);
            w2.start(packed + L.l2a, lane, v);
            wh.start(packed + L.ha, lane, v);
            relu_inplace<1>(h1);  // relu(feature) feeds both branches (PERD3QN.py:200-201)
        }
        lds_barrier();
        if (on) publish_tile(lds_h, PS, v, lane, h1[0]);
        lds_barrier();
        RL_PMARK(3);
        if (on) {
            layer_hidden<4>(w2, packed + L.l2a, lane, v, lds_h, h2);
            RL_PMARK(4);
            w1.start(packed + L.l2b, lane, v);  // the value branch's first chunks arrive while the advantage head runs
            relu_inplace<1>(h2);
            head_mfma(wh, h2, adv);
            wh.start(packed + L.hb, lane, v);
            RL_PMARK(5);
            layer_hidden<4>(w1, packed + L.l2b, lane, v, lds_h, h2);
            RL_PMARK(6);
            relu_inplace<1>(h2);
            head_mfma(wh, h2, val);
            RL_PMARK(7);
#pragma unroll
            for (int r = 0; r < 4; ++r) lds_part[v][j][4 * h + r] = adv[r];
            if (h == 0) lds_part[v][j][8] = val[0];
        }
    } else {
        f32x16 h1[2], h2[2];
        float q4[4];
        WRing<8, 2, 4, 3> w1, w2;
        HeadW<2, 4> wh;
        if (on) {
            w1.start(packed + L.l1, lane, v);   // tiles v and v+4
            stage_x(lds_h, io.obs, io.row, lane, v);
        }
        lds_barrier();
        if (on) {
            layer_in(w1, lane, lds_h, h1);
            w2.start(packed + L.l2a, lane, v);
            relu_inplace<2>(h1);
        }
        lds_barrier();
        if (on) {
            publish_tile(lds_h, PS, v, lane, h1[0]);
            publish_tile(lds_h, PS, v + 4, lane, h1[1]);
        }
        lds_barrier();
        if (on) {
            wh.start(packed + L.ha, lane, v);
            layer_hidden<8>(w2, packed + L.l2a, lane, v, lds_h, h2);
            relu_inplace<2>(h2);
            head_mfma(wh, h2, q4);
#pragma unroll
            for (int r = 0; r < 4; ++r) lds_part[v][j][4 * h + r] = q4[r];
        }
    }
    lds_barrier();
    RL_PMARK(8);
    if (on && v == 0 && h == 0) {
        float q[8];
        float sum9[9];
#pragma unroll
        for (int i = 0; i < 9; ++i)
            sum9[i] = (i < 8 || KIND == RL_D3QN || KIND == RL_PERD3QN)
                          ? ((lds_part[0][j][i] + lds_part[1][j][i]) + lds_part[2][j][i]) + lds_part[3][j][i] : 0.0f;
        if (KIND == RL_D3QN || KIND == RL_PERD3QN) {
            gfloat* ba = packed + L.ha + head_bias_off(4);
            const float bv = packed[L.hb + head_bias_off(4)];
            float adv[8], mean = 0.0f;  // advantage.mean() of the [1,8] tensor == per-row mean when batched
#pragma unroll
            for (int i = 0; i < 8; ++i) { adv[i] = sum9[i] + ba[i]; mean += adv[i]; }
            mean *= 0.125f;
            const float val = sum9[8] + bv;
#pragma unroll
            for (int i = 0; i < 8; ++i) q[i] = adv[i] + val - mean;
        } else {
            gfloat* bq = packed + L.ha + head_bias_off(KIND == RL_DQN ? 2 : 8);
#pragma unroll
            for (int i = 0; i < 8; ++i) q[i] = sum9[i] + bq[i];
            if (KIND == RL_PPO) {
                float m = q[0], sm = 0.0f;  // softmax over the 8 logits (PPO.py:105)
#pragma unroll
                for (int i = 1; i < 8; ++i) m = fmaxf(m, q[i]);
#pragma unroll
                for (int i = 0; i < 8; ++i) { q[i] = expf(q[i] - m); sm += q[i]; }
#pragma unroll
                for (int i = 0; i < 8; ++i) q[i] = q[i] / sm;
            }
        }
        if (io.valid) {
            if (io.out) {
                f32x4* o = (f32x4*)(io.out + io.row * 8);
                o[0] = f32x4{q[0], q[1], q[2], q[3]};
                o[1] = f32x4{q[4], q[5], q[6], q[7]};
            }
            if (io.actions) {
                const rl_u4 r = rl_philox4x32(io.seed, io.key_epoch, io.key_world, io.key_tick, RL_SITE_ACT, io.key_index);
                const float u = (float)rl_u24(r.x);
                int a = 0;
                if (KIND == RL_PPO) {  // Categorical(prob).sample() as inverse CDF
                    float cum = 0.0f; a = 7; bool found = false;
#pragma unroll
                    for (int i = 0; i < 8; ++i) { cum += q[i]; if (!found && u < cum) { a = i; found = true; } }
                } else if (u < io.eps) a = (int)(r.y >> 29);
                else {
#pragma unroll
                    for (int i = 1; i < 8; ++i) if (q[i] > q[a]) a = i;  // first maximum
                }
                io.actions[io.row] = (int8_t)a;
            }
        }
    }
    RL_PMARK(9);
    lds_barrier();  // lds_h / lds_part are reused by the next tile
}

}  // namespace
