// rl_policy_dev.h -- device side of the policy (shared by rl_policy.hip and the fused tick of rl_world.hip): batched policy inference for ReinLife's brains on MI355X (gfx950), hand-written HIP + MFMA.
//
// Reference (paths under /root/reference/ReinLife/Models):
//   DQN      Qnet.forward            DQN.py:126-130      153 -> 128 -> 64 -> 8
//   D3QN     dueling_ddqn.forward    D3QN.py:161-165     153 -> 128 -> (128 -> 8 || 128 -> 1), q = adv + val - mean(adv)
//   PERD3QN  DuelingDDQN.forward     PERD3QN.py:198-202  (same network)
//   PPO      PPO.pi                  PPO.py:101-106      153 -> 256 -> 256 -> 8 -> softmax
//   action selection                 DQN.py:132-139, D3QN.py:167-173, PERD3QN.py:204-210, PPO.py:164-169
// The reference runs one batch-1 forward per agent; here 32 agents (observation rows) of one brain form a TILE, and a tile is computed by
// TWO waves that share a SIMD (policy_tile1s<PAIR> for the dueling kinds, policy_pair2 for DQN / PPO: each role owns half of every layer's
// output features, the halves meet through LDS twice per tile) -- inside the multi-tick kernel (k_run's policy half: up to four tiles of a
// world on its workgroup's eight waves) and in the stand-alone launches (k_policy_pair: four tiles of one brain per 512-thread workgroup;
// k_policy_dense from 1,536 dueling tiles on).  One arithmetic everywhere.  (The 4-wave "N-split" tile of rounds 1-2,
// every wave a quarter of each layer, was deleted in round 5.)
//
//   * f32-grade results from the f16 matrix pipe ("2 x f16, block-scaled"): every f32 operand row is scaled by a power
//     of two so that its largest element lands in [2^10, 2^11) -- per observation / activation ROW (the B operand's
//     columns) and per output FEATURE of the weights (the A operand's rows), so the scale factors leave the MFMA as one
//     multiply per accumulator register -- and split into two f16 parts x = hi + lo (hi = x rounded toward zero to 11
//     bits, lo = the exact remainder, again up to 11 bits: 22 bits of every operand, measured against the row
//     maximum).  A product is hi.hi + hi.lo + lo.hi, three v_mfma_f32_32x32x16_f16 per 16 k with f32 accumulation (the
//     cross terms in their own accumulator); the dropped lo.lo term is 2^-22 relative.  Measured
//     difference to the reference's torch outputs on the golden vectors: ~1e-7 (the bar is 1e-5) -- the same as an f32
//     numpy forward.  The power-of-two scaling makes the scheme independent of the operands' magnitude (f16 alone
//     would overflow at 65504 and lose small values).  History: f32-input MFMA (8 x 64 cycles per 16 k) -> three bf16
//     planes, six products (6 x 32) -> this (3 x 32 cycles, 4 bytes per weight instead of 6).
//   * transposed formulation  H_out[feature][row] = W[feature][k] . H_in[k][row]: the WEIGHTS are the MFMA A operand
//     and the ACTIVATIONS the B operand.  The 32x32 f32 accumulator layout (lane = row, register r of half h =
//     feature (r&3) + 8(r>>2) + 4h) is dtype-independent, so a lane's registers 8c..8c+7 are exactly its 8 B-operand
//     values of K-chunk (tile, c) of the next layer when the next layer's weights are packed in that k order:
//     activations are never transposed; the weights are pre-packed so every A fragment is one coalesced
//     16-byte-per-lane load (rl_policy_pack_weights).
//   * the narrow heads (8 / 1 outputs) also run on the matrix pipe (head_mfma), followed by the dueling combine / softmax
//     and the epsilon-greedy / categorical draw (Philox) in the same kernel.
#pragma once
#include "rl_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
// weight pointers come out of a runtime-indexed brain table, which makes them generic (flat_load, counted on lgkmcnt AND
// vmcnt); they always point to device global memory, so say so: global_load + a prefetch ring that survives LDS barriers
typedef const float __attribute__((address_space(1))) gfloat;
typedef const f32x4 __attribute__((address_space(1))) gf32x4;

constexpr int kInChunks = 10;        // input layer: K = 153 padded to 160 = 10 chunks of 16 (lanes 0-31: k 16c..16c+7, lanes 32-63: +8)
constexpr int kPlanes = 2;           // hi, lo
constexpr int kScaleExp = 10;        // a scaled row's maximum lies in [2^10, 2^11)

// packed sizes in 4-byte units: a fragment is 8 f16 = 16 bytes per lane, two planes per fragment; every output tile of
// an MFMA layer carries 64 epilogue constants [half h][unscale 16 | bias 16] in accumulator order
__host__ __device__ constexpr int64_t frag_floats(int chunks, int tout) { return (int64_t)chunks * tout * kPlanes * 64 * 4; }
__host__ __device__ constexpr int64_t in_layer_floats(int tout) { return frag_floats(kInChunks, tout) + (int64_t)tout * 64; }
__host__ __device__ constexpr int64_t hid_layer_floats(int tin, int tout) { return frag_floats(2 * tin, tout) + (int64_t)tout * 64; }
// head: fragments (output rows >= n_out are zero), then unscale[8], then bias[8]
__host__ __device__ constexpr int64_t head_consts_off(int tin) { return frag_floats(2 * tin, 1); }
__host__ __device__ constexpr int64_t head_floats(int tin, int nout) { return head_consts_off(tin) + 16; }

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

// power-of-two scale of a row whose largest magnitude is mx: s = 2^(kScaleExp - exponent(mx)), r = 1 / s.  Host and device
// use the same rule.  mx == 0 (or tiny) is clamped to an exponent that keeps both factors finite.
__host__ __device__ inline void row_scale(float mx, float& s, float& r)
{
    union { float f; int i; } u;
    u.f = mx;
    int eb = (u.i >> 23) & 0xff;
    eb = eb < 32 ? 32 : (eb > 230 ? 230 : eb);
    u.i = (254 + kScaleExp - eb) << 23; s = u.f;
    u.i = (eb - kScaleExp) << 23; r = u.f;
}

// ---------------------------------------------------------------------------------------------------------------
// device building blocks (everything fully unrolled: accumulators must stay in registers)
// ---------------------------------------------------------------------------------------------------------------
// LDS-only workgroup barrier: lds_barrier() would also wait vmcnt(0), i.e. drain the weight-prefetch ring at every
// layer boundary; waves only exchange activations / partial sums through LDS.
#ifndef RL_HAVE_LDS_BARRIER  /* rl_world.hip defines the same barrier before including this header */
#ifdef RL_FULL_FENCE
__device__ inline void lds_barrier() { __syncthreads(); }
#else
__device__ inline void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
#endif
#endif

__device__ inline f32x16 mfma16(const f32x4& a, const f32x4& b, f32x16 c)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// Two values and their row's power-of-two scale -> packed f16 pairs: hi = f16(x * s) (round to nearest even), lo = f16(x * s - hi)
// with x * s - hi evaluated exactly (one fused multiply-add in f32).  Four v_fma_mix instructions per pair -- the mixed-precision
// FMA reads f32 and f16 sources and writes an f16 half directly; the earlier sequence (2 multiplies, cvt_pkrtz, 2 cvt back, 2
// subtractions, cvt_pkrtz) was eight, and these splits are a third of the one-wave tile's instruction stream.
// lo of an element m times smaller than the row maximum is an f16 subnormal from m > 2^14 on: what is lost there is below
// 2^-24 of the row maximum, i.e. below f32's own resolution of the dot product.
__device__ inline void split_pair(float x0, float x1, float sc, unsigned& hi, unsigned& lo)
{
    unsigned h = 0, l = 0;
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h) : "v"(x0), "v"(sc));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h) : "v"(x1), "v"(sc));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(l) : "v"(x0), "v"(sc), "v"(h));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(x1), "v"(sc), "v"(h));
#endif
    hi = h; lo = l;
}
__device__ inline void split8(const float (&x)[8], float sc, f32x4& hi, f32x4& lo)
{
    unsigned h[4], l[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) split_pair(x[2 * q], x[2 * q + 1], sc, h[q], l[q]);
    hi = f32x4{__builtin_bit_cast(float, h[0]), __builtin_bit_cast(float, h[1]), __builtin_bit_cast(float, h[2]), __builtin_bit_cast(float, h[3])};
    lo = f32x4{__builtin_bit_cast(float, l[0]), __builtin_bit_cast(float, l[1]), __builtin_bit_cast(float, l[2]), __builtin_bit_cast(float, l[3])};
}

// The packed weights are STEP-major: [K-chunk][output tile][plane][lane][8 f16], so the fragments of one chunk are
// 1 KiB apart (immediate offsets of one running pointer).  The pointer is made opaque at every step so that the
// compiler neither precomputes nor hoists hundreds of 64-bit addresses, and a sched_barrier per step bounds the
// prefetch distance to exactly one step.
//
// Weight prefetch ring of one layer for one wave: D K-chunks of A fragments (2 planes each) in flight.  start() only
// needs the packed pointer, so it is issued BEFORE the wait that precedes the layer (observation staging, the LDS
// exchange of the previous layer, the head): the first L2 round trip of every layer overlaps that wait.
template <int TOUT, int NT, int TSTRIDE, int D>
struct WRing {
    f32x4 a[D][NT][kPlanes];
    gf32x4* p;
    __device__ inline void start(gfloat* __restrict__ pw, int lane, int t0)
    {
#ifdef RL_ABL_W  // tuning experiment: every chunk re-reads the first fragments of the layer (cache hits; results are WRONG)
        p = (gf32x4*)pw + lane;
#else
        p = (gf32x4*)pw + t0 * kPlanes * 64 + lane;
#endif
#pragma unroll
        for (int d = 0; d < D; ++d)
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int pl = 0; pl < kPlanes; ++pl) a[d][t][pl] = p[((d * TOUT + t * TSTRIDE) * kPlanes + pl) * 64];
    }
    // take chunk s out of the ring and refill its slot with chunk s + D (of NS)
    template <int NS>
    __device__ inline void next(int s, f32x4 (&ac)[NT][kPlanes])
    {
        const int cur = s % D;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int pl = 0; pl < kPlanes; ++pl) ac[t][pl] = a[cur][t][pl];
#ifndef RL_ABL_W
        p += TOUT * kPlanes * 64;
#endif
        asm volatile("" : "+v"(p));
        if (s + D < NS) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int pl = 0; pl < kPlanes; ++pl) a[cur][t][pl] = p[(((D - 1) * TOUT + t * TSTRIDE) * kPlanes + pl) * 64];
        }
    }
};

// The 32 epilogue constants of one output tile (see epilogue_tile): tile t2, half h -> [unscale 16 | bias 16].
struct EpiConsts {
    f32x4 un[4], bi[4];
    __device__ inline void start(gfloat* __restrict__ consts, int t2, int half)
    {
        gf32x4* c = (gf32x4*)(consts + (t2 * 2 + half) * 32);
#pragma unroll
        for (int q = 0; q < 4; ++q) { un[q] = c[q]; bi[q] = c[4 + q]; }
    }
};



// layer epilogue of one output tile: back to the unscaled domain, bias, optional ReLU.
// consts = the layer's epilogue block (behind its fragments): tile t2, half h -> [unscale 16 | bias 16] in register order.
// The 32 constants are REQUESTED before the layer's K loop (EpiConsts::start) and only consumed here: read at their use
// site they put an L2 round trip on the tile's dependency chain after every layer.
template <bool RELU>
__device__ inline void epilogue_tile(f32x16& h, const EpiConsts& k, float row_un)
{
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float y = __builtin_fmaf(h[4 * q + e], k.un[q][e] * row_un, k.bi[q][e]);
            h[4 * q + e] = RELU ? fmaxf(y, 0.0f) : y;
        }
    }
}
template <bool RELU>
__device__ inline void epilogue_tile(f32x16& h, gfloat* __restrict__ consts, int t2, int half, float row_un)
{
    EpiConsts k;
    k.start(consts, t2, half);
    epilogue_tile<RELU>(h, k, row_un);
}

__device__ inline float reg_max(const f32x16& h)
{
    float m = fabsf(h[0]);
#pragma unroll
    for (int r = 1; r < 16; ++r) m = fmaxf(m, fabsf(h[r]));
    return m;
}

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
    int lds_actions_off;   // optional (multi-tick kernel): byte offset of the world's action array in the DYNAMIC LDS region
    int lds_slot;          // (-1 = none), and the slot of tile row (lane & 31) in it
    int x_lds_off;         // one-wave tile in the multi-tick kernel: byte offset of this lane's row in the LDS mirror, or -1 (read obs)
    int c_lds_off;         // policy_tile1s: byte offset of the brain's epilogue constants in LDS ([l1 | l2a | l2b][256] floats)
#ifdef RL_PHASE_PROFILE
    long long* prof;       // tuning build: shader-clock stamps (slots 48..), non-null in the profiled workgroup only
#endif
};

#ifdef RL_PHASE_PROFILE
#define RL_PMARK(i) do { if (io.prof && threadIdx.x == 0) io.prof[100 + (i)] = (long long)clock64(); } while (0)
#else
#define RL_PMARK(i) do { } while (0)
#endif

// ---------------------------------------------------------------------------------------------------------------
// One WAVE per 32-row tile (the dueling kinds: D3QN / PERD3QN).
//
// The 4-wave tile of rounds 1-2 (removed in round 5) spent ~25 instructions per MFMA: every wave stages, scales and splits rows, publishes and re-reads
// activations through LDS, crosses five workgroup barriers and keeps a weight ring for its quarter of the features -- 9,400
// instructions per tile, and the launch is bound by instruction issue (SQ counters: profiles/r02a_policy_tick_sq_counters.txt).
// Here ONE wave owns all four output tiles of every layer, so that
//   * the accumulator layout IS the next layer's B operand for the wave itself: no LDS, no barrier, no exchange at all;
//   * a weight fragment chunk (8 KB, contiguous in the packed layout) is requested once per tile instead of once per wave;
//   * the observation rows are read straight into B-operand order (lane = row, half = k-group: 2 x 16 B per 16-k chunk);
//   * four independent accumulators take the three partial products in turn (an accumulator is reused every 4th MFMA, beyond
//     the 16-pass latency), so no separate cross-term accumulator is needed.
// ~3,000 instructions per tile, 360 of them MFMAs; needs ~230 VGPRs (one or two waves per SIMD), which is what a launch of a
// few tiles per SIMD wants anyway.  Summation order: per 16-k chunk hi.lo, hi.hi, lo.hi into ONE f32 accumulator.
// ---------------------------------------------------------------------------------------------------------------
template <int NS, int D>
__device__ inline void k_loop_reg4(WRing<4, 4, 1, D>& w, const f32x4 (&B)[NS][kPlanes], f32x16 (&acc)[4])
{
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        f32x4 ac[4][kPlanes];
        w.template next<NS>(s, ac);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = mfma16(ac[t][0], B[s][1], acc[t]);  // hi.lo
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = mfma16(ac[t][0], B[s][0], acc[t]);  // hi.hi
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = mfma16(ac[t][1], B[s][0], acc[t]);  // lo.hi
        __builtin_amdgcn_sched_barrier(0);
    }
}

// head (8 / 1 outputs padded to one 32-row A tile): K = 128 over the wave's own activation fragments; three accumulator
// chains (one per partial product) keep consecutive MFMAs independent.  out[r] = output 4 * (lane >> 5) + r of row lane & 31.
template <int D>
__device__ inline void head_reg(gfloat* __restrict__ hw, const f32x4 (&B)[8][kPlanes], float row_un, int lane, float (&out)[4])
{
    WRing<1, 1, 1, D> w;
    w.start(hw, lane, 0);
    const f32x4 un4 = ((gf32x4*)(hw + head_consts_off(4)))[lane >> 5];
    f32x16 a0, a1, a2;
#pragma unroll
    for (int r = 0; r < 16; ++r) { a0[r] = 0.0f; a1[r] = 0.0f; a2[r] = 0.0f; }
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        f32x4 ac[1][kPlanes];
        w.template next<8>(s, ac);
        a0 = mfma16(ac[0][0], B[s][1], a0);
        a1 = mfma16(ac[0][0], B[s][0], a1);
        a2 = mfma16(ac[0][1], B[s][0], a2);
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) out[r] = ((a0[r] + a1[r]) + a2[r]) * (un4[r] * row_un);
}

// epilogue of the four output tiles of a hidden layer (unscale, bias, ReLU), row maximum over all 128 features of the lane's
// row, and the split of the result into the next layer's eight B fragments (chunk 2t + c = registers 8c .. 8c+7 of tile t)
__device__ inline void layer_out_to_B(f32x16 (&acc)[4], gfloat* __restrict__ consts, int half, float row_un_in, f32x4 (&B)[8][kPlanes], float& row_un_out)
{
    float m = 0.0f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        EpiConsts k;
        k.start(consts, t, half);
        epilogue_tile<true>(acc[t], k, row_un_in);
        m = fmaxf(m, reg_max(acc[t]));
    }
    m = fmaxf(m, __shfl_xor(m, 32));   // both k-halves of a row (lanes j and j + 32) use one factor
    float sc;
    row_scale(m, sc, row_un_out);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            float x[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = acc[t][8 * c + e];
            split8(x, sc, B[2 * t + c][0], B[2 * t + c][1]);
        }
}

#ifdef RL_PHASE_PROFILE
#define RL_PMARK1(i) do { if (io.prof && lane == 0) io.prof[100 + (i)] = (long long)clock64(); } while (0)
#else
#define RL_PMARK1(i) do { } while (0)
#endif
// dueling combine (per-row mean: PERD3QN.py:202 at batch 1), outputs, action.  adv: the lane's four advantages (outputs 4h .. 4h+3
// of row j, bias not added); val_plus_bias: the row's value (used by the lanes of half 0).
template <int KIND>
__device__ inline void tile1_finish(const TileIO& io, int lane, const float (&adv)[4], float val_plus_bias, const rl_u4& draw, const f32x4& ba)
{
    const int h = lane >> 5;
    float a4[4] = {adv[0] + ba.x, adv[1] + ba.y, adv[2] + ba.z, adv[3] + ba.w};
    float o4[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) o4[r] = __shfl_xor(a4[r], 32);   // the other half's four advantages
    if (h == 0) {
        const float advs[8] = {a4[0], a4[1], a4[2], a4[3], o4[0], o4[1], o4[2], o4[3]};
        float mean = 0.0f;
#pragma unroll
        for (int i = 0; i < 8; ++i) mean += advs[i];
        mean *= 0.125f;
        const float v = val_plus_bias;
        float q[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) q[i] = advs[i] + v - mean;
        if (io.valid) {
            if (io.out) {
                f32x4* o = (f32x4*)(io.out + io.row * 8);
                o[0] = f32x4{q[0], q[1], q[2], q[3]};
                o[1] = f32x4{q[4], q[5], q[6], q[7]};
            }
            if (io.actions) {
                const float u = (float)rl_u24(draw.x);
                int a = 0;
                if (u < io.eps) a = (int)(draw.y >> 29);
                else {
                    float best = q[0];
#pragma unroll
                    for (int i = 1; i < 8; ++i) { const bool gt = q[i] > best; a = gt ? i : a; best = gt ? q[i] : best; }  // first maximum
                }
                io.actions[io.row] = (int8_t)a;
                if (io.lds_actions_off >= 0) {
                    extern __shared__ __attribute__((aligned(16))) char rl_dyn_lds[];
                    ((signed char*)rl_dyn_lds)[io.lds_actions_off + io.lds_slot] = (signed char)a;
                }
            }
        }
    }
}

template <int KIND, bool COHERENT, bool XLDS = false>
__device__ inline void policy_tile1(const TileIO& io, int lane)
{
    static_assert(KIND == RL_D3QN || KIND == RL_PERD3QN, "one-wave tile: dueling kinds");
    const int h = lane >> 5;
    const Layout L = layout_of(KIND);
    gfloat* __restrict__ packed = io.packed;
    WRing<4, 4, 1, 2> w;
    RL_PMARK1(1);
    w.start(packed + L.l1, lane, 0);
    // ---- the lane's half of its observation row, chunk by chunk: x[row][16c + 8h + 0..7]
    f32x4 B1[kInChunks][kPlanes];
    {
        // chunk 9 of the upper half (k = 152 .. 159) holds one real input: it reads k = 149 .. 152 instead and keeps the last
        // element, so that no load leaves the row
        const int64_t rbase = io.row * RL_OBS_DIM;
        if (XLDS && io.x_lds_off >= 0) {   // the row is mirrored in LDS (same float32 values): 20 x ds_read_b128, no trip through L2
            extern __shared__ __attribute__((aligned(16))) char rl_dyn_lds[];
            const float* xr = (const float*)(rl_dyn_lds + io.x_lds_off);
#pragma unroll
            for (int c = 0; c < kInChunks; ++c)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    if (c == kInChunks - 1 && h == 1) B1[c][q] = f32x4{xr[149], xr[150], xr[151], xr[152]};   // (not 16-byte aligned)
                    else B1[c][q] = *(const f32x4*)(xr + 16 * c + 8 * h + 4 * q);
                }
        } else
#pragma unroll
        for (int c = 0; c < kInChunks; ++c)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int k0 = (c == kInChunks - 1 && h == 1) ? 149 : 16 * c + 8 * h + 4 * q;
                if (COHERENT) {
                    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)io.obs, 0, 0x7fffffff, 0x00027000);
                    B1[c][q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)((rbase + k0) * 4), 0, 16 /* sc1 */));
                } else
                    B1[c][q] = *(const f32x4u*)(io.obs + rbase + k0);
            }
    }
    rl_u4 draw = {0u, 0u, 0u, 0u};
    // (a greedy brain never looks at its draw: u < 0 is false whatever u is -- and in this one-wave tile the ~100 instructions of the
    // Philox block sit on the tile's only dependency chain)
    if (io.actions && io.eps > 0.0f) draw = rl_philox4x32(io.seed, io.key_epoch, io.key_world, io.key_tick, RL_SITE_ACT, io.key_index);
    if (h == 1) { B1[kInChunks - 1][0] = f32x4{B1[kInChunks - 1][0].w, 0.0f, 0.0f, 0.0f}; B1[kInChunks - 1][1] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; }
    float m = 0.0f;
#pragma unroll
    for (int c = 0; c < kInChunks; ++c)
#pragma unroll
        for (int q = 0; q < 2; ++q)
            m = fmaxf(m, fmaxf(fmaxf(fabsf(B1[c][q].x), fabsf(B1[c][q].y)), fmaxf(fabsf(B1[c][q].z), fabsf(B1[c][q].w))));
    m = fmaxf(m, __shfl_xor(m, 32));
    float sc0, un0;
    row_scale(m, sc0, un0);
#pragma unroll
    for (int c = 0; c < kInChunks; ++c) {
        const float x[8] = {B1[c][0].x, B1[c][0].y, B1[c][0].z, B1[c][0].w, B1[c][1].x, B1[c][1].y, B1[c][1].z, B1[c][1].w};
        split8(x, sc0, B1[c][0], B1[c][1]);
    }
    // ---- input layer
    f32x16 acc[4];
    RL_PMARK1(2);
    k_loop_reg4<kInChunks>(w, B1, acc);
    RL_PMARK1(3);
    w.start(packed + L.l2a, lane, 0);
    f32x4 B2[8][kPlanes], B3[8][kPlanes];
    float un1, un2;
    layer_out_to_B(acc, packed + L.l1 + frag_floats(kInChunks, 4), h, un0, B2, un1);   // relu(feature) feeds both branches (PERD3QN.py:200-201)
    // ---- advantage branch
    RL_PMARK1(4);
    k_loop_reg4<8>(w, B2, acc);
    RL_PMARK1(5);
    layer_out_to_B(acc, packed + L.l2a + frag_floats(8, 4), h, un1, B3, un2);
    float adv[4], val[4];
    RL_PMARK1(6);
    head_reg<4>(packed + L.ha, B3, un2, lane, adv);
    RL_PMARK1(7);
    // ---- value branch
    w.start(packed + L.l2b, lane, 0);
    k_loop_reg4<8>(w, B2, acc);
    RL_PMARK1(8);
    layer_out_to_B(acc, packed + L.l2b + frag_floats(8, 4), h, un1, B3, un2);
    head_reg<4>(packed + L.hb, B3, un2, lane, val);
    RL_PMARK1(9);
    const f32x4 ba = ((gf32x4*)(packed + L.ha + head_consts_off(4) + 8))[h];
    tile1_finish<KIND>(io, lane, adv, val[0] + packed[L.hb + head_consts_off(4) + 8], draw, ba);
}

// ---------------------------------------------------------------------------------------------------------------
// Dense launches (thousands of tiles): kDenseTiles one-wave tiles of one brain per workgroup, the weights through LDS.
//
// What bounds the stand-alone policy launches in the throughput regime is the weight stream into the CUs: every
// 32-row tile pulls the brain's 263 KB through its CU's vector L1.  In k_policy_dense a workgroup of 4 waves = 4 tiles of ONE brain
// fetches every weight stage ONCE (256 threads x 2 x 16 bytes = 8 KB: two K-chunks of a tile pair, or four K-chunks of a head), parks
// it in LDS (three stage buffers, one workgroup barrier per stage) and all four waves take their MFMA A operands from there: a
// quarter of the bytes per row through L2 -> L1; two such workgroups per CU run out of step.  The tile itself (policy_tile1ds, at the
// end of this file) is the hand-scheduled one-wave tile on those stages.
// ---------------------------------------------------------------------------------------------------------------
constexpr int kStageUnits = 512;     // 16-byte units per stage (8 KB)
constexpr int kStages = 10 + 8 + 2 + 8 + 2;   // input layer, hidden advantage, head advantage, hidden value, head value
constexpr int kDenseTiles = 4;       // tiles (waves) per workgroup

// ---------------------------------------------------------------------------------------------------------------
// The one-wave tile, SCHEDULED BY HAND (the multi-tick kernel's policy half is ONE tile's dependency chain).
//
// Measured on gfx950 (tools/ubench/valu_rate.hip, mfma_valu_overlap.hip): a wave that is alone on its SIMD issues one VALU
// instruction per ~8 cycles (11.5 for v_fma_mix*), but up to four of them ride for free in the 34-cycle shadow of each of its
// OWN v_mfma_f32_32x32x16_f16.  policy_tile1 runs its 360 MFMAs and its ~2,100 VALU instructions (epilogues, row maxima, f16
// splits) in separate phases, so it pays their SUM (~12k + ~17k cycles).  Here every layer is taken as TWO passes over K (output
// tiles 0,1 then 2,3: two accumulators per pass, an accumulator is reused every other MFMA), and each MFMA is followed by a
// fixed slice of independent VALU work (a "slot"; a sched_barrier after every slot pins the order):
//     input layer, pass 1   the f16 split of the NEXT K-chunk of the observation row
//     pass 2 of any layer   the epilogue (unscale, bias, ReLU, row maximum) of pass 1's two tiles
//     hidden layer, pass 1  the split of the previous layer's activations, chunk by chunk, just ahead of their use
//     heads                 likewise the split of their input
// What stays exposed per layer is the epilogue of tiles 2,3, the row maximum / scale, and the first chunk's split.
// The arithmetic per accumulator (order of the partial products and chunks, epilogue, split) is that of policy_tile1, so the
// results are identical bit for bit.  The epilogue constants come from a copy in LDS (`c_lds_off`: [3 layers][256] floats of
// the brain, filled once per launch): their loads sit two slots ahead of their use, which an L2 round trip does not allow.
// ---------------------------------------------------------------------------------------------------------------
// LDS block of one brain for policy_tile1s: [l1 | l2a | l2b][256] epilogue constants, then the heads' [unscale 8 | bias 8] (advantage, value)
constexpr int kTileConstFloats = 3 * 256 + 2 * 16;

// Weight ring over the 2 * NS steps of a layer taken as two passes: step i = K-chunk i % NS of output tiles 2 * (i / NS), +1.
template <int NS, int D, int STEPS = 2 * NS, int TOUT = 4>   // STEPS = NS: one pass only (the caller offsets the base by its tile pair); TOUT: output tiles of the layer
struct WRingH {
    f32x4 a[D][2][kPlanes];
    gf32x4* p;   // tile-pair fragment block of the step the next refill asks for
#ifdef RL_ABL_WH  // tuning experiment (RL_EXTRA_HIPCC_FLAGS=-DRL_ABL_WH): every step re-reads the layer's first fragments (L1 hits; results WRONG)
    static __host__ __device__ constexpr int off(int i) { return 0; }
#else
    static __host__ __device__ constexpr int off(int i) { return ((i % NS) * TOUT + 2 * (i / NS)) * kPlanes * 64; }   // in 16-byte units
#endif
    __device__ inline void start(gfloat* __restrict__ pw, int lane)
    {
        static_assert(D < NS, "ring deeper than a pass");
        p = (gf32x4*)pw + lane;
#pragma unroll
        for (int d = 0; d < D; ++d)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int pl = 0; pl < kPlanes; ++pl) a[d][t][pl] = p[off(d) + (t * kPlanes + pl) * 64];
        p += off(D);
        asm volatile("" : "+v"(p));
    }
    __device__ inline void take(int i, f32x4 (&ac)[2][kPlanes])
    {
        const int cur = i % D;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int pl = 0; pl < kPlanes; ++pl) ac[t][pl] = a[cur][t][pl];
        if (i + D < STEPS) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int pl = 0; pl < kPlanes; ++pl) a[cur][t][pl] = p[(t * kPlanes + pl) * 64];
            if (i + D + 1 < STEPS) { p += off(i + D + 1) - off(i + D); asm volatile("" : "+v"(p)); }
        }
    }
};

__device__ inline void split_hi(float x0, float x1, float sc, float& hi)
{
    unsigned h = 0;
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h) : "v"(x0), "v"(sc));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h) : "v"(x1), "v"(sc));
#endif
    hi = __builtin_bit_cast(float, h);
}
__device__ inline void split_lo(float x0, float x1, float sc, float hi, float& lo)
{
    unsigned l = 0;
    [[maybe_unused]] const unsigned h = __builtin_bit_cast(unsigned, hi);   // (the host pass has no use for it)
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(l) : "v"(x0), "v"(sc), "v"(h));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(x1), "v"(sc), "v"(h));
#endif
    lo = __builtin_bit_cast(float, l);
}
// Slice k of NSLOT of the split of one 8-element B chunk (the same instructions as split8, in the same order): the eight
// half-pairs hi0 lo0 hi1 lo1 hi2 lo2 hi3 lo3 are dealt to the slots in order.  raw(e) = element e of the chunk.
template <int NSLOT, typename Raw>
__device__ inline void split_slice(int k, Raw&& raw, float sc, f32x4& hi, f32x4& lo)
{
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        if (j < (k * 8) / NSLOT || j >= ((k + 1) * 8) / NSLOT) continue;
        const int q = j >> 1;
        if ((j & 1) == 0) { float v; split_hi(raw(2 * q), raw(2 * q + 1), sc, v); hi[q] = v; }
        else { float v; split_lo(raw(2 * q), raw(2 * q + 1), sc, hi[q], v); lo[q] = v; }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// What an observation row IS (environment.py:362-370, 392-456), used by the input layer when the rows are known to be the world
// kernels' own (template argument XM of policy_tile1s / policy_pair2, chosen per tile by k_run from the state it holds -- run_obs_flags,
// rl_run.hip; the stand-alone launches take XM = 0, the general code):
//   * the three 7x7 planes hold only -1, 0, 1/2, 1 -- except the health plane when cell (0,0) holds an agent (np.vectorize then infers
//     float64 and the plane carries health / 200, environment.py:392-402; otherwise it is truncated to int64: -1, 0, 1);
//   * features 147..152 are health / 200 (|.| < 2 while |health| < 400), 0/1, same / n <= 1, n / max_agents, 0/1 and ate_super_food = +-1.
// Hence  RL_XF_SCALE: |health| < 400 for every agent and n < 2 max_agents  =>  every row's largest magnitude lies in [1, 2): the row
//                     scale is 2^kScaleExp without looking at the row (no row-maximum pass);
//        the `lo` half of the split of a K-chunk that only holds plane values is EXACTLY zero (a multiple of 1/2 times 2^10 is an f16):
//        chunks 0-2 (features 0..47) and 7-8 (112..143) whenever RL_XF_SCALE holds, chunks 3-6 (48..111: the health plane) also when
//        RL_XF_INT_HEALTH holds.  Their W_hi . x_lo products are exact zeros: the MFMAs are not issued and the halves not computed --
//        the accumulators receive the same values in the same order, so the results are BIT-IDENTICAL to the general path (tests: every
//        comparison of k_run with the stand-alone kernels, which take the general path).
// XM is a COMPILE-TIME property of a tile instantiation (0 nothing known, 1 RL_XF_SCALE, 2 + RL_XF_INT_HEALTH) and the caller branches
// between whole tiles: the same decisions as wave-uniform branches INSIDE one copy of the layer made the register allocator spill (65 -
// 289 VGPRs: a conditional MFMA leaves two live versions of its sixteen-register accumulator) and the kernel 1.5 - 4 us slower.
// ---------------------------------------------------------------------------------------------------------------
enum { RL_XF_SCALE = 1, RL_XF_INT_HEALTH = 2 };
template <int V> struct IntC { static constexpr int value = V; };
// 0: the chunk's lo half may be anything (chunk 9: the six scalar features); 1: zero under RL_XF_SCALE; 2: zero under RL_XF_SCALE + RL_XF_INT_HEALTH
__host__ __device__ constexpr int in_chunk_class(int c) { return (c <= 2 || c == 7 || c == 8) ? 1 : (c >= 3 && c <= 6) ? 2 : 0; }
// One pass of the INPUT layer for output tiles 2 * HALF, + 1 of the ring, steps S .. 9: per chunk the two hi.lo MFMAs (not issued when the chunk's lo
// half is known to be zero: XM >= the chunk's class), then hi.hi, hi.hi, lo.hi, lo.hi with shadow(step, k, slot) behind the
// k-th of those four (slot = 4 * step + k).  The accumulation order per accumulator is k_pass's.
template <int HALF, int XM, int S, typename Ring, typename Shadow>
__device__ __forceinline__ void k_pass_inx_steps(Ring& w, const f32x4 (&B)[kInChunks][kPlanes], f32x16& a0, f32x16& a1, Shadow& shadow)
{
    constexpr int cls = in_chunk_class(S);
    f32x4 ac[2][kPlanes];
    w.take(HALF * kInChunks + S, ac);
    if constexpr (!(cls != 0 && XM >= cls)) {
        a0 = mfma16(ac[0][0], B[S][1], a0); __builtin_amdgcn_sched_barrier(0);   // hi.lo
        a1 = mfma16(ac[1][0], B[S][1], a1); __builtin_amdgcn_sched_barrier(0);
    }
    a0 = mfma16(ac[0][0], B[S][0], a0); shadow(IntC<S>{}, IntC<0>{}, IntC<4 * S + 0>{}); __builtin_amdgcn_sched_barrier(0);   // hi.hi
    a1 = mfma16(ac[1][0], B[S][0], a1); shadow(IntC<S>{}, IntC<1>{}, IntC<4 * S + 1>{}); __builtin_amdgcn_sched_barrier(0);
    a0 = mfma16(ac[0][1], B[S][0], a0); shadow(IntC<S>{}, IntC<2>{}, IntC<4 * S + 2>{}); __builtin_amdgcn_sched_barrier(0);   // lo.hi
    a1 = mfma16(ac[1][1], B[S][0], a1); shadow(IntC<S>{}, IntC<3>{}, IntC<4 * S + 3>{}); __builtin_amdgcn_sched_barrier(0);
    if constexpr (S + 1 < kInChunks) k_pass_inx_steps<HALF, XM, S + 1>(w, B, a0, a1, shadow);
}
template <int HALF, int XM, typename Ring, typename Shadow>
__device__ __forceinline__ void k_pass_inx(Ring& w, const f32x4 (&B)[kInChunks][kPlanes], f32x16& a0, f32x16& a1, Shadow&& shadow)
{
#pragma unroll
    for (int r = 0; r < 16; ++r) { a0[r] = 0.0f; a1[r] = 0.0f; }
    k_pass_inx_steps<HALF, XM, 0>(w, B, a0, a1, shadow);
}
// The split of input chunk C behind the four always-present MFMAs of the step before it: slot k does hi pair k; the lo pairs (0, 1 behind
// slot 1; 2, 3 behind slot 3) only when the chunk's lo half is not known to be zero.  The instructions are split8's.
template <int XM, int C, int K, typename Raw>
__device__ __forceinline__ void in_split_slot(Raw&& raw, float sc, f32x4& hi, f32x4& lo)
{
    constexpr int cls = in_chunk_class(C);
    { float v; split_hi(raw(2 * K), raw(2 * K + 1), sc, v); hi[K] = v; }
    if constexpr (K == 1 || K == 3) {
        if constexpr (!(cls != 0 && XM >= cls)) {
            float v;
            split_lo(raw(2 * K - 2), raw(2 * K - 1), sc, hi[K - 1], v); lo[K - 1] = v;
            split_lo(raw(2 * K), raw(2 * K + 1), sc, hi[K], v); lo[K] = v;
        }
    }
}

// One pass of a layer: K loop over NS chunks for output tiles 2 * HALF, +1.  shadow(slot), slot = 0 .. 6 * NS - 1, follows MFMA `slot`.
template <int NS, int HALF, typename Ring, typename Shadow>   // Ring: take(step, fragments) -- WRingH (registers) or WStageH (LDS stages)
__device__ inline void k_pass(Ring& w, const f32x4 (&B)[NS][kPlanes], f32x16& a0, f32x16& a1, Shadow&& shadow)
{
#pragma unroll
    for (int r = 0; r < 16; ++r) { a0[r] = 0.0f; a1[r] = 0.0f; }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        f32x4 ac[2][kPlanes];
        w.take(HALF * NS + s, ac);
        a0 = mfma16(ac[0][0], B[s][1], a0); shadow(6 * s + 0); __builtin_amdgcn_sched_barrier(0);   // hi.lo
        a1 = mfma16(ac[1][0], B[s][1], a1); shadow(6 * s + 1); __builtin_amdgcn_sched_barrier(0);
        a0 = mfma16(ac[0][0], B[s][0], a0); shadow(6 * s + 2); __builtin_amdgcn_sched_barrier(0);   // hi.hi
        a1 = mfma16(ac[1][0], B[s][0], a1); shadow(6 * s + 3); __builtin_amdgcn_sched_barrier(0);
        a0 = mfma16(ac[0][1], B[s][0], a0); shadow(6 * s + 4); __builtin_amdgcn_sched_barrier(0);   // lo.hi
        a1 = mfma16(ac[1][1], B[s][0], a1); shadow(6 * s + 5); __builtin_amdgcn_sched_barrier(0);
    }
}

__device__ inline void max3_abs(float& m, float a, float b)   // m = max(m, |a|, |b|) in one instruction
{
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(m) : "v"(a), "v"(b));
#endif
}

// Epilogue of two finished output tiles, one accumulator register per step (32 steps): y = relu(acc * (unscale * row_un) + bias),
// m = max(m, y).  The constants of quarter Q (4 registers) are read from LDS while quarter Q - 1 is worked on.
struct EpiStream {
    const float* c;   // LDS: the layer's 256 constants + 32 * half  ([tile][half][unscale 16 | bias 16])
    f32x4 un[2], bi[2];
    __device__ inline void fetch(int t, int Q)   // quarter Q = 0 .. 7 of the tile pair (t = first tile of the pair)
    {
        const float* q = c + (t + (Q >> 2)) * 64 + 4 * (Q & 3);
        un[Q & 1] = *(const f32x4*)q;
        bi[Q & 1] = *(const f32x4*)(q + 16);
    }
    __device__ inline void step(int t, int e, f32x16& a0, f32x16& a1, float row_un, float& m)   // e = 0 .. 31
    {
        const int Q = e >> 2, r = e & 15, ee = e & 3;
        if (ee == 0 && Q + 1 < 8) fetch(t, Q + 1);
        f32x16& a = e < 16 ? a0 : a1;
        const float y = fmaxf(__builtin_fmaf(a[r], un[Q & 1][ee] * row_un, bi[Q & 1][ee]), 0.0f);
        a[r] = y;
        m = fmaxf(m, y);
    }
};

// head (8 / 1 outputs padded to one A tile) over K = 128, its input split chunk by chunk ahead of the MFMAs that use it.
// raw(c, e): element e of B chunk c (= register 8 * (c & 1) + e of activation tile c >> 1).
template <int D, int NCH = 8, typename Raw>   // NCH: K-chunks of the head's input this wave holds (8 = 128 features)
__device__ inline void head_stream(WRing<1, 1, 1, D>& w, const f32x4& un4, Raw&& raw, float sc, float row_un, float (&out)[4])
{
    f32x16 a0, a1, a2;
#pragma unroll
    for (int r = 0; r < 16; ++r) { a0[r] = 0.0f; a1[r] = 0.0f; a2[r] = 0.0f; }
    f32x4 B[NCH][kPlanes];
#pragma unroll
    for (int k = 0; k < 3; ++k) split_slice<3>(k, [&](int e) { return raw(0, e); }, sc, B[0][0], B[0][1]);
#pragma unroll
    for (int s = 0; s < NCH; ++s) {
        f32x4 ac[1][kPlanes];
        w.template next<NCH>(s, ac);
        a0 = mfma16(ac[0][0], B[s][1], a0);
        if (s + 1 < NCH) split_slice<3>(0, [&](int e) { return raw(s + 1, e); }, sc, B[(s + 1) % NCH][0], B[(s + 1) % NCH][1]);
        __builtin_amdgcn_sched_barrier(0);
        a1 = mfma16(ac[0][0], B[s][0], a1);
        if (s + 1 < NCH) split_slice<3>(1, [&](int e) { return raw(s + 1, e); }, sc, B[(s + 1) % NCH][0], B[(s + 1) % NCH][1]);
        __builtin_amdgcn_sched_barrier(0);
        a2 = mfma16(ac[0][1], B[s][0], a2);
        if (s + 1 < NCH) split_slice<3>(2, [&](int e) { return raw(s + 1, e); }, sc, B[(s + 1) % NCH][0], B[(s + 1) % NCH][1]);
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) out[r] = ((a0[r] + a1[r]) + a2[r]) * (un4[r] * row_un);
}

// PAIR: two waves per tile on ONE SIMD (waves i and i + 4 of a workgroup share one: tools/ubench/simd_map.hip), which hide each
// other's waits.  Input layer: role r computes output tiles 2r, 2r+1 (its half of the 128 features); the two exchange their
// partial row maxima (pair_lds->pmax) and then their halves of the split activations (pair_lds->ex, which may alias the LDS
// mirror of the Agent.state rows: every tile wave has read its row before the first of the two workgroup barriers in here).
// Then role 0 = advantage branch (+ the finish, after the caller's barrier), role 1 = value branch, whose result reaches role 0
// through pair_lds->val.  Waves without a tile must meet the same two barriers.
struct PairLds {
    float* pmax;   // [2 roles][64]
    f32x4* ex;     // [8 chunks][2 planes][64]
    float* val;    // [32]
};
struct Tile1Part {
    float head[4];
    rl_u4 draw;
};
template <int KIND, bool COHERENT, bool PAIR = false, int XM = 0>   // XM: what is known about the rows (in_chunk_class): 0 nothing, 1 RL_XF_SCALE, 2 + RL_XF_INT_HEALTH
__device__ inline void policy_tile1s(const TileIO& io, int lane, int role = 0, const PairLds* pair_lds = nullptr, Tile1Part* part = nullptr)
{
    static_assert(KIND == RL_D3QN || KIND == RL_PERD3QN, "one-wave tile: dueling kinds");
    extern __shared__ __attribute__((aligned(16))) char rl_dyn_lds[];
    constexpr int D = 3;
    const int h = lane >> 5;
    const Layout L = layout_of(KIND);
    gfloat* __restrict__ packed = io.packed;
    const float* const consts = (const float*)(rl_dyn_lds + io.c_lds_off) + 32 * h;   // + 256 per layer: l1, l2a, l2b
    const float* const hconsts = (const float*)(rl_dyn_lds + io.c_lds_off) + 768;     // heads: [un 8 | bias 8] advantage, value
    RL_PMARK1(1);
    WRingH<kInChunks, D, PAIR ? kInChunks : 2 * kInChunks> w1;
    w1.start(packed + L.l1 + (PAIR ? role * (2 * kPlanes * 64 * 4) : 0), lane);
    // ---- the lane's half of its observation row, chunk by chunk: x[row][16c + 8h + 0..7] (see policy_tile1)
    f32x4 X[kInChunks][2];
    {
        const int64_t rbase = io.row * RL_OBS_DIM;
        if (io.x_lds_off >= 0) {
            const float* xr = (const float*)(rl_dyn_lds + io.x_lds_off);
#pragma unroll
            for (int c = 0; c < kInChunks; ++c)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    if (c == kInChunks - 1 && h == 1) X[c][q] = f32x4{xr[149], xr[150], xr[151], xr[152]};
                    else X[c][q] = *(const f32x4*)(xr + 16 * c + 8 * h + 4 * q);
                }
        } else
#pragma unroll
        for (int c = 0; c < kInChunks; ++c)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int k0 = (c == kInChunks - 1 && h == 1) ? 149 : 16 * c + 8 * h + 4 * q;
                if (COHERENT) {
                    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)io.obs, 0, 0x7fffffff, 0x00027000);
                    X[c][q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)((rbase + k0) * 4), 0, 16 /* sc1 */));
                } else
                    X[c][q] = *(const f32x4u*)(io.obs + rbase + k0);
            }
    }
    rl_u4 draw = {0u, 0u, 0u, 0u};
    if ((!PAIR || role == 0) && io.actions && io.eps > 0.0f) draw = rl_philox4x32(io.seed, io.key_epoch, io.key_world, io.key_tick, RL_SITE_ACT, io.key_index);
    if (h == 1) { X[kInChunks - 1][0] = f32x4{X[kInChunks - 1][0].w, 0.0f, 0.0f, 0.0f}; X[kInChunks - 1][1] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; }
    // the row's scale: known without looking when the rows are the world kernels' own (RL_XF_SCALE, see in_chunk_class) -- else from the row's
    // largest magnitude: four independent chains of max3(m, |a|, |b|), 40 instructions (the compiler's tree over fabsf / fmaxf is 92, and
    // ONE chain of 40 is slower than that tree: a dependent VALU instruction costs a lone wave more than its issue slot)
    float sc0, un0;
    if constexpr (XM >= 1) row_scale(1.0f, sc0, un0);
    else {
        float m4[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int c = 0; c < kInChunks; ++c)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                max3_abs(m4[(2 * c + q) & 3], X[c][q].x, X[c][q].y);
                max3_abs(m4[(2 * c + q + 2) & 3], X[c][q].z, X[c][q].w);
            }
        float m = fmaxf(fmaxf(m4[0], m4[1]), fmaxf(m4[2], m4[3]));
        m = fmaxf(m, __shfl_xor(m, 32));
        row_scale(m, sc0, un0);
    }
    f32x4 B1[kInChunks][kPlanes];
    auto xraw = [&](int c, int e) { return X[c][e >> 2][e & 3]; };
    // ---- input layer
    f32x16 F[4];
    EpiStream ep;
    if constexpr (XM >= 1) {
        auto x0 = [&](int e) { return xraw(0, e); };
        in_split_slot<XM, 0, 0>(x0, sc0, B1[0][0], B1[0][1]); in_split_slot<XM, 0, 1>(x0, sc0, B1[0][0], B1[0][1]);
        in_split_slot<XM, 0, 2>(x0, sc0, B1[0][0], B1[0][1]); in_split_slot<XM, 0, 3>(x0, sc0, B1[0][0], B1[0][1]);
        RL_PMARK1(2);
        k_pass_inx<0, XM>(w1, B1, F[0], F[1], [&](auto st, auto k, auto) {
            constexpr int c = decltype(st)::value + 1;
            if constexpr (c < kInChunks) in_split_slot<XM, c, decltype(k)::value>([&](int e) { return xraw(c, e); }, sc0, B1[c][0], B1[c][1]);
        });
    } else {
#pragma unroll
        for (int k = 0; k < 6; ++k) split_slice<6>(k, [&](int e) { return xraw(0, e); }, sc0, B1[0][0], B1[0][1]);
        RL_PMARK1(2);
        k_pass<kInChunks, 0>(w1, B1, F[0], F[1], [&](int slot) {
            const int c = slot / 6 + 1;
            if (c < kInChunks) split_slice<6>(slot % 6, [&](int e) { return xraw(c, e); }, sc0, B1[c][0], B1[c][1]);
        });
    }
    float mrow = 0.0f;
    const int64_t l2 = (PAIR && role) ? L.l2b : L.l2a, hd = (PAIR && role) ? L.hb : L.ha;
    WRingH<8, D> w2;
    f32x4 B2[8][kPlanes];
    float sc1, un1;
    auto fraw = [&](int c, int e) { return F[c >> 1][8 * (c & 1) + e]; };
    if (PAIR) {
        w2.start(packed + l2, lane);
        ep.c = consts + role * 128;   // tiles 2 * role, + 1
        ep.fetch(0, 0);
#pragma unroll
        for (int e = 0; e < 32; ++e) ep.step(0, e, F[0], F[1], un0, mrow);
        mrow = fmaxf(mrow, __shfl_xor(mrow, 32));
        pair_lds->pmax[role * 64 + lane] = mrow;
        lds_barrier();
        mrow = fmaxf(mrow, pair_lds->pmax[(role ^ 1) * 64 + lane]);
        row_scale(mrow, sc1, un1);
#pragma unroll
        for (int c = 0; c < 4; ++c) {   // own chunks: registers 8 (c & 1) .. of own tile c >> 1 = chunk 4 * role + c of the layer
            f32x4 hi, lo;
#pragma unroll
            for (int k = 0; k < 6; ++k) split_slice<6>(k, [&](int e) { return fraw(c, e); }, sc1, hi, lo);
            pair_lds->ex[((4 * role + c) * kPlanes + 0) * 64 + lane] = hi;
            pair_lds->ex[((4 * role + c) * kPlanes + 1) * 64 + lane] = lo;
        }
        lds_barrier();
#pragma unroll
        for (int c = 0; c < 8; ++c)
#pragma unroll
            for (int pl = 0; pl < kPlanes; ++pl) B2[c][pl] = pair_lds->ex[(c * kPlanes + pl) * 64 + lane];
    } else {
        ep.c = consts;
        if constexpr (XM >= 1)
            k_pass_inx<1, XM>(w1, B1, F[2], F[3], [&](auto, auto, auto sl) {
                constexpr int slot = decltype(sl)::value;
                if constexpr (slot == 0) ep.fetch(0, 0);
                if constexpr (slot >= 2 && slot < 34) ep.step(0, slot - 2, F[0], F[1], un0, mrow);
            });
        else
        k_pass<kInChunks, 1>(w1, B1, F[2], F[3], [&](int slot) {
            if (slot == 0) ep.fetch(0, 0);
            if (slot >= 2 && slot < 34) ep.step(0, slot - 2, F[0], F[1], un0, mrow);
        });
        RL_PMARK1(3);
        w2.start(packed + l2, lane);
        ep.fetch(2, 0);
#pragma unroll
        for (int e = 0; e < 32; ++e) ep.step(2, e, F[2], F[3], un0, mrow);
        mrow = fmaxf(mrow, __shfl_xor(mrow, 32));
        row_scale(mrow, sc1, un1);
        // relu(feature) feeds both branches (PERD3QN.py:200-201): B2 chunk 2t + c = registers 8c .. 8c+7 of tile t
#pragma unroll
        for (int k = 0; k < 6; ++k) split_slice<6>(k, [&](int e) { return fraw(0, e); }, sc1, B2[0][0], B2[0][1]);
    }
    RL_PMARK1(4);
    // ---- advantage branch
    f32x16 A[4];
    k_pass<8, 0>(w2, B2, A[0], A[1], [&](int slot) {
        const int c = slot / 6 + 1;
        if (!PAIR && c < 8) split_slice<6>(slot % 6, [&](int e) { return fraw(c, e); }, sc1, B2[c][0], B2[c][1]);
    });
    mrow = 0.0f;
    ep.c = consts + ((PAIR && role) ? 512 : 256);
    k_pass<8, 1>(w2, B2, A[2], A[3], [&](int slot) {
        if (slot == 0) ep.fetch(0, 0);
        if (slot >= 2 && slot < 34) ep.step(0, slot - 2, A[0], A[1], un1, mrow);
    });
    RL_PMARK1(5);
    WRing<1, 1, 1, D> wh;
    wh.start(packed + hd, lane, 0);
    ep.fetch(2, 0);
#pragma unroll
    for (int e = 0; e < 32; ++e) ep.step(2, e, A[2], A[3], un1, mrow);
    mrow = fmaxf(mrow, __shfl_xor(mrow, 32));
    float sc2, un2;
    row_scale(mrow, sc2, un2);
    auto araw = [&](int c, int e) { return A[c >> 1][8 * (c & 1) + e]; };
    float adv[4], val[4];
    RL_PMARK1(6);
    if (PAIR) {
        head_stream<D>(wh, *(const f32x4*)(hconsts + ((PAIR && role) ? 16 : 0) + 4 * h), araw, sc2, un2, adv);
        if (role) { if (h == 0) pair_lds->val[lane] = adv[0] + hconsts[16 + 8]; }
        else {
#pragma unroll
            for (int r = 0; r < 4; ++r) part->head[r] = adv[r];
            part->draw = draw;
        }
        RL_PMARK1(9);
        return;
    }
    head_stream<D>(wh, *(const f32x4*)(hconsts + 4 * h), araw, sc2, un2, adv);
    WRingH<8, D> w3;   // (started before the head its twelve fragments are spilled and reloaded inside the value branch's MFMA stream)
    w3.start(packed + L.l2b, lane);
    RL_PMARK1(7);
    // ---- value branch
    k_pass<8, 0>(w3, B2, A[0], A[1], [&](int) {});
    mrow = 0.0f;
    ep.c = consts + 512;
    k_pass<8, 1>(w3, B2, A[2], A[3], [&](int slot) {
        if (slot == 0) ep.fetch(0, 0);
        if (slot >= 2 && slot < 34) ep.step(0, slot - 2, A[0], A[1], un1, mrow);
    });
    RL_PMARK1(8);
    wh.start(packed + L.hb, lane, 0);
    ep.fetch(2, 0);
#pragma unroll
    for (int e = 0; e < 32; ++e) ep.step(2, e, A[2], A[3], un1, mrow);
    mrow = fmaxf(mrow, __shfl_xor(mrow, 32));
    row_scale(mrow, sc2, un2);
    head_stream<D>(wh, *(const f32x4*)(hconsts + 16 + 4 * h), araw, sc2, un2, val);
    RL_PMARK1(9);
    tile1_finish<KIND>(io, lane, adv, val[0] + hconsts[16 + 8], draw, *(const f32x4*)(hconsts + 8 + 4 * h));
}


// ---------------------------------------------------------------------------------------------------------------
// FOUR waves per tile for the dueling kinds in 1024-thread workgroups (128 VGPRs per wave, FOUR waves per SIMD: the SIMD issues a VALU
// instruction every 2 cycles instead of every 4, tools/ubench/valu_rate.hip, and the tick half runs on sixteen waves).  The layers are
// split by OUTPUT TILE: wave q of a tile owns output tile q of the input layer and of both branches' hidden layers -- ONE accumulator
// per big layer (16 registers), the canonical chain per accumulator (chunk 0 .. NS-1, hi.lo, hi.hi, lo.hi: section 5.2.1), so the bits
// are those of policy_tile1s / policy_tile1.  The four waves of a tile sit on ONE SIMD (waves t, t + 4, t + 8, t + 12 of the
// workgroup): four independent accumulator chains keep that SIMD's matrix pipe busy, and no weight byte is fetched twice per tile.
//   rows        wave q reads K-chunks q, q + 4, q + 8 of the tile's rows from the LDS mirror of the Agent.state rows (nothing else of
//               them, ever), the partial row maxima cross through LDS (barrier 1), it splits ITS chunks into the tile's exchange slice
//               (barrier 2), and the input layer's K loop takes the B operand from there: a split costs VALU issue that only hides
//               under the issuing wave's OWN MFMAs (~2 v_fma_mix per MFMA) -- split by all four waves inside the loop, the first
//               version, the input layer took 9.6k cycles for 120 MFMAs;
//   exchanges   after each big layer every wave knows 32 of the 128 activations of a row: partial row maxima through LDS (barriers 3
//               and 5), then each wave splits ITS two K-chunks of the next layer's B operand into the slice (barriers 4 and 6); the
//               slice aliases the mirror -- every row has been read when barrier 1 is passed;
//   heads       advantage head on wave 0, value head on wave 1 (three accumulators each, as head_stream): their own two chunks stay in
//               registers, the other six come out of the slice (24 KB per tile: 3 waves x 2 chunks x 2 planes x 1 KB, twice);
//   finish      the value crosses through LDS (the caller's barrier), wave 0 combines, picks the action (tile1_finish).
// Waves without a tile meet the same six barriers.
// ---------------------------------------------------------------------------------------------------------------
struct QuadLds {
    float* pmax;   // [3 buffers][32 rows][4 waves]: stage 0 (the row) and stage 2 (advantage branch) share buffer 0 -- two barriers lie between them --, stage 1 (input layer) 1, stage 3 (value branch) 2
    f32x4* ex;     // the tile's exchange slice, kQuadExBytes
    float* val;    // [32]
#ifdef RL_PHASE_PROFILE
    long long* prof;   // tuning build: arrival stamps of tile 0's four waves (slots 116 + 4 * point + q)
#endif
};
#ifdef RL_PHASE_PROFILE
#define RL_QMARK(pt) do { if (ql->prof && lane == 0) ql->prof[116 + 4 * (pt) + q] = (long long)clock64(); } while (0)
#else
#define RL_QMARK(pt) do { } while (0)
#endif
constexpr int kQuadExBytes = 24 * 1024;
constexpr int kQuadFloats = 3 * 32 * 4 + 32;   // per tile: partial row maxima (three buffers), the row values
constexpr int kQuadBarriers = 6;

// chunk c of this lane's half row (x[row][16c + 8h + 0..7], zero beyond 152).  XL: from the LDS mirror (`xr`: the lane's row there);
// else from memory (`goff`: byte offset of the lane's row in io.obs).  The choice is made per WAVE, not per lane: memory loads inside
// the K loop share the weight ring's counter (vmcnt), and a wait for one of them is a wait for every weight load issued before it.
template <bool COHERENT, bool XL>
__device__ inline void quad_load_x(const TileIO& io, const float* xr, int goff, int c, int h, f32x4& x0, f32x4& x1)
{
    if (XL) {   // (a mirror row is zero from float 153 to 159: write_observations)
        x0 = *(const f32x4*)(xr + 16 * c + 8 * h);
        x1 = *(const f32x4*)(xr + 16 * c + 8 * h + 4);
        return;
    }
    const bool last = c == kInChunks - 1 && h == 1;   // k 152 .. 159: one value
    const int k0 = last ? 149 : 16 * c + 8 * h;
    if (COHERENT) {
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)io.obs, 0, 0x7fffffff, 0x00027000);
        x0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, goff + 4 * k0, 0, 16 /* sc1 */));
        x1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, goff + 4 * (last ? k0 : k0 + 4), 0, 16));
    } else {
        const float* g = (const float*)((const char*)io.obs + goff);
        x0 = *(const f32x4u*)(g + k0);
        x1 = *(const f32x4u*)(g + (last ? k0 : k0 + 4));
    }
}
template <bool XL>
__device__ inline void quad_fix_x(int c, int h, f32x4& x0, f32x4& x1)
{
    if (!XL && c == kInChunks - 1 && h == 1) { x0 = f32x4{x0.w, 0.0f, 0.0f, 0.0f}; x1 = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; }
}
__device__ inline float quad_row_max(const QuadLds* ql, int stage, int q, int lane, float m)   // partial maximum in -> the row's maximum out (one barrier)
{
    m = fmaxf(m, __shfl_xor(m, 32));
    float* pm = ql->pmax + (stage * 32 + (lane & 31)) * 4;
    if (lane < 32) pm[q] = m;
    lds_barrier();
    const f32x4 v = *(const f32x4*)pm;
    return fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w));
}
// One-tile weight ring of a big layer: step i = K-chunk i of output tile `tile` (TOUT tiles per chunk), fragments hi / lo.
template <int NS, int D, int TOUT = 4>
struct WRingQ {
    f32x4 a[D][kPlanes];
    gf32x4* p;
    __device__ inline void start(gfloat* __restrict__ pw, int lane, int tile)
    {
        p = (gf32x4*)pw + tile * kPlanes * 64 + lane;
#pragma unroll
        for (int d = 0; d < D; ++d)
#pragma unroll
            for (int pl = 0; pl < kPlanes; ++pl) a[d][pl] = p[(d * TOUT * kPlanes + pl) * 64];
        p += D * TOUT * kPlanes * 64;
        asm volatile("" : "+v"(p));
    }
    __device__ inline void take(int i, f32x4 (&ac)[kPlanes])
    {
        const int cur = i % D;
#pragma unroll
        for (int pl = 0; pl < kPlanes; ++pl) ac[pl] = a[cur][pl];
        if (i + D < NS) {
#ifdef RL_ABL_HALFW   /* tuning experiment: half the weight bytes (the lo plane is the hi plane; results WRONG) */
            a[cur][0] = p[0]; a[cur][1] = a[cur][0];
#else
#pragma unroll
            for (int pl = 0; pl < kPlanes; ++pl) a[cur][pl] = p[pl * 64];
#endif
            p += TOUT * kPlanes * 64;
            asm volatile("" : "+v"(p));
        }
    }
};
// epilogue of one output tile from the brain's LDS constants ([unscale 16 | bias 16] of this lane's half): y = relu(acc * (unscale * row_un) + bias)
__device__ inline float quad_epilogue(f32x16& a, const float* c, float row_un)
{
    float m = 0.0f;
#pragma unroll
    for (int Q = 0; Q < 4; ++Q) {
        const f32x4 un = *(const f32x4*)(c + 4 * Q), bi = *(const f32x4*)(c + 16 + 4 * Q);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float y = fmaxf(__builtin_fmaf(a[4 * Q + e], un[e] * row_un, bi[e]), 0.0f);
            a[4 * Q + e] = y;
            m = fmaxf(m, y);
        }
    }
    return m;
}
// A head on one wave (Q = 0: advantage, own chunks 0, 1; Q = 1: value, own chunks 2, 3): three accumulators over the eight chunks in order
template <int Q, int D>
__device__ inline void quad_head(gfloat* __restrict__ hw, int lane, const f32x4 (&own)[2][kPlanes], const f32x4* __restrict__ others, const f32x4& un4, float row_un, float (&out)[4])
{
    WRing<1, 1, 1, D> w;
    w.start(hw, lane, 0);
    f32x16 a0, a1, a2;
#pragma unroll
    for (int r = 0; r < 16; ++r) { a0[r] = 0.0f; a1[r] = 0.0f; a2[r] = 0.0f; }
    auto src = [&](int c, int pl) -> f32x4 {
        const int wv = c >> 1;
        if (wv == Q) return own[c & 1][pl];
        const int idx = Q == 0 ? wv - 1 : (wv == 0 ? 0 : wv - 1);
        return others[((idx * 2 + (c & 1)) * kPlanes + pl) * 64 + lane];
    };
    f32x4 B[2][kPlanes];
#pragma unroll
    for (int pl = 0; pl < kPlanes; ++pl) B[0][pl] = src(0, pl);
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        f32x4 ac[1][kPlanes];
        w.template next<8>(s, ac);
        if (s + 1 < 8) {
#pragma unroll
            for (int pl = 0; pl < kPlanes; ++pl) B[(s + 1) & 1][pl] = src(s + 1, pl);
        }
        a0 = mfma16(ac[0][0], B[s & 1][1], a0);
        a1 = mfma16(ac[0][0], B[s & 1][0], a1);
        a2 = mfma16(ac[0][1], B[s & 1][0], a2);
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) out[r] = ((a0[r] + a1[r]) + a2[r]) * (un4[r] * row_un);
}

template <int KIND, bool COHERENT, bool XL>
__device__ inline void quad_input_layer(const TileIO& io, int lane, int q, const QuadLds* ql, rl_u4& draw, f32x16& F, float& un0)
{
    extern __shared__ __attribute__((aligned(16))) char rl_dyn_lds[];
    const int h = lane >> 5;
    const Layout L = layout_of(KIND);
    const float* const xr = (const float*)(rl_dyn_lds + (XL ? io.x_lds_off : 0));   // (+ 8 h floats: folded into the loads' offsets)
    const int goff = XL ? 0 : (int)(io.row * (RL_OBS_DIM * 4));
    // ---- this wave's K-chunks of the row: q, q + 4, q + 8 (waves 2, 3: two chunks) -- all it ever reads of the rows
    f32x4 X[3][2];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int c = q + 4 * i;   // (uniform)
        if (c < kInChunks) { quad_load_x<COHERENT, XL>(io, xr, goff, c, h, X[i][0], X[i][1]); quad_fix_x<XL>(c, h, X[i][0], X[i][1]); }
        else { X[i][0] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; X[i][1] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; }
    }
    WRingQ<kInChunks, 4> w1;
    w1.start(io.packed + L.l1, lane, q);
    if (q == 0 && io.actions && io.eps > 0.0f) draw = rl_philox4x32(io.seed, io.key_epoch, io.key_world, io.key_tick, RL_SITE_ACT, io.key_index);
    float m4[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        max3_abs(m4[0], X[i][0].x, X[i][0].y); max3_abs(m4[1], X[i][0].z, X[i][0].w); max3_abs(m4[2], X[i][1].x, X[i][1].y); max3_abs(m4[3], X[i][1].z, X[i][1].w);
    }
    RL_PMARK1(1);
    RL_QMARK(0);
    float sc0;
    row_scale(quad_row_max(ql, 0, q, lane, fmaxf(fmaxf(m4[0], m4[1]), fmaxf(m4[2], m4[3]))), sc0, un0);   // barrier 1: every row of the tile has been read
    // ---- the wave's chunks split into the tile's exchange slice (which aliases the mirror): [chunk][plane][lane]
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int c = q + 4 * i;
        if (c < kInChunks) {
            const float x[8] = {X[i][0].x, X[i][0].y, X[i][0].z, X[i][0].w, X[i][1].x, X[i][1].y, X[i][1].z, X[i][1].w};
            f32x4 hi, lo;
            split8(x, sc0, hi, lo);
            ql->ex[(c * kPlanes + 0) * 64 + lane] = hi;
            ql->ex[(c * kPlanes + 1) * 64 + lane] = lo;
        }
    }
    lds_barrier();   // 2
    RL_PMARK1(2);
    // ---- input layer: output tile q, the B operand chunk by chunk out of the slice (one chunk ahead)
#pragma unroll
    for (int r = 0; r < 16; ++r) F[r] = 0.0f;
    f32x4 B[2][kPlanes];
#pragma unroll
    for (int pl = 0; pl < kPlanes; ++pl) B[0][pl] = ql->ex[pl * 64 + lane];
#pragma unroll
    for (int s = 0; s < kInChunks; ++s) {
        f32x4 ac[kPlanes];
        w1.take(s, ac);
        if (s + 1 < kInChunks) {
#pragma unroll
            for (int pl = 0; pl < kPlanes; ++pl) B[(s + 1) & 1][pl] = ql->ex[((s + 1) * kPlanes + pl) * 64 + lane];
        }
        F = mfma16(ac[0], B[s & 1][1], F);
        F = mfma16(ac[0], B[s & 1][0], F);
        F = mfma16(ac[1], B[s & 1][0], F);
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int KIND, bool COHERENT>
__device__ inline void policy_quad(const TileIO& io, int lane, int q, const QuadLds* ql, Tile1Part* part)
{
    static_assert(KIND == RL_D3QN || KIND == RL_PERD3QN, "four-wave tile: dueling kinds");
    extern __shared__ __attribute__((aligned(16))) char rl_dyn_lds[];
    const int h = lane >> 5;
    const Layout L = layout_of(KIND);
    gfloat* __restrict__ packed = io.packed;
    const float* const cc = (const float*)(rl_dyn_lds + io.c_lds_off);
    const float* const consts = cc + 64 * q + 32 * h;   // + 256 per layer: l1, l2a, l2b
    const float* const hconsts = cc + 768;              // heads: [un 8 | bias 8] advantage, value
    rl_u4 draw = {0u, 0u, 0u, 0u};
    // ---- the row's maximum (barrier 1) and the input layer's output tile q: rows from the mirror, or -- if ANY row of the wave is not
    // there (more rows than the mirror holds, several rounds of tiles, a launch's first tick without preload) -- all from memory, which
    // recycle_world drained in exactly those cases
    f32x16 F;
    float un0;
    if (__builtin_amdgcn_ballot_w64(io.x_lds_off < 0) == 0ull) quad_input_layer<KIND, COHERENT, true>(io, lane, q, ql, draw, F, un0);
    else quad_input_layer<KIND, COHERENT, false>(io, lane, q, ql, draw, F, un0);
    RL_PMARK1(3);
    // ---- both branches' weight rings start here: their first trip overlaps the exchange
    WRingQ<8, 3> wa, wv;
    wa.start(packed + L.l2a, lane, q);
    wv.start(packed + L.l2b, lane, q);
    float sc1, un1;
    RL_QMARK(1);
    row_scale(quad_row_max(ql, 1, q, lane, quad_epilogue(F, consts, un0)), sc1, un1);   // barrier 3: every wave is through its K loop
    RL_PMARK1(4);
    {   // this wave's two K-chunks of the hidden layers' input: registers 0 .. 7 / 8 .. 15 = chunks 2q, 2q + 1
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const float x[8] = {F[8 * c + 0], F[8 * c + 1], F[8 * c + 2], F[8 * c + 3], F[8 * c + 4], F[8 * c + 5], F[8 * c + 6], F[8 * c + 7]};
            f32x4 hi, lo;
            split8(x, sc1, hi, lo);
            ql->ex[((2 * q + c) * kPlanes + 0) * 64 + lane] = hi;
            ql->ex[((2 * q + c) * kPlanes + 1) * 64 + lane] = lo;
        }
    }
    lds_barrier();   // 4
    RL_PMARK1(5);
    // ---- hidden layers: output tile q of the advantage branch and of the value branch (relu(feature) feeds both: PERD3QN.py:200-201)
    f32x16 A, V;
#pragma unroll
    for (int r = 0; r < 16; ++r) { A[r] = 0.0f; V[r] = 0.0f; }
    f32x4 B[2][kPlanes];
#pragma unroll
    for (int pl = 0; pl < kPlanes; ++pl) B[0][pl] = ql->ex[pl * 64 + lane];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        f32x4 aa[kPlanes], av[kPlanes];
        wa.take(s, aa);
        wv.take(s, av);
        if (s + 1 < 8) {
#pragma unroll
            for (int pl = 0; pl < kPlanes; ++pl) B[(s + 1) & 1][pl] = ql->ex[(((s + 1) * kPlanes) + pl) * 64 + lane];
        }
        A = mfma16(aa[0], B[s & 1][1], A);
        V = mfma16(av[0], B[s & 1][1], V);
        A = mfma16(aa[0], B[s & 1][0], A);
        V = mfma16(av[0], B[s & 1][0], V);
        A = mfma16(aa[1], B[s & 1][0], A);
        V = mfma16(av[1], B[s & 1][0], V);
        __builtin_amdgcn_sched_barrier(0);
    }
    RL_PMARK1(6);
    RL_QMARK(2);
    const float ma = quad_epilogue(A, consts + 256, un1), mv = quad_epilogue(V, consts + 512, un1);
    {   // both branches' row maxima in one exchange (stages 2 and 3)
        const float pa = fmaxf(ma, __shfl_xor(ma, 32)), pv = fmaxf(mv, __shfl_xor(mv, 32));
        float* pm = ql->pmax + (lane & 31) * 4;
        if (lane < 32) { pm[q] = pa; pm[256 + q] = pv; }
    }
    lds_barrier();   // 5: the hidden layers' input has been read by every wave
    RL_PMARK1(7);
    float sca, una, scv, unv;
    {
        const float* pm = ql->pmax + (lane & 31) * 4;
        const f32x4 a = *(const f32x4*)pm, v = *(const f32x4*)(pm + 256);
        row_scale(fmaxf(fmaxf(a.x, a.y), fmaxf(a.z, a.w)), sca, una);
        row_scale(fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)), scv, unv);
    }
    f32x4 own[2][kPlanes];   // wave 0: its advantage chunks; wave 1: its value chunks
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const float xa[8] = {A[8 * c + 0], A[8 * c + 1], A[8 * c + 2], A[8 * c + 3], A[8 * c + 4], A[8 * c + 5], A[8 * c + 6], A[8 * c + 7]};
        const float xv[8] = {V[8 * c + 0], V[8 * c + 1], V[8 * c + 2], V[8 * c + 3], V[8 * c + 4], V[8 * c + 5], V[8 * c + 6], V[8 * c + 7]};
        f32x4 ah, al, vh, vl;
        split8(xa, sca, ah, al);
        split8(xv, scv, vh, vl);
        f32x4* const adv_o = ql->ex, * const val_o = ql->ex + 3 * 2 * kPlanes * 64;
        if (q == 0) { own[c][0] = ah; own[c][1] = al; }
        else { adv_o[(((q - 1) * 2 + c) * kPlanes + 0) * 64 + lane] = ah; adv_o[(((q - 1) * 2 + c) * kPlanes + 1) * 64 + lane] = al; }
        if (q == 1) { own[c][0] = vh; own[c][1] = vl; }
        else {
            const int iv = q == 0 ? 0 : q - 1;
            val_o[((iv * 2 + c) * kPlanes + 0) * 64 + lane] = vh; val_o[((iv * 2 + c) * kPlanes + 1) * 64 + lane] = vl;
        }
    }
    lds_barrier();   // 6
    RL_PMARK1(8);
    // ---- heads
    if (q == 0) {
        float adv[4];
        quad_head<0, 3>(packed + L.ha, lane, own, ql->ex, *(const f32x4*)(hconsts + 4 * h), una, adv);
#pragma unroll
        for (int r = 0; r < 4; ++r) part->head[r] = adv[r];
        part->draw = draw;
        RL_PMARK1(9);
    } else if (q == 1) {
        float val[4];
        quad_head<1, 3>(packed + L.hb, lane, own, ql->ex + 3 * 2 * kPlanes * 64, *(const f32x4*)(hconsts + 16 + 4 * h), unv, val);
        if (h == 0) ql->val[lane] = val[0] + hconsts[16 + 8];
    }
}


// ---------------------------------------------------------------------------------------------------------------
// Two waves per tile for the plain kinds -- DQN (153 -> 128 -> 64 -> 8) and PPO (153 -> 256 -> 256 -> 8 -> softmax) -- with the protocol of
// policy_tile1s<PAIR>: both roles meet two workgroup barriers inside (partial row maxima, then the split activations of the input layer
// through pair_lds->ex), the caller adds a third, and role 0 finishes (pair_finish).  Every layer is split by OUTPUT features:
//   input layer   role r: the upper / lower half of the output tiles (DQN: tiles 2r, 2r+1 -- exactly the dueling pair's; PPO: 4r .. 4r+3 in
//                 two passes over K, the epilogue of the first pass in the shadow of the second);
//   hidden layer  DQN: role r computes output tile r (24 MFMAs: main / cross accumulators); PPO: tiles 4r .. 4r+3 in two passes, its B
//                 operand -- all 16 K-chunks of the 256 activations -- read chunk by chunk from pair_lds->ex, one step ahead of its use
//                 (128 registers if held);
//   head          each role multiplies ITS OWN features (the head's K-chunks 2r, 2r+1 / 8r .. 8r+7) and scales them with its own row
//                 factor; role 1's four partial outputs per lane cross through pair_lds->val, role 0 adds them (pair_finish).
// So no weight byte is fetched twice per tile and the two waves (any two SIMD slots: only LDS and the barriers connect them) fill each
// other's waits.  Summation order differs from the 4-wave tile's (policy_tile): ~1e-7 apart, each checked against the oracle; the
// stand-alone k_policy_pair runs THIS code, so rl_run and the two-launch loop can be compared bit for bit.
// LDS constants of a brain (`c_lds_off`): per kind, see tile_const_floats / tile_const_src.
// ---------------------------------------------------------------------------------------------------------------
__host__ __device__ constexpr int tile_const_floats(int kind)
{
    return kind == RL_DQN ? 256 + 128 + 16 : kind == RL_PPO ? 512 + 512 + 16 : 3 * 256 + 2 * 16;
}
constexpr int kTileConstMax = 512 + 512 + 16;
// element j of a brain's LDS constant block -> offset in its packed weights
__host__ __device__ inline int64_t tile_const_src(int kind, int j)
{
    const Layout L = layout_of(kind);
    if (kind == RL_DQN) {
        if (j < 256) return L.l1 + frag_floats(kInChunks, 4) + j;
        if (j < 384) return L.l2a + frag_floats(8, 2) + (j - 256);
        return L.ha + head_consts_off(2) + (j - 384);
    }
    if (kind == RL_PPO) {
        if (j < 512) return L.l1 + frag_floats(kInChunks, 8) + j;
        if (j < 1024) return L.l2a + frag_floats(16, 8) + (j - 512);
        return L.ha + head_consts_off(8) + (j - 1024);
    }
    if (j < 256) return L.l1 + frag_floats(kInChunks, 4) + j;
    if (j < 512) return L.l2a + frag_floats(8, 4) + (j - 256);
    if (j < 768) return L.l2b + frag_floats(8, 4) + (j - 512);
    if (j < 784) return L.ha + head_consts_off(4) + (j - 768);
    return L.hb + head_consts_off(4) + (j - 784);
}
__host__ __device__ constexpr int pair_ex_bytes(int kind) { return (kind == RL_PPO ? 16 : 8) * kPlanes * 64 * 16; }
constexpr int kPairValFloats = 256;   // role 1's head partials: 4 per lane

// One pass of the PPO hidden layer: K loop over 16 chunks for output tiles (pair HALF of the role's four), the B operand from LDS
// (ex[chunk][plane][lane]), requested one step ahead.  shadow(slot) as in k_pass.
template <int HALF, typename Ring, typename Shadow>
__device__ inline void k_pass_lds16(Ring& w, const f32x4* __restrict__ ex, int lane, f32x16& a0, f32x16& a1, Shadow&& shadow)
{
    constexpr int NS = 16;
#pragma unroll
    for (int r = 0; r < 16; ++r) { a0[r] = 0.0f; a1[r] = 0.0f; }
    f32x4 B[2][kPlanes];
#pragma unroll
    for (int pl = 0; pl < kPlanes; ++pl) B[0][pl] = ex[(0 * kPlanes + pl) * 64 + lane];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        f32x4 ac[2][kPlanes];
        w.take(HALF * NS + s, ac);
        if (s + 1 < NS) {
#pragma unroll
            for (int pl = 0; pl < kPlanes; ++pl) B[(s + 1) & 1][pl] = ex[((s + 1) * kPlanes + pl) * 64 + lane];
        }
        const f32x4 (&b)[kPlanes] = B[s & 1];
        a0 = mfma16(ac[0][0], b[1], a0); shadow(6 * s + 0); __builtin_amdgcn_sched_barrier(0);   // hi.lo
        a1 = mfma16(ac[1][0], b[1], a1); shadow(6 * s + 1); __builtin_amdgcn_sched_barrier(0);
        a0 = mfma16(ac[0][0], b[0], a0); shadow(6 * s + 2); __builtin_amdgcn_sched_barrier(0);   // hi.hi
        a1 = mfma16(ac[1][0], b[0], a1); shadow(6 * s + 3); __builtin_amdgcn_sched_barrier(0);
        a0 = mfma16(ac[0][1], b[0], a0); shadow(6 * s + 4); __builtin_amdgcn_sched_barrier(0);   // lo.hi
        a1 = mfma16(ac[1][1], b[0], a1); shadow(6 * s + 5); __builtin_amdgcn_sched_barrier(0);
    }
}

template <int KIND, bool COHERENT, int XM = 0>   // XM: see policy_tile1s
__device__ inline void policy_pair2(const TileIO& io, int lane, int role, const PairLds* pair_lds, Tile1Part* part)
{
    static_assert(KIND == RL_DQN || KIND == RL_PPO, "the plain kinds (the dueling pair is policy_tile1s<PAIR>)");
    extern __shared__ __attribute__((aligned(16))) char rl_dyn_lds[];
    constexpr int D = 3;
    constexpr int T1 = KIND == RL_PPO ? 8 : 4;             // output tiles of the input layer
    constexpr int OWN1 = T1 / 2;                           // ... of which a role computes this many
    const int h = lane >> 5;
    const Layout L = layout_of(KIND);
    gfloat* __restrict__ packed = io.packed;
    const float* const cbase = (const float*)(rl_dyn_lds + io.c_lds_off);
    const float* const c1 = cbase + 32 * h;                                   // input layer: [tile][half][unscale 16 | bias 16]
    const float* const c2 = cbase + T1 * 64 + 32 * h;                         // hidden layer
    const float* const hconsts = cbase + T1 * 64 + (KIND == RL_PPO ? 512 : 128);   // head: [unscale 8 | bias 8]
    // ---- input layer: the role's OWN1 output tiles
    RL_PMARK1(1);
    WRingH<kInChunks, D, (OWN1 / 2) * kInChunks, T1> w1;
    w1.start(packed + L.l1 + role * (OWN1 * kPlanes * 64 * 4), lane);
    f32x4 X[kInChunks][2];
    {
        const int64_t rbase = io.row * RL_OBS_DIM;
        if (io.x_lds_off >= 0) {
            const float* xr = (const float*)(rl_dyn_lds + io.x_lds_off);
#pragma unroll
            for (int c = 0; c < kInChunks; ++c)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    if (c == kInChunks - 1 && h == 1) X[c][q] = f32x4{xr[149], xr[150], xr[151], xr[152]};
                    else X[c][q] = *(const f32x4*)(xr + 16 * c + 8 * h + 4 * q);
                }
        } else
#pragma unroll
        for (int c = 0; c < kInChunks; ++c)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int k0 = (c == kInChunks - 1 && h == 1) ? 149 : 16 * c + 8 * h + 4 * q;   // (see policy_tile1)
                if (COHERENT) {
                    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)io.obs, 0, 0x7fffffff, 0x00027000);
                    X[c][q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)((rbase + k0) * 4), 0, 16 /* sc1 */));
                } else
                    X[c][q] = *(const f32x4u*)(io.obs + rbase + k0);
            }
    }
    rl_u4 draw = {0u, 0u, 0u, 0u};
    if (role == 0 && io.actions && (KIND == RL_PPO || io.eps > 0.0f))
        draw = rl_philox4x32(io.seed, io.key_epoch, io.key_world, io.key_tick, RL_SITE_ACT, io.key_index);
    if (h == 1) { X[kInChunks - 1][0] = f32x4{X[kInChunks - 1][0].w, 0.0f, 0.0f, 0.0f}; X[kInChunks - 1][1] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; }
    // the row's scale and the chunks whose lo half is exactly zero: see in_chunk_class (the dueling pair: policy_tile1s)
    float sc0, un0;
    if constexpr (XM >= 1) row_scale(1.0f, sc0, un0);
    else {
        float m4[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int c = 0; c < kInChunks; ++c)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                max3_abs(m4[(2 * c + q) & 3], X[c][q].x, X[c][q].y);
                max3_abs(m4[(2 * c + q + 2) & 3], X[c][q].z, X[c][q].w);
            }
        float m = fmaxf(fmaxf(m4[0], m4[1]), fmaxf(m4[2], m4[3]));
        m = fmaxf(m, __shfl_xor(m, 32));
        row_scale(m, sc0, un0);
    }
    f32x4 B1[kInChunks][kPlanes];
    auto xraw = [&](int c, int e) { return X[c][e >> 2][e & 3]; };
    f32x16 F[OWN1];
    EpiStream ep;
    float mrow = 0.0f;
    if constexpr (XM >= 1) {
        auto x0 = [&](int e) { return xraw(0, e); };
        in_split_slot<XM, 0, 0>(x0, sc0, B1[0][0], B1[0][1]); in_split_slot<XM, 0, 1>(x0, sc0, B1[0][0], B1[0][1]);
        in_split_slot<XM, 0, 2>(x0, sc0, B1[0][0], B1[0][1]); in_split_slot<XM, 0, 3>(x0, sc0, B1[0][0], B1[0][1]);
        RL_PMARK1(2);
        k_pass_inx<0, XM>(w1, B1, F[0], F[1], [&](auto st, auto k, auto) {
            constexpr int c = decltype(st)::value + 1;
            if constexpr (c < kInChunks) in_split_slot<XM, c, decltype(k)::value>([&](int e) { return xraw(c, e); }, sc0, B1[c][0], B1[c][1]);
        });
    } else {
#pragma unroll
        for (int k = 0; k < 6; ++k) split_slice<6>(k, [&](int e) { return xraw(0, e); }, sc0, B1[0][0], B1[0][1]);
        RL_PMARK1(2);
        k_pass<kInChunks, 0>(w1, B1, F[0], F[1], [&](int slot) {
            const int c = slot / 6 + 1;
            if (c < kInChunks) split_slice<6>(slot % 6, [&](int e) { return xraw(c, e); }, sc0, B1[c][0], B1[c][1]);
        });
    }
    RL_PMARK1(3);
    ep.c = c1 + role * OWN1 * 64;
    // (the hidden layer's first weight chunks are requested before the exposed part of the epilogue and the two exchanges, not after them)
    WRingH<16, D, 32, 8> w2p;      // PPO: hidden layer, the role's four output tiles in two passes
    WRing<2, 1, 1, D> w2d;         // DQN: hidden layer, output tile `role`
    WRing<1, 1, 1, 2> whd;         // DQN: the head's two K-chunks of that tile
    if (KIND == RL_PPO) {
        if constexpr (XM >= 1)
            k_pass_inx<1, XM>(w1, B1, F[OWN1 - 2], F[OWN1 - 1], [&](auto, auto, auto sl) {
                constexpr int slot = decltype(sl)::value;
                if constexpr (slot == 0) ep.fetch(0, 0);
                if constexpr (slot >= 2 && slot < 34) ep.step(0, slot - 2, F[0], F[1], un0, mrow);
            });
        else
        k_pass<kInChunks, 1>(w1, B1, F[OWN1 - 2], F[OWN1 - 1], [&](int slot) {
            if (slot == 0) ep.fetch(0, 0);
            if (slot >= 2 && slot < 34) ep.step(0, slot - 2, F[0], F[1], un0, mrow);
        });
        RL_PMARK1(4);
        w2p.start(packed + L.l2a + role * (4 * kPlanes * 64 * 4), lane);
        ep.fetch(2, 0);
#pragma unroll
        for (int e = 0; e < 32; ++e) ep.step(2, e, F[OWN1 - 2], F[OWN1 - 1], un0, mrow);
    } else {
        w2d.start(packed + L.l2a, lane, role);
        whd.start(packed + L.ha, lane, 2 * role);
        ep.fetch(0, 0);
#pragma unroll
        for (int e = 0; e < 32; ++e) ep.step(0, e, F[0], F[1], un0, mrow);
    }
    // ---- the row's scale over ALL features of the layer, then the split activations of both roles through LDS
    mrow = fmaxf(mrow, __shfl_xor(mrow, 32));
    pair_lds->pmax[role * 64 + lane] = mrow;
    RL_PMARK1(5);
    lds_barrier();
    RL_PMARK1(6);
    mrow = fmaxf(mrow, pair_lds->pmax[(role ^ 1) * 64 + lane]);
    float sc1, un1;
    row_scale(mrow, sc1, un1);
    auto fraw = [&](int c, int e) { return F[c >> 1][8 * (c & 1) + e]; };
#pragma unroll
    for (int c = 0; c < 2 * OWN1; ++c) {   // own chunks: registers 8 (c & 1) .. of own tile c >> 1 = chunk 2 * OWN1 * role + c of the layer
        f32x4 hi, lo;
#pragma unroll
        for (int k = 0; k < 6; ++k) split_slice<6>(k, [&](int e) { return fraw(c, e); }, sc1, hi, lo);
        pair_lds->ex[((2 * OWN1 * role + c) * kPlanes + 0) * 64 + lane] = hi;
        pair_lds->ex[((2 * OWN1 * role + c) * kPlanes + 1) * 64 + lane] = lo;
    }
    RL_PMARK1(7);
    lds_barrier();
    RL_PMARK1(8);
    float head4[4];
    float sc2, un2;
    if (KIND == RL_DQN) {
        // ---- hidden layer 128 -> 64: output tile `role`; main / cross accumulator chains (consecutive MFMAs independent)
        f32x4 B2[8][kPlanes];
#pragma unroll
        for (int c = 0; c < 8; ++c)
#pragma unroll
            for (int pl = 0; pl < kPlanes; ++pl) B2[c][pl] = pair_lds->ex[(c * kPlanes + pl) * 64 + lane];
        WRing<2, 1, 1, D>& w2 = w2d;
        WRing<1, 1, 1, 2>& wh = whd;
        f32x16 acc, cross;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[r] = 0.0f; cross[r] = 0.0f; }
#pragma unroll
        for (int s2 = 0; s2 < 8; ++s2) {
            f32x4 ac[1][kPlanes];
            w2.template next<8>(s2, ac);
            cross = mfma16(ac[0][0], B2[s2][1], cross);   // hi.lo
            acc = mfma16(ac[0][0], B2[s2][0], acc);       // hi.hi
            cross = mfma16(ac[0][1], B2[s2][0], cross);   // lo.hi
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] += cross[r];
        f32x16 dummy = acc;
        float m2 = 0.0f;
        ep.c = c2 + role * 64;
        ep.fetch(0, 0);
#pragma unroll
        for (int e = 0; e < 16; ++e) ep.step(0, e, acc, dummy, un1, m2);
        m2 = fmaxf(m2, __shfl_xor(m2, 32));
        row_scale(m2, sc2, un2);
        auto araw = [&](int c, int e) { return acc[8 * (c & 1) + e]; };
        head_stream<2, 2>(wh, *(const f32x4*)(hconsts + 4 * h), araw, sc2, un2, head4);
    } else {
        // ---- hidden layer 256 -> 256: output tiles 4 role .. 4 role + 3 in two passes, B from LDS
        WRingH<16, D, 32, 8>& w2 = w2p;
        f32x16 A[4];
        float m2 = 0.0f;
        k_pass_lds16<0>(w2, pair_lds->ex, lane, A[0], A[1], [&](int) {});
        RL_PMARK1(9);
        ep.c = c2 + role * 4 * 64;
        k_pass_lds16<1>(w2, pair_lds->ex, lane, A[2], A[3], [&](int slot) {
            if (slot == 0) ep.fetch(0, 0);
            if (slot >= 2 && slot < 34) ep.step(0, slot - 2, A[0], A[1], un1, m2);
        });
        RL_PMARK1(13);
        WRing<1, 1, 1, D> wh;
        wh.start(packed + L.ha, lane, 8 * role);
        ep.fetch(2, 0);
#pragma unroll
        for (int e = 0; e < 32; ++e) ep.step(2, e, A[2], A[3], un1, m2);
        m2 = fmaxf(m2, __shfl_xor(m2, 32));
        row_scale(m2, sc2, un2);
        RL_PMARK1(14);
        auto araw = [&](int c, int e) { return A[c >> 1][8 * (c & 1) + e]; };
        head_stream<D>(wh, *(const f32x4*)(hconsts + 4 * h), araw, sc2, un2, head4);
        RL_PMARK1(15);
    }
    if (role) {
        // (the lane index taken afresh: computed from `lane` this address is formed at the top of the tile, lives through all of it in a
        // VGPR the 250-register tile does not have -- spilled -- and comes back by a scratch reload + s_waitcnt vmcnt(0) right here, on the
        // role that finishes last)
        int l2 = 0;
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l2));
#endif
        *(f32x4*)(pair_lds->val + 4 * l2) = f32x4{head4[0], head4[1], head4[2], head4[3]};
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) part->head[r] = head4[r];
        part->draw = draw;
    }
}

// Role 0 after the caller's barrier: the two roles' head partials, bias, softmax (PPO.py:105) / Q values, the action (DQN.py:132-139:
// epsilon-greedy, first maximum; PPO.py:164-169: Categorical(prob).sample() as inverse CDF over the Philox uniform).
template <int KIND>
__device__ inline void pair_finish(const TileIO& io, int lane, const Tile1Part& part, const PairLds* pair_lds)
{
    extern __shared__ __attribute__((aligned(16))) char rl_dyn_lds[];
    constexpr int T1 = KIND == RL_PPO ? 8 : 4;
    const int h = lane >> 5;
    const float* const hconsts = (const float*)(rl_dyn_lds + io.c_lds_off) + T1 * 64 + (KIND == RL_PPO ? 512 : 128);
    const f32x4 other = *(const f32x4*)(pair_lds->val + 4 * lane), bias = *(const f32x4*)(hconsts + 8 + 4 * h);
    float a4[4] = {(part.head[0] + other.x) + bias.x, (part.head[1] + other.y) + bias.y, (part.head[2] + other.z) + bias.z, (part.head[3] + other.w) + bias.w};
    float o4[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) o4[r] = __shfl_xor(a4[r], 32);
    if (h == 0) {
        float q[8] = {a4[0], a4[1], a4[2], a4[3], o4[0], o4[1], o4[2], o4[3]};
        if (KIND == RL_PPO) {
            // softmax (PPO.py:105).  This runs on ONE wave per tile while every other wave of the workgroup waits at the barrier behind
            // the tiles: exp as v_exp_f32(x log2 e) (the arguments are <= 0: no range handling; ~1e-6 relative) and ONE division, not
            // eight calls of expf and eight divisions -- ~100 instructions instead of ~350 (2.8k -> 0.9k counts of the stamped build).
            float mx = q[0], sm = 0.0f;
#pragma unroll
            for (int i = 1; i < 8; ++i) mx = fmaxf(mx, q[i]);
#pragma unroll
            for (int i = 0; i < 8; ++i) { q[i] = __builtin_amdgcn_exp2f((q[i] - mx) * 1.44269504088896340736f); sm += q[i]; }
            const float inv = 1.0f / sm;
#pragma unroll
            for (int i = 0; i < 8; ++i) q[i] *= inv;
        }
        if (io.valid) {
            if (io.out) {
                f32x4* o = (f32x4*)(io.out + io.row * 8);
                o[0] = f32x4{q[0], q[1], q[2], q[3]};
                o[1] = f32x4{q[4], q[5], q[6], q[7]};
            }
            if (io.actions) {
                const float u = (float)rl_u24(part.draw.x);
                int a = 0;
                if (KIND == RL_PPO) {
                    float cum = 0.0f; a = 7; bool found = false;
#pragma unroll
                    for (int i = 0; i < 8; ++i) { cum += q[i]; if (!found && u < cum) { a = i; found = true; } }
                } else if (u < io.eps) a = (int)(part.draw.y >> 29);
                else {
#pragma unroll
                    for (int i = 1; i < 8; ++i) if (q[i] > q[a]) a = i;   // first maximum
                }
                io.actions[io.row] = (int8_t)a;
                if (io.lds_actions_off >= 0) ((signed char*)rl_dyn_lds)[io.lds_actions_off + io.lds_slot] = (signed char)a;
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------
// The dense tile: the weights through LDS stages AND the tile scheduled around the matrix pipe (as policy_tile1s: two passes per layer
// over output-tile pairs, the VALU work in the MFMAs' shadows).  A first version ran policy_tile1's phases on 8 KB stages of whole
// K-chunks (10,880 tiles: 157 us against 177-188 for the register-fed tiles); this one 148 us.  With the weight bytes out of the way the
// launch is bound by each wave's own chain and its 30 stage barriers (a lone 4-tile workgroup: 18 us).  A stage is 8 KB = two K-chunks of
// one tile pair (or four K-chunks of a head).  Same arithmetic per accumulator as every one-wave tile: bit-identical results.
// ---------------------------------------------------------------------------------------------------------------
struct WStage2 {
    f32x4* buf;       // LDS: 3 x kStageUnits
    gf32x4* src;      // the brain's packed weights in 16-byte units, + thread index (256 threads)
    f32x4 q[2][2];    // two stages in flight to LDS, two 4 KB blocks each
    int tid, lane;
    // source of block u (0, 1) of stage I: layers -- chunks c0 + u of tile pair h (4 KB each); heads -- the stage's two halves
    static __device__ inline int64_t off(const Layout& L, int I, int u)   // 16-byte units
    {
        if (I >= 18 && I < 20) return (L.ha + (int64_t)(I - 18) * 2048) / 4 + u * 256;
        if (I >= 28) return (L.hb + (int64_t)(I - 28) * 2048) / 4 + u * 256;
        const int64_t base = (I < 10 ? L.l1 : I < 18 ? L.l2a : L.l2b) / 4;
        const int jl = I < 10 ? I : I < 18 ? I - 10 : I - 20, per_pass = I < 10 ? kInChunks / 2 : 4;
        const int h = jl / per_pass, c0 = 2 * (jl % per_pass);
        return base + (int64_t)(c0 + u) * 512 + 256 * h;
    }
    __device__ inline void load(const Layout& L, int I, f32x4 (&dst)[2]) { dst[0] = src[off(L, I, 0)]; dst[1] = src[off(L, I, 1)]; }
    __device__ inline void put(int I, const f32x4 (&v)[2]) { buf[(I % 3) * kStageUnits + tid] = v[0]; buf[(I % 3) * kStageUnits + 256 + tid] = v[1]; }
    __device__ inline void start(const Layout& L)
    {
        f32x4 s0[2], s1[2];
        load(L, 0, s0); load(L, 1, s1); load(L, 2, q[0]); load(L, 3, q[1]);
        put(0, s0); put(1, s1);
        lds_barrier();
    }
    // Stage I: stage I + 2 is written to LDS from its registers (requested two stages ago), stage I + 4 is requested into the registers
    // that just became free, and ONE barrier publishes stage I + 2 -- so that the fragments of stage I + 1 (published a stage ago) can be
    // read while the MFMAs of stage I run.  Three buffers, and ONE RULE for the readers: stage J is read into registers before the
    // wave meets stage J's barrier (begin(J)); behind that barrier a faster wave's begin(J + 1) overwrites stage J's buffer.
    __device__ inline void begin(const Layout& L, int I)
    {
        if (I + 2 < kStages) put(I + 2, q[I & 1]);
        if (I + 4 < kStages) load(L, I + 4, q[I & 1]);
        lds_barrier();
    }
    __device__ inline f32x4 frag(int I, int unit) const { return buf[(I % 3) * kStageUnits + unit * 64 + lane]; }
};
// The stages of one layer seen as the weight ring of k_pass: step i (chunk i % NS of tile pair i / NS) = half (i & 1) of stage S0 + (i >> 1).
template <int NS, int S0>
struct WStageH {
    WStage2& ws;
    const Layout& L;
    f32x4 fr[2][8];   // the current stage's fragments and the next stage's on their way out of LDS
    __device__ inline WStageH(WStage2& w, const Layout& l) : ws(w), L(l) {}
    __device__ inline void read(int jl, f32x4 (&dst)[8])
    {
#pragma unroll
        for (int k = 0; k < 8; ++k) dst[k] = ws.frag(S0 + jl, k);
    }
    __device__ inline void take(int i, f32x4 (&ac)[2][kPlanes])
    {
        const int jl = i >> 1;
        if ((i & 1) == 0) {
            if (i == 0) read(0, fr[0]);                  // (published two stages ago)
            ws.begin(L, S0 + jl);
            if (jl + 1 < NS) read(jl + 1, fr[(jl + 1) & 1]);   // before the wave meets stage jl + 1's barrier
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int pl = 0; pl < kPlanes; ++pl) ac[t][pl] = fr[jl & 1][(i & 1) * 4 + t * kPlanes + pl];
    }
};
// head over two stages of four K-chunks, its input split chunk by chunk ahead of the MFMAs that use it (head_stream on stages)
template <int S0, typename Raw>
__device__ inline void head_stream_stage(WStage2& ws, const Layout& L, const f32x4& un4, Raw&& raw, float sc, float row_un, float (&out)[4])
{
    f32x16 a0, a1, a2;
#pragma unroll
    for (int r = 0; r < 16; ++r) { a0[r] = 0.0f; a1[r] = 0.0f; a2[r] = 0.0f; }
    f32x4 B[8][kPlanes];
#pragma unroll
    for (int k = 0; k < 3; ++k) split_slice<3>(k, [&](int e) { return raw(0, e); }, sc, B[0][0], B[0][1]);
#pragma unroll
    for (int hs = 0; hs < 2; ++hs) {
        f32x4 fr[4][kPlanes];
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int pl = 0; pl < kPlanes; ++pl) fr[c][pl] = ws.frag(S0 + hs, c * kPlanes + pl);
        ws.begin(L, S0 + hs);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int s = 4 * hs + c;
            a0 = mfma16(fr[c][0], B[s][1], a0);
            if (s + 1 < 8) split_slice<3>(0, [&](int e) { return raw(s + 1, e); }, sc, B[(s + 1) & 7][0], B[(s + 1) & 7][1]);
            __builtin_amdgcn_sched_barrier(0);
            a1 = mfma16(fr[c][0], B[s][0], a1);
            if (s + 1 < 8) split_slice<3>(1, [&](int e) { return raw(s + 1, e); }, sc, B[(s + 1) & 7][0], B[(s + 1) & 7][1]);
            __builtin_amdgcn_sched_barrier(0);
            a2 = mfma16(fr[c][1], B[s][0], a2);
            if (s + 1 < 8) split_slice<3>(2, [&](int e) { return raw(s + 1, e); }, sc, B[(s + 1) & 7][0], B[(s + 1) & 7][1]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) out[r] = ((a0[r] + a1[r]) + a2[r]) * (un4[r] * row_un);
}

// One wave's tile inside the 4-tile workgroup (every wave must call it: the stage barriers).  lds_consts: the brain's
// kTileConstFloats epilogue / head constants (LDS, filled by the caller before ws.start()).
template <int KIND>
__device__ inline void policy_tile1ds(const TileIO& io, int lane, WStage2& ws, const float* lds_consts)
{
    static_assert(KIND == RL_D3QN || KIND == RL_PERD3QN, "dense tile: dueling kinds");
    const int h = lane >> 5;
    const Layout L = layout_of(KIND);
    const float* const consts = lds_consts + 32 * h;
    const float* const hconsts = lds_consts + 768;
    ws.start(L);
    f32x4 X[kInChunks][2];
    {
        const int64_t rbase = io.row * RL_OBS_DIM;
#pragma unroll
        for (int c = 0; c < kInChunks; ++c)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int k0 = (c == kInChunks - 1 && h == 1) ? 149 : 16 * c + 8 * h + 4 * q;   // (see policy_tile1)
                X[c][q] = *(const f32x4u*)(io.obs + rbase + k0);
            }
    }
    rl_u4 draw = {0u, 0u, 0u, 0u};
    if (io.actions && io.eps > 0.0f) draw = rl_philox4x32(io.seed, io.key_epoch, io.key_world, io.key_tick, RL_SITE_ACT, io.key_index);
    if (h == 1) { X[kInChunks - 1][0] = f32x4{X[kInChunks - 1][0].w, 0.0f, 0.0f, 0.0f}; X[kInChunks - 1][1] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; }
    float m4[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int c = 0; c < kInChunks; ++c)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            max3_abs(m4[(2 * c + q) & 3], X[c][q].x, X[c][q].y);
            max3_abs(m4[(2 * c + q + 2) & 3], X[c][q].z, X[c][q].w);
        }
    float m = fmaxf(fmaxf(m4[0], m4[1]), fmaxf(m4[2], m4[3]));
    m = fmaxf(m, __shfl_xor(m, 32));
    float sc0, un0;
    row_scale(m, sc0, un0);
    f32x4 B1[kInChunks][kPlanes];
    auto xraw = [&](int c, int e) { return X[c][e >> 2][e & 3]; };
#pragma unroll
    for (int k = 0; k < 6; ++k) split_slice<6>(k, [&](int e) { return xraw(0, e); }, sc0, B1[0][0], B1[0][1]);
    // ---- input layer
    f32x16 F[4];
    EpiStream ep;
    float mrow = 0.0f;
    {
        WStageH<kInChunks, 0> w1(ws, L);
        k_pass<kInChunks, 0>(w1, B1, F[0], F[1], [&](int slot) {
            const int c = slot / 6 + 1;
            if (c < kInChunks) split_slice<6>(slot % 6, [&](int e) { return xraw(c, e); }, sc0, B1[c][0], B1[c][1]);
        });
        ep.c = consts;
        k_pass<kInChunks, 1>(w1, B1, F[2], F[3], [&](int slot) {
            if (slot == 0) ep.fetch(0, 0);
            if (slot >= 2 && slot < 34) ep.step(0, slot - 2, F[0], F[1], un0, mrow);
        });
    }
    ep.fetch(2, 0);
#pragma unroll
    for (int e = 0; e < 32; ++e) ep.step(2, e, F[2], F[3], un0, mrow);
    mrow = fmaxf(mrow, __shfl_xor(mrow, 32));
    float sc1, un1;
    row_scale(mrow, sc1, un1);
    f32x4 B2[8][kPlanes];
    auto fraw = [&](int c, int e) { return F[c >> 1][8 * (c & 1) + e]; };
#pragma unroll
    for (int k = 0; k < 6; ++k) split_slice<6>(k, [&](int e) { return fraw(0, e); }, sc1, B2[0][0], B2[0][1]);
    // ---- advantage branch
    f32x16 A[4];
    {
        WStageH<8, 10> w2(ws, L);
        k_pass<8, 0>(w2, B2, A[0], A[1], [&](int slot) {
            const int c = slot / 6 + 1;
            if (c < 8) split_slice<6>(slot % 6, [&](int e) { return fraw(c, e); }, sc1, B2[c][0], B2[c][1]);
        });
        mrow = 0.0f;
        ep.c = consts + 256;
        k_pass<8, 1>(w2, B2, A[2], A[3], [&](int slot) {
            if (slot == 0) ep.fetch(0, 0);
            if (slot >= 2 && slot < 34) ep.step(0, slot - 2, A[0], A[1], un1, mrow);
        });
    }
    ep.fetch(2, 0);
#pragma unroll
    for (int e = 0; e < 32; ++e) ep.step(2, e, A[2], A[3], un1, mrow);
    mrow = fmaxf(mrow, __shfl_xor(mrow, 32));
    float sc2, un2;
    row_scale(mrow, sc2, un2);
    auto araw = [&](int c, int e) { return A[c >> 1][8 * (c & 1) + e]; };
    float adv[4], val[4];
    head_stream_stage<18>(ws, L, *(const f32x4*)(hconsts + 4 * h), araw, sc2, un2, adv);
    // ---- value branch
    {
        WStageH<8, 20> w3(ws, L);
        k_pass<8, 0>(w3, B2, A[0], A[1], [&](int) {});
        mrow = 0.0f;
        ep.c = consts + 512;
        k_pass<8, 1>(w3, B2, A[2], A[3], [&](int slot) {
            if (slot == 0) ep.fetch(0, 0);
            if (slot >= 2 && slot < 34) ep.step(0, slot - 2, A[0], A[1], un1, mrow);
        });
    }
    ep.fetch(2, 0);
#pragma unroll
    for (int e = 0; e < 32; ++e) ep.step(2, e, A[2], A[3], un1, mrow);
    mrow = fmaxf(mrow, __shfl_xor(mrow, 32));
    row_scale(mrow, sc2, un2);
    head_stream_stage<28>(ws, L, *(const f32x4*)(hconsts + 16 + 4 * h), araw, sc2, un2, val);
    tile1_finish<KIND>(io, lane, adv, val[0] + hconsts[16 + 8], draw, *(const f32x4*)(hconsts + 8 + 4 * h));
}

}  // namespace
