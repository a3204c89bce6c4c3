// M1 harness: the dueling network on a 16-row tile with v_mfma_f32_16x16x32_f16, arithmetic of the canonical 32-row tiles (one wave, both branches)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "rl_policy_dev.h"
typedef float f32x4_t __attribute__((ext_vector_type(4)));

namespace {
__device__ inline f32x4 mfma1632(const f32x4& a, const f32x4& b, f32x4 c)
{
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
// LDS: B operand planes in canonical slot order: entry (chunk s, plane, h, row) = 16 B
__device__ inline int ex_idx(int s, int plane, int h, int row) { return ((s * 2 + plane) * 2 + h) * 16 + row; }

// one layer on the 16-row tile: NS chunks of K, TOUT output tiles; B from LDS `ex`, weights fragments at `frag` ([chunk][tile][plane][lane] 16 B)
template <int NS, int TOUT>
__device__ inline void layer16(gfloat* frag, const f32x4* ex, int lane, f32x4 (&d)[TOUT][2])
{
    const int l16 = lane & 15, g = lane >> 4, h = g & 1, G = g >> 1;
    gf32x4* fr = (gf32x4*)frag;
#pragma unroll
    for (int t = 0; t < TOUT; ++t)
#pragma unroll
        for (int fb = 0; fb < 2; ++fb) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < (3 * NS) / 2; ++j) {
                const int st = 2 * j + G, s = st / 3, ty = st % 3;
                const int wpl = ty == 2 ? 1 : 0, xpl = ty == 0 ? 1 : 0;
                const f32x4 a = fr[((s * TOUT + t) * 2 + wpl) * 64 + 16 * fb + l16 + 32 * h];
                const f32x4 b = ex[ex_idx(s, xpl, h, l16)];
                acc = mfma1632(a, b, acc);
            }
            d[t][fb] = acc;
        }
}
// epilogue of a layer's TOUT tiles (consts: [tile][half][un 16 | bias 16] in the canonical register order), row maximum
template <int TOUT>
__device__ inline float epilogue16(gfloat* consts, int lane, f32x4 (&d)[TOUT][2], float row_un)
{
    const int g = lane >> 4;
    float m = 0.f;
#pragma unroll
    for (int t = 0; t < TOUT; ++t)
#pragma unroll
        for (int fb = 0; fb < 2; ++fb) {
            const int base = t * 64 + (g & 1) * 32 + 4 * (2 * fb + (g >> 1));
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float y = fmaxf(__builtin_fmaf(d[t][fb][r], consts[base + r] * row_un, consts[base + 16 + r]), 0.0f);
                d[t][fb][r] = y; m = fmaxf(m, y);
            }
        }
    m = fmaxf(m, __shfl_xor(m, 16)); m = fmaxf(m, __shfl_xor(m, 32));
    return m;
}
// split a layer's outputs into the LDS planes of the next layer's B operand
template <int TOUT>
__device__ inline void split_to_ex16(const f32x4 (&d)[TOUT][2], float sc, f32x4* ex, int lane)
{
    const int l16 = lane & 15, g = lane >> 4;
#pragma unroll
    for (int t = 0; t < TOUT; ++t)
#pragma unroll
        for (int fb = 0; fb < 2; ++fb) {
            float h0, h1, l0, l1;
            split_hi(d[t][fb][0], d[t][fb][1], sc, h0); split_lo(d[t][fb][0], d[t][fb][1], sc, h0, l0);
            split_hi(d[t][fb][2], d[t][fb][3], sc, h1); split_lo(d[t][fb][2], d[t][fb][3], sc, h1, l1);
            float* eh = (float*)&ex[ex_idx(2 * t + fb, 0, g & 1, l16)] + 2 * (g >> 1);
            float* el = (float*)&ex[ex_idx(2 * t + fb, 1, g & 1, l16)] + 2 * (g >> 1);
            eh[0] = h0; eh[1] = h1; el[0] = l0; el[1] = l1;
        }
}
// head over K = 128 (8 chunks): three chains, outputs o = 4 g + r (g < 2) of the 16 rows
__device__ inline void head16(gfloat* hw, const f32x4* ex, int lane, float row_un, float (&out)[4])
{
    const int l16 = lane & 15, g = lane >> 4, h = g & 1, G = g >> 1;
    gf32x4* fr = (gf32x4*)hw;
    f32x4 a[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        a[k] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int wpl = k == 2 ? 1 : 0, xpl = k == 0 ? 1 : 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int s = 2 * j + G;
            const f32x4 wa = fr[(s * 2 + wpl) * 64 + l16 + 32 * h];
            const f32x4 xb = ex[ex_idx(s, xpl, h, l16)];
            a[k] = mfma1632(wa, xb, a[k]);
        }
    }
    gfloat* hc = hw + head_consts_off(4);
#pragma unroll
    for (int r = 0; r < 4; ++r) { const int o = (4 * g + r) & 7; out[r] = ((a[0][r] + a[1][r]) + a[2][r]) * (hc[o] * row_un); }
}
}  // namespace

extern "C" __global__ __launch_bounds__(64) void half16_forward(const float* packed_, const float* obs, int n_rows, float* out)
{
    __shared__ f32x4 ex0[10 * 2 * 2 * 16], ex1[8 * 2 * 2 * 16], ex2[8 * 2 * 2 * 16];
    gfloat* packed = (gfloat*)packed_;
    const Layout L = layout_of(RL_PERD3QN);
    const int lane = threadIdx.x, l16 = lane & 15, g = lane >> 4;
    const int row = l16 < n_rows ? l16 : 0;
    // ---- the row: lane (row, g) takes chunks g, g + 4, g + 8
    float xs[3][16];
    float m = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int c = g + 4 * i;
#pragma unroll
        for (int e = 0; e < 16; ++e) { const int k = 16 * c + e; xs[i][e] = (c < 10 && k < 153) ? obs[(size_t)row * 153 + k] : 0.f; m = fmaxf(m, fabsf(xs[i][e])); }
    }
    m = fmaxf(m, __shfl_xor(m, 16)); m = fmaxf(m, __shfl_xor(m, 32));
    float sc0, un0; row_scale(m, sc0, un0);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int c = g + 4 * i;
        if (c < 10)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                float hi[4], lo[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) { split_hi(xs[i][8 * hh + 2 * q], xs[i][8 * hh + 2 * q + 1], sc0, hi[q]); split_lo(xs[i][8 * hh + 2 * q], xs[i][8 * hh + 2 * q + 1], sc0, hi[q], lo[q]); }
                ex0[ex_idx(c, 0, hh, l16)] = f32x4{hi[0], hi[1], hi[2], hi[3]};
                ex0[ex_idx(c, 1, hh, l16)] = f32x4{lo[0], lo[1], lo[2], lo[3]};
            }
    }
    __syncthreads();
    // ---- input layer
    f32x4 d1[4][2];
    layer16<10, 4>(packed + L.l1, ex0, lane, d1);
    const float m1 = epilogue16<4>(packed + L.l1 + frag_floats(kInChunks, 4), lane, d1, un0);
    float sc1, un1; row_scale(m1, sc1, un1);
    split_to_ex16<4>(d1, sc1, ex1, lane);
    __syncthreads();
    // ---- advantage branch
    f32x4 d2[4][2];
    layer16<8, 4>(packed + L.l2a, ex1, lane, d2);
    const float m2 = epilogue16<4>(packed + L.l2a + frag_floats(8, 4), lane, d2, un1);
    float sc2, un2; row_scale(m2, sc2, un2);
    split_to_ex16<4>(d2, sc2, ex2, lane);
    __syncthreads();
    float adv[4];
    head16(packed + L.ha, ex2, lane, un2, adv);
    __syncthreads();
    // ---- value branch
    layer16<8, 4>(packed + L.l2b, ex1, lane, d2);
    const float m3 = epilogue16<4>(packed + L.l2b + frag_floats(8, 4), lane, d2, un1);
    float sc3, un3; row_scale(m3, sc3, un3);
    split_to_ex16<4>(d2, sc3, ex2, lane);
    __syncthreads();
    float val[4];
    head16(packed + L.hb, ex2, lane, un3, val);
    // ---- dueling combine (tile1_finish's order)
    gfloat* hca = packed + L.ha + head_consts_off(4);
    gfloat* hcv = packed + L.hb + head_consts_off(4);
    float a4[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) a4[r] = adv[r] + hca[8 + ((4 * g + r) & 7)];
    float o4[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) o4[r] = __shfl_xor(a4[r], 16);   // lane (row, 1)'s four advantages for lane (row, 0)
    if (g == 0) {
        const float advs[8] = {a4[0], a4[1], a4[2], a4[3], o4[0], o4[1], o4[2], o4[3]};
        float mean = 0.0f;
#pragma unroll
        for (int i = 0; i < 8; ++i) mean += advs[i];
        mean *= 0.125f;
        const float v = val[0] + hcv[8];
        if (l16 < n_rows)
#pragma unroll
            for (int i = 0; i < 8; ++i) out[(size_t)l16 * 8 + i] = advs[i] + v - mean;
    }
}
