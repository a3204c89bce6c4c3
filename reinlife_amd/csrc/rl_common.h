// Shared host/device helpers of libreinlife_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/reinlife_hip.h"

// Tuning / test switches.  Process-level defaults are read from the environment ONCE (first use: RL_WORLD_BLOCK, RL_WORLD_GENERIC,
// RL_POLICY_VARIANT, RL_RUN_ALWAYS) and changed only by rl_set_option(); rl_create() snapshots them into the handle,
// so no launch ever calls getenv and a handle's kernels do not change under it.  The handle-less rl_policy_forward reads the process level.
enum { RL_PV_AUTO = 0, RL_PV_WAVE = 2, RL_PV_DENSE, RL_PV_PAIR };   // (1 was the 4-wave "nsplit" tile of rounds 1-2, removed in round 5)
struct rl_options {
    int world_block;      // 0 = by world count; 256 / 512 / 1024 = workgroup size of the world kernels and of rl_run
    int world_generic;    // 1 = the generic (not shape-specialised) world code also for the default 30x30 / 100-agent shape
    int policy_variant;   // RL_PV_*: which stand-alone policy kernel (auto: the tiles of rl_run's policy half, one arithmetic everywhere)
    int run_always;       // 1 = DeviceWorlds.run() takes the multi-tick launch at any world count (host side reads it through rl_get_option)
};
const rl_options& rl_options_current();

struct rl_world {
    rl_config cfg;
    rl_options opt;      // snapshot of the process-level options at rl_create
    rl_state st;
    int bound;
    int device;          // HIP device the state buffers live on (-1: unknown)
    int32_t* err_flag;   // device, optional
    int cells;           // width*height
    int cpad;            // cells rounded up to 64
    int hash_size;       // power of two >= 2*slot_cap
    size_t smem_bytes;   // dynamic LDS of the world kernels
    int block;           // threads per world workgroup
    void* work;          // bound policy work buffer (device) or null
    int lists_valid;     // the row lists in `work` describe the current world state
    int lists_parity;    // which counter half holds them
    int parity_next;     // half the next list producer (world launch or k_bucket) writes
    long long* prof;     // optional device int64[32]: shader-clock stamps of one world's phases
    int prof_world;
};

void rl_set_error(const char* fmt, ...);

// Row-list entry of the policy work buffer: (world << 12) | slot (slot_cap <= 4096, n_worlds < 2^19), so that the policy
// kernel gets world and slot without an integer division by the runtime slot capacity.
__host__ __device__ inline int rl_list_entry(int world, int slot) { return (world << 12) | slot; }
__host__ __device__ inline int rl_list_world(int e) { return e >> 12; }
__host__ __device__ inline int rl_list_slot(int e) { return e & 4095; }

// ---------------------------------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al., SC'11).  counter = (index, site, tick, world), key = (seed_lo, seed_hi ^ epoch*phi)
// ---------------------------------------------------------------------------------------------------------------
struct rl_u4 { uint32_t x, y, z, w; };

__host__ __device__ inline uint32_t rl_mulhi(uint32_t a, uint32_t b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __umulhi(a, b);
#else
    return (uint32_t)(((uint64_t)a * b) >> 32);
#endif
}

__host__ __device__ inline rl_u4 rl_philox4x32(uint64_t seed, uint32_t epoch, uint32_t world, uint32_t tick,
                                               uint32_t site, uint32_t index)
{
    uint32_t c0 = index, c1 = site, c2 = tick, c3 = world;
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32) ^ (epoch * 0x9E3779B9u);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        // one 32x32->64 product per multiplier (v_mad_u64_u32 on the device: both halves from ONE quarter-rate instruction
        // instead of v_mul_hi_u32 + v_mul_lo_u32)
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t h0 = (uint32_t)(p0 >> 32), l0 = (uint32_t)p0;
        const uint32_t h1 = (uint32_t)(p1 >> 32), l1 = (uint32_t)p1;
        const uint32_t n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
        c0 = n0; c1 = l1; c2 = n2; c3 = l0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return rl_u4{c0, c1, c2, c3};
}

// 24-bit uniform in [0,1), exactly representable in float and double
__host__ __device__ inline double rl_u24(uint32_t x) { return (double)(x >> 8) * (1.0 / 16777216.0); }
