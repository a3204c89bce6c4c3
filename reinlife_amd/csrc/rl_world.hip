// rl_world.hip -- the ReinLife world kernels on MI355X (gfx950): k_world (step / update / fused tick / observe), k_reset (world
// generator, refill), k_capture (transition capture) and their host launchers.  The device code lives in rl_world_dev.h; the
// multi-tick kernel (k_run) in rl_run.hip.
#include "rl_world_dev.h"

#ifdef RL_PHASE_PROFILE   /* the stamped tuning build only: sections of the tick can be skipped (results WRONG) */
int g_rl_ablate = 0;
extern "C" __attribute__((visibility("default"))) void rl_debug_set_ablate(int mask) { g_rl_ablate = mask; }
#endif

namespace {

template <int T, int MODE, bool LEAN, bool FIXED = false>
__global__ __launch_bounds__(T) void k_world(const KParams p_in)
{
    KParams p = p_in;
    if (FIXED) {
        p.W = kFixW; p.H = kFixH; p.C = kFixC; p.Cp = kFixCp; p.nW = kFixCp / 64; p.PS = kFixW; p.invW = kFixInvW;
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
        int t0, t1, t2, t3, t4, t5, t6, t7, t8;
        asm volatile("s_load_dword %0, %9, 0x0\n\ts_load_dword %1, %9, 0x40\n\ts_load_dword %2, %9, 0x80\n\t"
                     "s_load_dword %3, %9, 0xc0\n\ts_load_dword %4, %9, 0x100\n\ts_load_dword %5, %9, 0x140\n\t"
                     "s_load_dword %6, %9, 0x180\n\ts_load_dword %7, %9, 0x1c0\n\ts_load_dword %8, %9, 0x200\n\ts_waitcnt lgkmcnt(0)"
                     : "=s"(t0), "=s"(t1), "=s"(t2), "=s"(t3), "=s"(t4), "=s"(t5), "=s"(t6), "=s"(t7), "=s"(t8)
                     : "s"(ka) : "memory");
        static_assert(sizeof(KParams) >= 0x200 + 4 && sizeof(KParams) <= 0x240, "the warm-up loads must cover the argument block");
    }
    Smem s;
    if (FIXED) carve(s, smem_raw, kFixCp, kFixCap, kFixHash, kFixPp);
    else carve(s, smem_raw, p.Cp, p.cap, p.hash_size, plane_words(p.PS, p.H, p.Cp));
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
            if (kPlanesEarly) patch_placed_planes(p, s);
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

template <int T>
__global__ __launch_bounds__(T) void k_reset(const KParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    Smem s;
    carve(s, smem_raw, p.Cp, p.cap, p.hash_size, plane_words(p.PS, p.H, p.Cp));
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

template <int MODE, bool LEAN>
int launch_world_v(const rl_world* h, const KParams& p, hipStream_t stream)
{
    const int blk = pick_block(h);
    static const size_t fixed_bytes = rl_world_smem_bytes(kFixCp, kFixCap, kFixHash, kFixW, kFixH);  // inside the default 64 KB window
    const bool fixed = LEAN && MODE == MODE_TICK && p.W == kFixW && p.H == kFixH && p.cap == kFixCap && p.hash_size == kFixHash &&
                       fixed_bytes <= 64 * 1024 && !h->opt.world_generic;  // (option: run the generic code -- tests, A/B)
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

}  // namespace


// ---------------------------------------------------------------------------------------------------------------
// host entry points used by rl_capi.hip
// ---------------------------------------------------------------------------------------------------------------
size_t rl_world_smem_bytes(int cpad, int cap, int hash, int plane_stride, int height)
{
    Smem s;
    return carve(s, nullptr, cpad, cap, hash, plane_words(plane_stride, height, cpad));
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
int rl_world_launch_reset(rl_world* h, int n_agents, int threshold, float* obs, int32_t* refill_count, int families, hipStream_t st)
{
    KParams p = make_params(h);
    set_list_production(h, p, true);
    p.reset_n_agents = n_agents; p.refill_threshold = threshold; p.obs_only = obs; p.refill_count = refill_count;
    p.reset_families = families;
    if (int rc = rl_world_prepare_bytes(h->smem_bytes)) return rc;
    const int blk = pick_block(h);
    if (blk == 1024) hipLaunchKernelGGL((k_reset<1024>), dim3(h->cfg.n_worlds), dim3(1024), h->smem_bytes, st, p);
    else if (blk == 512) hipLaunchKernelGGL((k_reset<512>), dim3(h->cfg.n_worlds), dim3(512), h->smem_bytes, st, p);
    else hipLaunchKernelGGL((k_reset<256>), dim3(h->cfg.n_worlds), dim3(256), h->smem_bytes, st, p);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { rl_set_error("reset kernel launch failed: %s", hipGetErrorString(e)); return RL_E_LAUNCH; }
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
