"""Race evidence for the hand-fenced kernels (race detection; GPU).

The kernels replace __syncthreads() by `s_waitcnt lgkmcnt(0); s_barrier` (lds_barrier: global stores and the weight-prefetch ring stay in
flight across barriers), and since round 6 the multi-tick launch leaves out a barrier between ticks (RL_SEAM_OPEN).  Builds:
    product                                    lib/libreinlife_hip.so
    RL_FULL_FENCE (every lds_barrier = __syncthreads(): waits for ALL counters, fences) + RL_SEAM_OPEN=0     lib/libreinlife_hip_fence.so
        RL_LIB_TAG=fence RL_EXTRA_HIPCC_FLAGS="-DRL_FULL_FENCE -DRL_SEAM_OPEN=0" python reinlife_amd/build.py
Every case below runs free (policy-driven, refills, multi-tick launches of uneven lengths) in a process per library from the same seeds;
what a case leaves behind -- the whole state arena, both Agent.state buffers, state_prime, actions, rewards, counters, Tracker sums,
replay rings -- is hashed.  Same seeds -> same digests, or the elided waits are load-bearing somewhere.
    python tools/fence_ab.py [libA.so libB.so]          (default: the two above)
    python tools/fence_ab.py --child                     (one library, REINLIFE_HIP_LIB: prints the digests)"""
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CASES = [  # name, worlds, brain kinds, static, launches (ticks each), mode
    ("c4 256 worlds, dueling kernel", 256, ["PERD3QN", "PERD3QN"], True, [700, 1, 333, 2, 964], "plain"),
    ("c5 256 worlds, mixed-kind kernel", 256, ["PPO", "PERD3QN"], False, [500, 3, 497], "plain"),
    ("c4 64 worlds, TRAIN 1 (Tracker + epsilon schedule)", 64, ["PERD3QN", "D3QN"], True, [400, 7, 393], "train"),
    ("3 kinds 48 worlds, TRAIN 2 (capture + policy outputs)", 48, ["DQN", "PPO", "D3QN"], False, [40, 5, 35], "capture"),
    ("two-launch loop, 128 worlds", 128, ["PERD3QN", "PERD3QN"], True, [300], "two-launch"),
    ("two-launch loop, mixed kinds, 96 worlds", 96, ["PPO", "DQN"], False, [300], "two-launch"),
]


def child():
    import numpy as np
    import torch
    import bench
    from reinlife_amd import _lib
    from reinlife_amd.worlds import DeviceWorlds, pack_brain_weights
    out = {}
    for name, R, kinds, static, launches, mode in CASES:
        dw = DeviceWorlds(n_worlds=R, seed=977, width=30, height=30, max_agents=100, n_brains=len(kinds), static_families=static)
        dw.set_brains([(_lib.KIND_BY_METHOD[n], 0.05 * k, pack_brain_weights(_lib.KIND_BY_METHOD[n], bench.brain_weights(n, 100 + k))) for k, n in enumerate(kinds)])
        if mode in ("train", "capture"):
            dw.enable_tracking(True)
        if mode == "capture":
            dw.enable_capture(1 << 18, with_prob=True)   # (no wrap-around: WHICH rows survive a wrap depends on the worlds' arrival order at the ring's counter)
        dw.reset_synthetic(100)
        rng = np.random.RandomState(3)
        h = hashlib.sha256()
        steps = 0
        for n in launches:
            if mode == "two-launch":
                for _ in range(n):
                    dw.act(); dw.tick_refill(70, 100)
            else:
                sched = rng.uniform(0, 0.3, size=(n, len(kinds))).astype(np.float32) if mode != "plain" else None
                dw.run(n, 70, 100, eps_schedule=sched, want_q=(mode == "capture"))
            torch.cuda.synchronize(); dw.check_error_flag()
            for key in sorted(dw.s.keys()):      # every rl_state array
                h.update(key.encode()); h.update(dw.s[key].cpu().numpy().tobytes())
            for t in (dw.actions, dw.n_acted, dw.reward, dw.done, dw.src1, dw.src2, dw.obs_state(), dw.obs_state_prime(), dw.prev_state()):
                h.update(t.cpu().numpy().tobytes())
            if mode in ("train", "capture"):
                for t in (dw.trk_tick, dw.trk_sum, dw.trk_cnt, dw.trk_pop):
                    h.update(t.cpu().numpy().tobytes())
            if mode == "capture":   # the rings as MULTISETS of transitions: a ring's row order across worlds is the order of the worlds' atomics (timing, by design)
                for r in dw.replays:
                    cnt = int(r["count"].item())
                    assert cnt <= r["state"].shape[0], "ring wrapped: enlarge it"
                    cols = [r[k][:cnt].cpu().numpy().reshape(cnt, -1).astype(np.float64) for k in ("state", "state_prime", "action", "reward", "done", "age", "prob") if r.get(k) is not None]
                    rows = np.concatenate(cols, axis=1)
                    rows = rows[np.lexsort(rows.T[::-1])]
                    h.update(np.int64(cnt).tobytes()); h.update(np.ascontiguousarray(rows).tobytes())
            steps = int(dw.acted_total.item())
        out[name] = {"sha256": h.hexdigest()[:32], "agent_steps": steps, "refills": int(dw.refill_count.item()), "ticks": sum(launches)}
        del dw
    print(json.dumps(out))


def main():
    libs = sys.argv[1:3] if len(sys.argv) >= 3 else [os.path.join(ROOT, "reinlife_amd", "lib", "libreinlife_hip.so"),
                                                     os.path.join(ROOT, "reinlife_amd", "lib", "libreinlife_hip_fence.so")]
    res = []
    for lib in libs:
        env = dict(os.environ, REINLIFE_HIP_LIB=os.path.abspath(lib))
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, capture_output=True, text=True)
        if p.returncode != 0:
            print(p.stderr[-3000:]); raise SystemExit("child failed for %s" % lib)
        res.append(json.loads(p.stdout.strip().splitlines()[-1]))
    from reinlife_amd import build
    print("# tools/fence_ab.py on kernel sources %s: %s  vs  %s" % (build.source_hash(), *[os.path.basename(x) for x in libs]))
    bad = 0
    for name, *_ in CASES:
        a, b = res[0][name], res[1][name]
        same = a == b
        bad += not same
        print("%-58s %6d ticks %10d agent-steps %5d refills  %s  %s" % (name, a["ticks"], a["agent_steps"], a["refills"], a["sha256"], "== same bits" if same else "!= " + b["sha256"]))
    print("RESULT: %s" % ("all %d cases bit-identical between the two builds" % len(CASES) if not bad else "%d case(s) DIFFER" % bad))
    raise SystemExit(1 if bad else 0)


if __name__ == "__main__":
    child() if "--child" in sys.argv else main()
