"""TEST INFRASTRUCTURE ONLY -- the CPU baseline leg of bench.py: the oracle (oracle/rl_oracle.c, a plain-C port of the
reference's world tick) plus a BATCHED float32 policy forward (numpy sgemm over all agents of a worker's worlds -- what the
reference's torch-CPU networks would cost if they were batched; the reference itself runs one batch-1 forward per agent,
Helpers/trainer.py:88-89), timed on the host cores with one PROCESS per core (no GIL, BLAS pinned to one thread each).

Four rates are reported, single-thread and all-core, over the same workload as the GPU line (30x30 worlds filled to 100
agents, refill below 70, greedy brains):  world_only (step + update_env + refill),  policy_only (batched forward + argmax),
full_tick (both; the number bench.py puts in cpu_baseline.value),  policy_plus_step (variant (i) of SURVEY.md 8d: get_action +
step, with update_env + refill running untimed in between -- the counterpart of the GPU line's variant_policy_plus_step).
"""
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

SHAPES = {"DQN": [(128, 153), (64, 128), (8, 64)],
          "D3QN": [(128, 153), (128, 128), (8, 128), (128, 128), (1, 128)],
          "PPO": [(256, 153), (256, 256), (8, 256), (1, 256)]}


def unpack(kind, flat):
    """flat float32 state-dict vector -> [(W, b), ...] in registration order."""
    out, o = [], 0
    for n_out, n_in in SHAPES["D3QN" if kind == "PERD3QN" else kind]:
        W = flat[o:o + n_out * n_in].reshape(n_out, n_in); o += n_out * n_in
        b = flat[o:o + n_out]; o += n_out
        out.append((np.ascontiguousarray(W.T), b))  # x @ W.T as one sgemm on a contiguous operand
    return out


def forward(kind, layers, x):
    """Batched float32 forward of the reference networks (DQN.py:126-130, PERD3QN.py:198-202, PPO.py:101-106)."""
    relu = lambda v: np.maximum(v, 0.0, out=v)  # noqa: E731
    if kind == "DQN":
        h = relu(x @ layers[0][0] + layers[0][1])
        h = relu(h @ layers[1][0] + layers[1][1])
        return h @ layers[2][0] + layers[2][1]
    if kind in ("D3QN", "PERD3QN"):
        f = relu(x @ layers[0][0] + layers[0][1])
        adv = relu(f @ layers[1][0] + layers[1][1]) @ layers[2][0] + layers[2][1]
        val = relu(f @ layers[3][0] + layers[3][1]) @ layers[4][0] + layers[4][1]
        return adv + val - adv.mean(axis=1, keepdims=True)
    h = relu(x @ layers[0][0] + layers[0][1])
    h = relu(h @ layers[1][0] + layers[1][1])
    lg = h @ layers[2][0] + layers[2][1]
    e = np.exp(lg - lg.max(axis=1, keepdims=True))
    return e / e.sum(axis=1, keepdims=True)


def _worker(idx, n_worlds, brains, static_families, seed, weights, duration, start_at):
    from oracle import oracle as orc
    ow = orc.OracleWorlds(n_worlds=n_worlds, n_brains=len(brains), static_families=static_families, seed=seed,
                          world_base=idx * n_worlds)
    ow.reset_synthetic(100)
    layers = [unpack(k, w) for k, w in zip(brains, weights)]
    acts = np.zeros((n_worlds, ow.cap), np.int8)
    rng = np.random.RandomState(seed + idx)
    cap = ow.cap

    def tick():
        n = ow.s["n_agents"]
        t0 = time.perf_counter()
        live = np.arange(cap)[None, :] < n[:, None]
        br = ow.s["a_brain"]
        for b, kind in enumerate(brains):
            ws, ks = np.nonzero(live & (br == b))
            if len(ws) == 0:
                continue
            out = forward(kind, layers[b], ow.obs2[ws, ks])
            if kind == "PPO":  # Categorical(prob).sample() as inverse CDF (PPO.py:164-169)
                a = (out.cumsum(axis=1) < rng.random_sample((len(ws), 1)).astype(np.float32)).sum(axis=1).clip(0, 7)
            else:
                a = out.argmax(axis=1)
            acts[ws, ks] = a
        t1 = time.perf_counter()
        steps = int(n.sum())
        ow.step(acts)
        ts = time.perf_counter()
        ow.update()
        ow.refill(70, 100)
        t2 = time.perf_counter()
        return steps, t1 - t0, t2 - t1, ts - t1

    for _ in range(3):
        tick()
    while time.time() < start_at:  # common start: every process is warmed up before the window opens
        time.sleep(0.005)
    t_start = time.perf_counter()
    steps = ticks = 0
    t_pol = t_world = t_step = 0.0
    while time.perf_counter() - t_start < duration:
        s, a, b, c = tick()
        steps += s; ticks += 1; t_pol += a; t_world += b; t_step += c
    return (idx, steps, ticks, t_pol, t_world, time.perf_counter() - t_start, t_step)


def _run(n_procs, n_worlds, brains, static_families, seed, wfile, duration):
    """One independent interpreter per core (`python -m oracle.cpu_bench --worker ...`): nothing is inherited from the
    parent (which holds a HIP context), and no GIL is shared."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    start_at = time.time() + 3.0 + 0.05 * n_procs  # interpreter + numpy import + oracle warm-up
    cmd = [sys.executable, "-m", "oracle.cpu_bench", "--worker", "--n-worlds", str(n_worlds), "--brains", ",".join(brains),
           "--static", str(int(static_families)), "--seed", str(seed), "--weights", wfile, "--duration", str(duration),
           "--start-at", repr(start_at)]
    procs = [subprocess.Popen(cmd + ["--idx", str(i)], cwd=root, stdout=subprocess.PIPE, text=True) for i in range(n_procs)]
    res = []
    for p in procs:
        out, _ = p.communicate(timeout=duration + 180)
        if p.returncode != 0:
            raise RuntimeError("cpu_bench worker failed (rc %d)" % p.returncode)
        res.append(json.loads(out.strip().splitlines()[-1]))
    steps = sum(r[1] for r in res)
    wall = max(r[5] for r in res)
    t_pol = sum(r[3] for r in res)
    t_world = sum(r[4] for r in res)
    t_step = sum(r[6] for r in res)
    return {"full_tick": round(steps / wall, 1),
            # variant (i) of SURVEY.md 8d / BASELINE.md 3: get_action + step only (update_env and the refill run, untimed)
            "policy_plus_step": round(steps / ((t_pol + t_step) / n_procs), 1),
            # per-leg rates: agent-steps per second of that leg's own time, summed over the processes running in parallel
            "world_only": round(steps / (t_world / n_procs), 1), "policy_only": round(steps / (t_pol / n_procs), 1),
            "ticks": sum(r[2] for r in res), "agent_steps": steps, "seconds": round(wall, 2)}


def run(brains, static_families, seed, weights, cores=None, worlds_per_proc=8, single_seconds=4.0, all_seconds=6.0):
    """-> dict for bench.py's `cpu_baseline`.  `weights`: flat float32 state-dict vectors, one per brain."""
    from oracle import oracle as orc
    orc.build()
    cores = cores or max(1, os.cpu_count() or 1)
    saved = {k: os.environ.get(k) for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS")}
    os.environ.update(OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")  # inherited by the workers
    wfile = tempfile.NamedTemporaryFile(suffix=".npz", delete=False).name
    np.savez(wfile, *[np.asarray(w, np.float32) for w in weights])
    try:
        single = _run(1, worlds_per_proc, brains, static_families, seed, wfile, single_seconds)
        allc = _run(cores, worlds_per_proc, brains, static_families, seed, wfile, all_seconds) if cores > 1 else single
    finally:
        os.unlink(wfile)
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    return {"value": allc["full_tick"], "unit": "agent-steps/s", "cores": cores, "kind": "port",
            "sample": "%d processes x %d worlds x %.0f s of the same workload (refill below 70): oracle/rl_oracle.c world tick + batched "
                      "float32 numpy (sgemm, 1 BLAS thread per process) policy forward; then 1 process x %.0f s"
                      % (cores, worlds_per_proc, all_seconds, single_seconds),
            "all_cores": allc, "single_thread": single}


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--worker", action="store_true")
    ap.add_argument("--idx", type=int, default=0)
    ap.add_argument("--n-worlds", type=int, default=8)
    ap.add_argument("--brains", default="PERD3QN,PERD3QN")
    ap.add_argument("--static", type=int, default=1)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--weights")
    ap.add_argument("--duration", type=float, default=4.0)
    ap.add_argument("--start-at", type=float, default=0.0)
    a = ap.parse_args()
    z = np.load(a.weights)
    wts = [z["arr_%d" % i] for i in range(len(z.files))]
    print(json.dumps(_worker(a.idx, a.n_worlds, a.brains.split(","), bool(a.static), a.seed, wts, a.duration, a.start_at)))
