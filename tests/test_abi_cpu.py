"""CPU: libreinlife_hip.so loads, exports every symbol include/reinlife_hip.h declares, and its host-only entry
points (validation, weight packing, Philox) behave.  No compute call is made (no GPU here)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from reinlife_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "reinlife_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(rl_[a-z_0-9]+)\s*\(", hdr))
    assert len(declared) >= 19
    lib = _lib.lib()
    for name in sorted(declared):
        assert hasattr(lib, name), "missing export %s" % name
    assert declared == {n for n, _, _ in _lib.ABI}, "ctypes ABI table out of sync with the header"


def test_the_library_exports_the_header_and_nothing_else():
    """The product library is built with -fvisibility=hidden + a version script: its dynamic symbol table is EXACTLY the functions
    include/reinlife_hip.h declares -- no measurement switch (rl_debug_*: tuning builds only), no C++ internals."""
    import subprocess
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "reinlife_hip.h")).read(), flags=re.S)
    declared = set(re.findall(r"\b(rl_[a-z_0-9]+)\s*\(", hdr))
    _lib.lib()
    from reinlife_amd import build
    nm = "/opt/rocm/lib/llvm/bin/llvm-nm" if os.path.exists("/opt/rocm/lib/llvm/bin/llvm-nm") else "nm"
    out = subprocess.run([nm, "-D", "--defined-only", build.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if ln.strip()}
    assert exported == declared, (sorted(exported - declared), sorted(declared - exported))
    # (the switches are compiled under RL_TUNING / RL_PHASE_PROFILE only: the export table above is the check)


def test_create_validates_its_configuration_and_states_its_own_limits():
    """width / height >= 3 is the reference's rule (Grid asserts it, grid.py:23-24).  The upper bounds are THIS build's: a world lives in
    one workgroup's LDS (width * height <= 4096 cells, a side <= 255), where the reference's Grid is unbounded -- rl_last_error() says so."""
    lib = _lib.lib()
    h = C.c_void_p()
    ok = _lib.Config(30, 30, 100, 2, 256, 1, 1, 0, 1, 0, 0)
    assert lib.rl_create(C.byref(ok), C.byref(h)) == 0
    acts = (C.c_int8 * 256)()
    assert lib.rl_step(h, acts, None, None, None) == -2          # RL_E_UNBOUND: rl_bind_state was not called
    assert b"rl_bind_state" in lib.rl_last_error()
    lib.rl_destroy(h)
    for bad in (dict(width=2), dict(height=2), dict(slot_cap=100), dict(slot_cap=128), dict(n_brains=0), dict(n_worlds=0),
                dict(width=100, height=100)):
        cfg = _lib.Config(30, 30, 100, 2, 256, 1, 1, 0, 1, 0, 0)
        for k, v in bad.items():
            setattr(cfg, k, v)
        assert lib.rl_create(C.byref(cfg), C.byref(h)) < 0, bad     # Grid asserts width/height >= 3 (grid.py:23-24)
        assert lib.rl_last_error()
        if bad == dict(width=100, height=100):
            assert b"LIMIT of this build" in lib.rl_last_error() and b"unbounded" in lib.rl_last_error()


def test_create_accepts_the_world_shapes_the_lds_budget_allows():
    """rl_create sizes the world's LDS (160 KB per workgroup; the stand-alone kernels' row-major observation planes -- the multi-tick
    launch pads them where that fits, rl_run.hip host_plane_stride): every shape the budget allows creates."""
    lib = _lib.lib()
    h = C.c_void_p()
    for w, hgt, agents, cap in ((3, 3, 2, 64), (30, 30, 100, 256), (64, 64, 100, 256), (64, 64, 500, 1024), (10, 255, 100, 256), (255, 10, 100, 256),
                                (28, 146, 200, 448), (40, 100, 300, 640), (63, 65, 60, 128), (30, 30, 400, 832)):
        cfg = _lib.Config(w, hgt, agents, 2, cap, 4, 1, 0, 1, 0, 0)
        assert lib.rl_create(C.byref(cfg), C.byref(h)) == 0, (w, hgt, agents, cap, lib.rl_last_error())
        lib.rl_destroy(h)


def test_philox_matches_oracle_and_known_answers():
    from oracle import oracle as orc
    lib = _lib.lib()
    out = (C.c_uint32 * 4)()
    lib.rl_philox(0, 0, 0, 0, 0, 0, C.byref(out))
    assert list(out) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    rng = np.random.RandomState(0)
    for _ in range(50):
        a = [int(x) for x in rng.randint(0, 2**31, size=6)]
        lib.rl_philox(a[0] * 7919 + 3, a[1] % 5, a[2], a[3], a[4] % 10, a[5], C.byref(out))
        assert list(out) == orc.philox(a[0] * 7919 + 3, a[1] % 5, a[2], a[3], a[4] % 10, a[5])


def _f16_planes_value(u16, shape):
    """[..., 2 planes, 64 lanes, 8] f16 (hi, lo) -> hi + lo, the scaled weights the two planes stand for."""
    parts = u16.view(np.float16).reshape(shape).astype(np.float64)
    return parts[..., 0, :, :] + parts[..., 1, :, :]


def _feature_of(t2, lane_half, r):
    return 32 * t2 + (r & 3) + 8 * (r >> 2) + 4 * lane_half


@pytest.mark.parametrize("name", ["DQN", "D3QN", "PERD3QN", "PPO"])
def test_weight_packing_keeps_every_weight_to_22_bits(name):
    """rl_policy_pack_weights: MFMA layers are stored as two f16 planes of the weight scaled by a power of two per output
    feature (hi + lo == scale * w to 22 bits of the row maximum), permuted into fragment order; the per-feature
    unscale factors and the biases follow in accumulator order as f32.  Every parameter must be recoverable, none
    duplicated, and unscale * scaled weight must give the weight back."""
    from oracle import oracle as orc
    lib = _lib.lib()
    kind = _lib.KIND_BY_METHOD[name]
    n = lib.rl_policy_n_params(kind)
    assert n == orc.lib().rlo_policy_n_params(kind)
    rng = np.random.RandomState(3)
    flat = (rng.uniform(0.25, 1.0, size=n) * rng.choice([-1.0, 1.0], size=n) * 2.0 ** rng.randint(-6, 3, size=n)).astype(np.float32)
    packed = np.zeros(lib.rl_policy_packed_floats(kind), np.float32)
    assert lib.rl_policy_pack_weights(kind, flat.ctypes.data_as(C.c_void_p), packed.ctypes.data_as(C.c_void_p)) == 0
    u16 = packed.view(np.uint16)
    # (input layer tiles, [(tin, tout, head outputs)...]) per kind, in packed order; parameters in state-dict order
    t1, branches = {"DQN": (4, [(4, 2, 8)]), "D3QN": (4, [(4, 4, 8), (4, 4, 1)]), "PERD3QN": (4, [(4, 4, 8), (4, 4, 1)]),
                    "PPO": (8, [(8, 8, 8)])}[name]
    off = [0]  # in 4-byte units
    par = [0]

    def take_params(count):
        v = flat[par[0]:par[0] + count]
        par[0] += count
        return v

    def check_mfma_layer(chunks, tout, n_in, k_of):
        """fragments [chunks][tout][2][64][8] + consts [tout][2][unscale 16 | bias 16]; k_of(chunk, lane_half, e) -> input index"""
        W = take_params(tout * 32 * n_in).reshape(tout * 32, n_in).astype(np.float64)
        b = take_params(tout * 32).astype(np.float64)
        cnt = chunks * tout * 2 * 64 * 4
        val = _f16_planes_value(u16[off[0] * 2:(off[0] + cnt) * 2], (chunks, tout, 2, 64, 8))
        off[0] += cnt
        consts = packed[off[0]:off[0] + tout * 64].reshape(tout, 2, 2, 16).astype(np.float64)  # [t2][half][unscale|bias][r]
        off[0] += tout * 64
        un = np.zeros(tout * 32); bias = np.zeros(tout * 32)
        for t2 in range(tout):
            for hh in range(2):
                for r in range(16):
                    un[_feature_of(t2, hh, r)] = consts[t2, hh, 0, r]
                    bias[_feature_of(t2, hh, r)] = consts[t2, hh, 1, r]
        assert np.array_equal(bias, b)
        assert np.all(np.log2(un) == np.round(np.log2(un)))  # powers of two
        seen = np.zeros_like(W, dtype=bool)
        for c in range(chunks):
            for t2 in range(tout):
                for lane in range(64):
                    o = 32 * t2 + (lane & 31)
                    for e in range(8):
                        k = k_of(c, lane >> 5, e)
                        if k is None or k >= n_in:
                            assert val[c, t2, lane, e] == 0
                            continue
                        assert not seen[o, k]
                        seen[o, k] = True
                        rowmax = np.abs(W[o]).max()
                        assert abs(val[c, t2, lane, e] * un[o] - W[o, k]) <= rowmax * 2.0 ** -21
        assert seen.all()

    check_mfma_layer(10, t1, 153, lambda c, hh, e: 16 * c + 8 * hh + e)
    for tin, tout, nout in branches:
        def k_hidden(c, hh, e):
            t, cc = divmod(c, 2)
            r = 8 * cc + e
            return 32 * t + (r & 3) + 8 * (r >> 2) + 4 * hh
        check_mfma_layer(2 * tin, tout, tin * 32, k_hidden)
        # head: fragments [tin*2][1][2][64][8] (rows >= nout zero), then unscale[8], bias[8]
        W = take_params(nout * tout * 32).reshape(nout, tout * 32).astype(np.float64)
        b = take_params(nout).astype(np.float64)
        cnt = tout * 2 * 2 * 64 * 4
        val = _f16_planes_value(u16[off[0] * 2:(off[0] + cnt) * 2], (tout * 2, 2, 64, 8))
        off[0] += cnt
        un, bias = packed[off[0]:off[0] + 8].astype(np.float64), packed[off[0] + 8:off[0] + 16].astype(np.float64)
        off[0] += 16
        assert np.array_equal(bias[:nout], b) and np.all(bias[nout:] == 0)
        for c in range(tout * 2):
            for lane in range(64):
                o = lane & 31
                for e in range(8):
                    k = k_hidden(c, lane >> 5, e)
                    if o >= nout:
                        assert val[c, lane, e] == 0
                    else:
                        assert abs(val[c, lane, e] * un[o] - W[o, k]) <= np.abs(W[o]).max() * 2.0 ** -21
    assert off[0] == len(packed)
    assert par[0] == n - (257 if name == "PPO" else 0)   # PPO's value head (fc_v, last in the state dict) is not evaluated when acting
    assert lib.rl_policy_pack_weights(9, flat.ctypes.data_as(C.c_void_p), packed.ctypes.data_as(C.c_void_p)) < 0


def test_device_worlds_refuses_to_run_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from reinlife_amd.worlds import DeviceWorlds
    with pytest.raises(_lib.ReinLifeHipError):
        DeviceWorlds(n_worlds=1)


def test_options_are_process_level_snapshotted_by_handles_and_never_read_from_the_environment_later(monkeypatch):
    """rl_set_option / rl_get_option (include/reinlife_hip.h "options"): values start from the environment once, a handle keeps the
    snapshot rl_create took, unknown names / values are errors.  (No launch reads the environment: see `grep getenv csrc/`.)"""
    from reinlife_amd import _lib
    lib = _lib.lib()
    get = lambda h, n: lib.rl_get_option(h, n.encode())  # noqa: E731
    try:
        assert get(None, "policy_variant") == 0 and get(None, "world_block") == 0      # auto: ONE arithmetic, block by world count
        _lib.set_option("policy_variant", "wave"); _lib.set_option("world_block", 512)
        monkeypatch.setenv("RL_WORLD_BLOCK", "256")                                     # later changes of the environment: not seen ...
        cfg = _lib.Config(30, 30, 100, 2, 256, 4, 1, 0, 1, 0, 7)
        h = C.c_void_p()
        assert lib.rl_create(C.byref(cfg), C.byref(h)) == 0
        assert (get(h, "policy_variant"), get(h, "world_block"), get(h, "world_generic"), get(h, "run_always")) == (2, 512, 0, 0)
        _lib.set_option("policy_variant", "pair"); _lib.set_option("world_block", 1024); _lib.set_option("run_always", 1)
        assert (get(h, "policy_variant"), get(h, "world_block"), get(h, "run_always")) == (2, 512, 0)   # the handle keeps its snapshot
        assert (get(None, "policy_variant"), get(None, "world_block"), get(None, "run_always")) == (4, 1024, 1)
        lib.rl_destroy(h)
        _lib.set_option("world_block", None)                                            # ... until somebody asks for the environment's value
        assert get(None, "world_block") == 256
        for name, value in (("policy_variant", "fast"), ("policy_variant", "nsplit"), ("world_block", "300"), ("policy_per_kind", "1"), ("no_such_option", "1")):   # ("nsplit" / policy_per_kind: the 4-wave tile of rounds 1-2 is gone)
            assert lib.rl_set_option(name.encode(), value.encode()) != 0 and lib.rl_last_error()
        assert get(None, "no_such_option") == -1
    finally:
        monkeypatch.delenv("RL_WORLD_BLOCK", raising=False)
        for name in ("policy_variant", "world_block", "world_generic", "run_always"):
            _lib.set_option(name, None)
    assert get(None, "policy_variant") == 0 and get(None, "world_block") == 0
    src = os.path.join(ROOT, "reinlife_amd", "csrc")
    users = [f for f in sorted(os.listdir(src)) if "getenv(" in open(os.path.join(src, f)).read()]
    assert users == ["rl_capi.hip"], users   # the one place that reads the environment: options_from_env


def test_the_header_says_what_the_product_library_runs_and_the_library_agrees():
    """include/reinlife_hip.h's "Supported" paragraph of rl_run against rl_run_supported() of the PRODUCT library (the
    header named 256- / 1024-thread instantiations the product refuses): 512-thread workgroups up to 768 worlds, any world count with
    run_always, the forced 256- / 1024-thread instantiations answered 0 with an error that names the tuning build -- and the library's
    symbol table holds k_run<512, ...> and nothing else of that family (reference loop: Helpers/trainer.py:85-99)."""
    import subprocess
    from reinlife_amd import build
    hdr = open(os.path.join(ROOT, "include", "reinlife_hip.h")).read()
    para = hdr[hdr.index(" * Supported (rl_run_supported() != 0)"):]
    para = para[:para.index("*/")]
    assert "slot_cap <= 512" in para and "n_worlds <= 768" in para and '"run_always"' in para
    assert "tuning" in para and "libreinlife_hip_tune.so" in para and "256- and 1024-thread" in para
    lib = _lib.lib()
    nm = subprocess.run(["nm", "-C", build.LIB_PATH], capture_output=True, text=True, check=True).stdout
    blocks = set(re.findall(r"k_run<(\d+),", nm))
    assert blocks == {"512"}, blocks
    brains = (_lib.Brain * 2)(_lib.Brain(_lib.PERD3QN, 0.0, None), _lib.Brain(_lib.PERD3QN, 0.0, None))
    mixed = (_lib.Brain * 2)(_lib.Brain(_lib.PPO, 0.0, None), _lib.Brain(_lib.PERD3QN, 0.0, None))

    def supported(n_worlds, which=brains, **opts):
        for k, v in opts.items():
            _lib.set_option(k, v)
        try:
            cfg = _lib.Config(30, 30, 100, 2, 256, n_worlds, 1, 0, 1, 0, 7)
            h = C.c_void_p()
            assert lib.rl_create(C.byref(cfg), C.byref(h)) == 0
            r = lib.rl_run_supported(h, which, 2)
            err = lib.rl_last_error().decode()
            lib.rl_destroy(h)
            return r, err
        finally:
            for k in opts:
                _lib.set_option(k, None)
    assert supported(256)[0] == 1 and supported(768)[0] == 1 and supported(256, mixed)[0] == 1
    r, err = supported(1024)
    assert r == 0 and "tuning-build" in err and "256-thread" in err
    assert supported(1024, run_always=1)[0] == 1 and supported(4096, mixed, run_always=1)[0] == 1     # ... the 512-thread kernel, several workgroups per CU in turn
    for block in (256, 1024):
        r, err = supported(256, world_block=block)
        assert r == 0 and "tuning-build" in err and ("%d-thread" % block) in err


def test_builds_are_serialised_by_a_lock_and_files_are_moved_into_place():
    """reinlife_amd/build.py: N ranks that find the library stale together must not compile into the same files while
    others dlopen them.  build() runs under an flock on lib/.build.lock (checked here with two processes: the second waits for the first)
    and writes every object and the library under a temporary name before os.replace() (checked in the source: no compiler or linker
    command line names a final path as its output)."""
    import subprocess
    import sys
    import time
    from reinlife_amd import build
    holder = subprocess.Popen([sys.executable, "-c",
                               "import sys, time; sys.path.insert(0, %r)\nfrom reinlife_amd import build\nwith build._BuildLock():\n    print('held', flush=True); time.sleep(1.5)\n" % ROOT],
                              stdout=subprocess.PIPE, text=True)
    assert holder.stdout.readline().strip() == "held"
    t0 = time.perf_counter()
    with build._BuildLock():
        waited = time.perf_counter() - t0
    holder.wait()
    assert waited > 0.8, "the lock did not wait for its holder (%.2f s)" % waited
    t0 = time.perf_counter()
    with build._BuildLock():
        assert time.perf_counter() - t0 < 0.5            # free again
    src = open(os.path.join(ROOT, "reinlife_amd", "build.py")).read()
    assert '"-o", obj + tmp_tag]' in src and '"-o", LIB_PATH + tmp_tag]' in src and src.count("os.replace(obj + tmp_tag, obj)") == 1 and src.count("os.replace(LIB_PATH + tmp_tag, LIB_PATH)") == 1
    assert build.library_is_current()                    # (and the shipped stamps are those of the shipped sources)


def test_measurements_bench_reports_from_files_are_those_of_the_shipped_kernel_sources():
    """bench.py takes three figures from tracked files -- PMC HBM traffic of the multi-tick launch (profiles/run_traffic.json), of the two
    stand-alone kernels (profiles/tick_traffic.json) and the latency-bound model (profiles/latency_model.json) -- and drops each when the
    kernel sources no longer hash to its stamp.  The shipped tree must carry stamps of the shipped sources (tick_traffic.json
    had been three rounds stale, its fields null in every driver line): after a kernel change, re-take them (tools/final_round6.sh)."""
    import json
    from reinlife_amd import build
    now = build.source_hash()
    for name, keys in (("run_traffic.json", ("hbm_bytes_per_tick",)), ("tick_traffic.json", ("hbm_bytes_per_launch", "policy_hbm_bytes_per_launch")),
                       ("latency_model.json", ("workloads",))):
        d = json.load(open(os.path.join(ROOT, "profiles", name)))
        assert d["kernel_src_sha16"] == now, "%s was measured on kernel sources %s, the tree holds %s" % (name, d["kernel_src_sha16"], now)
        for k in keys:
            assert d.get(k), (name, k)
    m = json.load(open(os.path.join(ROOT, "profiles", "latency_model.json")))["workloads"]
    for wl in ("c4", "c5"):
        w = m[wl]
        assert 0.2 < w["frac_of_latency_bound"] < 0.8
        assert w["policy_floor_counts"] == max(w["mfma_slowest_simd"]["counts"], w["weight_stream"]["counts"])
        assert abs(w["weight_stream"]["counts"] - w["weight_stream"]["kb_per_world_tick"] * 1024 / w["weight_stream"]["cu_load_bytes_per_clock"]) < 1.0
        parts = w["policy_floor_counts"] + w["barriers"]["counts"] + sum(w["one_wave_sections_counts"].values())
        assert abs(parts - w["bound_counts"]) < 1.0 and abs(w["bound_counts"] / w["stamped_tick_counts"] - w["frac_of_latency_bound"]) < 1e-3


def test_the_latency_bound_of_the_bench_line_reproduces_from_the_tracked_profiles(tmp_path):
    """roofline.latency_bound_us: profiles/latency_model.json is what tools/latency_bound.py makes of the tracked unit costs
    (profiles/r06_ubench.txt) and stamps (profiles/r06_stamps.txt) -- run it again and compare ('reproducible from profiles/')."""
    import json
    import subprocess
    import sys
    out = tmp_path / "model.json"
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "latency_bound.py"), os.path.join(ROOT, "profiles", "r06_ubench.txt"),
                    os.path.join(ROOT, "profiles", "r06_stamps.txt"), str(out)], check=True, capture_output=True, cwd=ROOT)
    new, kept = json.load(open(out)), json.load(open(os.path.join(ROOT, "profiles", "latency_model.json")))
    assert new["workloads"] == kept["workloads"] and new["unit_costs_counts"] == kept["unit_costs_counts"]
    u = kept["unit_costs_counts"]
    assert 31.5 < u["mfma_pipe_counts"] < 32.5 and 40 < u["barrier_counts"] < 70 and 7 < u["valu_dependent_counts"] < 10
    c4 = kept["workloads"]["c4"]
    assert c4["mfma_slowest_simd"]["mfmas"] == 360 and abs(c4["mfma_slowest_simd"]["counts"] - 360 * u["mfma_pipe_counts"]) < 1
