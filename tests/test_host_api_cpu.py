"""CPU: host-side logic of the Python mirror -- exact Environment.reset replay, state-dict wire contract, API surface."""
import inspect
import json
import os

import numpy as np

import golden_io


def test_host_reset_replays_the_reference_reset_under_the_same_numpy_seed():
    from reinlife_amd.World.environment import host_reset
    from oracle import oracle as orc
    g = np.load(os.path.join(golden_io.GOLDEN_DIR, "resets.npz"))
    for seed, nb, w, h in ((1, 2, 30, 30), (2, 3, 30, 30), (3, 5, 30, 20), (4, 2, 7, 5)):
        key = "s%d_b%d_%dx%d" % (seed, nb, w, h)
        np.random.seed(seed)
        snap = host_reset(w, h, nb)
        assert np.array_equal(snap["cell_type"], g[key + "_cell_type"]), key
        for k in ("i", "j", "gene", "uid"):
            assert np.array_equal(snap[k], g[key + "_" + k]), (key, k)
        ow = orc.OracleWorlds(1, w, h, 100, nb)   # and the first observation, through the oracle
        ow.load_world(0, snap)
        assert np.array_equal(ow.observe()[0, : len(snap["i"])], g[key + "_obs"]), key


def test_brains_keep_the_reference_state_dict_contract_and_constructor_keywords():
    from reinlife_amd import Models, _lib
    keys = json.load(open(os.path.join(golden_io.GOLDEN_DIR, "state_dict_keys.json")))
    nets = {"DQN": Models.DQN().agent, "D3QN": Models.D3QN().eval_net, "PERD3QN": Models.PERD3QN().eval_net, "PPO": Models.PPO().model}
    for name, net in nets.items():
        got = [[k, list(v.shape)] for k, v in net.state_dict().items()]
        assert got == keys[name], name
    b = Models.PERD3QN(training=False)
    assert b.method == "PERD3QN" and b.epsilon == 0 and b.state_dict_flat().size == _lib.lib().rl_policy_n_params(_lib.PERD3QN)
    assert list(inspect.signature(Models.DQN.__init__).parameters)[1:] == ["input_dim", "output_dim", "max_epi", "learning_rate", "train_freq", "load_model", "training"]
    assert list(inspect.signature(Models.PPO.__init__).parameters)[1:] == ["input_dim", "output_dim", "learning_rate", "gamma", "lmbda", "eps_clip", "k_epoch", "train_freq", "load_model"]


def test_trainer_tester_environment_signatures_match_the_reference():
    import reinlife_amd
    tr = list(inspect.signature(reinlife_amd.trainer).parameters)
    assert tr[:15] == ["brains", "n_episodes", "width", "height", "visualize_results", "google_colab", "update_interval",
                       "print_results", "max_agents", "render", "static_families", "training", "save", "limit_reproduction",
                       "incentivize_killing"]
    te = list(inspect.signature(reinlife_amd.tester).parameters)
    assert te[:8] == ["brains", "width", "height", "max_agents", "pastel_colors", "static_families", "limit_reproduction", "fps"]
    en = list(inspect.signature(reinlife_amd.Environment.__init__).parameters)
    assert en[1:16] == ["width", "height", "brains", "grid_size", "max_agents", "update_interval", "print_results",
                        "static_families", "interactive_results", "google_colab", "training", "save", "pastel_colors",
                        "limit_reproduction", "incentivize_killing"]
    d = inspect.signature(reinlife_amd.trainer).parameters
    assert d["n_episodes"].default == 10_000 and d["max_agents"].default == 100 and d["save"].default is True


def test_saver_writes_the_reference_layout_and_loadable_state_dicts(tmp_path, monkeypatch):
    """saver.py:58-194 layout; the .pt files carry the reference's state-dict keys, parameters_*.json the reference's scalar
    brain attributes (tests/golden/brain_param_keys.json, recorded from the reference's brains)."""
    import torch
    from reinlife_amd import Models
    from reinlife_amd.Helpers.saver import SavedAgent, Saver
    monkeypatch.chdir(tmp_path)
    brains = [Models.PERD3QN(), Models.DQN(max_epi=100), Models.PPO(), Models.D3QN()]
    results = {"Avg Population Size": {0: [1.5]}, "Avg Number of Populations": [2.0]}
    exp = Saver("experiments").save([SavedAgent(g, b) for g, b in enumerate(brains)], True, results, {"Width": 30})
    exp2 = Saver("experiments").save([SavedAgent(g, b) for g, b in enumerate(brains[:2])], False, results, {"Width": 30})
    assert exp.endswith("_V1") and exp2.endswith("_V2")
    keys = json.load(open(os.path.join(golden_io.GOLDEN_DIR, "state_dict_keys.json")))
    want_params = json.load(open(os.path.join(golden_io.GOLDEN_DIR, "brain_param_keys.json")))
    for g, b in enumerate(brains):
        sd = torch.load(os.path.join(exp, b.method, "brain_gene_%d.pt" % g))
        assert [[k, list(v.shape)] for k, v in sd.items()] == keys[b.method]
        fresh = type(b)(load_model=os.path.join(exp, b.method, "brain_gene_%d.pt" % g))  # load_model= round trip
        assert all(torch.equal(x, y) for x, y in zip(fresh._net().state_dict().values(), b._net().state_dict().values()))
        params = json.load(open(os.path.join(exp, b.method, "parameters_gene_%d.json" % g)))
        for k, v in want_params[b.method].items():
            assert k in params and (params[k] == v or k in ("load_model",)), (b.method, k, params.get(k), v)
    assert os.path.exists(os.path.join(exp2, "PERD3QN", "brain_1.pt")) and os.path.exists(os.path.join(exp2, "DQN", "parameters_1.json"))
    assert json.load(open(os.path.join(exp, "results.json")))["Avg Number of Populations"] == [2.0]


def test_packed_weights_cache_notices_every_way_the_weights_can_change(monkeypatch):
    """Models.*.packed_weights() repacks exactly when the weights changed: in-place updates, load_state_dict, a Parameter object swapped
    for a new one, a replaced sub-module, a replaced network -- and not otherwise (the key used to cache the tensor LIST)."""
    import torch
    from reinlife_amd import Models, worlds
    calls = []
    monkeypatch.setattr(worlds, "pack_brain_weights", lambda kind, flat, device="cuda:0": calls.append(float(flat.sum())) or object())
    b = Models.PERD3QN()
    p0 = b.packed_weights("cuda:0")
    assert b.packed_weights("cuda:0") is p0 and len(calls) == 1                      # unchanged: the cached pack
    with torch.no_grad():
        b.eval_net.fc.weight.add_(1.0)                                               # in place (an optimizer step)
    assert b.packed_weights("cuda:0") is not p0 and len(calls) == 2
    b.eval_net.load_state_dict(b.target_net.state_dict())                            # load_state_dict copies in place
    b.packed_weights("cuda:0"); assert len(calls) == 3
    b.eval_net.adv_fc1.weight = torch.nn.Parameter(torch.zeros_like(b.eval_net.adv_fc1.weight))   # a NEW Parameter object in the slot
    b.packed_weights("cuda:0"); assert len(calls) == 4 and calls[-1] != calls[-2]
    b.eval_net.value_fc2 = torch.nn.Linear(128, 1)                                   # a replaced sub-module
    b.packed_weights("cuda:0"); assert len(calls) == 5
    b.eval_net = type(b.eval_net)(153, 8)                                            # a replaced network
    b.packed_weights("cuda:0"); assert len(calls) == 6
    b.packed_weights("cuda:0"); assert len(calls) == 6
    b.packed_weights("cuda:1"); assert len(calls) == 7                               # another device: its own pack


def test_two_tapes_prepared_before_either_is_used_hold_two_different_draws():
    """DeviceWorlds.make_tape (worlds.TapeRing): every call returns a fresh struct over its own device buffer -- a caller that prepares the
    step's and the update's tape before launching either keeps both (one aliased buffer silently replayed the second
    tape on both).  Here on CPU tensors, the tapes read back through the struct's own pointers, as a kernel would."""
    import ctypes as C
    import numpy as np
    from reinlife_amd import _lib
    from reinlife_amd.worlds import TapeRing
    R, cap = 3, 64
    ring = TapeRing(R, cap, "cpu")
    rng = np.random.RandomState(5)

    def draws():
        return [dict(food_k=rng.randint(0, 900, _lib.FOOD_TRIES), food_u=rng.rand(_lib.FOOD_TRIES), repro_u=rng.rand(cap), birth_k=rng.randint(0, 900, cap + 1),
                     produce_u=rng.rand(), produce_choice=rng.randint(0, 10)) for _ in range(R)]

    def read(tape, name, dt, shape):
        n = int(np.prod(shape))
        return np.ctypeslib.as_array(C.cast(getattr(tape, name), C.POINTER(np.ctypeslib.as_ctypes_type(dt))), (n,)).reshape(shape).copy()

    made = [(draws(),) for _ in range(TapeRing.SLOTS)]
    tapes = [ring.make(d[0]) for d in made]                       # every tape of the ring prepared before any is read
    assert len({t.food_k for t in tapes}) == TapeRing.SLOTS and len({id(t) for t in tapes}) == TapeRing.SLOTS
    for (d,), t in zip(made, tapes):
        assert np.array_equal(read(t, "food_k", np.int32, (R, _lib.FOOD_TRIES)), np.stack([w["food_k"] for w in d]))
        assert np.array_equal(read(t, "repro_u", np.float64, (R, cap)), np.stack([w["repro_u"] for w in d]))
        assert np.array_equal(read(t, "birth_k", np.int32, (R, cap + 1)), np.stack([w["birth_k"] for w in d]))
        assert np.array_equal(read(t, "produce_u", np.float64, (R,)), np.array([w["produce_u"] for w in d]))
        assert np.array_equal(read(t, "produce_choice", np.int32, (R,)), np.array([w["produce_choice"] for w in d]))
    # the ring wraps: the fifth tape lives where the first one did (documented lifetime), with its own draws; a short list leaves zeros behind
    fifth = ring.make(draws()[:1])
    assert fifth.food_k == tapes[0].food_k
    assert not read(fifth, "food_u", np.float64, (R, _lib.FOOD_TRIES))[1:].any()
