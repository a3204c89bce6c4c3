"""Test helper (GPU suites): the policy half of a multi-tick launch checked against the oracle AT THE BENCHED SIZE.

The 256-world oracle runs feed the launch's actions to the oracle, so a policy tile that read a stale or foreign observation row would
still produce a VALID action the oracle then follows.  PolicyCheck closes that: it snapshots the oracle's Agent.state rows (obs2) and
the keys of the draws before a launch and afterwards compares, for every live agent of every world,
  * the launch's outputs (Q values / PPO probabilities, rl_run_opts.policy_out) with an f32 forward of the ORACLE's rows (batched sgemm
    of oracle/cpu_bench.py, itself pinned to the C oracle's scalar forward on a sample; D3QN.py:161-173, PPO.py:101-106,164-169), atol 1e-5,
    and the chosen actions with the oracle's selection rule applied to those outputs (exact: epsilon-greedy / inverse-CDF sample);
  * where the launch writes no outputs (the TRAIN 0 instantiation bench.py times): the chosen actions with the selection rule applied
    to the f32 forward -- greedy rows wherever the f32 top-2 gap exceeds 1e-5, sampled rows (PPO) up to `ppo_slack` rows per call whose
    uniform draw may sit within the 2e-7 the two probability vectors differ by."""
import numpy as np


class PolicyCheck:
    def __init__(self, names, weights, eps, ppo_slack=1):
        from oracle import cpu_bench
        self.names, self.weights, self.eps, self.ppo_slack = names, weights, eps, ppo_slack
        self.layers = [cpu_bench.unpack(n, w) for n, w in zip(names, weights)]
        self.rows = self.q_rows = 0
        self.max_dq = 0.0
        self.ppo_mismatches = 0

    def before(self, ow):
        """Snapshot what the policy is about to read: call right before the launch of the tick."""
        self.snap = (ow.obs2.copy(), ow.s["n_agents"].copy(), ow.s["a_brain"].copy(), ow.s["tick"].copy(), ow.s["epoch"].copy(), ow.cfg)

    def after(self, actions, q=None, what=""):
        from oracle import cpu_bench, oracle as orc
        obs, n, brain, tick, epoch, cfg = self.snap
        live = np.arange(obs.shape[1])[None, :] < n[:, None]
        for b, name in enumerate(self.names):
            ws, ks = np.nonzero(live & (brain == b))
            if len(ws) == 0:
                continue
            kind = orc.KIND_BY_NAME[name]
            ref = cpu_bench.forward(name, self.layers[b], obs[ws, ks])
            sub = slice(0, 300)   # (the sgemm forward is itself pinned to the oracle's scalar forward)
            np.testing.assert_allclose(ref[sub], orc.policy_forward(kind, self.weights[b], obs[ws[sub], ks[sub]]), rtol=0, atol=2e-6)
            got = actions[ws, ks]
            self.rows += len(ws)
            if q is not None:
                out = q[ws, ks]
                np.testing.assert_allclose(out, ref, rtol=0, atol=1e-5, err_msg="%s %s outputs vs the f32 forward of the oracle's rows" % (what, name))
                self.max_dq = max(self.max_dq, float(np.abs(out - ref).max()))
                self.q_rows += len(ws)
                want = orc.select_actions(cfg, kind, out, ws, ks, tick, epoch, self.eps[b])
                assert np.array_equal(got, want), "%s %s: actions are not the selection rule applied to the launch's own outputs" % (what, name)
            want = orc.select_actions(cfg, kind, ref, ws, ks, tick, epoch, self.eps[b])
            bad = got != want
            if name == "PPO":
                self.ppo_mismatches += int(bad.sum())
                assert int(bad.sum()) <= self.ppo_slack, "%s PPO: %d sampled actions differ from the rule on the f32 probabilities" % (what, int(bad.sum()))
            elif bad.any():   # an explored (random) action is the same draw on both sides; a greedy one may flip only inside the tolerance
                srt = np.sort(ref, axis=1)
                gap = srt[:, -1] - srt[:, -2]
                assert float(gap[bad].max()) < 1e-5, "%s %s: an action differs where the f32 top-2 gap is %.3g" % (what, name, float(gap[bad].max()))
