"""GPU (-m gpu): BatchedEnv (through the C ABI) + dqn_glue.TransitionAssembler against fixture G11 = every
agent.perceive() call of the REFERENCE's own Game.play (tests/golden/gen_game.py): the tables of a scenario are stepped
in lock step on the device -- legal lists, `face`, the rule agent for the roles without a network (Env.step_auto), the
step, masked re-deals -- and the transitions the assembler closes must be the reference's, call for call
(replicate_reference_quirk=True) or minus the stale cross-episode ones (default)."""
import importlib

import numpy as np
import pytest
import torch

import game_policy as gp

pytestmark = pytest.mark.gpu


class DeviceBackend:
    def __init__(self, pkg, T, seed):
        self.pkg, self.env = pkg, pkg.BatchedEnv(T, seed=seed, device="cuda:0")

    def reset(self, mask):
        self.env.reset(None if mask is None else torch.from_numpy(np.asarray(mask)))

    def roles(self):
        return self.env.role.cpu().numpy()

    def legal(self):
        off, rows, ids = self.env.legal()
        n = int(off[-1])
        return off.cpu().numpy(), rows[:max(n, 1)].cpu().numpy(), ids[:max(n, 1)].cpu().numpy()

    def observe(self, variant):
        return self.env.observe(variant).cpu().numpy()

    def auto_choose(self):
        return self.env.auto_choose(0b111).cpu().numpy()

    def step_ids(self, ids):
        done, r, _ = self.env.step(torch.from_numpy(ids), self.pkg.STEP_IDS, auto_reset=False)
        return done.cpu().numpy(), r.cpu().numpy()


@pytest.mark.parametrize("quirk", [True, False])
@pytest.mark.parametrize("name", list(gp.SCENARIOS))
def test_device_env_and_assembler_reproduce_the_reference_game_loop(golden, name, quirk):
    pkg = importlib.import_module("doudizhu-rl_amd")
    glue = importlib.import_module("doudizhu-rl_amd.dqn_glue")
    g = golden("game_play.npz")
    T, E = int(g["tables"]), int(g["episodes"])
    sc = gp.SCENARIOS[name]
    be = DeviceBackend(pkg, T, sc["seed"])
    got, wins = gp.replay_scenario(be, glue, name, T, E, quirk)
    want = gp.expected_from_fixture(g, name, T, quirk, sc["train"])
    n = 0
    for t in range(T):
        assert len(got[t]) == len(want[t]), (t, len(got[t]), len(want[t]))
        for k, (a, b) in enumerate(zip(got[t], want[t])):
            assert a[:5] == b[:5], (t, k, a[:5], b[:5])
            assert np.array_equal(a[5], b[5]) and np.array_equal(a[6], b[6]), (t, k)
            n += 1
    assert n > 900 and np.array_equal(wins, g[f"{name}.wins"])
    assert be.env.status() == 0
