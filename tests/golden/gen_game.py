#!/usr/bin/env python3
"""Generate tests/golden/game_play.npz (fixture G11) -- runs ONLY in the build container.

  G11  every agent.perceive(s0, a0, reward, s1, a1, done) call the REFERENCE's own training loop makes
       (game.py:90-181: Game.play -> lord_turn / down_turn / up_turn -> step / feedback), in call order, for seeded
       episodes played back to back by ONE Game object per table -- including what the reference does between episodes
       (it never clears *_s0 / *_a0, game.py:39-40,132-137).

game.py is imported from /root/reference and executed here.  What had to be supplied (the fixture is data: inputs and
the outputs of the reference's code):
  * config.get_logger (config.py:34-47) is replaced by a function returning a null logger BEFORE game.py is imported:
    the original makes directories and a log file under the reference tree (game.py:7 calls it at import), and this
    script must never write there.
  * env_cls: the native modules `env` / `r` are absent (envi.py:10-13), so the Env handed to Game() is a one-table view
    of this repo's CPU oracle with the members game.py touches (face, valid_actions, step_manual, step_auto, reset,
    prepare; game.py:95-106,121-125,170-171) and the return conventions of envi.py:63-77.  Deals, the prob planes and
    step_auto's decomposers are this repo's specs (PARITY UNPINNED); what G11 pins is everything ABOVE the
    env: who gets feedback when, with which (s0, a0), which reward sign, which s1 / a1, in which order.
  * dqns_dict: recording agents with the three methods game.py calls (e_greedy_action, greedy_action, perceive;
    dqn.py:21-71) and deterministic choices (tests/game_policy.py) so that a batched replay needs no RNG.
"""
import logging
import os
import sys
import zlib

import numpy as np
import torch

sys.dont_write_bytecode = True  # never write into /root/reference
REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
sys.path.insert(0, REF)

from oracle import oracle  # noqa: E402
import game_policy as gp  # noqa: E402

import config as ref_config  # noqa: E402  (reference: constants only)

ref_config.get_logger = lambda: ("0000_0000", logging.getLogger("ddz_null"), os.devnull)
import game as ref_game  # noqa: E402  (reference: the training loop itself)

TABLES, EPISODES = 6, 12


def crc(x):
    return zlib.crc32(np.ascontiguousarray(x, np.float32).tobytes())


def make_env_cls(variant, seed, gid):
    class OracleBackedEnv:
        """the members of envi.Env that game.py uses, on one oracle table"""

        def __init__(self, debug=False, seed_=None, **_):
            self.o = oracle.OracleEnv(1, seed=seed, gid_base=gid)
            self.plies = []

        def reset(self):        # envi.py:30-36: adapter bookkeeping only (lives in the oracle's state rows)
            pass

        def prepare(self):      # native: shuffle + deal; the lord moves first (game.py:171-173)
            self.o.reset()

        @property
        def face(self):         # envi.py:87-96 (+ variants): f32 [P,15,4]
            return torch.from_numpy(self.o.observe(variant)[0].copy())

        def valid_actions(self, tensor=True):   # envi.py:98-116
            _, rows, _ = self.o.legal()
            return torch.from_numpy(oracle.rows_to_onehot(rows))

        def step_manual(self, onehot):          # envi.py:63-70 -> (r, done, _)
            row = np.zeros((1, 16), np.int8)
            row[0, :15] = np.asarray(onehot).reshape(15, 4).sum(1)
            self.o.legal()
            role = int(self.o.field(10)[0, 0])
            done, r, illegal, _ = self.o.step(oracle.STEP_ROWS, row, auto_reset=False)
            assert not illegal[0]
            self.plies.append((role, oracle.lookup(row[0, :15])))
            return int(r[0]), bool(done[0]), None

        def step_auto(self):                    # envi.py:72-77 -> (cards, r, _); game.py:106 reads r as `done`
            ids = self.o.auto_choose(0b111)
            self.o.legal()
            role = int(self.o.field(10)[0, 0])
            done, r, illegal, _ = self.o.step(oracle.STEP_IDS, ids, auto_reset=False)
            assert not illegal[0] and bool(done[0]) == bool(r[0])
            self.plies.append((role, int(ids[0])))
            return None, int(r[0]), None

    def ctor(debug=False, seed=None):
        return OracleBackedEnv(debug=debug, seed_=seed)
    return ctor


class RecordingAgent:
    """what game.py needs of DQNFirst (dqn.py:10-71), with the deterministic choices of tests/game_policy.py"""

    def __init__(self, role, log):
        self.role, self.log = role, log

    @staticmethod
    def _sizes(face):
        return int(round(float(face[0].sum()))), int(round(float(face[1].sum())))

    def e_greedy_action(self, face, actions):
        h, tk = self._sizes(face)
        return actions[gp.e_greedy_index(h, tk, actions.shape[0])]

    def greedy_action(self, face, actions):
        h, tk = self._sizes(face)
        return actions[gp.greedy_index(h, tk, actions.shape[0])]

    def perceive(self, s0, a0, reward, s1, a1, done):
        self.log.append((gp.ROLE_ID[self.role], float(reward), bool(done), crc(s0), crc(s1),
                         np.asarray(a0).reshape(15, 4).sum(1).astype(np.int8),
                         np.asarray(a1).reshape(15, 4).sum(1).astype(np.int8)))
        return None


def run_scenario(name, sc):
    out = {k: [] for k in ("table", "role", "reward", "done", "s0_crc", "s1_crc", "a0", "a1", "episode")}
    plies, wins = [], []
    for t in range(TABLES):
        log = []
        nets = {role: True for role in sc["ai"]}                      # truthy placeholders: net_cls is only handed on
        dqns = {role: (lambda net, role=role: RecordingAgent(role, log)) for role in sc["ai"]}
        train = {role: role in sc["train"] for role in ("lord", "down", "up")}
        g = ref_game.Game(make_env_cls(sc["variant"], sc["seed"], t), nets, dqns, train_dict=train)
        for ep in range(EPISODES):
            n0 = len(log)
            g.play()                                                   # the reference's loop
            for rec in log[n0:]:
                out["table"].append(t); out["episode"].append(ep)
                for k, v in zip(("role", "reward", "done", "s0_crc", "s1_crc", "a0", "a1"), rec):
                    out[k].append(v)
        plies.append(np.array(g.env.plies, np.int32))
        wins.append((g.up_total_wins, g.lord_total_wins, g.down_total_wins))
    res = {f"{name}.{k}": np.array(v) for k, v in out.items()}
    res[f"{name}.wins"] = np.array(wins, np.int32)                    # per table: up / lord / down (game.py:142,155,166)
    res[f"{name}.ply_table"] = np.concatenate([np.full(len(p), t, np.int32) for t, p in enumerate(plies)])
    res[f"{name}.ply_role"] = np.concatenate([p[:, 0] for p in plies])
    res[f"{name}.ply_action"] = np.concatenate([p[:, 1] for p in plies])
    n, nd = len(out["role"]), int(np.sum(out["done"]))
    print(f"{name}: {TABLES} tables x {EPISODES} episodes, {sum(len(p) for p in plies)} plies, {n} perceive() calls "
          f"({nd} terminal), wins up/lord/down {np.sum(wins, 0).tolist()}")
    return res


def main():
    data = {"tables": np.int32(TABLES), "episodes": np.int32(EPISODES)}
    for name, sc in gp.SCENARIOS.items():
        data.update(run_scenario(name, sc))
    np.savez_compressed(os.path.join(HERE, "game_play.npz"), **data)
    print("wrote game_play.npz")


if __name__ == "__main__":
    main()
