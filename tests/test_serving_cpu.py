"""CPU: the payload codec of doudizhu-rl_amd/serving.py (pure host code): state rows -> the reference's serving payloads
(server/client.py:6-25) -> state rows, on oracle-played tables."""
import importlib
import json

import numpy as np


def test_state_to_payloads_round_trip(oracle):
    serving = importlib.import_module("doudizhu-rl_amd.serving")
    T = 200
    ref = oracle.OracleEnv(T, seed=5)
    ref.reset()
    for _ in range(23):
        ref.legal(); ref.step(oracle.STEP_RANDOM)
    payloads = serving.state_to_payloads(ref.state)
    assert len(payloads) == T and set(payloads[0]) == {"role_id", "cur_cards", "history", "left", "last_taken"}
    full = ref.state.reshape(T, 11, 16)
    back = serving.payloads_to_state(json.loads(json.dumps(payloads)))     # JSON turns the role keys into strings
    for t in range(T):
        p, role = payloads[t], int(full[t, 10, 0])
        assert p["role_id"] == role and all(3 <= c <= 17 for c in p["cur_cards"]) and p["cur_cards"] == sorted(p["cur_cards"])
        assert len(p["cur_cards"]) == p["left"][role] == int(full[t, role, 15])
        assert sum(len(p["history"][r]) + p["left"][r] for r in range(3)) == 54     # every card is somewhere
        assert np.array_equal(back[t, role], full[t, role])                         # the actor's hand row, incl. its size
        assert np.array_equal(back[t, 3:10, :15], full[t, 3:10, :15])               # history, recent handouts, taken
        assert [back[t, r, 15] for r in range(3)] == [full[t, r, 15] for r in range(3)]
        assert back[t, 10, 0] == role
