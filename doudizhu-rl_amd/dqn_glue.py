"""Batched transition assembly for the DQN loop of the reference.

The reference closes the transition of role X when X's next observation appears, or at the
terminal ply (game.py:109-167):
  * X acts in state s0 with action a0 (game.py:95-104);
  * the next time X is to move, feedback(X, done=False) stores (s0, a0, 0, s1 = face now,
    a1 = greedy action now, False) (game.py:109-127, called from :131-133, :146-148, :158-160);
  * when any role empties its hand every role with a pending (s0, a0) gets
    (s0, a0, +/-reward_dict[role], s1 = face after the terminal ply, a1 = zeros[15,4], True)
    (game.py:113-123, :134-141, :149-155, :161-167): winners +reward, losers -reward; lord and
    farmers are opposite sides, the two farmers win and lose together.
Here that bookkeeping is done for T tables at once with tensor ops (any device); pending
slots are cleared at the terminal ply (the reference never clears `*_a0` between episodes,
SURVEY.md appendix A "quirks": not replicated).

Usage per lock-step iteration (all tensors [T, ...]):
    closed = asm.before_step(role, face, chosen_onehot, greedy_onehot)
    ... env.step(auto_reset=False) ...; terminal_face = env.observe(variant)
    ended = asm.after_step(role, done, r, terminal_face); env.reset(mask=done)
Both calls return a dict of the transitions they closed (s0, a0, reward, s1, a1, done,
table, role) ready to append to a replay buffer.
"""
import torch

REWARD_DICT = {"up": 50.0, "lord": 100.0, "down": 50.0}  # game.py:13-14; role ids 0 up, 1 lord, 2 down


class TransitionAssembler:
    def __init__(self, n_tables, planes, device, reward_dict=None):
        rd = dict(REWARD_DICT if reward_dict is None else reward_dict)
        self.T, self.P, self.device = int(n_tables), int(planes), torch.device(device)
        self.reward = torch.tensor([rd["up"], rd["lord"], rd["down"]], dtype=torch.float32, device=self.device)
        self.s0 = torch.zeros((self.T, 3, self.P, 15, 4), dtype=torch.float32, device=self.device)
        self.a0 = torch.zeros((self.T, 3, 15, 4), dtype=torch.float32, device=self.device)
        self.pending = torch.zeros((self.T, 3), dtype=torch.bool, device=self.device)

    @staticmethod
    def _pack(s0, a0, reward, s1, a1, done, table, role):
        return {"s0": s0, "a0": a0, "reward": reward, "s1": s1, "a1": a1, "done": done, "table": table,
                "role": role}

    def before_step(self, role, face, chosen, greedy, active=None):
        """role int[T] actor of each table, face f32[T,P,15,4] its observation, chosen / greedy
        f32[T,15,4] the action it is about to play and the greedy action (a1 of the closing
        transition, game.py:125).  Closes the actor's previous transition, opens a new one."""
        role = role.to(self.device).long()
        ar = torch.arange(self.T, device=self.device)
        if active is None:
            active = torch.ones(self.T, dtype=torch.bool, device=self.device)
        close = self.pending[ar, role] & active
        idx = close.nonzero(as_tuple=True)[0]
        r_idx = role[idx]
        out = self._pack(self.s0[idx, r_idx].clone(), self.a0[idx, r_idx].clone(),
                         torch.zeros(idx.numel(), dtype=torch.float32, device=self.device),
                         face[idx].clone(), greedy[idx].clone(),
                         torch.zeros(idx.numel(), dtype=torch.bool, device=self.device), idx, r_idx)
        act = active.nonzero(as_tuple=True)[0]
        self.s0[act, role[act]] = face[act]
        self.a0[act, role[act]] = chosen[act]
        self.pending[act, role[act]] = True
        return out

    def after_step(self, role, done, r, terminal_face):
        """role int[T] the actor that just moved, done u8[T], r i8[T] (-1 lord won, +1 farmers
        won; rule_play.py:14), terminal_face f32[T,P,15,4] = env.observe() after the ply.
        Closes every pending transition of the finished tables."""
        done = done.to(self.device).bool()
        t_idx, r_idx = (self.pending & done[:, None]).nonzero(as_tuple=True)
        lord_won = (r.to(self.device)[t_idx] < 0)
        is_lord = r_idx == 1
        sign = torch.where(lord_won == is_lord, 1.0, -1.0)
        out = self._pack(self.s0[t_idx, r_idx].clone(), self.a0[t_idx, r_idx].clone(),
                         sign * self.reward[r_idx], terminal_face[t_idx].clone(),
                         torch.zeros((t_idx.numel(), 15, 4), dtype=torch.float32, device=self.device),
                         torch.ones(t_idx.numel(), dtype=torch.bool, device=self.device), t_idx, r_idx)
        self.pending[done] = False
        return out


def td_target(transitions, q_next, gamma=0.95):
    """y = r + (1 - done) * gamma * Q_target(s1, a1)  (dqn.py:40-41, config.py:8 GAMMA)."""
    return transitions["reward"] + (~transitions["done"]).float() * gamma * q_next.view(-1)
