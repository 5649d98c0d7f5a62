"""Checkpoint / win-rate / log conventions of the reference, for batched runs.

The reference names everything after the start time `BEGIN = "%m%d_%H%M"` (Asia/Shanghai,
config.py:35-36) and turns the first underscores of a name into directories
(`name_dir`, config.py:30-31):

  models    <MODEL_DIR>/<mmdd>/<HHMM>/<role>_<episode>[_<wins>].pt     net.py:11-17, game.py:64-75,233-237
  win rates <WIN_DIR>/<mmdd>/<HHMM>.json  {"lord": [...], "down": [...], "up": [...]}   game.py:58-83
  log line  game.py:217-230 (recent/total win rate and mean loss per role)

A win belongs to the role that emptied its hand (game.py:129-167), which is what the engine's
`stats()` counts as lord_wins / up_wins / down_wins.  Nothing here touches the GPU.
"""
import json
import os
from datetime import datetime, timedelta, timezone

ROLES = ("lord", "down", "up")  # the reference's iteration order (game.py:31,233)


def name_dir(name, max_split=2):
    """'0805_1409_lord_4000' -> '0805/1409/lord_4000' (config.py:30-31)."""
    return os.path.join(*str(name).split("_", max_split))


def begin_stamp(now=None):
    """'%m%d_%H%M' in Asia/Shanghai (config.py:35-36); `now` = aware or naive-UTC datetime."""
    if now is None:
        now = datetime.now(timezone.utc)
    if now.tzinfo is None:
        now = now.replace(tzinfo=timezone.utc)
    try:
        from zoneinfo import ZoneInfo
        local = now.astimezone(ZoneInfo("Asia/Shanghai"))
    except Exception:  # no tz database: China has no DST, UTC+8 is exact
        local = now.astimezone(timezone(timedelta(hours=8)))
    return local.strftime("%m%d_%H%M")


def model_path(model_dir, name, max_split=2):
    """Path of Net.save(name) / Net.load(name) (net.py:11-17, 19-25)."""
    return os.path.join(model_dir, name_dir(name, max_split)) + ".pt"


def checkpoint_name(begin, role, episode, wins=None):
    """'<BEGIN>_<role>_<episode>' (game.py:236) or '<BEGIN>_<role>_<episode>_<wins>' (game.py:65,71,73)."""
    if role not in ROLES:
        raise ValueError("role must be lord / down / up")
    base = "{}_{}_{}".format(begin, role, episode)
    return base if wins is None else "{}_{}".format(base, wins)


def save_state_dict(module_or_state, model_dir, name, max_split=2):
    """Net.save (net.py:11-17): state_dict at model_path(...), directories created."""
    import torch
    path = model_path(model_dir, name, max_split)
    os.makedirs(os.path.dirname(path), exist_ok=True)
    state = module_or_state.state_dict() if hasattr(module_or_state, "state_dict") else module_or_state
    torch.save(state, path)
    return path


def load_state_dict(model_dir=None, name=None, abspath=None, max_split=2, map_location="cpu"):
    """Net.load (net.py:19-28) without executing anything from the file (weights_only)."""
    import torch
    path = abspath if abspath else model_path(model_dir, name, max_split)
    return torch.load(path, map_location=map_location, weights_only=True)


class WinRateBook:
    """Recent / total wins per role over log intervals, fed from `BatchedEnv.stats()` deltas
    (the batched counterpart of the counters in game.py:129-167, 58-88)."""

    def __init__(self, begin=None):
        self.begin = begin or begin_stamp()
        self.total = dict.fromkeys(ROLES, 0)
        self.recent = dict.fromkeys(ROLES, 0)
        self.history = {r: [] for r in ROLES}  # one entry per log interval (game.py:59-61)
        self.loss_sum = dict.fromkeys(ROLES, 0.0)
        self.loss_count = dict.fromkeys(ROLES, 0)
        self.episodes = 0
        self._last = None

    def update(self, stats):
        """stats: the cumulative dict of BatchedEnv.stats(); adds the wins since the last call."""
        cur = {"lord": int(stats["lord_wins"]), "up": int(stats["up_wins"]), "down": int(stats["down_wins"])}
        eps = int(stats["episodes"])
        if self._last is not None:
            for r in ROLES:
                d = cur[r] - self._last[0][r]
                self.recent[r] += d
                self.total[r] += d
            self.episodes += eps - self._last[1]
        self._last = (cur, eps)

    def add_loss(self, role, loss):
        """accumulate_loss (game.py:45-56): falsy losses are not counted."""
        if role not in ROLES:
            raise ValueError("role must be lord / down / up")
        if loss:
            self.loss_count[role] += 1
            self.loss_sum[role] += float(loss)

    def log_message(self, recent_episodes, seconds):
        """The reference's progress message (game.py:217-230) for the interval just finished."""
        n = max(1, int(recent_episodes))
        tot = max(1, self.episodes)
        mean = lambda r: self.loss_sum[r] / (self.loss_count[r] + 1e-3)  # noqa: E731
        lines = ["Reach at round {}, recent {} rounds takes {:.2f}seconds".format(self.episodes, n, seconds)]
        for label, r in (("Up  ", "up"), ("Lord", "lord"), ("Down", "down")):
            lines.append("\t{} recent/total win: {:.2%}/{:.2%} [Mean loss: {:.2f}]".format(
                label, self.recent[r] / n, self.total[r] / tot, mean(r)))
        return "\n".join(lines) + "\n"

    def close_interval(self, win_dir=None):
        """save_win_rates + reset_recent (game.py:58-88): append the interval, write the JSON."""
        for r in ROLES:
            self.history[r].append(self.recent[r])
        path = None
        if win_dir is not None:
            path = os.path.join(win_dir, name_dir(self.begin)) + ".json"
            os.makedirs(os.path.dirname(path), exist_ok=True)
            with open(path, "w") as f:
                json.dump({"lord": self.history["lord"], "down": self.history["down"], "up": self.history["up"]}, f)
        self.recent = dict.fromkeys(ROLES, 0)
        self.loss_sum = dict.fromkeys(ROLES, 0.0)
        self.loss_count = dict.fromkeys(ROLES, 0)
        return path
