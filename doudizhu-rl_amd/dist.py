"""Multi-GPU: tables are independent (one Env per Game, game.py:28), so they shard by
global table id with no traffic while stepping; the only exchange is the end-of-batch
gather of the packed trajectories (RCCL all_gather over xGMI; `gloo` in the CPU tests).

One process per GPU (torchrun); every rank owns the contiguous id range
[rank * T_local, (rank + 1) * T_local).  The RNG is keyed by the *global* table id, so the
union of the shards is bit-identical to a single-process run over all tables.
"""
import torch
import torch.distributed as dist

TRAJ_BYTES = 32


def shard_tables(total_tables, rank, world_size):
    """(n_local, table_id_base) of `rank`; the remainder goes to the low ranks."""
    total_tables, rank, world_size = int(total_tables), int(rank), int(world_size)
    if not 0 <= rank < world_size:
        raise ValueError("rank out of range")
    q, r = divmod(total_tables, world_size)
    n_local = q + (1 if rank < r else 0)
    base = rank * q + min(rank, r)
    return n_local, base


class _Pending:
    """Handle of an in-flight trajectory exchange; result() waits and returns the tensor
    ([n_iters, T_total, 32] on the ranks that receive, None elsewhere)."""

    def __init__(self, works, finish):
        self._works, self._finish, self._out = works, finish, None

    def result(self):
        if self._finish is not None:
            for w in self._works:
                w.wait()
            self._out = self._finish()
            self._finish = None
        return self._out


def gather_trajectories(traj, group=None, dst=None, async_op=False, shard_sizes=None, compact=False,
                        _force_collective=False):
    """traj: uint8 [n_iters, T_local, 32] of this rank -> uint8 [n_iters, T_total, 32], tables in
    global id order, on every rank (dst=None: all_gather) or only on rank `dst` (gather to the
    learner; the other ranks get None).  One collective per call (few, large messages: xGMI is
    per-link bound); ragged shards are padded to the largest.  With async_op the call returns a
    handle at once -- the collective runs on the process group's own stream, so the next
    rollout on the current stream overlaps it -- and handle.result() waits for it.
    shard_sizes (tables per rank, if the caller knows them) skips the size exchange and its
    host sync.  compact=True (GPU tensors): the records are packed to 8 bytes (the action as its
    canonical id, engine.pack_trajectory) before the collective -- 4x fewer bytes over xGMI -- and the
    result is uint8 [n_iters, T_total, 8]; unpack_trajectory() reads both forms."""
    if traj.dtype != torch.uint8 or traj.dim() != 3 or traj.shape[2] not in (TRAJ_BYTES, 8):
        raise ValueError("traj must be uint8 [n_iters, T_local, 32] (or already packed: [..., 8])")
    if compact and traj.shape[2] == TRAJ_BYTES:
        from .engine import pack_trajectory
        traj = pack_trajectory(traj)
    rec = traj.shape[2]
    # (_force_collective: tests run the real collective on a one-rank group -- RCCL on a single-GPU box)
    if not dist.is_available() or not dist.is_initialized() or (dist.get_world_size(group) == 1 and not _force_collective):
        pend = _Pending([], lambda: traj)
        return pend if async_op else pend.result()
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    n_iters, t_local = traj.shape[0], traj.shape[1]
    if shard_sizes is not None:
        all_sizes = [int(x) for x in shard_sizes]
        if len(all_sizes) != world or all_sizes[rank] != t_local:
            raise ValueError("shard_sizes does not match this process group")
    else:
        sizes = torch.tensor([t_local], dtype=torch.int64, device=traj.device)
        all_sizes = [torch.zeros_like(sizes) for _ in range(world)]
        dist.all_gather(all_sizes, sizes, group=group)
        all_sizes = [int(s.item()) for s in all_sizes]
    t_max = max(all_sizes)
    send = traj.contiguous()
    if t_local != t_max:
        pad = torch.zeros((n_iters, t_max - t_local, rec), dtype=torch.uint8, device=traj.device)
        send = torch.cat([send, pad], dim=1).contiguous()
    receives = dst is None or rank == dst
    recv = (torch.empty((world, n_iters, t_max, rec), dtype=torch.uint8, device=traj.device)
            if receives else None)
    if dst is None:
        work = dist.all_gather_into_tensor(recv.view(-1), send.view(-1), group=group, async_op=True)
    else:
        glist = [recv[r] for r in range(world)] if receives else None
        work = dist.gather(send, glist, dst=dst, group=group, async_op=True)

    def finish():
        if not receives:
            return None
        return torch.cat([recv[r, :, :all_sizes[r]] for r in range(world)], dim=1).contiguous()

    pend = _Pending([work], finish)
    pend._keep = (send, recv)  # the buffers must outlive the collective
    return pend if async_op else pend.result()


def unpack_trajectory(traj, action_rows=None):
    """Named int views of trajectory records: [..., 32] (layout: DESIGN.md / include/ddz_env.h) or the
    compact [..., 8] form (then "id" is the canonical action id -- 0x3FFF for a record that holds no action
    of the table -- and "row" is filled in when `action_rows` = engine.action_table() is given; "episode"
    holds the low 14 bits)."""
    if traj.shape[-1] == 8:
        t = traj.to(torch.int64)
        w = lambda lo: t[..., lo] | (t[..., lo + 1] << 8) | (t[..., lo + 2] << 16) | (t[..., lo + 3] << 24)  # noqa: E731
        w0, w1 = w(0), w(4)
        rc = (w0 >> 26) & 3
        out = {"id": w0 & 0x3FFF, "n_legal": (w0 >> 14) & 0x1FF, "role": (w0 >> 23) & 3, "done": (w0 >> 25) & 1,
               "reward": torch.where(rc == 2, -torch.ones_like(rc), rc), "flags": (w0 >> 28) & 3,
               "choice": (w1 & 0x3FF) - 1, "ply": (w1 >> 10) & 0xFF, "episode": (w1 >> 18) & 0x3FFF}
        if action_rows is not None:
            ids = out["id"].to(action_rows.device)
            none = ids >= action_rows.shape[0]                      # 0x3FFF: the record holds no action -> zero row
            out["row"] = torch.where(none.unsqueeze(-1), torch.zeros_like(action_rows[:1]),
                                     action_rows[ids.clamp(max=action_rows.shape[0] - 1)])
        return out
    t = traj.to(torch.int64)
    u16 = lambda lo: t[..., lo] | (t[..., lo + 1] << 8)  # noqa: E731
    u32 = lambda lo: u16(lo) | (u16(lo + 2) << 16)       # noqa: E731
    rew = t[..., 18]
    choice = u32(28)
    return {
        "row": traj[..., :16].view(torch.int8) if traj.dtype == torch.uint8 else traj[..., :16],
        "role": t[..., 16], "done": t[..., 17], "reward": torch.where(rew > 127, rew - 256, rew),
        "flags": t[..., 19], "n_legal": u16(20), "ply": u16(22), "episode": u32(24),
        "choice": torch.where(choice >= (1 << 31), choice - (1 << 32), choice),
    }
