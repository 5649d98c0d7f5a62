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


def gather_trajectories(traj, group=None, dst=None, async_op=False, shard_sizes=None):
    """traj: uint8 [n_iters, T_local, 32] of this rank -> uint8 [n_iters, T_total, 32], tables in
    global id order, on every rank (dst=None: all_gather) or only on rank `dst` (gather to the
    learner; the other ranks get None).  One collective per call (few, large messages: xGMI is
    per-link bound); ragged shards are padded to the largest.  With async_op the call returns a
    handle at once -- the collective runs on the process group's own stream, so the next
    rollout on the current stream overlaps it -- and handle.result() waits for it.
    shard_sizes (tables per rank, if the caller knows them) skips the size exchange and its
    host sync."""
    if traj.dtype != torch.uint8 or traj.dim() != 3 or traj.shape[2] != TRAJ_BYTES:
        raise ValueError("traj must be uint8 [n_iters, T_local, 32]")
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
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
        pad = torch.zeros((n_iters, t_max - t_local, TRAJ_BYTES), dtype=torch.uint8, device=traj.device)
        send = torch.cat([send, pad], dim=1).contiguous()
    receives = dst is None or rank == dst
    recv = (torch.empty((world, n_iters, t_max, TRAJ_BYTES), dtype=torch.uint8, device=traj.device)
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


def unpack_trajectory(traj):
    """Named int views of packed records [..., 32] (layout: DESIGN.md / include/ddz_env.h)."""
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
