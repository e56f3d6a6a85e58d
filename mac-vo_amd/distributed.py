"""Multi-GPU execution: one process per GPU, independent sequences per rank, one pose gather.

The reference runs sequences one after another in a single process (``Scripts/Experiment/Experiment_MACVO.py:55-58``)
and has no collective anywhere.  Sequences are independent, so the MI355X-native plan is the
embarrassingly parallel one: rank r owns sequences ``{s : s % world == r}``; nothing is exchanged while frames are
processed; at the end every rank contributes its ``[T, 7]`` poses to ONE all_gather (RCCL over xGMI when the backend
is "nccl", gloo on CPU for the tests).  The payload is ~28 B per frame, i.e. latency- not bandwidth-bound — no ring
all-reduce, no bucketing.
"""
from __future__ import annotations

import torch


def shard_sequences(n_sequences: int, rank: int, world: int) -> list[int]:
    """Round-robin ownership of independent sequences."""
    return [s for s in range(n_sequences) if s % world == rank]


def gather_poses(poses: torch.Tensor, dist=None, lengths: torch.Tensor | None = None) -> torch.Tensor:
    """All-gather per-rank pose tracks.

    poses ``[T, 7]`` (same T on every rank -> one all_gather_into_tensor) or ragged with ``lengths`` given
    (two-phase: gather lengths, pad to the max, gather payload).  Returns ``[world, T_max, 7]``; with no process
    group it is ``poses[None]``.
    """
    if dist is None or not dist.is_initialized():
        return poses[None]
    world = dist.get_world_size()
    T = torch.tensor([poses.shape[0]], dtype=torch.int64, device=poses.device)
    if lengths is not None:
        all_T = [torch.zeros_like(T) for _ in range(world)]
        dist.all_gather(all_T, T)
        t_max = int(torch.stack(all_T).max().item())
        if poses.shape[0] < t_max:
            pad = torch.zeros((t_max - poses.shape[0], poses.shape[1]), dtype=poses.dtype, device=poses.device)
            poses = torch.cat([poses, pad], dim=0)
    poses = poses.contiguous()
    out = torch.empty((world,) + tuple(poses.shape), dtype=poses.dtype, device=poses.device)
    if poses.is_cuda:
        dist.all_gather_into_tensor(out, poses)
    else:  # gloo: list form
        parts = [torch.empty_like(poses) for _ in range(world)]
        dist.all_gather(parts, poses)
        out = torch.stack(parts)
    return out
