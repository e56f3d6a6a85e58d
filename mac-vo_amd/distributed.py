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


def gather_tracks(poses: torch.Tensor, time_ns: torch.Tensor | None = None, dist=None):
    """All-gather per-rank trajectories: ``poses [T,7]`` fp32 + ``time_ns [T]`` int64 + ``T``.

    Two phases, both tiny: (1) all_gather of the track lengths, (2) ONE all_gather of a ``[T_max, 9]`` fp32 payload
    (7 pose floats + the int64 timestamp carried as two 32-bit words), rows beyond a rank's own T zero-padded.  Ragged
    tracks are therefore always safe; the lengths come back so callers can strip the padding.
    Returns ``(poses [world, T_max, 7] fp32, time_ns [world, T_max] int64, lengths [world] int64)``; without a process
    group the inputs are returned with a leading dimension of 1.
    """
    assert poses.dim() == 2 and poses.shape[1] == 7 and poses.dtype == torch.float32, "poses must be [T,7] float32"
    T = poses.shape[0]
    dev = poses.device
    if time_ns is None:
        time_ns = torch.zeros((T,), dtype=torch.int64, device=dev)
    assert time_ns.shape == (T,) and time_ns.dtype == torch.int64, "time_ns must be [T] int64"
    if dist is None or not dist.is_initialized():
        return poses[None], time_ns[None], torch.full((1,), T, dtype=torch.int64, device=dev)     # (a fill launch: no pageable host-to-device copy, no host wait)
    world = dist.get_world_size()
    mine = torch.full((1,), T, dtype=torch.int64, device=dev)
    lengths = torch.empty((world,), dtype=torch.int64, device=dev)
    _all_gather(dist, lengths, mine)
    t_max = int(lengths.max().item())
    payload = torch.zeros((t_max, 9), dtype=torch.float32, device=dev)
    payload[:T, :7] = poses
    payload[:T, 7:] = time_ns.contiguous().view(torch.float32).view(T, 2)     # bit pattern, not a conversion
    out = torch.empty((world, t_max, 9), dtype=torch.float32, device=dev)
    _all_gather(dist, out, payload)
    ts = out[:, :, 7:].contiguous().view(torch.int64).view(world, t_max)
    return out[:, :, :7].contiguous(), ts, lengths


def _all_gather(dist, out: torch.Tensor, mine: torch.Tensor) -> None:
    if mine.is_cuda and dist.get_backend() != "gloo":   # RCCL over xGMI (backend "nccl")
        dist.all_gather_into_tensor(out, mine.contiguous())
    else:              # gloo: list form, on host tensors (gloo has no device all_gather: device tracks are staged through the host —
        #                the 2-ranks-on-one-GPU test and CPU tests only; 36 B per frame)
        src = mine.contiguous().cpu()
        parts = [torch.empty_like(src) for _ in range(out.shape[0])]
        dist.all_gather(parts, src)
        out.copy_(torch.stack(parts).view_as(out))


def gather_poses(poses: torch.Tensor, dist=None, time_ns: torch.Tensor | None = None):
    """``gather_tracks`` returning ``(poses [world, T_max, 7], lengths [world])`` (padding rows are zero)."""
    p, _, lengths = gather_tracks(poses, time_ns, dist)
    return p, lengths
