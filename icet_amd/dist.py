"""Sharding independent scan pairs over the GPUs of one node.

The reference has no inter-process communication on this path: every scan pair is an independent
``ICET`` object (include/icet.h:36-116) and the only reduction inside a pair is V -> 1 of 27 floats.
So the path shards across PAIRS only: pair k goes to rank k mod world (round-robin, as BASELINE.json's
config 4 states), each rank runs its shard on its own GPU with no data-path collective, and ONE
all-gather of 48 floats per pair (X 6 + pred_stds 6 + 6x6 covariance) over RCCL/xGMI returns every
solution to every rank.  At 2048 pairs that is 393 KB in total -- latency-bound, nowhere near the
7 x ~153 GB/s xGMI links -- so a single fused gather at the end is the right granularity.

One process per GPU (torch.distributed, backend "nccl" == RCCL on ROCm); the same code runs under
"gloo" on CPU for the world_size-2 tests, with any callable as the local solver.
"""
import numpy as np
import torch
import torch.distributed as dist

RESULT_WIDTH = 48     # X[6] | pred_stds[6] | cov[36]


def shard_indices(n_pairs, rank, world):
    """Global pair ids owned by `rank`: k with k % world == rank (round-robin)."""
    return list(range(rank, n_pairs, world))


def shard_size(n_pairs, rank, world):
    return len(range(rank, n_pairs, world))


def max_shard_size(n_pairs, world):
    return (n_pairs + world - 1) // world


def gather_results(local, n_pairs, rank=None, world=None, group=None):
    """All-gather per-rank result blocks and undo the round-robin interleave.

    local: (shard_size, 48) float32 tensor on this rank's device (cuda for nccl, cpu for gloo).
    Returns (n_pairs, 48) in GLOBAL pair order on every rank.  Ragged shards (n_pairs % world != 0)
    are padded to the largest shard for the collective and the padding rows are dropped afterwards.
    """
    if world is None:
        world = dist.get_world_size(group) if dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank(group) if dist.is_initialized() else 0
    if local.dim() != 2 or local.shape[1] != RESULT_WIDTH:
        raise ValueError("local must be (shard, %d)" % RESULT_WIDTH)
    if local.shape[0] != shard_size(n_pairs, rank, world):
        raise ValueError("rank %d holds %d rows, expected %d" % (rank, local.shape[0], shard_size(n_pairs, rank, world)))
    if world == 1 and not dist.is_initialized():
        return local.clone()
    # (with an initialised process group the collective runs even for one rank: the RCCL path is then exercised by a 1-GPU test)
    m = max_shard_size(n_pairs, world)
    buf = torch.zeros((m, RESULT_WIDTH), dtype=local.dtype, device=local.device)
    buf[: local.shape[0]] = local
    out = torch.empty((world * m, RESULT_WIDTH), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, buf, group=group)
    # row (r*m + j) holds global pair r + world*j  ->  global order is the (j, r) transpose
    full = out.view(world, m, RESULT_WIDTH).transpose(0, 1).reshape(world * m, RESULT_WIDTH)
    return full[:n_pairs].contiguous()


def solve_sharded(n_pairs, solve_local, device, rank=None, world=None, group=None):
    """Run `solve_local(global_ids) -> (len(ids), 48) tensor on `device`` on this rank's shard and gather."""
    if world is None:
        world = dist.get_world_size(group) if dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank(group) if dist.is_initialized() else 0
    ids = shard_indices(n_pairs, rank, world)
    local = solve_local(ids) if ids else torch.zeros((0, RESULT_WIDTH), dtype=torch.float32, device=device)
    local = torch.as_tensor(np.asarray(local) if not torch.is_tensor(local) else local, dtype=torch.float32, device=device)
    return gather_results(local.reshape(-1, RESULT_WIDTH), n_pairs, rank, world, group)
