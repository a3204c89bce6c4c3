"""Replica sharding across the GPUs of a node: worlds are independent, so each rank owns a contiguous
block of global replica ids and the only collective of a job is ONE all-gather of the metric counters (a row per rank)."""
import torch

collectives_executed = 0   # collectives this module has issued in this process (bench.py reports it; tests assert it)


def shard(n_worlds_total, rank, world_size):
    """(first global replica id, number of replicas) of this rank: contiguous blocks, remainder to the low ranks."""
    base, rem = divmod(n_worlds_total, world_size)
    count = base + (1 if rank < rem else 0)
    first = rank * base + min(rank, rem)
    return first, count


def backend_of(dist):
    """Name of the process group's backend ("nccl" = RCCL on ROCm, "gloo", ...); "" for a stand-in without get_backend."""
    try:
        return str(dist.get_backend())
    except Exception:  # noqa: BLE001
        return ""


def gather_rows(row, dist):
    """ONE all-gather of a flat float64 tensor per rank -> [world_size, row.numel()], identical on every rank.  Under a CPU-only
    backend (gloo) with device tensors the few hundred bytes hop through the host (the 2-rank dry runs that share one GPU:
    tests/test_hip_round5.py; the result then stays on the host); under nccl (= RCCL) the gather runs on the device over xGMI."""
    row = row.reshape(-1).contiguous()
    if row.is_cuda and backend_of(dist) == "gloo":
        row = row.cpu()
    flat = torch.empty(dist.get_world_size() * row.numel(), dtype=row.dtype, device=row.device)
    dist.all_gather_into_tensor(flat, row)   # (flat output: the shape every backend accepts)
    return flat.view(dist.get_world_size(), row.numel())


def reduce_counters(counters, elapsed_s, dist=None):
    """counters: 1-D float64 tensor of additive metrics; returns (summed counters, max elapsed, per-rank table) over all ranks.

    ONE collective: every rank contributes a row [counters..., elapsed] to an all-gather of O(100 B) per rank (pure latency
    over xGMI, never per step); sums, the maximum of the elapsed times and the per-rank table (straggler diagnosis: bench.py
    prints min / max of the per-rank rates) all come out of it.  Runs whenever a process group exists -- also at world size 1,
    where RCCL executes the same call."""
    row = torch.cat([counters.to(torch.float64), torch.tensor([elapsed_s], dtype=torch.float64, device=counters.device)])
    if dist is not None and dist.is_initialized():
        global collectives_executed
        collectives_executed += 1
        table = gather_rows(row, dist)
    else:
        table = row[None, :].clone()
    return table[:, :-1].sum(0), float(table[:, -1].max().item()), table.cpu()
