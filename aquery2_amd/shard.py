"""Row-range sharding of a table over the GPUs of one node, and the ONE exchange step that merges
per-shard group tables.  torch.distributed is plumbing here: RCCL ("nccl") on GPUs,
gloo in the CPU tests.  No data-path collective besides this merge: every shard scans its own rows.

Shards are contiguous row ranges in rank order, so concatenating the shards' group tables in rank
order (each already in local first-occurrence order) and grouping the concatenation again yields the
reference's global first-occurrence order -- no row ids need to cross the wire.
"""
import torch


def shard_rows(n_total, world, rank):
    """rows [lo, hi) of shard `rank` (the last shard takes the remainder)"""
    per = n_total // world
    lo = rank * per
    hi = n_total if rank == world - 1 else lo + per
    return lo, hi


def gather_group_tables(dist, cols, ngroups):
    """all_gather the first `ngroups` rows of each 1-D int64 tensor in `cols` from every rank and return the
    rank-ordered concatenations (padding removed).  Two collectives: the sizes, then one packed payload."""
    world = dist.get_world_size()
    dev = cols[0].device
    sizes = torch.zeros(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(sizes, torch.tensor([ngroups], dtype=torch.int64, device=dev))
    sizes = sizes.tolist()
    gmax = max(max(sizes), 1)
    pack = torch.zeros(gmax, len(cols), dtype=torch.int64, device=dev)
    for j, c in enumerate(cols):
        pack[:ngroups, j] = c[:ngroups]
    allp = torch.zeros(world * gmax, len(cols), dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(allp, pack)
    rows = torch.cat([allp[r * gmax: r * gmax + sizes[r]] for r in range(world)])
    return [rows[:, j].contiguous() for j in range(len(cols))]
