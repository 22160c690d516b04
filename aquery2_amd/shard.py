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


def gather_group_tables(dist, cols, ngroups, gmax=None):
    """all_gather the first `ngroups` rows of each 1-D int64 tensor in `cols` from every rank and return the
    rank-ordered concatenations (padding removed).
    gmax given (an upper bound of every rank's group count, e.g. the group-by hint): ONE collective -- the payload's
    first row carries the count.  Otherwise the sizes are exchanged first (two collectives)."""
    world = dist.get_world_size()
    dev = cols[0].device
    if gmax is None:
        sizes = torch.zeros(world, dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(sizes, torch.tensor([ngroups], dtype=torch.int64, device=dev))
        gmax = max(int(sizes.max().item()), 1)
    if ngroups > gmax:
        raise ValueError(f"gather_group_tables: {ngroups} groups exceed gmax={gmax}")
    pack = torch.zeros(gmax + 1, len(cols), dtype=torch.int64, device=dev)
    pack[0, 0] = ngroups
    for j, c in enumerate(cols):
        pack[1:ngroups + 1, j] = c[:ngroups]
    allp = torch.zeros(world * (gmax + 1), len(cols), dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(allp, pack)
    allp = allp.view(world, gmax + 1, len(cols))
    sizes = allp[:, 0, 0].tolist()
    rows = torch.cat([allp[r, 1:1 + sizes[r]] for r in range(world)])
    return [rows[:, j].contiguous() for j in range(len(cols))]


class GroupTableExchange:
    """The lean form of the merge used by bench.py: the shard's table is packed by one kernel (aqg_groupby_pack), ONE
    all_gather moves (gmax + 1) int64 pairs per rank, and aqg_groupby_merge_packed re-aggregates the concatenation --
    three library / collective calls per step, buffers allocated once.  `xdev` = "cuda" (RCCL) or "cpu" (gloo rehearsal)."""

    def __init__(self, dev, dist, gmax, key_tag, op, xdev="cuda"):
        self.dev, self.dist, self.gmax, self.key_tag, self.op, self.xdev = dev, dist, gmax, key_tag, op, xdev
        self.world = dist.get_world_size()
        self.pack = torch.zeros((gmax + 1) * 2, dtype=torch.int64, device="cuda")
        self.all = torch.zeros(self.world * (gmax + 1) * 2, dtype=torch.int64, device="cuda")
        if xdev != "cuda":
            self.all_x = torch.zeros_like(self.all, device=xdev)
        self.merged = None

    def __call__(self, gb, agg_index=0):
        self.dev.groupby_pack(gb, agg_index, self.gmax, self.pack.data_ptr())
        if self.xdev == "cuda":
            self.dist.all_gather_into_tensor(self.all, self.pack)
        else:
            self.dev.sync()
            self.dist.all_gather_into_tensor(self.all_x, self.pack.to(self.xdev))
            self.all.copy_(self.all_x)
            torch.cuda.synchronize()
        self.merged = self.dev.groupby_merge_packed(self.all.data_ptr(), self.world, self.gmax, self.key_tag, self.op, self.merged)
        return self.merged
