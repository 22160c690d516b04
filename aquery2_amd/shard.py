"""Row-range sharding of a table over the GPUs of one node: shard arithmetic, the halo / carry exchanges of scans and windows
 and a torch-level form of the group-table merge for the CPU (gloo) tests.  The merge the product runs is in the
library itself (aqg_groupby_agg_sharded: RCCL inside the C-ABI).  torch.distributed is plumbing here.

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


# (The group-table merge of bench.py lives in the library since round 2: aqg_comm_* / aqg_groupby_agg_sharded, include/aqg.h.
#  gather_group_tables above remains the torch-level form the gloo tests drive with the oracle's kernels.)


# ---- reductions and scans over row-range shards: the product path is the LIBRARY's (round 3) -----------------------------------
# aqg_reduce_sharded / aqg_corr_sharded / aqg_scan_sharded (include/aqg.h, csrc/sharded.hip): one all-gather of moments / neighbours /
# halos on the communicator, every fold in C++ -- a C++ host shards them without Python.  These are the thin callers; the functions
# further down are the torch-level forms of the same exchanges that the CPU (gloo) tests drive and that feed halos by hand on one GPU.

def reduce_sharded(comm, op, x_local):
    """op over the whole column from this rank's rows (aquery2_amd.Comm.reduce_sharded -> aqg_reduce_sharded)"""
    return comm.reduce_sharded(op, x_local)


def corr_sharded(comm, x_local, y_local):
    return comm.corr_sharded(x_local, y_local)


def scan_sharded(comm, op, x_local, w=0):
    """this rank's rows of the scan / window / shift over the whole column (aqg_scan_sharded)"""
    return comm.scan_sharded(op, x_local, w)


# ---- scans and windows over row-range shards ("windows need a fixed halo") ---------------------------------------
# A shard needs the last w - 1 rows of the shard before it (sliding windows), its last row (deltas / prev / ratios), or one
# value folded over everything before it (running min / max).  `exchange_tails` is the one collective; the functions below
# are pure device code so that they can be tested on one GPU by feeding the halos by hand.

def exchange_tails(dist, tail, head=None):
    """all_gather every rank's `tail` (1-D tensor, the same length on every rank: its last h rows) and return the tail of the
    previous rank (None on rank 0).  With `head` (its first rows) also returns the head of the next rank (None on the last)."""
    world, rank = dist.get_world_size(), dist.get_rank()
    out = torch.zeros(world * tail.numel(), dtype=tail.dtype, device=tail.device)
    dist.all_gather_into_tensor(out, tail.contiguous())
    prev_tail = out.view(world, -1)[rank - 1].clone() if rank > 0 else None
    if head is None:
        return prev_tail
    outh = torch.zeros(world * head.numel(), dtype=head.dtype, device=head.device)
    dist.all_gather_into_tensor(outh, head.contiguous())
    next_head = outh.view(world, -1)[rank + 1].clone() if rank + 1 < world else None
    return prev_tail, next_head


def _concat(dev, parts, dtype):
    """device buffer holding the concatenation of DevBufs / None entries"""
    from .capi import DevBuf
    import ctypes as C
    import numpy as np
    parts = [p for p in parts if p is not None and p.n]
    n = sum(p.n for p in parts)
    buf = dev.empty(max(n, 1), dtype)
    off = 0
    isz = np.dtype(dtype).itemsize
    for p in parts:
        dev._chk(dev.lib.aqg_d2d(dev.ctx, C.c_void_p(buf.ptr + off * isz), C.c_void_p(p.ptr), C.c_size_t(p.n * isz)), "aqg_d2d")
        off += p.n
    return DevBuf(dev, buf.ptr, dtype, n, owned=False), buf


def window_scan_with_halo(dev, op, x_local, halo_prev, w):
    """sumw / avgw / minw / maxw / ratiow of this shard's rows given the last (w - 1) rows (ratiow: w rows) of the shard
    before it (`halo_prev`: DevBuf or None on the first shard).  Returns a DevBuf view of the n_local results (+ its owner)."""
    from .capi import DevBuf, TAG2NP
    import numpy as np
    h = halo_prev.n if halo_prev is not None else 0
    cat, owner = _concat(dev, [halo_prev, x_local], x_local.dtype)
    res = dev.scan(op, cat, w, keep=True)
    ot = np.dtype(TAG2NP[dev.lib.aqg_scan_out_dtype(op, x_local.tag)])
    view = DevBuf(dev, res.ptr + h * ot.itemsize, ot, x_local.n, owned=False)
    view._keep = (res, owner)
    return view


def shift_scan_with_neighbours(dev, op, x_local, prev_last, next_first):
    """deltas / prev / aggnext with the neighbouring shards' boundary rows (1-element DevBufs or None at the table's ends)"""
    from .capi import DevBuf
    import numpy as np
    h = 1 if prev_last is not None else 0
    cat, owner = _concat(dev, [prev_last, x_local, next_first], x_local.dtype)
    res = dev.scan(op, cat, 0, keep=True)
    view = DevBuf(dev, res.ptr + h * np.dtype(x_local.dtype).itemsize, x_local.dtype, x_local.n, owned=False)
    view._keep = (res, owner)
    return view


def running_minmax_with_carry(dev, op, x_local, carry):
    """mins / maxs of this shard given the fold of every earlier row (`carry`: 1-element DevBuf or None): the carry is scanned
    as a row in front of the shard's rows"""
    from .capi import DevBuf
    import numpy as np
    h = 1 if carry is not None else 0
    cat, owner = _concat(dev, [carry, x_local], x_local.dtype)
    res = dev.scan(op, cat, 0, keep=True)
    view = DevBuf(dev, res.ptr + h * np.dtype(x_local.dtype).itemsize, x_local.dtype, x_local.n, owned=False)
    view._keep = (res, owner)
    return view


def exchange_totals(dist, total, device="cpu"):
    """all_gather every rank's column total and row count; returns (sum of the totals of the ranks before this one, rows before
    this rank).  `total`: python int of up to 128 bits (integer columns; travels as two int64 halves) or float."""
    world, rank = dist.get_world_size(), dist.get_rank()
    total, rows = total
    if isinstance(total, float):
        mine = torch.tensor([total, float(rows)], dtype=torch.float64, device=device)
        out = torch.zeros(world * 2, dtype=torch.float64, device=device)
        dist.all_gather_into_tensor(out, mine)
        o = out.view(world, 2).tolist()
        carry = 0.0
        for r in range(rank):
            carry += o[r][0]
        return (carry if rank else -0.0), int(sum(o[r][1] for r in range(rank)))
    u = int(total) & ((1 << 128) - 1)
    halves = [(u >> s) & 0xFFFFFFFF for s in (0, 32, 64, 96)]          # four 32-bit limbs: every one fits an int64 as is
    mine = torch.tensor(halves + [int(rows)], dtype=torch.int64, device=device)
    out = torch.zeros(world * 5, dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(out, mine)
    o = out.view(world, 5).tolist()
    carry = 0
    for r in range(rank):
        v = o[r][0] | (o[r][1] << 32) | (o[r][2] << 64) | (o[r][3] << 96)
        carry += v - (1 << 128) if v >> 127 else v                        # back to a signed 128-bit value
    return carry, int(sum(o[r][4] for r in range(rank)))


def running_sums_with_carry(dev, op, x_local, carry, rows_before):
    """sums / avgs of this shard given the sum and the number of every earlier row (from exchange_totals)"""
    return dev.scan_resume(op, x_local, carry, rows_before, keep=True)
