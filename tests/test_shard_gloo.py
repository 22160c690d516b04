"""world_size-2 test of the row-range sharding + group-table merge (aquery2_amd/shard.py) over gloo on CPU.
The per-shard group-by and the re-aggregation are done by the oracle here (the GPU runs the same
orchestration with the HIP kernels in bench.py); checked against the oracle on the whole table."""
import os
import sys

import numpy as np
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _worker(rank, world, port, q):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    import torch
    import torch.distributed as dist
    import checker as ck
    from aquery2_amd import shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    oracle = ck.load_oracle()
    n_total = 200_003
    lo, hi = shard.shard_rows(n_total, world, rank)
    # the generator is counter-based: a shard generates exactly its rows of the global table
    id1 = oracle.gen_column(ck.GEN_ID1, 42, lo, hi - lo, n_total, 100)
    v1 = oracle.gen_column(ck.GEN_V1, 42, lo, hi - lo, n_total, 100)
    gb = oracle.groupby([id1])
    sums = np.array(ck.i128_to_int(oracle.grouped_reduce(ck.RED_SUM, v1, gb)), dtype=np.int64)
    keys = id1[gb["first_rows"]].astype(np.int64)
    mk, ms = shard.gather_group_tables(dist, [torch.from_numpy(keys), torch.from_numpy(sums)], gb["ngroups"])
    mk, ms = mk.numpy().astype(np.int32), ms.numpy()
    mg = oracle.groupby([mk])
    merged_sums = ck.i128_to_int(oracle.grouped_reduce(ck.RED_SUM, ms, mg))
    merged_keys = mk[mg["first_rows"]]
    if rank == 0:
        full_id1 = oracle.gen_column(ck.GEN_ID1, 42, 0, n_total, n_total, 100)
        full_v1 = oracle.gen_column(ck.GEN_V1, 42, 0, n_total, n_total, 100)
        fg = oracle.groupby([full_id1])
        want_sums = ck.i128_to_int(oracle.grouped_reduce(ck.RED_SUM, full_v1, fg))
        want_keys = full_id1[fg["first_rows"]]
        q.put((merged_keys.tolist() == want_keys.tolist(), merged_sums == want_sums, len(merged_sums)))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_groupby_merge_matches_whole_table():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    ok_keys, ok_sums, g = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok_keys and ok_sums and g == 100


def test_shard_rows_cover_table():
    sys.path.insert(0, os.path.dirname(HERE))
    from aquery2_amd import shard
    for n, w in ((10, 3), (1_000_000_007, 8), (5, 8)):
        spans = [shard.shard_rows(n, w, r) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))


def _tails_worker(rank, world, port, q):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    import torch
    import torch.distributed as dist
    from aquery2_amd import shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rows = torch.arange(rank * 100, rank * 100 + 100, dtype=torch.int32)        # this shard's rows of an ordered column
    prev_tail, next_head = shard.exchange_tails(dist, rows[-4:], rows[:1])
    # the collective of sharded running sums: totals of up to 128 bits (negative ones too) and the row counts before a rank
    totals = [-(1 << 100) - 7, (1 << 70) + 3, 12345]
    carry, before = shard.exchange_totals(dist, (totals[rank], 100 + rank))
    fcarry, fbefore = shard.exchange_totals(dist, (0.5 + rank, 100 + rank))
    sums_ok = (carry == sum(totals[:rank]) and before == sum(100 + r for r in range(rank))
               and fcarry == sum(0.5 + r for r in range(rank)) and fbefore == before)
    q.put((rank, None if prev_tail is None else prev_tail.tolist(), None if next_head is None else next_head.tolist(), sums_ok))
    dist.barrier()
    dist.destroy_process_group()


def test_halo_exchange_delivers_the_neighbours_rows():
    """the one collective of sharded windows / shifts: every rank gets the last rows of the rank before it (and the first row of
    the rank after it); the device side is covered on the GPU by test_scans_and_windows_over_row_range_shards"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_tails_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(3):
        r, prev_tail, next_head, sums_ok = q.get(timeout=120)
        assert sums_ok, r
        got[r] = (prev_tail, next_head)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0] == (None, [100])
    assert got[1] == ([96, 97, 98, 99], [200])
    assert got[2] == ([196, 197, 198, 199], None)
