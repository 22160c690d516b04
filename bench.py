#!/usr/bin/env python3
"""bench.py -- rows/sec of the h2o group-by hot path on MI355X, against the HBM roofline.

  python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run)

Workload (BASELINE.json metric "rows/sec on h2o groupby 1e9-row"): h2o Q1
`SELECT id1, sum(v1) FROM source GROUP BY id1` (reference benchmark/h2o/groupby.sql:1) on 1e9 synthetic
h2o-shaped rows PER GPU (K=100, seed 42, generated in HBM by the library's counter-based generator;
SURVEY.md 8d).  One step = one pass of the hot path over the resident columns: fused hash group-by +
sum (aqg_groupby_agg through the C-ABI) and, for N>1, the one RCCL exchange that merges the per-shard
group tables (all_gather of {key, partial sum}, re-aggregated by the same kernel) -- weak scaling.
The JSON line carries `roofline` (algorithmic 8 B/row over the HIP-event duration of the dominant
kernel) and, on rank 0 at N=1, `cpu_baseline` (the reference's own headers, or the oracle port, timed
on a bounded sample of the same workload on the host cores).
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 TB/s achievable copy)
Q1_BYTES_PER_ROW = 8           # id1 int32 + v1 int32


def cpu_baseline(sample_rows):
    """The reference CPU path (hash build -> ht_postproc -> per-group gather + sum), single thread --
    the post-processor is single-threaded in the reference (omp simd only).  Checker code: timed, never shipped."""
    import checker as ck
    oracle = ck.load_oracle()
    ref = ck.load_ref(fast=True)
    id1 = oracle.gen_column(ck.GEN_ID1, 42, 0, sample_rows, 10**9, 100)
    v1 = oracle.gen_column(ck.GEN_V1, 42, 0, sample_rows, 10**9, 100)
    kind, lib = ("reference", ref) if ref is not None else ("port", oracle)
    secs, groups, split = lib.time_groupby_sum([id1], [v1])
    if secs <= 0:
        kind, lib = "port", oracle
        secs, groups, split = lib.time_groupby_sum([id1], [v1])
    return {"value": sample_rows / secs, "unit": "rows/s", "cores": 1, "kind": kind,
            "sample": f"h2o Q1 on the first {sample_rows:.0e} rows of the same synthetic table "
                      f"(build {split[0]:.2f}s / ht_postproc {split[1]:.2f}s / gather+sum {split[2]:.2f}s; "
                      f"{groups} groups; host has {os.cpu_count()} cores, reference post-processor is single-threaded)"}


def secondary_entries(dev, n, ck, aq, id1, v1):
    """The other BASELINE configurations and the un-hinted / host-buffer forms of the headline, measured after the timed region on
    rank 0 at N=1 (never part of `value`): h2o Q5 (config 2), the moving windows of config 3, the fused join + group-by of config 4,
    and `seam_a` = Q1 the way the header layer calls it (hint 0, fresh handle), without and with the 8 GB host -> HBM ingest.
    Each entry: best-of-3 HIP-event time of the whole C-ABI call, algorithmic bytes, fraction of the 8 TB/s peak."""
    out = []
    def timed(fn, reps=3):
        best = 1e30
        for _ in range(reps):
            dev.sync(); dev.timer_start(); fn(); best = min(best, dev.timer_stop_ms())
        return best
    def entry(name, ms, abytes, **extra):
        gbs = abytes / (ms * 1e-3) / 1e9
        e = {"name": name, "ms": round(ms, 4), "rows_per_s": n / (ms * 1e-3), "algorithmic_bytes": int(abytes), "achieved_GBps": round(gbs, 1),
             "frac": round(gbs / HBM_PEAK_GBS, 4)}
        e.update(extra)
        out.append(e)
    K = 100
    # ---- config 2: h2o Q5 sum(v1), sum(v2), sum(v3) by id6 (benchmark/h2o/groupby.sql:9), N/K groups
    id6, v2, v3 = (dev.gen_column(c, 42, 0, n, n, K) for c in (ck.GEN_ID6, ck.GEN_V2, ck.GEN_V3))
    h = {}
    def q5():
        h["q5"] = dev.groupby_agg([id6], [ck.RED_SUM] * 3, [v1, v2, v3], hint=n // K + 1024, handle=h.get("q5"))
    ms = timed(q5)
    G = h["q5"].ngroups
    entry("h2o_q5_sum_v1_v2_v3_by_id6", ms, 16 * n + 44 * G, groups=int(G), plan=int(h["q5"].plan), kernel_ms_last_stage=round(dev.last_kernel_ms(), 4))
    h["q5"].destroy(); id6.free()
    # h2o Q3 `sum(v1), avg(v3) BY id3` and Q7 `max(v1), min(v2) BY id3` (benchmark/h2o/groupby.sql:5,14): the same plan over id3
    try:
        id3 = dev.gen_column(ck.GEN_ID3, 42, 0, n, n, K)
        for name, ops, vals in (("h2o_q3_sum_v1_avg_v3_by_id3", [ck.RED_SUM, ck.RED_AVG], [v1, v3]), ("h2o_q7_max_v1_min_v2_by_id3", [ck.RED_MAX, ck.RED_MIN], [v1, v2])):
            def q37():
                h["q37"] = dev.groupby_agg([id3], ops, vals, hint=n // K + 1024, handle=h.get("q37"))
            ms = timed(q37)
            entry(name, ms, 12 * n + 28 * h["q37"].ngroups, groups=int(h["q37"].ngroups), plan=int(h["q37"].plan))
            h.pop("q37").destroy()
        id3.free()
    except Exception as e:                                    # noqa: BLE001 -- a secondary entry never costs the line
        out.append({"name": "h2o_q3_q7", "error": str(e)[:300]})
    v2.free(); v3.free()
    # ---- config 2 again: h2o Q2 (two keys, 1e4 groups: the dense-domain plan) and Q10 (six keys, ~N groups: the wide-tuple partition
    # plan + the ordering tail; 48 GB of output), benchmark/h2o/groupby.sql:7,23
    try:
        id2 = dev.gen_column(ck.GEN_ID2, 42, 0, n, n, K)
        def q2():
            h["q2"] = dev.groupby_agg([id1, id2], [ck.RED_SUM], [v1], hint=16384, handle=h.get("q2"))
        ms = timed(q2)
        entry("h2o_q2_sum_v1_by_id1_id2", ms, 12 * n, groups=int(h["q2"].ngroups), kernel_ms=round(dev.last_kernel_ms(), 4))
        h["q2"].destroy()
        if n <= 1_000_000_000:
            ids = [id1, id2] + [dev.gen_column(c, 42, 0, n, n, K) for c in (ck.GEN_ID3, ck.GEN_ID4, ck.GEN_ID5, ck.GEN_ID6)]
            v3 = dev.gen_column(ck.GEN_V3, 42, 0, n, n, K)
            def q10():
                h["q10"] = dev.groupby_agg(ids, [ck.RED_SUM, ck.RED_COUNT], [v3, v3], hint=n, handle=h.get("q10"))
            ms = timed(q10, reps=2)
            G = h["q10"].ngroups
            entry("h2o_q10_sum_v3_count_by_id1_to_id6", ms, 28 * n + 48 * G, groups=int(G), plan=int(h["q10"].plan))
            h["q10"].destroy()
            for c in ids[2:]: c.free()
            v3.free()
        id2.free()
    except Exception as e:                                    # noqa: BLE001 -- a secondary entry never costs the line
        out.append({"name": "h2o_q2_q10", "error": str(e)[:300]})
    # ---- config 3: moving windows over an ordered series (tests/stock.a shapes; aggregations.h:127-281)
    price = dev.gen_column(ck.GEN_PRICE, 42, 0, n, n, K)
    big = dev.empty(n, ck.I128)
    for name, op, w, bpr in (("avgw5_price", ck.SCAN_AVGW, 5, 12), ("sumw5_price", ck.SCAN_SUMW, 5, 20), ("minw10_price", ck.SCAN_MINW, 10, 8)):
        ot = dev.lib.aqg_scan_out_dtype(op, price.tag)
        o = aq.DevBuf(dev, big.ptr, aq.capi.TAG2NP[ot], n, owned=False)
        ms = timed(lambda: dev.scan(op, price, w, keep=True, out=o), reps=5)
        entry(name, ms, bpr * n, kernel_ms=round(dev.last_kernel_ms(), 4))
    big.free(); price.free()
    # ---- seam A at high cardinality: h2o Q5 the way the header layer runs it -- aqg_groupby_build (hint 0: the group count is estimated from
    # a sample) + one aqg_grouped_reduce per aggregate (include/aquery/hasher.h, device.h), all resident; the host's per-group loop of the
    # generated code is not part of it (DESIGN.md section 5 gives its cost per group)
    try:
        id6, v2, v3 = (dev.gen_column(c, 42, 0, n, n, K) for c in (ck.GEN_ID6, ck.GEN_V2, ck.GEN_V3))
        t_b = timed(lambda: (h.__setitem__("sa", dev.groupby_build([id6])), None)[1], reps=2)
        gsa = h["sa"]
        outs = [dev.empty(gsa.ngroups, ck.I128) for _ in range(3)]
        def reduces():
            for x, o in zip((v1, v2, v3), outs):
                dev._chk(dev.lib.aqg_grouped_reduce(dev.ctx, gsa.h, ck.RED_SUM, x.tag, ctypes.c_void_p(x.ptr), ctypes.c_void_p(o.ptr)), "aqg_grouped_reduce")
        t_r = timed(reduces)
        entry("seam_a_q5_build_plus_three_grouped_sums", t_b + t_r, 16 * n + 44 * gsa.ngroups, groups=int(gsa.ngroups), build_ms=round(t_b, 3), three_reduces_ms=round(t_r, 3))
        gsa.destroy(); id6.free(); v2.free(); v3.free()
        for o in outs: o.free()
    except Exception as e:                                    # noqa: BLE001 -- a secondary entry never costs the line
        out.append({"name": "seam_a_q5", "error": str(e)[:300]})
    # ---- the reference's flagship shape: per-group windows inside the generated group loop (benchmark/quries/Aquery/q7.a
    # `SELECT stocksymbol, avgs(5, price) FROM trade ASSUMING ASC time GROUP BY stocksymbol`, mem_opt.cpp:53-63): the grouping is built once
    # (aqg_groupby_build), then ONE aqg_grouped_scan call answers every symbol -- price into the flat layout (value-carrying radix passes
    # over the group ids) + the segmented window kernel.  Algorithmic bytes: 4 price + 4 group id + 8 out per row.
    try:
        for label, Ksym in (("1e5_symbols", max(n // 100_000, 1)), ("100_symbols", None)):
            sym = dev.gen_column(ck.GEN_ID1, 42, 0, n, n, K) if Ksym is None else dev.gen_column(ck.GEN_ID3, 42, 0, n, n, Ksym)
            price = dev.gen_column(ck.GEN_PRICE, 42, 0, n, n, K)
            o = dev.empty(n, np.float64)
            t_b = timed(lambda: h.__setitem__("q7", dev.groupby_build([sym])), reps=1)
            gbq = h["q7"]
            ms = timed(lambda: dev.grouped_scan(gbq, ck.SCAN_AVGW, price, 5, keep=True, out=o))
            xf = dev.grouped_flatten(gbq, price, keep=True)
            ms_scan = timed(lambda: dev.grouped_scan(gbq, ck.SCAN_AVGW, xf, 5, flat=True, keep=True, out=o))
            entry("q7_avgs5_by_symbol" if Ksym is not None else "q7_avgs5_by_symbol_100_symbols", ms, 16 * n, groups=int(gbq.ngroups), symbols=label,
                  build_ms=round(t_b, 3), scan_only_ms=round(ms_scan, 4), scan_only_frac=round(12 * n / (ms_scan * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                  kernel_ms=round(dev.last_kernel_ms(), 4))
            gbq.destroy(); xf.free(); o.free(); price.free(); sym.free()
        # h2o Q9 `pow(corr(v1, v2), 2) BY id2, id4` (benchmark/h2o/groupby.sql:20): build over two keys + ONE aqg_grouped_corr
        id2, id4, v2 = (dev.gen_column(c, 42, 0, n, n, K) for c in (ck.GEN_ID2, ck.GEN_ID4, ck.GEN_V2))
        t_b = timed(lambda: h.__setitem__("q9", dev.groupby_build([id2, id4])), reps=1)
        ms = timed(lambda: dev.grouped_corr(h["q9"], v1, v2))
        entry("h2o_q9_corr_v1_v2_by_id2_id4", ms, 12 * n, groups=int(h["q9"].ngroups), build_ms=round(t_b, 3))
        h["q9"].destroy(); id2.free(); id4.free(); v2.free()
    except Exception as e:                                    # noqa: BLE001 -- a secondary entry never costs the line
        out.append({"name": "q7_q9", "error": str(e)[:300]})
    # ---- config 4 (one shard of it): fact JOIN small(id4, w), sum(v1 * w) by id1, fused
    id4 = dev.gen_column(ck.GEN_ID4, 42, 0, n, n, K)
    rng = np.random.default_rng(4)
    dim_key, dim_w = dev.to_device(rng.permutation(np.arange(1, K + 1, dtype=np.int32))), dev.to_device(rng.integers(1, 50, K).astype(np.int32))
    def jf():
        h["j"] = dev.join_groupby_sum(dim_key, dim_w, id4, id1, v1, hint=128, handle=h.get("j"))
    ms = timed(jf)
    entry("join_small_id4_sum_v1w_by_id1_fused", ms, 12 * n, kernel_ms=round(dev.last_kernel_ms(), 4))
    h["j"].destroy(); id4.free()
    # ---- seam A: the call the header layer makes (hint 0, a fresh handle per call), resident columns
    def q1_nohint(k, v):
        g = dev.groupby_agg([k], [ck.RED_SUM], [v], hint=0)
        g.destroy()
    ms = timed(lambda: q1_nohint(id1, v1))
    entry("seam_a_q1_hint0_fresh_handle", ms, 8 * n)
    # ---- seam A from host buffers: the two columns as the data source hands them over (pageable host memory), aqg_col_pin + Q1
    hid1, hv1 = id1.to_host(), v1.to_host()
    best, best_up = 1e30, 1e30
    for _ in range(2):
        dev.col_unpin_all(); dev.sync()
        t0 = time.perf_counter()
        dk, dv = dev.col_pin(hid1), dev.col_pin(hv1)
        dev.sync()
        t1 = time.perf_counter()
        best_up = min(best_up, t1 - t0)
        dev.col_unpin_all(); dev.sync()
        t0 = time.perf_counter()
        dk, dv = dev.col_pin(hid1), dev.col_pin(hv1)
        q1_nohint(dk, dv)
        dev.sync()
        best = min(best, time.perf_counter() - t0)
    dev.col_unpin_all()
    entry("seam_a_q1_from_host_buffers_h2d_included", best * 1e3, 8 * n, h2d_GBps=round(8 * n / best_up / 1e9, 1), h2d_ms=round(best_up * 1e3, 2),
          note="8 B/row cross PCIe once (aqg_col_pin: page-locked chunks + DMA), then the same call; bounded by the link, not by HBM")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", type=float, default=1e9, help="rows per GPU")
    ap.add_argument("--cpu-sample", type=float, default=1e8, help="rows of the CPU baseline sample (0 = skip)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary entries (configs 2-4, seam A) on rank 0 at N=1")
    ap.add_argument("--workload", choices=["q1", "join"], default="q1",
                    help="q1 (default, the graded metric): sum(v1) by id1.  join: BASELINE config 4, fact JOIN small(id4, w) ON id4, "
                         "sum(v1 * w) by id1 through the fused aqg_join_groupby_sum, same shard merge")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    n = int(args.rows)
    n_total = n * world

    import torch
    import aquery2_amd
    import checker as ck

    # AQG_BENCH_REHEARSAL=1: every rank on GPU 0 and the exchange over gloo -- a way to run the N>1 code path on a one-GPU
    # box (RCCL refuses two ranks on one device).  Never set by the driver; the numbers of such a run mean nothing.
    # AQG_BENCH_FORCE_COMM=1 (under torchrun with ONE rank): the multi-rank code path -- process group, communicator, sharded call --
    # with a world of one; AQG_BENCH_FORCE_TORCH_ALLGATHER=1 additionally takes the fallback transport.  Never set by the driver.
    multi = world > 1 or os.environ.get("AQG_BENCH_FORCE_COMM") == "1"
    rehearsal = world > 1 and os.environ.get("AQG_BENCH_REHEARSAL") == "1"
    gpu = 0 if rehearsal else local_rank
    xdev = "cpu" if rehearsal else "cuda"       # where the exchanged tensors live
    dist = None
    if multi:
        import torch.distributed as dist
        torch.cuda.set_device(gpu)
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", gpu))
    torch.cuda.set_device(gpu)

    # The library and torch share ONE stream, so that RCCL's all_gather is ordered after the pack kernel and the merge
    # after the all_gather without host syncs.  torch's default stream has handle 0, which the C-ABI reads as "create your
    # own", so a side stream is made current for the whole run.
    side = torch.cuda.Stream(device=gpu)
    torch.cuda.set_stream(side)
    stream = torch.cuda.current_stream().cuda_stream
    assert stream, "expected a non-default stream handle"
    dev = aquery2_amd.Device(gpu, stream=stream)
    # The exchange lives in the library (include/aqg.h: aqg_comm_*, aqg_groupby_agg_sharded): one communicator per rank, RCCL's
    # ncclAllGather on the library's stream.  torch.distributed only carries the 128-byte RCCL id from rank 0 to the others (and
    # the barrier / max-over-ranks of the timing).  Rehearsal and self-merge runs plug a host-side all-gather into the same C code.
    comm = None
    transport = "rccl in the library"
    if multi and not rehearsal:
        idt = torch.zeros(128, dtype=torch.uint8, device=xdev)
        if rank == 0:
            idt = torch.tensor(list(aquery2_amd.Comm.unique_id()), dtype=torch.uint8, device=xdev)
        dist.broadcast(idt, 0)
        torch.cuda.synchronize()
        ok = torch.ones(1, dtype=torch.int32, device=xdev)
        try:
            comm = aquery2_amd.Comm(dev, rank, world, nccl_id=bytes(idt.cpu().tolist()))
        except Exception as e:                                # noqa: BLE001 -- reported below; every rank must take the same path
            comm = None
            print(f"[bench] rank {rank}: the library's own RCCL communicator failed ({e}); falling back to torch's all-gather", file=sys.stderr, flush=True)
            ok.zero_()
        if os.environ.get("AQG_BENCH_FORCE_TORCH_ALLGATHER") == "1":
            ok.zero_()
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:
            # the same C code (aqg_groupby_agg_sharded) with torch.distributed's RCCL all-gather plugged in as the transport, on the
            # shared stream: a run that lands here still measures the exchange, and says so in its line
            if comm is not None:
                comm.destroy()
            def torch_allgather(send, recv, nbytes, stream):
                src = torch.as_tensor(aquery2_amd.DevBuf(dev, send, np.uint8, nbytes, owned=False), device="cuda")
                dst = torch.as_tensor(aquery2_amd.DevBuf(dev, recv, np.uint8, world * nbytes, owned=False), device="cuda")
                dist.all_gather_into_tensor(dst, src)
                return 0
            comm = aquery2_amd.Comm(dev, rank, world, allgather=torch_allgather)
            transport = "torch.distributed all_gather (library communicator failed)"
    elif multi:
        def gloo_allgather(send, recv, nbytes, stream):
            hs = np.empty(nbytes, np.uint8)
            dev._chk(dev.lib.aqg_d2h(dev.ctx, ctypes.c_void_p(hs.ctypes.data), ctypes.c_void_p(send), ctypes.c_size_t(nbytes)), "aqg_d2h")
            outs = torch.empty(world * nbytes, dtype=torch.uint8)
            dist.all_gather_into_tensor(outs, torch.from_numpy(hs))
            ho = outs.numpy()
            dev._chk(dev.lib.aqg_h2d(dev.ctx, ctypes.c_void_p(recv), ctypes.c_void_p(ho.ctypes.data), ctypes.c_size_t(world * nbytes)), "aqg_h2d")
            return 0
        comm = aquery2_amd.Comm(dev, rank, world, allgather=gloo_allgather)
    id1 = dev.gen_column(ck.GEN_ID1, 42, rank * n, n, n_total, 100)
    v1 = dev.gen_column(ck.GEN_V1, 42, rank * n, n, n_total, 100)
    join = args.workload == "join"
    if join:
        id4 = dev.gen_column(ck.GEN_ID4, 42, rank * n, n, n_total, 100)
        rng = np.random.default_rng(4)                    # the dimension table small(id4, w): identical on every rank
        dim_key_h = rng.permutation(np.arange(1, 101, dtype=np.int32))
        dim_w_h = rng.integers(1, 50, 100).astype(np.int32)
        dim_key, dim_w = dev.to_device(dim_key_h), dev.to_device(dim_w_h)
    dev.sync()

    # AQG_BENCH_SELFMERGE=1 (N=1 only): the whole sharded call with a world of one (pack, a device copy standing in for the
    # all-gather, concatenate, re-aggregate, finalise) -- what the exchange adds on top of the row pass.  Never set by the driver.
    selfmerge = world == 1 and os.environ.get("AQG_BENCH_SELFMERGE") == "1"
    if selfmerge:
        def self_allgather(send, recv, nbytes, stream):
            dev._chk(dev.lib.aqg_d2d(dev.ctx, ctypes.c_void_p(recv), ctypes.c_void_p(send), ctypes.c_size_t(nbytes)), "aqg_d2d")
            return 0
        comm = aquery2_amd.Comm(dev, 0, 1, allgather=self_allgather)
    state = {"gb": None, "merged": None}
    GMAX = 128                     # upper bound of a shard's group count (h2o K=100): fixed-size payload, ONE collective per step
    kernel_ms = []

    # set-up outside any step: the argument arrays of the N=1 Q1 call are marshalled once (a prepared call: every step still runs the
    # whole aqg_groupby_agg)
    if not join and comm is None:
        state["q1"], state["gb"] = dev.prepare_groupby_agg([id1], [ck.RED_SUM], [v1], hint=128)
    if comm is not None:                       # communicator warm-up (RCCL sets its channels up on the first collective)
        w = comm.groupby_agg_sharded([id1], [ck.RED_SUM], [v1], row_base=rank * n, hint=128, gmax=GMAX)
        w.destroy()
        if not join:                           # the sharded Q1 call, marshalled once like the N=1 one
            state["q1s"], state["merged"] = comm.prepare_groupby_agg_sharded([id1], [ck.RED_SUM], [v1], row_base=rank * n, hint=128, gmax=GMAX)

    def step(record):
        if join:
            gb = dev.join_groupby_sum(dim_key, dim_w, id4, id1, v1, hint=128, handle=state["gb"])
            state["gb"] = gb
            if record:
                kernel_ms.append(dev.last_kernel_ms())
            if comm is not None:               # the shard tables of the fused join: one partial (the 128-bit sum) per group
                state["merged"] = comm.groupby_exchange(gb, [ck.RED_SUM], row_base=rank * n, gmax=GMAX, handle=state["merged"])
        elif comm is not None:
            # ONE library call: this rank's group-by, pack, ONE all-gather of 129 x 3 words per rank, re-aggregation on every rank
            state["merged"] = state["q1s"]()
            if record:
                kernel_ms.append(dev.last_kernel_ms())     # the pass over this shard's rows (the re-aggregation does not replace it)
        else:
            state["gb"] = state["q1"]()
            if record:
                kernel_ms.append(dev.last_kernel_ms())

    def fence():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()
        dev.sync()

    for _ in range(args.warmup):
        step(False)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    fence()
    elapsed = time.perf_counter() - t0
    if multi:
        t = torch.tensor([elapsed], dtype=torch.float64, device=xdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # sanity: the merged / local result is the exact sum of v1 (checked against a second HIP reduction)
    final = state["merged"] if comm is not None else state["gb"]
    total = sum(ck.i128_to_int(final.result(0, ck.RED_SUM, ck.INT64 if join else ck.INT32)))
    if join:
        # independent check: sum(v1) by id4 (a plain group-by), dotted with w on the host
        by4 = dev.groupby_agg([id4], [ck.RED_SUM], [v1], hint=128)
        w_of = dict(zip(dim_key_h.tolist(), dim_w_h.tolist()))
        local_sum = sum(int(s_) * w_of[int(k_)] for k_, s_ in zip(by4.keys(0, np.int32), ck.i128_to_int(by4.result(0, ck.RED_SUM, ck.INT32))))
        by4.destroy()
    else:
        local_sum = int(dev.reduce(ck.RED_SUM, v1))
    if multi:
        t = torch.tensor([local_sum], dtype=torch.int64, device=xdev)
        dist.all_reduce(t)
        local_sum = int(t.item())
    assert total == local_sum, (total, local_sum)
    assert final.ngroups == 100

    if rank == 0:
        ms_step = 1e3 * elapsed / args.steps
        k_ms = float(np.mean(kernel_ms))
        bpr = 12 if join else Q1_BYTES_PER_ROW            # join: id4 + id1 + v1
        achieved = bpr * n / (k_ms * 1e-3) / 1e9
        line = {
            "metric": "rows/sec on h2o join + groupby 1e9-row" if join else "rows/sec on h2o groupby 1e9-row", "value": n_total * args.steps / elapsed, "unit": "rows/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int64", "data": "synthetic",
            "config": {"workload": (f"h2o_join_small_id4_w_then_sum_v1_times_w_by_id1_{n:.0e}_rows_per_gpu" if join else
                                    f"h2o_groupby_q1_sum_v1_by_id1_{n:.0e}_rows_per_gpu"), "rows_per_gpu": n, "K": 100,
                       "seed": 42, "groups": int(final.ngroups), "parallelism": f"row-range shards x{world}", **({"exchange": transport} if multi else {})},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         # `traffic` is a PMC counter of THIS run or null: bench.py collects none (counters need their own rocprofv3
                         # passes).  The profiled figure at 1e9 rows (FETCH_SIZE x2 + WRITE_SIZE, separate passes) is carried with
                         # the file it comes from: Q1 8.024 GB read + 0.012 GB written, join 12.001 + 0.018.
                         "traffic": None,
                         "traffic_profiled": {"bytes_at_1e9_rows": 12.019e9 if join else 8.036e9,
                                              "source": "profiles/r1_groupby_join_1e9_pmc.md" if join else "profiles/r3_bench_q1_1e9_pmc.md"},
                         "kernel": "starjoin_kernel" if join else "agg32_kernel<1,false,false,4>", "kernel_ms": k_ms, "algorithmic_bytes": bpr * n},
        }
        if world == 1 and comm is None and args.cpu_sample > 0 and not join:
            line["cpu_baseline"] = cpu_baseline(int(args.cpu_sample))
        if world == 1 and comm is None and not join and not args.no_secondary:
            state["gb"].destroy()
            try:
                line["secondary"] = secondary_entries(dev, n, ck, aquery2_amd, id1, v1)
            except Exception as e:                            # noqa: BLE001 -- the headline line is printed whatever happens behind it
                line["secondary_error"] = str(e)[:500]
        print(json.dumps(line), flush=True)
    if multi:
        dist.destroy_process_group()
    dev.close()


if __name__ == "__main__":
    main()
