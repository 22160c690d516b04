"""ctypes binding of the C-ABI (include/aqg.h).  Test/bench harness only: numpy in, numpy out.

There is NO fallback: if libaqg.so is missing or no GPU is visible, construction raises.
"""
import ctypes as C
import os
import weakref

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

# dtype tags (reference server/aquery_types.h:1-5)
INT32, FLOAT, STR, DOUBLE, LDOUBLE, INT64, INT128, INT16, DATE, TIME, INT8 = range(11)
UINT32, UINT64, UINT128, UINT16, UINT8, BOOL = 11, 12, 13, 14, 15, 16
TIMESTAMP = 18
ERROR = 22
I128 = np.dtype([("lo", "<u8"), ("hi", "<i8")])
U128 = np.dtype([("lo", "<u8"), ("hi", "<u8")])
TAG2NP = {
    INT8: np.dtype(np.int8), INT16: np.dtype(np.int16), INT32: np.dtype(np.int32), INT64: np.dtype(np.int64),
    UINT8: np.dtype(np.uint8), UINT16: np.dtype(np.uint16), UINT32: np.dtype(np.uint32), UINT64: np.dtype(np.uint64),
    FLOAT: np.dtype(np.float32), DOUBLE: np.dtype(np.float64), BOOL: np.dtype(np.uint8), INT128: I128, UINT128: U128,
}
NP2TAG = {v: k for k, v in TAG2NP.items() if k != BOOL}
NP2TAG[np.dtype(np.bool_)] = BOOL
VEC_VEC, VEC_SCALAR, SCALAR_VEC = 0, 1, 2
PLAN_FAST_LDS, PLAN_SMALL_LDS, PLAN_BIG_LDS, PLAN_DENSE, PLAN_PART_ONE, PLAN_PART_TWO, PLAN_PART_ROUND1, PLAN_PART_WIDE, PLAN_SORTED_TAIL, PLAN_HBM_TABLE, PLAN_BUILD_PARTITIONED, PLAN_GID_PARTITION, PLAN_PACKED_VALUES, PLAN_RANGE_PARTITIONS, PLAN_ROW_EMIT, PLAN_PACKED_KEYS, PLAN_BUILD_LOOKUP = (1 << i for i in range(17))


class AqgError(RuntimeError):
    def __init__(self, what, code, detail=""):
        super().__init__(f"{what}: status {code} {detail}")
        self.code = code


def lib_path():
    return os.path.join(HERE, "libaqg.so")


_LIB = None


def load_library():
    """dlopen the in-tree libaqg.so; raises if it has not been built (no fallback)."""
    global _LIB
    if _LIB is None:
        p = lib_path()
        if not os.path.exists(p):
            raise FileNotFoundError(f"{p} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                                    "or `make -C aquery2_amd/csrc`")
        lib = C.CDLL(p)
        lib.aqg_version.restype = C.c_char_p
        lib.aqg_last_error.restype = C.c_char_p
        lib.aqg_dtype_size.restype = C.c_size_t
        lib.aqg_ctx_stream.restype = C.c_void_p
        lib.aqg_groupby_ngroups.restype = C.c_uint32
        lib.aqg_groupby_nrows.restype = C.c_uint32
        lib.aqg_groupby_first_rows64.restype = C.c_void_p
        for f in ("aqg_groupby_reversemap", "aqg_groupby_counts", "aqg_groupby_first_rows", "aqg_groupby_agg_result"):
            getattr(lib, f).restype = C.c_void_p
        _LIB = lib
    return _LIB


def tag_of(a):
    return NP2TAG[np.asarray(a).dtype]


class DevBuf:
    """A device allocation (HBM) with a dtype and element count."""

    def __init__(self, dev, ptr, dtype, n, owned=True):
        self.dev, self.ptr, self.dtype, self.n, self.owned = dev, ptr, np.dtype(dtype), int(n), owned

    @property
    def tag(self):
        if getattr(self, "_tag", None) is not None:
            return self._tag
        return NP2TAG[self.dtype] if self.dtype in NP2TAG else (INT128 if self.dtype == I128 else UINT128)

    @property
    def __cuda_array_interface__(self):
        """zero-copy view for torch.as_tensor(buf, device="cuda") (bench.py's RCCL merge); 128-bit
        columns are exposed as 2n int64 words"""
        if self.dtype.names:
            return {"shape": (2 * self.n,), "typestr": "<i8", "data": (int(self.ptr), False), "version": 3}
        return {"shape": (self.n,), "typestr": self.dtype.str, "data": (int(self.ptr), False), "version": 3}

    def to_host(self):
        out = np.empty(self.n, dtype=self.dtype)
        if self.n:
            self.dev._chk(self.dev.lib.aqg_d2h(self.dev.ctx, out.ctypes.data_as(C.c_void_p), C.c_void_p(self.ptr),
                                               C.c_size_t(out.nbytes)), "aqg_d2h")
        return out

    def free(self):
        if self.owned and self.ptr and self.dev.ctx:
            self.dev.lib.aqg_free(self.dev.ctx, C.c_void_p(self.ptr))
        self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class GroupBy:
    def __init__(self, dev, handle):
        self.dev, self.h = dev, handle
        dev._handles.add(self)

    @property
    def ngroups(self):
        return self.dev.lib.aqg_groupby_ngroups(self.h)

    @property
    def plan(self):
        """AQG_PLAN_* bits of the plan the last call through this handle took"""
        self.dev.lib.aqg_groupby_plan.restype = C.c_uint32
        return self.dev.lib.aqg_groupby_plan(self.h)

    def _view(self, fn, dtype, n):
        p = getattr(self.dev.lib, fn)(self.h)
        if not p:
            return None
        return DevBuf(self.dev, p, dtype, n, owned=False).to_host()

    def reversemap(self):
        return self._view("aqg_groupby_reversemap", np.uint32, self.dev.lib.aqg_groupby_nrows(self.h))

    def counts(self):
        return self._view("aqg_groupby_counts", np.uint32, self.ngroups)

    def first_rows(self):
        return self._view("aqg_groupby_first_rows", np.uint32, self.ngroups)

    def first_rows64(self):
        """global row id of every group's first row (handles of aqg_groupby_agg_sharded)"""
        return self._view("aqg_groupby_first_rows64", np.int64, self.ngroups)

    def keys_raw(self, k, elem_bytes):
        """key column k of every group as raw bytes, (ngroups, elem_bytes) uint8 (keys of any element type)"""
        out = self.dev.empty(max(1, self.ngroups * elem_bytes), np.uint8)
        self.dev._chk(self.dev.lib.aqg_groupby_keys(self.h, k, C.c_void_p(out.ptr)), "aqg_groupby_keys")
        self.dev.sync()
        return out.to_host()[:self.ngroups * elem_bytes].reshape(self.ngroups, elem_bytes)

    def keys(self, k, dtype):
        out = self.dev.empty(self.ngroups, dtype)
        self.dev._chk(self.dev.lib.aqg_groupby_keys(self.h, k, C.c_void_p(out.ptr)), "aqg_groupby_keys")
        self.dev.sync()
        return out.to_host()

    def result(self, j, op, val_tag):
        ot = self.dev.lib.aqg_reduce_out_dtype(op, val_tag)
        p = self.dev.lib.aqg_groupby_agg_result(self.h, j)
        return DevBuf(self.dev, p, TAG2NP[ot], self.ngroups, owned=False).to_host()

    def postproc(self):
        G, n = self.ngroups, self.dev.lib.aqg_groupby_nrows(self.h)
        off, rows = self.dev.empty(G + 1, np.uint32), self.dev.empty(max(n, 1), np.uint32)
        self.dev._chk(self.dev.lib.aqg_groupby_postproc(self.h, C.c_void_p(off.ptr), C.c_void_p(rows.ptr)), "aqg_groupby_postproc")
        self.dev.sync()
        return off.to_host(), rows.to_host()[:n]

    def destroy(self):
        if self.h and self.dev.ctx:
            self.dev.lib.aqg_groupby_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)


class Comm:
    """One communicator of the library's own exchange (include/aqg.h: aqg_comm_*).  `nccl_id`: the AQG_COMM_ID_BYTES bytes rank 0
    got from Comm.unique_id() -- RCCL; `allgather`: a Python callable (send_ptr, recv_ptr, nbytes, stream) -> 0 -- a caller-supplied
    transport (tests / one-GPU rehearsals)."""

    def __init__(self, dev, rank, world, nccl_id=None, allgather=None):
        self.dev, self.rank, self.world = dev, rank, world
        h = C.c_void_p()
        if allgather is not None:
            self._cb = ALLGATHER_FN(lambda user, send, recv, nbytes, stream: int(allgather(send, recv, nbytes, stream)))
            dev._chk(dev.lib.aqg_comm_init_custom(dev.ctx, rank, world, self._cb, None, C.byref(h)), "aqg_comm_init_custom")
        else:
            assert nccl_id is not None and len(nccl_id) == 128
            buf = (C.c_char * 128).from_buffer_copy(bytes(nccl_id))
            dev._chk(dev.lib.aqg_comm_init_rccl(dev.ctx, rank, world, buf, C.byref(h)), "aqg_comm_init_rccl")
        self.h = h

    @staticmethod
    def unique_id(lib=None):
        lib = lib or load_library()
        buf = (C.c_char * 128)()
        rc = lib.aqg_comm_unique_id(buf)
        if rc != 0:
            raise AqgError("aqg_comm_unique_id", rc)
        return bytes(buf.raw)

    def groupby_agg_sharded(self, keys, ops, vals, row_base, hint=0, gmax=0, handle=None):
        d = self.dev
        kd, dts, ptrs = d._keyargs(keys)
        vd = [d._dev(v) if v is not None else None for v in vals]
        vdt = (C.c_int * max(1, len(vd)))(*[(v.tag if v is not None else INT32) for v in vd])
        vp = (C.c_void_p * max(1, len(vd)))(*[(v.ptr if v is not None else None) for v in vd])
        opa = (C.c_int * max(1, len(ops)))(*ops)
        h = handle.h if handle is not None else C.c_void_p()
        d._chk(d.lib.aqg_groupby_agg_sharded(self.h, len(kd), dts, ptrs, len(ops), opa, vdt, vp, C.c_uint32(kd[0].n), C.c_uint64(row_base),
                                             C.c_uint32(hint), C.c_uint32(gmax), C.byref(h)), "aqg_groupby_agg_sharded")
        gb = handle if handle is not None else GroupBy(d, h)
        gb._keep = (kd, vd)
        return gb

    def prepare_groupby_agg_sharded(self, keys, ops, vals, row_base, hint=0, gmax=0):
        """groupby_agg_sharded with its argument arrays marshalled once (Device.prepare_groupby_agg's counterpart): returns a function
        that runs the same call again on the same device columns and result handle"""
        d = self.dev
        kd, dts, ptrs = d._keyargs(keys)
        vd = [d._dev(v) if v is not None else None for v in vals]
        vdt = (C.c_int * max(1, len(vd)))(*[(v.tag if v is not None else INT32) for v in vd])
        vp = (C.c_void_p * max(1, len(vd)))(*[(v.ptr if v is not None else None) for v in vd])
        opa = (C.c_int * max(1, len(ops)))(*ops)
        h = C.c_void_p()
        args = (self.h, len(kd), dts, ptrs, len(ops), opa, vdt, vp, C.c_uint32(kd[0].n), C.c_uint64(row_base), C.c_uint32(hint), C.c_uint32(gmax), C.byref(h))
        fn, chk = d.lib.aqg_groupby_agg_sharded, d._chk
        gb = GroupBy(d, h)
        gb._keep = (kd, vd, dts, ptrs, vdt, vp, opa)

        def run():
            rc = fn(*args)
            if rc != 0:
                chk(rc, "aqg_groupby_agg_sharded")
            return gb
        return run, gb

    def str_encode_sharded(self, strs):
        """global dictionary codes of this rank's strings (list of bytes): aqg_str_encode_sharded; returns (DevBuf of uint32 codes, global distinct count)"""
        d = self.dev
        bufs = [C.create_string_buffer(b) for b in strs]
        arr = (C.c_char_p * max(1, len(bufs)))(*[C.cast(b, C.c_char_p) for b in bufs])
        out = d.empty(max(1, len(bufs)), np.uint32)
        nd = C.c_uint32()
        d._chk(d.lib.aqg_str_encode_sharded(self.h, arr, C.c_uint32(len(bufs)), C.c_void_p(out.ptr), C.byref(nd)), "aqg_str_encode_sharded")
        out.n = len(bufs)
        return out, nd.value

    def reduce_sharded(self, op, x):
        """aqg_reduce over a column sharded by row range (`x`: this rank's rows); the whole column's result on every rank"""
        d = self.dev
        xd = d._dev(x)
        buf = (C.c_ubyte * 16)()
        d._chk(d.lib.aqg_reduce_sharded(self.h, op, xd.tag, C.c_void_p(xd.ptr), C.c_uint32(xd.n), buf), "aqg_reduce_sharded")
        ot = d.lib.aqg_reduce_out_dtype(op, xd.tag)
        v = np.frombuffer(bytes(buf), dtype=TAG2NP[ot], count=1)[0]
        if ot in (INT128, UINT128):
            lo, hi = int(v["lo"]), int(v["hi"])
            return (hi << 64) + lo if ot == INT128 else (hi << 64) | lo
        return v

    def corr_sharded(self, x, y):
        d = self.dev
        xd, yd = d._dev(x), d._dev(y)
        out = C.c_double()
        d._chk(d.lib.aqg_corr_sharded(self.h, xd.tag, C.c_void_p(xd.ptr), yd.tag, C.c_void_p(yd.ptr), C.c_uint32(xd.n), C.byref(out)), "aqg_corr_sharded")
        return out.value

    def scan_sharded(self, op, x, w=0):
        d = self.dev
        xd = d._dev(x)
        ot = d.lib.aqg_scan_out_dtype(op, xd.tag)
        out = d.empty(xd.n, TAG2NP[ot])
        d._chk(d.lib.aqg_scan_sharded(self.h, op, xd.tag, C.c_void_p(xd.ptr), C.c_uint32(xd.n), C.c_uint32(w), C.c_void_p(out.ptr)), "aqg_scan_sharded")
        d.sync()
        return out.to_host()

    def groupby_exchange(self, local, merge_ops, row_base, gmax=0, handle=None):
        """the exchange alone over an existing shard table (aqg_groupby_exchange): partial p = aggregate p of `local`"""
        d = self.dev
        opa = (C.c_int * max(1, len(merge_ops)))(*merge_ops)
        h = handle.h if handle is not None else C.c_void_p()
        d._chk(d.lib.aqg_groupby_exchange(self.h, local.h, len(merge_ops), opa, C.c_uint64(row_base), C.c_uint32(gmax), C.byref(h)), "aqg_groupby_exchange")
        return handle if handle is not None else GroupBy(d, h)

    def destroy(self):
        if self.h and self.dev.ctx:
            self.dev.lib.aqg_comm_destroy(self.h)
        self.h = None


class ThreadRanks:
    """`world` ranks as THREADS of one process, each with its own context (and stream) on the same GPU, and an all-gather made of
    device copies and a barrier: drives the C code of the sharded group-by end to end where only one GPU is there (RCCL refuses two
    ranks on one device).  run(fn) calls fn(rank, dev, comm) on every rank and returns the results in rank order."""

    def __init__(self, world, device=0):
        import threading
        self.world = world
        self.devs = [Device(device) for _ in range(world)]
        self.barrier = threading.Barrier(world)
        self.slots = [None] * world
        self.comms = [Comm(self.devs[r], r, world, allgather=self._make(r)) for r in range(world)]

    def _make(self, rank):
        def gather(send, recv, nbytes, stream):
            dev = self.devs[rank]
            dev.sync()                                   # the pack kernel behind `send` has run
            self.slots[rank] = send
            self.barrier.wait()
            for r in range(self.world):
                dev._chk(dev.lib.aqg_d2d(dev.ctx, C.c_void_p(recv + r * nbytes), C.c_void_p(self.slots[r]), C.c_size_t(nbytes)), "aqg_d2d")
            dev.sync()
            self.barrier.wait()                          # nobody reuses its send buffer before everybody has copied it
            return 0
        return gather

    def run(self, fn):
        import threading
        out, err = [None] * self.world, [None] * self.world
        def body(r):
            try:
                out[r] = fn(r, self.devs[r], self.comms[r])
            except BaseException as e:                   # noqa: BLE001 -- reported to the caller below
                err[r] = e
                self.barrier.abort()
        th = [threading.Thread(target=body, args=(r,)) for r in range(self.world)]
        for t in th: t.start()
        for t in th: t.join()
        for e in err:
            if e is not None and not isinstance(e, __import__("threading").BrokenBarrierError):
                raise e
        for e in err:
            if e is not None:
                raise e
        return out

    def close(self):
        for c in self.comms: c.destroy()
        for d in self.devs: d.close()


class Device:
    """One aqg context on one GPU.  numpy-level wrappers upload, run the HIP path, download."""

    def __init__(self, device=0, stream=None):
        self.lib = load_library()
        ctx = C.c_void_p()
        rc = self.lib.aqg_ctx_create(int(device), C.c_void_p(stream) if stream else None, C.byref(ctx))
        if rc != 0:
            raise AqgError("aqg_ctx_create (no MI355X visible? the HIP path has no CPU fallback)", rc)
        self.ctx = ctx
        self.stream = int(stream) if stream else 0    # 0: the library made its own stream
        self._handles = weakref.WeakSet()

    def close(self):
        if self.ctx:
            for h in list(self._handles):   # handles hold device memory of this context: release them first
                h.destroy()
            self.lib.aqg_ctx_destroy(self.ctx)
            self.ctx = None

    def _chk(self, rc, what):
        if rc != 0:
            raise AqgError(what, rc, self.lib.aqg_last_error(self.ctx).decode())

    def sync(self):
        self._chk(self.lib.aqg_sync(self.ctx), "aqg_sync")

    # -- memory
    def empty(self, n, dtype):
        dtype = np.dtype(dtype)
        p = C.c_void_p()
        self._chk(self.lib.aqg_malloc(self.ctx, C.c_size_t(max(int(n), 1) * dtype.itemsize + 64), C.byref(p)), "aqg_malloc")
        return DevBuf(self, p.value, dtype, n)

    def to_device(self, a):
        a = np.ascontiguousarray(a)
        if a.dtype == np.bool_:
            a = a.astype(np.uint8)
        b = self.empty(a.size, a.dtype)
        if a.size:
            self._chk(self.lib.aqg_h2d(self.ctx, C.c_void_p(b.ptr), a.ctypes.data_as(C.c_void_p), C.c_size_t(a.nbytes)), "aqg_h2d")
        return b

    def _dev(self, a):
        return a if isinstance(a, DevBuf) else self.to_device(a)

    def key_col(self, tag, data):
        """a key column of a type that is not a numpy scalar type: DATE / TIME / TIMESTAMP as an (n, bytes) uint8 array, FLOAT /
        DOUBLE / 128-bit columns as their arrays, STR as a list of bytes objects (encoded to uint32 codes by aqg_str_encode)"""
        if tag == STR:
            bufs = [C.create_string_buffer(b) for b in data]
            arr = (C.c_char_p * max(1, len(bufs)))(*[C.cast(b, C.c_char_p) for b in bufs])
            out = self.empty(max(1, len(bufs)), np.uint32)
            nd = C.c_uint32()
            self._chk(self.lib.aqg_str_encode(self.ctx, arr, C.c_uint32(len(bufs)), C.c_void_p(out.ptr), C.byref(nd)), "aqg_str_encode")
            out.n = len(bufs)
            out.ndistinct = nd.value
            return out
        a = np.ascontiguousarray(data)
        raw = self.to_device(a.reshape(-1).view(np.uint8))
        buf = DevBuf(self, raw.ptr, np.uint8, a.shape[0], owned=False)
        buf._raw, buf._tag = raw, tag
        return buf

    # -- generators
    def col_pin(self, a):
        """device mirror of a borrowed host column (aqg_col_pin: asynchronous, stream-ordered upload); `a` must stay alive"""
        assert a.flags["C_CONTIGUOUS"]
        d = C.c_void_p()
        self._chk(self.lib.aqg_col_pin(self.ctx, C.c_void_p(a.ctypes.data), C.c_size_t(a.nbytes), C.byref(d)), "aqg_col_pin")
        buf = DevBuf(self, d.value, a.dtype, a.size, owned=False)
        buf._host = a
        return buf

    def col_pin_last(self):
        """(registered, staged, pageable) chunk counts of the most recent col_pin upload"""
        a, b, c = C.c_uint32(), C.c_uint32(), C.c_uint32()
        self._chk(self.lib.aqg_col_pin_last(self.ctx, C.byref(a), C.byref(b), C.byref(c)), "aqg_col_pin_last")
        return a.value, b.value, c.value

    def col_unpin_all(self):
        self._chk(self.lib.aqg_col_unpin_all(self.ctx), "aqg_col_unpin_all")

    def gen_column(self, col, seed, row_base, n, n_total, K, out=None):
        dt = np.float32 if col == 8 else np.int32
        out = out or self.empty(n, dt)
        self._chk(self.lib.aqg_gen_column(self.ctx, col, C.c_uint64(seed), C.c_uint64(row_base), C.c_uint32(n),
                                          C.c_uint64(n_total), C.c_uint32(K), C.c_void_p(out.ptr)), "aqg_gen_column")
        return out

    # -- element-wise
    def ewise(self, op, l, r, ot=None, keep=False, out=None):
        l_vec = isinstance(l, DevBuf) or np.ndim(l) > 0
        r_vec = isinstance(r, DevBuf) or np.ndim(r) > 0
        kind = VEC_VEC if (l_vec and r_vec) else (VEC_SCALAR if l_vec else SCALAR_VEC)
        ld = self._dev(l) if l_vec else None
        rd = self._dev(r) if r_vec else None
        ls = None if l_vec else np.atleast_1d(np.asarray(l))
        rs = None if r_vec else np.atleast_1d(np.asarray(r))
        lt = ld.tag if l_vec else tag_of(ls)
        rt = rd.tag if r_vec else tag_of(rs)
        n = ld.n if l_vec else rd.n
        if ot is None:
            ot = self.lib.aqg_ewise_out_dtype(op, lt, rt)
        if ot == ERROR:
            raise AqgError("aqg_ewise_out_dtype", ERROR)
        out = out if out is not None else self.empty(n, TAG2NP[ot])
        lp = C.c_void_p(ld.ptr) if l_vec else ls.ctypes.data_as(C.c_void_p)
        rp = C.c_void_p(rd.ptr) if r_vec else rs.ctypes.data_as(C.c_void_p)
        self._chk(self.lib.aqg_ewise(self.ctx, op, kind, lt, lp, rt, rp, ot, C.c_void_p(out.ptr), C.c_uint32(n)), "aqg_ewise")
        return out if keep else out.to_host()

    def unary(self, op, x, param=0):
        xd = self._dev(x)
        ot = DOUBLE if op == 0 else xd.tag
        out = self.empty(xd.n, TAG2NP[ot])
        self._chk(self.lib.aqg_unary(self.ctx, op, xd.tag, C.c_void_p(xd.ptr), C.c_uint32(xd.n), C.c_uint32(param), ot,
                                     C.c_void_p(out.ptr)), "aqg_unary")
        return out.to_host()

    # -- reductions
    def reduce(self, op, x):
        xd = self._dev(x)
        buf = (C.c_ubyte * 16)()
        self._chk(self.lib.aqg_reduce(self.ctx, op, xd.tag, C.c_void_p(xd.ptr), C.c_uint32(xd.n), buf), "aqg_reduce")
        ot = self.lib.aqg_reduce_out_dtype(op, xd.tag)
        v = np.frombuffer(bytes(buf), dtype=TAG2NP[ot], count=1)[0]
        if ot in (INT128, UINT128):
            lo, hi = int(v["lo"]), int(v["hi"])
            return (hi << 64) + lo if ot == INT128 else (hi << 64) | lo
        return v

    def corr(self, x, y):
        xd, yd = self._dev(x), self._dev(y)
        out = C.c_double()
        self._chk(self.lib.aqg_corr(self.ctx, xd.tag, C.c_void_p(xd.ptr), yd.tag, C.c_void_p(yd.ptr), C.c_uint32(xd.n),
                                    C.byref(out)), "aqg_corr")
        return out.value

    # -- scans
    def scan(self, op, x, w=0, keep=False, out=None):
        xd = self._dev(x)
        ot = self.lib.aqg_scan_out_dtype(op, xd.tag)
        out = out if out is not None else self.empty(xd.n, TAG2NP[ot])
        self._chk(self.lib.aqg_scan(self.ctx, op, xd.tag, C.c_void_p(xd.ptr), C.c_uint32(xd.n), C.c_uint32(w),
                                    C.c_void_p(out.ptr)), "aqg_scan")
        return out if keep else out.to_host()

    def scan_resume(self, op, x, carry, row_offset, keep=False, out=None):
        """sums / avgs of one row-range shard: `carry` = sum of every earlier row (python int for integer columns, float for
        floating ones), `row_offset` = number of earlier rows (aqg_scan_resume)"""
        xd = self._dev(x)
        ot = self.lib.aqg_scan_out_dtype(op, xd.tag)
        out = out if out is not None else self.empty(xd.n, TAG2NP[ot])
        if np.issubdtype(xd.dtype, np.floating):
            raw = np.array([float(carry), 0.0], dtype=np.float64).tobytes()
        else:
            raw = (int(carry) & ((1 << 128) - 1)).to_bytes(16, "little")
        buf = C.create_string_buffer(raw, 16)
        self._chk(self.lib.aqg_scan_resume(self.ctx, op, xd.tag, C.c_void_p(xd.ptr), C.c_uint32(xd.n), buf, C.c_uint64(int(row_offset)),
                                           C.c_void_p(out.ptr)), "aqg_scan_resume")
        return out if keep else out.to_host()

    # -- gather / filter
    def gather(self, x, idx):
        xd, idd = self._dev(x), self._dev(np.ascontiguousarray(idx, dtype=np.uint32) if not isinstance(idx, DevBuf) else idx)
        out = self.empty(idd.n, xd.dtype)
        self._chk(self.lib.aqg_gather(self.ctx, xd.tag, C.c_void_p(xd.ptr), C.c_void_p(idd.ptr), C.c_uint32(idd.n),
                                      C.c_void_p(out.ptr)), "aqg_gather")
        return out.to_host()

    def compact(self, x, mask):
        xd = self._dev(x)
        md = self._dev(np.ascontiguousarray(mask).astype(np.uint8) if not isinstance(mask, DevBuf) else mask)
        out = self.empty(xd.n, xd.dtype)
        m = C.c_uint32()
        self._chk(self.lib.aqg_compact(self.ctx, xd.tag, C.c_void_p(xd.ptr), C.c_void_p(md.ptr), C.c_uint32(xd.n),
                                       C.c_void_p(out.ptr), C.byref(m)), "aqg_compact")
        return out.to_host()[:m.value].copy()

    def mask_to_index(self, mask):
        md = self._dev(np.ascontiguousarray(mask).astype(np.uint8) if not isinstance(mask, DevBuf) else mask)
        out = self.empty(md.n, np.uint32)
        m = C.c_uint32()
        self._chk(self.lib.aqg_mask_to_index(self.ctx, C.c_void_p(md.ptr), C.c_uint32(md.n), C.c_void_p(out.ptr), C.byref(m)),
                  "aqg_mask_to_index")
        return out.to_host()[:m.value].copy()

    # -- group by
    def _keyargs(self, keys):
        kd = [self._dev(k) for k in keys]
        dts = (C.c_int * len(kd))(*[k.tag for k in kd])
        ptrs = (C.c_void_p * len(kd))(*[k.ptr for k in kd])
        return kd, dts, ptrs

    def groupby_build(self, keys, hint=0):
        kd, dts, ptrs = self._keyargs(keys)
        h = C.c_void_p()
        self._chk(self.lib.aqg_groupby_build(self.ctx, len(kd), dts, ptrs, C.c_uint32(kd[0].n), C.c_uint32(hint), C.byref(h)),
                  "aqg_groupby_build")
        gb = GroupBy(self, h)
        gb._keep = kd
        return gb

    def groupby_agg(self, keys, ops, vals, hint=0, handle=None):
        kd, dts, ptrs = self._keyargs(keys)
        vd = [self._dev(v) if v is not None else None for v in vals]
        vdt = (C.c_int * len(vd))(*[(v.tag if v is not None else INT32) for v in vd])
        vp = (C.c_void_p * len(vd))(*[(v.ptr if v is not None else None) for v in vd])
        opa = (C.c_int * len(ops))(*ops)
        h = handle.h if handle is not None else C.c_void_p()
        self._chk(self.lib.aqg_groupby_agg(self.ctx, len(kd), dts, ptrs, len(ops), opa, vdt, vp, C.c_uint32(kd[0].n),
                                           C.c_uint32(hint), C.byref(h)), "aqg_groupby_agg")
        gb = handle if handle is not None else GroupBy(self, h)
        gb._keep = (kd, vd)            # keys of non-integer columns are fetched from the caller's column: keep the uploads alive
        gb._val_tags = [v.tag if v is not None else INT32 for v in vd]
        gb._ops = list(ops)
        return gb

    def prepare_groupby_agg(self, keys, ops, vals, hint=0):
        """the same call as groupby_agg with its argument arrays marshalled once: returns a function that runs it again on the
        same device columns and handle (a prepared statement; the per-call Python work drops to one foreign call)"""
        kd, dts, ptrs = self._keyargs(keys)
        vd = [self._dev(v) if v is not None else None for v in vals]
        vdt = (C.c_int * len(vd))(*[(v.tag if v is not None else INT32) for v in vd])
        vp = (C.c_void_p * len(vd))(*[(v.ptr if v is not None else None) for v in vd])
        opa = (C.c_int * len(ops))(*ops)
        h = C.c_void_p()
        args = (self.ctx, len(kd), dts, ptrs, len(ops), opa, vdt, vp, C.c_uint32(kd[0].n), C.c_uint32(hint), C.byref(h))
        fn, chk = self.lib.aqg_groupby_agg, self._chk
        gb = GroupBy(self, h)                      # the handle is created by the first run
        gb._val_tags = [v.tag if v is not None else INT32 for v in vd]
        gb._ops = list(ops)
        gb._keep = (kd, vd, dts, ptrs, vdt, vp, opa)

        def run():
            rc = fn(*args)
            if rc != 0:
                chk(rc, "aqg_groupby_agg")
            return gb
        return run, gb

    def join_groupby_sum(self, dim_keys, dim_vals, fact_fk, group_keys, fact_vals, hint=0, handle=None):
        """fact JOIN dim ON fk = key, then sum(val * w) by group key -- one fused pass (aqg_join_groupby_sum)"""
        dk, dv, fk, gk, fv = (self._dev(a) for a in (dim_keys, dim_vals, fact_fk, group_keys, fact_vals))
        h = handle.h if handle is not None else C.c_void_p()
        self._chk(self.lib.aqg_join_groupby_sum(self.ctx, dk.tag, C.c_void_p(dk.ptr), dv.tag, C.c_void_p(dv.ptr), C.c_uint32(dk.n),
                                                C.c_void_p(fk.ptr), gk.tag, C.c_void_p(gk.ptr), fv.tag, C.c_void_p(fv.ptr), C.c_uint32(fk.n),
                                                C.c_uint32(hint), C.byref(h)), "aqg_join_groupby_sum")
        gb = handle if handle is not None else GroupBy(self, h)
        gb._keep = (dk, dv, fk, gk, fv)
        return gb
    def groupby_pack(self, gb, agg_index, gmax, out_ptr):
        """exchange payload of a shard's group table: (gmax + 1) int64 pairs at device address `out_ptr` (aqg_groupby_pack)"""
        self._chk(self.lib.aqg_groupby_pack(gb.h, agg_index, C.c_uint32(gmax), C.c_void_p(out_ptr)), "aqg_groupby_pack")
    def groupby_merge_packed(self, gathered_ptr, world, gmax, key_tag, op, handle=None):
        """merge the gathered payloads of `world` shards (aqg_groupby_merge_packed)"""
        h = handle.h if handle is not None else C.c_void_p()
        self._chk(self.lib.aqg_groupby_merge_packed(self.ctx, C.c_void_p(gathered_ptr), C.c_uint32(world), C.c_uint32(gmax), key_tag, op, C.byref(h)),
                  "aqg_groupby_merge_packed")
        return handle if handle is not None else GroupBy(self, h)
    def grouped_reduce(self, gb, op, x):
        xd = self._dev(x)
        ot = self.lib.aqg_reduce_out_dtype(op, xd.tag)
        out = self.empty(gb.ngroups, TAG2NP[ot])
        self._chk(self.lib.aqg_grouped_reduce(self.ctx, gb.h, op, xd.tag, C.c_void_p(xd.ptr), C.c_void_p(out.ptr)), "aqg_grouped_reduce")
        self.sync()
        return out.to_host()

    # -- per-group scans / windows / two-column aggregates over the flat layout of a build (include/aqg.h)
    def group_offsets(self, gb):
        self.lib.aqg_groupby_offsets.restype = C.c_void_p
        p = self.lib.aqg_groupby_offsets(gb.h)
        assert p, "aqg_groupby_offsets"
        return DevBuf(self, p, np.uint32, gb.ngroups + 1, owned=False).to_host()

    def grouped_flatten(self, gb, x, keep=False):
        xd = self._dev(x)
        out = self.empty(xd.n, xd.dtype)
        self._chk(self.lib.aqg_grouped_flatten(self.ctx, gb.h, xd.tag, C.c_void_p(xd.ptr), C.c_void_p(out.ptr)), "aqg_grouped_flatten")
        return out if keep else out.to_host()

    def grouped_scan(self, gb, op, x, w=0, flat=False, keep=False, out=None):
        """per-group scan of a column: `x` in row layout (flat=False: flatten + scan in one call) or already flat"""
        xd = self._dev(x)
        ot = self.lib.aqg_scan_out_dtype(op, xd.tag)
        out = out if out is not None else self.empty(xd.n, TAG2NP[ot])
        fn = self.lib.aqg_grouped_scan_flat if flat else self.lib.aqg_grouped_scan
        self._chk(fn(self.ctx, gb.h, op, xd.tag, C.c_void_p(xd.ptr), C.c_uint32(w), C.c_void_p(out.ptr)), "aqg_grouped_scan")
        return out if keep else out.to_host()

    def grouped_reduce_flat(self, gb, op, xflat):
        xd = self._dev(xflat)
        ot = self.lib.aqg_reduce_out_dtype(op, xd.tag)
        out = self.empty(gb.ngroups, TAG2NP[ot])
        self._chk(self.lib.aqg_grouped_reduce_flat(self.ctx, gb.h, op, xd.tag, C.c_void_p(xd.ptr), C.c_void_p(out.ptr)), "aqg_grouped_reduce_flat")
        self.sync()
        return out.to_host()

    def grouped_corr(self, gb, x, y):
        xd, yd = self._dev(x), self._dev(y)
        out = self.empty(gb.ngroups, np.float64)
        self._chk(self.lib.aqg_grouped_corr(self.ctx, gb.h, xd.tag, C.c_void_p(xd.ptr), yd.tag, C.c_void_p(yd.ptr), C.c_void_p(out.ptr)), "aqg_grouped_corr")
        self.sync()
        return out.to_host()

    # -- join
    def join_pairs(self, build, probe):
        bd, pd = self._dev(build), self._dev(probe)
        m = C.c_uint64()
        self._chk(self.lib.aqg_join_count(self.ctx, bd.tag, C.c_void_p(bd.ptr), C.c_uint32(bd.n), C.c_void_p(pd.ptr),
                                          C.c_uint32(pd.n), C.byref(m)), "aqg_join_count")
        pr, br = self.empty(m.value, np.uint32), self.empty(m.value, np.uint32)
        self._chk(self.lib.aqg_join_pairs(self.ctx, bd.tag, C.c_void_p(bd.ptr), C.c_uint32(bd.n), C.c_void_p(pd.ptr),
                                          C.c_uint32(pd.n), C.c_void_p(pr.ptr), C.c_void_p(br.ptr), C.c_uint64(m.value),
                                          C.byref(m)), "aqg_join_pairs")
        return pr.to_host(), br.to_host()

    def join_count(self, build, probe):
        """number of matching (probe row, build row) pairs, in 64 bits (aqg_join_count)"""
        bd, pd = self._dev(build), self._dev(probe)
        m = C.c_uint64()
        self._chk(self.lib.aqg_join_count(self.ctx, bd.tag, C.c_void_p(bd.ptr), C.c_uint32(bd.n), C.c_void_p(pd.ptr),
                                          C.c_uint32(pd.n), C.byref(m)), "aqg_join_count")
        return m.value

    def join_lookup(self, build, probe):
        bd, pd = self._dev(build), self._dev(probe)
        out = self.empty(pd.n, np.uint32)
        self._chk(self.lib.aqg_join_lookup(self.ctx, bd.tag, C.c_void_p(bd.ptr), C.c_uint32(bd.n), C.c_void_p(pd.ptr),
                                           C.c_uint32(pd.n), C.c_void_p(out.ptr)), "aqg_join_lookup")
        return out.to_host()

    # -- timing
    def timer_start(self):
        self._chk(self.lib.aqg_timer_start(self.ctx), "aqg_timer_start")

    def last_kernel_ms(self):
        ms = C.c_float()
        self._chk(self.lib.aqg_last_kernel_ms(self.ctx, C.byref(ms)), "aqg_last_kernel_ms")
        return ms.value

    def timer_stop_ms(self):
        ms = C.c_float()
        self._chk(self.lib.aqg_timer_stop_ms(self.ctx, C.byref(ms)), "aqg_timer_stop_ms")
        return ms.value
