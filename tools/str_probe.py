"""scratch: aqg_str_encode of 1e8 short strings, device dictionary vs the host map (AQG_STR_HOST=1 in the environment)"""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import aquery2_amd
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
npool = int(float(sys.argv[2])) if len(sys.argv) > 2 else 1_000_000
contents = [b"id%09d" % (i * 7919 % 1_000_000_007) for i in range(npool)]
bufs = [C.create_string_buffer(c) for c in contents]
addrs = np.array([C.addressof(b) for b in bufs], dtype=np.uint64)
rng = np.random.default_rng(1)
ptrs = np.ascontiguousarray(addrs[rng.integers(0, npool, n)])
gpu = aquery2_amd.Device(0)
out = gpu.empty(n, np.uint32)
nd = C.c_uint32()
for rep in range(2):
    t0 = time.perf_counter()
    gpu._chk(gpu.lib.aqg_str_encode(gpu.ctx, C.c_void_p(ptrs.ctypes.data), C.c_uint32(n), C.c_void_p(out.ptr), C.byref(nd)), "aqg_str_encode")
    dt = time.perf_counter() - t0
    print(f"{'host map' if os.environ.get('AQG_STR_HOST') else 'device dictionary'}: {n:.0e} strings, {nd.value} distinct: {dt:.3f} s = {n / dt / 1e6:.1f} M rows/s", flush=True)
print("checksum", int(out.to_host().astype(np.uint64).sum()))
