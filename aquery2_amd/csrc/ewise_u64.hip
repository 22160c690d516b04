// ewise_u64.hip -- the aqg_ewise kernels whose arithmetic runs in uint64_t (see ewise_impl.hpp)
#include "ewise_impl.hpp"
template int aqgew::dispatch_ot<uint64_t>(aqg_ctx*, int, int, int, int, const void*, int, const void*, void*, uint32_t, int);
