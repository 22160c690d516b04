// ewise_i64.hip -- the aqg_ewise kernels whose arithmetic runs in int64_t (see ewise_impl.hpp)
#include "ewise_impl.hpp"
template int aqgew::dispatch_ot<int64_t>(aqg_ctx*, int, int, int, int, const void*, int, const void*, void*, uint32_t, int);
