// ewise_i32.hip -- the aqg_ewise kernels whose arithmetic runs in int32_t (see ewise_impl.hpp)
#include "ewise_impl.hpp"
template int aqgew::dispatch_ot<int32_t>(aqg_ctx*, int, int, int, int, const void*, int, const void*, void*, uint32_t, int);
