// ewise_f64.hip -- the aqg_ewise kernels whose arithmetic runs in double (see ewise_impl.hpp)
#include "ewise_impl.hpp"
template int aqgew::dispatch_ot<double>(aqg_ctx*, int, int, int, int, const void*, int, const void*, void*, uint32_t, int);
