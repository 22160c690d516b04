// module prologue in the shape of the reference's header.cxx:1-13 (every generated out.cpp starts with it)
#include "./server/libaquery.h"
#include "./server/gc.h"
__AQEXPORT__(void) __AQ_Init_GC__(Context* cxt) {
    GC::gc_handle = static_cast<GC*>(cxt->gc);
    GC::scratch_space = nullptr;
}
