// Module prologue: every generated out.cpp starts with this file (the reference concatenates its header.cxx in front of the
// emitted functions, engine/storage.py:312-331; reference header.cxx:1-13).  Same two includes and the same exported
// __AQ_Init_GC__ the host calls right after dlopen (server/server.cpp:152-161), plus TWO more exported hooks: __AQ_Result_Changed__ (below) and
// __AQ_End_Session__, which the host calls before dlclose (aquery_host; server.cpp:604-609 is where the reference unloads).
// A module keeps device state of its own -- mirrors of the borrowed columns, the groupings HashTableFactory::get made -- in the
// header-only runtime inside this DSO; the hook releases it (a host that does not know the hook still gets it released by the
// static destructors that run at dlclose).
#include "./server/libaquery.h"
#include "./server/gc.h"
__AQEXPORT__(void) __AQ_Init_GC__(Context* cxt) {
    GC::gc_handle = static_cast<GC*>(cxt->gc);
    GC::scratch_space = nullptr;
}
// The data source has replaced its result set (a new 'Q' statement: server/server.cpp:285-295 -> monetdbe_query frees the previous
// result): the borrowed column pointers the module saw before are dead, and their addresses may come back holding other data -- the
// device mirrors keyed by them are dropped.  Called by the host after every statement that produces a result set.
__AQEXPORT__(void) __AQ_Result_Changed__(Context* cxt) {
    (void)cxt;
    aq::dev::Runtime::get().drop_pins();
}
__AQEXPORT__(void) __AQ_End_Session__(Context* cxt) {
    (void)cxt;
    aq::dev::Runtime::get().release_session();
    aq::dev::Runtime::get().drop_pins();
}
