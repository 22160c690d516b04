// Result egress parity: the shapes of print_shapes.inc printed by include/aquery's TableInfo::print / printall (the same source text is
// printed by the reference inside oracle/ref_harness.cpp -> tests/golden/print_shapes.txt).  Host-only: nothing here touches the GPU.
#include "header.cxx"
#include "./server/aggregations.h"
#include "./server/hasher.h"
#include "print_shapes.inc"
int main() {
    Context* cxt = new Context();
    __AQ_Init_GC__(cxt);
    aq_print_shapes();
    return 0;
}
