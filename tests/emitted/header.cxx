// the product's module prologue (include/header.cxx), found the way generated code includes it
#include "../../include/header.cxx"
