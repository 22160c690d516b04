// forwarding header: generated code includes "./server/gc.h" (header.cxx:1, engine/storage.py:150)
#pragma once
#include "../aquery/gc.h"
