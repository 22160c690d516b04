// forwarding header: generated code includes "./server/types.h" (header.cxx:1, engine/storage.py:150)
#pragma once
#include "../aquery/types.h"
