// forwarding header: generated code includes "./server/aggregations.h" (header.cxx:1, engine/storage.py:150)
#pragma once
#include "../aquery/aggregations.h"
