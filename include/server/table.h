// forwarding header: generated code includes "./server/table.h" (header.cxx:1, engine/storage.py:150)
#pragma once
#include "../aquery/table.h"
