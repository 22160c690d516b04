// forwarding header: generated code includes "./server/hasher.h" (header.cxx:1, engine/storage.py:150)
#pragma once
#include "../aquery/hasher.h"
