// forwarding header: generated code includes "./server/libaquery.h" (header.cxx:1, engine/storage.py:150)
#pragma once
#include "../aquery/libaquery.h"
