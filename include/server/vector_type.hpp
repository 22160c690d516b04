// forwarding header: generated code includes "./server/vector_type.hpp" (header.cxx:1, engine/storage.py:150)
#pragma once
#include "../aquery/vector_type.hpp"
