// forwarding header: generated code includes "./server/table_ext_monetdb.hpp" (engine/ast.py:506)
#pragma once
#include "../aquery/libaquery.h"
