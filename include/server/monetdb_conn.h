// forwarding header: the SQL back-end connector is out of scope; DataSource lives in libaquery.h
#pragma once
#include "../aquery/libaquery.h"
