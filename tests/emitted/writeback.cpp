// Write-back of a group-by result into the data source and reading it back (tests/q4.a:20-26 of the reference):
//   CREATE TABLE ticks2(ID INT, max REAL, min REAL)
//   INSERT INTO ticks2 SELECT ID AS ID, max(ratios(endofdayprice)) AS max, min(ratios(endofdayprice)) AS min FROM ticks GROUP BY ID
//   SELECT ID, max, min FROM ticks2
// The INSERT ... SELECT is a special group-by (vector function inside an aggregate): emitted shape of engine/ast.py:620-794, ending in
// `out->monetdb_append_table(cxt->curr_server, "ticks2")` (engine/ast.py:507) instead of a print.
#include "header.cxx"
#include "./server/monetdb_conn.h"
#include "./server/aggregations.h"
#include "./server/hasher.h"
#include "./server/table_ext_monetdb.hpp"

__AQEXPORT__(int) dll_5wB7tq(Context* cxt) {
	using namespace std;
	using namespace types;
	auto server = static_cast<DataSource*>(cxt->curr_server);
auto len_1a = server->cnt;
auto ID_2b = ColRef<int>(len_1a, server->getCol(0, types::Type_t::AINT32));
auto endofdayprice_3c = ColRef<int>(len_1a, server->getCol(1, types::Type_t::AINT32));
const char* names_4d[] = {"ID", "max", "min"};
auto out_5e = new TableInfo<int,value_type<decays<decltype(max(ratios(endofdayprice_3c)))>>,value_type<decays<decltype(min(ratios(endofdayprice_3c)))>>>("out_5e", names_4d);
decltype(auto) col_6f = out_5e->get_col<0>();
decltype(auto) col_7g = out_5e->get_col<1>();
decltype(auto) col_8h = out_5e->get_col<2>();
uint32_t len_9i = ID_2b.size;
typedef record<decays<decltype(ID_2b)>::value_t> record_type10j;
auto g11k = HashTableFactory<record_type10j, transTypes<record_type10j, hasher>>::get<decays<decltype(ID_2b)>>(ID_2b);
auto sz_g11k = g11k.size;
auto vecs_12l = g11k.values;
col_6f.resize(sz_g11k);
col_7g.resize(sz_g11k);
col_8h.resize(sz_g11k);
GC::scratch_space = GC::gc_handle ? &(GC::gc_handle->scratch) : nullptr;
for (uint32_t i13 = 0; i13 < sz_g11k; ++i13) {
auto &key_14m = (*g11k.keys)[i13];
auto &val_15n = vecs_12l[i13];
col_6f[i13] = (get<0>(key_14m));

col_7g[i13] = (max(ratios(endofdayprice_3c[val_15n])));

col_8h[i13] = (min(ratios(endofdayprice_3c[val_15n])));

GC::scratch_space->release();
}
GC::scratch_space = nullptr;
out_5e->monetdb_append_table(cxt->curr_server, "ticks2");
puts("done.");
return 0;
}

// a plain projection written back: SELECT ID, endofdayprice - mins(endofdayprice) AS gain INTO gains FROM ticks ASSUMING ASC date
// (whole columns produced on the device: the write-back fetches them together, asynchronously)
__AQEXPORT__(int) dll_8kD2uv(Context* cxt) {
	using namespace std;
	using namespace types;
	auto server = static_cast<DataSource*>(cxt->curr_server);
auto len_1a = server->cnt;
auto ID_2b = ColRef<int>(len_1a, server->getCol(0, types::Type_t::AINT32));
auto endofdayprice_3c = ColRef<int>(len_1a, server->getCol(1, types::Type_t::AINT32));
const char* names_4d[] = {"ID", "gain", "avg3"};
auto out_5e = new TableInfo<int,value_type<decays<decltype((endofdayprice_3c - mins(endofdayprice_3c)))>>,value_type<decays<decltype(avgw(3, endofdayprice_3c))>>>("out_5e", names_4d);
out_5e->get_col<0>().initfrom(ID_2b, "ID");
out_5e->get_col<1>().initfrom((endofdayprice_3c - mins(endofdayprice_3c)), "gain");
out_5e->get_col<2>().initfrom(avgw(3, endofdayprice_3c), "avg3");
out_5e->monetdb_append_table(cxt->curr_server, "gains");
puts("done.");
return 0;
}
