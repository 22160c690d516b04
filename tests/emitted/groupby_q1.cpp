// Emitted shape of tests/q1.sql  `SELECT sum(c), b, d FROM testq1 GROUP BY a,b,d` as the special group-by of HEAD
// (engine/ast.py:620-794): HashTableFactory front door, scalar aggregates per group inside the scratch arena,
// plus count(*) (`val.size`) and an avg column.  ORDER BY runs in the SQL engine in the hybrid design and is not part
// of the module.
#include "header.cxx"
#include "./server/monetdb_conn.h"
#include "./server/aggregations.h"
#include "./server/hasher.h"

__AQEXPORT__(int) dll_3kR9pQ(Context* cxt) {
	using namespace std;
	using namespace types;
	auto server = static_cast<DataSource*>(cxt->curr_server);
	auto timer = chrono::high_resolution_clock::now();
auto len_5Zb1 = server->cnt;
auto a_1x = ColRef<int>(len_5Zb1, server->getCol(0, types::Type_t::AINT32));
auto b_2y = ColRef<int>(len_5Zb1, server->getCol(1, types::Type_t::AINT32));
auto c_3z = ColRef<int>(len_5Zb1, server->getCol(2, types::Type_t::AINT32));
auto d_4w = ColRef<int>(len_5Zb1, server->getCol(3, types::Type_t::AINT32));
const char* names_7q[] = {"sumc", "b", "d", "cnt", "avgc"};
auto out_8r = new TableInfo<value_type<decays<decltype(sum(c_3z))>>,int,int,int,double>("out_8r", names_7q);
decltype(auto) col_a1 = out_8r->get_col<0>();
decltype(auto) col_a2 = out_8r->get_col<1>();
decltype(auto) col_a3 = out_8r->get_col<2>();
decltype(auto) col_a4 = out_8r->get_col<3>();
decltype(auto) col_a5 = out_8r->get_col<4>();
uint32_t len_b1 = a_1x.size;
typedef record<decays<decltype(a_1x)>::value_t,decays<decltype(b_2y)>::value_t,decays<decltype(d_4w)>::value_t> record_typeXq;
auto gK3 = HashTableFactory<record_typeXq, transTypes<record_typeXq, hasher>>::get<decays<decltype(a_1x)>, decays<decltype(b_2y)>, decays<decltype(d_4w)>>(a_1x, b_2y, d_4w);
auto sz_gK3 = gK3.size;
auto vecs_c1 = gK3.values;
col_a1.resize(sz_gK3);
col_a2.resize(sz_gK3);
col_a3.resize(sz_gK3);
col_a4.resize(sz_gK3);
col_a5.resize(sz_gK3);
GC::scratch_space = GC::gc_handle ? &(GC::gc_handle->scratch) : nullptr;
for (uint32_t i_d1 = 0; i_d1 < sz_gK3; ++i_d1) {
auto &key_e1 = (*gK3.keys)[i_d1];
auto &val_e2 = vecs_c1[i_d1];
col_a1[i_d1] = (sum(c_3z[val_e2]));

col_a2[i_d1] = (get<1>(key_e1));

col_a3[i_d1] = (get<2>(key_e1));

col_a4[i_d1] = (val_e2.size);

col_a5[i_d1] = (avg(c_3z[val_e2]));

GC::scratch_space->release();
}
GC::scratch_space = nullptr;
print(*out_8r);
puts("done.");
return 0;
}
