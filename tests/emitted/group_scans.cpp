// What the reference's own queries put INSIDE the generated group loop (engine/ast.py:691-794), at sizes where one launch per
// group is hopeless: per-group windows into one flat buffer, reductions of per-group scans, two-column aggregates and element-wise
// expressions over `col[val]`.  Each function is the emitted shape of one reference query; the only addition is the raw dump of
// the output table at the end (tests/emitted/dump_cols.h) where the generated code would print.
//   dll_q7      benchmark/quries/Aquery/q7.a   SELECT stocksymbol, avgs(5, price) FROM trade ASSUMING ASC time GROUP BY stocksymbol
//   dll_memopt  mem_opt.cpp:28-65              the older AQHashTable shape: avgw(10, sales[vecs[i]], col[i]) into buf + offsets
//   dll_q4      tests/q4.a:23                  SELECT ID, max(ratios(endofdayprice)), min(ratios(endofdayprice)) FROM ticks GROUP BY ID
//   dll_q9      benchmark/h2o/groupby.sql:20   SELECT id2, id4, pow(corr(v1, v2), 2) AS r2 FROM source GROUP BY id2, id4
//   dll_q8      benchmark/h2o/groupby.sql:17   SELECT id6, subvec(v3,0,2) AS v3 FROM source GROUP BY id6   (engine/expr.py:237: `v3[val].subvec(0, 2)`)
//   dll_expr    tests/stock.a:24 per symbol    SELECT sym, max(price - mins(price)), sum(price + price), mins(2, price) ... GROUP BY sym
#include "header.cxx"
#include "./server/monetdb_conn.h"
#include "./server/aggregations.h"
#include "./server/hasher.h"
#include "dump_cols.h"

__AQEXPORT__(int) dll_q7(Context* cxt) {
	using namespace std;
	using namespace types;
	auto server = static_cast<DataSource*>(cxt->curr_server);
auto len_1a = server->cnt;
auto stocksymbol_2b = ColRef<int>(len_1a, server->getCol(0, types::Type_t::AINT32));
auto price_3c = ColRef<int>(len_1a, server->getCol(1, types::Type_t::AINT32));
const char* names_4d[] = {"stocksymbol", "avgw5yprice"};
auto out_5e = new TableInfo<int,vector_type<value_type<decays<decltype(avgw(5, price_3c))>>>>("out_5e", names_4d);
decltype(auto) col_6f = out_5e->get_col<0>();
decltype(auto) col_7g = out_5e->get_col<1>();
uint32_t len_8h = stocksymbol_2b.size;
typedef record<decays<decltype(stocksymbol_2b)>::value_t> record_type9i;
auto g10j = HashTableFactory<record_type9i, transTypes<record_type9i, hasher>>::get<decays<decltype(stocksymbol_2b)>>(stocksymbol_2b);
auto sz_g10j = g10j.size;
auto vecs_11k = g10j.values;
col_6f.resize(sz_g10j);
col_7g.resize(sz_g10j);
auto buf_col_7g = static_cast<double *>(calloc(len_8h, sizeof(double)));
for (uint32_t i12 = 0; i12 < sz_g10j; ++i12) {
col_7g[i12].init_from(vecs_11k[i12].size, buf_col_7g + g10j.offsets[i12]);
}
GC::scratch_space = GC::gc_handle ? &(GC::gc_handle->scratch) : nullptr;
for (uint32_t i13 = 0; i13 < sz_g10j; ++i13) {
auto &key_14l = (*g10j.keys)[i13];
auto &val_15m = vecs_11k[i13];
col_6f[i13] = (get<0>(key_14l));

avgw(5, price_3c[val_15m], col_7g[i13]);

GC::scratch_space->release();
}
GC::scratch_space = nullptr;
aqtest::dump_table("q7.out", *out_5e);
puts("done.");
return 0;
}

__AQEXPORT__(int) dll_memopt(Context* cxt) {
	using namespace std;
	using namespace types;
	auto server = static_cast<DataSource*>(cxt->curr_server);
auto len_1 = server->cnt;
auto mont_2 = ColRef<int>(len_1, server->getCol(0, types::Type_t::AINT32));
auto sales_3 = ColRef<int>(len_1, server->getCol(1, types::Type_t::AINT32));
const char* names_4[] = {"mont", "avgw10ysales"};
auto out_5 = new TableInfo<int,vector_type<double>>("out_5", names_4);
decltype(auto) col_6 = out_5->get_col<0>();
decltype(auto) col_7 = out_5->get_col<1>();
typedef record<decays<decltype(mont_2)>::value_t> record_type8;
AQHashTable<record_type8, transTypes<record_type8, hasher>> g9 {(uint32_t)len_1};
g9.hashtable_push_all<decays<decltype(mont_2)>>(mont_2, len_1);
auto vecs_10 = g9.ht_postproc(len_1);
auto arr_values = g9.values().data();
auto arr_len = g9.size();
col_6.reserve(arr_len);
col_7.resize(arr_len);
auto buf_col_7 = new double[len_1];
for (uint32_t i = 0; i < arr_len; ++i) {
	col_7[i].init_from(vecs_10[i].size, buf_col_7 + g9.ht_base[i]);
}
for (uint32_t i = 0; i < arr_len; ++i) {
auto &key_11 = arr_values[i];
col_6.emplace_back(get<0>(key_11));

avgw(10, sales_3[vecs_10[i]], col_7[i]);

}
aqtest::dump_table("memopt.out", *out_5);
puts("done.");
return 0;
}

__AQEXPORT__(int) dll_q4(Context* cxt) {
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
aqtest::dump_table("q4.out", *out_5e);
puts("done.");
return 0;
}

__AQEXPORT__(int) dll_q9(Context* cxt) {
	using namespace std;
	using namespace types;
	auto server = static_cast<DataSource*>(cxt->curr_server);
auto len_1a = server->cnt;
auto id2_2b = ColRef<int>(len_1a, server->getCol(0, types::Type_t::AINT32));
auto id4_3c = ColRef<int>(len_1a, server->getCol(1, types::Type_t::AINT32));
auto v1_4d = ColRef<int>(len_1a, server->getCol(2, types::Type_t::AINT32));
auto v2_5e = ColRef<int>(len_1a, server->getCol(3, types::Type_t::AINT32));
const char* names_6f[] = {"id2", "id4", "r2"};
auto out_7g = new TableInfo<int,int,double>("out_7g", names_6f);
decltype(auto) col_8h = out_7g->get_col<0>();
decltype(auto) col_9i = out_7g->get_col<1>();
decltype(auto) col_10j = out_7g->get_col<2>();
uint32_t len_11k = id2_2b.size;
typedef record<decays<decltype(id2_2b)>::value_t,decays<decltype(id4_3c)>::value_t> record_type12l;
auto g13m = HashTableFactory<record_type12l, transTypes<record_type12l, hasher>>::get<decays<decltype(id2_2b)>, decays<decltype(id4_3c)>>(id2_2b, id4_3c);
auto sz_g13m = g13m.size;
auto vecs_14n = g13m.values;
col_8h.resize(sz_g13m);
col_9i.resize(sz_g13m);
col_10j.resize(sz_g13m);
GC::scratch_space = GC::gc_handle ? &(GC::gc_handle->scratch) : nullptr;
for (uint32_t i15 = 0; i15 < sz_g13m; ++i15) {
auto &key_16o = (*g13m.keys)[i15];
auto &val_17p = vecs_14n[i15];
col_8h[i15] = (get<0>(key_16o));

col_9i[i15] = (get<1>(key_16o));

col_10j[i15] = (pow(corr(v1_4d[val_17p], v2_5e[val_17p]), 2));

GC::scratch_space->release();
}
GC::scratch_space = nullptr;
aqtest::dump_table("q9.out", *out_7g);
puts("done.");
return 0;
}

__AQEXPORT__(int) dll_expr(Context* cxt) {
	using namespace std;
	using namespace types;
	auto server = static_cast<DataSource*>(cxt->curr_server);
auto len_1a = server->cnt;
auto sym_2b = ColRef<int>(len_1a, server->getCol(0, types::Type_t::AINT32));
auto price_3c = ColRef<int>(len_1a, server->getCol(1, types::Type_t::AINT32));
const char* names_4d[] = {"sym", "maxdrawup", "sum2", "above", "minw2yprice"};
auto out_5e = new TableInfo<int,value_type<decays<decltype(max((price_3c - mins(price_3c))))>>,value_type<decays<decltype(sum((price_3c + price_3c)))>>,
                            value_type<decays<decltype(sum((price_3c - 100)))>>,vector_type<value_type<decays<decltype(minw(2, price_3c))>>>>("out_5e", names_4d);
decltype(auto) col_6f = out_5e->get_col<0>();
decltype(auto) col_7g = out_5e->get_col<1>();
decltype(auto) col_8h = out_5e->get_col<2>();
decltype(auto) col_9i = out_5e->get_col<3>();
decltype(auto) col_10j = out_5e->get_col<4>();
uint32_t len_11k = sym_2b.size;
typedef record<decays<decltype(sym_2b)>::value_t> record_type12l;
auto g13m = HashTableFactory<record_type12l, transTypes<record_type12l, hasher>>::get<decays<decltype(sym_2b)>>(sym_2b);
auto sz_g13m = g13m.size;
auto vecs_14n = g13m.values;
col_6f.resize(sz_g13m);
col_7g.resize(sz_g13m);
col_8h.resize(sz_g13m);
col_9i.resize(sz_g13m);
col_10j.resize(sz_g13m);
auto buf_col_10j = static_cast<int *>(calloc(len_11k, sizeof(int)));
for (uint32_t i15 = 0; i15 < sz_g13m; ++i15) {
col_10j[i15].init_from(vecs_14n[i15].size, buf_col_10j + g13m.offsets[i15]);
}
GC::scratch_space = GC::gc_handle ? &(GC::gc_handle->scratch) : nullptr;
for (uint32_t i16 = 0; i16 < sz_g13m; ++i16) {
auto &key_17o = (*g13m.keys)[i16];
auto &val_18p = vecs_14n[i16];
col_6f[i16] = (get<0>(key_17o));

col_7g[i16] = (max((price_3c[val_18p] - mins(price_3c[val_18p]))));

col_8h[i16] = (sum((price_3c[val_18p] + price_3c[val_18p])));

col_9i[i16] = (sum((price_3c[val_18p] - 100)));

minw(2, price_3c[val_18p], col_10j[i16]);

GC::scratch_space->release();
}
GC::scratch_space = nullptr;
aqtest::dump_table("expr.out", *out_5e);
puts("done.");
return 0;
}

__AQEXPORT__(int) dll_q8(Context* cxt) {
	using namespace std;
	using namespace types;
	auto server = static_cast<DataSource*>(cxt->curr_server);
auto len_1a = server->cnt;
auto id6_2b = ColRef<int>(len_1a, server->getCol(0, types::Type_t::AINT32));
auto v3_3c = ColRef<int>(len_1a, server->getCol(1, types::Type_t::AINT32));
const char* names_4d[] = {"id6", "v3"};
auto out_5e = new TableInfo<int,vector_type<int>>("out_5e", names_4d);
decltype(auto) col_6f = out_5e->get_col<0>();
decltype(auto) col_7g = out_5e->get_col<1>();
uint32_t len_8h = id6_2b.size;
typedef record<decays<decltype(id6_2b)>::value_t> record_type9i;
auto g10j = HashTableFactory<record_type9i, transTypes<record_type9i, hasher>>::get<decays<decltype(id6_2b)>>(id6_2b);
auto sz_g10j = g10j.size;
auto vecs_11k = g10j.values;
col_6f.resize(sz_g10j);
col_7g.resize(sz_g10j);
GC::scratch_space = GC::gc_handle ? &(GC::gc_handle->scratch) : nullptr;
for (uint32_t i13 = 0; i13 < sz_g10j; ++i13) {
auto &key_14l = (*g10j.keys)[i13];
auto &val_15m = vecs_11k[i13];
col_6f[i13] = (get<0>(key_14l));

col_7g[i13] = (v3_3c[val_15m].subvec(0, 2));

GC::scratch_space->release();
}
GC::scratch_space = nullptr;
aqtest::dump_table("q8.out", *out_5e);
puts("done.");
return 0;
}
