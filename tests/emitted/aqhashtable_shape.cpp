// The older, still-executable group-by shape of the reference (the one mem_opt.cpp:28-65 spells by hand):
// AQHashTable + per-row hashtable_push + ht_postproc, then a vector-valued aggregate (avgw) per group written into
// one flat buffer sliced by group offsets.
#include "header.cxx"
#include "./server/monetdb_conn.h"
#include "./server/aggregations.h"
#include "./server/hasher.h"

__AQEXPORT__(int) dll_7Hs2mk(Context* cxt) {
	using namespace std;
	using namespace types;
	auto server = static_cast<DataSource*>(cxt->curr_server);
	auto timer = chrono::high_resolution_clock::now();
auto len_1 = server->cnt;
auto a_2 = ColRef<int>(len_1, server->getCol(0, types::Type_t::AINT32));
auto b_3 = ColRef<int>(len_1, server->getCol(1, types::Type_t::AINT32));
auto c_4 = ColRef<int>(len_1, server->getCol(2, types::Type_t::AINT32));
const char* names_5[] = {"a", "b", "avgw2yc"};
auto out_6 = new TableInfo<int,int,vector_type<double>>("out_6", names_5);
decltype(auto) col_7 = out_6->get_col<0>();
decltype(auto) col_8 = out_6->get_col<1>();
decltype(auto) col_9 = out_6->get_col<2>();
typedef record<decays<decltype(a_2)>::value_t,decays<decltype(b_3)>::value_t> record_type10;
AQHashTable<record_type10, transTypes<record_type10, hasher>> g11 {(uint32_t)len_1};
for (uint32_t i12 = 0; i12 < len_1; ++i12){
	g11.hashtable_push(forward_as_tuple(a_2[i12], b_3[i12]), i12);
}
auto vecs_13 = g11.ht_postproc(len_1);
auto arr_values = g11.values().data();
auto arr_len = g11.size();
col_7.reserve(arr_len);
col_8.reserve(arr_len);
col_9.resize(arr_len);
auto buf_col_9 = new double[len_1];
for (uint32_t i = 0; i < arr_len; ++i) {
	col_9[i].init_from(vecs_13[i].size, buf_col_9 + g11.ht_base[i]);
}
for (uint32_t i = 0; i < arr_len; ++i) {
auto &key_14 = arr_values[i];
col_7.emplace_back(get<0>(key_14));
col_8.emplace_back(get<1>(key_14));

avgw(2, c_4[vecs_13[i]], col_9[i]);

}
out_6->printall(",", "\n", nullptr, nullptr, 10);
puts("done.");
return 0;
}
