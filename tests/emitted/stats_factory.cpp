// ColRef::populate_stats (reference server/table.h:76-91) and the perfect-hash front door of HashTableFactory::get
// (server/hasher.h:327-357): with the key columns' TableStats populated, `get` sizes its plan from the 2^bits domain instead of a
// sample; the result (first-occurrence group order, descending row lists) must not change.
#include "header.cxx"
#include "./server/monetdb_conn.h"
#include "./server/aggregations.h"
#include "./server/hasher.h"

__AQEXPORT__(int) dll_stats(Context* cxt) {
	using namespace std;
	using namespace types;
	auto server = static_cast<DataSource*>(cxt->curr_server);
auto len_1 = server->cnt;
auto a_1 = ColRef<int>(len_1, server->getCol(0, types::Type_t::AINT32));
auto b_2 = ColRef<int>(len_1, server->getCol(1, types::Type_t::AINT32));
auto c_3 = ColRef<int>(len_1, server->getCol(2, types::Type_t::AINT32));
printf("before %d %d\n", (int)a_1.stats.bits, (int)b_2.stats.bits);
typedef record<decays<decltype(a_1)>::value_t,decays<decltype(b_2)>::value_t> record_typeA;
auto g0 = HashTableFactory<record_typeA, transTypes<record_typeA, hasher>>::get<decays<decltype(a_1)>, decays<decltype(b_2)>>(a_1, b_2);
bool ok = a_1.populate_stats() && b_2.populate_stats();
printf("stats %d a %d %d b %d %d\n", (int)ok, (int)a_1.stats.minima, (int)a_1.stats.bits, (int)b_2.stats.minima, (int)b_2.stats.bits);
auto g1 = HashTableFactory<record_typeA, transTypes<record_typeA, hasher>>::get<decays<decltype(a_1)>, decays<decltype(b_2)>>(a_1, b_2);
auto g2 = HashTableFactory<record_typeA, transTypes<record_typeA, hasher>, 3>::get<decays<decltype(a_1)>, decays<decltype(b_2)>>(a_1, b_2);   // threshold below the key width
printf("groups %u %u %u\n", g0.size, g1.size, g2.size);
int same = g0.size == g1.size && g1.size == g2.size;
for (uint32_t i = 0; same && i < g0.size; ++i) {
    same = (*g0.keys)[i] == (*g1.keys)[i] && (*g1.keys)[i] == (*g2.keys)[i] && g0.values[i].size == g1.values[i].size && g0.offsets[i] == g1.offsets[i] && g1.offsets[i] == g2.offsets[i];
    for (uint32_t j = 0; same && j < g0.values[i].size; ++j) same = g0.values[i][j] == g1.values[i][j] && g1.values[i][j] == g2.values[i][j];
}
printf("same %d\n", same);
for (uint32_t i = 0; i < g1.size; ++i) printf("%d,%d,%lld\n", get<0>((*g1.keys)[i]), get<1>((*g1.keys)[i]), (long long)sum(c_3[g1.values[i]]));
puts("done.");
	return 0;
}
