// Group-by over string keys in the emitted shape (the frozen sample of generated code, reference mem_opt.cpp:17-65, groups by
// `ColRef<const char*>`).  The reference's tuple == compares `const char*` keys as POINTERS and astring_view keys by content
// (server/types.h:281-334, server/hasher.h:97-106); both go through HashTableFactory::get here.
#include "header.cxx"
#include "./server/hasher.h"
#include "./server/monetdb_conn.h"
#include "./server/aggregations.h"

__AQEXPORT__(int) dll_strkeys(Context* cxt) {
	using namespace std;
	using namespace types;
	auto server = static_cast<DataSource*>(cxt->curr_server);
auto len_4ycjiV = server->cnt;
auto mont_8AE = ColRef<const char*>(len_4ycjiV, server->getCol(0, types::Type_t::ASTR));
auto sales_2RB = ColRef<int>(len_4ycjiV, server->getCol(2, types::Type_t::AINT32));
const char* names_6pIt[] = {"mont", "sumsales", "cnt"};
auto out_2LuaMH = new TableInfo<const char*,value_type<decays<decltype(sum(sales_2RB))>>,int>("out_2LuaMH", names_6pIt);
decltype(auto) col_EeW23s = out_2LuaMH->get_col<0>();
decltype(auto) col_5gY1Dm = out_2LuaMH->get_col<1>();
decltype(auto) col_7hZ = out_2LuaMH->get_col<2>();
typedef record<decays<decltype(mont_8AE)>::value_t> record_typegj3e8Xf;
auto gMz = HashTableFactory<record_typegj3e8Xf, transTypes<record_typegj3e8Xf, hasher>>::get<decays<decltype(mont_8AE)>>(mont_8AE);
auto sz_gMz = gMz.size;
auto vecs_gMz = gMz.values;
col_EeW23s.resize(sz_gMz);
col_5gY1Dm.resize(sz_gMz);
col_7hZ.resize(sz_gMz);
GC::scratch_space = GC::gc_handle ? &(GC::gc_handle->scratch) : nullptr;
for (uint32_t i = 0; i < sz_gMz; ++i) {
auto &key_3iNX3qG = (*gMz.keys)[i];
auto &val_7jjv8Mo = vecs_gMz[i];
col_EeW23s[i] = (get<0>(key_3iNX3qG));

col_5gY1Dm[i] = (sum(sales_2RB[val_7jjv8Mo]));

col_7hZ[i] = (val_7jjv8Mo.size);

GC::scratch_space->release();
}
GC::scratch_space = nullptr;
for (uint32_t i = 0; i < sz_gMz; ++i) printf("%s,%lld,%d\n", col_EeW23s[i], (long long)col_5gY1Dm[i], col_7hZ[i]);
puts("--");
// the same strings behind distinct pointers: pointer keys see 12 groups, astring_view keys see the 4 contents
auto own_ptr = ColRef<const char*>(len_4ycjiV, server->getCol(1, types::Type_t::ASTR));
typedef record<decays<decltype(own_ptr)>::value_t> record_typeP;
auto gP = HashTableFactory<record_typeP, transTypes<record_typeP, hasher>>::get<decays<decltype(own_ptr)>>(own_ptr);
auto own_sv = ColRef<astring_view>(len_4ycjiV, server->getCol(1, types::Type_t::ASV));
typedef record<decays<decltype(own_sv)>::value_t> record_typeS;
auto gS = HashTableFactory<record_typeS, transTypes<record_typeS, hasher>>::get<decays<decltype(own_sv)>>(own_sv);
printf("pointer groups %u, string-view groups %u\n", gP.size, gS.size);
for (uint32_t i = 0; i < gS.size; ++i) {
    auto& val = gS.values[i];
    printf("%s,%lld,%u,%u\n", (const char*)get<0>((*gS.keys)[i]), (long long)sum(sales_2RB[val]), val.size, val[0]);
}
puts("done.");
return 0;
}
