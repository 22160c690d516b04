// Emitted shape of tests/stock.a q1, q2, q4 (scalar results of vector expressions) and q3's filter mask.
#include "header.cxx"
#include "./server/monetdb_conn.h"
#include "./server/aggregations.h"

__AQEXPORT__(int) dll_q1(Context* cxt) {
	using namespace std;
	using namespace types;
	auto server = static_cast<DataSource*>(cxt->curr_server);
	auto timer = chrono::high_resolution_clock::now();
auto len_a1 = server->cnt;
auto timestamp_b2 = ColRef<int>(len_a1, server->getCol(0, types::Type_t::AINT32));
auto price_c3 = ColRef<int>(len_a1, server->getCol(1, types::Type_t::AINT32));
const char* names_d4[] = {"max_price_min_timestamp"};
auto out_e5 = new TableInfo<value_type<decays<decltype(max((price_c3 - min(timestamp_b2))))>>>("out_e5", names_d4);
out_e5->get_col<0>().initfrom(max((price_c3 - min(timestamp_b2))), "max_price_min_timestamp");
print(*out_e5);
puts("done.");
return 0;
}

__AQEXPORT__(int) dll_q2(Context* cxt) {
	using namespace std;
	using namespace types;
	auto server = static_cast<DataSource*>(cxt->curr_server);
	auto timer = chrono::high_resolution_clock::now();
auto len_a1 = server->cnt;
auto price_c3 = ColRef<int>(len_a1, server->getCol(1, types::Type_t::AINT32));
const char* names_d4[] = {"max_price_mins_price"};
auto out_e5 = new TableInfo<value_type<decays<decltype(max((price_c3 - mins(price_c3))))>>>("out_e5", names_d4);
out_e5->get_col<0>().initfrom(max((price_c3 - mins(price_c3))), "max_price_mins_price");
print(*out_e5);
puts("done.");
return 0;
}

// q3: the WHERE clause runs in the SQL engine in the hybrid design; the mask expression is what a C++-side
// filter would evaluate (price - timestamp > 1) -- printed as 0/1
__AQEXPORT__(int) dll_q3mask(Context* cxt) {
	using namespace std;
	using namespace types;
	auto server = static_cast<DataSource*>(cxt->curr_server);
	auto timer = chrono::high_resolution_clock::now();
auto len_a1 = server->cnt;
auto timestamp_b2 = ColRef<int>(len_a1, server->getCol(0, types::Type_t::AINT32));
auto price_c3 = ColRef<int>(len_a1, server->getCol(1, types::Type_t::AINT32));
auto mask_f6 = ((price_c3 - timestamp_b2) > 1);
for (uint32_t i = 0; i < mask_f6.size; ++i) putchar(mask_f6[i] ? '1' : '0');
putchar('\n');
auto kept_g7 = price_c3[mask_f6];
print(kept_g7);
puts("done.");
return 0;
}
