// Emitted shape of `SELECT DISTINCT a, c FROM t` (engine/ast.py:481-489 appends `out->distinct();`) plus the host utilities
// of TableInfo the generator's ORDER BY / view code targets (server/table.h:429-461 order_by / materialize, :262-343 ColView).
#include "header.cxx"
#include "./server/monetdb_conn.h"
#include "./server/aggregations.h"
#include "./server/hasher.h"

__AQEXPORT__(int) dll_distinct(Context* cxt) {
	using namespace std;
	using namespace types;
	auto server = static_cast<DataSource*>(cxt->curr_server);
auto len_1 = server->cnt;
auto a_2 = ColRef<int>(len_1, server->getCol(0, types::Type_t::AINT32));
auto c_3 = ColRef<int>(len_1, server->getCol(2, types::Type_t::AINT32));
const char* names_4[] = {"a", "c"};
auto out_5 = new TableInfo<int,int>("out_5", names_4);
out_5->get_col<0>().initfrom(a_2, "a");
out_5->get_col<1>().initfrom(c_3, "c");
out_5->distinct();
printf("distinct rows %u\n", out_5->get_col<0>().size);
// ORDER BY a DESC, c ASC over the distinct rows; first rows through a view, then materialised
auto ord_6 = out_5->order_by<-1, 1>();
auto view_7 = ColView<int>(out_5->get_col<0>(), *ord_6);
printf("top a %d %d %d\n", view_7[0], view_7[1], view_7[2]);
auto sorted_8 = out_5->materialize_copy(*ord_6);
sorted_8->printall(",", "\n", nullptr, nullptr, 5);
long long chk = 0;
for (uint32_t i = 0; i < sorted_8->get_col<0>().size; ++i) chk = chk * 31 + sorted_8->get_col<0>()[i] * 131 + sorted_8->get_col<1>()[i];
printf("checksum %lld\n", chk);
puts("done.");
return 0;
}
