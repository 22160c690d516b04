// Emitted shape of tests/moving_avg.a:
//   SELECT Mont, avgs(3, sales) FROM sale ASSUMING ASC Mont                 (engine/ast.py:340-447, 2-arg avgs -> avgw)
//   SELECT Mont, mins(2, sales) FROM sale ASSUMING DESC Mont GROUP BY Mont   (engine/ast.py:656-790 group loop)
// Column binding, output table, initfrom and print lines are written exactly as the code generator writes them;
// only the uuid suffixes are made up.
#include "header.cxx"
#include "./server/monetdb_conn.h"
#include "./server/aggregations.h"
#include "./server/hasher.h"

__AQEXPORT__(int) dll_2Cxoox(Context* cxt) {
	using namespace std;
	using namespace types;
	auto server = static_cast<DataSource*>(cxt->curr_server);
	auto timer = chrono::high_resolution_clock::now();
auto len_4ycjiV = server->cnt;
auto mont_8AE = ColRef<int>(len_4ycjiV, server->getCol(0, types::Type_t::AINT32));
auto sales_2RB = ColRef<int>(len_4ycjiV, server->getCol(1, types::Type_t::AINT32));
const char* names_6pIt[] = {"Mont", "avgw3ysales"};
auto out_2LuaMH = new TableInfo<int,value_type<decays<decltype(avgw(3, sales_2RB))>>>("out_2LuaMH", names_6pIt);
out_2LuaMH->get_col<0>().initfrom(mont_8AE, "Mont");
out_2LuaMH->get_col<1>().initfrom(avgw(3, sales_2RB), "avgw3ysales");
print(*out_2LuaMH);
FILE* fp_5LQeym = fopen("moving_avg_output.csv", "wb");
out_2LuaMH->printall(";", "\n", nullptr, fp_5LQeym);
fclose(fp_5LQeym);
puts("done.");
return 0;
}

__AQEXPORT__(int) dll_6Ywxmn(Context* cxt) {
	using namespace std;
	using namespace types;
	auto server = static_cast<DataSource*>(cxt->curr_server);
	auto timer = chrono::high_resolution_clock::now();
auto len_1a = server->cnt;
auto mont_3c = ColRef<int>(len_1a, server->getCol(0, types::Type_t::AINT32));
auto sales_4d = ColRef<int>(len_1a, server->getCol(1, types::Type_t::AINT32));
const char* names_5e[] = {"Mont", "minw2ysales"};
auto out_6f = new TableInfo<int,vector_type<value_type<decays<decltype(minw(2, sales_4d))>>>>("out_6f", names_5e);
decltype(auto) col_7g = out_6f->get_col<0>();
decltype(auto) col_8h = out_6f->get_col<1>();
uint32_t len_9i = mont_3c.size;
typedef record<decays<decltype(mont_3c)>::value_t> record_typegj3e8Xf;
auto gMzMTEvd = HashTableFactory<record_typegj3e8Xf, transTypes<record_typegj3e8Xf, hasher>>::get<decays<decltype(mont_3c)>>(mont_3c);
auto sz_gMzMTEvd = gMzMTEvd.size;
auto vecs_x1 = gMzMTEvd.values;
col_7g.resize(sz_gMzMTEvd);
col_8h.resize(sz_gMzMTEvd);
auto buf_col_8h = static_cast<int *>(calloc(len_9i, sizeof(int)));
for (uint32_t i1 = 0; i1 < sz_gMzMTEvd; ++i1) {
col_8h[i1].init_from(vecs_x1[i1].size, buf_col_8h + gMzMTEvd.offsets[i1]);
}
GC::scratch_space = GC::gc_handle ? &(GC::gc_handle->scratch) : nullptr;
for (uint32_t i2 = 0; i2 < sz_gMzMTEvd; ++i2) {
auto &key_3iNX3qG = (*gMzMTEvd.keys)[i2];
auto &val_7jjv8Mo = vecs_x1[i2];
col_7g[i2] = (get<0>(key_3iNX3qG));

minw(2, sales_4d[val_7jjv8Mo], col_8h[i2]);

GC::scratch_space->release();
}
GC::scratch_space = nullptr;
FILE* fp_zz = fopen("flatten.csv", "wb");
out_6f->printall(",", "\n", nullptr, fp_zz);
fclose(fp_zz);
puts("done.");
return 0;
}
