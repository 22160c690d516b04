// Emitted shape of tests/funcs.a: three user functions (engine/ast.py:1610-1720 emits `auto name = [](const auto& ...) {...};`)
// composed inside the special group-by loop (engine/ast.py:620-794):
//   FUNCTION covariance(x, y) { xmean := avg(x); ymean := avg(y); avg((x - xmean) * (y - ymean)) }
//   FUNCTION sd(x) { sqrt(covariance(x, x)) }
//   FUNCTION pairCorr(x, y) { covariance(x, y) / (sd(x) * sd(y)) }
//   SELECT pairCorr(c, b), a, sum(b) FROM test1 GROUP BY a     (tests/funcs.a multiplies by the column d, which makes the
//   output vector-valued; recorded here in its scalar form.  ORDER BY runs in the SQL engine)
#include "header.cxx"
#include "./server/monetdb_conn.h"
#include "./server/aggregations.h"
#include "./server/hasher.h"

auto covariance = [](const auto& x, const auto& y) {
	auto xmean = avg(x);
	auto ymean = avg(y);
	return avg(((x - xmean) * (y - ymean)));
};
auto sd = [](const auto& x) {
	return sqrt(covariance(x, x));
};
auto paircorr = [](const auto& x, const auto& y) {
	return (covariance(x, y) / (sd(x) * sd(y)));
};

__AQEXPORT__(int) dll_funcs(Context* cxt) {
	using namespace std;
	using namespace types;
	auto server = static_cast<DataSource*>(cxt->curr_server);
auto len_1 = server->cnt;
auto a_1 = ColRef<int>(len_1, server->getCol(0, types::Type_t::AINT32));
auto b_2 = ColRef<int>(len_1, server->getCol(1, types::Type_t::AINT32));
auto c_3 = ColRef<int>(len_1, server->getCol(2, types::Type_t::AINT32));
auto d_4 = ColRef<int>(len_1, server->getCol(3, types::Type_t::AINT32));
const char* names_5[] = {"paircorr", "a", "sumb"};
auto out_6 = new TableInfo<double,int,value_type<decays<decltype(sum(b_2))>>>("out_6", names_5);
decltype(auto) col_7 = out_6->get_col<0>();
decltype(auto) col_8 = out_6->get_col<1>();
decltype(auto) col_9 = out_6->get_col<2>();
typedef record<decays<decltype(a_1)>::value_t> record_typeA;
auto gA = HashTableFactory<record_typeA, transTypes<record_typeA, hasher>>::get<decays<decltype(a_1)>>(a_1);
auto sz_gA = gA.size;
auto vecs_gA = gA.values;
col_7.resize(sz_gA);
col_8.resize(sz_gA);
col_9.resize(sz_gA);
GC::scratch_space = GC::gc_handle ? &(GC::gc_handle->scratch) : nullptr;
for (uint32_t i = 0; i < sz_gA; ++i) {
auto &key_k = (*gA.keys)[i];
auto &val_v = vecs_gA[i];
col_7[i] = (paircorr(c_3[val_v], b_2[val_v]));

col_8[i] = (get<0>(key_k));

col_9[i] = (sum(b_2[val_v]));

GC::scratch_space->release();
}
GC::scratch_space = nullptr;
out_6->printall(",", "\n");
puts("done.");
return 0;
}
