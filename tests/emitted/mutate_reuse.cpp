// A device-produced vector that the host WRITES and the device then reads again (device.h): the registry must
// not serve a stale device mirror after mutable host access was handed out.  Shapes: operator[] returning _Ty&, operator=(scalar),
// emplace_back within capacity, erase.  Column binding as emitted by engine/ast.py:367-370.
#include "header.cxx"
#include "./server/monetdb_conn.h"
#include "./server/aggregations.h"
#include "./server/hasher.h"

__AQEXPORT__(int) dll_mutate(Context* cxt) {
	using namespace std;
	using namespace types;
	auto server = static_cast<DataSource*>(cxt->curr_server);
auto len_1 = server->cnt;
auto a_1 = ColRef<int>(len_1, server->getCol(0, types::Type_t::AINT32));
auto b_2 = ColRef<int>(len_1, server->getCol(1, types::Type_t::AINT32));
auto x = a_1 + b_2;                 // produced on the device, host copy not yet valid
x[0] = 500;                         // host write through operator[] (downloads first)
x[2] = -7;
auto y = x + x;                     // the device must see 500 and -7
printf("%lld %lld %lld %lld\n", (long long)y[0], (long long)y[1], (long long)y[2], (long long)sum(x));
auto z = a_1 - b_2;                 // device-produced, never touched by the host ...
auto w = z + z;                     // ... chained on the device
w.erase(w.begin());                 // host mutation of a device result
auto s2 = sum(w);
printf("%lld %u\n", (long long)s2, (unsigned)w.size);
auto m = maxs(a_1);                 // device scan result
m[len_1 - 1] = 1000;                // host write of the last element
printf("%lld\n", (long long)max(m));
puts("done.");
	return 0;
}
