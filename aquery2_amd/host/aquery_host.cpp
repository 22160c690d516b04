// aquery_host -- the host runtime shim of the MI355X AQuery library: what the reference's server/server.cpp does
// between the compiler and a generated module, for the slice this library serves.
//
//   aquery_host <dll.so> <messages> [--root DIR]
//
// `messages` is a recorded message list, one message per line, in the reference's protocol (server/server.cpp:284-350; the Python
// front end sends the same strings through receive_args, engine/storage.py:249-278):
//     Q<sql>        a statement for the SQL data source (the reference hands it to embedded MonetDB: server.cpp:285-295)
//     P<function>   dlsym(dll, function)(cxt): a generated post-processor function (server.cpp:296-307) -- the hot path
//     O<limit>      print the data source's current result set, at most <limit> rows (server.cpp:332-350)
//     #...          comment
// Session shape as in threaded_main (server.cpp:236-616): dlopen -> __AQ_Init_GC__(cxt) (server.cpp:152-161, header.cxx) ->
// cxt->init_session() -> messages -> __AQ_End_Session__(cxt) (this library's hook: the module releases its device state) ->
// cxt->end_session() -> dlclose; monet_time / postproc_time are accumulated around Q and P like cfg->stats (server.cpp:290-305).
//
// The SQL side is NOT rebuilt (MonetDB is out of scope and absent): MiniSqlSource below understands exactly the
// statements the recorded lists of the reference's own test scripts need -- CREATE TABLE, COPY ... INTO / LOAD DATA INFILE from a
// CSV file, SELECT <columns> FROM <table> [ORDER BY <column> [ASC|DESC]] -- and hands result columns out as borrowed host arrays
// through DataSource::getCol, which is all a generated function sees of the data source (engine/ast.py:367-370).
#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

#include "aquery/libaquery.h"

namespace {

std::string upper(std::string s) { for (auto& c : s) c = (char)std::toupper((unsigned char)c); return s; }
std::string trim(const std::string& s) {
    size_t a = s.find_first_not_of(" \t\r\n"), b = s.find_last_not_of(" \t\r\n;");
    return a == std::string::npos ? std::string() : s.substr(a, b - a + 1);
}
std::vector<std::string> split(const std::string& s, char sep) {
    std::vector<std::string> out; std::string cur;
    for (char c : s) { if (c == sep) { out.push_back(trim(cur)); cur.clear(); } else cur += c; }
    out.push_back(trim(cur));
    return out;
}

struct Column {
    std::string name;
    int type = types::AINT32;                 // AINT32, AINT64, ADOUBLE, AFLOAT, ASTR
    std::vector<int> i32; std::vector<long long> i64; std::vector<double> f64; std::vector<float> f32;
    std::vector<std::string> text; std::vector<const char*> ptr;
    size_t size() const { return type == types::AINT32 ? i32.size() : type == types::AINT64 ? i64.size() : type == types::ADOUBLE ? f64.size() : type == types::AFLOAT ? f32.size() : text.size(); }
    void push(const std::string& v) {
        switch (type) {
        case types::AINT32: i32.push_back((int)std::strtol(v.c_str(), nullptr, 10)); break;
        case types::AINT64: i64.push_back(std::strtoll(v.c_str(), nullptr, 10)); break;
        case types::ADOUBLE: f64.push_back(std::strtod(v.c_str(), nullptr)); break;
        case types::AFLOAT: f32.push_back(std::strtof(v.c_str(), nullptr)); break;
        default: text.push_back(v); break;
        }
    }
    double key(size_t i) const { return type == types::AINT32 ? i32[i] : type == types::AINT64 ? (double)i64[i] : type == types::ADOUBLE ? f64[i] : type == types::AFLOAT ? f32[i] : 0.0; }
    void* data() {
        switch (type) {
        case types::AINT32: return i32.data();
        case types::AINT64: return i64.data();
        case types::ADOUBLE: return f64.data();
        case types::AFLOAT: return f32.data();
        default: ptr.clear(); for (auto& s : text) ptr.push_back(s.c_str()); return ptr.data();
        }
    }
    Column take(const std::vector<size_t>& rows) const {
        Column c; c.name = name; c.type = type;
        for (size_t r : rows) {
            switch (type) {
            case types::AINT32: c.i32.push_back(i32[r]); break;
            case types::AINT64: c.i64.push_back(i64[r]); break;
            case types::ADOUBLE: c.f64.push_back(f64[r]); break;
            case types::AFLOAT: c.f32.push_back(f32[r]); break;
            default: c.text.push_back(text[r]); break;
            }
        }
        return c;
    }
    std::string str(size_t i) const {
        char b[64];
        switch (type) {
        case types::AINT32: std::snprintf(b, sizeof b, "%d", i32[i]); return b;
        case types::AINT64: std::snprintf(b, sizeof b, "%lld", i64[i]); return b;
        case types::ADOUBLE: std::snprintf(b, sizeof b, "%lf", f64[i]); return b;
        case types::AFLOAT: std::snprintf(b, sizeof b, "%f", f32[i]); return b;
        default: return text[i];
        }
    }
};
struct Table { std::vector<Column> cols; };

int sql_type(const std::string& t) {
    const std::string u = upper(t);
    if (u.rfind("BIGINT", 0) == 0 || u == "LONG") return types::AINT64;
    if (u.rfind("DOUBLE", 0) == 0 || u == "REAL") return types::ADOUBLE;
    if (u.rfind("FLOAT", 0) == 0) return types::AFLOAT;
    if (u.rfind("INT", 0) == 0 || u == "INTEGER" || u == "SMALLINT" || u == "TINYINT") return types::AINT32;
    return types::ASTR;
}

// the statements of the recorded message lists; anything else sets the error state (haserror), like a failed monetdbe_query
struct MiniSqlSource : DataSource {
    std::map<std::string, Table> tables;
    std::vector<Column> result;
    std::string err, root;
    bool failed = false;
    MiniSqlSource() { status = true; }
    void connect(Context* c) override { cxt = c; }
    void close() override {}
    bool haserror() override { return failed; }
    void getDSTable(const char*, void*) override {}
    void* getCol(int i, int = 0) override { return i >= 0 && (size_t)i < result.size() ? result[(size_t)i].data() : nullptr; }
    void fail(const std::string& m) { failed = true; err = m; last_error = err.c_str(); std::fprintf(stderr, "[aquery_host] SQL: %s\n", m.c_str()); }

    void exec(const char* q) override {
        failed = false;
        const std::string s = trim(q), u = upper(s);
        if (u.rfind("CREATE TABLE", 0) == 0) return create(s);
        if (u.rfind("COPY", 0) == 0 || u.rfind("LOAD DATA", 0) == 0) return load(s, u);
        if (u.rfind("SELECT", 0) == 0) return select(s, u);
        if (u.rfind("DROP TABLE", 0) == 0) { auto w = split(s, ' '); tables.erase(w.back()); return; }
        fail("statement not understood by the stand-in data source: " + s);
    }
    void create(const std::string& s) {
        const size_t lp = s.find('('), rp = s.rfind(')');
        if (lp == std::string::npos || rp == std::string::npos) return fail("CREATE TABLE without a column list");
        auto head = split(trim(s.substr(0, lp)), ' ');
        if (upper(s).find("IF NOT EXISTS") != std::string::npos && tables.count(head.back())) return;      // (monetdb_append_table creates its target this way)
        Table t;
        for (auto& def : split(s.substr(lp + 1, rp - lp - 1), ',')) {
            auto w = split(def, ' ');
            if (w.size() < 2) return fail("column definition: " + def);
            Column c; c.name = w[0]; c.type = sql_type(w[1]);
            t.cols.push_back(c);
        }
        tables[head.back()] = t;
    }
    static std::string quoted(const std::string& s, size_t from) {
        size_t a = s.find_first_of("'\"", from);
        if (a == std::string::npos) return "";
        size_t b = s.find(s[a], a + 1);
        return b == std::string::npos ? "" : s.substr(a + 1, b - a - 1);
    }
    // COPY [OFFSET k] INTO t FROM 'file' ... DELIMITERS 'c'      |      LOAD DATA INFILE "file" INTO TABLE t FIELDS TERMINATED BY "c"
    void load(const std::string& s, const std::string& u) {
        const size_t into = u.find(" INTO ");
        if (into == std::string::npos) return fail("load without INTO");
        std::istringstream rest(s.substr(into + 6));
        std::string tname; rest >> tname;
        if (upper(tname) == "TABLE") rest >> tname;
        auto it = tables.find(tname);
        if (it == tables.end()) return fail("unknown table " + tname);
        const size_t fpos = u.rfind("COPY", 0) == 0 ? u.find(" FROM ") : u.find("INFILE");
        std::string file = quoted(s, fpos == std::string::npos ? 0 : fpos);
        size_t dpos = u.find("DELIMITERS"); if (dpos == std::string::npos) dpos = u.find("TERMINATED BY");
        std::string delim = dpos == std::string::npos ? "," : quoted(s, dpos);
        if (delim.empty()) delim = ",";
        size_t skip = 1;                                                   // the header line (the reference emits COPY OFFSET 2)
        if (size_t o = u.find("OFFSET"); o != std::string::npos) skip = (size_t)std::max(0L, std::strtol(s.c_str() + o + 6, nullptr, 10) - 1);
        std::ifstream in(file[0] == '/' ? file : root + file);
        if (!in) return fail("cannot open " + file);
        std::string line;
        for (size_t k = 0; std::getline(in, line); ++k) {
            if (k < skip || trim(line).empty()) continue;
            auto f = split(line, delim[0]);
            for (size_t c = 0; c < it->second.cols.size(); ++c) it->second.cols[c].push(c < f.size() ? f[c] : "");
        }
    }
    void select(const std::string& s, const std::string& u) {
        const size_t from = u.find(" FROM ");
        if (from == std::string::npos) return fail("SELECT without FROM");
        auto names = split(s.substr(6, from - 6), ',');
        std::istringstream rest(s.substr(from + 6));
        std::string tname; rest >> tname;
        auto it = tables.find(tname);
        if (it == tables.end()) return fail("unknown table " + tname);
        Table& t = it->second;
        const size_t n = t.cols.empty() ? 0 : t.cols[0].size();
        std::vector<size_t> rows(n);
        for (size_t i = 0; i < n; ++i) rows[i] = i;
        if (size_t ob = u.find("ORDER BY"); ob != std::string::npos) {
            std::istringstream o(s.substr(ob + 8));
            std::string col, dir; o >> col >> dir;
            col = trim(col);
            const Column* kc = nullptr;
            for (auto& c : t.cols) if (c.name == col) kc = &c;
            if (!kc) return fail("ORDER BY unknown column " + col);
            const bool desc = upper(dir).rfind("DESC", 0) == 0;
            std::stable_sort(rows.begin(), rows.end(), [&](size_t a, size_t b) { return desc ? kc->key(a) > kc->key(b) : kc->key(a) < kc->key(b); });
        }
        result.clear();
        for (auto& nm : names) {
            const Column* c = nullptr;
            for (auto& x : t.cols) if (x.name == nm || nm == "*") { c = &x; if (nm != "*") break; result.push_back(x.take(rows)); }
            if (nm == "*") continue;
            if (!c) return fail("unknown column " + nm);
            result.push_back(c->take(rows));
        }
        cnt = (long long)rows.size();
    }
    // write-back of a result table (TableInfo::monetdb_append_table -> monetdbe_append in the reference, server/table_ext_monetdb.hpp:76):
    // the rows are copied into the catalog, so that later statements of the message list SELECT them back
    int append(const char* table, int ncols, const AppendColumn* cols) override {
        auto it = tables.find(table);
        if (it == tables.end()) { fail(std::string("append: unknown table ") + table); return -1; }
        Table& t = it->second;
        if ((int)t.cols.size() != ncols) { fail("append: column count"); return -1; }
        for (int c = 0; c < ncols; ++c) {
            Column& d = t.cols[(size_t)c];
            const uint32_t n = cols[c].count;
            auto as_double = [&](uint32_t i) -> double {
                switch (cols[c].type) {
                case types::AINT32: return static_cast<const int*>(cols[c].data)[i];
                case types::AINT64: return (double)static_cast<const long long*>(cols[c].data)[i];
                case types::AFLOAT: return static_cast<const float*>(cols[c].data)[i];
                case types::ADOUBLE: return static_cast<const double*>(cols[c].data)[i];
                case types::AINT16: return static_cast<const short*>(cols[c].data)[i];
                case types::AINT8: return static_cast<const signed char*>(cols[c].data)[i];
                case types::AUINT32: return static_cast<const unsigned*>(cols[c].data)[i];
                default: return 0;
                }
            };
            for (uint32_t i = 0; i < n; ++i) {
                switch (d.type) {
                case types::AINT32: d.i32.push_back((int)as_double(i)); break;
                case types::AINT64: d.i64.push_back(cols[c].type == types::AINT64 ? static_cast<const long long*>(cols[c].data)[i] : (long long)as_double(i)); break;
                case types::ADOUBLE: d.f64.push_back(as_double(i)); break;
                case types::AFLOAT: d.f32.push_back((float)as_double(i)); break;
                default: d.text.push_back(cols[c].type == types::ASTR ? static_cast<const char* const*>(cols[c].data)[i] : std::to_string(as_double(i))); break;
                }
            }
        }
        return 0;
    }
    void print_results(const char* sep = " ", const char* end = "\n", uint32_t limit = std::numeric_limits<uint32_t>::max()) override {
        const size_t n = result.empty() ? 0 : result[0].size();
        for (size_t i = 0; i < n && i < limit; ++i) {
            for (size_t c = 0; c < result.size(); ++c) std::printf("%s%s", result[c].str(i).c_str(), c + 1 < result.size() ? sep : "");
            std::printf("%s", end);
        }
    }
};

double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

} // namespace

int main(int argc, char** argv) {
    if (argc < 3) { std::fprintf(stderr, "usage: %s dll.so messages [--root DIR]\n", argv[0]); return 2; }
    std::string root;
    for (int i = 3; i + 1 < argc; ++i) if (std::string(argv[i]) == "--root") root = std::string(argv[i + 1]) + "/";
    Context* cxt = new Context();
    Config cfg{};
    cfg.backend_type = BACKEND_AQuery; cfg.running = 1; cfg.has_dll = 1;
    cxt->cfg = &cfg;
    MiniSqlSource ds;
    ds.root = root;
    ds.connect(cxt);
    cxt->curr_server = &ds;
    cxt->alt_server[BACKEND_AQuery] = cxt->alt_server[BACKEND_MonetDB] = &ds;

    std::ifstream in(argv[2]);
    if (!in) { std::fprintf(stderr, "[aquery_host] cannot open message list %s\n", argv[2]); return 2; }
    void* handle = dlopen(argv[1], RTLD_NOW);
    if (!handle) { std::fprintf(stderr, "[aquery_host] dlopen: %s\n", dlerror()); return 1; }
    if (auto init = reinterpret_cast<void (*)(Context*)>(dlsym(handle, "__AQ_Init_GC__"))) init(cxt);
    cxt->init_session();
    int rc = 0;
    std::string msg;
    while (std::getline(in, msg)) {
        if (msg.empty() || msg[0] == '#') continue;
        const double t0 = now_ms();
        switch (msg[0]) {
        case 'Q':
            ds.exec(msg.c_str() + 1);
            // the previous result set is gone (its buffers may be reused): the module drops the device mirrors of its columns
            if (auto changed = reinterpret_cast<void (*)(Context*)>(dlsym(handle, "__AQ_Result_Changed__"))) changed(cxt);
            cfg.stats.monet_time += (long long)((now_ms() - t0) * 1e6);
            break;
        case 'P':
            if (ds.haserror()) { std::fprintf(stderr, "[aquery_host] skipping %s: the data source is in its error state\n", msg.c_str()); rc |= 1; break; }
            if (auto fn = reinterpret_cast<code_snippet>(dlsym(handle, msg.c_str() + 1))) { rc |= fn(cxt); std::fflush(stdout); }
            else { std::fprintf(stderr, "[aquery_host] dlsym %s: %s\n", msg.c_str() + 1, dlerror()); rc |= 1; }
            cfg.stats.postproc_time += (long long)((now_ms() - t0) * 1e6);
            break;
        case 'O': {
            const long lim = std::strtol(msg.c_str() + 1, nullptr, 10);
            if (lim != 0 && !ds.haserror()) ds.print_results(" ", "\n", lim < 0 ? std::numeric_limits<uint32_t>::max() : (uint32_t)lim);
        } break;
        default: std::fprintf(stderr, "[aquery_host] message '%c' is outside this host's scope (modules, procedures, triggers)\n", msg[0]); break;
        }
    }
    if (auto fini = reinterpret_cast<void (*)(Context*)>(dlsym(handle, "__AQ_End_Session__"))) fini(cxt);
    cxt->end_session();
    dlclose(handle);
    std::fprintf(stderr, "[aquery_host] sql %.3f ms, post-processing %.3f ms\n", cfg.stats.monet_time / 1e6, cfg.stats.postproc_time / 1e6);
    return rc;
}
