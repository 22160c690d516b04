// libaquery.h -- the slice of the host ABI that generated post-processor modules see
// (reference server/libaquery.h:82-161 Context, server/DataSource_conn.h:27-54 DataSource,
// `__AQEXPORT__`, header.cxx `__AQ_Init_GC__`).  Only what the column-batch hot path needs: the message
// loop, triggers, stored procedures and the SQL back ends are out of scope.
#pragma once
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <limits>
#include <string>
#include <unordered_map>
#include <vector>

#include "gc.h"
#include "table.h"

enum Log_level : int { LOG_INFO, LOG_ERROR, LOG_SILENT };
enum Backend_Type : int { BACKEND_AQuery, BACKEND_MonetDB, BACKEND_MariaDB, BACKEND_DuckDB, BACKEND_SQLite, BACKEND_TOTAL };

struct QueryStats { long long monet_time; long long postproc_time; };
struct Config {
    int running, new_query, server_mode, backend_type, has_dll, n_buffers;
    QueryStats stats;
};

struct Context;
// one column handed back to the data source by TableInfo::monetdb_append_table (the reference builds a monetdbe_column {type, data,
// count, name} per column, server/table_ext_monetdb.hpp:88-116): `type` is the AQuery type tag, `data` a host array of `count` elements
struct AppendColumn { const char* name; int type; const void* data; uint32_t count; };
// the abstract data source generated code reads columns from: `server->cnt`, `server->getCol(i, type)`
struct DataSource {
    void* server = nullptr;
    Context* cxt = nullptr;
    bool status = false;
    char* query = nullptr;
    Backend_Type DataSourceType = BACKEND_AQuery;
    void* res = nullptr;
    void* ret_col = nullptr;
    long long cnt = 0;
    const char* last_error = nullptr;
    void* handle = nullptr;
    DataSource() = default;
    virtual void connect(Context*) = 0;
    virtual void exec(const char* q) = 0;
    virtual void* getCol(int col_idx, int type) = 0;
    virtual void getDSTable(const char* name, void* tbl) = 0;
    // write-back of a result table (monetdbe_append in the reference, server/table_ext_monetdb.hpp:76): 0 = stored
    virtual int append(const char* table, int ncols, const AppendColumn* cols) { (void)table; (void)ncols; (void)cols; return -1; }
    virtual void close() = 0;
    virtual bool haserror() = 0;
    virtual void print_results(const char* = " ", const char* = "\n", uint32_t = std::numeric_limits<uint32_t>::max()) {}
    virtual ~DataSource() {}
};

// In-memory data source standing in for the SQL engine's result set (the reference needs libmonetdbe, which
// is not available): columns are borrowed host arrays, exactly what monetdb getCol hands out.
struct ColumnDataSource : DataSource {
    std::vector<void*> columns;
    ColumnDataSource() { status = true; }
    void set(long long rows, std::vector<void*> cols) { cnt = rows; columns = std::move(cols); }
    void connect(Context* c) override { cxt = c; }
    void exec(const char*) override {}
    void* getCol(int i, int = 0) override { return i >= 0 && (size_t)i < columns.size() ? columns[i] : nullptr; }
    void getDSTable(const char*, void*) override {}
    void close() override {}
    bool haserror() override { return false; }
};

struct Context {
    typedef int (*printf_type)(const char* format, ...);
    Config* cfg = nullptr;
    void* curr_server = nullptr;
    void* alt_server[BACKEND_TOTAL] = {nullptr};
    Log_level log_level = LOG_INFO;
    const char* aquery_root_path = "";
    void* gc = nullptr;
    printf_type print = &std::printf;
    std::unordered_map<std::string, void*> tables;
    std::unordered_map<std::string, uColRef*> cols;

    Context() { gc = new GC(); }
    virtual ~Context() { delete static_cast<GC*>(gc); }
    template <class... Types> void log(Types... args) { if (log_level == LOG_INFO) print(args...); }
    template <class... Types> void err(Types... args) { if (log_level <= LOG_ERROR) print(args...); }
    void init_session() {}
    // end of a module's session: drop the device mirrors of borrowed columns (their host buffers go away) and the groupings that
    // HashTableFactory::get made for the module (device handles, row ids, key vectors: 8 GB per 1e9-row group-by otherwise)
    void end_session() { aq::dev::Runtime::get().release_session(); aq::dev::Runtime::get().drop_pins(); }
};

#define __DLLEXPORT__
#define __AQEXPORT__(_Ty) extern "C" _Ty __DLLEXPORT__
typedef int (*code_snippet)(void*);

// ---- TableInfo::monetdb_append_table (declared in table.h) ----------------------------------------------------------------------------
namespace aq {
inline const char* sql_type_name(int t) {     // reference types::SQL_Type, server/types.h:77-78
    static const char* names[] = {"INT", "REAL", "TEXT", "DOUBLE", "DOUBLE", "BIGINT", "HUGEINT", "SMALLINT", "DATE", "TIME", "TINYINT",
                                  "INT", "BIGINT", "HUGEINT", "SMALLINT", "TINYINT", "BOOL", "HUGEINT", "TIMESTAMP", "CHAR", "TEXT", "NULL", "ERROR"};
    return t >= 0 && t < (int)(sizeof names / sizeof names[0]) ? names[t] : "ERROR";
}
}
template <class... Types>
void TableInfo<Types...>::monetdb_append_table(void* srv, const char* alt_name) {
    if (!alt_name) alt_name = this->name;
    auto* ds = static_cast<DataSource*>(srv);
    if (!ds || sizeof...(Types) == 0) { std::puts("Error! Empty table."); return; }
    std::vector<const void*> ptrs;
    std::apply([&](auto&... c) { (ptrs.push_back(c.container), ...); }, cols);
    aq::dev::Runtime::get().fetch_all(ptrs);
    std::string create = std::string("CREATE TABLE IF NOT EXISTS ") + alt_name + " (";
    std::vector<AppendColumn> ac;
    std::apply([&](auto&... c) {
        ((create += std::string(c.name) + ' ' + aq::sql_type_name((int)types::Types<typename std::decay_t<decltype(c)>::value_t>::getType()) + ", ",
          ac.push_back(AppendColumn{c.name, (int)types::Types<typename std::decay_t<decltype(c)>::value_t>::getType(), (const void*)c.container, c.size})), ...);
    }, cols);
    create.resize(create.size() - 2);
    create += ")";
    ds->exec(create.c_str());
    if (ds->haserror()) { std::puts(ds->last_error ? ds->last_error : "Error! CREATE TABLE failed."); return; }
    if (ds->append(alt_name, (int)ac.size(), ac.data()) != 0) std::puts("Error! The data source does not take appended tables.");
}
