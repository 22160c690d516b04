// Minimal host for generated modules, in the shape of the reference's server/server.cpp 'P' message handling
// (:271 dlopen, :152-161 __AQ_Init_GC__, :301-305 dlsym + call): loads a module, installs an in-memory DataSource
// holding the reference's own tiny fixtures, runs the named entry points.
//   host_main <module.so> <dataset> <function>...
#include <dlfcn.h>

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "aquery/libaquery.h"

static std::vector<std::vector<int>> dataset(const std::string& name) {
    if (name == "moving_avg_asc")   // data/moving_avg.csv ordered by Month (what ASSUMING ASC Mont delivers)
        return {{1, 2, 3, 4, 5}, {100, 120, 140, 140, 130}};
    if (name == "moving_avg_desc")
        return {{5, 4, 3, 2, 1}, {130, 140, 140, 120, 100}};
    if (name == "stock")            // tests/stock.a:3-18
        return {{1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16}, {15, 19, 16, 17, 15, 13, 5, 8, 7, 13, 11, 14, 10, 5, 2, 5}};
    if (name == "stock_desc")
        return {{16, 15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1}, {5, 2, 5, 10, 14, 11, 13, 7, 8, 5, 13, 15, 17, 16, 19, 15}};
    if (name == "test_csv")         // data/test.csv columns a, b, c, d
        return {{1, 2, 2, 1, 1, 4, 2, 2, 1, 3, 1, 3, 2, 3, 2, 2, 2, 3, 2, 1},
                {1, 1, 4, 2, 2, 2, 1, 1, 2, 2, 2, 2, 1, 3, 2, 3, 4, 4, 3, 2},
                {2, 2, 3, 2, 3, 1, 3, 1, 3, 4, 3, 1, 4, 4, 3, 4, 1, 1, 2, 3},
                {2, 2, 4, 2, 4, 4, 3, 2, 4, 2, 3, 2, 2, 4, 1, 4, 2, 2, 2, 1}};
    if (name == "synthetic") {      // 200,000 rows, 1,000 groups: a = lcg % 1000, b = 7, c = lcg % 97, d = 1 (same LCG in tests/test_gpu_emitted.py)
        std::vector<std::vector<int>> t(4, std::vector<int>(200000));
        unsigned long long x = 12345;
        for (int i = 0; i < 200000; ++i) {
            x = x * 6364136223846793005ULL + 1442695040888963407ULL;
            t[0][i] = (int)((x >> 33) % 1000);
            t[1][i] = 7;
            t[2][i] = (int)((x >> 20) % 97);
            t[3][i] = 1;
        }
        return t;
    }
    if (name == "synthetic_big") {  // 3,000,000 rows, ~1,000,000 groups on column a (h2o Q5-like cardinality through the header loop)
        const int n = 3000000;
        std::vector<std::vector<int>> t(4, std::vector<int>(n));
        unsigned long long x = 777;
        for (int i = 0; i < n; ++i) {
            x = x * 6364136223846793005ULL + 1442695040888963407ULL;
            t[0][i] = (int)((x >> 33) % 1000003);
            t[1][i] = 7;
            t[2][i] = (int)((x >> 20) % 97);
            t[3][i] = 1;
        }
        return t;
    }
    // counter-based columns (splitmix64 of seed + row; the same in tests/test_gpu_emitted.py) for the at-size group-loop modules
    auto mix = [](unsigned long long z) { z += 0x9E3779B97F4A7C15ULL; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; return z ^ (z >> 31); };
    if (name == "trade" || name == "trade_small") {   // stocksymbol, price: 1e7 rows over 1e5 symbols (small: 5e4 rows over 300)
        const int n = name == "trade" ? 10000000 : 50000, S = name == "trade" ? 100000 : 300;
        std::vector<std::vector<int>> t(2, std::vector<int>(n));
        for (int i = 0; i < n; ++i) { t[0][i] = (int)(mix(1000 + (unsigned long long)i) % S); t[1][i] = 50 + (int)(mix(77000000000ULL + i) % 451); }
        return t;
    }
    if (name == "h2o9") {                             // id2, id4 in 1..100, v1 in 1..5, v2 in 1..15 (h2o G1_1e7_1e2 shapes): 1e4 groups
        const int n = 10000000;
        std::vector<std::vector<int>> t(4, std::vector<int>(n));
        for (int i = 0; i < n; ++i) {
            t[0][i] = 1 + (int)(mix(1ULL * i) % 100); t[1][i] = 1 + (int)(mix(5000000000ULL + i) % 100);
            t[2][i] = 1 + (int)(mix(9000000000ULL + i) % 5); t[3][i] = 1 + (int)(mix(13000000000ULL + i) % 15);
        }
        return t;
    }
    return {};
}

// "strings": column 0 holds `const char*` (12 rows over 4 distinct months; equal strings share ONE pointer, as a string heap hands
// them out), column 1 the same strings behind DISTINCT pointers (content equality only), column 2 sales
static const char* kMonths[] = {"jan", "feb", "mar", "apr"};
static const int kMonthOf[] = {2, 0, 2, 1, 3, 0, 0, 1, 2, 3, 3, 1};
static std::vector<std::string> g_copies;

int main(int argc, char** argv) {
    if (argc < 4) { std::fprintf(stderr, "usage: %s module.so dataset function...\n", argv[0]); return 2; }
    Context* cxt = new Context();
    Config cfg{};
    cfg.backend_type = BACKEND_AQuery;
    cxt->cfg = &cfg;
    std::vector<std::vector<int>> cols;
    std::vector<const char*> shared_ptrs, own_ptrs;
    std::vector<void*> ptrs;
    if (std::string(argv[2]) == "strings") {
        const int n = (int)(sizeof kMonthOf / sizeof kMonthOf[0]);
        g_copies.reserve(n);
        cols.push_back(std::vector<int>(n));
        for (int i = 0; i < n; ++i) {
            shared_ptrs.push_back(kMonths[kMonthOf[i]]);
            g_copies.emplace_back(kMonths[kMonthOf[i]]);
            own_ptrs.push_back(g_copies.back().c_str());
            cols[0][i] = 100 + 7 * i;
        }
        ptrs = {shared_ptrs.data(), own_ptrs.data(), cols[0].data()};
    } else {
        cols = dataset(argv[2]);
        for (auto& c : cols) ptrs.push_back(c.data());
    }
    if (cols.empty()) { std::fprintf(stderr, "unknown dataset %s\n", argv[2]); return 2; }
    ColumnDataSource ds;
    ds.set((long long)cols[0].size(), ptrs);
    ds.connect(cxt);
    cxt->curr_server = &ds;
    cxt->alt_server[BACKEND_AQuery] = &ds;

    void* handle = dlopen(argv[1], RTLD_NOW);
    if (!handle) { std::fprintf(stderr, "dlopen: %s\n", dlerror()); return 1; }
    if (auto init = reinterpret_cast<void (*)(Context*)>(dlsym(handle, "__AQ_Init_GC__"))) init(cxt);
    int rc = 0;
    for (int i = 3; i < argc; ++i) {
        auto fn = reinterpret_cast<code_snippet>(dlsym(handle, argv[i]));
        if (!fn) { std::fprintf(stderr, "dlsym %s: %s\n", argv[i], dlerror()); return 1; }
        rc |= fn(cxt);
        std::fflush(stdout);
    }
    if (auto fini = reinterpret_cast<void (*)(Context*)>(dlsym(handle, "__AQ_End_Session__"))) fini(cxt);   // the module's own device state
    cxt->end_session();
    return rc;
}
