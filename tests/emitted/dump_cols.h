// Test-only epilogue for the at-size emitted-shape modules: writes every column of an output table as raw little-endian bytes to
// <path>.<k> (vector-valued columns flattened, group after group), so that the Python side can compare the module's results with the
// oracle bit for bit.  Uses the public API only (size / begin()), i.e. the same host accesses `printall` would make.
#pragma once
#include <cstdio>
#include <string>

namespace aqtest {
template <class T> inline void dump_col(FILE* f, const ColRef<T>& c) {
    if constexpr (is_vector_type<T>) {
        for (uint32_t g = 0; g < c.size; ++g) {
            const auto& v = c[g];
            if (v.size) std::fwrite((const void*)v.begin(), sizeof(*v.begin()), v.size, f);
        }
    } else if (c.size) std::fwrite((const void*)c.begin(), sizeof(T), c.size, f);
}
template <class... Ts, size_t... Is> inline void dump_table_impl(const std::string& path, TableInfo<Ts...>& t, std::index_sequence<Is...>) {
    ((void)([&] {
        FILE* f = std::fopen((path + "." + std::to_string(Is)).c_str(), "wb");
        dump_col(f, t.template get_col<Is>());
        std::fclose(f);
    }()), ...);
}
template <class... Ts> inline void dump_table(const char* path, TableInfo<Ts...>& t) { dump_table_impl(path, t, std::index_sequence_for<Ts...>{}); }
} // namespace aqtest
