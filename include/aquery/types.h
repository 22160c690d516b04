// types.h -- type vocabulary of the AQuery library API (clean-room; same names and results as the
// reference's server/types.h so that generated code compiles unchanged).  Result-type rules:
//   GetLongType  (reference server/types.h:205-210)  unsigned -> unsigned __int128, fp -> double, else __int128
//   GetFPType    (:199-204)  4-byte types -> float, everything else -> double
//   Coercion     (:264-275)  wider wins; equal width: fp wins, else signed unless both unsigned
// The same rules are exported by the C-ABI (aqg_long_type / aqg_fp_type / aqg_coercion) and checked
// against the reference in tests/test_cabi.py and tests/test_oracle_vs_ref.py.
#pragma once
#include <cstddef>
#include <cstdint>
#include <limits>
#include <string>
#include <string_view>
#include <tuple>
#include <type_traits>
#include <utility>

using std::size_t;

#if defined(__SIZEOF_INT128__) && !defined(_WIN32)
#define __AQ__HAS__INT128__
#endif

template <class T> struct vector_base {};

template <class T> constexpr static inline bool is_vector(const T&) { return false; }
template <class T> struct is_vector_impl : std::false_type {};
template <class T> constexpr static bool is_vector_type = is_vector_impl<T>::value;

template <class T> constexpr size_t aq_szof = sizeof(T);
template <> inline constexpr size_t aq_szof<void> = 0;

// "same" in the reference's sense: both bool, or same signedness + same floating-ness + same size;
// classes are the same when one derives from the other
template <class A, class B> struct aqis_same_impl {
    static constexpr bool compute() {
        if constexpr (std::is_same_v<A, bool> || std::is_same_v<B, bool>) return std::is_same_v<A, bool> && std::is_same_v<B, bool>;
        else if constexpr (std::is_class_v<A> && std::is_class_v<B>) return std::is_base_of_v<A, B> || std::is_base_of_v<B, A>;
        else if constexpr (std::is_class_v<A> || std::is_class_v<B>) return false;
        else return std::is_signed_v<A> == std::is_signed_v<B> && std::is_floating_point_v<A> == std::is_floating_point_v<B> &&
                    aq_szof<A> == aq_szof<B>;
    }
    static constexpr bool value = compute();
};
template <class A, class B, class... R> constexpr bool aqis_same = aqis_same_impl<A, B>::value && aqis_same<B, R...>;
template <class A, class B> constexpr bool aqis_same<A, B> = aqis_same_impl<A, B>::value;

namespace types {
// tags in the order of the reference's server/aquery_types.h:1-5 (they cross the C-ABI as ints)
enum Type_t : int {
    AINT32, AFLOAT, ASTR, ADOUBLE, ALDOUBLE, AINT64, AINT128, AINT16, ADATE, ATIME, AINT8, AUINT32, AUINT64, AUINT128,
    AUINT16, AUINT8, ABOOL, VECTOR, ATIMESTAMP, ACHAR, ASV, NONE, ERROR
};

struct date_t {
    unsigned char day = 0, month = 0;
    short year = 0;
    date_t() = default;
    date_t(unsigned char d, unsigned char m, short y) : day(d), month(m), year(y) {}
    bool operator==(const date_t& o) const { return day == o.day && month == o.month && year == o.year; }
    bool operator<(const date_t& o) const { return std::tie(year, month, day) < std::tie(o.year, o.month, o.day); }
    constexpr static unsigned string_length() { return 11; }
};
struct time_t {
    unsigned int ms = 0;
    unsigned char seconds = 0, minutes = 0, hours = 0;
    time_t() = default;
    time_t(unsigned int ms_, unsigned char s, unsigned char m, unsigned char h) : ms(ms_), seconds(s), minutes(m), hours(h) {}
    bool operator==(const time_t& o) const { return ms == o.ms && seconds == o.seconds && minutes == o.minutes && hours == o.hours; }
    bool operator<(const time_t& o) const { return std::tie(hours, minutes, seconds, ms) < std::tie(o.hours, o.minutes, o.seconds, o.ms); }
    constexpr static unsigned string_length() { return 16; }
};
struct timestamp_t {
    date_t date;
    time_t time;
    timestamp_t() = default;
    timestamp_t(const date_t& d, const time_t& t) : date(d), time(t) {}
    bool operator==(const timestamp_t& o) const { return date == o.date && time == o.time; }
    constexpr static unsigned string_length() { return date_t::string_length() + time_t::string_length(); }
};

#ifdef __AQ__HAS__INT128__
using LL_Type = __int128_t;
using ULL_Type = __uint128_t;
#else
using LL_Type = long long;
using ULL_Type = unsigned long long;
#endif

template <class T> struct Types {
    typedef T type;
    inline constexpr static Type_t getType() {
        if constexpr (aqis_same<int, T>) return AINT32;
        else if constexpr (aqis_same<float, T>) return AFLOAT;
        else if constexpr (aqis_same<const char*, T>) return ASTR;
        else if constexpr (aqis_same<double, T>) return ADOUBLE;
        else if constexpr (aqis_same<long double, T>) return ALDOUBLE;
        else if constexpr (aqis_same<long, T>) return AINT64;
        else if constexpr (aqis_same<short, T>) return AINT16;
        else if constexpr (aqis_same<date_t, T>) return ADATE;
        else if constexpr (aqis_same<time_t, T>) return ATIME;
        else if constexpr (aqis_same<unsigned char, T>) return AUINT8;
        else if constexpr (aqis_same<char, T>) return AINT8;
        else if constexpr (aqis_same<unsigned int, T>) return AUINT32;
        else if constexpr (aqis_same<unsigned long, T>) return AUINT64;
        else if constexpr (aqis_same<unsigned short, T>) return AUINT16;
        else if constexpr (aqis_same<bool, T>) return ABOOL;
        else if constexpr (aqis_same<timestamp_t, T>) return ATIMESTAMP;
        else if constexpr (aqis_same<std::string_view, T> || aqis_same<std::string, T>) return ASV;
#ifdef __AQ__HAS__INT128__
        else if constexpr (aqis_same<__int128_t, T>) return AINT128;
        else if constexpr (aqis_same<__uint128_t, T>) return AUINT128;
#endif
        else if constexpr (is_vector_type<T>) return VECTOR;
        else return NONE;
    }
};

template <class T> struct GetFPTypeImpl { using type = std::conditional_t<sizeof(T) == sizeof(float), float, double>; };
template <class T> using GetFPType = typename GetFPTypeImpl<std::decay_t<T>>::type;

template <class T> struct GetLongTypeImpl {
    using type = std::conditional_t<std::is_unsigned_v<T>, ULL_Type, std::conditional_t<std::is_floating_point_v<T>, double, LL_Type>>;
};
template <class T> using GetLongType = typename GetLongTypeImpl<std::decay_t<T>>::type;

template <class T> struct GetSignedType_impl {
    using type = std::conditional_t<aqis_same<T, unsigned char>, char,
                 std::conditional_t<aqis_same<T, unsigned short>, short,
                 std::conditional_t<aqis_same<T, unsigned int>, int,
                 std::conditional_t<aqis_same<T, unsigned long>, long,
#ifdef __AQ__HAS__INT128__
                 std::conditional_t<aqis_same<T, unsigned __int128>, __int128_t, T>
#else
                 T
#endif
                 >>>>;
};
template <class T> using GetSignedType = typename GetSignedType_impl<T>::type;

template <class T1, class T2, class... Ts> struct Coercion { using type = typename Coercion<T1, typename Coercion<T2, Ts...>::type>::type; };
template <class T1, class T2> struct Coercion<T1, T2> {
private:
    static constexpr bool fp1 = std::is_floating_point_v<T1>, fp2 = std::is_floating_point_v<T2>;
    static constexpr bool u1 = std::is_unsigned_v<T1>, u2 = std::is_unsigned_v<T2>;
    // the wider operand; on a tie the floating one, else the signed one (T2 when T1 is unsigned)
    using wide = std::conditional_t<(sizeof(T1) < sizeof(T2)), T2,
                 std::conditional_t<(sizeof(T1) > sizeof(T2)), T1,
                 std::conditional_t<fp1, T1, std::conditional_t<fp2, T2, std::conditional_t<u1, T2, T1>>>>>;
    using arith = std::conditional_t<(fp1 || fp2), GetFPType<wide>, std::conditional_t<!(u1 && u2), GetSignedType<wide>, wide>>;
    static constexpr bool same = aqis_same<T1, T2>;
    static constexpr bool stringy = aqis_same<T1, const char*> || aqis_same<T2, const char*>;
public:
    using type = std::conditional_t<same, T1, std::conditional_t<stringy, const char*, arith>>;
};
} // namespace types

// C string view used for string columns (pointer-sized; compared by content)
union astring_view {
    const unsigned char* str = nullptr;
    const signed char* sstr;
    const char* rstr;
    size_t ptr;
    constexpr astring_view(const char* s) noexcept : rstr(s) {}
    constexpr astring_view(const signed char* s) noexcept : sstr(s) {}
    constexpr astring_view(const unsigned char* s) noexcept : str(s) {}
    constexpr astring_view() noexcept = default;
    bool operator==(const astring_view& r) const {
        const unsigned char *a = str, *b = r.str;
        while (*a && *b) { if (*a != *b) return false; ++a; ++b; }
        return !(*a || *b);
    }
    operator const char*() const { return rstr; }
    operator const unsigned char*() const { return str; }
    operator const signed char*() const { return sstr; }
};

template <class T> constexpr bool is_cstr() {
    using D = std::decay_t<T>;
    return std::is_same_v<D, const char*> || std::is_same_v<D, char*> || std::is_same_v<D, const signed char*> || std::is_same_v<D, signed char*> ||
           std::is_same_v<D, const unsigned char*> || std::is_same_v<D, unsigned char*> || std::is_same_v<D, astring_view>;
}

// rT<T...> from lT<T...>  (e.g. transTypes<record<int,int>, hasher> = hasher<int,int>)
template <class lT, template <typename...> class rT> struct transTypes_s;
template <template <typename...> class lT, typename... T, template <typename...> class rT> struct transTypes_s<lT<T...>, rT> { using type = rT<T...>; };
template <class lT, template <typename...> class rT> using transTypes = typename transTypes_s<lT, rT>::type;

template <class... Types> using record = std::tuple<Types...>;

template <class T> struct decayS { using type = std::decay_t<T>; };
template <template <typename...> class T, typename... Types> struct decayS<T<Types...>> { using type = T<std::decay_t<Types>...>; };
template <class T> using decays = typename decayS<std::decay_t<T>>::type;
template <class T> using decay_inner = typename decayS<T>::type;

template <class, template <class...> class T> struct instance_of_impl : std::false_type {};
template <class... T1, template <class...> class T2> struct instance_of_impl<T2<T1...>, T2> : std::true_type {};
template <class T1, template <class...> class T2> constexpr bool instance_of = instance_of_impl<T1, T2>::value;

template <template <class...> class T, class... Types> struct decayed_impl { typedef T<Types...> type; };
template <template <typename...> class VT, class... Types> using decayed_t = typename decayed_impl<VT, Types...>::type;

template <class First = void, class... Rest> struct get_first_impl { typedef First first; };
template <class... T> using get_first = typename get_first_impl<T...>::first;
template <class T> struct value_type_impl { typedef T type; };
template <template <class...> class VT, class... V> struct value_type_impl<VT<V...>> { typedef get_first<V...> type; };
template <class T> using value_type = typename value_type_impl<T>::type;
template <class T> struct value_type_rec_impl { typedef T type; };
template <template <class...> class VT, class... V> struct value_type_rec_impl<VT<V...>> {
    typedef std::conditional_t<std::is_base_of_v<vector_base<get_first<V...>>, VT<V...>>, typename value_type_rec_impl<get_first<V...>>::type, VT<V...>> type;
};
template <class T> using value_type_r = typename value_type_rec_impl<T>::type;

// NULL sentinels (reference server/types.h:452-462): INT_MIN, -NaN
template <class T> struct nullval_impl { constexpr static T value = 0; };
template <> struct nullval_impl<int> { constexpr static int value = std::numeric_limits<int>::min(); };
template <> struct nullval_impl<float> { constexpr static float value = -std::numeric_limits<float>::quiet_NaN(); };
template <> struct nullval_impl<double> { constexpr static double value = -std::numeric_limits<double>::quiet_NaN(); };
template <class T> constexpr static T nullval = nullval_impl<T>::value;

template <class T> inline constexpr uint32_t aq_fp_precision = std::is_same_v<T, float> ? 7 : (std::is_same_v<T, double> ? 16 : 0);

template <int i, template <int...> class rT, class Seq = std::make_integer_sequence<int, i>> struct applyIntegerSequence_impl;
template <int i, template <int...> class rT, int... Is> struct applyIntegerSequence_impl<i, rT, std::integer_sequence<int, Is...>> { using type = rT<Is...>; };
template <int i, template <int...> class rT> using applyIntegerSequence = typename applyIntegerSequence_impl<i, rT>::type;
