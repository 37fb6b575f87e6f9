// ungar_amd :: compile-time variable hierarchies (layout contract of the hot path).
//
// Same user-facing surface and the same offset rules as the reference's Hana-based
//   include/ungar/variable.hpp   (UNGAR_VARIABLE :1013-1016, var_c :896-903, `<<=` :222-266,
//                                  `*` / `,` :914-1011, operator() bypass lookup :423-500,
//                                  Size()/Index() :142-155, ForEach :567-579)
// re-implemented on plain C++20 (class-type NTTP names, tuples, constexpr recursion) -- Boost.Hana
// is not available on the target boxes.  Rules:
//   * child index = parent index + sum of the sizes of the preceding siblings;
//   * `K * var` is an array laid out contiguously with stride var.Size();
//   * a leaf of size Q is a unit quaternion occupying 4 scalars;
//   * lookup `root(a, i, b, j...)` walks by unique NAME; each integer picks the element of the next
//     array met on the way to the preceding name; an ambiguous name is a compile-time error.
// Indices/offsets are checked bit-exact against the reference's own engine (tests/golden/).
#pragma once

#include <algorithm>
#include <array>
#include <cstddef>
#include <ranges>
#include <string>
#include <string_view>
#include <tuple>
#include <type_traits>
#include <utility>

#include "hana_support.hpp"
#include "io/logging.hpp"
#include "linalg.hpp"

namespace Ungar {

using namespace std::literals;  // the reference exports the std literal operators to `using namespace Ungar` code

using real_t = double;
using index_t = Eigen::Index;
using time_step_t = index_t;

/// Sentinel leaf size: "unit quaternion, 4 scalars" (reference utils.hpp:232, Q = Eigen::Dynamic).
inline constexpr index_t Q = Eigen::Dynamic;

// ---- integral constants: 30_c, N + 1_c, N * x ---------------------------------------------------
template <index_t V>
struct Const {
    static constexpr index_t value = V;
    using value_type = index_t;
    constexpr operator index_t() const {  // NOLINT
        return V;
    }
    constexpr index_t operator()() const {
        return V;
    }
};
template <index_t A, index_t B>
constexpr Const<A + B> operator+(Const<A>, Const<B>) {
    return {};
}
template <index_t A, index_t B>
constexpr Const<A - B> operator-(Const<A>, Const<B>) {
    return {};
}
template <index_t A, index_t B>
constexpr Const<A * B> operator*(Const<A>, Const<B>) {
    return {};
}
template <index_t A, index_t B>
constexpr Const<A / B> operator/(Const<A>, Const<B>) {
    return {};
}
template <class T>
inline constexpr bool is_const_v = false;
template <index_t V>
inline constexpr bool is_const_v<Const<V>> = true;

inline namespace Literals {
namespace detail {
template <char... Cs>
constexpr index_t ParseDigits() {
    index_t v = 0;
    ((v = v * 10 + static_cast<index_t>(Cs - '0')), ...);
    return v;
}
}  // namespace detail
template <char... Cs>
constexpr auto operator""_c() {
    return Const<detail::ParseDigits<Cs...>()>{};
}
constexpr index_t operator""_idx(const unsigned long long i) {
    return static_cast<index_t>(i);
}
constexpr time_step_t operator""_step(const unsigned long long i) {
    return static_cast<time_step_t>(i);
}
}  // namespace Literals

template <index_t V>
using idx_ = Const<V>;
template <index_t V>
inline constexpr Const<V> idx_c{};

template <std::integral I>
constexpr auto enumerate(const I n) {
    return std::views::iota(static_cast<I>(0), n);
}
template <index_t V>
constexpr auto enumerate(Const<V>) {
    return std::views::iota(static_cast<index_t>(0), V);
}

/// `range | cast_to<T>` (reference data_types.hpp:393-396): the elements converted to T
template <class T>
inline constexpr auto cast_to = std::views::transform([](auto&& el) -> decltype(auto) { return static_cast<T>(std::forward<decltype(el)>(el)); });

template <std::size_t N>
struct fixed_string {
    constexpr fixed_string(const char (&str)[N]) {  // NOLINT
        std::copy_n(str, N, data);
    }
    constexpr std::size_t size() const {
        return N;
    }
    constexpr const char* c_str() const {
        return data;
    }
    constexpr std::string_view view() const {
        return {data, N - 1};
    }
    template <std::size_t M>
    constexpr bool operator==(const fixed_string<M>& o) const {
        return view() == o.view();
    }
    char data[N]{};
};

// ---- the variable tree ------------------------------------------------------------------------------
/// Results of Variable::GetOpt.
template <class V>
struct VariableReference {
    V variable;
    constexpr const V& get() const { return variable; }
};
template <class V>
struct VariableOptional {
    VariableReference<V> reference;
    static constexpr bool has_value() { return true; }
    constexpr explicit operator bool() const { return true; }
    constexpr const VariableReference<V>& value() const { return reference; }
    constexpr const VariableReference<V>& operator*() const { return reference; }
    constexpr const VariableReference<V>* operator->() const { return &reference; }
};
struct NoVariable {
    static constexpr bool has_value() { return false; }
    constexpr explicit operator bool() const { return false; }
};

template <fixed_string NAME, index_t SIZE, class... Children>
class Variable;

/// `K * var`
template <class Var, index_t K>
struct VariableProductExpr {
    Var var;
};
/// `(a, b, K * c, ...)`
template <class... Items>
struct VariableTuple {
    std::tuple<Items...> items;
};

namespace detail {

template <class T>
inline constexpr bool is_variable_v = false;
template <fixed_string N, index_t S, class... C>
inline constexpr bool is_variable_v<Variable<N, S, C...>> = true;
template <class T>
inline constexpr bool is_product_v = false;
template <class V, index_t K>
inline constexpr bool is_product_v<VariableProductExpr<V, K>> = true;
template <class T>
inline constexpr bool is_tuple_v = false;
template <class... I>
inline constexpr bool is_tuple_v<VariableTuple<I...>> = true;
template <class T>
inline constexpr bool is_array_v = false;
template <class V, std::size_t K>
inline constexpr bool is_array_v<std::array<V, K>> = true;

/// Stored form of a child: a Variable, or std::array<Variable, K>.
template <class Item>
struct ChildOf {
    using type = Item;
};
template <class V, index_t K>
struct ChildOf<VariableProductExpr<V, K>> {
    using type = std::array<V, static_cast<std::size_t>(K)>;
};

template <class Child>
constexpr index_t ChildSize() {
    if constexpr (is_array_v<Child>) return static_cast<index_t>(std::tuple_size_v<Child>) * Child::value_type::Size();
    else return Child::Size();
}

template <class Child, fixed_string NAME>
constexpr index_t CountName();

}  // namespace detail

namespace Concepts {
template <class T>
concept Variable = detail::is_variable_v<std::remove_cvref_t<T>>;
}

template <fixed_string NAME, index_t SIZE, class... Children>
class Variable {
  public:
    static_assert(sizeof...(Children) == 0 || SIZE == 0, "only branch variables (SIZE 0) have sub-variables");
    constexpr Variable() = default;

    static constexpr auto Name() {
        return NAME;
    }
    static constexpr index_t Size() {
        if constexpr (SIZE != 0) return SIZE == Q ? 4 : SIZE;
        else return (index_t{0} + ... + detail::ChildSize<Children>());
    }
    constexpr index_t Index() const {
        return _index;
    }
    static constexpr bool IsLeaf() {
        return SIZE != 0;
    }
    static constexpr bool IsBranch() {
        return SIZE == 0;
    }
    static constexpr bool IsScalar() {
        return SIZE == 1;
    }
    static constexpr bool IsQuaternion() {
        return SIZE == Q;
    }
    static constexpr bool IsVector() {
        return !IsScalar() && !IsQuaternion();
    }

    // ---- composition: var_c<"x"> <<= (a, b, K * c) ---------------------------------------------------
    template <class Item>
        requires(detail::is_variable_v<Item> || detail::is_product_v<Item>)
    constexpr auto operator<<=(const Item& item) const {
        return Compose(std::tuple<Item>{item});
    }
    template <class... Items>
    constexpr auto operator<<=(const VariableTuple<Items...>& t) const {
        return Compose(t.items);
    }

    // ---- lookup: root(var, i, var2, j, ...) ------------------------------------------------------------
    template <class... Args>
    constexpr auto operator()(const Args&... args) const {
        return Resolve(*this, args...);
    }
    /// m-variable spelling of the same lookup (reference mvariable.hpp: var.Get(path...)).
    template <class... Args>
    constexpr auto Get(const Args&... args) const {
        return Resolve(*this, args...);
    }
    /// Optional lookup (reference mvariable.hpp:90-111 and the GetOpt members of its three m-variable macros): whether `var` lives in this
    /// hierarchy is a compile-time fact, so the result is either an engaged VariableOptional -- `GetOpt(var, i)->get()` / `.value().get()` is the
    /// sub-variable, as with the reference's optional of a reference wrapper -- or NoVariable (`has_value()` false), never an error.
    template <class Target, class... Idx>
    constexpr auto GetOpt(const Target& target, const Idx... idx) const {
        if constexpr (Name() == Target::Name() || CountDescendantsNamed<Target::Name()>() == 1) {
            using Found = decltype(Resolve(*this, target, idx...));
            return VariableOptional<Found>{VariableReference<Found>{Resolve(*this, target, idx...)}};
        } else {
            return NoVariable{};
        }
    }
    /// Verbose form: X.At<"x">(1)
    template <fixed_string CHILD, class... Idx>
    constexpr auto At(const Idx... idx) const {
        return Find<CHILD>(*this, static_cast<index_t>(idx)...).first;
    }

    /// Pre-order traversal (self, then children in declaration order, array elements in order).
    template <class F>
    constexpr void ForEach(F&& f) const {
        f(*this);
        std::apply(
            [&](const auto&... child) {
                (ForEachChild(child, f), ...);
            },
            _children);
    }

    constexpr auto CloneWithIndexOffset(const index_t offset) const {
        Variable v = *this;
        v._index += offset;
        std::apply([&](auto&... child) { (OffsetChild(child, offset), ...); }, v._children);
        return v;
    }

    template <fixed_string N>
    static constexpr index_t CountDescendantsNamed() {
        return (index_t{0} + ... + detail::CountName<Children, N>());
    }

    const auto& Children_() const {
        return _children;
    }

  private:
    template <fixed_string N2, index_t S2, class... C2>
    friend class Variable;

    template <class Child>
    static constexpr void OffsetChild(Child& child, const index_t offset) {
        if constexpr (detail::is_array_v<Child>) {
            for (auto& e : child) e = e.CloneWithIndexOffset(offset);
        } else {
            child = child.CloneWithIndexOffset(offset);
        }
    }
    template <class Child, class F>
    static constexpr void ForEachChild(const Child& child, F& f) {
        if constexpr (detail::is_array_v<Child>) {
            for (const auto& e : child) e.ForEach(f);
        } else {
            child.ForEach(f);
        }
    }

    template <class... Items>
    constexpr auto Compose(const std::tuple<Items...>& items) const {
        static_assert(SIZE == 0 && sizeof...(Children) == 0, "sub-variables can only be attached to an empty branch variable");
        Variable<NAME, 0, typename detail::ChildOf<Items>::type...> out;
        out._index = _index;
        index_t offset = _index;
        std::apply(
            [&](const auto&... item) {
                std::size_t slot = 0;
                ((PlaceItem<Items>(out, item, offset, slot)), ...);
                (void)slot;
            },
            items);
        return out;
    }
    template <class Item, class Out>
    static constexpr void PlaceItem(Out& out, const Item& item, index_t& offset, std::size_t& slot) {
        PlaceAt(out, item, offset, slot, std::make_index_sequence<std::tuple_size_v<decltype(out._children)>>{});
        ++slot;
    }
    template <class Out, class Item, std::size_t... Is>
    static constexpr void PlaceAt(Out& out, const Item& item, index_t& offset, const std::size_t slot, std::index_sequence<Is...>) {
        ((Is == slot ? PlaceInto(std::get<Is>(out._children), item, offset) : void()), ...);
    }
    template <class Slot, class Item>
    static constexpr void PlaceInto(Slot& slotRef, const Item& item, index_t& offset) {
        if constexpr (detail::is_product_v<Item>) {
          if constexpr (std::is_same_v<Slot, typename detail::ChildOf<Item>::type>) {
            for (std::size_t i = 0; i < slotRef.size(); ++i) {
                slotRef[i] = item.var.CloneWithIndexOffset(offset - item.var.Index());
                offset += std::remove_cvref_t<decltype(item.var)>::Size();
            }
          }
        } else if constexpr (std::is_same_v<Slot, Item>) {
            slotRef = item.CloneWithIndexOffset(offset - item.Index());
            offset += Item::Size();
        }
    }

    // ---- name-based search --------------------------------------------------------------------------------
    /// Finds the unique descendant (or self) named TARGET starting at `from`, consuming indices for
    /// the arrays met on the way.  Returns {variable, number of indices consumed}.
    template <fixed_string TARGET, class From, class... Idx>
    static constexpr auto Find(const From& from, const Idx... idx) {
        if constexpr (From::Name() == TARGET) {
            return std::pair{from, std::size_t{0}};
        } else {
            static_assert(From::template CountDescendantsNamed<TARGET>() >= 1, "variable not found in this hierarchy");
            return FindInChildren<TARGET, 0>(from, idx...);
        }
    }
    template <fixed_string TARGET, std::size_t I, class From, class... Idx>
    static constexpr auto FindInChildren(const From& from, const Idx... idx) {
        using Tuple = std::remove_cvref_t<decltype(from._children)>;
        static_assert(I < std::tuple_size_v<Tuple>, "variable not found in this hierarchy");
        using Child = std::tuple_element_t<I, Tuple>;
        constexpr index_t here = detail::CountName<Child, TARGET>();
        if constexpr (here == 0) {
            return FindInChildren<TARGET, I + 1>(from, idx...);
        } else {
            // uniqueness among siblings (reference variable.hpp:736-738 static_asserts on ambiguity)
            static_assert(From::template CountDescendantsNamed<TARGET>() == here, "ambiguous variable name: disambiguate with a longer path");
            const auto& child = std::get<I>(from._children);
            if constexpr (detail::is_array_v<Child>) {
                static_assert(sizeof...(Idx) >= 1, "an array lies on the path to this variable: pass its element index");
                return FindInArray<TARGET>(child, idx...);
            } else {
                return Find<TARGET>(child, idx...);
            }
        }
    }
    template <fixed_string TARGET, class Array, class... Rest>
    static constexpr auto FindInArray(const Array& arr, const index_t i, const Rest... rest) {
        auto r = Find<TARGET>(arr[static_cast<std::size_t>(i)], rest...);
        return std::pair{r.first, r.second + 1};
    }

    /// Processes `var, idx..., var2, idx2...` left to right.
    template <class From>
    static constexpr auto Resolve(const From& from) {
        return from;
    }
    template <class From, class Target, class... Rest>
    static constexpr auto Resolve(const From& from, const Target&, const Rest&... rest) {
        static_assert(detail::is_variable_v<Target>, "lookup arguments must start with a variable");
        return ResolveIdx<Target::Name()>(from, std::tuple<>{}, rest...);
    }
    // gather the integer arguments following a variable
    template <fixed_string TARGET, class From, class... Got>
    static constexpr auto ResolveIdx(const From& from, const std::tuple<Got...>& got) {
        return std::apply([&](auto... i) { return Find<TARGET>(from, static_cast<index_t>(i)...).first; }, got);
    }
    template <fixed_string TARGET, class From, class... Got, class Next, class... Rest>
    static constexpr auto ResolveIdx(const From& from, const std::tuple<Got...>& got, const Next& next, const Rest&... rest) {
        if constexpr (detail::is_variable_v<Next>) {
            const auto found = std::apply([&](auto... i) { return Find<TARGET>(from, static_cast<index_t>(i)...).first; }, got);
            return Resolve(found, next, rest...);
        } else {
            return ResolveIdx<TARGET>(from, std::tuple_cat(got, std::tuple<index_t>{static_cast<index_t>(next)}), rest...);
        }
    }

    index_t _index = 0;
    std::tuple<Children...> _children{};
};

namespace detail {
template <class Child, fixed_string NAME>
constexpr index_t CountName() {
    if constexpr (is_array_v<Child>) {
        using V = typename Child::value_type;
        return (V::Name() == NAME ? 1 : 0) + V::template CountDescendantsNamed<NAME>();
    } else {
        return (Child::Name() == NAME ? 1 : 0) + Child::template CountDescendantsNamed<NAME>();
    }
}
}  // namespace detail

/// var_c<"name", size> -- a leaf; var_c<"name"> -- an empty branch to be filled with `<<=`.
template <fixed_string NAME, index_t SIZE = 0>
inline constexpr Variable<NAME, SIZE> var_c{};

template <index_t K, fixed_string N, index_t S, class... C>
constexpr auto operator*(Const<K>, const Variable<N, S, C...>& v) {
    return VariableProductExpr<Variable<N, S, C...>, K>{v};
}

template <class A, class B>
    requires((detail::is_variable_v<A> || detail::is_product_v<A>) && (detail::is_variable_v<B> || detail::is_product_v<B>))
constexpr auto operator,(const A& a, const B& b) {
    return VariableTuple<A, B>{{a, b}};
}
template <class... Items, class B>
    requires(detail::is_variable_v<B> || detail::is_product_v<B>)
constexpr auto operator,(const VariableTuple<Items...>& t, const B& b) {
    return VariableTuple<Items..., B>{std::tuple_cat(t.items, std::tuple<B>{b})};
}
template <class A, class... Items>
    requires(detail::is_variable_v<A> || detail::is_product_v<A>)
constexpr auto operator,(const A& a, const VariableTuple<Items...>& t) {
    return VariableTuple<A, Items...>{std::tuple_cat(std::tuple<A>{a}, t.items)};
}

#define UNGAR_LEAF_VARIABLE(name, size) constexpr auto name = ::Ungar::var_c<#name, size>
#define UNGAR_BRANCH_VARIABLE(name) constexpr auto name = ::Ungar::var_c<#name>
#define UNGAR_VARIABLE(name, ...) constexpr auto name = ::Ungar::var_c<#name __VA_OPT__(, __VA_ARGS__)>

}  // namespace Ungar
