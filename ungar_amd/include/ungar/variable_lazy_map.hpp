// ungar_amd :: typed views over a flat scalar buffer laid out by a Variable hierarchy.
//
// Reference: include/ungar/variable_lazy_map.hpp:87-307 (Get/GetTuple/GetImpl) and
// include/ungar/variable_map.hpp:41-221.  View rule: size 1 -> reference to the
// scalar; unit quaternion -> Map<Quaternion> over (x,y,z,w); size <= 32 -> fixed-size Map; larger ->
// dynamic-size Map.  No copies; works for Scalar in {real_t, ad_scalar_t}.
#pragma once

#include <algorithm>
#include <map>
#include <memory>
#include <tuple>

#include "data_types.hpp"

namespace Ungar {

template <class Scalar, Concepts::Variable Var, bool MUTABLE>
class VariableLazyMap {
  public:
    using Ptr = std::conditional_t<MUTABLE, Scalar*, const Scalar*>;
    constexpr VariableLazyMap(Ptr data, const Var& var) : _data{data}, _var{var} {
    }
    /// Over a vector-like buffer (`VariableLazyMap{map.Get(), variables}`, reference variable_lazy_map.hpp:87-100; see the deduction guide below).
    template <class Underlying>
        requires requires(Underlying& u) { u.data(); u.size(); }
    constexpr VariableLazyMap(Underlying& underlying, const Var& var) : _data{underlying.data()}, _var{var} {
        assert(static_cast<index_t>(underlying.size()) == Var::Size());
    }

    /// View of the sub-variable addressed by `args` (same grammar as Variable::operator()).
    template <class... Args>
    decltype(auto) Get(const Args&... args) const {
        return View(_var(args...));
    }
    template <class... Vars>
    auto GetTuple(const Vars&... vars) const {
        return std::tuple<decltype(Get(vars))...>{Get(vars)...};
    }
    Ptr Data() const {
        return _data;
    }
    /// View of an already resolved sub-variable, e.g. `variables(srbd_state, k, i)` (reference variable_lazy_map.hpp:251-275).
    template <Concepts::Variable V>
    decltype(auto) Get1(const V& v) const {
        return View(v);
    }

    /// View of an already resolved sub-variable (a result of Variable::operator()).
    template <class V>
    decltype(auto) View(const V& v) const {
        Ptr p = _data + (v.Index() - _var.Index());
        if constexpr (V::IsScalar()) {
            return (*p);  // Scalar& / const Scalar&
        } else if constexpr (V::IsQuaternion()) {
            return Eigen::Map<std::conditional_t<MUTABLE, Quaternion<Scalar>, const Quaternion<Scalar>>>{p};
        } else if constexpr (V::Size() <= 32) {
            return Eigen::Map<std::conditional_t<MUTABLE, Vector<Scalar, V::Size()>, const Vector<Scalar, V::Size()>>>{p};
        } else {
            return Eigen::Map<std::conditional_t<MUTABLE, VectorX<Scalar>, const VectorX<Scalar>>>{p, V::Size()};
        }
    }

  private:
    Ptr _data;
    Var _var;
};

/// Writable iff the buffer is not const and does not view const data (a Map<const ...>): decided by what data() returns.
template <class Underlying, Concepts::Variable Var>
VariableLazyMap(Underlying&, const Var&)
    -> VariableLazyMap<std::remove_const_t<typename std::remove_cvref_t<Underlying>::Scalar>, Var,
                       !std::is_const_v<Underlying> && !std::is_const_v<std::remove_pointer_t<decltype(std::declval<Underlying&>().data())>>>;

/// reference variable_lazy_map.hpp:341-361
template <class Underlying, Concepts::Variable Var>
auto MakeVariableLazyMap(Underlying& underlying, const Var& var) {
    using S = std::remove_const_t<typename std::remove_cvref_t<Underlying>::Scalar>;
    // writable iff the buffer is not const and does not view const data (a Map<const ...>): decided by what data() returns
    constexpr bool mut = !std::is_const_v<Underlying> && !std::is_const_v<std::remove_pointer_t<decltype(underlying.data())>>;
    assert(underlying.size() == Var::Size());
    return VariableLazyMap<S, Var, mut>{underlying.data(), var};
}
/// m-variable spelling (reference mvariable_lazy_map.hpp:339-359).
template <class Underlying, Concepts::Variable Var>
auto MakeMVariableLazyMap(Underlying& underlying, const Var& var) {
    return MakeVariableLazyMap(underlying, var);
}

/// Owning map (reference variable_map.hpp:41-221).  Like the reference's, Get hands out REFERENCES to view
/// objects owned by the map (`Eigen::Map<Vector3r>&`, `real_t&`, ...), so a view can be bound once and
/// reused; the reference builds its table of views eagerly at construction, here a view is created the
/// first time it is asked for and then kept (same observable types and lifetimes).
template <class Scalar, Concepts::Variable Var>
class VariableMap {
  public:
    explicit VariableMap(const Var& var) : _underlying(Var::Size()), _var{var} {
        _underlying.setZero();
    }
    VariableMap(const VariableMap& o) : _underlying{o._underlying}, _var{o._var} {  // views are per-buffer: not copied
    }
    VariableMap(VariableMap&&) = default;  // the heap buffer moves along, cached views stay valid
    VariableMap& operator=(const VariableMap& o) {
        if (this != &o) {
            if (_underlying.size() == o._underlying.size()) std::copy(o._underlying.data(), o._underlying.data() + o._underlying.size(), _underlying.data());
            else {
                _views.clear();
                _underlying = o._underlying;
            }
        }
        return *this;
    }
    VariableMap& operator=(VariableMap&&) = default;

    template <class... Args>
    decltype(auto) Get(const Args&... args) {
        if constexpr (sizeof...(Args) == 0) return (_underlying);
        else return Cached<true>(_var(args...));
    }
    template <class... Args>
    decltype(auto) Get(const Args&... args) const {
        if constexpr (sizeof...(Args) == 0) return (_underlying);
        else return Cached<false>(_var(args...));
    }
    template <class... Vars>
    auto GetTuple(const Vars&... vars) {
        return std::tuple<decltype(Get(vars))...>{Get(vars)...};
    }
    template <class... Vars>
    auto GetTuple(const Vars&... vars) const {
        return std::tuple<decltype(Get(vars))...>{Get(vars)...};
    }
    /// View of an already resolved sub-variable (reference variable_map.hpp:166-200).
    template <Concepts::Variable V>
    decltype(auto) Get1(const V& v) {
        return Cached<true>(v);
    }
    template <Concepts::Variable V>
    decltype(auto) Get1(const V& v) const {
        return Cached<false>(v);
    }

  private:
    struct Erased {
        virtual ~Erased() = default;
    };
    template <class M>
    struct Holder : Erased {
        explicit Holder(M m) : view{std::move(m)} {
        }
        M view;
    };
    template <bool MUTABLE, class V>
    decltype(auto) Cached(const V& v) const {
        using P = std::conditional_t<MUTABLE, Scalar*, const Scalar*>;
        P p = const_cast<P>(_underlying.data()) + (v.Index() - _var.Index());
        if constexpr (V::IsScalar()) {
            return (*p);
        } else {
            using M = std::remove_cvref_t<decltype(VariableLazyMap<Scalar, Var, MUTABLE>{p, _var}.View(v))>;
            // key: offset, size, mutability, quaternion-ness (a 4-vector and a quaternion may share an offset)
            const std::tuple<index_t, index_t, bool, bool> key{v.Index(), V::Size(), MUTABLE, V::IsQuaternion()};
            auto it = _views.find(key);
            if (it == _views.end()) {
                VariableLazyMap<Scalar, Var, MUTABLE> lazy{const_cast<P>(_underlying.data()), _var};
                it = _views.emplace(key, std::make_unique<Holder<M>>(lazy.View(v))).first;
            }
            if constexpr (MUTABLE) return (static_cast<Holder<M>&>(*it->second).view);
            else return (static_cast<const Holder<M>&>(*it->second).view);
        }
    }

    VectorX<Scalar> _underlying;
    Var _var;
    mutable std::map<std::tuple<index_t, index_t, bool, bool>, std::unique_ptr<Erased>> _views;
};

template <class Scalar, Concepts::Variable Var>
auto MakeVariableMap(const Var& var) {
    return VariableMap<Scalar, Var>{var};
}

}  // namespace Ungar
