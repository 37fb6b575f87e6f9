// ungar_amd :: scalar / vector typedefs of the host API (reference include/ungar/data_types.hpp:89-330).
#pragma once

#include <string>
#include <vector>
#include <string_view>

#include "assert.hpp"
#include "io/logging.hpp"
#include "linalg.hpp"
#include "variable.hpp"

namespace Ungar {

using namespace std::literals;

template <class S, index_t N>
using Vector = Eigen::Vector<S, N>;
template <class S>
using VectorX = Eigen::VectorX<S>;
template <class S>
using Vector2 = Eigen::Vector<S, 2>;
template <class S>
using Vector3 = Eigen::Vector<S, 3>;
template <class S>
using Vector4 = Eigen::Vector<S, 4>;
template <class S>
using Quaternion = Eigen::Quaternion<S>;

using VectorXr = VectorX<real_t>;
using Vector2r = Vector2<real_t>;
using Vector3r = Vector3<real_t>;
using Vector4r = Vector4<real_t>;
using Quaternionr = Quaternion<real_t>;
using MapToVectorXr = Eigen::Map<VectorXr>;
using MapToConstVectorXr = Eigen::Map<const VectorXr>;
using MapToVector3r = Eigen::Map<Vector3r>;
using MapToConstVector3r = Eigen::Map<const Vector3r>;
using MapToQuaternionr = Eigen::Map<Quaternionr>;
using MapToConstQuaternionr = Eigen::Map<const Quaternionr>;

template <class S>
using MatrixX = Linalg::DenseMatrix<S>;
using MatrixXr = MatrixX<real_t>;
template <class S>
using SparseMatrix = Linalg::SparseView<S>;  // row-major compressed view (function.hpp:375-383)

namespace Linalg {
/// Row-major dense copy of a compressed row-major sparse matrix (either algebra: the accessors share Eigen's names).
template <class Sparse>
inline std::vector<real_t> ToDense(const Sparse& m) {
    std::vector<real_t> d(static_cast<std::size_t>(m.rows() * m.cols()), 0.0);
    for (index_t r = 0; r < m.rows(); ++r)
        for (int k = m.outerIndexPtr()[r]; k < m.outerIndexPtr()[r + 1]; ++k) d[static_cast<std::size_t>(r * m.cols() + m.innerIndexPtr()[k])] = m.valuePtr()[k];
    return d;
}
}  // namespace Linalg

namespace Concepts {
template <class T>
concept Scalar = std::is_arithmetic_v<T> || requires(T a) { a.IsLiteral(); };
}

}  // namespace Ungar
