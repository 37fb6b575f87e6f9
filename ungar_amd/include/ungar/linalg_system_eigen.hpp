// ungar_amd :: the host facade on top of the REAL Eigen (selected by UNGAR_AMD_USE_SYSTEM_EIGEN, see linalg.hpp).
//
// Nothing is added to `namespace Eigen` here; the hooks Eigen needs for the recorded scalar (NumTraits, the AD-safe
// normalisation / inverse / slerp of the reference's autodiff/support/quaternion.hpp) live in autodiff/data_types.hpp next
// to the scalar's definition.
#pragma once

#include <Eigen/Core>
#include <Eigen/Geometry>
#include <Eigen/SparseCore>

namespace Ungar::Linalg {
template <class S>
using DenseMatrix = Eigen::Matrix<S, Eigen::Dynamic, Eigen::Dynamic>;
template <class S>
using SparseView = Eigen::Map<const Eigen::SparseMatrix<S, Eigen::RowMajor>>;  // the reference's return type (function.hpp:217, 237)
template <class S>
inline SparseView<S> MakeSparseView(std::ptrdiff_t rows, std::ptrdiff_t cols, std::ptrdiff_t nnz, const int* starts, const int* indices, const S* values) {
    return SparseView<S>{rows, cols, nnz, starts, indices, values};
}
}  // namespace Ungar::Linalg
