/* ungar_amd -- C ABI of the MI355X batched derivative-evaluation engine.
 *
 * This is the drop-in boundary for Ungar's hot path.  In the reference the
 * object underneath `Ungar::Autodiff::Function` is a CppADCodeGen `GenericModel<double>` obtained
 * from a dlopen'ed library of generated C (include/ungar/autodiff/function.hpp:364-365); Function
 * reaches it through the virtual calls listed beside each entry point below.  Here the object is a
 * `ungar_model` whose evaluation entry points take DEVICE pointers, a batch of independent
 * shooting nodes, and a HIP stream.  Plain C types only: no torch, no Eigen, no HIP headers
 * (`stream` is a hipStream_t passed as void*).
 *
 * Conventions
 *   - every function returns 0 on success, a negative UNGAR_E_* code otherwise; the message for the
 *     last error on the calling thread is ungar_last_error();
 *   - all floating-point data is FP64 (reference real_t, include/ungar/data_types.hpp:89);
 *   - sparsity is canonical row-major CSR (rows ascending, columns ascending within a row); the
 *     reference's own within-row order is generator-defined (function.hpp:367-374);
 *   - the caller owns every buffer; the library owns compiled code only; no hidden allocation and
 *     no host synchronisation on the batched path (calls are stream-ordered).
 */
#ifndef UNGAR_AMD_H_
#define UNGAR_AMD_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UNGAR_OK 0
#define UNGAR_E_INVALID (-1)   /* bad argument (null pointer, negative count, unknown model) */
#define UNGAR_E_HIP (-2)       /* a HIP runtime call failed; see ungar_last_error() */
#define UNGAR_E_UNSUPPORTED (-3) /* derivative not enabled for this model (function.hpp:219, 238) */
#define UNGAR_E_COMPILE (-4)   /* run-time kernel compilation failed */
#define UNGAR_E_IO (-5)

typedef struct ungar_model ungar_model;

/* Sizes of a shooting-node model  x+ = f(x, u; w, p).
 * replaces GenericModel::Domain()/Range() as used for IndependentVariableSize()/ParameterSize()/
 * DependentVariableSize() (function.hpp:350-361): independent = nx + nu, parameters = nw + np,
 * dependent = ny. */
typedef struct ungar_model_info {
    int64_t nx;       /* state size (differentiated) */
    int64_t nu;       /* input size (differentiated) */
    int64_t nw;       /* per-node parameters (not differentiated) */
    int64_t np;       /* per-instance parameters (not differentiated) */
    int64_t ny;       /* outputs (= nx for the node models) */
    int64_t jac_nnz;  /* structural non-zeros of d y / d (x,u); 0 if Jacobian not enabled */
    int64_t hes_nnz;  /* structural non-zeros of the upper-triangular Hessian of y[0]; 0 if n/a */
} ungar_model_info;

/* Strided view of one operand over a batch of `count` nodes.  Node i = (instance b, knot k) with
 * b = i / knots, k = i % knots; element e of that node lives at
 *     base[b * instance_stride + k * knot_stride + e * element_stride]      (strides in doubles).
 * This one descriptor covers the unit-fastest SoA layout (element_stride = count), flat node-major
 * AoS, and the reference's per-instance VariableMap buffers [X | U | parameters]
 * (example/mpc/quadrotor.example.cpp:103-117) without copies. */
typedef struct ungar_operand {
    double* base;
    int64_t instance_stride;
    int64_t knot_stride;
    int64_t element_stride;
} ungar_operand;

typedef struct ungar_node_batch {
    int64_t count;  /* number of shooting nodes = instances * knots */
    int64_t knots;  /* knots per instance (>= 1) */
    ungar_operand x, u, w, p;  /* inputs  (w may be null when nw == 0; p.knot_stride normally 0) */
    ungar_operand f;           /* output: y, ny elements per node (may be null: skip) */
    ungar_operand jac;         /* output: Jacobian values, jac_nnz (sparse) or ny*(nx+nu) (dense) per node */
} ungar_node_batch;

/* ---- model lifetime ------------------------------------------------------------------------ */

/* Opens one of the built-in node models: "quadrotor", "rc_car", "srbd", "anymal".
 * replaces FunctionFactory::Make -> DynamicLib::model(name)  (function.hpp:497, 589-604). */
int ungar_model_open(const char* name, ungar_model** out);
void ungar_model_close(ungar_model* model);

/* replaces GenericModel::getName (function.hpp:531). */
const char* ungar_model_name(const ungar_model* model);
int ungar_model_get_info(const ungar_model* model, ungar_model_info* info);

/* replaces GenericModel::JacobianSparsity(rows, cols) / JacobianSparsitySet (function.hpp:98-105).
 * Pointers stay valid for the model's lifetime. */
int ungar_model_jacobian_sparsity(const ungar_model* model, const int32_t** rows, const int32_t** cols, int64_t* nnz);
/* replaces GenericModel::HessianSparsity(0, rows, cols) (function.hpp:135-145). */
int ungar_model_hessian_sparsity(const ungar_model* model, const int32_t** rows, const int32_t** cols, int64_t* nnz);

/* replaces GenericModel::is{ForwardZero,SparseJacobian,SparseHessian}Available (function.hpp:340-348). */
int ungar_model_has_forward_zero(const ungar_model* model);
int ungar_model_has_sparse_jacobian(const ungar_model* model);
int ungar_model_has_sparse_hessian(const ungar_model* model);

/* ---- batched evaluation (device pointers, stream-ordered) ----------------------------------- */

/* y = f(x,u;w,p) for every node.  replaces GenericModel::ForwardZero (function.hpp:186-189). */
int ungar_model_forward_zero(const ungar_model* model, const ungar_node_batch* batch, void* stream);

/* y and the structural non-zeros of dy/d(x,u) in CSR value order.
 * replaces GenericModel::SparseJacobian (function.hpp:224-228). */
int ungar_model_sparse_jacobian(const ungar_model* model, const ungar_node_batch* batch, void* stream);

/* y and the DENSE row-major ny x (nx+nu) block [A | B] (structural zeros written as 0.0) -- the
 * per-node block of the reference's block-bidiagonal equality Jacobian. */
int ungar_model_dense_jacobian(const ungar_model* model, const ungar_node_batch* batch, void* stream);

/* Gauss-Newton contraction  G = J^T diag(d) J  per node on the FP64 matrix cores
 * (v_mfma_f64_16x16x4_f64).  replaces the Eigen sparse triple product at
 * include/ungar/optimization/soft_sqp.hpp:257-264.
 *   jac : node-major dense blocks, rows x cols row-major, leading dimension ld_j, node stride js
 *   d   : rows weights per node (node stride ds) or null for the identity
 *   g   : cols x cols row-major FULL symmetric block per node, leading dimension ld_g, node stride gs */
int ungar_gn_hessian(const double* jac, int64_t js, int64_t ld_j, const double* d, int64_t ds, double* g, int64_t gs, int64_t ld_g,
                     int32_t rows, int32_t cols, int64_t count, void* stream);

/* ---- diagnostics ------------------------------------------------------------------------------ */
const char* ungar_last_error(void);
/* "ungar_amd <version> gfx950 hip <runtime version>" */
const char* ungar_version(void);

#ifdef __cplusplus
}
#endif
#endif /* UNGAR_AMD_H_ */
