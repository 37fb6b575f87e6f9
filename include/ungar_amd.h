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

/* ABI version of THIS header.  It changes whenever a struct below grows or an entry point's parameter list changes (4: ungar_ocp_qp with its
 * equality-row fields, `status` in ungar_ocp_line_search_select / _accept; 5: `instances` in ungar_shooting_merit_args, the entry points
 * ungar_shooting_trial_rows_listed / ungar_shooting_select_listed of the staged line search; 6: the entry points the C++ driver of round 5 calls --
 * ungar_shooting_trial_elements, ungar_function_{forward_zero,sparse_jacobian,sparse_hessian}_nodes_split, ungar_ocp_riccati_route, ungar_shooting_assemble_route,
 * ungar_measurement_build -- so that a driver header never meets a library without them; 7: the wave-tile layout of the dense block,
 * ungar_model_tile_layout / ungar_model_tile_doubles / ungar_model_dense_jacobian_tiles / ungar_tiles_gather; ungar_function_get_tape, `value_stride` in
 * ungar_shooting_merit_args).  ungar_abi_version() returns the version the LIBRARY was built with:
 * a wrapper compiled against an older header must compare the two before its first call -- a mismatch silently shifts arguments otherwise.
 * (ungar_amd/__init__.py and Ungar::BatchedSoftSQPOptimizer do.) */
#define UNGAR_AMD_ABI_VERSION 7
int32_t ungar_abi_version(void);

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

/* Opens one of the built-in node models: "quadrotor_cost" / "srbd_cost" / "rc_car_cost" (scalar stage costs of the quadrotor,
 * quadruped and RC-car OCPs: value, gradient, upper Hessian), "anymal_cost" (tracking cost of the full-body quadruped, x = [q(19); v(18)],
 * u = 12 torques, p = [x_ref(37), 5 weights]: the engine's own -- the reference has no full-body OCP), "rc_car_ineq" (3 inequality rows per knot of the RC-car OCP),
 * "srbd_feet" (world foot positions of the quadruped: the node-local part of its foot-contact equality rows), "srbd_ineq" (12 inequality rows per knot of the quadruped OCP and their Jacobian),
 * "quadrotor_ineq" (8 rotor-speed bound rows per knot of the quadrotor OCP),
 * "quadrotor", "rc_car", "srbd", "anymal" (structured
 * implicit differentiation, phased body with an LDS home), "anymal_reg" (same program, plain
 * straight-line body) or "anymal_ad" (same function, derivatives by taping ABA); and the rigid-body quantities of
 * ANYmal B as batched node models y = f(x, u) (rbd/quantities/<name>.hpp:42-43 in the reference): "anymal_rnea"
 * (x = [q; v], u = a, y = joint torques), "anymal_crba" (x = q, y = M(q) 18 x 18 row-major), "anymal_minv" (x = q,
 * y = M(q)^-1, value only), "anymal_feet" (x = q, y = 4 x [position(3); rotation(9)] of LF/LH/RF/RH_FOOT),
 * "anymal_centroidal" (x = [q; v], y = centroidal momentum [linear; angular]).
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

/* Scalar node models (one output, e.g. the stage cost "quadrotor_cost"): y, its gradient w.r.t. (x, u) into
 * batch->jac when that operand is given (dense 1 x (nx+nu) row), and the UPPER triangle of the Hessian w.r.t.
 * (x, u) into `hes`, hes_nnz values per node in the CSR order of ungar_model_hessian_sparsity.
 * replaces GenericModel::SparseHessian (function.hpp:252-257).  UNGAR_E_UNSUPPORTED for vector-valued models. */
int ungar_model_sparse_hessian(const ungar_model* model, const ungar_node_batch* batch, const ungar_operand* hes, void* stream);

/* Gauss-Newton contraction  G = J^T diag(d) J  per node on the FP64 matrix cores
 * (v_mfma_f64_16x16x4_f64).  replaces the Eigen sparse triple product at
 * include/ungar/optimization/soft_sqp.hpp:257-264.
 *   jac : node-major dense blocks, rows x cols row-major, leading dimension ld_j, node stride js
 *   d   : rows weights per node (node stride ds) or null for the identity
 *   g   : cols x cols row-major FULL symmetric block per node, leading dimension ld_g, node stride gs */
int ungar_gn_hessian(const double* jac, int64_t js, int64_t ld_j, const double* d, int64_t ds, double* g, int64_t gs, int64_t ld_g,
                     int32_t rows, int32_t cols, int64_t count, void* stream);

/* Same contraction, UPPER triangle only: entries with row <= col are written, the rest of `g` is left
 * untouched.  The reference keeps Hessians upper-triangular (function.hpp:236-274) and its QP reads only
 * the upper triangle of the objective matrix (soft_sqp.hpp:143-152), so this is what the SQP consumer
 * needs: 10 of 16 tile products, half the bytes written. */
int ungar_gn_hessian_upper(const double* jac, int64_t js, int64_t ld_j, const double* d, int64_t ds, double* g, int64_t gs, int64_t ld_g,
                           int32_t rows, int32_t cols, int64_t count, void* stream);

/* The same upper-triangular contraction for a Jacobian in the UNIT-FASTEST layout, i.e. exactly what
 * ungar_model_dense_jacobian writes fastest: entry (r, c) of node i at jac[(r * cols + c) * j_es + i]
 * (nodes contiguous), weights at d[r * d_es + i] (or null).  The result is node-major as above.  Removes
 * the transpose between the node kernel and the Gauss-Newton term (soft_sqp.hpp:257-264). */
int ungar_gn_hessian_upper_unit_fastest(const double* jac, int64_t j_es, const double* d, int64_t d_es, double* g, int64_t gs, int64_t ld_g,
                                        int32_t rows, int32_t cols, int64_t count, void* stream);

/* The same upper-triangular contraction for unit-fastest Jacobians with ONE LANE PER NODE on the FP64 vector ALU (which has
 * the matrix instruction's FP64 rate on gfx950 and needs no transposition of this layout): every operand a coalesced load
 * of 64 consecutive nodes.  Output strides are the caller's: G(a, b) of node i at g[(a * ld_g + b) * g_es + i * g_ns] --
 * unit-fastest (g_es >= count, g_ns = 1: coalesced, what ungar_ocp_riccati_solve reads through its operand strides) or
 * node-major blocks (g_es = 1, g_ns = block stride).  Entries with row > col are not written.  Any cols; (soft_sqp.hpp:257-264). */
int ungar_gn_hessian_upper_lanes(const double* jac, int64_t j_es, const double* d, int64_t d_es, double* g, int64_t g_es, int64_t g_ns, int64_t ld_g,
                                 int32_t rows, int32_t cols, int64_t count, void* stream);

/* The same operands and result as ungar_gn_hessian_upper_lanes with ONE LANE PER (node, 7 x 7 block of G): a workgroup owns 16
 * consecutive nodes, streams their Jacobian rows once through LDS (double buffered) and every lane accumulates its block on
 * the FP64 vector ALU -- each Jacobian byte leaves HBM once (the lane-per-node kernel re-reads it 8 times through L1 / L2).
 * Compiled for the block widths of the built-in models (cols = 49, 37, 17); other widths are forwarded to the
 * lane-per-node kernel.  This is the kernel of BASELINE config 4's chain node Jacobians -> Gauss-Newton term.
 * (soft_sqp.hpp:257-264). */
int ungar_gn_hessian_upper_tiles(const double* jac, int64_t j_es, const double* d, int64_t d_es, double* g, int64_t g_es, int64_t g_ns, int64_t ld_g,
                                 int32_t rows, int32_t cols, int64_t count, void* stream);

/* ---- wave tiles: the dense block stored as register images of the wavefronts that compute it --------------------------------
 * What bounds the widest node kernel ('anymal', 37 x 49) is its result-store path, and the store path wants (a) ONE store instruction =
 * 1 KiB contiguous and (b) the wavefronts that run at the same time filling one contiguous band together (DESIGN.md sections 3 / 4.5;
 * profiles/r06a_store_ceiling_*.log: 6.1 TB/s against 5.0-5.4 for every per-entry or per-wavefront layout).  A tile operand is therefore
 *     [band of `band_tiles` tiles][unit][tile of the band][64 lanes][2 images]
 * tile = `nodes_per_tile` consecutive nodes (node i = instance * knots + knot); a lane = (leg, node of the tile): lane = 16 * (n / 4) + 4 * leg + n % 4;
 * an image = one value per lane; unit p = images 2 p and 2 p + 1 side by side per lane (16 bytes per lane, 1 KiB per unit).
 * Entry (row, col) of the dense block -- row * (nx + nu) + col -- sits in slot 4 * image + leg with entry_of_slot[slot] == entry
 * (-1: a padding slot, 3 of them); every entry appears exactly once.  Element `slot` of node i:
 *     t = i / nodes_per_tile, n = i % nodes_per_tile, image = slot / 4, leg = slot % 4, lane = 16 * (n / 4) + 4 * leg + n % 4
 *     tiles[(((t / band_tiles) * (images / 2) + image / 2) * band_tiles + t % band_tiles) * unit_doubles + 2 * lane + image % 2]
 * The layout belongs to the MODEL (its generated program decides which entries share a store); only 'anymal' has one.
 * replaces, for that model, the dense operand of GenericModel::SparseJacobian's values (function.hpp:224-228). */
typedef struct ungar_tile_layout {
    int32_t nodes_per_tile;        /* 16 */
    int32_t band_tiles;            /* 64 */
    int32_t images;                /* store images per tile (even) */
    int32_t unit_doubles;          /* 128: two images */
    int32_t entries;               /* ny * (nx + nu) */
    int32_t reserved;
    const int16_t* entry_of_slot;  /* HOST table, 4 * images entries; owned by the library */
} ungar_tile_layout;
/* UNGAR_E_UNSUPPORTED for models without a tile program.  Needs no device. */
int ungar_model_tile_layout(const ungar_model* model, ungar_tile_layout* layout);
/* Doubles of a tile operand for `count` nodes (whole bands: the last band is padded); < 0 on error. */
int64_t ungar_model_tile_doubles(const ungar_model* model, int64_t count);
/* Value (batch->f, may be null) and dense Jacobian of every node of the batch into `tiles` (batch->jac is ignored; tile 0 = nodes 0..15 of the
 * batch).  Same inputs, same values as ungar_model_dense_jacobian up to the last bit of a few entries (contraction differences between two
 * compilations of the same program).  replaces GenericModel::SparseJacobian (function.hpp:224-228). */
int ungar_model_dense_jacobian_tiles(const ungar_model* model, const ungar_node_batch* batch, double* tiles, void* stream);
/* jac[b * instance_stride + k * knot_stride + entry * element_stride] = tile slot of that entry, for every node (b, k) of `count` nodes with
 * `knots` knots per instance: a tile operand converted into any strided operand (unit-fastest, node-major, a VariableMap buffer). */
int ungar_tiles_gather(const ungar_model* model, const double* tiles, int64_t count, int64_t knots, const ungar_operand* jac, void* stream);

/* ---- layout conversion ----------------------------------------------------------------------------------- */

/* dst[n * dst_node_stride + e * dst_element_stride] = src[n * src_node_stride + e * src_element_stride]   for n < count, e < elements
 * (strides in doubles), through 64 x 64 LDS tiles: coalesced on both sides when one operand is unit-fastest (node stride 1) and the other
 * instance-major (element stride 1).  The reference hands instance-major data (`Eigen::Map`s over one VariableMap buffer per instance,
 * function.hpp:186-257); the node kernels are fastest on unit-fastest operands -- for wide Jacobians (ANYmal: 5.8x) it pays to transpose
 * the inputs in and the Jacobian out around the unit-fastest launch.  Device pointers; src and dst must not overlap. */
int ungar_transpose_nodes(const double* src, int64_t src_node_stride, int64_t src_element_stride, double* dst, int64_t dst_node_stride, int64_t dst_element_stride,
                          int64_t count, int32_t elements, void* stream);

/* ---- whole-horizon assembly ------------------------------------------ */

/* Sparsity of the equality-constraint Jacobian  d g / d [X | U]  of a horizon-N OCP built on `model`,
 *     g = [x_0 - x_m ; x_{k+1} - f(x_k, u_k)]_{k<N}     (example/mpc/quadrotor.example.cpp:246-266),
 * in canonical CSR: (N+1) nx rows, (N+1) nx + N nu columns, nnz = nx + N (jac_nnz + nx).
 * Pass null arrays to query `nnz` only.  Host arrays; replaces the JacobianSparsity query that the
 * reference makes on the whole-horizon model (function.hpp:98-105). */
int ungar_ocp_equality_sparsity(const ungar_model* model, int64_t horizon, int32_t* row_starts, int32_t* cols, int64_t* nnz);

/* Uploads the model's node pattern (a few KB) to the CURRENT device so that ungar_ocp_assemble_equality allocates
 * nothing and does not synchronise; optional (the first assembly call on a device does it otherwise), thread-safe,
 * once per device.  No reference counterpart: the reference's sparsity lives in host arrays (function.hpp:98-134). */
int ungar_model_prepare(const ungar_model* model);

/* Assembles, for `batch` instances, the constraint values g ((N+1) nx per instance) and the CSR value
 * array of the block-bidiagonal Jacobian from the node kernels' outputs.
 *   x    : states x_k, k = 0..N (instance/knot/element strides; a VariableMap buffer works directly)
 *   xm   : measured state per instance (knot_stride ignored)
 *   f    : node values, jac : node DENSE blocks, both over batch*N nodes as written by
 *          ungar_model_dense_jacobian (node (b, k) at b * instance_stride + k * knot_stride;
 *          the unit-fastest layout of the node kernels is instance_stride = N, knot_stride = 1)
 *   g, values : outputs, instance-major (instance_stride = per-instance size or more).
 * replaces, for this structure, the whole-horizon SparseJacobian evaluation (function.hpp:224-228). */
int ungar_ocp_assemble_equality(const ungar_model* model, int64_t horizon, int64_t batch, const ungar_operand* x, const ungar_operand* xm,
                                const ungar_operand* f, const ungar_operand* jac, const ungar_operand* g, const ungar_operand* values,
                                void* stream);

/* ---- batched SQP iteration for shooting problems ---------------------------------
 * Device counterparts of the pieces of SoftSQPOptimizer::Optimize (include/ungar/optimization/soft_sqp.hpp:62-112) for
 * problems with stage-wise cost / soft inequalities and the dynamics constraint g = [x_0 - x_m; x_{k+1} - f(x_k, u_k)]:
 * thousands of independent MPC instances per launch, one 64-lane workgroup per instance, everything stream-ordered.
 * Node operands use (instance_stride, knot_stride, element_stride); per-instance operands ignore knot_stride. */

/* Relaxed barrier of the soft inequality constraints h <= 0, applied to z = -h
 * (soft_inequality_constraint.hpp:77-205; soft_sqp.hpp:105-130). */
#define UNGAR_BARRIER_POLY 0
#define UNGAR_BARRIER_LOG 1
typedef struct ungar_barrier {
    int32_t type;
    int32_t reserved;
    double stiffness, epsilon;
} ungar_barrier;

/* Stage data of the QP subproblem from the node kernels' outputs (soft_sqp.hpp:143-155, 247-264 per knot):
 *   b_k = f_k - x_{k+1},   dx0 = x_m - x_0,
 *   hess_k = hess cost_k + J_h^T diag(b''(-h)) J_h   (dense (nx+nu)^2 row-major; the upper triangle is written, the strict lower triangle is left untouched),
 *   grad_k = grad cost_k - J_h^T b'(-h). */
typedef struct ungar_ocp_stage_qp_args {
    int64_t nx, nu, horizon, batch;
    ungar_operand X, xm, f;        /* states k = 0..N; measured state; node values k < N (ungar_model_dense_jacobian's f) */
    ungar_operand cost_grad;       /* dense 1 x (nx+nu) per node (jac operand of ungar_model_sparse_hessian) */
    ungar_operand cost_hes;        /* hes_nnz values per node in the order of ungar_model_hessian_sparsity */
    const int32_t* hes_rows;       /* HOST arrays of that pattern (hes_nnz <= 160, rows <= cols < 256) */
    const int32_t* hes_cols;
    int64_t hes_nnz;
    int64_t nh;                    /* inequality rows per knot (<= 64), 0: none */
    ungar_operand h, h_jac;        /* values and dense nh x (nx+nu) Jacobian per node (ungar_model_dense_jacobian of an inequality node) */
    ungar_barrier barrier;
    ungar_operand b, hess, grad, dx0; /* outputs */
} ungar_ocp_stage_qp_args;
int ungar_ocp_stage_qp(const ungar_ocp_stage_qp_args* args, void* stream);

/* Solves  min sum_k 1/2 [dx;du]^T hess_k [dx;du] + grad_k^T [dx;du] + terminal   s.t.  dx_0 = dx0,
 * dx_{k+1} = A_k dx_k + B_k du_k + b_k   for every instance by the discrete Riccati recursion (exact solution of the KKT
 * system the reference hands to OSQP, soft_sqp.hpp:143-158).  `regularization` is added to the diagonal of every stage
 * Hessian and of the terminal one (the reference's 1e-6 I, :149-151).  workspace: ungar_ocp_riccati_workspace() doubles
 * of device memory.  status (device, may be null): 0, or k+1 if the reduced input Hessian of knot k was not positive definite. */
typedef struct ungar_ocp_qp {
    int64_t nx, nu, horizon, batch;
    ungar_operand jac;             /* dense nx x (nx+nu) [A|B] per node */
    ungar_operand b, hess, grad;   /* per node: nx, (nx+nu)^2 (upper triangle read), nx+nu */
    ungar_operand hess_terminal, grad_terminal; /* per instance: nx^2 (upper triangle read), nx; bases may be null */
    ungar_operand dx0;             /* per instance */
    ungar_operand dX, dU;          /* outputs: dx_k (k = 0..N), du_k (k < N) */
    double* workspace;
    int64_t workspace_doubles;
    double regularization;
    int32_t* status;
    /* Stage equality rows  E_k [dx_k; du_k] + e_k = 0  for k < N -- the foot-contact rows of example/mpc/quadruped.example.cpp:279-304 are hard
     * equalities of the reference's QP next to the dynamics (soft_sqp.hpp:155-157).  ne rows per node (0: none), eq = dense row-major
     * ne x (nx+nu) per node, eq_values = ne per node.  A row that is identically zero (an inactive contact) is skipped; the input parts
     * of the other rows of a node must be linearly independent (status k+1 otherwise). */
    int64_t ne;
    ungar_operand eq, eq_values;
    int64_t hess_terminal_ld;      /* leading dimension of hess_terminal (0: nx): the terminal block may be the corner of a wider one */
} ungar_ocp_qp;
int64_t ungar_ocp_riccati_workspace(int64_t nx, int64_t nu, int64_t horizon, int64_t batch);
int ungar_ocp_riccati_solve(const ungar_ocp_qp* qp, void* stream);
/* Which recursion a QP of these stage sizes runs (ne: equality rows INSIDE the recursion; rows eliminated by ungar_shooting_assemble count as 0):
 *   0  the LDS-resident kernels, sizes known at run time (any nx + nu <= 256);
 *   1  the register-resident one-wavefront kernel compiled into the library (the reference's own stage sizes);
 *   2  the same kernel TEMPLATE instantiated for (nx, nu) by the kernel factory: `hipcc --genco` through the model cache's machinery
 *      (UNGAR_CODEGEN_FOLDER/ungar_amd_kernels, keyed by the kernel sources, the sizes, the toolchain), occupancy picked from the compiled candidates.
 * prepare != 0 builds / loads a factory kernel NOW (seconds, once per size and cache folder) instead of inside the first ungar_ocp_riccati_solve; a
 * factory failure (no compiler on the machine) is reported on stderr once and the call returns 0: the solve then takes the LDS-resident kernels.
 * Replaces the reference's "any NLPProblem" contract for the QP step (optimization/concepts.hpp:153-262, soft_sqp.hpp:143-158). */
int ungar_ocp_riccati_route(int64_t nx, int64_t nu, int64_t ne, int32_t prepare);

/* Merit terms of the line search per instance (soft_sqp.hpp:68-87): theta = multiplier * |g|_2, phi = sum of the stage
 * costs (+ terminal) + sum of the barrier over the inequality values; and, when cost_grad and dX are given, the slope
 * grad(objective) . step used by the Armijo branch (backtracking_line_search.hpp:99-101). */
typedef struct ungar_ocp_merit_args {
    int64_t nx, nu, horizon, batch, nh;
    ungar_operand X, xm, f;        /* states, measured state, node values at (X, U) */
    ungar_operand cost;            /* 1 value per node; base may be null */
    ungar_operand cost_terminal;   /* 1 value per instance; base may be null */
    ungar_operand h;               /* nh values per node; base may be null */
    ungar_barrier barrier;
    double violation_multiplier;
    ungar_operand cost_grad, cost_grad_terminal, dX, dU; /* optional: slope */
    double *theta, *phi, *slope;   /* device, one per instance (slope may be null) */
} ungar_ocp_merit_args;
int ungar_ocp_merit(const ungar_ocp_merit_args* args, void* stream);

/* Xt = X + alpha dX,  Ut = U + alpha dU for every instance. */
int ungar_ocp_trial_point(int64_t nx, int64_t nu, int64_t horizon, int64_t batch, const ungar_operand* X, const ungar_operand* U, const ungar_operand* dX,
                          const ungar_operand* dU, double alpha, const ungar_operand* Xt, const ungar_operand* Ut, void* stream);

/* One candidate step size of BacktrackingLineSearch::Do (backtracking_line_search.hpp:116-151) for every instance that has
 * not accepted a larger one: the three-way test on (theta, phi) -> (theta_trial, phi_trial); accepting instances copy the
 * trial point into (X, U) and record alpha in accepted[instance] (0 = still searching).  status (device, may be null): the report of
 * ungar_ocp_riccati_solve; an instance whose QP was not solved (non-zero) takes no step -- the reference asserts there (soft_sqp.hpp:223-230). */
typedef struct ungar_line_search_parameters {
    double alpha_min, theta_min, theta_max, eta, gamma_phi, gamma_theta, gamma_alpha; /* reference defaults: 1e-4 1e-6 1e-2 1e-4 1e-6 1e-6 0.5 */
} ungar_line_search_parameters;
int ungar_ocp_line_search_accept(int64_t nx, int64_t nu, int64_t horizon, int64_t batch, const ungar_line_search_parameters* parameters, double alpha,
                                 const double* theta0, const double* phi0, const double* slope, const double* theta_trial, const double* phi_trial,
                                 double* accepted, const ungar_operand* X, const ungar_operand* U, const ungar_operand* Xt, const ungar_operand* Ut,
                                 const int32_t* status, void* stream);

/* The whole backtracking search in THREE launches instead of three per candidate (the GPU-native form of
 * backtracking_line_search.hpp:116-151: with thousands of instances somebody always needs a short step, so all candidates get
 * evaluated anyway -- as 14 x 6 small launches one after the other, or as a few large ones):
 *   ungar_ocp_trial_points       Xt(c * batch + i) = X(i) + alphas[c] dX(i), likewise Ut: `candidates * batch` STACKED trial points
 *                                (candidate c of instance i is stacked instance c * batch + i; alphas: host array, at most 16);
 *   node values / ungar_ocp_merit_stacked over the candidates * batch stacked instances (args->batch = candidates * batch; operands
 *                                X, f, cost, h, theta, phi indexed by the stacked instance, xm by instance % period, period = batch);
 *   ungar_ocp_line_search_select per instance, the FIRST candidate (largest step) that passes the test is copied into (X, U) and its
 *                                step size recorded in accepted[instance] (0 = none acceptable: instance left unchanged).
 * Same result as the candidate-by-candidate loop over ungar_ocp_trial_point / ungar_ocp_line_search_accept. */
int ungar_ocp_trial_points(int64_t nx, int64_t nu, int64_t horizon, int64_t batch, const ungar_operand* X, const ungar_operand* U, const ungar_operand* dX,
                           const ungar_operand* dU, const double* alphas, int64_t candidates, const ungar_operand* Xt, const ungar_operand* Ut, void* stream);
int ungar_ocp_merit_stacked(const ungar_ocp_merit_args* args, int64_t period, void* stream);
int ungar_ocp_line_search_select(int64_t nx, int64_t nu, int64_t horizon, int64_t batch, const ungar_line_search_parameters* parameters, const double* alphas,
                                 int64_t candidates, const double* theta0, const double* phi0, const double* slope, const double* theta_trial, const double* phi_trial,
                                 double* accepted, const ungar_operand* X, const ungar_operand* U, const ungar_operand* Xt, const ungar_operand* Ut,
                                 const int32_t* status, void* stream);

/* ---- shooting problems with carried quantities and stage equality rows (the reference's three OCPs as written) ------------------
 * The objective of example/mpc/quadrotor.example.cpp:222-227 and rc_car.example.cpp:216-220 couples u_k with u_{k-1}; the foot-contact
 * rows of quadruped.example.cpp:279-304 couple the foot positions of knots k and k-1.  Both are stage-local once the quantity of the
 * previous knot is CARRIED in the stage state: z_k = [c_k; x_k], c_{k+1} = kappa(x_k, u_k) (kappa = u_k, or a recorded function).
 * Every node (instance, knot k <= N) owns one ROW of nv = nc+nx+nu+nw+np doubles [c | x | u | w | p] of a node-major DEVICE array
 * rows[batch][N+1][nv]; stage functions are ungar_functions over a contiguous slice of the row -- dynamics / carry over [x|u ; w|p]
 * (n = nx+nu), cost / equality / inequality rows over [c|x|u ; w|p] (n = nc+nx+nu) -- evaluated with ungar_function_*_nodes into
 * node-major outputs of N+1 knots per instance (values in the order of ungar_function_*_sparsity).  Row N holds x_N; its cost is the
 * terminal cost (input derivatives ignored).  One SQP iteration = SoftSQPOptimizer::Optimize's loop body (soft_sqp.hpp:62-112):
 *   stage derivatives -> ungar_shooting_assemble -> ungar_ocp_riccati_solve (nx := nc+nx, ne rows) -> ungar_shooting_merit ->
 *   ungar_shooting_trial_rows -> stage values on the stacked rows -> ungar_shooting_merit (stacked) -> ungar_shooting_select.
 * include/../ungar_amd/include/ungar/optimization/batched_soft_sqp.hpp drives exactly this from C++20. */
typedef struct ungar_stage_pattern {
    const int32_t* rows; /* DEVICE arrays */
    const int32_t* cols;
    int64_t nnz;
} ungar_stage_pattern;

typedef struct ungar_shooting_dims {
    int64_t nx, nu, nc, nw, np, horizon, batch;
    int32_t carry_inputs; /* 1: c_{k+1} = u_k (nc == nu), no carry function */
    int32_t reserved;
} ungar_shooting_dims;

/* QP data of every node (soft_sqp.hpp:143-155, 247-264): with nz = nc+nx, nd = nz+nu,
 *   AB[batch][N][nz*nd]   rows 0..nc-1 carry Jacobian (identity on u for carry_inputs), rows nc.. dynamics Jacobian, c columns zero
 *   b[batch][N][nz]       [0; f_k - x_{k+1}]          dz0[batch][nz] = [0; x_m - x_0]
 *   W[batch][N+1][nd*nd]  upper triangle of hess cost + J_h^T diag(b''(-h)) J_h + regularization on the (x, u) diagonal (knot N: cost only)
 *   w[batch][N+1][nd]     grad cost - J_h^T b'(-h)
 *   E[batch][N][ne*nd]    dense equality-row Jacobian (values: the equality function's own output). */
typedef struct ungar_shooting_assemble_args {
    ungar_shooting_dims dims;
    const double* rows;
    const double* xm;                       /* batch x nx measured states */
    const double *f, *f_jac;                /* dynamics value (nx) and sparse Jacobian values per node */
    const double* carry_jac;                /* null with carry_inputs */
    const double *cost_grad, *cost_hes;     /* sparse gradient (1 x nd) and upper Hessian values per node */
    const double *h, *h_jac;                /* nh inequality values and sparse Jacobian values per node (null: none) */
    const double* eq_jac;                   /* sparse Jacobian values of the ne equality rows (null: none) */
    ungar_stage_pattern f_pattern, carry_pattern, cost_grad_pattern, cost_hes_pattern, h_pattern, eq_pattern;
    int64_t nh, ne;
    ungar_barrier barrier;
    double regularization;
    double *AB, *b, *W, *w, *E, *dz0;       /* outputs */
    /* eliminate_equalities != 0: the stage equality rows are eliminated here, node by node in parallel, instead of inside the sequential
     * recursion: Gauss-Jordan on [D | C | e] picks one pivot input per active row, u_j = -(E'_i . [z; u] + e'_i); the substitution turns
     * (AB, b, W, w) into an UNCONSTRAINED stage problem in which u_j is a decoupled dummy, so ungar_ocp_riccati_solve is called with
     * ne = 0 and ungar_shooting_recover_inputs restores the eliminated inputs afterwards.  E then receives the reduced rows E', eq_reduced
     * (batch x N x ne) their residuals e', eq_pivots (batch x N x ne) the pivot input of every row (-1: row inactive -- identically zero, or REDUNDANT: reduced
     * by the rows before it to rounding noise, i.e. to less than 1e-12 of its own largest original entry, residual included; the reference's OSQP tolerates
     * such rows too -- and -2: a row that the inputs of its knot cannot meet, e.g. a duplicate row with a different residual -- reported through `status` by
     * the recover call).  eq: the equality function's values (N+1 knots). */
    int32_t eliminate_equalities;
    int32_t reserved;
    const double* eq;
    double* eq_reduced;
    int32_t* eq_pivots;
} ungar_shooting_assemble_args;
int ungar_shooting_assemble(const ungar_shooting_assemble_args* args, void* stream);
/* Which assembly kernel a stage problem takes (nz = carried + state size, ne stage equality rows eliminated by the call, nh inequality rows):
 *   0 the workgroup kernel (any size up to the limits of INTEGRATION.md section 5), 1 the one-wavefront kernel compiled into the library (the reference's
 *   quadruped OCP), 2 the same kernel template instantiated for (nz, nu, ne) by the kernel factory (see ungar_ocp_riccati_route), 3 the run-time-size
 *   one-wavefront kernel of the problems without equality rows.  prepare != 0 builds / loads a factory kernel now. */
int ungar_shooting_assemble_route(int64_t nz, int64_t nu, int64_t ne, int64_t nh, int32_t prepare);
/* After the Riccati solve of a problem assembled with eliminate_equalities: dU[b][k][j] = -(E'_i . [dz_k; du_k] + e'_i) for every reduced
 * row i with pivot input j; status[b] = -(k+1) for a row that cannot be met (status may be null). */
int ungar_shooting_recover_inputs(const ungar_shooting_dims* dims, int64_t ne, const double* E, const double* eq_reduced, const int32_t* eq_pivots, const double* dZ,
                                  double* dU, int32_t* status, void* stream);

/* carry_inputs problems (c_{k+1} = u_k): copies the inputs of row k into the carried slots of row k+1 for every k < N, in place, after `rows` was written from
 * outside (row 0's carried slots stay the caller's: the input applied before the horizon).  quadrotor.example.cpp:222-227 reads u_{k-1} the same way. */
int ungar_shooting_refresh_carried_inputs(const ungar_shooting_dims* dims, double* rows, void* stream);

/* theta = multiplier * |[x_0 - x_m; x_{k+1} - f_k; e_k]|_2, objective = sum_{k<=N} cost_k, phi = objective + barrier terms, and (cost_grad,
 * dZ, dU given) slope = grad objective . step, per instance (soft_sqp.hpp:68-87).  period > 0: dims.batch counts STACKED trial points
 * (candidate c of instance i at c * period + i); xm, dZ, dU are indexed by instance % period. */
typedef struct ungar_shooting_merit_args {
    ungar_shooting_dims dims;
    const double* rows;
    const double* xm;
    const double *f, *cost, *h, *eq;        /* node values, N+1 knots per (stacked) instance; h / eq may be null */
    int64_t nh, ne;
    ungar_barrier barrier;
    double violation_multiplier;
    const double* cost_grad;                /* null: no slope */
    ungar_stage_pattern cost_grad_pattern;
    const double *dZ, *dU;                  /* dZ[batch][N+1][nz], dU[batch][N][nu] */
    double *theta, *phi, *objective, *slope; /* device, one per (stacked) instance; objective / slope may be null */
    int64_t period;
    int64_t rows_stride;                    /* 0: node-major rows; > 0: unit-fastest rows (ungar_shooting_trial_rows with trial_stride) */
    const int32_t* instances;               /* period > 0: null, or the device list of ungar_shooting_trial_rows_listed -- stacked point s belongs to instance
                                             * instances[s % period] (xm, dZ, dU are that instance's) */
    int64_t value_stride;                   /* 0: f, cost, h, eq are dense arrays of nx / 1 / nh / ne values per node; > 0: all four point INTO one array of
                                             * value_stride doubles per node (the output of one function that evaluates them together) */
} ungar_shooting_merit_args;
int ungar_shooting_merit(const ungar_shooting_merit_args* args, void* stream);

/* trial[c * batch + b][k] = rows[b][k] with [c|x|u] += alphas[c] * [dZ; dU] (parameters copied; `alphas` host array, at most 16 candidates; a step of
 * exactly 0 copies the rows whatever dZ / dU hold -- how BatchedSoftSQPOptimizer makes the unit-fastest image of the current rows its derivative kernels read).
 * With carry_inputs the carried slots of row k+1 are the trial inputs of row k; otherwise the caller refreshes them with the carry function
 * (output operand = the carried slots of rows 1..N). */
int ungar_shooting_trial_rows(const ungar_shooting_dims* dims, const double* rows, const double* dZ, const double* dU, const double* alphas, int64_t candidates,
                              double* trial, int64_t trial_stride, void* stream);
/* The same for a LISTED subset of the instances -- the later stages of a staged search evaluate the remaining candidates only for the instances the
 * first ones left unresolved: trial point (c, i), i < listed, is instance instances[i] (device array of instance indices) at c * listed + i.  listed = 0
 * (instances null): all instances, as above.  Merit terms of the listed points: ungar_shooting_merit with dims.batch = candidates * listed,
 * period = listed, instances = the list. */
int ungar_shooting_trial_rows_listed(const ungar_shooting_dims* dims, const double* rows, const double* dZ, const double* dU, const double* alphas, int64_t candidates,
                                     const int32_t* instances, int64_t listed, double* trial, int64_t trial_stride, void* stream);
/* The same for a WINDOW of the row: elements [first_element, first_element + elements) of every stacked row, unit-fastest (trial_stride > 0) at
 * trial[(e - first_element) * trial_stride + i]; elements = 0 means the whole row.  The variables [0, nc + nx + nu) are all that a candidate step changes; the
 * parameter part [nc + nx + nu, row size) of the CURRENT rows (candidates = 1, alphas = {0}) is the same for every candidate and is kept as one image per
 * ungar_function_*_nodes_split call instead of being copied per candidate. */
int ungar_shooting_trial_elements(const ungar_shooting_dims* dims, const double* rows, const double* dZ, const double* dU, const double* alphas, int64_t candidates,
                                  const int32_t* instances, int64_t listed, int64_t first_element, int64_t elements, double* trial, int64_t trial_stride, void* stream);
/* trial_stride: 0 = node-major trial rows (like `rows`); > 0 = UNIT-FASTEST: element e of stacked node i = (c * batch + b) * (N+1) + k at
 * trial[e * trial_stride + i] (trial_stride >= candidates * batch * (N+1)).  The stage functions then read the trial rows with coalesced loads
 * (ungar_operand {base + offset * trial_stride, instance_stride 1, knot_stride 0, element_stride trial_stride}) and touch only the elements they
 * use; pass the same stride to ungar_shooting_merit (rows_stride) and ungar_shooting_select (trial_stride). */

/* Backtracking search over the stacked candidates (backtracking_line_search.hpp:116-151) and the iteration bookkeeping of
 * SoftSQPOptimizer::Optimize (soft_sqp.hpp:88-99) per instance: the first acceptable candidate's [c|x|u] is copied into `rows`, accepted[b] = its
 * step size (0: none); active[b] (may be null) is cleared when no step is acceptable or the objective decreased by less than 1e-6, and instances
 * that are not active are left untouched.  status (may be null): the Riccati solve's per-instance report; an instance whose QP was not solved
 * (non-zero) takes no step and stops -- the reference asserts on a failed QP (soft_sqp.hpp:223-230).
 * stage: 0 = the whole search in one call.  The candidates may also be offered in several calls, largest steps first (the order of
 * backtracking_line_search.hpp:116-151 is kept: an instance takes the first acceptable candidate over all calls): UNGAR_SEARCH_NOT_LAST on every call but
 * the last (an instance without an acceptable candidate is then left for the next call and counted in *unresolved, a device counter the caller
 * zeroes), UNGAR_SEARCH_NOT_FIRST on every call but the first (instances that took a step in an earlier call are skipped). */
#define UNGAR_SEARCH_NOT_FIRST 1
#define UNGAR_SEARCH_NOT_LAST 2
int ungar_shooting_select(const ungar_shooting_dims* dims, const ungar_line_search_parameters* parameters, const double* alphas, int64_t candidates,
                          const double* theta0, const double* phi0, const double* objective0, const double* slope, const double* theta_trial,
                          const double* phi_trial, const double* objective_trial, double* accepted, int32_t* active, const int32_t* status, double* rows, const double* trial,
                          int64_t trial_stride, int32_t stage, int32_t* unresolved, void* stream);
/* The same over a listed subset (instances / listed as in ungar_shooting_trial_rows_listed; the *_trial arrays hold candidates x listed values; theta0 ...
 * accepted, active, status, rows stay indexed by instance).  next_instances (may be null; device array of dims.batch entries, not `instances`): in a call
 * that is not the last, the instances left unresolved are written there at the positions the counter *unresolved hands out -- the list of the next call,
 * *unresolved its length (read it back; the order of the entries is arbitrary, the instances are independent). */
int ungar_shooting_select_listed(const ungar_shooting_dims* dims, const ungar_line_search_parameters* parameters, const double* alphas, int64_t candidates,
                                 const double* theta0, const double* phi0, const double* objective0, const double* slope, const double* theta_trial,
                                 const double* phi_trial, const double* objective_trial, double* accepted, int32_t* active, const int32_t* status, double* rows, const double* trial,
                                 int64_t trial_stride, int32_t stage, int32_t* unresolved, const int32_t* instances, int64_t listed, int32_t* next_instances, void* stream);

/* ---- device memory for host code that is not compiled with hipcc (the C++20 facade) ------------------------------------------------ */
int ungar_device_malloc(void** out, int64_t bytes);
int ungar_device_free(void* ptr);
int ungar_device_upload(void* dst_device, const void* src_host, int64_t bytes);     /* synchronous */
int ungar_device_download(void* dst_host, const void* src_device, int64_t bytes);   /* synchronous (after all work of the null stream) */
int ungar_device_zero(void* dst_device, int64_t bytes, void* stream);
int ungar_device_copy(void* dst_device, const void* src_device, int64_t bytes, void* stream);  /* stream-ordered */
int ungar_device_synchronize(void);
/* A few bytes (<= 64) of device memory as they are after everything `stream` has been given so far: an asynchronous copy into pinned host memory and a
 * completion word the stream writes behind it, which the host polls -- the read-back of a count between two launches of a loop (the unresolved instances of
 * a line-search stage) without a device-wide, interrupt-driven wait (50 us against ~15 around 4 bytes).  Falls back to hipStreamSynchronize where the runtime has
 * no stream memory operations. */
int ungar_device_read_polled(void* dst_host, const void* src_device, int64_t bytes, void* stream);

/* ---- run-time function factory (any recorded function, not only the built-in node models) ----- */

/* One node of a recorded expression tape, in topological order (operands refer to EARLIER nodes).
 * `op` codes (ungar_amd/csrc/tape/graph.hpp): 0 const(value), 1 input(a = index into [x; p]),
 * 2 add, 3 sub, 4 mul, 5 div, 6 neg, 7 sin, 8 cos, 9 tan, 10 asin, 11 acos, 12 atan, 13 exp,
 * 14 log, 15 sqrt, 16 abs, 17 sign, 18 pow(a,b), 19 atan2(a,b), 20..24 CondExp{Lt,Le,Eq,Ge,Gt}:
 * (a cmp b) ? c : d.   replaces the CppAD operation sequence recorded between CppAD::Independent
 * and ADFun(xp, y) (function.hpp:456-466). */
typedef struct ungar_tape_node {
    int32_t op, a, b, c, d;
    int32_t reserved;
    double value;
} ungar_tape_node;

typedef struct ungar_function ungar_function;

typedef struct ungar_function_info {
    int64_t n;        /* independent (decision) variables   -- IndependentVariableSize() */
    int64_t p;        /* parameters                         -- ParameterSize()            */
    int64_t m;        /* dependent variables                -- DependentVariableSize()    */
    int64_t jac_nnz;  /* nnz of the m x n sparse Jacobian (parameter columns trimmed)  */
    int64_t hes_nnz;  /* nnz of the n x n UPPER-TRIANGULAR Hessian of y[0]              */
    int64_t cache_hit; /* 1 if the code object was reused from the on-disk cache          */
} ungar_function_info;

#define UNGAR_ENABLE_NONE 1u      /* EnabledDerivatives::NONE     (autodiff/data_types.hpp:95-100) */
#define UNGAR_ENABLE_JACOBIAN 2u  /* EnabledDerivatives::JACOBIAN */
#define UNGAR_ENABLE_HESSIAN 4u   /* EnabledDerivatives::HESSIAN  */
#define UNGAR_ENABLE_ALL 6u       /* EnabledDerivatives::ALL      */

/* Derives the enabled sparse derivatives of the recorded y = f([x; p]), lowers value / Jacobian /
 * Hessian to HIP kernels, compiles them with hipcc for gfx950 (or reuses the hashed on-disk code
 * object under <folder>/<name>/ungar_amd/ unless `recompile`), and loads them.
 * replaces Autodiff::MakeFunction / FunctionFactory::Make (function.hpp:589-613).
 * `folder` null or "" -> $UNGAR_CODEGEN_FOLDER, else $TMPDIR/ungar_codegen (data_types.hpp:39-41). */
int ungar_function_make(const ungar_tape_node* nodes, int64_t num_nodes, const int32_t* outputs, int64_t m, int64_t n, int64_t p,
                        const char* name, uint32_t enabled_derivatives, const char* folder, int recompile, ungar_function** out);
void ungar_function_free(ungar_function* fn);
int ungar_function_get_info(const ungar_function* fn, ungar_function_info* info);
const char* ungar_function_code_object(const ungar_function* fn);
/* The tape `fn` was made from, as the caller passed it (owned by `fn`; outputs: info.m node indices; folder: the model-cache folder it was made in, may be null).
 * What a caller needs to make ONE function out of several over shared inputs -- BatchedSoftSQPOptimizer evaluates the values of all stage functions of a
 * shooting problem at the trial points of the line search in one launch this way.  No reference counterpart (CppAD's ADFun keeps its operation sequence too). */
int ungar_function_get_tape(const ungar_function* fn, const ungar_tape_node** nodes, int64_t* num_nodes, const int32_t** outputs, const char** folder);
/* replaces GenericModel::JacobianSparsity / HessianSparsity(0, ...) (function.hpp:98-105, 135-145). */
int ungar_function_jacobian_sparsity(const ungar_function* fn, const int32_t** rows, const int32_t** cols, int64_t* nnz);
int ungar_function_hessian_sparsity(const ungar_function* fn, const int32_t** rows, const int32_t** cols, int64_t* nnz);

/* Batched evaluation over `batch` independent instances (device pointers, stream-ordered).
 * Element e of instance i: base[i * instance_stride + e * element_stride] (knot_stride unused).
 * xp holds n+p elements per instance; outputs m / jac_nnz / hes_nnz values per instance.
 * replaces GenericModel::ForwardZero / SparseJacobian / SparseHessian (function.hpp:186-257). */
int ungar_function_forward_zero(const ungar_function* fn, const ungar_operand* xp, const ungar_operand* y, int64_t batch, void* stream);
int ungar_function_sparse_jacobian(const ungar_function* fn, const ungar_operand* xp, const ungar_operand* jac, int64_t batch, void* stream);
int ungar_function_sparse_hessian(const ungar_function* fn, const ungar_operand* xp, const ungar_operand* hes, int64_t batch, void* stream);

/* The same over shooting NODES: node i = (instance i / knots, knot i % knots), element e at
 * base[instance * instance_stride + knot * knot_stride + e * element_stride] for both operands -- the nodes k < N (or k <= N) of every
 * instance of a batch of horizons in one launch, inputs read from and outputs written into per-instance buffers in place. */
int ungar_function_forward_zero_nodes(const ungar_function* fn, const ungar_operand* xp, const ungar_operand* y, int64_t count, int64_t knots, void* stream);
int ungar_function_sparse_jacobian_nodes(const ungar_function* fn, const ungar_operand* xp, const ungar_operand* jac, int64_t count, int64_t knots, void* stream);
int ungar_function_sparse_hessian_nodes(const ungar_function* fn, const ungar_operand* xp, const ungar_operand* hes, int64_t count, int64_t knots, void* stream);
/* The same with the independent variables [0, n) and the parameters [n, n + p) through SEPARATE operands, both addressed by (instance = i / knots, knot =
 * i % knots).  A parameter operand with instance stride 0 serves every instance from one image: the batched SQP evaluates its stage functions at
 * (candidate step, node) pairs -- instance := candidate, knot := node -- where only the variables differ between candidates, and keeps one unit-fastest
 * image of the node parameters (ungar_shooting_trial_elements) instead of copying them for every candidate. */
int ungar_function_forward_zero_nodes_split(const ungar_function* fn, const ungar_operand* x, const ungar_operand* p, const ungar_operand* y, int64_t count, int64_t knots, void* stream);
/* The same with a parameter operand of `parameter_instances` instances that instance i reads at index i % parameter_instances: candidate steps stacked
 * INSTANCE-wise (instance := candidate x instance, knot := knot < knots) over one image of the instances' parameters -- the carried quantities of the trial
 * points of a line search, which exist for the knots 0 .. N - 1 of N + 1 only, in one launch instead of one per candidate. */
int ungar_function_forward_zero_nodes_periodic(const ungar_function* fn, const ungar_operand* x, const ungar_operand* p, int64_t parameter_instances, const ungar_operand* y,
                                               int64_t count, int64_t knots, void* stream);
int ungar_function_sparse_jacobian_nodes_split(const ungar_function* fn, const ungar_operand* x, const ungar_operand* p, const ungar_operand* jac, int64_t count, int64_t knots,
                                               void* stream);
int ungar_function_sparse_hessian_nodes_split(const ungar_function* fn, const ungar_operand* x, const ungar_operand* p, const ungar_operand* hes, int64_t count, int64_t knots,
                                              void* stream);

/* Single-instance HOST call (what Ungar::Autodiff::Function::operator()/Jacobian/Hessian need -- reference function.hpp:206-259, an in-process C call there):
 * xp_host in, the result in out_host, synchronously.  what: 0 value (m doubles), 1 Jacobian values (jac_nnz), 2 Hessian values (hes_nnz).
 * Node-sized functions (n + p <= 128 inputs, <= 512 results) are served by a RESIDENT kernel: one lane that stays on the device between calls, is handed a
 * call through a word of mapped host memory and answers through another -- no launch per call.  It is launched by the first call and returns by itself once
 * nobody has called for UNGAR_AMD_HOST_CALL_RESIDENT_US microseconds (default 100; 0: never resident) or 50 ms after its launch; while it is there, a
 * device-wide synchronisation waits for at most that long.  Larger functions: one launch per call on a private stream, operands through pinned / mapped
 * host buffers.  Not re-entrant per function (the reference's Function is not either). */
int ungar_function_eval_host(ungar_function* fn, int32_t what, const double* xp_host, double* out_host);
/* 1: single-instance host calls of this derivative (what as above) are served by the resident kernel; 0: one launch per call. */
int32_t ungar_function_host_call_resident(const ungar_function* fn, int32_t what);

/* ---- diagnostics ------------------------------------------------------------------------------ */
const char* ungar_last_error(void);
/* "ungar_amd <version> gfx950 hip <runtime version>" */
const char* ungar_version(void);
/* 0: the shipped library.  1: the measurement build (ungar_amd/lib/measurement/libungar_amd.so, same ABI): A/B routes between kernels, per-phase
 * clocks and experiment knobs selectable through UNGAR_AMD_* / UNGAR_GN_* environment variables (csrc/runtime/measurement.hpp) -- what tools/ and the
 * agreement tests between two routes load.  The shipped library reads none of them. */
int32_t ungar_measurement_build(void);

#ifdef __cplusplus
}
#endif
#endif /* UNGAR_AMD_H_ */
