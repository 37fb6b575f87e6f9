// ungar_amd :: implementation of the C ABI declared in include/ungar_amd.h (compiled by hipcc).
//
// Thin by design: argument checking, the model registry, and launches.  No allocation and no
// synchronisation on the batched entry points.
#include "measurement.hpp"
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../../include/ungar_amd.h"
#include "../kernels/node_kernel.hpp"
#include "../kernels/ocp_assembly.hpp"

using ungar_amd::kernels::NodeLaunch;
using ungar_amd::kernels::OperandView;

#define UNGAR_AMD_DECLARE_MODEL(ns)                                                           \
    extern "C" int ungar_amd_launch_##ns(int mode, const NodeLaunch* a, void* stream);        \
    extern "C" const int* ungar_amd_pattern_##ns(int which, int* nnz);                        \
    extern "C" void ungar_amd_dims_##ns(int* d);

UNGAR_AMD_DECLARE_MODEL(quadrotor)
UNGAR_AMD_DECLARE_MODEL(rc_car)
UNGAR_AMD_DECLARE_MODEL(srbd)
UNGAR_AMD_DECLARE_MODEL(anymal)
UNGAR_AMD_DECLARE_MODEL(anymal_ad)
UNGAR_AMD_DECLARE_MODEL(anymal_reg)
UNGAR_AMD_DECLARE_MODEL(quadrotor_cost)
UNGAR_AMD_DECLARE_MODEL(srbd_cost)
UNGAR_AMD_DECLARE_MODEL(srbd_ineq)
UNGAR_AMD_DECLARE_MODEL(quadrotor_ineq)
UNGAR_AMD_DECLARE_MODEL(rc_car_ineq)
UNGAR_AMD_DECLARE_MODEL(srbd_feet)
UNGAR_AMD_DECLARE_MODEL(rc_car_cost)
UNGAR_AMD_DECLARE_MODEL(anymal_cost)
UNGAR_AMD_DECLARE_MODEL(anymal_rnea)
UNGAR_AMD_DECLARE_MODEL(anymal_crba)
UNGAR_AMD_DECLARE_MODEL(anymal_minv)
UNGAR_AMD_DECLARE_MODEL(anymal_feet)
UNGAR_AMD_DECLARE_MODEL(anymal_centroidal)

extern "C" int ungar_amd_launch_ocp_assemble(const ungar_amd::kernels::OcpAssemblyArgs* a, void* stream);

extern "C" int ungar_amd_launch_gn_hessian(const double* jac, long long js, long long ldj, const double* d, long long ds, double* g,
                                            long long gs, long long ldg, int rows, int cols, long long count, int upperOnly, void* stream);

namespace {
thread_local std::string g_lastError;
}

namespace ungar_amd::runtime {
int Fail(int code, const std::string& msg) {
    g_lastError = msg;
    return code;
}
}  // namespace ungar_amd::runtime

namespace {

using ungar_amd::runtime::Fail;

struct BuiltinEntry {
    const char* name;
    int (*launch)(int, const NodeLaunch*, void*);
    const int* (*pattern)(int, int*);  // which = 0 / 1: Jacobian rows / cols; 2 / 3: Hessian rows / cols (scalar models)
    void (*dims)(int*);
    bool scalar = false;  // one output (a stage cost) with gradient and upper-triangular Hessian
};

const BuiltinEntry kBuiltins[] = {
    {"quadrotor", ungar_amd_launch_quadrotor, ungar_amd_pattern_quadrotor, ungar_amd_dims_quadrotor},
    {"rc_car", ungar_amd_launch_rc_car, ungar_amd_pattern_rc_car, ungar_amd_dims_rc_car},
    {"srbd", ungar_amd_launch_srbd, ungar_amd_pattern_srbd, ungar_amd_dims_srbd},
    {"anymal", ungar_amd_launch_anymal, ungar_amd_pattern_anymal, ungar_amd_dims_anymal},
    {"anymal_ad", ungar_amd_launch_anymal_ad, ungar_amd_pattern_anymal_ad, ungar_amd_dims_anymal_ad},
    {"anymal_reg", ungar_amd_launch_anymal_reg, ungar_amd_pattern_anymal_reg, ungar_amd_dims_anymal_reg},
    {"srbd_ineq", ungar_amd_launch_srbd_ineq, ungar_amd_pattern_srbd_ineq, ungar_amd_dims_srbd_ineq},
    {"quadrotor_ineq", ungar_amd_launch_quadrotor_ineq, ungar_amd_pattern_quadrotor_ineq, ungar_amd_dims_quadrotor_ineq},
    {"anymal_rnea", ungar_amd_launch_anymal_rnea, ungar_amd_pattern_anymal_rnea, ungar_amd_dims_anymal_rnea},
    {"anymal_crba", ungar_amd_launch_anymal_crba, ungar_amd_pattern_anymal_crba, ungar_amd_dims_anymal_crba},
    {"anymal_minv", ungar_amd_launch_anymal_minv, ungar_amd_pattern_anymal_minv, ungar_amd_dims_anymal_minv},
    {"anymal_feet", ungar_amd_launch_anymal_feet, ungar_amd_pattern_anymal_feet, ungar_amd_dims_anymal_feet},
    {"anymal_centroidal", ungar_amd_launch_anymal_centroidal, ungar_amd_pattern_anymal_centroidal, ungar_amd_dims_anymal_centroidal},
    {"rc_car_ineq", ungar_amd_launch_rc_car_ineq, ungar_amd_pattern_rc_car_ineq, ungar_amd_dims_rc_car_ineq},
    {"srbd_feet", ungar_amd_launch_srbd_feet, ungar_amd_pattern_srbd_feet, ungar_amd_dims_srbd_feet},
    {"rc_car_cost", ungar_amd_launch_rc_car_cost, ungar_amd_pattern_rc_car_cost, ungar_amd_dims_rc_car_cost, true},
    {"anymal_cost", ungar_amd_launch_anymal_cost, ungar_amd_pattern_anymal_cost, ungar_amd_dims_anymal_cost, true},
    {"quadrotor_cost", ungar_amd_launch_quadrotor_cost, ungar_amd_pattern_quadrotor_cost, ungar_amd_dims_quadrotor_cost, true},
    {"srbd_cost", ungar_amd_launch_srbd_cost, ungar_amd_pattern_srbd_cost, ungar_amd_dims_srbd_cost, true},
};

OperandView View(const ungar_operand& o) {
    return {o.base, o.instance_stride, o.knot_stride, o.element_stride};
}

}  // namespace

struct ungar_model {
    std::string name;
    ungar_model_info info{};
    std::vector<int32_t> jacRows, jacCols, hesRows, hesCols;
    int (*launch)(int, const NodeLaunch*, void*) = nullptr;
    // device copies of the node pattern (rows, cols, row starts) for the whole-horizon assembly, one per device,
    // uploaded by ungar_model_prepare (or by the first assembly call on that device)
    mutable std::mutex patternMutex;
    mutable std::map<int, int*> devPattern;
    ~ungar_model() {
        for (auto& [dev, ptr] : devPattern)
            if (ptr) (void)hipFree(ptr);
    }
};

namespace {
/// Device pointer of the node pattern on the CURRENT device; uploads it on first use (thread-safe; a failed upload
/// leaves nothing behind).
int DevicePattern(const ungar_model* model, const int** out) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return Fail(UNGAR_E_HIP, std::string("hipGetDevice: ") + hipGetErrorString(e));
    std::lock_guard<std::mutex> lock(model->patternMutex);
    auto it = model->devPattern.find(dev);
    if (it != model->devPattern.end()) {
        *out = it->second;
        return UNGAR_OK;
    }
    const int nx = static_cast<int>(model->info.ny), nn = static_cast<int>(model->info.jac_nnz);
    std::vector<int> host(static_cast<std::size_t>(2 * nn + nx + 1), 0);
    for (int k = 0; k < nn; ++k) {
        host[static_cast<std::size_t>(k)] = model->jacRows[static_cast<std::size_t>(k)];
        host[static_cast<std::size_t>(nn + k)] = model->jacCols[static_cast<std::size_t>(k)];
        ++host[static_cast<std::size_t>(2 * nn + model->jacRows[static_cast<std::size_t>(k)] + 1)];
    }
    for (int r = 0; r < nx; ++r) host[static_cast<std::size_t>(2 * nn + r + 1)] += host[static_cast<std::size_t>(2 * nn + r)];
    int* ptr = nullptr;
    e = hipMalloc(&ptr, host.size() * sizeof(int));
    if (e == hipSuccess) {
        e = hipMemcpy(ptr, host.data(), host.size() * sizeof(int), hipMemcpyHostToDevice);
        if (e != hipSuccess) (void)hipFree(ptr);
    }
    if (e != hipSuccess) return Fail(UNGAR_E_HIP, std::string("node pattern upload: ") + hipGetErrorString(e));
    model->devPattern[dev] = ptr;
    *out = ptr;
    return UNGAR_OK;
}
}  // namespace

namespace {

int Evaluate(const ungar_model* model, const ungar_node_batch* batch, void* stream, int mode, const ungar_operand* hes = nullptr) {
    if (!model || !batch) return Fail(UNGAR_E_INVALID, "null model or batch");
    if (batch->count < 0 || batch->knots < 1) return Fail(UNGAR_E_INVALID, "batch.count must be >= 0 and batch.knots >= 1");
    if (batch->count == 0) return UNGAR_OK;
    if (batch->count % batch->knots != 0) return Fail(UNGAR_E_INVALID, "batch.count must be a multiple of batch.knots");
    if (!batch->x.base || (model->info.nu > 0 && !batch->u.base) || (model->info.np > 0 && !batch->p.base) ||
        (model->info.nw > 0 && !batch->w.base))
        return Fail(UNGAR_E_INVALID, "null input operand for model '" + model->name + "'");
    if (mode == ungar_amd::kernels::kModeValue && !batch->f.base) return Fail(UNGAR_E_INVALID, "forward_zero needs an output operand f");
    if (mode == ungar_amd::kernels::kModeHessian) {
        if (model->info.hes_nnz == 0) return Fail(UNGAR_E_UNSUPPORTED, "model '" + model->name + "' has no Hessian (vector-valued node model)");
        if (!hes || !hes->base) return Fail(UNGAR_E_INVALID, "Hessian evaluation needs an output operand hes");
    } else if (mode != ungar_amd::kernels::kModeValue && !batch->jac.base) {
        return Fail(UNGAR_E_INVALID, "Jacobian evaluation needs an output operand jac");
    }
    if (mode != ungar_amd::kernels::kModeValue && model->info.jac_nnz == 0)
        return Fail(UNGAR_E_UNSUPPORTED, "model '" + model->name + "' was built without a Jacobian");
    NodeLaunch a{batch->count, batch->knots, View(batch->x), View(batch->u), View(batch->w), View(batch->p), View(batch->f), View(batch->jac)};
    if (hes) a.hes = View(*hes);
    const int err = model->launch(mode, &a, stream);
    if (err != 0)
        return Fail(UNGAR_E_HIP, std::string("kernel launch failed for model '") + model->name + "': " + hipGetErrorString(static_cast<hipError_t>(err)));
    return UNGAR_OK;
}

}  // namespace

extern "C" {

int ungar_model_open(const char* name, ungar_model** out) {
    if (!name || !out) return Fail(UNGAR_E_INVALID, "ungar_model_open: null argument");
    for (const BuiltinEntry& e : kBuiltins) {
        if (std::strcmp(e.name, name) != 0) continue;
        auto* m = new ungar_model;
        m->name = name;
        int d[5];
        e.dims(d);
        int nnz = 0;
        const int* rows = e.pattern(0, &nnz);
        const int* cols = e.pattern(1, &nnz);
        m->jacRows.assign(rows, rows + nnz);
        m->jacCols.assign(cols, cols + nnz);
        m->info = {d[0], d[1], d[2], d[3], d[4], nnz, 0};
        if (e.scalar) {
            int hnnz = 0;
            const int* hr = e.pattern(2, &hnnz);
            const int* hc = e.pattern(3, &hnnz);
            m->hesRows.assign(hr, hr + hnnz);
            m->hesCols.assign(hc, hc + hnnz);
            m->info.hes_nnz = hnnz;
        }
        m->launch = e.launch;
        *out = m;
        return UNGAR_OK;
    }
    return Fail(UNGAR_E_INVALID, std::string("unknown model '") + name + "' (built-ins: quadrotor, rc_car, srbd, anymal, anymal_ad, anymal_reg, srbd_ineq, quadrotor_ineq, rc_car_ineq, srbd_feet, quadrotor_cost, srbd_cost, rc_car_cost, anymal_cost, anymal_rnea, anymal_crba, anymal_minv, anymal_feet, anymal_centroidal)");
}

void ungar_model_close(ungar_model* model) {
    delete model;
}

const char* ungar_model_name(const ungar_model* model) {
    return model ? model->name.c_str() : "";
}

int ungar_model_get_info(const ungar_model* model, ungar_model_info* info) {
    if (!model || !info) return Fail(UNGAR_E_INVALID, "ungar_model_get_info: null argument");
    *info = model->info;
    return UNGAR_OK;
}

int ungar_model_jacobian_sparsity(const ungar_model* model, const int32_t** rows, const int32_t** cols, int64_t* nnz) {
    if (!model || !rows || !cols || !nnz) return Fail(UNGAR_E_INVALID, "ungar_model_jacobian_sparsity: null argument");
    *rows = model->jacRows.data();
    *cols = model->jacCols.data();
    *nnz = static_cast<int64_t>(model->jacRows.size());
    return UNGAR_OK;
}

int ungar_model_hessian_sparsity(const ungar_model* model, const int32_t** rows, const int32_t** cols, int64_t* nnz) {
    if (!model || !rows || !cols || !nnz) return Fail(UNGAR_E_INVALID, "ungar_model_hessian_sparsity: null argument");
    if (model->info.hes_nnz == 0) return Fail(UNGAR_E_UNSUPPORTED, "model '" + model->name + "' has no Hessian (vector-valued node model)");
    *rows = model->hesRows.data();
    *cols = model->hesCols.data();
    *nnz = static_cast<int64_t>(model->hesRows.size());
    return UNGAR_OK;
}

int ungar_model_has_forward_zero(const ungar_model* model) {
    return model != nullptr;
}
int ungar_model_has_sparse_jacobian(const ungar_model* model) {
    return model && model->info.jac_nnz > 0;
}
int ungar_model_has_sparse_hessian(const ungar_model* model) {
    return model && model->info.hes_nnz > 0;
}

int ungar_model_forward_zero(const ungar_model* model, const ungar_node_batch* batch, void* stream) {
    return Evaluate(model, batch, stream, ungar_amd::kernels::kModeValue);
}
int ungar_model_sparse_jacobian(const ungar_model* model, const ungar_node_batch* batch, void* stream) {
    return Evaluate(model, batch, stream, ungar_amd::kernels::kModeSparseJacobian);
}
int ungar_model_dense_jacobian(const ungar_model* model, const ungar_node_batch* batch, void* stream) {
    return Evaluate(model, batch, stream, ungar_amd::kernels::kModeDenseJacobian);
}

int ungar_model_sparse_hessian(const ungar_model* model, const ungar_node_batch* batch, const ungar_operand* hes, void* stream) {
    return Evaluate(model, batch, stream, ungar_amd::kernels::kModeHessian, hes);
}

extern "C" int ungar_amd_launch_gn_hessian_upper_soa(const double* jac, long long jes, const double* d, long long des, double* g, long long gs,
                                                      long long ldg, int rows, int cols, long long count, void* stream);
int ungar_gn_hessian_upper_unit_fastest(const double* jac, int64_t j_es, const double* d, int64_t d_es, double* g, int64_t gs, int64_t ld_g,
                                        int32_t rows, int32_t cols, int64_t count, void* stream) {
    if (!jac || !g) return Fail(UNGAR_E_INVALID, "ungar_gn_hessian_upper_unit_fastest: null jac or g");
    if (rows <= 0 || cols <= 0 || count < 0 || ld_g < cols || j_es < count) return Fail(UNGAR_E_INVALID, "ungar_gn_hessian_upper_unit_fastest: bad dimensions");
    if (cols > 64) return Fail(UNGAR_E_UNSUPPORTED, "ungar_gn_hessian_upper_unit_fastest: cols > 64 not supported");
    if (count == 0) return UNGAR_OK;
    const int err = ungar_amd_launch_gn_hessian_upper_soa(jac, j_es, d, d_es, g, gs, ld_g, rows, cols, count, stream);
    if (err != 0) return Fail(UNGAR_E_HIP, std::string("gn_hessian (unit-fastest) launch failed: ") + hipGetErrorString(static_cast<hipError_t>(err)));
    return UNGAR_OK;
}
extern "C" int ungar_amd_launch_gn_hessian_lanes(const double* jac, long long jes, const double* d, long long des, double* g, long long ges, long long gns,
                                                  long long ldg, int rows, int cols, long long count, void* stream);
int ungar_gn_hessian_upper_lanes(const double* jac, int64_t j_es, const double* d, int64_t d_es, double* g, int64_t g_es, int64_t g_ns, int64_t ld_g, int32_t rows,
                                 int32_t cols, int64_t count, void* stream) {
    if (!jac || !g) return Fail(UNGAR_E_INVALID, "ungar_gn_hessian_upper_lanes: null jac or g");
    if (rows <= 0 || cols <= 0 || count < 0 || ld_g < cols || j_es < count || (d && d_es < count)) return Fail(UNGAR_E_INVALID, "ungar_gn_hessian_upper_lanes: bad dimensions");
    if (count == 0) return UNGAR_OK;
    const int err = ungar_amd_launch_gn_hessian_lanes(jac, j_es, d, d_es, g, g_es, g_ns, ld_g, rows, cols, count, stream);
    if (err != 0) return Fail(UNGAR_E_HIP, std::string("gn_hessian (lane per node) launch failed: ") + hipGetErrorString(static_cast<hipError_t>(err)));
    return UNGAR_OK;
}
extern "C" void ungar_amd_anymal_tile_layout(int* images, const short** entryOfSlot);                                            // quad_anymal_tiles.hip
extern "C" int ungar_amd_launch_anymal_tiles(const ungar_amd::kernels::NodeLaunch* a, double* tiles, void* stream);
extern "C" int ungar_amd_launch_anymal_tiles_gather(const double* tiles, const ungar_amd::kernels::OperandView* dst, long long count, long long knots, void* stream);
namespace {
constexpr int kTileNodes = 16, kTileBandTiles = 64, kTileUnitDoubles = 128;  // quad_tile_kernel.hpp
bool HasTiles(const ungar_model* model) {
    return model && model->name == "anymal";
}
}  // namespace
int ungar_model_tile_layout(const ungar_model* model, ungar_tile_layout* layout) {
    if (!model || !layout) return Fail(UNGAR_E_INVALID, "ungar_model_tile_layout: null argument");
    if (!HasTiles(model)) return Fail(UNGAR_E_UNSUPPORTED, "model '" + model->name + "' has no wave-tile program (only 'anymal' has)");
    int images = 0;
    const short* table = nullptr;
    ungar_amd_anymal_tile_layout(&images, &table);
    *layout = {kTileNodes, kTileBandTiles, images, kTileUnitDoubles, static_cast<int32_t>(model->info.ny * (model->info.nx + model->info.nu)), 0, table};
    return UNGAR_OK;
}
int64_t ungar_model_tile_doubles(const ungar_model* model, int64_t count) {
    ungar_tile_layout l{};
    const int rc = ungar_model_tile_layout(model, &l);
    if (rc != UNGAR_OK) return rc;
    if (count < 0) return Fail(UNGAR_E_INVALID, "ungar_model_tile_doubles: negative count");
    const int64_t tiles = (count + l.nodes_per_tile - 1) / l.nodes_per_tile, bands = (tiles + l.band_tiles - 1) / l.band_tiles;
    return bands * l.band_tiles * (l.images / 2) * l.unit_doubles;
}
int ungar_model_dense_jacobian_tiles(const ungar_model* model, const ungar_node_batch* batch, double* tiles, void* stream) {
    if (!model || !batch || !tiles) return Fail(UNGAR_E_INVALID, "ungar_model_dense_jacobian_tiles: null argument");
    if (!HasTiles(model)) return Fail(UNGAR_E_UNSUPPORTED, "model '" + model->name + "' has no wave-tile program (only 'anymal' has)");
    if (batch->count < 0 || batch->knots < 1) return Fail(UNGAR_E_INVALID, "batch.count must be >= 0 and batch.knots >= 1");
    if (batch->count == 0) return UNGAR_OK;
    if (batch->count % batch->knots != 0) return Fail(UNGAR_E_INVALID, "batch.count must be a multiple of batch.knots");
    if (!batch->x.base || !batch->u.base || !batch->p.base) return Fail(UNGAR_E_INVALID, "null input operand for model '" + model->name + "'");
    const NodeLaunch a{batch->count, batch->knots, View(batch->x), View(batch->u), View(batch->w), View(batch->p), View(batch->f), {}};
    const int err = ungar_amd_launch_anymal_tiles(&a, tiles, stream);
    if (err != 0) return Fail(UNGAR_E_HIP, std::string("tile kernel launch failed for model '") + model->name + "': " + hipGetErrorString(static_cast<hipError_t>(err)));
    return UNGAR_OK;
}
int ungar_tiles_gather(const ungar_model* model, const double* tiles, int64_t count, int64_t knots, const ungar_operand* jac, void* stream) {
    if (!model || !tiles || !jac || !jac->base) return Fail(UNGAR_E_INVALID, "ungar_tiles_gather: null argument");
    if (!HasTiles(model)) return Fail(UNGAR_E_UNSUPPORTED, "model '" + model->name + "' has no wave-tile program (only 'anymal' has)");
    if (count < 0 || knots < 1 || count % knots != 0) return Fail(UNGAR_E_INVALID, "ungar_tiles_gather: count must be a non-negative multiple of knots");
    if (count == 0) return UNGAR_OK;
    const ungar_amd::kernels::OperandView dst = View(*jac);
    const int err = ungar_amd_launch_anymal_tiles_gather(tiles, &dst, count, knots, stream);
    if (err != 0) return Fail(UNGAR_E_HIP, std::string("tile gather launch failed: ") + hipGetErrorString(static_cast<hipError_t>(err)));
    return UNGAR_OK;
}
extern "C" int ungar_amd_launch_transpose_nodes(const double* src, long long sns, long long ses, double* dst, long long dns, long long des, long long count, int elements,
                                                 void* stream);
int ungar_transpose_nodes(const double* src, int64_t src_node_stride, int64_t src_element_stride, double* dst, int64_t dst_node_stride, int64_t dst_element_stride,
                          int64_t count, int32_t elements, void* stream) {
    if (!src || !dst) return Fail(UNGAR_E_INVALID, "ungar_transpose_nodes: null src or dst");
    if (count < 0 || elements < 0) return Fail(UNGAR_E_INVALID, "ungar_transpose_nodes: bad dimensions");
    if (count == 0 || elements == 0) return UNGAR_OK;
    const int err = ungar_amd_launch_transpose_nodes(src, src_node_stride, src_element_stride, dst, dst_node_stride, dst_element_stride, count, elements, stream);
    if (err != 0) return Fail(UNGAR_E_HIP, std::string("transpose_nodes launch failed: ") + hipGetErrorString(static_cast<hipError_t>(err)));
    return UNGAR_OK;
}
extern "C" int ungar_amd_gn_hessian_tiles_supported(int cols);
extern "C" int ungar_amd_launch_gn_hessian_tiles(const double* jac, long long jes, const double* d, long long des, double* g, long long ges, long long gns,
                                                  long long ldg, int rows, int cols, long long count, void* stream);
int ungar_gn_hessian_upper_tiles(const double* jac, int64_t j_es, const double* d, int64_t d_es, double* g, int64_t g_es, int64_t g_ns, int64_t ld_g, int32_t rows,
                                 int32_t cols, int64_t count, void* stream) {
    if (!jac || !g) return Fail(UNGAR_E_INVALID, "ungar_gn_hessian_upper_tiles: null jac or g");
    if (rows <= 0 || cols <= 0 || count < 0 || ld_g < cols || j_es < count || (d && d_es < count)) return Fail(UNGAR_E_INVALID, "ungar_gn_hessian_upper_tiles: bad dimensions");
    if (count == 0) return UNGAR_OK;
    // Block shapes without a compiled instance of the (node, block)-per-lane kernel take the lane-per-node kernel: same operands, same result.
    if (!ungar_amd_gn_hessian_tiles_supported(cols)) return ungar_gn_hessian_upper_lanes(jac, j_es, d, d_es, g, g_es, g_ns, ld_g, rows, cols, count, stream);
    const int err = ungar_amd_launch_gn_hessian_tiles(jac, j_es, d, d_es, g, g_es, g_ns, ld_g, rows, cols, count, stream);
    if (err != 0) return Fail(UNGAR_E_HIP, std::string("gn_hessian (lane per block) launch failed: ") + hipGetErrorString(static_cast<hipError_t>(err)));
    return UNGAR_OK;
}
namespace {
int GnHessian(const double* jac, int64_t js, int64_t ld_j, const double* d, int64_t ds, double* g, int64_t gs, int64_t ld_g, int32_t rows, int32_t cols,
              int64_t count, int upperOnly, void* stream);
}
int ungar_gn_hessian(const double* jac, int64_t js, int64_t ld_j, const double* d, int64_t ds, double* g, int64_t gs, int64_t ld_g,
                     int32_t rows, int32_t cols, int64_t count, void* stream) {
    return GnHessian(jac, js, ld_j, d, ds, g, gs, ld_g, rows, cols, count, 0, stream);
}
int ungar_gn_hessian_upper(const double* jac, int64_t js, int64_t ld_j, const double* d, int64_t ds, double* g, int64_t gs, int64_t ld_g,
                           int32_t rows, int32_t cols, int64_t count, void* stream) {
    return GnHessian(jac, js, ld_j, d, ds, g, gs, ld_g, rows, cols, count, 1, stream);
}
namespace {
int GnHessian(const double* jac, int64_t js, int64_t ld_j, const double* d, int64_t ds, double* g, int64_t gs, int64_t ld_g, int32_t rows, int32_t cols,
              int64_t count, int upperOnly, void* stream) {
    if (!jac || !g) return Fail(UNGAR_E_INVALID, "ungar_gn_hessian: null jac or g");
    if (rows <= 0 || cols <= 0 || count < 0 || ld_j < cols || ld_g < cols) return Fail(UNGAR_E_INVALID, "ungar_gn_hessian: bad dimensions");
    if (cols > 64) return Fail(UNGAR_E_UNSUPPORTED, "ungar_gn_hessian: cols > 64 not supported (one wavefront tile set per node)");
    if (count == 0) return UNGAR_OK;
    const int err = ungar_amd_launch_gn_hessian(jac, js, ld_j, d, ds, g, gs, ld_g, rows, cols, count, upperOnly, stream);
    if (err != 0) return Fail(UNGAR_E_HIP, std::string("gn_hessian launch failed: ") + hipGetErrorString(static_cast<hipError_t>(err)));
    return UNGAR_OK;
}
}  // namespace

int ungar_model_prepare(const ungar_model* model) {
    if (!model) return Fail(UNGAR_E_INVALID, "ungar_model_prepare: null model");
    if (model->info.ny != model->info.nx) return UNGAR_OK;  // only dynamics nodes take part in the whole-horizon assembly
    const int* pattern = nullptr;
    return DevicePattern(model, &pattern);
}

int ungar_ocp_equality_sparsity(const ungar_model* model, int64_t horizon, int32_t* row_starts, int32_t* cols, int64_t* nnz) {
    if (!model || !nnz || horizon < 1) return Fail(UNGAR_E_INVALID, "ungar_ocp_equality_sparsity: bad argument");
    if (model->info.ny != model->info.nx) return Fail(UNGAR_E_UNSUPPORTED, "whole-horizon assembly needs a dynamics node model (ny == nx)");
    const int64_t nx = model->info.nx, nu = model->info.nu, nn = model->info.jac_nnz;
    *nnz = nx + horizon * (nn + nx);
    if (!row_starts || !cols) return UNGAR_OK;
    int64_t e = 0;
    for (int64_t r = 0; r < nx; ++r) {
        row_starts[r] = static_cast<int32_t>(e);
        cols[e++] = static_cast<int32_t>(r);
    }
    for (int64_t k = 0; k < horizon; ++k) {
        std::size_t p = 0;
        for (int64_t r = 0; r < nx; ++r) {
            row_starts[nx + k * nx + r] = static_cast<int32_t>(e);
            for (; p < model->jacRows.size() && model->jacRows[p] == r && model->jacCols[p] < nx; ++p)
                cols[e++] = static_cast<int32_t>(k * nx + model->jacCols[p]);
            cols[e++] = static_cast<int32_t>((k + 1) * nx + r);
            for (; p < model->jacRows.size() && model->jacRows[p] == r; ++p)
                cols[e++] = static_cast<int32_t>((horizon + 1) * nx + k * nu + (model->jacCols[p] - nx));
        }
    }
    row_starts[(horizon + 1) * nx] = static_cast<int32_t>(e);
    return UNGAR_OK;
}

int ungar_ocp_assemble_equality(const ungar_model* model, int64_t horizon, int64_t batch, const ungar_operand* x, const ungar_operand* xm,
                                const ungar_operand* f, const ungar_operand* jac, const ungar_operand* g, const ungar_operand* values,
                                void* stream) {
    if (!model || !x || !xm || !f || !jac || !g || !values || horizon < 1 || batch < 0)
        return Fail(UNGAR_E_INVALID, "ungar_ocp_assemble_equality: bad argument");
    if (batch == 0) return UNGAR_OK;
    if (!x->base || !xm->base || !f->base || !jac->base || !g->base || !values->base)
        return Fail(UNGAR_E_INVALID, "ungar_ocp_assemble_equality: null operand base");
    if (model->info.ny != model->info.nx) return Fail(UNGAR_E_UNSUPPORTED, "whole-horizon assembly needs a dynamics node model (ny == nx)");
    const int nx = static_cast<int>(model->info.nx), nn = static_cast<int>(model->info.jac_nnz);
    const int* pattern = nullptr;  // allocation-free after ungar_model_prepare (or after the first call on this device)
    if (const int rc = DevicePattern(model, &pattern); rc != UNGAR_OK) return rc;
    ungar_amd::kernels::OcpAssemblyArgs a{x->base, x->instance_stride, x->knot_stride, x->element_stride,
                                          xm->base, xm->instance_stride, xm->element_stride,
                                          f->base, f->instance_stride, f->knot_stride, f->element_stride,
                                          jac->base, jac->instance_stride, jac->knot_stride, jac->element_stride,
                                          g->base, g->instance_stride,
                                          values->base, values->instance_stride,
                                          pattern, pattern + nn, pattern + 2 * nn,
                                          nx, static_cast<int>(model->info.nu), static_cast<int>(horizon), nn, batch};
    const int err = ungar_amd_launch_ocp_assemble(&a, stream);
    if (err != 0) return Fail(UNGAR_E_HIP, std::string("ocp assembly launch failed: ") + hipGetErrorString(static_cast<hipError_t>(err)));
    return UNGAR_OK;
}

const char* ungar_last_error(void) {
    return g_lastError.c_str();
}

int32_t ungar_abi_version(void) {
    return UNGAR_AMD_ABI_VERSION;
}

int32_t ungar_measurement_build(void) {
    return UNGAR_AMD_MEASUREMENT_BUILD;
}

const char* ungar_version(void) {
    static std::string v = [] {
        int rt = 0;
        (void)hipRuntimeGetVersion(&rt);
        return std::string("ungar_amd 0.1 gfx950 hip ") + std::to_string(rt);
    }();
    return v.c_str();
}

}  // extern "C"
