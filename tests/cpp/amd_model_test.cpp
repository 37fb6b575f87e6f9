// The reference-side adapter of INTEGRATION.md section 2 (include/ungar_amd_model.hpp), compiled and exercised: a tape built BY
// HAND (what a walker over a CppAD operation sequence would emit) for the reference's known-answer function
//     y = [p |x|^2, 2 x0^2],  x in R^4, p in R            (test/autodiff/function.test.cpp:70-89)
// and, as a second model, its scalar first component with the Hessian enabled (:120-131: Hessian = 2 p I).
// `sparsity` mode needs no GPU (UNGAR_AMD_COMPILE_ONLY=1); `gpu` mode evaluates through ungar_function_eval_host.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/ungar_amd_model.hpp"

using namespace ungar_amd;

static int g_failures = 0;
#define EXPECT_TRUE(cond)                                               \
    do {                                                                \
        if (!(cond)) {                                                  \
            ++g_failures;                                               \
            std::printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #cond); \
        }                                                               \
    } while (0)

int main(int argc, char** argv) {
    const bool gpu = argc > 1 && std::strcmp(argv[1], "gpu") == 0;
    const std::string folder = argc > 2 ? argv[2] : "/tmp/ungar_amd_model_test";
    try {
        TapeBuilder t;
        int32_t x[4], sq = -1;
        for (int i = 0; i < 4; ++i) x[i] = t.Input(i);
        const int32_t p = t.Input(4);
        for (int i = 0; i < 4; ++i) {
            const int32_t xi2 = t.Binary(kMul, x[i], x[i]);
            sq = i == 0 ? xi2 : t.Binary(kAdd, sq, xi2);
        }
        const int32_t y0 = t.Binary(kMul, p, sq);
        const int32_t y1 = t.Binary(kMul, t.Constant(2.0), t.Binary(kMul, x[0], x[0]));
        AmdModel jac(t.Nodes(), {y0, y1}, 4, 1, "amd_model_test_jacobian", UNGAR_ENABLE_JACOBIAN, folder, true);
        AmdModel hes(t.Nodes(), {y0}, 4, 1, "amd_model_test_hessian", UNGAR_ENABLE_ALL, folder, true);
        const int32_t *rows = nullptr, *cols = nullptr;
        int64_t nnz = 0;
        jac.JacobianSparsity(&rows, &cols, &nnz);
        EXPECT_TRUE(nnz == 5 && jac.Info().n == 4 && jac.Info().p == 1 && jac.Info().m == 2);  // [[x x x x], [x . . .]], parameter column trimmed
        const int wantR[5] = {0, 0, 0, 0, 1}, wantC[5] = {0, 1, 2, 3, 0};
        for (int k = 0; k < 5 && k < nnz; ++k) EXPECT_TRUE(rows[k] == wantR[k] && cols[k] == wantC[k]);
        hes.HessianSparsity(&rows, &cols, &nnz);
        EXPECT_TRUE(nnz == 4);
        for (int k = 0; k < 4 && k < nnz; ++k) EXPECT_TRUE(rows[k] == k && cols[k] == k);  // upper triangle of 2 p I
        if (gpu) {
            const double xp[5] = {0.3, -1.2, 0.7, 2.0, 1.5};
            double y[2], jv[5], hv[4];
            jac.ForwardZero(xp, y);
            jac.SparseJacobian(xp, jv);
            hes.SparseHessian(xp, hv);
            const double n2 = 0.3 * 0.3 + 1.2 * 1.2 + 0.7 * 0.7 + 2.0 * 2.0;
            EXPECT_TRUE(std::fabs(y[0] - 1.5 * n2) < 1e-14 && std::fabs(y[1] - 2.0 * 0.09) < 1e-15);
            for (int k = 0; k < 4; ++k) EXPECT_TRUE(std::fabs(jv[k] - 2.0 * 1.5 * xp[k]) < 1e-14);
            EXPECT_TRUE(std::fabs(jv[4] - 4.0 * 0.3) < 1e-15);
            for (int k = 0; k < 4; ++k) EXPECT_TRUE(std::fabs(hv[k] - 2.0 * 1.5) < 1e-15);
        }
    } catch (const std::exception& e) {
        std::printf("EXCEPTION %s\n", e.what());
        return 2;
    }
    if (g_failures) std::printf("FAILED %d checks\n", g_failures);
    else std::printf("amd_model_test OK (%s)\n", gpu ? "gpu" : "sparsity");
    return g_failures ? 1 : 0;
}
