/* The drop-in boundary is a C ABI: this file is compiled as C99 with -pedantic -Werror (tests/test_abi.py) to prove that
 * include/ungar_amd.h carries no C++ in its declarations, and exercises the host-only entry points. */
#include "ungar_amd.h"
#include <stdio.h>
#include <string.h>
int main(void) {
    ungar_model* m = 0;
    int rc = ungar_model_open("quadrotor_cost", &m);
    ungar_model_info info;
    if (rc == UNGAR_OK && ungar_model_get_info(m, &info) == UNGAR_OK) printf("%s nx=%lld ny=%lld hes_nnz=%lld version=%s\n", ungar_model_name(m), (long long)info.nx, (long long)info.ny, (long long)info.hes_nnz, ungar_version());
    {   /* wave tiles (ABI 7): a model without a tile program says so; the layout of 'anymal' is a host-side query */
        ungar_tile_layout layout;
        ungar_model* q = 0;
        double dummy[2] = {0, 0};
        ungar_node_batch nb;
        memset(&nb, 0, sizeof nb);
        if (ungar_model_tile_layout(m, &layout) != UNGAR_E_UNSUPPORTED || ungar_model_tile_doubles(m, 16) >= 0) return 40;
        if (ungar_model_open("anymal", &q) != UNGAR_OK) return 41;
        if (ungar_model_tile_layout(q, &layout) != UNGAR_OK || layout.nodes_per_tile != 16 || layout.band_tiles != 64 || layout.images % 2 != 0 || layout.entries != 37 * 49 ||
            !layout.entry_of_slot || ungar_model_tile_doubles(q, 17) != (int64_t)64 * (layout.images / 2) * layout.unit_doubles)
            return 42;
        if (ungar_model_dense_jacobian_tiles(q, &nb, 0, 0) != UNGAR_E_INVALID || ungar_model_dense_jacobian_tiles(q, 0, dummy, 0) != UNGAR_E_INVALID) return 43; /* null operand */
        nb.count = 5; nb.knots = 2;
        if (ungar_model_dense_jacobian_tiles(q, &nb, dummy, 0) != UNGAR_E_INVALID) return 44; /* count not a multiple of knots */
        if (ungar_tiles_gather(q, dummy, 4, 1, 0, 0) != UNGAR_E_INVALID || ungar_tiles_gather(q, dummy, 5, 2, (const ungar_operand*)&nb.jac, 0) != UNGAR_E_INVALID) return 45;
        ungar_model_close(q);
    }
    ungar_model_close(m);
    if (rc != UNGAR_OK) return rc;
    /* Error behaviour of the batched entry points: arguments are validated before anything touches a device, so a bad call returns
     * UNGAR_E_INVALID (never crashes, never launches) and leaves a message for ungar_last_error() -- also on a machine without a GPU. */
    {
        double dummy[4] = {0, 0, 0, 0};
        double alphas[17] = {1, .5, .25, .125, .0625, .03125, .015625, .0078125, .00390625, .001953125, .0009765625, .00048828125, .000244140625, .0001220703125, 6e-5, 3e-5, 1e-5};
        ungar_operand op = {0, 0, 0, 0};
        ungar_line_search_parameters ls = {1e-4, 1e-6, 1e-2, 1e-4, 1e-6, 1e-6, 0.5};
        ungar_ocp_merit_args merit;
        int bad = 0;
        op.base = dummy;
        bad += ungar_transpose_nodes(0, 1, 1, dummy, 1, 1, 4, 1, 0) != UNGAR_E_INVALID;
        bad += ungar_transpose_nodes(dummy, 1, 1, dummy, 1, 1, -1, 1, 0) != UNGAR_E_INVALID;
        bad += ungar_transpose_nodes(dummy, 1, 1, dummy, 1, 1, 0, 1, 0) != UNGAR_OK; /* empty batch: nothing to do */
        bad += ungar_gn_hessian_upper_tiles(0, 4, 0, 0, dummy, 4, 1, 2, 1, 2, 4, 0) != UNGAR_E_INVALID;
        bad += ungar_gn_hessian_upper_tiles(dummy, 2, 0, 0, dummy, 4, 1, 2, 1, 2, 4, 0) != UNGAR_E_INVALID; /* element stride < count */
        bad += ungar_gn_hessian_upper_lanes(dummy, 4, 0, 0, dummy, 4, 1, 1, 1, 2, 4, 0) != UNGAR_E_INVALID;  /* ld_g < cols */
        bad += ungar_ocp_trial_points(13, 4, 30, 8, &op, &op, &op, &op, alphas, 17, &op, &op, 0) != UNGAR_E_INVALID; /* > 16 candidates */
        bad += ungar_ocp_trial_points(13, 4, 30, 8, &op, &op, &op, &op, 0, 14, &op, &op, 0) != UNGAR_E_INVALID;
        bad += ungar_ocp_line_search_select(13, 4, 30, 8, &ls, alphas, 14, 0, dummy, dummy, dummy, dummy, dummy, &op, &op, &op, &op, 0, 0) != UNGAR_E_INVALID;
        merit.nx = 13; merit.nu = 4; merit.horizon = 30; merit.batch = 10; merit.nh = 0;
        merit.X = op; merit.xm = op; merit.f = op; merit.cost = op; merit.cost_terminal = op; merit.h = op;
        merit.barrier.type = 0; merit.barrier.reserved = 0; merit.barrier.stiffness = 1; merit.barrier.epsilon = 1;
        merit.violation_multiplier = 1; merit.cost_grad = op; merit.cost_grad_terminal = op; merit.dX = op; merit.dU = op;
        merit.theta = dummy; merit.phi = dummy; merit.slope = 0;
        bad += ungar_ocp_merit_stacked(&merit, 4, 0) != UNGAR_E_INVALID; /* the stacked batch is not a multiple of the period */
        {   /* shooting problems with carried quantities: dimensions and candidate counts are checked before anything is launched */
            ungar_shooting_dims dims = {13, 4, 4, 15, 21, 30, 8, 1, 0};
            ungar_shooting_dims wrong = {13, 4, 3, 15, 21, 30, 8, 1, 0}; /* carry_inputs needs nc == nu */
            ungar_shooting_assemble_args asm_args;
            void* ptr = dummy;
            memset(&asm_args, 0, sizeof asm_args);
            asm_args.dims = dims;
            bad += ungar_shooting_trial_rows(&dims, dummy, dummy, dummy, alphas, 17, dummy, 0, 0) != UNGAR_E_INVALID;
            bad += ungar_shooting_trial_rows(&wrong, dummy, dummy, dummy, alphas, 14, dummy, 0, 0) != UNGAR_E_INVALID;
            bad += ungar_shooting_trial_rows(&dims, dummy, dummy, dummy, alphas, 14, dummy, 5, 0) != UNGAR_E_INVALID; /* stride below the stacked node count */
            bad += ungar_shooting_trial_rows_listed(&dims, dummy, dummy, dummy, alphas, 14, 0, 3, dummy, 0, 0) != UNGAR_E_INVALID; /* listed instances without the list */
            bad += ungar_shooting_trial_rows_listed(&dims, dummy, dummy, dummy, alphas, 14, (const int32_t*)dummy, dims.batch + 1, dummy, 0, 0) != UNGAR_E_INVALID; /* more listed than there are */
            bad += ungar_shooting_assemble(&asm_args, 0) != UNGAR_E_INVALID; /* null operands */
            bad += ungar_shooting_select(&dims, &ls, alphas, 14, 0, dummy, dummy, dummy, dummy, dummy, dummy, dummy, 0, 0, dummy, dummy, 0, 0, 0, 0) != UNGAR_E_INVALID;
            /* listed variants: a list exactly when listed > 0; the next list is filled through the counter of a call that is not the last, and is not the list being read */
            bad += ungar_shooting_select_listed(&dims, &ls, alphas, 2, dummy, dummy, dummy, dummy, dummy, dummy, dummy, dummy, 0, 0, dummy, dummy, 0, UNGAR_SEARCH_NOT_LAST,
                                                (int32_t*)dummy, 0, 3, 0, 0) != UNGAR_E_INVALID;
            bad += ungar_shooting_select_listed(&dims, &ls, alphas, 2, dummy, dummy, dummy, dummy, dummy, dummy, dummy, dummy, 0, 0, dummy, dummy, 0, 0, (int32_t*)dummy, 0, 0,
                                                (int32_t*)dummy, 0) != UNGAR_E_INVALID; /* next list in the LAST call */
            bad += ungar_shooting_select_listed(&dims, &ls, alphas, 2, dummy, dummy, dummy, dummy, dummy, dummy, dummy, dummy, 0, 0, dummy, dummy, 0, UNGAR_SEARCH_NOT_LAST,
                                                (int32_t*)dummy, (const int32_t*)dummy, 3, (int32_t*)dummy, 0) != UNGAR_E_INVALID; /* next list == list */
            bad += ungar_device_malloc(&ptr, -1) != UNGAR_E_INVALID;
            bad += ungar_device_malloc(&ptr, 0) != UNGAR_OK || ptr != 0;
            bad += ungar_function_forward_zero_nodes(0, &op, &op, 4, 2, 0) != UNGAR_E_INVALID;
            bad += ungar_function_host_call_resident(0, 1) != 0; /* no function: no resident kernel */
            bad += ungar_function_eval_host(0, 0, dummy, dummy) != UNGAR_E_INVALID;
        }
        bad += ungar_last_error()[0] == 0;
        printf("argument checks: %d unexpected\n", bad);
        if (bad) return 100 + bad;
    }
    return rc;
}
