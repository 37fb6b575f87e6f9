/* The drop-in boundary is a C ABI: this file is compiled as C99 with -pedantic -Werror (tests/test_abi.py) to prove that
 * include/ungar_amd.h carries no C++ in its declarations, and exercises the host-only entry points. */
#include "ungar_amd.h"
#include <stdio.h>
int main(void) {
    ungar_model* m = 0;
    int rc = ungar_model_open("quadrotor_cost", &m);
    ungar_model_info info;
    if (rc == UNGAR_OK && ungar_model_get_info(m, &info) == UNGAR_OK) printf("%s nx=%lld ny=%lld hes_nnz=%lld version=%s\n", ungar_model_name(m), (long long)info.nx, (long long)info.ny, (long long)info.hes_nnz, ungar_version());
    ungar_model_close(m);
    return rc;
}
