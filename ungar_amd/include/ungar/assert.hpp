// ungar_amd :: UNGAR_ASSERT (reference include/ungar/assert.hpp): active in every build type, reports the
// failed condition with its location and aborts.
#pragma once

#include <cstdio>
#include <cstdlib>

#define UNGAR_ASSERT(condition)                                                                                 \
    do {                                                                                                        \
        if (!(condition)) {                                                                                     \
            std::fprintf(stderr, "[ungar] assertion failed: %s (%s:%d)\n", #condition, __FILE__, __LINE__);     \
            std::abort();                                                                                       \
        }                                                                                                       \
    } while (false)
