// ungar_amd :: measurement switches.
//
// A/B routes between kernels, per-phase clocks and experiment knobs are read from the environment ONLY in the measurement build of
// the library (-DUNGAR_AMD_MEASUREMENT -> ungar_amd/lib/measurement/libungar_amd.so, loaded by tools/ and by the agreement tests
// that compare two routes).  In the shipped library UNGAR_MEASUREMENT_SWITCH(<name>) is a null pointer at compile time: the route
// it guards is not selectable, the kernels only it reaches are not instantiated, and the name does not appear in the binary
// (tests/test_abi.py::test_shipped_library_has_no_measurement_switches).  Environment variables that ARE part of the interface
// (UNGAR_CODEGEN_FOLDER, UNGAR_HIPCC, UNGAR_AMD_JIT_FLAGS, UNGAR_AMD_SCALAR_STORES, ...) are listed in INTEGRATION.md and, where they
// change generated code, are part of the model-cache key.
#pragma once

#include <cstdlib>

#if defined(UNGAR_AMD_MEASUREMENT)
#define UNGAR_MEASUREMENT_SWITCH(name) (std::getenv(name))
#define UNGAR_AMD_MEASUREMENT_BUILD 1
#else
#define UNGAR_MEASUREMENT_SWITCH(name) (static_cast<const char*>(nullptr))
#define UNGAR_AMD_MEASUREMENT_BUILD 0
#endif
