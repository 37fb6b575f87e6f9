// ungar_amd :: 'anymal_reg' -- the structured ANYmal B program emitted as ONE straight-line body
// (register allocation left to the compiler).  Comparison arm of the phased 'anymal' kernel.
#include "../gen/anymal_reg_gen.hpp"
#include "node_kernel.hpp"

UNGAR_AMD_DEFINE_NODE_MODEL(anymal_reg, 64)
