// ungar_amd :: reference include path `ungar/mvariable_lazy_map.hpp`.
#pragma once
#include "mvariable.hpp"
#include "variable_lazy_map.hpp"
