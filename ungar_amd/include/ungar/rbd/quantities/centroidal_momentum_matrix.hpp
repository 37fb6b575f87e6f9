// ungar_amd :: reference include path ungar/rbd/quantities/centroidal_momentum_matrix.hpp; all quantities live in quantities.hpp.
#pragma once
#include "quantities.hpp"
