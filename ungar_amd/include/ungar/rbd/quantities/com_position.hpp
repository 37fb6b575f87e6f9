// ungar_amd :: reference include path ungar/rbd/quantities/com_position.hpp; all quantities live in quantities.hpp.
#pragma once
#include "quantities.hpp"
