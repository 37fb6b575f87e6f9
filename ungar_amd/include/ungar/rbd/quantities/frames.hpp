// ungar_amd :: reference include path ungar/rbd/quantities/frames.hpp; all quantities live in quantities.hpp.
#pragma once
#include "quantities.hpp"
