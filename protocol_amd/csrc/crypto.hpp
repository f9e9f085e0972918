// Shared between the host crypto (crypto.cpp) and the batched device kernels (ecdsa.hip).
#pragma once
#include <vector>

#include "fr_host.hpp"

namespace et {

// Poseidon 5x5 parameters (canonical integers): (8 + 60) * 5 round constants and the 5 x 5 MDS matrix, row-major
void poseidon5_parameters(std::vector<FrRaw>& round_constants, std::vector<FrRaw>& mds);

}  // namespace et
