// rocrand_tables.cpp -- exposes rocRAND 4.2's precomputed XORWOW jump tables
// (rocrand_xorwow_precomputed.h:1127,3313) so that tests can compare the matrices the oracle and the
// product derive by repeated squaring against the library the reference's HIP build links.
// TEST INFRASTRUCTURE ONLY.
#ifndef __HIP_PLATFORM_AMD__
#define __HIP_PLATFORM_AMD__ 1
#endif
#include <rocrand/rocrand_xorwow.h>
#include <rocrand/rocrand_version.h>

extern "C" {
const unsigned int *rocrand_h_xorwow_jump_matrix(int i) { return h_xorwow_jump_matrices[i]; }
const unsigned int *rocrand_h_xorwow_sequence_jump_matrix(int i) { return h_xorwow_sequence_jump_matrices[i]; }
int rocrand_tables_version(void) { return ROCRAND_VERSION; }
}
