/*
 * ref_shim.h -- lets the reference's own device-function lines run on the host, one "thread" after
 * another.  TEST INFRASTRUCTURE ONLY; used solely by oracle/Makefile's `ref` target, which reads
 * the reference sources where they lie (/root/reference) and writes only into oracle/_ref/.
 *
 * Nothing here restates reference code.  It supplies what a plain C++ compile lacks:
 *  - the CUDA built-in index variables, as ordinary globals the driver sets per "thread";
 *  - the cuRAND names, mapped to rocRAND's host-callable XORWOW exactly as the reference's own
 *    `make hip` route maps them (hipify: curand -> hiprand -> rocrand).
 * HIP's __device__/__global__/__host__ expand to nothing in a non-HIP compile, so the
 * reference lines and the rocRAND engine compile unmodified.
 */
#ifndef REF_SHIM_H_
#define REF_SHIM_H_

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

#ifndef __HIP_PLATFORM_AMD__
#define __HIP_PLATFORM_AMD__ 1
#endif
#include <rocrand/rocrand_xorwow.h>
#include <rocrand/rocrand_uniform.h>

typedef rocrand_state_xorwow curandState_t;
#define curand_init rocrand_init
#define curand_uniform_double rocrand_uniform_double

struct ref_dim3 { unsigned x, y, z; };
static ref_dim3 ref_blockIdx, ref_blockDim, ref_threadIdx;
#define blockIdx ref_blockIdx
#define blockDim ref_blockDim
#define threadIdx ref_threadIdx

#define SAMPLES_PER_THREAD (ref_samples_per_thread)
static int ref_samples_per_thread = 50;

#endif
