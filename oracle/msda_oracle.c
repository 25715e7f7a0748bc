/*
 * TEST INFRASTRUCTURE — NOT PRODUCT CODE.  See msda_oracle_impl.h for the
 * reference file:line each piece restates.
 *
 * Parity status: PINNED.  the .npz files under tests/golden/ hold inputs and outputs produced by
 * importing the reference's own native_multiscale_deformable_attention
 * (/root/reference/src/msda_triton/frontend.py:15-68) in the build container
 * (generator: tests/golden/make_golden.py); tests/test_oracle.py checks this
 * restatement against every one of them.
 *
 * Build: make -C oracle   (gcc -O2 -fopenmp -shared -fPIC -> oracle/_build/libmsda_oracle.so)
 */
#include <math.h>
#include <stdint.h>

#define MSDA_ORACLE_MAX_LEVELS 32

#define REAL float
#define FN(name) name##_f32
#define FLOOR floorf
#include "msda_oracle_impl.h"
#undef REAL
#undef FN
#undef FLOOR

#define REAL double
#define FN(name) name##_f64
#define FLOOR floor
#include "msda_oracle_impl.h"
#undef REAL
#undef FN
#undef FLOOR

#ifdef _OPENMP
#include <omp.h>
int msda_oracle_num_threads(void) { return omp_get_max_threads(); }
void msda_oracle_set_num_threads(int n) { omp_set_num_threads(n); }
#else
int msda_oracle_num_threads(void) { return 1; }
void msda_oracle_set_num_threads(int n) { (void)n; }
#endif
