// f64 instantiation: compile with -ffp-contract=off (see build.py).
#define RT_REAL double
#include "rt_kernels.inc"
