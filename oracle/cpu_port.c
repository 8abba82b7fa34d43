/*
 * cpu_port.c -- TEST INFRASTRUCTURE ONLY (never linked into the product library).
 * Instantiates cpu_port_impl.h for f32 and f64.  See that file for what it restates.
 */
#ifndef _GNU_SOURCE
#define _GNU_SOURCE
#endif
#include <math.h>
#include <sched.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define REAL float
#define SUF f32
#include "cpu_port_impl.h"
#undef REAL
#undef SUF

#define REAL double
#define SUF f64
#include "cpu_port_impl.h"
#undef REAL
#undef SUF

#ifdef _OPENMP
int cpu_port_max_threads(void) { return omp_get_max_threads(); }
#else
int cpu_port_max_threads(void) { return 1; }
#endif
