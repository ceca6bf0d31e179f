// fsnap_device_common.h — vector types shared by the gfx950 translation units
// (fsnap_syrk.hip, fsnap_rows.hip, fsnap_chol.hip).  Internal; the public boundary is include/fsnap_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));
typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
typedef unsigned int u2 __attribute__((ext_vector_type(2)));
