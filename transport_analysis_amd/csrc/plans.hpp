// plans.hpp — the compiled FFT plans (M = 2^a or 5*2^a, 16 <= M <= 10240).
// A length-M complex-f64 series is 16*M bytes of LDS; 10240 fills the CU's 160 KiB.
#pragma once
#include "fft_engine.hpp"

namespace ta {

// X(NT, radices...)   — the last radix has unit stride and feeds the accumulators.
// NT = M/32 (2^a) or M/40 (5*2^a), at least 64: one wave per SIMD slot for the big
// plans, so a thread can hold the gathered pair, both accumulator sets and one
// butterfly's data without spilling (<= 512 VGPRs).
#define TA_PLANS_POW2(X)        \
    X(64, 4, 4)                 \
    X(64, 8, 4)                 \
    X(64, 8, 8)                 \
    X(64, 16, 8)                \
    X(64, 16, 16)               \
    X(64, 8, 8, 8)              \
    X(64, 16, 8, 8)             \
    X(64, 16, 16, 8)            \
    X(128, 16, 16, 16)          \
    X(256, 16, 8, 8, 8)

#define TA_PLANS_FIVE(X)        \
    X(64, 5, 4)                 \
    X(64, 5, 8)                 \
    X(64, 5, 16)                \
    X(64, 5, 8, 4)              \
    X(64, 5, 8, 8)              \
    X(64, 5, 16, 8)             \
    X(64, 5, 16, 16)            \
    X(64, 5, 8, 8, 8)           \
    X(128, 5, 16, 8, 8)         \
    X(256, 5, 16, 16, 8)

}  // namespace ta
