/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C99, fp32, no FMA contraction) of the ArcNerf volumetric
 * rendering hot path.  It is the checker for the HIP kernels in arcnerf_amd/csrc and the
 * timed CPU baseline in bench.py.  Nothing in the product path may import, link or call
 * this library: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
 *
 * Every function cites the reference file:line (relative to /root/reference) it restates.
 * Pinning status: checked against golden vectors generated from the reference's own
 * torch path (tests/golden/make_golden.py) for everything that has a torch implementation;
 * the CUDA-only sampler (K3) has no runnable reference here and is pinned only through the
 * torch-side invariants listed in SURVEY.md §8(c) ("parity unpinned" for K3/K4 bit streams).
 */
#ifndef ORC_COMMON_H
#define ORC_COMMON_H

#include <math.h>
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <stdlib.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_API __attribute__((visibility("default")))

/* ---- pcg32: arcnerf/ops/include/pcg32.h:38-165 ------------------------------------ */
#define ORC_PCG32_MULT 0x5851f42d4c957f2dULL

typedef struct {
    uint64_t state;
    uint64_t inc;
} orc_pcg32;

static inline uint32_t orc_pcg32_next_uint(orc_pcg32 *r) {
    uint64_t old = r->state;
    r->state = old * ORC_PCG32_MULT + r->inc;
    uint32_t xorshifted = (uint32_t)(((old >> 18u) ^ old) >> 27u);
    uint32_t rot = (uint32_t)(old >> 59u);
    return (xorshifted >> rot) | (xorshifted << ((~rot + 1u) & 31));
}

static inline void orc_pcg32_seed(orc_pcg32 *r, uint64_t initstate, uint64_t initseq) {
    r->state = 0u;
    r->inc = (initseq << 1u) | 1u;
    orc_pcg32_next_uint(r);
    r->state += initstate;
    orc_pcg32_next_uint(r);
}

static inline float orc_pcg32_next_float(orc_pcg32 *r) {
    union { uint32_t u; float f; } x;
    x.u = (orc_pcg32_next_uint(r) >> 9) | 0x3f800000u;
    return x.f - 1.0f;
}

static inline void orc_pcg32_advance(orc_pcg32 *r, int64_t delta_) {
    uint64_t cur_mult = ORC_PCG32_MULT, cur_plus = r->inc, acc_mult = 1u, acc_plus = 0u;
    uint64_t delta = (uint64_t)delta_;
    while (delta > 0) {
        if (delta & 1) {
            acc_mult *= cur_mult;
            acc_plus = acc_plus * cur_mult + cur_plus;
        }
        cur_plus = (cur_mult + 1) * cur_plus;
        cur_mult *= cur_mult;
        delta /= 2;
    }
    r->state = acc_mult * r->state + acc_plus;
}

#endif
