// Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3",
// SC'11) -- the counter-based generator that replaces the reference's OS-seeded thread_rng
// (sim/src/simulator.rs:342,630).  Shared by host code and HIP kernels of libesim.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define ESIM_HD __host__ __device__ __forceinline__
#else
#define ESIM_HD static inline
#endif

struct philox_out { uint32_t w0, w1, w2, w3; };

ESIM_HD uint32_t esim_mulhi32(uint32_t a, uint32_t b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __umulhi(a, b);
#else
    return (uint32_t)(((uint64_t)a * b) >> 32);
#endif
}

ESIM_HD philox_out philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                 uint32_t k0, uint32_t k1)
{
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        // one 32x32->64 product each (v_mad_u64_u32 on gfx950: half the quarter-rate multiplies of mul_hi + mul_lo)
        const uint64_t p0 = (uint64_t)M0 * c0, p1 = (uint64_t)M1 * c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        c0 = hi1 ^ c1 ^ k0; c1 = lo1;
        c2 = hi0 ^ c3 ^ k1; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    philox_out o = { c0, c1, c2, c3 };
    return o;
}

// 32-bit integer of the exposure draw for (citizen, step, slot): uniform = word * 2^-32.  Steps 4k .. 4k+3 share the block
// (citizen, k, slot), step t takes its word t & 3 -- all 128 bits of a block are used.
ESIM_HD uint32_t esim_u32_of(const philox_out &o, uint32_t step)
{
    const uint32_t h = step & 3u;
    return h == 0u ? o.w0 : h == 1u ? o.w1 : h == 2u ? o.w2 : o.w3;
}

ESIM_HD philox_out esim_draw_block(uint64_t seed, uint32_t citizen, uint32_t step, uint32_t slot)
{
    return philox4x32_10(citizen, step >> 2, slot, 0u, (uint32_t)seed, (uint32_t)(seed >> 32));
}

ESIM_HD uint32_t esim_u32(uint64_t seed, uint32_t citizen, uint32_t step, uint32_t slot)
{
    return esim_u32_of(esim_draw_block(seed, citizen, step, slot), step);
}

// draw slots (RNG contract)
enum { ESIM_SLOT_HOME = 0, ESIM_SLOT_WORK = 1, ESIM_SLOT_BUS = 2, ESIM_SLOT_BUS_ORDER = 3,
       ESIM_SLOT_VACCINE = 4, ESIM_SLOT_ROOM0 = 16 };
