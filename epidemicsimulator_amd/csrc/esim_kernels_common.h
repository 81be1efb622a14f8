// esim_kernels_common.h -- what every kernel file uses: launch shapes, DiseaseStatus from the citizen word, the global
// schedule, list appends, the diagnostics timers.  Included by esim_kernels.hip (one translation unit).
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/esim.h"
#include "esim_device.h"
#include "philox.h"

#define TPB 256
#define FIN_TPB 1024

__device__ __forceinline__ uint32_t status_of(uint32_t te, uint32_t t, uint32_t et, uint32_t it)
{
    if (te == TE_SUSCEPTIBLE) return ESIM_SUSCEPTIBLE;
    if (te == TE_VACCINATED) return ESIM_VACCINATED;
    if (te == TE_RECOVERED) return ESIM_RECOVERED;
    uint32_t d = t + TE_BIAS - te;               // steps since Exposed(0)
    if (d <= et) return ESIM_EXPOSED;            // Exposed(d), disease.rs:53-58
    if (d <= et + 1u + it) return ESIM_INFECTED; // Infected(d - et - 1), disease.rs:60-65
    return ESIM_RECOVERED;
}

// Counters that other waves bump with atomics are read past the L1 (they may have been cached by an
// earlier step of the persistent kernel).
__device__ __forceinline__ uint32_t ld(const uint32_t *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Citizen::execute_time_step's schedule (citizen.rs:176-206) for the whole population at once.
// The four hours are distinct (checked in esim_create), so every citizen takes the same arm
// up to the `uses_public_transport` guard, which only decides who is on the bus.
__device__ __forceinline__ void schedule(const Dev &d, const Ctrl *ctrl, uint32_t t, uint32_t &at_work, uint32_t &bus_dir)
{
    at_work = ctrl->at_work; bus_dir = ctrl->bus_dir;
    if (ctrl->lockdown) return;                                   // citizen.rs:176 (Q8)
    const uint32_t h = t % 24u;
    if (h == d.start_hour - 1u) bus_dir = 1u;                     // :179-184 (users of public transport)
    else if (h == d.start_hour) { at_work = 1u; bus_dir = 0u; }   // :186-189
    else if (h == d.end_hour - 1u) bus_dir = 2u;                  // :191-196
    else if (h == d.end_hour) { at_work = 0u; bus_dir = 0u; }     // :198-201
    else bus_dir = 0u;                                            // :202-204
}

// What is in force while step t runs: taken from the control block (one step at a time) or from the
// decision table of a pipelined chunk.
struct StepEnv { uint32_t t, mask, at_work, bus_dir; };

__device__ __forceinline__ StepEnv env_from_ctrl(const Dev &d, const Ctrl *ctrl)
{
    StepEnv e;
    e.t = ctrl->t; e.mask = ctrl->mask;
    schedule(d, ctrl, e.t, e.at_work, e.bus_dir);
    return e;
}

__device__ __forceinline__ StepEnv env_from_dec(const Dev &d, uint32_t t, uint32_t j)
{
    const Decision q = d.dec[j];
    StepEnv e;
    e.t = t; e.mask = q.mask; e.at_work = q.at_work; e.bus_dir = q.bus_dir;
    return e;
}

// Census of simulator.rs:178 from the exposure-time histogram; called by a whole block.
// out[0..4] = S,E,I,R,V of this shard after the tick of step t (before this step's exposures).
__device__ void census_block(const Dev &d, const Ctrl *ctrl, uint32_t t, uint32_t *out /* shared, >= 5 */)
{
    if (threadIdx.x < 5) out[threadIdx.x] = 0;
    __syncthreads();
    const int hi_e = (int)(t + TE_BIAS);                          // d = 0
    const int lo_e = hi_e - (int)d.exposed_time;                  // d = exposed_time
    const int hi_i = lo_e - 1;                                    // d = exposed_time + 1
    const int lo_i = hi_i - (int)d.infected_time;
    uint32_t e = 0, i = 0;
    for (int k = lo_i + (int)threadIdx.x; k <= hi_e; k += (int)blockDim.x) {
        if (k < 0) continue;
        const uint32_t v = ld(&d.hist[k]);
        if (k >= lo_e) e += v; else i += v;
    }
    if (e) atomicAdd(&out[1], e);
    if (i) atomicAdd(&out[2], i);
    __syncthreads();
    if (threadIdx.x == 0) {
        out[0] = ld(&ctrl->n_susceptible); out[4] = ld(&ctrl->n_vaccinated);
        out[3] = d.n - out[0] - out[4] - out[1] - out[2];
    }
    __syncthreads();
}

__device__ __forceinline__ void append(uint32_t *list, uint32_t *len, uint32_t v)
{
    list[atomicAdd(len, 1u)] = v;
}

// Diagnostics build (make prof): every wavefront of the chunk-pass kernels stores its timers in its own row of
// d.prof_buf -- no atomics, so the measurement does not serialise the kernel it measures.
#ifdef ESIM_WAVE_PROFILE
#define PROF_ROW 16u
#define PROF_NOW() ((uint32_t)wall_clock64())
#define PROF_PUT(d, i, v) do { if ((threadIdx.x & 63u) == 0) (d).prof_buf[(size_t)(((blockIdx.x * TPB + threadIdx.x) >> 6)) * PROF_ROW + (i)] = (uint32_t)(v); } while (0)
#define BOOKS_PROF(d, i, v) do { if (threadIdx.x == 0) (d).prof_buf[(size_t)16383 * PROF_ROW + (i)] = (uint32_t)(v); } while (0)
#define TINY_PROF(d, i) do { if (threadIdx.x == 0) (d).prof_buf[(size_t)16382 * PROF_ROW + (i)] = (uint32_t)wall_clock64(); } while (0)
#else
#define BOOKS_PROF(d, i, v) do { (void)sizeof(v); } while (0)
#define TINY_PROF(d, i) do { } while (0)
#define PROF_NOW() 0u
#define PROF_PUT(d, i, v) do { (void)sizeof(v); } while (0)
#endif


// Counting build (make count; tools/work_counts.py): what the chunk pass WORKS ON, counted exactly -- the sparse pass's own
// units of work, which its roofline is priced in.  Per-lane tallies are summed over the wavefront where all its
// lanes meet again and added to 64-bit device counters by lane 0; nothing of this exists in the product build.
#ifdef ESIM_COUNT_WORK
enum { WK_ENTRIES = 0,      // Infected log entries (and received commuter records) k_chunk_marks looked at
       WK_KEYS,             // hash keys looked up / claimed (home, work, room, route per entry)
       WK_CLAIMS,           // items claimed
       WK_RECORDS,          // interval records left with somebody else's item (slot records + overflow)
       WK_DIRECT,           // per-step counter atomics (schools, commuters of other shards)
       WK_FOLDED,           // overflow records k_chunk_fold summed
       WK_ITEMS,            // items k_chunk_draw took (buildings, rooms)
       WK_MEMBERS,          // member entries staged for drawing (id + citizen word each)
       WK_MEMBERS_IDX,      // ... of which through an index list (workers, room participants): 4 more bytes each
       WK_PAIRS,            // (member, slot of four time steps) pairs looked at
       WK_PAIRS_ACTIVE,     // ... that take at least one draw
       WK_BLOCKS,           // Philox4x32-10 blocks computed for exposure draws
       WK_DRAWS,            // Bernoulli draws the reference would make: (member, step) pairs, one per Infected room-mate in rooms
       WK_UNITS,            // deferred units drawn by k_chunk_units
       WK_ROUTE_PAIRS,      // (route, bus step) pairs ranked
       WK_RIDERS,           // riders ranked (one Philox block each for the order key) 
       WK_BUS_DRAWS,        // bus draws (one Philox block each)
       WK_HITS,             // successful draws (atomicMin issued)
       WK_N };
#define WORK_TALLY uint32_t wk_[WK_N]; for (int wk_i = 0; wk_i < WK_N; ++wk_i) wk_[wk_i] = 0u
#define WORK_ADD(i, x) (wk_[i] += (uint32_t)(x))
// (all lanes of the wavefront must be here together)
#define WORK_FLUSH(d) do { for (int wk_i = 0; wk_i < WK_N; ++wk_i) { uint32_t x_ = wk_[wk_i]; for (int o_ = 32; o_ > 0; o_ >>= 1) x_ += __shfl_xor(x_, o_, 64); \
        if ((threadIdx.x & 63u) == 0 && x_) atomicAdd(&(d).work_cnt[wk_i], (unsigned long long)x_); wk_[wk_i] = 0u; } } while (0)
#define WORK_ARG , uint32_t *wk_
#define WORK_PASS , wk_
#else
#define WORK_TALLY do { } while (0)
#define WORK_ADD(i, x) do { } while (0)
#define WORK_FLUSH(d) do { } while (0)
#define WORK_ARG
#define WORK_PASS
#endif
