// esim_kernels.hip -- gfx950 kernels of the per-timestep Citizen update loop.
//
// One time step (reference: Simulator::step, sim/src/simulator.rs:131-152) is, on the device:
//   k_tick     generate_exposures  (simulator.rs:155-260): schedule + census + infected-per-building
//   k_expose   apply_exposures, buildings (simulator.rs:268-358)
//   k_bus_*    apply_exposures, public transport (simulator.rs:360-401)
//   k_finish   apply_interventions (simulator.rs:455-556) + the StatisticEntry of the step
// All draws are Philox4x32-10 keyed (global citizen, step, slot); probabilities are integer
// thresholds ceil(q*2^53) from a host-built LUT, so every comparison is exact integer work.
#include <hip/hip_runtime.h>
#include "../../include/esim.h"
#include "esim_device.h"
#include "philox.h"

#define TPB 256

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

__device__ __forceinline__ uint32_t wave_sum(uint32_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// ------------------------------------------------------------------------------------ k_tick
// Citizen::execute_time_step (citizen.rs:168-216) for every citizen, the census of
// simulator.rs:178, the rider test of :181-186 and the infected-building push of :187-198.
__global__ __launch_bounds__(TPB) void k_tick(Dev d)
{
    Ctrl *ctrl = d.ctrl;
    if (ctrl->finished) return;
    const uint32_t t = ctrl->t;
    const bool lock = ctrl->lockdown != 0;
    const uint32_t h = t % 24u;
    uint32_t cS = 0, cE = 0, cI = 0, cR = 0, cV = 0, cB = 0;
    for (uint32_t c = blockIdx.x * TPB + threadIdx.x; c < d.n; c += gridDim.x * TPB) {
        const uint32_t st = d.state[c];
        const uint32_t fl = d.flags[c];
        uint32_t ns = st;
        if (!lock) {                                                      // citizen.rs:176
            const bool pt = fl & FL_USES_PT;
            if (h == d.start_hour - 1u && pt) ns |= ST_ON_BUS;            // :179-184
            else if (h == d.start_hour) ns = (ns | ST_AT_WORK) & ~ST_ON_BUS;   // :186-189
            else if (h == d.end_hour - 1u && pt) ns |= ST_ON_BUS;         // :191-196
            else if (h == d.end_hour) ns &= ~(ST_AT_WORK | ST_ON_BUS);    // :198-201
            else ns &= ~ST_ON_BUS;                                        // :202-204
        }
        const uint32_t cls = status_of(ns & ST_TE_MASK, t, d.exposed_time, d.infected_time);
        cS += cls == ESIM_SUSCEPTIBLE; cE += cls == ESIM_EXPOSED; cI += cls == ESIM_INFECTED;
        cR += cls == ESIM_RECOVERED;   cV += cls == ESIM_VACCINATED;
        if (ns & ST_ON_BUS) cB++;                                         // simulator.rs:181-186
        else if (cls == ESIM_INFECTED) {                                  // :187-198
            const bool at_work = (ns & ST_AT_WORK) && (fl & FL_HAS_WORK);
            const uint32_t b = at_work ? d.work[c] : d.home[c];
            atomicAdd(&d.cnt_bld[b], 1u);
            if (at_work && (fl & FL_WORK_SCHOOL)) atomicAdd(&d.cnt_room[d.room[c]], 1u);
        }
        if (ns != st) d.state[c] = (uint16_t)ns;
    }
    __shared__ uint32_t red[6][TPB / 64];
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    cS = wave_sum(cS); cE = wave_sum(cE); cI = wave_sum(cI); cR = wave_sum(cR); cV = wave_sum(cV); cB = wave_sum(cB);
    if (lane == 0) { red[0][wv] = cS; red[1][wv] = cE; red[2][wv] = cI; red[3][wv] = cR; red[4][wv] = cV; red[5][wv] = cB; }
    __syncthreads();
    if (threadIdx.x < 6) {
        uint32_t s = 0;
        for (uint32_t w = 0; w < TPB / 64; ++w) s += red[threadIdx.x][w];
        if (s) atomicAdd(threadIdx.x < 5 ? &ctrl->counts[threadIdx.x] : &ctrl->n_riders, s);
    }
}

// Did the vaccination programme start in this step?  (interventions.rs:132-141; the infected
// fraction only depends on the census, because exposures move S->E and leave I alone.)
__device__ __forceinline__ bool trigger_now(const Dev &d, const Ctrl *ctrl)
{
    const uint32_t total = ctrl->counts[0] + ctrl->counts[1] + ctrl->counts[2] + ctrl->counts[3] + ctrl->counts[4];
    const double x = (double)ctrl->counts[2] / (double)total;          // statistics.rs:252-254
    return !ctrl->vacc_active && d.thr_vacc < x;
}

// Threshold for Citizen::expose (citizen.rs:221-248): row 1 of the LUT is p - p*mask_effectiveness,
// which only applies to NON-compliant citizens while the global status is Everywhere (Q7).
__device__ __forceinline__ uint64_t threshold(const Dev &d, uint32_t fl, uint32_t mask, uint32_t n)
{
    const uint32_t row = (!(fl & FL_MASK_COMPLIANT) && mask == ESIM_MASK_EVERYWHERE) ? 1u : 0u;
    return d.thr[row * 256u + (n & 255u)];                              // `as u8`, citizen.rs:239
}

// All building draws of one susceptible citizen in step t (simulator.rs:308-350 seen from the
// candidate's side): the home list (building.rs:202), then the work list (building.rs:278) or
// the school-room multiset (building.rs:494-522).  Pure function of the infected counts.
__device__ __forceinline__ bool building_draws(const Dev &d, uint32_t c, uint32_t st, uint32_t fl,
                                               uint32_t t, uint32_t mask)
{
    const uint32_t g = d.id_base + c;
    const bool at_work = st & ST_AT_WORK;
    const bool same = fl & FL_SAME_AREA;
    // "If the Citizen is not currently in the Area, they haven't been exposed!" simulator.rs:324
    if (!at_work || same) {
        const uint32_t n = d.cnt_bld[d.home[c]];
        if (n && esim_u53(((uint64_t)d.seed_hi << 32) | d.seed_lo, g, t, ESIM_SLOT_HOME) < threshold(d, fl, mask, n))
            return true;
    }
    if ((fl & FL_HAS_WORK) && (at_work || same)) {
        const uint32_t n = d.cnt_bld[d.work[c]];
        if (n) {
            const uint64_t thr = threshold(d, fl, mask, n);
            const uint64_t seed = ((uint64_t)d.seed_hi << 32) | d.seed_lo;
            if (fl & FL_WORK_SCHOOL) {
                const uint32_t k = d.cnt_room[d.room[c]];               // one copy of the room per infected
                for (uint32_t j = 0; j < k; ++j)
                    if (esim_u53(seed, g, t, ESIM_SLOT_ROOM0 + j) < thr) return true;
            } else if (esim_u53(seed, g, t, ESIM_SLOT_WORK) < thr) return true;
        }
    }
    return false;
}

// ---------------------------------------------------------------------------------- k_expose
__global__ __launch_bounds__(TPB) void k_expose(Dev d)
{
    Ctrl *ctrl = d.ctrl;
    if (ctrl->finished) return;
    const uint32_t t = ctrl->t, mask = ctrl->mask;
    const bool trig = trigger_now(d, ctrl);
    uint32_t n_exp = 0, n_elig = 0;
    for (uint32_t c = blockIdx.x * TPB + threadIdx.x; c < d.n; c += gridDim.x * TPB) {
        const uint32_t st = d.state[c];
        if ((st & ST_TE_MASK) != TE_SUSCEPTIBLE) continue;               // is_susceptible(), simulator.rs:337
        const uint32_t fl = d.flags[c];
        if (building_draws(d, c, st, fl, t, mask)) {
            d.state[c] = (uint16_t)((st & ~ST_TE_MASK) | (t + TE_BIAS));  // Exposed(0), citizen.rs:244
            n_exp++;
        } else if (trig) {
            // eligible := everyone still Susceptible at the end of the trigger step (simulator.rs:487-513);
            // bus exposures of this step take the bit away again in k_bus_*.
            d.state[c] = (uint16_t)(st | ST_ELIGIBLE);
            n_elig++;
        }
    }
    __shared__ uint32_t red[2][TPB / 64];
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    n_exp = wave_sum(n_exp); n_elig = wave_sum(n_elig);
    if (lane == 0) { red[0][wv] = n_exp; red[1][wv] = n_elig; }
    __syncthreads();
    if (threadIdx.x < 2) {
        uint32_t s = 0;
        for (uint32_t w = 0; w < TPB / 64; ++w) s += red[threadIdx.x][w];
        if (s) atomicAdd(threadIdx.x == 0 ? &ctrl->exp_bld : &ctrl->elig_count, s);
    }
}

// ------------------------------------------------------------------------------------- buses
// A rider that is still Susceptible after the building phase draws once with the number of
// infected riders on the same bus (expose_citizens, simulator.rs:407-453).
__device__ __forceinline__ void bus_draw(const Dev &d, Ctrl *ctrl, uint32_t c, uint32_t st, uint32_t k,
                                         uint32_t t, uint32_t mask)
{
    const uint32_t fl = d.flags[c];
    const uint64_t seed = ((uint64_t)d.seed_hi << 32) | d.seed_lo;
    if (esim_u53(seed, d.id_base + c, t, ESIM_SLOT_BUS) < threshold(d, fl, mask, k)) {
        d.state[c] = (uint16_t)((st & ~(ST_TE_MASK | ST_ELIGIBLE)) | (t + TE_BIAS));
        atomicAdd(&ctrl->exp_bus, 1u);
        if (st & ST_ELIGIBLE) atomicSub(&ctrl->elig_count, 1u);           // simulator.rs:447-449
    }
}

// One wavefront per route with <= 64 riders: rank by (Philox key, id) with shuffles, buses are
// consecutive runs of bus_capacity ranks (replaces shuffle + pop, simulator.rs:362-388).
__global__ __launch_bounds__(TPB) void k_bus_small(Dev d)
{
    Ctrl *ctrl = d.ctrl;
    if (ctrl->finished || ctrl->n_riders == 0) return;
    const uint32_t t = ctrl->t, mask = ctrl->mask;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (blockIdx.x * TPB + threadIdx.x) >> 6, n_waves = (gridDim.x * TPB) >> 6;
    for (uint32_t ri = wave; ri < d.n_routes_small; ri += n_waves) {
        const uint32_t r = d.route_small[ri];
        const uint32_t off = d.route_off[r], s = d.route_off[r + 1] - off;
        uint32_t c = 0, st = 0;
        bool active = false, inf = false;
        if (lane < s) {
            c = d.route_riders[off + lane];
            st = d.state[c];
            active = st & ST_ON_BUS;
            inf = active && status_of(st & ST_TE_MASK, t, d.exposed_time, d.infected_time) == ESIM_INFECTED;
        }
        if (!__any(inf)) continue;                                        // no bus of this route has exposure_count > 0
        const uint32_t key = active ? philox4x32_10(d.id_base + c, t, ESIM_SLOT_BUS_ORDER, 0u, d.seed_lo, d.seed_hi).w0 : 0u;
        uint32_t rank = 0;
        for (uint32_t j = 0; j < s; ++j) {
            const uint32_t kj = __shfl(key, j, 64);
            const bool aj = __shfl((int)active, j, 64);
            rank += aj && (kj < key || (kj == key && j < lane));          // ids ascend with the lane
        }
        const uint32_t bus = rank / d.bus_capacity;
        uint32_t k = 0;
        for (uint32_t j = 0; j < s; ++j) {
            const uint32_t bj = __shfl(bus, j, 64);
            const bool ij = __shfl((int)inf, j, 64);
            k += ij && bj == bus;
        }
        if (active && k && (st & ST_TE_MASK) == TE_SUSCEPTIBLE) bus_draw(d, ctrl, c, st, k, t, mask);
    }
}

// One workgroup per route with > 64 riders (rare: a very large Output Area).  Same ordering rule,
// rank by counting through global scratch.
__global__ __launch_bounds__(TPB) void k_bus_big(Dev d)
{
    Ctrl *ctrl = d.ctrl;
    if (ctrl->finished || ctrl->n_riders == 0) return;
    const uint32_t t = ctrl->t, mask = ctrl->mask;
    for (uint32_t ri = blockIdx.x; ri < d.n_routes_big; ri += gridDim.x) {
        const uint32_t r = d.route_big[ri];
        const uint32_t off = d.route_off[r], s = d.route_off[r + 1] - off;
        int any_inf = 0;
        for (uint32_t i = threadIdx.x; i < s; i += TPB) {
            const uint32_t c = d.route_riders[off + i];
            const uint32_t st = d.state[c];
            const bool active = st & ST_ON_BUS;
            const bool inf = active && status_of(st & ST_TE_MASK, t, d.exposed_time, d.infected_time) == ESIM_INFECTED;
            d.bus_key[off + i] = active ? philox4x32_10(d.id_base + c, t, ESIM_SLOT_BUS_ORDER, 0u, d.seed_lo, d.seed_hi).w0 : 0u;
            d.bus_flag[off + i] = (uint8_t)((active ? 1u : 0u) | (inf ? 2u : 0u));
            d.bus_cnt[off + i] = 0u;
            any_inf |= inf;
        }
        any_inf = __syncthreads_or(any_inf);
        if (!any_inf) continue;
        for (uint32_t i = threadIdx.x; i < s; i += TPB) {
            const uint32_t fi = d.bus_flag[off + i];
            if (!(fi & 1u)) continue;
            const uint32_t key = d.bus_key[off + i];
            uint32_t rank = 0;
            for (uint32_t j = 0; j < s; ++j) {
                const uint32_t kj = d.bus_key[off + j];
                rank += (d.bus_flag[off + j] & 1u) && (kj < key || (kj == key && j < i));
            }
            const uint32_t bus = rank / d.bus_capacity;
            d.bus_idx[off + i] = bus;
            if (fi & 2u) atomicAdd(&d.bus_cnt[off + bus], 1u);
        }
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < s; i += TPB) {
            if (!(d.bus_flag[off + i] & 1u)) continue;
            const uint32_t c = d.route_riders[off + i];
            const uint32_t st = d.state[c];
            const uint32_t k = __hip_atomic_load(&d.bus_cnt[off + d.bus_idx[off + i]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (k && (st & ST_TE_MASK) == TE_SUSCEPTIBLE) bus_draw(d, ctrl, c, st, k, t, mask);
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------- exchange
// Sharded runs: pack the census and the infected counts of shared buildings/rooms, let the caller
// SUM-all-reduce, and scatter the totals back.
__global__ __launch_bounds__(TPB) void k_pack_a(Dev d)
{
    const Ctrl *ctrl = d.ctrl;
    const uint32_t i = blockIdx.x * TPB + threadIdx.x;
    const uint32_t nb = d.n_shared_bld, nr = d.n_shared_room;
    if (i < XA_HEADER) d.xa[i] = i < 5 ? ctrl->counts[i] : (i == 5 ? ctrl->n_riders : 0u);
    if (i < nb) { const int32_t l = d.shared_bld[i]; d.xa[XA_HEADER + i] = l >= 0 ? d.cnt_bld[l] : 0u; }
    if (i < nr) { const int32_t l = d.shared_room[i]; d.xa[XA_HEADER + nb + i] = l >= 0 ? d.cnt_room[l] : 0u; }
}

__global__ __launch_bounds__(TPB) void k_unpack_a(Dev d)
{
    Ctrl *ctrl = d.ctrl;
    const uint32_t i = blockIdx.x * TPB + threadIdx.x;
    const uint32_t nb = d.n_shared_bld, nr = d.n_shared_room;
    if (i < 5) ctrl->counts[i] = d.xa[i];
    if (i == 5) ctrl->n_riders = d.xa[5];
    if (i < nb) { const int32_t l = d.shared_bld[i]; if (l >= 0) d.cnt_bld[l] = d.xa[XA_HEADER + i]; }
    if (i < nr) { const int32_t l = d.shared_room[i]; if (l >= 0) d.cnt_room[l] = d.xa[XA_HEADER + nb + i]; }
}

__device__ __forceinline__ uint32_t vacc_candidate(const Dev &d, uint32_t i, uint32_t t)
{
    const philox_out o = philox4x32_10(i, t, ESIM_SLOT_VACCINE, 0u, d.seed_lo, d.seed_hi);
    const uint64_t x = ((uint64_t)o.w0 << 32) | o.w1;
    return (uint32_t)__umul64hi(x, (uint64_t)d.n_global);
}

// Liveness (eligible bit) of the first VACC_BATCH vaccination candidates, owner computes.
__global__ __launch_bounds__(TPB) void k_pack_b(Dev d)
{
    const Ctrl *ctrl = d.ctrl;
    const uint32_t i = blockIdx.x * TPB + threadIdx.x;
    if (i == 0) { d.xb[0] = ctrl->exp_bld; d.xb[1] = ctrl->exp_bus; d.xb[2] = ctrl->elig_count; d.xb[3] = ctrl->error; }
    if (i < VACC_BATCH) {
        const uint32_t j = vacc_candidate(d, i, ctrl->t);
        bool live = false;
        if (j >= d.id_base && j - d.id_base < d.n) live = d.state[j - d.id_base] & ST_ELIGIBLE;
        const unsigned long long m = __ballot(live);
        if ((threadIdx.x & 63u) == 0) { d.xb[XB_HEADER + (i >> 5)] = (uint32_t)m; d.xb[XB_HEADER + (i >> 5) + 1] = (uint32_t)(m >> 32); }
    }
}

// ---------------------------------------------------------------------------------- k_finish
// apply_interventions (simulator.rs:455-556): InterventionStatus::update_status
// (interventions.rs:110-184), the vaccination draw (simulator.rs:524-553), and the
// StatisticEntry of the step (statistics.rs:208-215, adjusted by citizen_exposed :275-287).
#define FIN_TPB 1024
__global__ __launch_bounds__(FIN_TPB) void k_finish(Dev d, int sharded)
{
    __shared__ uint32_t tab_key[VACC_TABLE];
    __shared__ uint32_t tab_idx[VACC_TABLE];
    __shared__ uint32_t wsum[FIN_TPB / 64];
    __shared__ uint32_t s_total;
    Ctrl *ctrl = d.ctrl;
    if (ctrl->finished) return;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    const uint32_t t = ctrl->t;
    const uint32_t total = ctrl->counts[0] + ctrl->counts[1] + ctrl->counts[2] + ctrl->counts[3] + ctrl->counts[4];
    const double x = (double)ctrl->counts[2] / (double)total;            // infected_percentage, statistics.rs:252
    const bool trig = !ctrl->vacc_active && d.thr_vacc < x;
    const bool have = ctrl->have_elig || trig;
    // sharded: totals over all shards come from exchange buffer B, the ctrl fields stay per-shard
    const uint32_t elig_count = sharded ? d.xb[2] : ctrl->elig_count;
    const uint32_t exp_bld = sharded ? d.xb[0] : ctrl->exp_bld;
    const uint32_t exp_bus = sharded ? d.xb[1] : ctrl->exp_bus;
    uint32_t vacc_now = 0;

    if (have) {
        if (elig_count <= d.vaccination_rate) {
            // choose_multiple hands back the whole set (simulator.rs:525-527)
            for (uint32_t c = tid; c < d.n; c += FIN_TPB) {
                const uint32_t st = d.state[c];
                if (st & ST_ELIGIBLE) d.state[c] = (uint16_t)((st & ~ST_TE_MASK) | TE_VACCINATED);
            }
            vacc_now = elig_count;
        } else {
            const uint32_t k = d.vaccination_rate;
            for (uint32_t i = tid; i < VACC_TABLE; i += FIN_TPB) { tab_key[i] = 0xFFFFFFFFu; tab_idx[i] = 0xFFFFFFFFu; }
            __syncthreads();
            uint32_t already = 0;
            for (uint32_t base = 0; already < k; base += VACC_BATCH) {
                uint32_t j[4], slot[4]; bool live[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const uint32_t i = base + tid * 4u + q;
                    j[q] = vacc_candidate(d, i, t);
                    if (sharded) live[q] = (d.xb[XB_HEADER + ((i - base) >> 5)] >> ((i - base) & 31u)) & 1u;
                    else live[q] = d.state[j[q]] & ST_ELIGIBLE;
                    slot[q] = 0;
                    if (live[q]) {
                        uint32_t sl = (j[q] * 2654435761u) >> 18;        // 14 bits
                        for (;;) {
                            const uint32_t old = atomicCAS(&tab_key[sl], 0xFFFFFFFFu, j[q]);
                            if (old == 0xFFFFFFFFu || old == j[q]) break;
                            sl = (sl + 1u) & (VACC_TABLE - 1u);
                        }
                        atomicMin(&tab_idx[sl], i);
                        slot[q] = sl;
                    }
                }
                __syncthreads();
                bool first[4]; uint32_t mine = 0;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    first[q] = live[q] && tab_idx[slot[q]] == base + tid * 4u + q;
                    mine += first[q];
                }
                // exclusive scan of `mine` in candidate order
                uint32_t incl = mine;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) { const uint32_t v = __shfl_up(incl, o, 64); if (lane >= (uint32_t)o) incl += v; }
                if (lane == 63) wsum[wv] = incl;
                __syncthreads();
                if (tid == 0) { uint32_t a = 0; for (uint32_t w = 0; w < FIN_TPB / 64; ++w) { const uint32_t v = wsum[w]; wsum[w] = a; a += v; } s_total = a; }
                __syncthreads();
                uint32_t pos = already + wsum[wv] + incl - mine;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (first[q]) {
                        if (pos < k) {
                            const uint32_t g = j[q];
                            if (g >= d.id_base && g - d.id_base < d.n) {
                                const uint32_t st = d.state[g - d.id_base];
                                d.state[g - d.id_base] = (uint16_t)((st & ~ST_TE_MASK) | TE_VACCINATED);  // unconditional, simulator.rs:551
                            }
                        }
                        pos++;
                    }
                }
                const uint32_t got = s_total;
                __syncthreads();
                already += got < k - already ? got : k - already;
                // every wave must reach an exit: one batch when sharded (liveness was exchanged for one),
                // a hard cap otherwise (an eligible fraction below ~1e-4 would need more candidates)
                if ((sharded || base >= (1u << 26)) && already < k) { if (tid == 0) ctrl->error = (uint32_t)(-ESIM_ERANGE); break; }
            }
            vacc_now = already;
        }
    }
    __syncthreads();
    if (tid == 0) {
        const uint32_t exps = exp_bld + exp_bus;
        if (sharded) ctrl->error |= d.xb[3];
        esim_step_result r;
        r.time_step = t;
        if (exps > ctrl->counts[0]) ctrl->error = (uint32_t)(-ESIM_ESIM);   // citizen_exposed underflow
        r.susceptible = ctrl->counts[0] - exps; r.exposed = ctrl->counts[1] + exps;
        r.infected = ctrl->counts[2]; r.recovered = ctrl->counts[3]; r.vaccinated = ctrl->counts[4];
        r.exposures_building = exp_bld; r.exposures_bus = exp_bus;
        // InterventionStatus::update_status, interventions.rs:110-184 (all comparisons strict)
        const uint32_t lockdown = d.thr_lockdown < x ? 1u : 0u;             // :116-128
        uint32_t mask = ctrl->mask;                                          // :142-180
        if (mask == ESIM_MASK_NONE) { if (d.thr_mask_pt < x) mask = ESIM_MASK_PUBLIC_TRANSPORT; }
        else if (mask == ESIM_MASK_PUBLIC_TRANSPORT) {
            if (x < d.thr_mask_pt) mask = ESIM_MASK_NONE;
            else if (d.thr_mask_all < x) mask = ESIM_MASK_EVERYWHERE;
        } else if (x < d.thr_mask_all) mask = ESIM_MASK_PUBLIC_TRANSPORT;
        // the direction everyone on a bus travels in, citizen.rs:179-204 with the lockdown used by THIS step
        if (!ctrl->lockdown) {
            const uint32_t h = t % 24u;
            if (h == d.start_hour - 1u) ctrl->bus_dir = 1u;
            else if (h == d.start_hour) ctrl->bus_dir = 0u;
            else if (h == d.end_hour - 1u) ctrl->bus_dir = 2u;
            else ctrl->bus_dir = 0u;
        }
        ctrl->lockdown = lockdown; ctrl->mask = mask;
        if (trig) { ctrl->vacc_active = 1u; ctrl->have_elig = 1u; }
        r.lockdown = lockdown; r.vaccination_active = ctrl->vacc_active; r.mask_status = mask;
        r.n_riders = ctrl->n_riders; r.vaccinated_now = vacc_now; r.eligible_count = have ? elig_count : 0u;
        r.disease_exists = (r.exposed != 0u || r.infected != 0u || r.susceptible != 0u) ? 1u : 0u;   // statistics.rs:289-291
        r.reserved = 0u;
        if (t <= d.max_steps) d.records[t] = r;
        ctrl->steps_done = t;
        if (!r.disease_exists && ctrl->stop_when_done) ctrl->finished = 1u;
        for (int i = 0; i < 5; ++i) ctrl->counts[i] = 0u;
        ctrl->n_riders = 0u; ctrl->exp_bld = 0u; ctrl->exp_bus = 0u;
        ctrl->t = t + 1u;
    }
}

// Reference-shaped view of the state word (esim_download_state).
__global__ __launch_bounds__(TPB) void k_decode_state(Dev d, uint8_t *status, uint16_t *timer, uint32_t *cur,
                                                      uint8_t *on_bus, uint8_t *eligible)
{
    const Ctrl *ctrl = d.ctrl;
    const uint32_t t = ctrl->t - 1u;             // last completed step
    for (uint32_t c = blockIdx.x * TPB + threadIdx.x; c < d.n; c += gridDim.x * TPB) {
        const uint32_t st = d.state[c], te = st & ST_TE_MASK, fl = d.flags[c];
        const uint32_t cls = status_of(te, t, d.exposed_time, d.infected_time);
        uint32_t tm = 0;
        if (te < TE_RECOVERED) {
            const uint32_t dd = t + TE_BIAS - te;
            if (cls == ESIM_EXPOSED) tm = dd; else if (cls == ESIM_INFECTED) tm = dd - d.exposed_time - 1u;
        }
        if (status) status[c] = (uint8_t)cls;
        if (timer) timer[c] = (uint16_t)tm;
        if (cur) cur[c] = ((st & ST_AT_WORK) && (fl & FL_HAS_WORK)) ? d.work[c] : d.home[c];
        if (on_bus) on_bus[c] = (st & ST_ON_BUS) ? (uint8_t)ctrl->bus_dir : 0;
        if (eligible) eligible[c] = (st & ST_ELIGIBLE) ? 1 : 0;
    }
}
