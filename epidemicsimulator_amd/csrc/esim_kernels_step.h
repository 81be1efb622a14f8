// esim_kernels_step.h -- the sequential forms of a time step: k_infected -> k_expose -> k_finish (the only form that can
// vaccinate), their one-launch fusion k_pipe for chunks that cannot be drawn in one pass, the persistent single-workgroup
// k_small, and the pack / unpack kernels of the coupled multi-GPU exchange.
#pragma once
// ---------------------------------------------------------------------------------- k_infected
// The Infected citizens of step t are the log slice with exposure step in
// [t - exposed_time - 1 - infected_time, t - exposed_time - 1].  simulator.rs:181-198: a rider
// joins its route's session, anybody else marks current_building_position.
__device__ __forceinline__ void infected_phase(const Dev &d, Ctrl *ctrl, const StepEnv &env, uint32_t vb, uint32_t nvb)
{
    const uint32_t t = env.t, at_work = env.at_work, bus_dir = env.bus_dir;
    const int hi = (int)(t + TE_BIAS) - (int)d.exposed_time - 1;
    const int lo = hi - (int)d.infected_time;
    if (hi < 0) return;
    const uint32_t p = t & (MARK_SLOTS - 1u);
    uint32_t *cnt_bld = d.cnt_bld[p], *cnt_room = d.cnt_room[p], *route_flag = d.route_flag[p];
    const uint32_t i0 = d.log_off[lo < 0 ? 0 : lo], i1 = d.log_off[hi + 1];
    for (uint32_t i = i0 + vb * blockDim.x + threadIdx.x; i < i1; i += nvb * blockDim.x) {
        const uint32_t c = d.log[i];
        const uint32_t fl = d.cit[c];
        if (status_of(CW_TE(fl), t, d.exposed_time, d.infected_time) != ESIM_INFECTED) continue;  // vaccinated since (Q10)
        if (bus_dir && (fl & FL_USES_PT)) {                                  // simulator.rs:181-186
            const uint32_t r = d.route_of[c];
            if (atomicExch(&route_flag[r], 1u) == 0u) {
                if (d.route_off[r + 1] - d.route_off[r] <= 64u) append(d.touched_route[p], &ctrl->n_touched_route[p], r);
                else append(d.touched_route_big[p], &ctrl->n_touched_route_big[p], r);
            }
        } else {                                                             // :187-198
            const bool atw = at_work && (fl & FL_HAS_WORK);
            const uint32_t b = atw ? d.work[c] : d.home[c];
            if (atomicAdd(&cnt_bld[b], 1u) == 0u) append(d.touched_bld[p], &ctrl->n_touched_bld[p], b);
            if (atw && (fl & FL_WORK_SCHOOL)) {
                const uint32_t r = d.room[c];
                if (atomicAdd(&cnt_room[r], 1u) == 0u) append(d.touched_room[p], &ctrl->n_touched_room[p], r);
            }
        }
    }
}

__global__ __launch_bounds__(TPB) void k_infected(Dev d)
{
    if (d.ctrl->finished) return;
    infected_phase(d, d.ctrl, env_from_ctrl(d, d.ctrl), blockIdx.x, gridDim.x);
}

// Threshold for Citizen::expose (citizen.rs:221-248): row 1 of the LUT is p - p*mask_effectiveness,
// which only applies to NON-compliant citizens while the global status is Everywhere (Q7).
__device__ __forceinline__ uint64_t threshold(const Dev &d, uint32_t fl, uint32_t mask, uint32_t n)
{
    const uint32_t row = (!(fl & FL_MASK_COMPLIANT) && mask == ESIM_MASK_EVERYWHERE) ? 1u : 0u;
    return d.thr[row * 256u + (n & 255u)];                              // `as u8`, citizen.rs:239
}

// Susceptible -> Exposed(0) (citizen.rs:244) exactly once per citizen even when several member
// lists reach the same citizen concurrently: CAS on the citizen word.  new_te_bits = te (and bus bit) part.
__device__ __forceinline__ bool expose_once(const Dev &d, Ctrl *ctrl, uint32_t m, uint32_t new_te_bits)
{
    uint32_t *w = d.cit + m;
    uint32_t old = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (;;) {
        if (CW_TE(old) != TE_SUSCEPTIBLE) return false;
        const uint32_t prev = atomicCAS(w, old, new_te_bits | (old & CW_FLAGS));
        if (prev == old) break;
        old = prev;
    }
    append(d.log, &ctrl->log_len, m);
    return true;
}

// All building draws of one susceptible citizen in step t seen from the candidate's side
// (simulator.rs:308-350): the home list (building.rs:202), then the work list (building.rs:278) or
// the school-room multiset (building.rs:494-522).  Pure function of the infected counts; the bus
// phase uses it to know whether the building phase exposes a rider (simulator.rs:436 only reaches
// riders that are still Susceptible after the buildings).
__device__ __forceinline__ bool building_draws(const Dev &d, uint32_t c, uint32_t fl, uint32_t t, uint32_t mask,
                                               uint32_t at_work)
{
    const uint32_t g = d.id_base + c;
    const uint64_t seed = ((uint64_t)d.seed_hi << 32) | d.seed_lo;
    const bool atw = at_work && (fl & FL_HAS_WORK);
    const bool same = fl & FL_SAME_AREA;
    const uint32_t *cnt_bld = d.cnt_bld[t & (MARK_SLOTS - 1u)], *cnt_room = d.cnt_room[t & (MARK_SLOTS - 1u)];
    // "If the Citizen is not currently in the Area, they haven't been exposed!" simulator.rs:324
    if (!atw || same) {
        const uint32_t n = cnt_bld[d.home[c]];
        if (n && esim_u32(seed, g, t, ESIM_SLOT_HOME) < threshold(d, fl, mask, n)) return true;
    }
    if ((fl & FL_HAS_WORK) && (at_work || same)) {
        const uint32_t n = cnt_bld[d.work[c]];
        if (n) {
            const uint64_t thr = threshold(d, fl, mask, n);
            if (fl & FL_WORK_SCHOOL) {
                const uint32_t k = cnt_room[d.room[c]];                 // one copy of the room per infected
                for (uint32_t j = 0; j < k; ++j)
                    if (esim_u32(seed, g, t, ESIM_SLOT_ROOM0 + j) < thr) return true;
            } else if (esim_u32(seed, g, t, ESIM_SLOT_WORK) < thr) return true;
        }
    }
    return false;
}

// One candidate of one member list (simulator.rs:308-350 for one citizen_id of find_exposures), given its
// state and flags.  kind 0: resident (home list), 1: worker (work list), 2: room participant (k draws).
__device__ __forceinline__ void member_eval(const Dev &d, Ctrl *ctrl, uint32_t m, uint32_t fl, uint32_t kind,
                                            uint32_t n, uint32_t k, uint32_t t, uint32_t mask, uint32_t at_work, uint32_t &n_exp)
{
    if (CW_TE(fl) != TE_SUSCEPTIBLE) return;                             // is_susceptible(), simulator.rs:337
    const bool same = fl & FL_SAME_AREA;
    // area of current_building_position == area of this building?  simulator.rs:324
    if (kind == 0u) { if (at_work && (fl & FL_HAS_WORK) && !same) return; }
    else if (!at_work && !same) return;
    const uint64_t thr = threshold(d, fl, mask, n);
    const uint64_t seed = ((uint64_t)d.seed_hi << 32) | d.seed_lo;
    const uint32_t g = d.id_base + m;
    bool hit = false;
    if (kind == 2u) { for (uint32_t j = 0; j < k && !hit; ++j) hit = esim_u32(seed, g, t, ESIM_SLOT_ROOM0 + j) < thr; }
    else hit = esim_u32(seed, g, t, kind == 0u ? ESIM_SLOT_HOME : ESIM_SLOT_WORK) < thr;
    if (hit && expose_once(d, ctrl, m, CW_MAKE(t + TE_BIAS, 0u))) n_exp++;   // Exposed(0), citizen.rs:244
}

// Members [lo, hi) of one list, walked by a group of 8 lanes (gl = lane in group): every lane takes up to four
// members per pass and issues their index / state / flag loads together, so a 30-member room costs one
// dependent chain instead of four.
__device__ __forceinline__ void member_list(const Dev &d, Ctrl *ctrl, const uint32_t *idx, uint32_t lo, uint32_t hi, uint32_t gl,
                                            uint32_t kind, uint32_t n, uint32_t k, uint32_t t, uint32_t mask, uint32_t at_work,
                                            uint32_t &n_exp)
{
    for (uint32_t base = lo + gl; base < hi; base += 32u) {
        uint32_t m[4], fl[4];
        bool ok[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const uint32_t q = base + 8u * u; ok[u] = q < hi; m[u] = ok[u] ? (idx ? idx[q] : q) : 0u; }
#pragma unroll
        for (int u = 0; u < 4; ++u) fl[u] = ok[u] ? d.cit[m[u]] : 0u;
#pragma unroll
        for (int u = 0; u < 4; ++u) if (ok[u]) member_eval(d, ctrl, m[u], fl[u], kind, n, k, t, mask, at_work, n_exp);
    }
}

// A rider that the building phase leaves Susceptible draws once with the number of infected
// riders on the same bus (expose_citizens, simulator.rs:407-453).
__device__ __forceinline__ void bus_draw(const Dev &d, Ctrl *ctrl, uint32_t c, uint32_t k, uint32_t t, uint32_t mask,
                                         uint32_t at_work)
{
    const uint32_t fl = d.cit[c];
    if (building_draws(d, c, fl, t, mask, at_work)) return;              // the buildings got there first
    const uint64_t seed = ((uint64_t)d.seed_hi << 32) | d.seed_lo;
    if (esim_u32(seed, d.id_base + c, t, ESIM_SLOT_BUS) < threshold(d, fl, mask, k)) {
        if (expose_once(d, ctrl, c, CW_MAKE(t + TE_BIAS, CW_BUS_EXPOSED))) {
            atomicAdd(&d.exp_step[2u * t + 1u], 1u);
            if (ctrl->have_elig) atomicSub(&ctrl->elig_count, 1u);       // simulator.rs:447-449 (a Susceptible is eligible)
        }
    }
}

// ---------------------------------------------------------------------------------- k_expose
// apply_exposures.  Work items: marked buildings (residents + workers), marked school rooms,
// marked routes of <= 64 riders -- one wavefront each, lanes over the members; then marked
// routes of > 64 riders, one workgroup each.
__device__ __forceinline__ void expose_phase(const Dev &d, Ctrl *ctrl, const StepEnv &env, uint32_t vb, uint32_t nvb)
{
    const uint32_t t = env.t, mask = env.mask, at_work = env.at_work;
    const uint32_t p = t & (MARK_SLOTS - 1u), q = (t + MARK_SLOTS - 1u) & (MARK_SLOTS - 1u);
    const uint32_t *cnt_bld = d.cnt_bld[p], *cnt_room = d.cnt_room[p];
    const uint32_t nb = ld(&ctrl->n_touched_bld[p]), nr = ld(&ctrl->n_touched_room[p]), nrt = ld(&ctrl->n_touched_route[p]);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (vb * blockDim.x + threadIdx.x) >> 6, n_waves = (nvb * blockDim.x) >> 6;
    uint32_t n_exp = 0;
    // (0) forget the marks of the previous step (other parity); nobody reads them any more
    {
        const uint32_t tid = vb * blockDim.x + threadIdx.x, nth = nvb * blockDim.x;
        const uint32_t ob = ctrl->n_touched_bld[q], orr = ctrl->n_touched_room[q], ort = ctrl->n_touched_route[q], orb = ctrl->n_touched_route_big[q];
        for (uint32_t i = tid; i < ob; i += nth) d.cnt_bld[q][d.touched_bld[q][i]] = 0u;
        for (uint32_t i = tid; i < orr; i += nth) d.cnt_room[q][d.touched_room[q][i]] = 0u;
        for (uint32_t i = tid; i < ort; i += nth) d.route_flag[q][d.touched_route[q][i]] = 0u;
        for (uint32_t i = tid; i < orb; i += nth) d.route_flag[q][d.touched_route_big[q][i]] = 0u;
    }
    // (1) marked buildings and school rooms: groups of 8 lanes per item, 8 items per wavefront pass
    const uint32_t grp = lane >> 3, gl = lane & 7u;
    for (uint32_t base = wave * 8u; base < nb + nr; base += n_waves * 8u) {
        const uint32_t it = base + grp;
        if (it >= nb + nr) continue;
        if (it < nb) {
            const uint32_t b = d.touched_bld[p][it];
            if (d.bld_type[b] == ESIM_SCHOOL) continue;                  // School::find_exposures works per room
            const uint32_t n = cnt_bld[b];                               // exposure_count, simulator.rs:307
            // Household / Workplace::find_exposures: every registered occupant (building.rs:202-204,278-280)
            member_list(d, ctrl, d.res_idx, d.res_off[b], d.res_off[b + 1], gl, 0u, n, 0u, t, mask, at_work, n_exp);
            member_list(d, ctrl, d.wrk_idx, d.wrk_off[b], d.wrk_off[b + 1], gl, 1u, n, 0u, t, mask, at_work, n_exp);
        } else {
            const uint32_t r = d.touched_room[p][it - nb];
            const uint32_t k = cnt_room[r];                              // one copy of the room per infected in it
            const uint32_t n = cnt_bld[d.room_bld[r]];                   // infected in the whole school
            member_list(d, ctrl, d.room_idx, d.room_off[r], d.room_off[r + 1], gl, 2u, n, k, t, mask, at_work, n_exp);
        }
    }
    // (2) marked routes of <= 64 riders, one wavefront each: rank by (Philox key, id) with shuffles; buses are
    // consecutive runs of bus_capacity ranks (replaces shuffle + pop, simulator.rs:362-388)
    for (uint32_t it = wave; it < nrt; it += n_waves) {
        const uint32_t r = d.touched_route[p][it];
        const uint32_t off = d.route_off[r], s = d.route_off[r + 1] - off;
        uint32_t c = 0, st = 0, key = 0;
        bool inf = false;
        if (lane < s) {
            c = d.route_riders[off + lane];
            st = d.cit[c];
            inf = status_of(CW_TE(st), t, d.exposed_time, d.infected_time) == ESIM_INFECTED;
            key = philox4x32_10(d.id_base + c, t, ESIM_SLOT_BUS_ORDER, 0u, d.seed_lo, d.seed_hi).w0;
        }
        uint32_t rank = 0;
        for (uint32_t j = 0; j < s; ++j) {
            const uint32_t kj = __shfl(key, j, 64);
            rank += kj < key || (kj == key && j < lane);                 // ids ascend with the lane
        }
        const uint32_t bus = rank / d.bus_capacity;
        uint32_t k = 0;
        for (uint32_t j = 0; j < s; ++j) {
            const uint32_t bj = __shfl(bus, j, 64);
            const bool ij = __shfl((int)inf, j, 64);
            k += ij && bj == bus;
        }
        if (lane < s && k && CW_TE(st) == TE_SUSCEPTIBLE) bus_draw(d, ctrl, c, k, t, mask, at_work);
    }
    // routes of > 64 riders (rare: a very large Output Area): rank by counting through global scratch
    const uint32_t nbig = ld(&ctrl->n_touched_route_big[p]);
    for (uint32_t ri = vb; ri < nbig; ri += nvb) {
        const uint32_t r = d.touched_route_big[p][ri];
        const uint32_t off = d.route_off[r], s = d.route_off[r + 1] - off;
        for (uint32_t i = threadIdx.x; i < s; i += blockDim.x) {
            const uint32_t c = d.route_riders[off + i];
            const bool inf = status_of(CW_TE(d.cit[c]), t, d.exposed_time, d.infected_time) == ESIM_INFECTED;
            d.bus_key[off + i] = philox4x32_10(d.id_base + c, t, ESIM_SLOT_BUS_ORDER, 0u, d.seed_lo, d.seed_hi).w0;
            d.bus_flag[off + i] = inf ? 1u : 0u;
            d.bus_cnt[off + i] = 0u;
        }
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < s; i += blockDim.x) {
            const uint32_t key = d.bus_key[off + i];
            uint32_t rank = 0;
            for (uint32_t j = 0; j < s; ++j) {
                const uint32_t kj = d.bus_key[off + j];
                rank += kj < key || (kj == key && j < i);
            }
            const uint32_t bus = rank / d.bus_capacity;
            d.bus_idx[off + i] = bus;
            if (d.bus_flag[off + i]) atomicAdd(&d.bus_cnt[off + bus], 1u);
        }
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < s; i += blockDim.x) {
            const uint32_t c = d.route_riders[off + i];
            const uint32_t k = __hip_atomic_load(&d.bus_cnt[off + d.bus_idx[off + i]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (k && CW_TE(d.cit[c]) == TE_SUSCEPTIBLE) bus_draw(d, ctrl, c, k, t, mask, at_work);
        }
        __syncthreads();
    }
    if (n_exp) atomicAdd(&d.exp_step[2u * t], n_exp);
}

__global__ __launch_bounds__(TPB) void k_expose(Dev d)
{
    if (d.ctrl->finished) return;
    expose_phase(d, d.ctrl, env_from_ctrl(d, d.ctrl), blockIdx.x, gridDim.x);
}

// ---------------------------------------------------------------------------------- exchange
// Sharded runs: pack the census and the infected counts of shared buildings/rooms, let the caller
// SUM-all-reduce, and scatter the totals back (marking what remote infected citizens touched).
__global__ __launch_bounds__(TPB) void k_pack_a(Dev d)
{
    __shared__ uint32_t cen[5];
    const Ctrl *ctrl = d.ctrl;
    const uint32_t i = blockIdx.x * TPB + threadIdx.x;
    const uint32_t nb = d.n_shared_bld, nr = d.n_shared_room;
    if (blockIdx.x == 0) {
        census_block(d, ctrl, ctrl->t, cen);
        uint32_t at_work, bus_dir;
        schedule(d, ctrl, ctrl->t, at_work, bus_dir);
        if (threadIdx.x < XA_HEADER) d.xa[threadIdx.x] = threadIdx.x < 5 ? cen[threadIdx.x] : (threadIdx.x == 5 && bus_dir ? d.n_pt : 0u);
    }
    const uint32_t p = ctrl->t & (MARK_SLOTS - 1u);
    if (i < nb) { const int32_t l = d.shared_bld[i]; d.xa[XA_HEADER + i] = l >= 0 ? d.cnt_bld[p][l] : 0u; }
    if (i < nr) { const int32_t l = d.shared_room[i]; d.xa[XA_HEADER + nb + i] = l >= 0 ? d.cnt_room[p][l] : 0u; }
}

__global__ __launch_bounds__(TPB) void k_unpack_a(Dev d)
{
    Ctrl *ctrl = d.ctrl;
    const uint32_t i = blockIdx.x * TPB + threadIdx.x;
    const uint32_t nb = d.n_shared_bld, nr = d.n_shared_room;
    const uint32_t p = ctrl->t & (MARK_SLOTS - 1u);
    if (i < 5) ctrl->counts[i] = d.xa[i];
    if (i == 5) ctrl->n_riders = d.xa[5];
    if (i < nb) {
        const int32_t l = d.shared_bld[i];
        const uint32_t tot = d.xa[XA_HEADER + i];
        if (l >= 0 && tot) { if (d.cnt_bld[p][l] == 0u) append(d.touched_bld[p], &ctrl->n_touched_bld[p], (uint32_t)l); d.cnt_bld[p][l] = tot; }
    }
    if (i < nr) {
        const int32_t l = d.shared_room[i];
        const uint32_t tot = d.xa[XA_HEADER + nb + i];
        if (l >= 0 && tot) { if (d.cnt_room[p][l] == 0u) append(d.touched_room[p], &ctrl->n_touched_room[p], (uint32_t)l); d.cnt_room[p][l] = tot; }
    }
}

__device__ __forceinline__ uint32_t vacc_candidate(const Dev &d, uint32_t i, uint32_t t)
{
    const philox_out o = philox4x32_10(i, t, ESIM_SLOT_VACCINE, 0u, d.seed_lo, d.seed_hi);
    const uint64_t x = ((uint64_t)o.w0 << 32) | o.w1;
    return (uint32_t)__umul64hi(x, (uint64_t)d.n_global);
}

// Member of citizens_eligible_for_vaccine (simulator.rs:97)?  The set is "Susceptible at the end of
// the trigger step" (simulator.rs:487-513) minus later bus exposures (:447-449); building exposures
// and vaccination never remove anybody (Q10).  All of that is recoverable from the state word.
__device__ __forceinline__ bool eligible(uint32_t st, uint32_t trigger_step)
{
    const uint32_t te = CW_TE(st);
    if (te == TE_SUSCEPTIBLE || te == TE_VACCINATED) return true;        // only eligible citizens are ever vaccinated
    if (te >= TE_RECOVERED) return false;
    return te > trigger_step + TE_BIAS && !(st & CW_BUS_EXPOSED);
}

// Liveness of the first VACC_BATCH vaccination candidates, owner computes (sharded runs).
__global__ __launch_bounds__(TPB) void k_pack_b(Dev d)
{
    const Ctrl *ctrl = d.ctrl;
    const uint32_t i = blockIdx.x * TPB + threadIdx.x;
    const uint32_t t = ctrl->t;
    // the programme may start in this very step: same test as k_finish
    const uint32_t total = ctrl->counts[0] + ctrl->counts[1] + ctrl->counts[2] + ctrl->counts[3] + ctrl->counts[4];
    const bool trig = !ctrl->vacc_active && d.thr_vacc < (double)ctrl->counts[2] / (double)total;
    const uint32_t tstep = trig ? t : ctrl->trigger_step;
    if (i == 0) {
        const uint32_t eb = d.exp_step[2u * t], eu = d.exp_step[2u * t + 1u];
        d.xb[0] = eb; d.xb[1] = eu;
        d.xb[2] = trig ? ctrl->n_susceptible - eb - eu : ctrl->elig_count;
        d.xb[3] = ERR_FIELD(ctrl->error);                                   // (summed over the shards: a field per code)
    }
    if (i < VACC_WINDOW) {
        bool live = false;
        if (ctrl->have_elig || trig) {
            const uint32_t j = vacc_candidate(d, i, t);
            if (j >= d.id_base && j - d.id_base < d.n) live = eligible(d.cit[j - d.id_base], tstep);
        }
        const unsigned long long m = __ballot(live);
        if ((threadIdx.x & 63u) == 0) { d.xb[XB_HEADER + (i >> 5)] = (uint32_t)m; d.xb[XB_HEADER + (i >> 5) + 1] = (uint32_t)(m >> 32); }
    }
}

__global__ __launch_bounds__(TPB) void k_infected_dec(Dev d, uint32_t t, uint32_t j)
{
    infected_phase(d, d.ctrl, env_from_dec(d, t, j), blockIdx.x, gridDim.x);
}

// ------------------------------------------------------------------------------------- k_pipe
// One pipelined step: workgroups [0, n_expose) draw the exposures of step t (and clear the marks of step
// t-1), the others mark for step t+1.  The two halves touch different ring slots and different state:
// marks of t+1 depend on citizens exposed >= exposed_time + 1 steps ago, never on step t's exposures.
__global__ __launch_bounds__(TPB) void k_pipe(Dev d, uint32_t t, uint32_t j, uint32_t n_expose, int mark_next)
{
    Ctrl *ctrl = d.ctrl;
    if (blockIdx.x < n_expose) {
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            // slot (t+2) was cleared by the previous k_pipe; its lists restart empty for the k_pipe after this one
            const uint32_t z = (t + 2u) & (MARK_SLOTS - 1u);
            ctrl->n_touched_bld[z] = 0u; ctrl->n_touched_room[z] = 0u; ctrl->n_touched_route[z] = 0u; ctrl->n_touched_route_big[z] = 0u;
        }
        expose_phase(d, ctrl, env_from_dec(d, t, j), blockIdx.x, n_expose);
    } else if (mark_next) {
        infected_phase(d, ctrl, env_from_dec(d, t + 1u, j + 1u), blockIdx.x - n_expose, gridDim.x - n_expose);
    }
}

// Vaccination bookkeeping for one citizen set to Vaccinated (simulator.rs:551).
__device__ __forceinline__ void vaccinate(const Dev &d, Ctrl *ctrl, uint32_t c)
{
    const uint32_t st = d.cit[c], te = CW_TE(st);
    if (te == TE_VACCINATED) return;                                     // chosen again: ids are never removed (Q10)
    if (te == TE_SUSCEPTIBLE) atomicSub(&ctrl->n_susceptible, 1u);
    else if (te == TE_RECOVERED) atomicSub(&ctrl->n_recovered_sentinel, 1u);
    else atomicSub(&d.hist[te], 1u);                                     // an Exposed/Infected/Recovered citizen is relabelled
    atomicAdd(&ctrl->n_vaccinated, 1u);
    d.cit[c] = CW_MAKE(TE_VACCINATED, st & (CW_BUS_EXPOSED | CW_FLAGS));
}

// ---------------------------------------------------------------------------------- k_finish
// apply_interventions (simulator.rs:455-556): InterventionStatus::update_status
// (interventions.rs:110-184), the vaccination draw (simulator.rs:524-553), the StatisticEntry of the
// step (statistics.rs:208-215, adjusted by citizen_exposed :275-287), and the hand-over to step t+1.
struct FinishShared {
    uint32_t tab_key[VACC_TABLE];
    uint32_t tab_idx[VACC_TABLE];
    uint32_t wsum[FIN_TPB / 64];
    uint32_t s_total;
    uint32_t cen[5];
};

// Called by one whole workgroup of FIN_TPB threads.
__device__ __forceinline__ void finish_phase(const Dev &d, Ctrl *ctrl, int sharded, FinishShared &sm)
{
    uint32_t (&tab_key)[VACC_TABLE] = sm.tab_key;
    uint32_t (&tab_idx)[VACC_TABLE] = sm.tab_idx;
    uint32_t (&wsum)[FIN_TPB / 64] = sm.wsum;
    uint32_t &s_total = sm.s_total;
    uint32_t (&cen)[5] = sm.cen;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    const uint32_t t = ctrl->t;
    uint32_t at_work, bus_dir;
    schedule(d, ctrl, t, at_work, bus_dir);
    // sharded: 0 one shard (local census is the census); 1 coupled shards (global census from exchange A)
    if (sharded) { if (tid < 5) cen[tid] = ctrl->counts[tid]; __syncthreads(); }
    else census_block(d, ctrl, t, cen);
    const uint32_t total = cen[0] + cen[1] + cen[2] + cen[3] + cen[4];
    const double x = (double)cen[2] / (double)total;                     // infected_percentage, statistics.rs:252
    const bool trig = !ctrl->vacc_active && d.thr_vacc < x;
    const bool have = ctrl->have_elig || trig;
    const uint32_t tstep = trig ? t : ctrl->trigger_step;
    // totals over all shards come from exchange buffer B when sharded, the ctrl fields stay per-shard
    const uint32_t my_exp_bld = ld(&d.exp_step[2u * t]), my_exp_bus = ld(&d.exp_step[2u * t + 1u]);
    const uint32_t exp_bld = sharded ? d.xb[0] : my_exp_bld;
    const uint32_t exp_bus = sharded ? d.xb[1] : my_exp_bus;
    const uint32_t local_elig = trig ? ld(&ctrl->n_susceptible) - my_exp_bld - my_exp_bus : ld(&ctrl->elig_count);
    const uint32_t elig_count = sharded ? d.xb[2] : local_elig;
    const uint32_t n_riders = sharded ? ctrl->n_riders : (bus_dir ? d.n_pt : 0u);
    __syncthreads();
    // this step's exposures enter the books before anybody is vaccinated
    if (tid == 0) {
        const uint32_t mine = my_exp_bld + my_exp_bus;
        atomicSub(&ctrl->n_susceptible, mine);
        atomicAdd(&d.hist[t + TE_BIAS], mine);
        d.log_off[t + TE_BIAS + 1u] = ld(&ctrl->log_len);
        if (trig) ctrl->elig_count = local_elig;
    }
    __syncthreads();
    uint32_t vacc_now = 0;
    if (have) {
        if (elig_count <= d.vaccination_rate) {
            // choose_multiple hands back the whole set (simulator.rs:525-527)
            for (uint32_t c = tid; c < d.n; c += FIN_TPB)
                if (eligible(d.cit[c], tstep)) vaccinate(d, ctrl, c);
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
                    if (sharded) live[q] = i < VACC_WINDOW ? ((d.xb[XB_HEADER + (i >> 5)] >> (i & 31u)) & 1u) != 0u : false;
                    else live[q] = eligible(d.cit[j[q]], tstep);
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
                            if (g >= d.id_base && g - d.id_base < d.n) vaccinate(d, ctrl, g - d.id_base);   // unconditional, simulator.rs:551
                        }
                        pos++;
                    }
                }
                const uint32_t got = s_total;
                __syncthreads();
                already += got < k - already ? got : k - already;
                // every wave must reach an exit: the window whose liveness was exchanged when sharded,
                // a hard cap otherwise (an eligible fraction below ~1e-4 would need more candidates)
                if (((sharded && base + VACC_BATCH >= VACC_WINDOW) || base >= (1u << 26)) && already < k) { if (tid == 0) ctrl->error = (uint32_t)(-ESIM_ERANGE); break; }
            }
            vacc_now = already;
        }
    }
    __syncthreads();
    if (tid == 0) {
        const uint32_t exps = exp_bld + exp_bus;
        if (sharded && d.xb[3]) { ctrl->peer_error |= d.xb[3]; if (!ctrl->error) ctrl->error = err_decode(d.xb[3]); }   // any shard's error is everybody's
        esim_step_result r;
        r.time_step = t;
        if (exps > cen[0]) ctrl->error = (uint32_t)(-ESIM_ESIM);          // citizen_exposed underflow, statistics.rs:275-287
        r.susceptible = cen[0] - exps; r.exposed = cen[1] + exps;
        r.infected = cen[2]; r.recovered = cen[3]; r.vaccinated = cen[4];
        r.exposures_building = exp_bld; r.exposures_bus = exp_bus;
        // InterventionStatus::update_status, interventions.rs:110-184 (all comparisons strict)
        const uint32_t lockdown = d.thr_lockdown < x ? 1u : 0u;             // :116-128
        uint32_t mask = ctrl->mask;                                          // :142-180
        if (mask == ESIM_MASK_NONE) { if (d.thr_mask_pt < x) mask = ESIM_MASK_PUBLIC_TRANSPORT; }
        else if (mask == ESIM_MASK_PUBLIC_TRANSPORT) {
            if (x < d.thr_mask_pt) mask = ESIM_MASK_NONE;
            else if (d.thr_mask_all < x) mask = ESIM_MASK_EVERYWHERE;
        } else if (x < d.thr_mask_all) mask = ESIM_MASK_PUBLIC_TRANSPORT;
        ctrl->at_work = at_work; ctrl->bus_dir = bus_dir;
        ctrl->lockdown = lockdown; ctrl->mask = mask;
        if (trig) { ctrl->vacc_active = 1u; ctrl->have_elig = 1u; ctrl->trigger_step = t; }
        r.lockdown = lockdown; r.vaccination_active = ctrl->vacc_active; r.mask_status = mask;
        r.n_riders = n_riders; r.vaccinated_now = vacc_now; r.eligible_count = have ? elig_count : 0u;
        r.disease_exists = (r.exposed != 0u || r.infected != 0u || r.susceptible != 0u) ? 1u : 0u;   // statistics.rs:289-291
        r.reserved = 0u;
        if (t <= d.max_steps) d.records[t] = r;
        ctrl->steps_done = t;
        if (!r.disease_exists && ctrl->stop_when_done) ctrl->finished = 1u;
        // the marks of step t-1 were cleared by this step's exposure pass: that slot's lists are free again
        const uint32_t q = (t + MARK_SLOTS - 1u) & (MARK_SLOTS - 1u);
        ctrl->n_touched_bld[q] = 0u; ctrl->n_touched_room[q] = 0u; ctrl->n_touched_route[q] = 0u; ctrl->n_touched_route_big[q] = 0u;
        ctrl->n_riders = 0u;
        for (int i = 0; i < 5; ++i) ctrl->counts[i] = 0u;
        ctrl->need_seq = 0u;                                                // (a cut chunk asked for this step in this form, k_chunk_vax)
        ctrl->t = t + 1u;
    }
    __syncthreads();
}

__global__ __launch_bounds__(FIN_TPB) void k_finish(Dev d, int sharded)
{
    __shared__ FinishShared sm;
    if (d.ctrl->finished) return;
    finish_phase(d, d.ctrl, sharded, sm);
}

// ------------------------------------------------------------------------------------- k_small
// While few citizens are Infected, a whole time step is a handful of dependent memory round trips and
// three kernel boundaries cost more than the work.  This persistent single-workgroup kernel runs the
// same three phases back to back (workgroup barriers instead of kernel boundaries) for up to `max_steps`
// steps, and returns as soon as a step's infected slice exceeds `small_max` (the multi-workgroup kernels
// take over), the run is finished, or the budget is used.  ctrl->small_done = steps it executed.
__global__ __launch_bounds__(FIN_TPB) void k_small(Dev d, uint32_t max_steps, uint32_t small_max, int mode)
{
    __shared__ FinishShared sm;
    __shared__ Ctrl sc;                                   // the control block lives in LDS for the whole launch
    __shared__ uint32_t go;
    if (threadIdx.x == 0) sc = *d.ctrl;
    __syncthreads();
    Ctrl *ctrl = &sc;
    uint32_t done = 0;
    for (; done < max_steps; ++done) {
        if (threadIdx.x == 0) {
            const uint32_t t = ctrl->t;
            const int hi = (int)(t + TE_BIAS) - (int)d.exposed_time - 1;
            const int lo = hi - (int)d.infected_time;
            const uint32_t len = d.log_off[hi + 1] - d.log_off[lo < 0 ? 0 : lo];
            go = (!ctrl->finished && !ctrl->error && len <= small_max && t <= d.max_steps) ? 1u : 0u;
        }
        __syncthreads();
        if (!go) break;                                   // block-uniform: every wave leaves together
        const StepEnv env = env_from_ctrl(d, ctrl);
        infected_phase(d, ctrl, env, 0u, 1u);
        __syncthreads();
        expose_phase(d, ctrl, env, 0u, 1u);
        __syncthreads();
        finish_phase(d, ctrl, mode, sm);
    }
    __syncthreads();
    if (threadIdx.x == 0) { sc.small_done = done; *d.ctrl = sc; }
}

// Reference-shaped view of the state (esim_download_state).
// Exposure log entries [first, first + n): the citizen and whether it was exposed on public transport (the bus bit of its
// word outlives a later vaccination: vaccinate() keeps it).
