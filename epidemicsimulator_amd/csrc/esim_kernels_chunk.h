// esim_kernels_chunk.h -- chunks of up to 96 time steps: the census ahead and the decisions of a chunk (k_future, k_decide),
// the one-pass form (k_chunk_marks, k_chunk_draw, k_chunk_units, k_chunk_books [+ k_chunk_count, k_chunk_scatter]) and the
// books of a chunk that ran step by step (k_batch_finish).
#pragma once
// ------------------------------------------------------------------------------- chunk set-up
// A citizen exposed in step t is Infected no earlier than step t + exposed_time + 1 (disease.rs:47-71), so the
// Infected census of the next <= exposed_time + 1 steps is already fixed -- as long as nobody is vaccinated.
// k_future writes that vector for this shard (sharded runs SUM-all-reduce it); k_decide then runs the
// intervention state machine (interventions.rs:110-184 needs nothing but the infected fraction) and the
// schedule (citizen.rs:176-206) over the chunk and stops in front of the step that would start vaccinating.
// Inclusive prefix sum over BF_WIN values in shared memory, by a workgroup of FIN_TPB = BF_WIN threads.
#define BF_WIN 1024
__device__ __forceinline__ void block_scan_1024(uint32_t *v, uint32_t *wtmp)
{
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    uint32_t x = v[tid];
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(x, o, 64); if (lane >= (uint32_t)o) x += y; }
    if (lane == 63u) wtmp[wv] = x;
    __syncthreads();
    if (tid == 0) { uint32_t a = 0; for (uint32_t w = 0; w < FIN_TPB / 64; ++w) { const uint32_t y = wtmp[w]; wtmp[w] = a; a += y; } }
    __syncthreads();
    v[tid] = x + wtmp[wv];
    __syncthreads();
}

// (Control block, histogram and census vector are read past the caches: in k_chunk_books the same workgroup has just
// written them.)
__device__ __forceinline__ void future_body(const Dev &d, uint32_t max_ahead, uint32_t limit_t, uint32_t *win, uint32_t *wtmp)
{
    Ctrl *ctrl = d.ctrl;
    const uint32_t t0 = ld(&ctrl->t), tid = threadIdx.x;                   // t0: first step of the chunk
    const uint32_t n_ahead = t0 > limit_t ? 0u : (limit_t - t0 + 1u < max_ahead ? limit_t - t0 + 1u : max_ahead);
    const int et = (int)d.exposed_time, it = (int)d.infected_time;
    const int base_idx = (int)(t0 + TE_BIAS) - et - 1 - it;               // lowest entry of the first Infected window
    { const int k = base_idx + (int)tid; win[tid] = (k >= 0 && k < (int)TE_SLOTS) ? ld(&d.hist[k]) : 0u; }
    __syncthreads();
    block_scan_1024(win, wtmp);                                            // win[i] = sum of hist[base_idx .. base_idx + i]
    if (tid < n_ahead) {
        // Infected window of step t0 + tid: entries [tid, tid + it] of the loaded range
        const uint32_t hi = win[tid + (uint32_t)it], lo = tid ? win[tid - 1u] : 0u;
        d.xf[tid] = hi - lo;
    }
    if (tid == 0) {
        ctrl->free_base = t0;
        // citizens Infected in at least one step of the chunk: exposure steps [first window's low end, last window's top]
        const uint32_t pairs = n_ahead ? win[n_ahead - 1u + (uint32_t)it] : 0u;
        ctrl->chunk_pairs = pairs;
        // Can this shard draw the chunk in one pass?  A citizen marks at most its home, its work building, its room and
        // its route.  The word after the census counts the shards that cannot, so that after the all-reduce every shard
        // takes the same form of the chunk (speculatively enqueued chunks advance on all shards or on none).
        const bool fits = d.items_cap && d.max_route <= CHUNK_ROUTE_MAX && d.n_routes < (1u << 25) &&
                          (unsigned long long)pairs * 4ull + 65536ull <= (unsigned long long)d.items_cap;
        d.xf[d.xf_n] = fits ? 0u : 1u;
    }
}

__global__ __launch_bounds__(FIN_TPB) void k_future(Dev d, uint32_t max_ahead, uint32_t limit_t)
{
    __shared__ uint32_t win[BF_WIN];
    __shared__ uint32_t wtmp[FIN_TPB / 64];
    future_body(d, max_ahead, limit_t, win, wtmp);
}

// Highest set bit index of m, -1 when m == 0.
__device__ __forceinline__ int top_bit(unsigned long long m) { return m ? 63 - __clzll((long long)m) : -1; }

// (g after f) for transition functions on the three mask states, two bits per state.
__device__ __forceinline__ uint32_t mask_compose(uint32_t g, uint32_t f)
{
    return ((g >> (2u * (f & 3u))) & 3u) | (((g >> (2u * ((f >> 2) & 3u))) & 3u) << 2) | (((g >> (2u * ((f >> 4) & 3u))) & 3u) << 4);
}

__device__ __forceinline__ void decide_body(const Dev &d, uint32_t max_ahead, uint32_t limit_t, int allow_parallel, int adj_folded = 0)
{
    // One wavefront, no serial loop.  Lane l evaluates the (strict) threshold tests of steps l and 64 + l
    // (interventions.rs:116-170).  Then, per step j of the chunk:
    //   lockdown in force      = the lockdown test of step j - 1                     (interventions.rs:116-128)
    //   at_work / bus_dir      = set by the last step <= j that ran its schedule arm  (citizen.rs:176-206: a locked-down
    //                            step runs none), found with ballots and count-leading-zeros
    //   mask status in force   = the three-state machine of interventions.rs:142-180 applied to steps 0..j-1: an
    //                            exclusive scan of transition functions under composition
    Ctrl *ctrl = d.ctrl;
    const uint32_t lane = threadIdx.x;
    const uint32_t t0 = ld(&ctrl->t);
    const uint32_t n_ahead = t0 > limit_t ? 0u : (limit_t - t0 + 1u < max_ahead ? limit_t - t0 + 1u : max_ahead);
    const uint32_t lim_in = n_ahead < FREE_MAX ? n_ahead : FREE_MAX;
    // under a vaccination programme only a chunk whose vaccinations are planned may run (k_chunk_vax); the Infected census ahead
    // then loses those the plan vaccinates before (prefix sums of xf_adj)
    const bool vax = ld(&ctrl->have_elig) != 0u;
    const bool ok = (vax ? (ld(&ctrl->vax_chunk) != 0u && ld(&ctrl->vax_fail) == 0u) : !ld(&ctrl->vacc_active)) && !ld(&ctrl->need_seq) && !ld(&ctrl->finished) && !ld(&ctrl->error) && ld(&ctrl->free_base) == t0;
    uint32_t adj[2] = { 0u, 0u };
    if (vax && ok && !adj_folded) {                       // (sharded: folded into buffer F before its all-reduce, k_shard_prep)
        uint32_t carry = 0u;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const uint32_t jj = 64u * r + lane;
            uint32_t x = jj < FREE_MAX ? ld(&d.xf_adj[jj]) : 0u;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(x, o, 64); if (lane >= (uint32_t)o) x += y; }
            adj[r] = x + carry;
            carry += __shfl(x, 63, 64);
        }
    }
    const uint32_t lock_init = ld(&ctrl->lockdown), mask_init = ld(&ctrl->mask), work_init = ld(&ctrl->at_work), bus_init = ld(&ctrl->bus_dir);
    unsigned long long m_vacc[2], m_lock[2];
    uint32_t f_mask[2];                                   // transition function of the lane's step in each round
    bool in[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const uint32_t j = 64u * r + lane;
        in[r] = j < lim_in && t0 + j <= d.max_steps;
        const double x = in[r] ? (double)(ld(&d.xf[j]) + adj[r]) / (double)d.n_global : 0.0;   // infected_percentage, statistics.rs:252
        m_vacc[r] = __ballot(in[r] && d.thr_vacc < x);
        m_lock[r] = __ballot(in[r] && d.thr_lockdown < x);
        const uint32_t from_none = (in[r] && d.thr_mask_pt < x) ? 1u : 0u;
        const uint32_t from_pt = !in[r] ? 1u : (x < d.thr_mask_pt ? 0u : (d.thr_mask_all < x ? 2u : 1u));
        const uint32_t from_all = (in[r] && x < d.thr_mask_all) ? 1u : 2u;
        f_mask[r] = from_none | (from_pt << 2) | (from_all << 4);
    }
    // steps before the one that starts the vaccination programme
    uint32_t n_ok = 0;
    if (ok) {
        const unsigned long long valid0 = __ballot(in[0]), valid1 = __ballot(in[1]);
        const uint32_t n_valid = (uint32_t)(__popcll(valid0) + __popcll(valid1));
        const uint32_t first_v = m_vacc[0] ? (uint32_t)__ffsll((long long)m_vacc[0]) - 1u : (m_vacc[1] ? 64u + (uint32_t)__ffsll((long long)m_vacc[1]) - 1u : n_valid);
        n_ok = (!vax && first_v < n_valid) ? first_v : n_valid;             // (a programme that runs cannot start again)
    }
    // exclusive scan of the mask transition functions (identity = 0b100100)
    uint32_t pre[2];
    uint32_t carry = 0x24u;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        uint32_t incl = f_mask[r];
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(incl, o, 64); if (lane >= (uint32_t)o) incl = mask_compose(incl, y); }
        uint32_t excl = __shfl_up(incl, 1, 64);
        if (lane == 0) excl = 0x24u;
        pre[r] = mask_compose(excl, carry);               // everything before this step, earlier round first
        carry = mask_compose(__shfl(incl, 63, 64), carry);
    }
    const Decision none = { 0u, 0u, 0u, 0u };
    Decision mine[2] = { none, none };
    unsigned long long run_mask[2];                       // steps that run their schedule arm (not locked down)
    uint32_t lockd[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const uint32_t j = 64u * r + lane;
        const bool prev_lock = j == 0 ? lock_init != 0u : (j == 64u ? ((m_lock[0] >> 63) & 1ull) != 0ull : ((m_lock[r] >> (lane - 1u)) & 1ull) != 0ull);
        lockd[r] = prev_lock ? 1u : 0u;
        run_mask[r] = __ballot(!prev_lock);
    }
    // at_work and bus_dir: last deciding step at or before j, over both rounds
    unsigned long long s1[2], s0[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const uint32_t h = (t0 + 64u * r + lane) % 24u;
        s1[r] = __ballot(!lockd[r] && h == d.start_hour);
        s0[r] = __ballot(!lockd[r] && h == d.end_hour);
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const unsigned long long le = lane == 63u ? ~0ull : ((1ull << (lane + 1u)) - 1ull);
        // position: the latest "starts work" / "goes home" arm among the steps that ran
        int last1 = top_bit(s1[r] & le), last0 = top_bit(s0[r] & le);
        if (r == 1) { last1 = last1 >= 0 ? last1 + 64 : top_bit(s1[0]); last0 = last0 >= 0 ? last0 + 64 : top_bit(s0[0]); }
        const uint32_t at_work = last1 > last0 ? 1u : (last0 > last1 ? 0u : work_init);
        // bus: the latest step that ran any arm decides (every arm assigns on_public_transport)
        int last_run = top_bit(run_mask[r] & le);
        if (r == 1) last_run = last_run >= 0 ? last_run + 64 : top_bit(run_mask[0]);
        uint32_t bus_dir = bus_init;
        if (last_run >= 0) {
            const uint32_t hh = (t0 + (uint32_t)last_run) % 24u;
            bus_dir = hh == d.start_hour - 1u ? 1u : (hh == d.end_hour - 1u ? 2u : 0u);
        }
        const uint32_t msk = (pre[r] >> (2u * mask_init)) & 3u;
        mine[r] = Decision{ lockd[r], msk, at_work, bus_dir };
    }
    if (allow_parallel) {
        // A one-pass chunk registers at most CHUNK_BUS_STEPS steps with riders on a bus (two a day -- unless a lockdown froze
        // them there, Q8: then every step is one).  Rather than give up the form, the chunk ends in front of the step that
        // would be one too many.
        const unsigned long long b0 = __ballot(lane < n_ok && mine[0].bus_dir != 0u), b1 = __ballot(64u + lane < n_ok && mine[1].bus_dir != 0u);
        const unsigned long long lt_ = (1ull << lane) - 1ull;
        const uint32_t before0 = (uint32_t)__popcll(b0 & lt_), before1 = (uint32_t)(__popcll(b0) + __popcll(b1 & lt_));
        const unsigned long long over0 = __ballot(((b0 >> lane) & 1ull) && before0 == CHUNK_BUS_STEPS);
        const unsigned long long over1 = __ballot(((b1 >> lane) & 1ull) && before1 == CHUNK_BUS_STEPS);
        if (over0) n_ok = (uint32_t)__ffsll((long long)over0) - 1u;
        else if (over1) n_ok = 64u + (uint32_t)__ffsll((long long)over1) - 1u;
    }
    // what is in force after the chunk = what would be in force during step n_ok, except that position and bus
    // are those of step n_ok - 1 (k_batch_finish reads them from there)
    if (lane < n_ok) d.dec[lane] = mine[0];
    if (64u + lane < n_ok) d.dec[64u + lane] = mine[1];
    {
        // entry n_ok: lockdown / mask after the last step of the chunk; computed by the lane that owns step n_ok when
        // it exists in the arrays, else from the scan totals
        const uint32_t jn = n_ok;
        uint32_t lock_after, mask_after;
        if (jn == 0) { lock_after = lock_init; mask_after = mask_init; }
        else {
            const uint32_t jl = jn - 1u;                                  // last step of the chunk
            lock_after = (uint32_t)((m_lock[jl >> 6] >> (jl & 63u)) & 1ull);
            // mask after step jl = f_jl applied to the mask in force during jl
            const uint32_t f_last = __shfl(jl < 64u ? f_mask[0] : f_mask[1], (int)(jl & 63u), 64);
            const uint32_t pre_last = __shfl(jl < 64u ? pre[0] : pre[1], (int)(jl & 63u), 64);
            mask_after = (mask_compose(f_last, pre_last) >> (2u * mask_init)) & 3u;
        }
        const uint32_t aw_last = jn ? __shfl(jn - 1u < 64u ? mine[0].at_work : mine[1].at_work, (int)((jn - 1u) & 63u), 64) : work_init;
        const uint32_t bd_last = jn ? __shfl(jn - 1u < 64u ? mine[0].bus_dir : mine[1].bus_dir, (int)((jn - 1u) & 63u), 64) : bus_init;
        if (lane == 0) d.dec[jn] = Decision{ lock_after, mask_after, aw_last, bd_last };
    }
    const unsigned long long bus_m0 = __ballot(lane < n_ok && mine[0].bus_dir != 0u), bus_m1 = __ballot(64u + lane < n_ok && mine[1].bus_dir != 0u);
    for (uint32_t i = lane; i < HOT_RESET; i += 64u) d.hot[i * HOT_STRIDE] = 0u;
    if (lane == 0) {
        ctrl->chunk_ok = n_ok; ctrl->chunk_t0 = t0;
        // the log slice of everybody Infected in some step of the chunk (k_chunk_marks starts from it)
        const int lo_te = (int)(t0 + TE_BIAS) - (int)d.exposed_time - 1 - (int)d.infected_time;
        const int hi_te = (int)(t0 + n_ok + TE_BIAS) - (int)d.exposed_time - 2;
        ctrl->chunk_i0 = (hi_te < 0 || n_ok == 0u) ? 0u : ld(&d.log_off[lo_te < 0 ? 0 : lo_te]);
        ctrl->chunk_i1 = (hi_te < 0 || n_ok == 0u) ? 0u : ld(&d.log_off[hi_te + 1]);
        // ... and of those that TURN Infected in it (exposure steps from the one whose stretch starts in step 0): what the
        // persistent map has to enter
        const int e_te = (int)(t0 + TE_BIAS) - (int)d.exposed_time - 1;
        ctrl->chunk_e0 = (hi_te < 0 || n_ok == 0u) ? 0u : ld(&d.log_off[e_te < 0 ? 0 : e_te]);
        ctrl->pmap_chunk = 0u;
        // riders are on a bus in at most CHUNK_BUS_STEPS steps of a one-pass chunk (two a day unless a lockdown froze them
        // there, Q8): that bounds the (route, bus step) pairs a wavefront of k_chunk_marks can register.  The same on all shards.
        const uint32_t bus_steps = (uint32_t)(__popcll(bus_m0) + __popcll(bus_m1));
        ctrl->chunk_parallel = (allow_parallel && ld(&d.xf[d.xf_n]) == 0u && bus_steps <= CHUNK_BUS_STEPS) ? 1u : 0u;
        ctrl->chunk_bus = bus_steps;
        ctrl->n_items = 0u; ctrl->n_newexp = 0u; ctrl->n_units = 0u; ctrl->unit_next = 0u; ctrl->n_route_pairs = 0u; ctrl->n_route_pairs_big = 0u;
    }
    d.cursor[lane] = 0u;
    if (lane < FREE_MAX - 64u) d.cursor[64u + lane] = 0u;
    // (the plan's census adjustments have been used -- by this kernel, or folded into buffer F by k_shard_prep --: zero for the next plan)
    d.xf_adj[lane] = 0u;
    if (lane < FREE_MAX + 2u - 64u) d.xf_adj[64u + lane] = 0u;
}

__global__ __launch_bounds__(64) void k_decide(Dev d, uint32_t max_ahead, uint32_t limit_t, int allow_parallel, int adj_folded)
{
    decide_body(d, max_ahead, limit_t, allow_parallel, adj_folded);
}

// ------------------------------------------------------------------ vaccination inside a chunk
// While a vaccination programme runs (simulator.rs:524-553) every step sets `vaccination_rate` citizens Vaccinated: the
// first k distinct members of citizens_eligible_for_vaccine in the candidate sequence of that step (RNG contract,
// DESIGN.md 2).  Candidates are pure functions of (i, step), and the eligible set only ever loses citizens that are exposed
// on public transport (simulator.rs:447-449, Q10) -- so who is vaccinated when is known for a whole chunk ahead, up to those
// few removals.  k_chunk_vax plans the chunk: workgroup j takes step t0 + j, walks its candidate sequence exactly as
// k_finish does, notes the chosen citizens (Dev::vax_ev) and leaves "Vaccinated at the end of step j" in their words
// (earliest step wins, atomicMax on the vax field).  Everything downstream reads the field: an Infected citizen stops
// marking after that step (k_chunk_marks), a Susceptible one takes no draw after it (member_pairs), the census moves
// (k_chunk_vax_adj for the Infected counts the decisions need, k_chunk_count for the records).  The plan is speculative in
// one respect only: a citizen it chose for step j may be exposed on a bus in a step s <= j of this very chunk, which removes
// it from the set before its turn.  k_chunk_count finds the earliest such step s* (Ctrl::chunk_cut); the chunk is then
// committed up to s* - 1 and the next chunk starts AT s*.  What happens in step s* itself does not depend on the plan from s*
// on, so everybody the cut chunk saw exposed on a bus in s* will be again: k_chunk_scatter marks them (CW_PLAN_SKIP) and the
// next plan leaves them out -- it cannot be cut at s* again.
// Sharded runs: a candidate's eligibility is known to the shard that owns the citizen.  k_vax_live writes, for every step of
// the chunk ahead, the liveness bits of its first PLAN_W candidates (own citizens only; the shards SUM-all-reduce buffer V, the
// bits being disjoint) and, in the header, this shard's eligible count, riders and whether it can plan at all.
// REPAIR (k_vax_live<true>, before k_chunk_vax<true> walks the steps from Ctrl::replan_from on again): the liveness as it truly stood
// in each of those steps (eligible by the final word, or exposed on a bus in a LATER step of the chunk); the other steps' rows are
// zeroed (the buffer is summed in place a second time).
template <bool REPAIR>
__global__ __launch_bounds__(TPB) void k_vax_live(Dev d, uint32_t max_ahead, uint32_t limit_t)
{
    Ctrl *ctrl = d.ctrl;
    const uint32_t j = blockIdx.y, i = blockIdx.x * TPB + threadIdx.x;
    const uint32_t t0 = ctrl->t;
    const uint32_t n_chunk = ctrl->chunk_ok, from = ctrl->replan_from;
    const uint32_t n_ahead = REPAIR ? ((ctrl->vax_chunk && ctrl->chunk_parallel && from < n_chunk) ? n_chunk : 0u)
                                    : (t0 > limit_t ? 0u : (limit_t - t0 + 1u < max_ahead ? limit_t - t0 + 1u : max_ahead));
    if (j == 0 && blockIdx.x == 0 && threadIdx.x < XV_HEADER) {
        const uint32_t k = threadIdx.x;
        d.xv[k] = k == 0 ? ctrl->elig_count : k == 1 ? d.n_pt : k == 2 ? ((ctrl->finished || ctrl->error) ? 1u : 0u) : 0u;
        if (k == 0 && !REPAIR) ctrl->vax_fail = 0u;
    }
    bool live = false;
    if (ctrl->have_elig && j < n_ahead && (!REPAIR || j >= from)) {
        const uint32_t cand = vacc_candidate(d, i, t0 + j);
        if (cand >= d.id_base && cand - d.id_base < d.n) {
            const uint32_t w = d.cit[cand - d.id_base];
            const uint32_t e = CW_TE(w) - TE_BIAS - t0;
            const bool later_bus = REPAIR && (w & CW_BUS_EXPOSED) && CW_TE(w) < TE_RECOVERED && e < n_chunk && e > j;
            live = (eligible(w, ctrl->trigger_step) || later_bus) && !(w & CW_PLAN_SKIP);
        }
    }
    const unsigned long long m = __ballot(live);
    if ((threadIdx.x & 63u) == 0) { uint32_t *row = d.xv + XV_HEADER + (size_t)j * (PLAN_W / 32u); row[i >> 5] = (uint32_t)m; row[(i >> 5) + 1u] = (uint32_t)(m >> 32); }
}

// sharded: the liveness of the candidates comes from buffer V (k_vax_live, all-reduced), every shard walks the same sequence
// and accepts the same candidates, and keeps the events of its own citizens.
// Launched with one workgroup more than there are steps, that one makes the census ahead (k_future's work: nothing the plan reads
// or writes, and a kernel boundary costs as much as the census).
// The citizens expose_min listed (exposed on a bus with a planned vaccination in their word) whose FINAL exposure is that one: they
// left the eligible set in that step, their planned vaccination is void (the field is cleared here, before any step is walked
// again: a walk may choose the same citizen anew for an EARLIER step), and the steps from the earliest such exposure on are the
// ones k_chunk_vax<true> walks again.
__global__ __launch_bounds__(FIN_TPB) void k_chunk_lost(Dev d)
{
    __shared__ uint32_t s_from;
    Ctrl *ctrl = d.ctrl;
    const uint32_t tid = threadIdx.x;
    const uint32_t n_chunk = ctrl->chunk_ok, t0 = ctrl->t;
    const uint32_t n_lost = d.hot[HOT_LOST * HOT_STRIDE];
    if (!ctrl->vax_chunk || !ctrl->chunk_parallel || n_chunk == 0u || n_lost == 0u) return;
    if (tid == 0) s_from = FREE_MAX + 1u;
    __syncthreads();
    uint32_t lo = FREE_MAX + 1u;
    auto look = [&](uint32_t m) {
        if (m >= d.n) return;
        const uint32_t w = d.cit[m], e = CW_TE(w) - TE_BIAS - t0;
        if (!(w & CW_BUS_EXPOSED) || CW_TE(w) >= TE_RECOVERED || e >= n_chunk || CW_VAX_REL(w) == CW_VAX_NONE) return;   // (listed twice: cleared already)
        lo = min(lo, e);
        atomicAnd(&d.cit[m], ~CW_VAX_MASK);
        if (d.world > 1u) d.xl[e] = 1u;
    };
    if (n_lost <= LOST_CAP) for (uint32_t i = tid; i < n_lost; i += FIN_TPB) look(d.lost_list[i]);
    else {
        // (more than the list holds: everybody exposed in the chunk is looked at)
        const uint32_t r = tid & (SUBQ - 1u);
        const uint32_t n_new = min(d.hot[(HOT_NEWEXP + r) * HOT_STRIDE], d.newexp_cap);
        const uint32_t *list = d.newexp + (size_t)r * d.newexp_cap;
        for (uint32_t i = tid / SUBQ; i < n_new; i += FIN_TPB / SUBQ) look(list[i]);
    }
    if (lo <= FREE_MAX) atomicMin(&s_from, lo);
    __syncthreads();
    if (d.world > 1u) return;                                                 // (sharded: the earliest step of ALL shards, k_lost_global)
    if (tid == 0 && s_from <= FREE_MAX) { ctrl->replan_from = s_from; ctrl->repair_ran = 1u; ctrl->vax_repairs += 1u; }
}

// Sharded: buffer L summed over the shards -- the plan is walked again from the earliest step in which ANY shard lost a citizen.
__global__ __launch_bounds__(64) void k_lost_global(Dev d)
{
    Ctrl *ctrl = d.ctrl;
    const uint32_t lane = threadIdx.x, n = ctrl->chunk_ok;
    if (!ctrl->vax_chunk || !ctrl->chunk_parallel || n == 0u) return;
    const unsigned long long m0 = __ballot(lane < n && d.xl[lane] != 0u), m1 = __ballot(64u + lane < n && d.xl[64u + lane] != 0u);
    if (lane == 0 && (m0 | m1)) {
        ctrl->replan_from = m0 ? (uint32_t)__ffsll((long long)m0) - 1u : 64u + (uint32_t)__ffsll((long long)m1) - 1u;
        ctrl->repair_ran = 1u; ctrl->vax_repairs += 1u;
    }
}

// REPAIR (k_chunk_vax<true>, after the draws of the chunk, before its counts; unsharded contexts on the per-chunk map): a citizen the
// plan vaccinates at the end of step f was exposed on a bus in step e <= f.  It left the eligible set with that exposure, so in
// every step from e on in which it was among the chosen the walk takes the next eligible candidate instead -- and nothing else
// changes: the chosen stay in the set (Q10), so every step's walk is a function of the words alone.  Round 2 cut the chunk at e
// and threw the draws of everything behind it away (uk64m's last 200 steps, with 17 000 bus exposures per 96 steps, cost 14 of
// the run's 33 ms that way).  Now the steps from the earliest such e on are walked AGAIN, with eligibility as it truly stood in
// each step (eligible by the final word, or exposed on a bus in a LATER step of this chunk): the lists of the chosen are rewritten,
// the lost citizens' fields are cleared, the newly chosen get theirs.  A newly chosen citizen (or one whose vaccination moves to
// an earlier step) is harmless when nothing it did behind that step mattered: it is not Infected in any later step of the chunk
// (its marks would have to go, and with them the counts others were drawn with) and was not exposed in a later one.  Otherwise the
// chunk is cut BEHIND that step (chunk_cut = j + 1): everything up to and including it stands.
template <bool REPAIR>
__global__ __launch_bounds__(FIN_TPB) void k_chunk_vax(Dev d, uint32_t max_ahead, uint32_t limit_t, int sharded)
{
    __shared__ FinishShared sm;
    __shared__ uint32_t n_local;
    if (blockIdx.x >= FREE_MAX) {
        if (REPAIR) return;
        __shared__ uint32_t win[BF_WIN];
        __shared__ uint32_t wtmp[FIN_TPB / 64];
        future_body(d, max_ahead, limit_t, win, wtmp);
        return;
    }
    Ctrl *ctrl = d.ctrl;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    const uint32_t j = blockIdx.x;
    const uint32_t t0 = ctrl->t;
    uint32_t n_chunk = 0u;
    if (REPAIR) {
        n_chunk = ctrl->chunk_ok;
        const uint32_t from = ctrl->replan_from;                              // (k_chunk_lost)
        if (!ctrl->vax_chunk || !ctrl->chunk_parallel || j >= n_chunk || j < from) return;
    }
    const uint32_t n_ahead = REPAIR ? n_chunk : (t0 > limit_t ? 0u : (limit_t - t0 + 1u < max_ahead ? limit_t - t0 + 1u : max_ahead));
    // (every workgroup -- and, sharded, every shard -- takes the same decision from the same words; nothing here writes them)
    const uint32_t elig_all = sharded ? d.xv[0] : ctrl->elig_count, riders_all = sharded ? d.xv[1] : d.n_pt;
    const bool plan = ctrl->have_elig && !ctrl->finished && !ctrl->error && !ctrl->need_seq && !(sharded && d.xv[2]) &&
                      elig_all > d.vaccination_rate + riders_all;      // the set cannot shrink to the "whole set" case inside the chunk
    if (REPAIR) { if (tid == 0) { d.vax_cnt[j] = 0u; d.vax_now[j] = 0u; n_local = 0u; } }
    else if (tid == 0) {
        for (uint32_t q = 0; q < 4u; ++q) d.vax_delta[q * (FREE_MAX + 2u) + j] = 0u;                // (xf_adj: zero since the last k_decide)
        d.vax_cnt[j] = 0u; d.vax_now[j] = 0u;
        n_local = 0u;
        if (j == 0) {
            d.hot[HOT_LOST * HOT_STRIDE] = 0u; ctrl->replan_from = FREE_MAX + 1u; ctrl->repair_ran = 0u;
            ctrl->vax_chunk = plan ? 1u : 0u;
            ctrl->vax_planned = plan ? n_ahead : 0u;
            ctrl->n_neg = 0u; ctrl->n_cancel = 0u;
            if (!sharded) ctrl->vax_fail = 0u;
            ctrl->chunk_cut = FREE_MAX + 1u;
            for (uint32_t z = FREE_MAX; z < FREE_MAX + 2u; ++z) for (uint32_t q = 0; q < 4u; ++q) d.vax_delta[q * (FREE_MAX + 2u) + z] = 0u;
        }
    }
    if (!plan || j >= n_ahead) return;
    const uint32_t t = t0 + j, k = d.vaccination_rate, tstep = ctrl->trigger_step;
    const uint32_t *bits = d.xv + XV_HEADER + (size_t)j * (PLAN_W / 32u);
    for (uint32_t i = tid; i < VACC_TABLE; i += FIN_TPB) { sm.tab_key[i] = 0xFFFFFFFFu; sm.tab_idx[i] = 0xFFFFFFFFu; }
    __syncthreads();
    uint32_t already = 0;
    for (uint32_t base = 0; already < k; base += VACC_BATCH) {
        uint32_t cj[4], slot[4], cw[4]; bool live[4], mine_c[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t i = base + tid * 4u + q;
            cj[q] = vacc_candidate(d, i, t);
            mine_c[q] = cj[q] >= d.id_base && cj[q] - d.id_base < d.n;
            cw[q] = mine_c[q] ? d.cit[cj[q] - d.id_base] : 0u;
            if (sharded) live[q] = i < PLAN_W && ((bits[i >> 5] >> (i & 31u)) & 1u) != 0u;
            else if (REPAIR) {
                // as the set truly stood in this step: who is exposed on a bus in a LATER step of the chunk was still in it
                const uint32_t e = CW_TE(cw[q]) - TE_BIAS - t0;
                const bool later_bus = (cw[q] & CW_BUS_EXPOSED) && CW_TE(cw[q]) < TE_RECOVERED && e < n_chunk && e > j;
                live[q] = (eligible(cw[q], tstep) || later_bus) && !(cw[q] & CW_PLAN_SKIP);
            }
            else live[q] = eligible(cw[q], tstep) && !(cw[q] & CW_PLAN_SKIP);
            slot[q] = 0;
            if (live[q]) {
                uint32_t sl = (cj[q] * 2654435761u) >> 18;        // 14 bits
                for (;;) {
                    const uint32_t old = atomicCAS(&sm.tab_key[sl], 0xFFFFFFFFu, cj[q]);
                    if (old == 0xFFFFFFFFu || old == cj[q]) break;
                    sl = (sl + 1u) & (VACC_TABLE - 1u);
                }
                atomicMin(&sm.tab_idx[sl], i);
                slot[q] = sl;
            }
        }
        __syncthreads();
        bool first[4]; uint32_t mine = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) { first[q] = live[q] && sm.tab_idx[slot[q]] == base + tid * 4u + q; mine += first[q]; }
        uint32_t incl = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t v = __shfl_up(incl, o, 64); if (lane >= (uint32_t)o) incl += v; }
        if (lane == 63) sm.wsum[wv] = incl;
        __syncthreads();
        if (tid == 0) { uint32_t a = 0; for (uint32_t w = 0; w < FIN_TPB / 64; ++w) { const uint32_t v = sm.wsum[w]; sm.wsum[w] = a; a += v; } sm.s_total = a; }
        __syncthreads();
        uint32_t pos = already + sm.wsum[wv] + incl - mine;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (first[q]) {
                if (pos < k && mine_c[q]) {
                    const uint32_t c = cj[q] - d.id_base;
                    d.vax_ev[(size_t)j * VACC_MAX_RATE + atomicAdd(&n_local, 1u)] = c;      // (the order inside a step does not matter)
                    // unconditional (simulator.rs:551) -- but a citizen that is Vaccinated already stays what it is
                    if (CW_TE(cw[q]) != TE_VACCINATED) {
                        const uint32_t old = atomicMax(&d.cit[c], (cw[q] & ~CW_VAX_MASK) | CW_VAX_FIELD(j));
                        if (!REPAIR && CW_TE(cw[q]) < TE_RECOVERED && (CW_VAX_REL(old) == CW_VAX_NONE || CW_VAX_REL(old) > j)) {
                            // The Infected census ahead (buffer F) counts everybody whose exposure step makes it Infected; those the plan
                            // vaccinates before leave it: the stretch of the chunk in which this citizen would have been Infected behind
                            // step j goes into the difference array xf_adj (k_decide adds its prefix sums to F) -- for the step that WINS:
                            // a step that takes the citizen over from a later one takes that one's stretch out again (round 3: a
                            // kernel of its own did this from the final words, k_chunk_vax_adj).
                            const int a = (int)CW_TE(cw[q]) - (int)TE_BIAS + (int)d.exposed_time + 1 - (int)t0, hi = min(a + (int)d.infected_time, (int)n_ahead - 1);
                            const int lo = max(a, (int)j + 1);
                            if (lo <= hi) { atomicSub(&d.xf_adj[lo], 1u); atomicAdd(&d.xf_adj[hi + 1], 1u); }
                            if (CW_VAX_REL(old) != CW_VAX_NONE) {
                                const int lo2 = max(a, (int)CW_VAX_REL(old) + 1);
                                if (lo2 <= hi) { atomicAdd(&d.xf_adj[lo2], 1u); atomicSub(&d.xf_adj[hi + 1], 1u); }
                            }
                        }
                        if (REPAIR && (CW_VAX_REL(old) == CW_VAX_NONE || CW_VAX_REL(old) > j)) {
                            // newly chosen for this step (or moved here from a later one): harmless unless something it did behind
                            // step j mattered -- Infected in a later step of the chunk, or exposed in one
                            const uint32_t te = CW_TE(cw[q]);
                            bool bad = false;
                            if (te < TE_RECOVERED) {
                                const int e = (int)te - (int)TE_BIAS - (int)t0;                                  // exposure step (may lie before the chunk)
                                const int a = e + (int)d.exposed_time + 1, b = a + (int)d.infected_time;       // Infected in steps a .. b
                                if (e > (int)j && e < (int)n_chunk) bad = true;
                                if (max(a, (int)j + 1) <= min(b, (int)n_chunk - 1)) bad = true;
                            }
                            if (bad) { atomicMin(&ctrl->chunk_cut, j + 1u); if (d.world > 1u) d.xc[j + 1u] = 1u; }
                        }
                    }
                }
                pos++;
            }
        }
        const uint32_t got = sm.s_total;
        __syncthreads();
        already += got < k - already ? got : k - already;
        if (sharded && base + VACC_BATCH >= PLAN_W && already < k) {
            // the exchanged window was too short: no plan -- or, walking a step again: the chunk ends in front of this step
            if (tid == 0) { if (REPAIR) { atomicMin(&ctrl->chunk_cut, j); d.xc[j] = 1u; } else atomicAdd(&ctrl->vax_fail, 1u); }
            break;
        }
        if (base >= (1u << 26) && already < k) { if (tid == 0) ctrl->error = (uint32_t)(-ESIM_ERANGE); break; }   // every wave must reach an exit
    }
    __syncthreads();
    if (tid == 0) { d.vax_cnt[j] = n_local; d.vax_now[j] = already; }
}

__device__ __forceinline__ void map_cancel(const Dev &d, Ctrl *ctrl, uint32_t c, uint32_t w, uint32_t t0, uint32_t j);
// The Infected census ahead (buffer F) counts everybody whose exposure step makes it Infected; those the plan vaccinates
// before leave it.  One thread per planned citizen: if its own step is the one that won, and the citizen is Infected in some
// later step of the chunk, that stretch goes into the difference array xf_adj (k_decide adds its prefix sums to F).
__global__ __launch_bounds__(TPB) void k_chunk_vax_adj(Dev d, uint32_t max_ahead, uint32_t limit_t, int pmap)
{
    const Ctrl *ctrl = d.ctrl;
    if (!ctrl->vax_chunk || ctrl->vax_fail) return;
    const uint32_t t0 = ctrl->t;
    const uint32_t n_ahead = t0 > limit_t ? 0u : (limit_t - t0 + 1u < max_ahead ? limit_t - t0 + 1u : max_ahead);
    const uint32_t j = blockIdx.x;
    if (j >= n_ahead) return;
    const uint32_t cnt = d.vax_cnt[j];
    for (uint32_t i = threadIdx.x; i < cnt; i += TPB) {
        const uint32_t cz = d.vax_ev[(size_t)j * VACC_MAX_RATE + i];
        const uint32_t w = d.cit[cz], te = CW_TE(w);
        if (CW_VAX_REL(w) != j || te >= TE_RECOVERED) continue;                // not the winning step / never exposed
        const int a = (int)te - (int)TE_BIAS + (int)d.exposed_time + 1 - (int)t0;   // Infected in steps [a, a + infected_time] of the chunk
        // persistent map: a citizen it holds already (k_map_enter handles those that turn Infected in this chunk) and whose stretch
        // reaches beyond step j is cancelled from step j + 1 on -- to the end of its stretch, whatever the chunk's length
        // (noted here, entered by k_map_enter once the chunk's length is known: a step beyond its end is not committed by it)
        if (pmap && (w & CW_IN_MAP) && ctrl->map_t == t0 && a + (int)d.infected_time > (int)j) {
            const uint32_t q = atomicAdd(&const_cast<Ctrl *>(ctrl)->n_cancel, 1u);
            if (q < NEG_CAP) { d.cancel_list[2u * q] = cz; d.cancel_list[2u * q + 1u] = j; } else RAISE(const_cast<Ctrl *>(ctrl), ESIM_ERANGE, ERR_AT_NEG_LIST);
        }
    }
}

// ---------------------------------------------------------------------- sharded chunks: the commuter exchange
// A building or school room whose members live on several shards is shared (esim_shard_population).  An Infected member
// standing in it matters to every shard that has members there: per chunk, each shard sends the citizen words of its own
// Infected whose work building is shared, with the building's and the room's index in the shared tables (k_shared_pack; the
// segments are all-gathered), and k_chunk_marks enters the received ones into its map next to its own.  The slice walked is
// the one of the longest chunk that can follow (the decisions come later); an entry whose stretch misses the chunk is dropped
// by the receiver.
__global__ __launch_bounds__(TPB) void k_shared_pack(Dev d, uint32_t max_ahead, uint32_t limit_t)
{
    const Ctrl *ctrl = d.ctrl;
    uint32_t *seg = d.xs + (size_t)d.rank * (1u + 3u * d.xs_cap);
    const uint32_t t0 = ctrl->t;
    const uint32_t n_ahead = t0 > limit_t ? 0u : (limit_t - t0 + 1u < max_ahead ? limit_t - t0 + 1u : max_ahead);
    if (n_ahead == 0u) return;
    const int lo_te = (int)(t0 + TE_BIAS) - (int)d.exposed_time - 1 - (int)d.infected_time;
    const int hi_te = (int)(t0 + n_ahead + TE_BIAS) - (int)d.exposed_time - 2;
    if (hi_te < 0) return;
    const uint32_t i0 = d.log_off[lo_te < 0 ? 0 : lo_te], i1 = d.log_off[hi_te + 1];
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t base = i0 + (blockIdx.x * TPB + threadIdx.x - lane); base < i1; base += gridDim.x * TPB) {
        const uint32_t i = base + lane;
        uint32_t w = 0u, sb = 0u, sr = 0xFFFFFFFFu;
        bool send = false;
        if (i < i1) {
            const uint32_t c = d.log[i];
            w = d.cit[c];
            if ((w & FL_HAS_WORK) && CW_TE(w) < TE_RECOVERED) {
                const int32_t k = d.shared_of_bld[d.work[c]];
                if (k >= 0) {
                    send = true; sb = (uint32_t)k;
                    if (w & FL_WORK_SCHOOL) { const int32_t q = d.shared_of_room[d.room[c]]; sr = q >= 0 ? (uint32_t)q : 0xFFFFFFFFu; }
                }
            }
        }
        const unsigned long long m = __ballot(send);
        if (!m) continue;
        if (d.xs_out) {
            // all-to-all: the record goes into the segment of every OTHER shard that has members in the building
            const uint32_t to = send ? d.shared_mask[sb] & ~(1u << d.rank) : 0u;
            for (uint32_t r = 0; r < d.world; ++r) {
                const unsigned long long mr = __ballot((to >> r) & 1u);
                if (!mr) continue;
                uint32_t *out = d.xs_out + (size_t)r * (1u + 3u * d.xs_cap);
                uint32_t pos = 0u;
                if (lane == 0) pos = atomicAdd(&out[0], (uint32_t)__popcll(mr));
                pos = __shfl(pos, 0, 64) + (uint32_t)__popcll(mr & ((1ull << lane) - 1ull));
                if (((to >> r) & 1u) && pos < d.xs_cap) { out[1u + 3u * pos] = w; out[2u + 3u * pos] = sb; out[3u + 3u * pos] = sr; }
            }
            continue;
        }
        uint32_t pos = 0u;
        if (lane == 0) pos = atomicAdd(&seg[0], (uint32_t)__popcll(m));       // one atomic per wavefront
        pos = __shfl(pos, 0, 64) + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        if (send && pos < d.xs_cap) { seg[1u + 3u * pos] = w; seg[2u + 3u * pos] = sb; seg[3u + 3u * pos] = sr; }
    }
}

// Before buffer F is all-reduced: this shard's "cannot draw the chunk in one pass" word also covers the commuters received (they
// claim items too) and a segment that overflowed anywhere; and the Infected census ahead loses those the plan vaccinates before
// (prefix sums of xf_adj), so that the sum over the shards is the census the decisions need.
__global__ __launch_bounds__(128) void k_shard_prep(Dev d, uint32_t max_ahead, uint32_t limit_t)
{
    Ctrl *ctrl = d.ctrl;
    if (threadIdx.x != 0) return;
    const uint32_t t0 = ctrl->t;
    const uint32_t n_ahead = t0 > limit_t ? 0u : (limit_t - t0 + 1u < max_ahead ? limit_t - t0 + 1u : max_ahead);
    uint32_t n_remote = 0u; bool overflow = false;
    for (uint32_t r = 0; r < d.world; ++r) {
        const uint32_t cnt = d.xs[(size_t)r * (1u + 3u * d.xs_cap)];
        if (cnt > d.xs_cap) overflow = true;
        if (r != d.rank) n_remote += min(cnt, d.xs_cap);
    }
    // (what this shard saw: the segments it received -- and, in the all-to-all form, those it sent; the status exchange takes the
    // maximum over the shards, so that the segments grow alike everywhere)
    ctrl->xs_need = 0u;
    for (uint32_t r = 0; r < d.world; ++r) {
        if (r != d.rank || !d.xs_out) ctrl->xs_need = max(ctrl->xs_need, d.xs[(size_t)r * (1u + 3u * d.xs_cap)]);
        if (d.xs_out && r != d.rank) { const uint32_t o = d.xs_out[(size_t)r * (1u + 3u * d.xs_cap)]; ctrl->xs_need = max(ctrl->xs_need, o); if (o > d.xs_cap) overflow = true; }
    }
    // (a shard in a device-side error state makes the chunk a no-op on EVERY shard: the word is summed)
    const bool fits = d.xf[d.xf_n] == 0u && !overflow && !ctrl->error && !ctrl->finished &&
                      ((unsigned long long)ctrl->chunk_pairs + n_remote) * 4ull + 65536ull <= (unsigned long long)d.items_cap;
    d.xf[d.xf_n] = fits ? 0u : 1u;
    if (ctrl->vax_chunk && !ctrl->vax_fail) {
        uint32_t a = 0u;
        for (uint32_t j = 0; j < n_ahead; ++j) { a += d.xf_adj[j]; d.xf[j] += a; }
    }
    for (uint32_t j = 0; j < FREE_MAX + 2u; ++j) { d.xc[j] = 0u; d.xl[j] = 0u; }
}

// Marks of the first step of a chunk (the later ones are made by the k_pipe of the step before).
// ------------------------------------------------------------------------- time-parallel chunk
// Inside a chunk nothing a draw depends on changes: who is Infected and where (known ahead), the mask
// status, the Philox counters.  A citizen's exposure step is therefore simply the EARLIEST step at which any of
// its draws succeeds (later draws would have been skipped by `is_susceptible()`, simulator.rs:337), and within a
// step a building exposure precedes a bus exposure (simulator.rs:268-401).  With the exposure step in the top
// bits of the citizen word and the bus bit right below, that is one atomicMin per successful draw -- so all
// steps of the chunk are drawn in ONE pass.
//   k_chunk_marks  an item per building / room / route that somebody Infected stands in during the chunk, and per item the
//                  stretches of steps in which each of them stands there (generate_exposures)
//   k_chunk_fold   the stretches that did not fit an item's own records, summed into its per-step counters; the prefix sums
//                  that let the draw pass take the items in equal shares
//   k_chunk_draw   the (member, slot of four marked steps) pairs of every item, densely over the lanes (apply_exposures); long
//                  member lists are cut into units
//   k_chunk_units  the units, dealt evenly; routes of more than 64 riders
//   k_chunk_books  exposure counts, records, log entries, clean-up, the next chunk's decisions
//                  (k_chunk_count / k_chunk_scatter: its two wide parts as kernels of their own while many are Infected)
__device__ __forceinline__ uint32_t hash64(unsigned long long k)
{
    k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
    return (uint32_t)k;
}

// Items live in an open-addressing hash map keyed by (building | n_bld + room | n_bld + n_room + route).  Whoever
// inserts the key claims the item: it takes the next id of its wavefront's own id range (a counter bumped once per claim
// would serialise the pass) and writes the item's record.  Everything the others add to the item -- their interval
// records, the per-step counters of those that found no record free, a route's registered bus steps -- is indexed by the
// hash SLOT, which the probe itself returns: nobody ever waits for anybody.
#define ITEM_UNUSED 0xFFFFFFFFu

// An interval record: a citizen that is Infected in steps [a, b] of the chunk, the flags that decide where it stands
// in each of them, and whether the record sits in its work building / room or in its home.
#define IV_VALID   0x80000000u
#define IV_PT      (1u << 14)
#define IV_HW      (1u << 15)
#define IV_AS_WORK (1u << 16)
__device__ __forceinline__ uint32_t iv_present(uint32_t iv, uint32_t j, const Decision &q)
{
    if (!(iv & IV_VALID) || j < (iv & 127u) || j > ((iv >> 7) & 127u)) return 0u;
    if (q.bus_dir && (iv & IV_PT)) return 0u;                                 // on a bus (simulator.rs:181-186)
    const bool at_work = q.at_work && (iv & IV_HW);
    return ((iv & IV_AS_WORK) != 0u) == at_work ? 1u : 0u;
}

// DiseaseStatus of a citizen in step t0 + j of a chunk: a vaccination planned for the end of step f of the chunk
// (k_chunk_vax) makes it Vaccinated from step f + 1 on, whatever it was (simulator.rs:551).
__device__ __forceinline__ uint32_t status_in_chunk(const Dev &d, uint32_t w, uint32_t t0, uint32_t j)
{
    const uint32_t f = CW_VAX_REL(w);
    if (f != CW_VAX_NONE && j > f) return ESIM_VACCINATED;
    return status_of(CW_TE(w), t0 + j, d.exposed_time, d.infected_time);
}

// Where an Infected citizen stands in step t0 + j of the chunk (simulator.rs:181-198): bit 0 in the home building,
// bit 1 in the work building, bit 2 on the bus.  0 when not Infected in that step.
__device__ __forceinline__ uint32_t where_in_step(const Dev &d, uint32_t w, uint32_t t0, uint32_t j, const Decision &q)
{
    if (status_in_chunk(d, w, t0, j) != ESIM_INFECTED) return 0u;
    if (q.bus_dir && (w & FL_USES_PT)) return 4u;
    return (q.at_work && (w & FL_HAS_WORK)) ? 2u : 1u;
}

#define FX(x, i) ((uint32_t)__builtin_amdgcn_readlane((int)(x), (int)(i)))
// A set of steps of a chunk (FREE_MAX = 96 bits).
struct M96 { unsigned long long lo; uint32_t hi; };
__device__ __forceinline__ M96 m96_and(M96 a, M96 b) { return M96{ a.lo & b.lo, a.hi & b.hi }; }
__device__ __forceinline__ M96 m96_andn(M96 a, M96 b) { return M96{ a.lo & ~b.lo, a.hi & ~b.hi }; }
__device__ __forceinline__ bool m96_any(M96 a) { return a.lo != 0ull || a.hi != 0u; }
// steps a..b (a <= b < 96)
__device__ __forceinline__ M96 m96_range(uint32_t a, uint32_t b)
{
    M96 r = { 0ull, 0u };
    if (a < 64u) { const uint32_t e = min(b, 63u); r.lo = (e == 63u ? ~0ull : ((1ull << (e + 1u)) - 1ull)) & (~0ull << a); }
    if (b >= 64u) { const uint32_t s0 = a > 64u ? a - 64u : 0u, e = min(b - 64u, 31u); r.hi = (e == 31u ? ~0u : ((1u << (e + 1u)) - 1u)) & (~0u << s0); }
    return r;
}

// The steps of the chunk in which the citizen of interval record iv stands where the record was left (iv_present as a set):
// AW / BUS = the steps in which those with a work place are at work / riders are on a bus.  With a wavefront-uniform record
// this is scalar arithmetic; a lane then only picks its step's bit.
__device__ __forceinline__ M96 iv_steps(uint32_t iv, const M96 &AW, const M96 &BUS)
{
    if (!(iv & IV_VALID)) return M96{ 0ull, 0u };
    const M96 I = m96_range(iv & 127u, (iv >> 7) & 127u);
    const M96 rest = (iv & IV_PT) ? m96_andn(I, BUS) : I;
    const M96 atw = (iv & IV_HW) ? m96_and(rest, AW) : M96{ 0ull, 0u };
    return (iv & IV_AS_WORK) ? atw : m96_andn(rest, atw);
}
__device__ __forceinline__ void iv_count(uint32_t iv, uint32_t lane, const M96 &AW, const M96 &BUS, uint32_t &c0, uint32_t &c1)
{
    const M96 at = iv_steps(iv, AW, BUS);
    c0 += (uint32_t)(at.lo >> lane) & 1u;
    c1 += lane < 32u ? (at.hi >> lane) & 1u : 0u;
}
__device__ __forceinline__ void schedule_masks(uint32_t lane, uint32_t n, const Decision &q0, const Decision &q1, M96 &AW, M96 &BUS)
{
    AW = M96{ __ballot(lane < n && q0.at_work != 0u), (uint32_t)__ballot(64u + lane < n && q1.at_work != 0u) };
    BUS = M96{ __ballot(lane < n && q0.bus_dir != 0u), (uint32_t)__ballot(64u + lane < n && q1.bus_dir != 0u) };
}

// ------------------------------------------------------------------------------- persistent item map
// k_chunk_marks rebuilds the map of items from the exposure log in every chunk, and k_chunk_scatter tears it down again -- although
// an Infected citizen stands in the same buildings for its whole stretch of infected_time + 1 steps, three to four chunks.  In the
// persistent form an item's records are ABSOLUTE: the citizen's exposure step (from which its Infected stretch
// follows in any chunk) instead of a stretch relative to one chunk's first step.  A citizen is entered ONCE, by the chunk in which
// it turns Infected (k_map_enter walks the log slice of those only; CW_IN_MAP in its word), and never taken out: a record whose
// stretch has passed contributes nothing.  A vaccination planned for an Infected citizen (k_chunk_vax) adds a CANCELLATION
// record -- same citizen, negative, from the step after the vaccination on -- instead of searching for the record: it stays for
// good when the step is committed and is zeroed through Dev::neg_list when it is not.  Routes are items like the others: their
// records are their Infected riders, and the draw pass finds the bus steps with an Infected rider from them.  Everything that
// advances the clock without maintaining the map invalidates it (Ctrl::map_t); the next chunk then starts with k_map_clear and
// enters everybody who is Infected in it -- which is also how dead items are shed every few chunks.
//   record: bits 0-12 exposure step + TE_BIAS | 13-25 cancellation: absent AFTER this step (absolute; only with PIV_NEG) |
//           26 rides public transport | 27 has a work place | 28 stands in its work building / room | 29 rider record of a route |
//           30 cancellation | 31 valid
#define PIV_VALID   0x80000000u
#define PIV_NEG     (1u << 30)
#define PIV_ROUTE   (1u << 29)
#define PIV_AS_WORK (1u << 28)
#define PIV_HW      (1u << 27)
#define PIV_PT      (1u << 26)
#define PIV_CUT_SHIFT 13u
// slot_state of a persistent slot: records appended so far | listed as an item | every record in `ovf` (a school building: its
// rooms read its per-step counters only)
#define PSLOT_COUNT   0x00FFFFFFu
#define PSLOT_LISTED  0x40000000u
#define PSLOT_ALL_OVF 0x20000000u
struct ChunkT { int te0; int it; int t0; uint32_t n; };      // te0: the exposure step (+ TE_BIAS) that turns Infected in step 0 of the chunk
__device__ __forceinline__ ChunkT chunk_t(const Dev &d, uint32_t t0, uint32_t n)
{
    return ChunkT{ (int)(t0 + TE_BIAS) - (int)d.exposed_time - 1, (int)d.infected_time, (int)t0, n };
}
// The steps of the chunk in which the citizen of a persistent record stands where the record was left.
__device__ __forceinline__ M96 piv_steps(uint32_t iv, const ChunkT &ct, const M96 &AW, const M96 &BUS)
{
    if (!(iv & PIV_VALID)) return M96{ 0ull, 0u };
    int a = (int)(iv & 0x1FFFu) - ct.te0;
    const int b = min(a + ct.it, (int)ct.n - 1);
    if (iv & PIV_NEG) a = max(a, (int)((iv >> PIV_CUT_SHIFT) & 0x1FFFu) - ct.t0 + 1);
    a = max(a, 0);
    if (a > b) return M96{ 0ull, 0u };
    const M96 I = m96_range((uint32_t)a, (uint32_t)b);
    if (iv & PIV_ROUTE) return m96_and(I, BUS);
    const M96 rest = (iv & PIV_PT) ? m96_andn(I, BUS) : I;
    const M96 atw = (iv & PIV_HW) ? m96_and(rest, AW) : M96{ 0ull, 0u };
    return (iv & PIV_AS_WORK) ? atw : m96_andn(rest, atw);
}
__device__ __forceinline__ void piv_count(uint32_t iv, uint32_t lane, const ChunkT &ct, const M96 &AW, const M96 &BUS, uint32_t &c0, uint32_t &c1)
{
    const M96 at = piv_steps(iv, ct, AW, BUS);
    const uint32_t b0 = (uint32_t)(at.lo >> lane) & 1u, b1 = lane < 32u ? (at.hi >> lane) & 1u : 0u;
    if (iv & PIV_NEG) { c0 -= b0; c1 -= b1; } else { c0 += b0; c1 += b1; }     // (a cancellation never outruns what it cancels: sums stay >= 0)
}


// The map is emptied (a rebuild follows): the slots of every listed item, the spilled per-step counters, the in-map bits of
// everybody the log holds from the earliest exposure step that can still be Infected.  Ctrl::map_t = 0xFFFFFFFF says "empty".
__global__ __launch_bounds__(TPB) void k_map_clear(Dev d)
{
    Ctrl *ctrl = d.ctrl;
    const uint32_t tid = blockIdx.x * TPB + threadIdx.x, nth = gridDim.x * TPB;
    const uint32_t per = d.items_cap / SUBQ;
    for (uint32_t s = 0; s < SUBQ; ++s) {
        const uint32_t cnt = min(d.used_cnt[s], per);
        for (uint32_t k = tid; k < cnt; k += nth) {
            const uint32_t v = s * per + k, h = d.hitems[v];
            if (h >= d.hcap) continue;
            // (the per-step counters need no zeroing: on the persistent map k_map_fold STORES them before anybody reads them)
            d.hkey[h] = HKEY_EMPTY; d.slot_state[h] = 0u; d.hitems[v] = ITEM_UNUSED;
        }
    }
    const int lo_te = (int)(ctrl->t + TE_BIAS) - (int)d.exposed_time - 1 - (int)d.infected_time;
    const uint32_t i0 = d.log_off[lo_te < 0 ? 0 : lo_te], i1 = ctrl->log_len;
    for (uint32_t i = i0 + tid; i < i1; i += nth) { const uint32_t c = d.log[i], w = d.cit[c]; if (w & CW_IN_MAP) d.cit[c] = w & ~CW_IN_MAP; }   // (nothing else runs)
}
__global__ __launch_bounds__(64) void k_map_reset(Dev d)
{
    const uint32_t lane = threadIdx.x;
    d.used_cnt[lane] = 0u; d.pbig_cnt[lane] = 0u;
    if (lane == 0) { d.ctrl->map_t = 0xFFFFFFFFu; d.ctrl->n_neg = 0u; }
}

// One record of a citizen into one slot of the persistent map; returns nothing the caller has to wait for.  kind 0 home building,
// 1 work building, 2 room, 3 route; all_ovf: a school building (every record in `ovf`).  A first record in `ovf` lists the slot
// for k_map_fold; a cancellation record notes its address in Dev::neg_list.
__device__ __forceinline__ void map_list(const Dev &d, Ctrl *ctrl, uint32_t slot, uint32_t base2, uint32_t cap2, uint32_t sch, uint32_t wave)
{
    const uint32_t r = wave & (SUBQ - 1u), at = atomicAdd(&d.pbig_cnt[r], 1u), cap = d.big_qcap * 3u / PBIG_STRIDE;
    if (at < cap) { uint32_t *bl = d.big_list + ((size_t)r * cap + at) * PBIG_STRIDE; bl[0] = slot; bl[1] = base2; bl[2] = cap2 | (sch != 0xFFFFFFFFu ? 0x80000000u : 0u); bl[3] = sch; }
    else RAISE(ctrl, ESIM_ERANGE, ERR_AT_BIG_LIST);
}
__device__ __forceinline__ void map_append(const Dev &d, Ctrl *ctrl, uint32_t slot, uint32_t pos, uint32_t rec, uint32_t base2, uint32_t cap2, bool all_ovf,
                                            uint32_t wave, uint32_t neg_step)
{
    uint32_t where;
    if (!all_ovf && pos < ITEM_RECS) { where = slot * SLOT_IV_STRIDE + pos; d.slot_iv[where] = rec; }
    else {
        const uint32_t q = all_ovf ? pos : pos - ITEM_RECS;
        if (q >= cap2) { RAISE(ctrl, ESIM_ERANGE, ERR_AT_OVF_FULL); return; }     // (a member leaves one record and at most one cancellation per item)
        d.ovf[base2 + q] = rec;
        where = 0x80000000u | (base2 + q);
        if (q == 0u && !all_ovf) map_list(d, ctrl, slot, base2, cap2, 0xFFFFFFFFu, wave);     // (a school is listed by its first member, see k_map_enter)
    }
    if (rec & PIV_NEG) {
        const uint32_t i = atomicAdd(&ctrl->n_neg, 1u);
        if (i < NEG_CAP) { d.neg_list[2u * i] = where; d.neg_list[2u * i + 1u] = neg_step; } else RAISE(ctrl, ESIM_ERANGE, ERR_AT_NEG_LIST);
    }
}

// Find (or, with `insert`, make) the slot of a key in the open-addressing map.  Returns ITEM_UNUSED when it is not there.
__device__ __forceinline__ uint32_t map_slot(const Dev &d, Ctrl *ctrl, unsigned long long key, bool insert)
{
    uint32_t h = hash64(key) & (d.hcap - 1u);
    for (uint32_t probe = 0; probe < d.hcap; ++probe) {
        unsigned long long seen = d.hkey[h];
        if (seen == HKEY_EMPTY) { if (!insert) return ITEM_UNUSED; seen = atomicCAS(&d.hkey[h], HKEY_EMPTY, key); if (seen == HKEY_EMPTY) return h; }
        if (seen == key) return h;
        h = (h + 1u) & (d.hcap - 1u);
    }
    RAISE(ctrl, ESIM_ERANGE, ERR_AT_HASH_FULL);
    return ITEM_UNUSED;
}

// Where the records of key `kind` of citizen c go beyond the slot's own: (base, capacity) of its stretch of `ovf`, two places per
// member (a record and a cancellation).
__device__ __forceinline__ void map_ovf_range(const Dev &d, uint32_t kind, uint32_t id, uint32_t &base2, uint32_t &cap2)
{
    uint32_t base, cap;
    if (kind < 2u) { base = d.ovf_off[id]; cap = d.ovf_off[id + 1u] - base; }
    else if (kind == 2u) { const uint32_t o = d.room_off[id]; base = d.ovf_room_base + o; cap = d.room_off[id + 1u] - o; }
    else { const uint32_t o = d.route_off[id]; base = d.ovf_route_base + o; cap = d.route_off[id + 1u] - o; }
    base2 = 2u * base; cap2 = 2u * cap;
}

// generate_exposures for the persistent map: one LANE per citizen that TURNS Infected in the chunk (a rebuild: per citizen that is
// Infected in it at all).  The citizen leaves one record with each item it can ever stand in while Infected -- its home, its
// work building and room, its route -- whatever this chunk's schedule is: the records outlive the chunk.
__global__ __launch_bounds__(TPB) void k_map_enter(Dev d)
{
    Ctrl *ctrl = d.ctrl;
    const uint32_t t0 = ctrl->chunk_t0, n = ctrl->chunk_ok;
    if (!ctrl->chunk_parallel || n == 0u) return;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (blockIdx.x * TPB + threadIdx.x) >> 6, n_waves = (gridDim.x * TPB) >> 6;
    const uint32_t map_t = ctrl->map_t;
    const bool rebuild = map_t != t0;
    if (rebuild && map_t != 0xFFFFFFFFu) { if (wave == 0 && lane == 0) RAISE(ctrl, ESIM_ESTATE, ERR_AT_MAP_STATE); return; }   // (the host clears before a rebuild)
    {
        const uint32_t q = (t0 + MARK_SLOTS - 1u) & (MARK_SLOTS - 1u);        // marks a sequential step left (see k_chunk_marks)
        const uint32_t tid = blockIdx.x * TPB + threadIdx.x, nth = gridDim.x * TPB;
        const uint32_t ob = ctrl->n_touched_bld[q], orr = ctrl->n_touched_room[q], ort = ctrl->n_touched_route[q], orb = ctrl->n_touched_route_big[q];
        for (uint32_t i = tid; i < ob; i += nth) d.cnt_bld[q][d.touched_bld[q][i]] = 0u;
        for (uint32_t i = tid; i < orr; i += nth) d.cnt_room[q][d.touched_room[q][i]] = 0u;
        for (uint32_t i = tid; i < ort; i += nth) d.route_flag[q][d.touched_route[q][i]] = 0u;
        for (uint32_t i = tid; i < orb; i += nth) d.route_flag[q][d.touched_route_big[q][i]] = 0u;
    }
    const uint32_t per = d.items_cap / SUBQ;
    // Work building, room and route are entered only into a map that is built for a schedule with working hours: under a lockdown
    // (everybody at home for the whole chunk, Q8) a rebuild enters the homes alone -- half the items --, and a chunk whose schedule
    // needs the others cannot run on such a map: it is a no-op, the host sees no progress and rebuilds.
    const Decision q0 = lane < n ? d.dec[lane] : Decision{ 0u, 0u, 0u, 0u };
    const Decision q1 = 64u + lane < n ? d.dec[64u + lane] : Decision{ 0u, 0u, 0u, 0u };
    const bool needs_work = __ballot(lane < n && (q0.at_work | q0.bus_dir) != 0u) != 0ull || __ballot(64u + lane < n && (q1.at_work | q1.bus_dir) != 0u) != 0ull;
    const bool map_work = rebuild ? needs_work : ctrl->map_work != 0u;
    if (needs_work && !map_work) { if (wave == 0 && lane == 0) ctrl->chunk_parallel = 0u; return; }
    if (wave == 0 && lane == 0) { ctrl->items_per_wave = per; ctrl->n_items = per * SUBQ; ctrl->pmap_chunk = 1u; ctrl->map_work = map_work ? 1u : 0u; }
    const uint32_t i0 = rebuild ? ctrl->chunk_i0 : ctrl->chunk_e0, i1 = ctrl->chunk_i1;
    const ChunkT ct = chunk_t(d, t0, n);
    const unsigned long long lt = (1ull << lane) - 1ull;
    const uint32_t sub = wave & (SUBQ - 1u);
    // the plan's vaccinations of citizens the map holds already (k_chunk_vax_adj noted them), those inside the chunk
    if (!rebuild) {
        const uint32_t nc = min(ctrl->n_cancel, NEG_CAP);
        for (uint32_t i = blockIdx.x * TPB + threadIdx.x; i < nc; i += gridDim.x * TPB) {
            const uint32_t cz = d.cancel_list[2u * i], j = d.cancel_list[2u * i + 1u];
            if (j < n && cz < d.n) map_cancel(d, ctrl, cz, d.cit[cz], t0, j);
        }
    }
    WORK_TALLY;
    const uint32_t E = i1 - i0;
    for (uint32_t round = 0; wave + n_waves * (round * 64u) < E; ++round) {
        const uint32_t idx = wave + n_waves * (round * 64u + lane);
        bool act = idx < E;
        uint32_t c = 0u, w = 0u;
        if (act) { c = d.log[i0 + idx]; w = d.cit[c]; }
        const uint32_t te = CW_TE(w);
        const int a = (int)te - ct.te0;                                        // first Infected step, relative to the chunk
        // a vaccination the plan places inside this chunk (one beyond its end is not committed by it: the next plan decides again)
        const uint32_t vrel = CW_VAX_REL(w) < n ? CW_VAX_REL(w) : CW_VAX_NONE;
        // not (any more) Infected in this chunk; entered already; or Vaccinated (by the chunk's plan) before it would turn Infected
        if (te >= TE_RECOVERED || (w & CW_IN_MAP) || a > (int)n - 1 || a + ct.it < 0 || (vrel != CW_VAX_NONE && (int)vrel < a)) act = false;
        if (act) WORK_ADD(WK_ENTRIES, 1);
        const bool school = w & FL_WORK_SCHOOL, has_work = (w & FL_HAS_WORK) && map_work;
        uint32_t id[4] = { 0u, 0u, 0u, 0u };
        bool use[4] = { act, act && has_work, act && has_work && school, false };
        if (act) id[0] = d.home[c];
        if (use[1]) id[1] = d.work[c];
        if (use[2]) { id[2] = d.room[c]; if (id[2] == 0xFFFFFFFFu) use[2] = false; }
        if (act && map_work && (w & FL_USES_PT)) { id[3] = d.route_of[c]; use[3] = id[3] != NO_ROUTE; }
        // find or make the slots: the first probes of all four keys go out together (a look before the compare-and-swap)
        unsigned long long key[4], seen[4];
        uint32_t slot[4];
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k) {
            key[k] = (unsigned long long)id[k] + (k == 2u ? d.n_bld : k == 3u ? d.n_bld + d.n_room : 0u);
            slot[k] = hash64(key[k]) & (d.hcap - 1u);
            seen[k] = key[k];
            if (use[k]) { WORK_ADD(WK_KEYS, 1); seen[k] = d.hkey[slot[k]]; }
        }
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k) if (use[k] && seen[k] == HKEY_EMPTY) { seen[k] = atomicCAS(&d.hkey[slot[k]], HKEY_EMPTY, key[k]); if (seen[k] == HKEY_EMPTY) seen[k] = key[k]; }
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k)
            if (use[k] && seen[k] != key[k]) {                                  // somebody else's key in the slot: linear probing from the next one
                uint32_t h = slot[k];
                bool found = false;
                for (uint32_t probe = 1; probe < d.hcap && !found; ++probe) {
                    h = (h + 1u) & (d.hcap - 1u);
                    unsigned long long o = d.hkey[h];
                    if (o == HKEY_EMPTY) { o = atomicCAS(&d.hkey[h], HKEY_EMPTY, key[k]); if (o == HKEY_EMPTY) o = key[k]; }
                    found = o == key[k];
                }
                if (!found) { RAISE(ctrl, ESIM_ERANGE, ERR_AT_HASH_FULL); use[k] = false; }
                slot[k] = h;
            }
        // the record(s): the citizen's exposure step and what decides where it stands; a plan that vaccinates it at the end of step
        // vrel cancels it from the step after
        const uint32_t fl = PIV_VALID | te | ((w & FL_USES_PT) ? PIV_PT : 0u) | ((w & FL_HAS_WORK) ? PIV_HW : 0u);
        const bool neg = act && vrel != CW_VAX_NONE && (int)vrel < a + ct.it;   // (to the end of its stretch, whatever the chunk's length)
        const uint32_t cut = neg ? ((t0 + vrel) << PIV_CUT_SHIFT) | PIV_NEG : 0u;
        // a school building keeps no record per member: the member joins the school's histograms of exposure steps (k_map_fold
        // sums windows of them); only a cancellation is a record there
        const bool ring = use[1] && school;
        int32_t sch = -1;
        if (ring) {
            sch = d.sch_of_bld[id[1]];
            if (sch < 0) { RAISE(ctrl, ESIM_ESTATE, ERR_AT_ITEM_CHECK); use[1] = false; }
            else {
                atomicAdd(&d.sch_ring[((size_t)sch * 2u) * SCH_RING + (te & (SCH_RING - 1u))], 1u);
                if (w & FL_USES_PT) atomicAdd(&d.sch_ring[((size_t)sch * 2u + 1u) * SCH_RING + (te & (SCH_RING - 1u))], 1u);
            }
        }
        uint32_t old[4] = { 0u, 0u, 0u, 0u };
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k)
            if (use[k]) {
                // (a school's members -- hundreds in a chunk -- all meet in ONE slot: they look at it with a plain load, and only a
                // cancellation, which needs a place, or whoever sees it unlisted pays for a returning atomic)
                if (k == 1u && ring) old[k] = neg ? atomicAdd(&d.slot_state[slot[k]], 1u) : d.slot_state[slot[k]];
                else old[k] = atomicAdd(&d.slot_state[slot[k]], neg ? 2u : 1u);
            }
        // whoever finds a slot that is not listed as an item lists it: the next ids of this wavefront's sub-list
        bool lists[4];
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k) {
            lists[k] = false;
            if (use[k] && !(old[k] & PSLOT_LISTED)) lists[k] = !(atomicOr(&d.slot_state[slot[k]], PSLOT_LISTED) & PSLOT_LISTED);
        }
        uint32_t before = 0u, n_claims = 0u;
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k) { const unsigned long long cm = __ballot(lists[k]); before += (uint32_t)__popcll(cm & lt); n_claims += (uint32_t)__popcll(cm); }
        uint32_t first_id = 0u;
        if (n_claims) {
            if (lane == 0) first_id = atomicAdd(&d.used_cnt[sub], n_claims);
            first_id = FX(first_id, 0);
            if (first_id + n_claims > per) { if (lane == 0) RAISE(ctrl, ESIM_ERANGE, ERR_AT_ITEM_IDS); n_claims = 0u; }
        }
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k) {
            if (lists[k] && n_claims) {
                WORK_ADD(WK_CLAIMS, 1);
                const uint32_t v = sub * per + first_id + before++;
                d.hitems[v] = slot[k];
                ItemRec rec = { (uint32_t)key[k], 0u, 0u, 0u, 0u, 0u, k == 2u ? slot[1] : 0xFFFFFFFFu, 0u };
                if (k < 2u) {
                    rec.a_lo = d.res_off[id[k]]; rec.a_hi = d.res_off[id[k] + 1u];
                    rec.b_lo = d.wrk_off[id[k]]; rec.b_hi = d.wrk_off[id[k] + 1u];
                    rec.aux = (uint32_t)d.bld_type[id[k]];
                } else if (k == 2u) { rec.a_lo = d.room_off[id[2]]; rec.a_hi = d.room_off[id[2] + 1u]; rec.aux = d.room_bld[id[2]]; }
                else { rec.a_lo = d.route_off[id[3]]; rec.a_hi = d.route_off[id[3] + 1u]; }
                d.item_rec[v] = rec;
            }
        }
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k) {
            if (!use[k]) continue;
            uint32_t base2, cap2;
            map_ovf_range(d, k, id[k], base2, cap2);
            const uint32_t rec = fl | (k == 3u ? PIV_ROUTE : (k == 0u ? 0u : PIV_AS_WORK));
            const uint32_t pos = old[k] & PSLOT_COUNT;
            if (k == 1u && ring) {
                if (lists[k]) map_list(d, ctrl, slot[k], base2, cap2, (uint32_t)sch, wave);      // (who lists the item lists the school for the fold)
                if (neg) map_append(d, ctrl, slot[k], pos, rec | cut, base2, cap2, true, wave, vrel);
                continue;
            }
            WORK_ADD(WK_RECORDS, 1);
            map_append(d, ctrl, slot[k], pos, rec, base2, cap2, false, wave, 0u);
            if (neg) map_append(d, ctrl, slot[k], pos + 1u, rec | cut, base2, cap2, false, wave, vrel);
        }
        if (act) d.cit[c] = w | CW_IN_MAP;                                    // (nobody else writes the word of a citizen that turns Infected here)
    }
    WORK_FLUSH(d);
}

// A vaccination planned for the end of step j of the chunk, of a citizen whose records the map holds already: a cancellation record
// from step j + 1 on with each of its items (k_chunk_vax_adj calls this for the citizens whose own step won).
__device__ __forceinline__ void map_cancel(const Dev &d, Ctrl *ctrl, uint32_t c, uint32_t w, uint32_t t0, uint32_t j)
{
    const uint32_t te = CW_TE(w);
    const bool mw = ctrl->map_work != 0u;                                     // (a map built under a lockdown holds the homes alone)
    const bool school = w & FL_WORK_SCHOOL, has_work = (w & FL_HAS_WORK) && mw;
    const uint32_t fl = PIV_VALID | PIV_NEG | ((t0 + j) << PIV_CUT_SHIFT) | te | ((w & FL_USES_PT) ? PIV_PT : 0u) | ((w & FL_HAS_WORK) ? PIV_HW : 0u);
    uint32_t id[4] = { d.home[c], has_work ? d.work[c] : 0u, (has_work && school) ? d.room[c] : 0xFFFFFFFFu, (mw && (w & FL_USES_PT)) ? d.route_of[c] : NO_ROUTE };
    const bool use[4] = { true, has_work, has_work && school && id[2] != 0xFFFFFFFFu, id[3] != NO_ROUTE };
    for (uint32_t k = 0; k < 4u; ++k) {
        if (!use[k]) continue;
        const uint32_t slot = map_slot(d, ctrl, (unsigned long long)id[k] + (k == 2u ? d.n_bld : k == 3u ? d.n_bld + d.n_room : 0u), false);
        if (slot == ITEM_UNUSED) { RAISE(ctrl, ESIM_ESTATE, ERR_AT_CANCEL_SLOT); continue; }     // (its record is there: CW_IN_MAP)
        uint32_t base2, cap2;
        map_ovf_range(d, k, id[k], base2, cap2);
        const uint32_t pos = atomicAdd(&d.slot_state[slot], 1u) & PSLOT_COUNT;
        map_append(d, ctrl, slot, pos, fl | (k == 3u ? PIV_ROUTE : (k == 0u ? 0u : PIV_AS_WORK)), base2, cap2, k == 1u && school, (blockIdx.x * blockDim.x + threadIdx.x) >> 6, j);
    }
}

// generate_exposures (simulator.rs:181-198) for every step of the chunk: one LANE per citizen that is Infected somewhere in
// the chunk.
__global__ __launch_bounds__(TPB) void k_chunk_marks(Dev d)
{
    Ctrl *ctrl = d.ctrl;
    const uint32_t t0 = ctrl->chunk_t0, n = ctrl->chunk_ok;
    if (!ctrl->chunk_parallel || n == 0u) return;
    {
        // the marks of step t0 - 1 (made by a sequential or pipelined step) would have been cleared by the exposure
        // pass of step t0; this chunk has none, so clear them here
        const uint32_t q = (t0 + MARK_SLOTS - 1u) & (MARK_SLOTS - 1u);
        const uint32_t tid = blockIdx.x * TPB + threadIdx.x, nth = gridDim.x * TPB;
        const uint32_t ob = ctrl->n_touched_bld[q], orr = ctrl->n_touched_room[q], ort = ctrl->n_touched_route[q], orb = ctrl->n_touched_route_big[q];
        for (uint32_t i = tid; i < ob; i += nth) d.cnt_bld[q][d.touched_bld[q][i]] = 0u;
        for (uint32_t i = tid; i < orr; i += nth) d.cnt_room[q][d.touched_room[q][i]] = 0u;
        for (uint32_t i = tid; i < ort; i += nth) d.route_flag[q][d.touched_route[q][i]] = 0u;
        for (uint32_t i = tid; i < orb; i += nth) d.route_flag[q][d.touched_route_big[q][i]] = 0u;
    }
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (blockIdx.x * TPB + threadIdx.x) >> 6, n_waves = (gridDim.x * TPB) >> 6;
    const uint32_t i0 = ctrl->chunk_i0, i1 = ctrl->chunk_i1;                 // log slice of the chunk's Infected (k_decide)
    // every wavefront owns a fixed range of item ids (a citizen claims at most four items), so no counter is shared
    uint32_t n_remote = 0u;
    if (d.world > 1u) for (uint32_t r = 0; r < d.world; ++r) if (r != d.rank) n_remote += min(d.xs[(size_t)r * (1u + 3u * d.xs_cap)], d.xs_cap);
    const uint32_t per_wave = 4u * ((i1 - i0 + n_remote + n_waves - 1u) / n_waves + (n_remote ? 1u : 0u));
    if (wave == 0 && lane == 0) { ctrl->items_per_wave = per_wave; ctrl->n_items = per_wave * n_waves; }
    if ((unsigned long long)per_wave * n_waves > d.items_cap) { if (lane == 0) ctrl->error = (uint32_t)(-ESIM_ERANGE); return; }
    uint32_t next_id = wave * per_wave;
    const Decision q0 = lane < n ? d.dec[lane] : Decision{ 0u, 0u, 0u, 0u };
    const Decision q1 = 64u + lane < n ? d.dec[64u + lane] : Decision{ 0u, 0u, 0u, 0u };
    // The chunk's schedule as step masks (wavefront-uniform): the steps in which those with a work place are at work, and the
    // steps in which riders are on a bus (at most CHUNK_BUS_STEPS, k_decide: a route item keeps one bit per such step -- "an
    // Infected rider of this route has registered the (route, step) pair").
    const M96 AW = { __ballot(lane < n && q0.at_work != 0u), (uint32_t)__ballot(64u + lane < n && q1.at_work != 0u) };
    const M96 BUS = { __ballot(lane < n && q0.bus_dir != 0u), (uint32_t)__ballot(64u + lane < n && q1.bus_dir != 0u) };
    const unsigned long long lt = (1ull << lane) - 1ull;
    const uint32_t pm0 = PROF_NOW();
    WORK_TALLY;
    uint32_t p_entries = 0u;
    uint32_t my_pairs = 0u;                                                   // (route, bus step) pairs this wavefront registered
    uint32_t ps[5] = { 0u, 0u, 0u, 0u, 0u };                                   // diagnostics: time per stage
    // ONE LANE PER INFECTED CITIZEN: the pass is a chain of dependent round trips (log entry -> word and keys -> hash claim ->
    // the item's lists / a record position -> the record), so what it needs is requests in flight, not lanes per citizen.
    // Where a citizen stands in each step follows from its Infected stretch and the schedule masks with a few 96-bit
    // operations.  Entry idx of the chunk's log slice (then of the commuters the other shards sent) belongs to wavefront
    // idx % n_waves -- every wavefront gets the same share, whatever the number of entries, and with it the same share of item
    // ids and of the draw pass's work.
    const uint32_t E = i1 - i0, total = E + n_remote;
    for (uint32_t round = 0; wave + n_waves * (round * 64u) < total; ++round) {
        const uint32_t pa = PROF_NOW();
        const uint32_t idx = wave + n_waves * (round * 64u + lane);
        bool act = idx < total;
        const bool remote = act && idx >= E;
        uint32_t c = 0u, w = 0u, r_bld = 0xFFFFFFFFu, r_room = 0xFFFFFFFFu;
        if (act && !remote) { c = d.log[i0 + idx]; w = d.cit[c]; }
        if (remote) {
            // Sharded: the Infected commuters the other shards sent (k_shared_pack, all-gathered): each stands in a building
            // (and room) that has members here too; it enters the map like a local citizen's work building and room.
            uint32_t e = idx - E;
            const uint32_t *seg = nullptr;
            for (uint32_t r = 0; r < d.world; ++r) {
                if (r == d.rank) continue;
                const uint32_t *sg = d.xs + (size_t)r * (1u + 3u * d.xs_cap);
                const uint32_t cnt = min(sg[0], d.xs_cap);
                if (e < cnt) { seg = sg; break; }
                e -= cnt;
            }
            act = false;
            if (seg) {
                w = seg[1u + 3u * e];
                const uint32_t sb = seg[2u + 3u * e], sr = seg[3u + 3u * e];
                const int32_t lb = d.shared_bld[sb];
                if (lb >= 0) {                                                // (else nobody of that building lives here)
                    act = true;
                    r_bld = (uint32_t)lb;
                    const int32_t lr = sr != 0xFFFFFFFFu ? d.shared_room[sr] : -1;
                    if (lr >= 0) r_room = (uint32_t)lr;                       // (a remote commuter's room may have no member here)
                }
            }
        }
        // the citizen is Infected in steps [a, b] of the chunk (one stretch: disease.rs:60-65), Vaccinated after the step its
        // plan names (k_chunk_vax); where it stands in each of them (simulator.rs:181-198) follows from its flags and the schedule
        const int a_abs = (int)CW_TE(w) - (int)TE_BIAS + (int)d.exposed_time + 1;
        const int b_rel = a_abs + (int)d.infected_time - (int)t0;
        const uint32_t iv_a = a_abs > (int)t0 ? (uint32_t)(a_abs - (int)t0) : 0u;
        const uint32_t iv_b = b_rel < 0 ? 0u : min(min((uint32_t)b_rel, n - 1u), CW_VAX_REL(w));   // (CW_VAX_NONE is the largest value)
        if (CW_TE(w) >= TE_RECOVERED || b_rel < 0 || iv_a > iv_b) act = false;
        M96 I = m96_range(iv_a, iv_b);
        if (!act) I = M96{ 0ull, 0u };
        const M96 onbus = (w & FL_USES_PT) ? m96_and(I, BUS) : M96{ 0ull, 0u };
        const M96 rest = m96_andn(I, onbus);
        const M96 atw = (w & FL_HAS_WORK) ? m96_and(rest, AW) : M96{ 0ull, 0u };
        const M96 ath = m96_andn(rest, atw);
        // (a commuter from another shard only counts where it works: its home and its route are its own shard's business)
        const bool any_home = !remote && m96_any(ath), any_work = m96_any(atw), any_bus = !remote && m96_any(onbus);
        const bool school = w & FL_WORK_SCHOOL;
        if (any_home || any_work || any_bus) { ++p_entries; WORK_ADD(WK_ENTRIES, 1); }
        // the four keys: home building, work building, room, route
        uint32_t src[4] = { 0u, r_bld, r_room, 0u };
        if (!remote && (any_home || any_work || any_bus)) {
            const uint4 k4 = d.where4[c];                                      // (one request for the four)
            src[0] = k4.x; src[1] = k4.y; src[2] = k4.z; src[3] = k4.w;
        }
        unsigned long long key[4];
        key[0] = any_home ? (unsigned long long)src[0] : HKEY_EMPTY;
        key[1] = any_work ? (unsigned long long)src[1] : HKEY_EMPTY;
        key[2] = (any_work && school && src[2] != 0xFFFFFFFFu) ? (unsigned long long)d.n_bld + src[2] : HKEY_EMPTY;
        key[3] = any_bus ? (unsigned long long)d.n_bld + d.n_room + src[3] : HKEY_EMPTY;
        const uint32_t pb = PROF_NOW() + (uint32_t)(key[0] & 0ull) + (uint32_t)(key[1] & 0ull) + (uint32_t)(key[2] & 0ull) + (uint32_t)(key[3] & 0ull);
        // claim or find the items: the first probes of all four keys go out together
        uint32_t slot[4];
        unsigned long long seen[4];
        // (a look before the CAS: the hundreds of Infected of one school all ask for the same key, and compare-and-swaps on one
        // address are served one after the other, loads are not -- a stale "empty" only costs the CAS it would have cost anyway)
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k) {
            slot[k] = hash64(key[k]) & (d.hcap - 1u);
            seen[k] = key[k];
            if (key[k] != HKEY_EMPTY) seen[k] = d.hkey[slot[k]];
        }
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k)
            if (key[k] != HKEY_EMPTY && seen[k] == HKEY_EMPTY) seen[k] = atomicCAS(&d.hkey[slot[k]], HKEY_EMPTY, key[k]);
        bool claimed[4], pending[4];
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k) {
            claimed[k] = false; pending[k] = false;
            if (key[k] == HKEY_EMPTY) continue;
            WORK_ADD(WK_KEYS, 1);
            if (seen[k] == HKEY_EMPTY) claimed[k] = true;
            else if (seen[k] == key[k]) pending[k] = true;
            else {
                // somebody else's key in the slot: linear probing
                uint32_t h = slot[k];
                bool found = false;
                for (uint32_t probe = 1; probe < d.hcap && !found; ++probe) {
                    h = (h + 1u) & (d.hcap - 1u);
                    const unsigned long long old = atomicCAS(&d.hkey[h], HKEY_EMPTY, key[k]);
                    if (old == HKEY_EMPTY) { claimed[k] = true; found = true; }
                    else if (old == key[k]) { pending[k] = true; found = true; }
                }
                if (!found) { ctrl->error = (uint32_t)(-ESIM_ERANGE); key[k] = HKEY_EMPTY; h = 0u; }
                slot[k] = h;
            }
        }
        const uint32_t pc = PROF_NOW() + (slot[0] & 0u) + (slot[1] & 0u) + (slot[2] & 0u) + (slot[3] & 0u);
        const uint32_t iv_home = IV_VALID | iv_a | (iv_b << 7) | ((w & FL_USES_PT) ? IV_PT : 0u) | ((w & FL_HAS_WORK) ? IV_HW : 0u);
        const uint32_t iv_work = iv_home | IV_AS_WORK;
        // the claimers take the next ids of this wavefront's range and write what the draw pass needs of the item; the claimer's
        // own stretch travels in it (a school's counts are looked up by slot from its rooms, so even its claimer's stretch goes
        // into a slot record), and a room's record names the slot of its school (the citizen's work building)
        // (ids in citizen order, a citizen's items side by side: the draw pass walks a wavefront's items in id order, and a
        // mix of short and long member lists along the way keeps its load even)
        uint32_t before = 0u, n_claims = 0u;
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k) { const unsigned long long cm = __ballot(claimed[k]); before += (uint32_t)__popcll(cm & lt); n_claims += (uint32_t)__popcll(cm); }
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k) {
            if (claimed[k]) {
                WORK_ADD(WK_CLAIMS, 1);
                const uint32_t v = next_id + before++;
                d.hitems[v] = slot[k];
                const uint32_t id = (uint32_t)key[k];
                ItemRec rec = { id, 0u, 0u, 0u, 0u, 0u, k == 2u ? slot[1] : 0xFFFFFFFFu,
                                k == 0u ? iv_home : (k == 1u && !school) || k == 2u ? iv_work : 0u };
                if (k == 0u || k == 1u) {
                    const uint4 b4 = *reinterpret_cast<const uint4 *>(&d.bld8[id]);
                    rec.a_lo = b4.x; rec.a_hi = b4.y; rec.b_lo = b4.z; rec.b_hi = b4.w;
                    rec.aux = d.bld8[id].type;
                } else if (k == 2u) {
                    const uint32_t r = id - d.n_bld;
                    rec.a_lo = d.room_off[r]; rec.a_hi = d.room_off[r + 1u];
                    rec.aux = d.room_bld[r];
                }
                d.item_rec[v] = rec;
            }
        }
        next_id += n_claims;
        const uint32_t pd = PROF_NOW();
        // Somebody else's building / room: my stretch goes into one of the slot's ITEM_RECS records; when they are taken,
        // into its per-step counters (`vec`), one atomic per step.  The route: which of my bus steps nobody has registered yet.
        uint32_t mine = 0u;                                                   // bit i: I ride, Infected, in the i-th bus step of the chunk
        if (any_bus) {
            uint32_t i = 0u;
            for (unsigned long long m = BUS.lo; m; m &= m - 1ull, ++i) mine |= (uint32_t)((onbus.lo >> __builtin_ctzll(m)) & 1ull) << i;
            for (uint32_t m = BUS.hi; m; m &= m - 1u, ++i) mine |= ((onbus.hi >> __builtin_ctz(m)) & 1u) << i;
        }
        bool add_rec[3];
        uint32_t old[4] = { 0u, 0u, 0u, 0u };
#pragma unroll
        for (uint32_t k = 0; k < 3u; ++k) {
            // (a school's Infected -- hundreds -- do not queue for record positions: they count themselves, see below)
            add_rec[k] = pending[k] && !(k == 1u && school);
            if (add_rec[k]) old[k] = atomicAdd(&d.slot_state[slot[k]], 1u);
        }
        if (claimed[1] && school) d.slot_state[slot[1]] = SLOT_COUNTERS_ONLY;     // (tells item_counts and the clean-up)
        if (any_bus && key[3] != HKEY_EMPTY) old[3] = atomicOr(&d.slot_state[slot[3]], mine);
        // positions beyond ITEM_RECS: the record goes into the building's / room's own stretch of `ovf` (one place per member, so
        // it cannot run out -- except for commuters from other shards, who are no members here: those add themselves to the
        // slot's per-step counters, one atomic per step, which nobody has to wait for; so does everybody in a school
        // building); the first to get there lists the slot for k_chunk_fold
        bool direct[3], first[3];
        uint32_t first_base[3] = { 0u, 0u, 0u }, first_cap[3] = { 0u, 0u, 0u };
        if (school && key[1] != HKEY_EMPTY) {
            // a school building: my stretch goes into the school's difference arrays (two atomics, four for a rider, whatever
            // the number of steps); whoever claimed the building's item lists it for k_chunk_fold
            const int32_t sch = d.sch_of_bld[(uint32_t)key[1]];
            if (sch < 0 || (uint32_t)sch >= d.n_sch) RAISE(ctrl, ESIM_ERANGE, ERR_AT_SCHOOL);
            else {
                uint32_t *dd = d.sch_diff + ((size_t)sch * SD_REPL + (wave & (SD_REPL - 1u))) * 2u * FREE_MAX;
                WORK_ADD(WK_DIRECT, (w & FL_USES_PT) ? 4 : 2);
                atomicAdd(&dd[iv_a], 1u);
                if (iv_b + 1u < FREE_MAX) atomicAdd(&dd[iv_b + 1u], 0xFFFFFFFFu);
                if (w & FL_USES_PT) {
                    atomicAdd(&dd[FREE_MAX + iv_a], 1u);
                    if (iv_b + 1u < FREE_MAX) atomicAdd(&dd[FREE_MAX + iv_b + 1u], 0xFFFFFFFFu);
                }
                if (claimed[1]) { first_base[1] = (uint32_t)sch; first_cap[1] = 0xFFFFFFFFu; }
            }
        }
#pragma unroll
        for (uint32_t k = 0; k < 3u; ++k) {
            direct[k] = false; first[k] = k == 1u && first_cap[1] == 0xFFFFFFFFu;
            if (!add_rec[k]) continue;
            WORK_ADD(WK_RECORDS, 1);
            const uint32_t iv = k == 0u ? iv_home : iv_work;
            if (old[k] < ITEM_RECS) { d.slot_iv[(size_t)slot[k] * SLOT_IV_STRIDE + old[k]] = iv; continue; }
            const uint32_t q = old[k] - ITEM_RECS, id = (uint32_t)key[k];
            uint32_t base, cap;
            if (k < 2u) { base = d.bld8[id].ovf_lo; cap = d.bld8[id].ovf_hi - base; }
            else { const uint32_t r = id - d.n_bld; const uint32_t o = d.room_off[r]; base = d.ovf_room_base + o; cap = d.room_off[r + 1u] - o; }
            if (q < cap) { d.ovf[base + q] = iv; first[k] = q == 0u; first_base[k] = base; first_cap[k] = cap; }
            else direct[k] = true;
        }
        {
            // (one reservation in this wavefront's list for everything the 64 citizens list)
            const unsigned long long f0 = __ballot(first[0]), f1 = __ballot(first[1]), f2 = __ballot(first[2]);
            const uint32_t n_first = (uint32_t)(__popcll(f0) + __popcll(f1) + __popcll(f2));
            if (n_first) {
                const uint32_t r = wave & (SUBQ - 1u);
                uint32_t at0 = 0u;
                if (lane == 0) at0 = atomicAdd(&d.hot[(HOT_BIG + r) * HOT_STRIDE], n_first);
                at0 = __shfl(at0, 0, 64);
                uint32_t *bl = d.big_list + (size_t)r * d.big_qcap * 3u;
                uint32_t pos = at0 + (uint32_t)__popcll(f0 & lt);
                if (first[0] && pos < d.big_qcap) { bl[3u * pos] = slot[0]; bl[3u * pos + 1u] = first_base[0]; bl[3u * pos + 2u] = first_cap[0]; }
                pos = at0 + (uint32_t)__popcll(f0) + (uint32_t)__popcll(f1 & lt);
                if (first[1] && pos < d.big_qcap) { bl[3u * pos] = slot[1]; bl[3u * pos + 1u] = first_base[1]; bl[3u * pos + 2u] = first_cap[1]; }
                pos = at0 + (uint32_t)__popcll(f0) + (uint32_t)__popcll(f1) + (uint32_t)__popcll(f2 & lt);
                if (first[2] && pos < d.big_qcap) { bl[3u * pos] = slot[2]; bl[3u * pos + 1u] = first_base[2]; bl[3u * pos + 2u] = first_cap[2]; }
                if (at0 + n_first > d.big_qcap && lane == 0) ctrl->error = (uint32_t)(-ESIM_ERANGE);
            }
        }
#pragma unroll
        for (uint32_t k = 0; k < 3u; ++k) {
            const M96 at = k == 0u ? ath : atw;
            for (unsigned long long sp = __ballot(direct[k]); sp; sp &= sp - 1ull) {
                const int src_lane = __builtin_ctzll(sp);
                const uint32_t sl = __shfl(slot[k], src_lane, 64), hi = __shfl(at.hi, src_lane, 64);
                const unsigned long long lo = ((unsigned long long)__shfl((uint32_t)(at.lo >> 32), src_lane, 64) << 32) | __shfl((uint32_t)at.lo, src_lane, 64);
                uint32_t *v = d.vec + (size_t)sl * FREE_MAX;
                WORK_ADD(WK_DIRECT, ((lo >> lane) & 1ull) + ((lane < 32u && ((hi >> lane) & 1u)) ? 1 : 0));
                if ((lo >> lane) & 1ull) atomicAdd(&v[lane], 1u);
                if (lane < 32u && ((hi >> lane) & 1u)) atomicAdd(&v[64u + lane], 1u);
            }
        }
        const uint32_t pe = PROF_NOW() + (old[0] & 0u) + (old[1] & 0u) + (old[2] & 0u) + (old[3] & 0u);
        // register the (route, step) pairs that are new: k_chunk_draw ranks the riders of each once
        const uint32_t new_bits = (any_bus && key[3] != HKEY_EMPTY) ? (mine & ~old[3]) : 0u;
        if (__any(new_bits != 0u)) {
            const bool big = w & FL_BIG_ROUTE;
            const uint32_t rt = src[3];                                       // the route itself, not its item: saves the pass a hop
            // routes of few riders: this wavefront's own stretch of the list, no shared counter; the others share one
            const uint32_t K = PAIR_K(per_wave, ctrl->chunk_bus);
            uint32_t *list = d.route_pairs + (size_t)wave * K;
            uint32_t n_big = 0u;
            for (uint32_t i = 0; i < CHUNK_BUS_STEPS; ++i) n_big += (uint32_t)__popcll(__ballot(big && ((new_bits >> i) & 1u)));
            uint32_t big_base = 0u;
            if (n_big) {
                if (lane == 0) big_base = atomicAdd(&d.hot[HOT_BIGPAIRS * HOT_STRIDE], n_big);
                big_base = __shfl(big_base, 0, 64);
                if (big_base + n_big > 2u * d.items_cap) { if (lane == 0) ctrl->error = (uint32_t)(-ESIM_ERANGE); n_big = 0u; big_base = 0xFFFFFFFFu; }
            }
            uint32_t i = 0u;
            auto put = [&](uint32_t j) {
                const bool f = (new_bits >> i) & 1u;
                const unsigned long long ms = __ballot(f && !big), mb = __ballot(f && big);
                if (f && !big) {
                    const uint32_t pos = my_pairs + (uint32_t)__popcll(ms & lt);
                    if (pos < K) list[pos] = (rt << 7) | j; else ctrl->error = (uint32_t)(-ESIM_ERANGE);
                }
                if (f && big && big_base != 0xFFFFFFFFu) d.route_pairs_big[big_base + (uint32_t)__popcll(mb & lt)] = (rt << 7) | j;
                my_pairs += (uint32_t)__popcll(ms);
                if (big_base != 0xFFFFFFFFu) big_base += (uint32_t)__popcll(mb);
                ++i;
            };
            for (unsigned long long m = BUS.lo; m; m &= m - 1ull) put((uint32_t)__builtin_ctzll(m));
            for (uint32_t m = BUS.hi; m; m &= m - 1u) put(64u + (uint32_t)__builtin_ctz(m));
            my_pairs = min(my_pairs, K);
        }
        { const uint32_t pf = PROF_NOW(); ps[0] += pb - pa; ps[1] += pc - pb; ps[2] += pd - pc; ps[3] += pe - pd; ps[4] += pf - pe; }
    }
    if (lane == 0) { d.pair_cnt[wave] = my_pairs; d.used_cnt[wave] = next_id - wave * per_wave; }
    WORK_FLUSH(d);
    const uint32_t pm1 = PROF_NOW();
    PROF_PUT(d, 8, pm0); PROF_PUT(d, 9, pm1); PROF_PUT(d, 10, p_entries);
    PROF_PUT(d, 11, ps[0]); PROF_PUT(d, 12, ps[1]); PROF_PUT(d, 13, ps[2]); PROF_PUT(d, 14, ps[3]); PROF_PUT(d, 15, ps[4]);
    (void)p_entries; (void)ps;
}

// The interval records k_chunk_marks put into `ovf` (slots with more than ITEM_RECS of them: schools, large work places),
// summed into the slots' per-step counters before the draw pass reads them.  One wavefront per listed slot, 64 records at a
// time; each lane turns its record into the set of steps in which that citizen stands there, one ballot per step counts them.
// PM: the persistent map -- the listed slots are those of Dev::pbig_cnt (they stay listed), their records are absolute, the
// sums are STORED (nobody else adds to them), and the prefix sums are those of the SUBQ sub-lists the item ids are handed out from.
template <bool PM>
__global__ __launch_bounds__(TPB) void k_chunk_fold(Dev d)
{
    Ctrl *ctrl = d.ctrl;
    const uint32_t n = ctrl->chunk_ok;
    if (!ctrl->chunk_parallel || n == 0u) return;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (blockIdx.x * TPB + threadIdx.x) >> 6, n_waves = (gridDim.x * TPB) >> 6;
    const uint32_t per_wave = ld(&ctrl->items_per_wave);
    const uint32_t n_own = PM ? SUBQ : n_waves;                              // owners of item id ranges
    if ((unsigned long long)per_wave * n_own > d.items_cap) return;
    if (blockIdx.x == 0) {
        // for k_chunk_draw: used_pref[k] = ids handed out by the wavefronts before k
        __shared__ uint32_t s_pref[CHUNK_WAVES_MAX + 1u];
        __shared__ uint32_t s_wtot[TPB / 64];
        const uint32_t pf0 = PROF_NOW();
        constexpr uint32_t RUN = CHUNK_WAVES_MAX / TPB;
        const uint32_t run = (n_own + TPB - 1u) / TPB, b = threadIdx.x * run;   // <= RUN
        uint32_t cnt[RUN];
        uint32_t sum = 0u;
#pragma unroll
        for (uint32_t k = 0; k < RUN; ++k) { cnt[k] = (k < run && b + k < n_own) ? min(d.used_cnt[b + k], per_wave) : 0u; sum += cnt[k]; }
        uint32_t x = sum;
        for (uint32_t o = 1; o < 64u; o <<= 1) { const uint32_t y = __shfl_up(x, o, 64); if (lane >= o) x += y; }
        if (lane == 63u) s_wtot[threadIdx.x >> 6] = x;
        if (threadIdx.x == 0) s_pref[0] = 0u;
        __syncthreads();
        const uint32_t pf1 = PROF_NOW();
        uint32_t acc = x - sum;
        for (uint32_t k = 0; k < (threadIdx.x >> 6); ++k) acc += s_wtot[k];
#pragma unroll
        for (uint32_t k = 0; k < RUN; ++k) if (k < run && b + k < n_own) { acc += cnt[k]; s_pref[b + k + 1u] = acc; }
        __syncthreads();
        const uint32_t pf2 = PROF_NOW();
        for (uint32_t w = threadIdx.x; w <= n_own; w += TPB) d.used_pref[w] = s_pref[w];
        BOOKS_PROF(d, 8, pf1 - pf0); BOOKS_PROF(d, 9, pf2 - pf1); BOOKS_PROF(d, 10, PROF_NOW() - pf2);
        (void)pf0; (void)pf1; (void)pf2;
    }
    // list `wave & 63`, every (n_waves / 64)-th entry of it
    const uint32_t qr = wave & (SUBQ - 1u), first = wave >> 6, step = n_waves >> 6;
    const uint32_t l_stride = PM ? PBIG_STRIDE : 3u, l_cap = PM ? d.big_qcap * 3u / PBIG_STRIDE : d.big_qcap;
    const uint32_t n_list = step ? min(PM ? d.pbig_cnt[qr] : ld(&d.hot[(HOT_BIG + qr) * HOT_STRIDE]), l_cap) : 0u;
    const ChunkT ct = chunk_t(d, ctrl->chunk_t0, n);
#ifdef ESIM_PROFILE_FOLD
    const uint32_t pq0 = PROF_NOW();
    uint32_t pq_rec = 0u, pq_n = 0u;
    PROF_PUT(d, 11, 0u); PROF_PUT(d, 12, 0u); PROF_PUT(d, 13, 0u);
#endif
    if (first >= n_list) return;
    WORK_TALLY;
    const uint32_t *bl = d.big_list + (size_t)qr * l_cap * l_stride;
    __shared__ uint32_t s_win[TPB / 64][2][TE_BIAS + FREE_MAX + 64u];        // (persistent map: a school's window sums, per wavefront)
    const Decision q0 = lane < n ? d.dec[lane] : Decision{ 0u, 0u, 0u, 0u };
    const Decision q1 = 64u + lane < n ? d.dec[64u + lane] : Decision{ 0u, 0u, 0u, 0u };
    const M96 AW = { __ballot(lane < n && q0.at_work != 0u), (uint32_t)__ballot(64u + lane < n && q1.at_work != 0u) };
    const M96 BUS = { __ballot(lane < n && q0.bus_dir != 0u), (uint32_t)__ballot(64u + lane < n && q1.bus_dir != 0u) };
    for (uint32_t g = first; g < n_list; g += 64u * step) {
        // 64 of this wavefront's entries at a time, a lane each: the slot, where its records are and how many (those that
        // did not fit counted themselves)
        uint32_t slot_l = 0u, base_l = 0u, n_ov_l = 0u, sch_l = 0xFFFFFFFFu;
        const uint32_t mine = g + lane * step;
        if (mine < n_list) {
            slot_l = bl[l_stride * mine]; base_l = bl[l_stride * mine + 1u];
            const uint32_t cap_l = bl[l_stride * mine + 2u] & 0x7FFFFFFFu, all_ovf = bl[l_stride * mine + 2u] >> 31;   // (bit 31: a school building of the persistent map)
            if (PM && all_ovf) { sch_l = bl[l_stride * mine + 3u]; if (sch_l >= d.n_sch) { sch_l = 0xFFFFFFFFu; RAISE(ctrl, ESIM_ERANGE, ERR_AT_BIG_LIST); } }
            if (!PM && all_ovf) {
                // a school building (k_chunk_marks: capacity word 0xFFFFFFFF, the school's number where the records would start)
                sch_l = base_l; base_l = 0u;
                if (sch_l >= d.n_sch || slot_l >= d.hcap) { sch_l = 0xFFFFFFFFu; slot_l = 0u; RAISE(ctrl, ESIM_ERANGE, ERR_AT_BIG_LIST); }
            } else
            if (slot_l < d.hcap && base_l <= d.ovf_n && cap_l <= d.ovf_n - base_l) {
                const uint32_t state = d.slot_state[slot_l];
                if (PM) { const uint32_t cnt = state & PSLOT_COUNT, inl = all_ovf ? 0u : ITEM_RECS; n_ov_l = min(cnt > inl ? cnt - inl : 0u, cap_l); }
                else n_ov_l = min((state > ITEM_RECS && state < SLOT_COUNTERS_ONLY) ? state - ITEM_RECS : 0u, cap_l);
            } else { slot_l = 0u; base_l = 0u; ctrl->error = (uint32_t)(-ESIM_ERANGE); }   // (no list entry k_chunk_marks wrote looks like this)
        }
        const uint32_t m = min(64u, (n_list - g + step - 1u) / step);
        // ... then slot by slot, lanes = records; the first 64 records of eight slots are fetched together
        for (uint32_t i8 = 0; i8 < m; i8 += 8u) {
        uint32_t ivs[8];
#pragma unroll
        for (uint32_t u = 0; u < 8u; ++u) ivs[u] = (i8 + u < m && lane < FX(n_ov_l, min(i8 + u, 63u))) ? d.ovf[FX(base_l, min(i8 + u, 63u)) + lane] : 0u;
#pragma unroll
        for (uint32_t u = 0; u < 8u; ++u) {
            const uint32_t i = i8 + u;
            if (i >= m) break;
            const uint32_t slot = FX(slot_l, i), base = FX(base_l, i), n_ov = FX(n_ov_l, i);
            uint32_t iv = ivs[u];
            uint32_t c0 = 0u, c1 = 0u;
            WORK_ADD(WK_FOLDED, lane == 0 ? n_ov : 0);
            if (PM && FX(sch_l, i) != 0xFFFFFFFFu) {
                // A school building: its members that are Infected in step j of the chunk are those with exposure steps te0 + j -
                // infected_time .. te0 + j -- a window sum over its histogram of exposure steps (prefix sums through LDS) --; they
                // stand there while those with a work place are at work, the riders among them not while riders are on a bus.
                const uint32_t sch = FX(sch_l, i), wv = threadIdx.x >> 6;
                const uint32_t *ring = d.sch_ring + (size_t)sch * 2u * SCH_RING;
                const int base_te = ct.te0 - ct.it;                               // exposure step of the window's low end in step 0
                const uint32_t len = (uint32_t)ct.it + n;                         // entries base_te .. base_te + len - 1 are needed
                uint32_t carry0 = 0u, carry1 = 0u;
                for (uint32_t r0 = 0; r0 < len; r0 += 64u) {
                    const int te = base_te + (int)(r0 + lane);
                    uint32_t x0 = (r0 + lane < len && te >= 0) ? ring[(uint32_t)te & (SCH_RING - 1u)] : 0u;
                    uint32_t x1 = (r0 + lane < len && te >= 0) ? ring[SCH_RING + ((uint32_t)te & (SCH_RING - 1u))] : 0u;
                    for (uint32_t o = 1; o < 64u; o <<= 1) { const uint32_t y0 = __shfl_up(x0, o, 64), y1 = __shfl_up(x1, o, 64); if (lane >= o) { x0 += y0; x1 += y1; } }
                    s_win[wv][0][r0 + lane + 1u] = x0 + carry0; s_win[wv][1][r0 + lane + 1u] = x1 + carry1;     // [k + 1] = sum of entries 0 .. k
                    carry0 += __shfl(x0, 63, 64); carry1 += __shfl(x1, 63, 64);
                }
                if (lane == 0) { s_win[wv][0][0] = 0u; s_win[wv][1][0] = 0u; }
                __builtin_amdgcn_wave_barrier();
                // step j: entries j .. j + infected_time of the window
                if (lane < n) {
                    const uint32_t all = s_win[wv][0][lane + (uint32_t)ct.it + 1u] - s_win[wv][0][lane], pt = s_win[wv][1][lane + (uint32_t)ct.it + 1u] - s_win[wv][1][lane];
                    c0 = ((AW.lo >> lane) & 1ull) ? all - (((BUS.lo >> lane) & 1ull) ? pt : 0u) : 0u;
                }
                if (64u + lane < n) {
                    const uint32_t j = 64u + lane;
                    const uint32_t all = s_win[wv][0][j + (uint32_t)ct.it + 1u] - s_win[wv][0][j], pt = s_win[wv][1][j + (uint32_t)ct.it + 1u] - s_win[wv][1][j];
                    c1 = ((AW.hi >> lane) & 1u) ? all - (((BUS.hi >> lane) & 1u) ? pt : 0u) : 0u;
                }
                __builtin_amdgcn_wave_barrier();
                // the ring's cells below the window are free again for exposure steps SCH_RING later
                for (uint32_t z = lane; z < 2u * FREE_MAX; z += 64u) {
                    const int te = base_te - 1 - (int)z;
                    if (te >= 0) { d.sch_ring[(size_t)sch * 2u * SCH_RING + ((uint32_t)te & (SCH_RING - 1u))] = 0u; d.sch_ring[((size_t)sch * 2u + 1u) * SCH_RING + ((uint32_t)te & (SCH_RING - 1u))] = 0u; }
                }
            }
            if (!PM && FX(sch_l, i) != 0xFFFFFFFFu) {
                // A school building of the per-chunk map: the prefix sums of its difference arrays are its Infected per step -- all of
                // them, and the riders among them --; they stand there while those with a work place are at work, the riders not
                // while riders are on a bus (what k_chunk_marks' `atw` says per citizen).  The arrays are zeroed for the next chunk.
                uint32_t *dd = d.sch_diff + (size_t)FX(sch_l, i) * SD_REPL * 2u * FREE_MAX;
                uint32_t a0 = 0u, a1 = 0u, p0 = 0u, p1 = 0u;
#pragma unroll
                for (uint32_t r = 0; r < SD_REPL; ++r) {
                    uint32_t *rr = dd + (size_t)r * 2u * FREE_MAX;
                    a0 += rr[lane]; p0 += rr[FREE_MAX + lane];
                    if (lane < FREE_MAX - 64u) { a1 += rr[64u + lane]; p1 += rr[FREE_MAX + 64u + lane]; }
                }
#pragma unroll
                for (uint32_t r = 0; r < SD_REPL; ++r) {
                    uint32_t *rr = dd + (size_t)r * 2u * FREE_MAX;
                    rr[lane] = 0u; rr[FREE_MAX + lane] = 0u;
                    if (lane < FREE_MAX - 64u) { rr[64u + lane] = 0u; rr[FREE_MAX + 64u + lane] = 0u; }
                }
                for (uint32_t o = 1; o < 64u; o <<= 1) {
                    const uint32_t ya = __shfl_up(a0, o, 64), yp = __shfl_up(p0, o, 64), yb = __shfl_up(a1, o, 64), yq = __shfl_up(p1, o, 64);
                    if (lane >= o) { a0 += ya; p0 += yp; a1 += yb; p1 += yq; }
                }
                a1 += __shfl(a0, 63, 64); p1 += __shfl(p0, 63, 64);
                if (lane < n) c0 = ((AW.lo >> lane) & 1ull) ? a0 - (((BUS.lo >> lane) & 1ull) ? p0 : 0u) : 0u;
                if (lane < FREE_MAX - 64u && 64u + lane < n) c1 = ((AW.hi >> lane) & 1u) ? a1 - (((BUS.hi >> lane) & 1u) ? p1 : 0u) : 0u;
            }
#ifdef ESIM_PROFILE_FOLD
            pq_rec += n_ov; ++pq_n;
#endif
            for (uint32_t b = 0; b < n_ov; b += 64u) {
                if (b) iv = b + lane < n_ov ? d.ovf[base + b + lane] : 0u;
                const uint32_t nb = min(64u, n_ov - b);
                if (nb <= 12u) {
                    // few records (a class room): one after the other, its set of steps in scalar registers, lanes = steps
                    for (uint32_t k = 0; k < nb; ++k) {
                        if (PM) { piv_count(FX(iv, k), lane, ct, AW, BUS, c0, c1); continue; }
                        const M96 at = iv_steps(FX(iv, k), AW, BUS);
                        c0 += (uint32_t)(at.lo >> lane) & 1u;
                        c1 += lane < 32u ? (at.hi >> lane) & 1u : 0u;
                    }
                    continue;
                }
                // many: lanes = records, one ballot per step (persistent map: cancellation records count negative)
                const M96 at = PM ? piv_steps(iv, ct, AW, BUS) : iv_steps(iv, AW, BUS);
                const bool negr = PM && (iv & PIV_NEG);
                const bool any_neg = PM && __any(negr);
                for (uint32_t j = 0; j < n && j < 64u; ++j) {
                    const bool here = (at.lo >> j) & 1ull;
                    uint32_t k = (uint32_t)__popcll(__ballot(here && !negr));
                    if (any_neg) k -= (uint32_t)__popcll(__ballot(here && negr));
                    if (lane == j) c0 += k;
                }
                for (uint32_t j = 64u; j < n; ++j) {
                    const bool here = (at.hi >> (j - 64u)) & 1u;
                    uint32_t k = (uint32_t)__popcll(__ballot(here && !negr));
                    if (any_neg) k -= (uint32_t)__popcll(__ballot(here && negr));
                    if (lane == j - 64u) c1 += k;
                }
            }
            uint32_t *v = d.vec + (size_t)slot * FREE_MAX;
            if (PM) {                                                         // (the map's counters belong to this kernel alone)
                if (lane < n) v[lane] = c0;
                if (64u + lane < n) v[64u + lane] = c1;
                continue;
            }
            // (added, not stored: commuters from other shards may have counted themselves there; atomics, because nobody has
            // to wait for them)
            if (lane < n && c0) atomicAdd(&v[lane], c0);
            if (64u + lane < n && c1) atomicAdd(&v[64u + lane], c1);
        }
        }
    }
    WORK_FLUSH(d);
#ifdef ESIM_PROFILE_FOLD
    PROF_PUT(d, 11, pq_n); PROF_PUT(d, 12, PROF_NOW() - pq0); PROF_PUT(d, 13, pq_rec);
#endif
}

// A successful draw of citizen m in step s (bus: on public transport).
__device__ __forceinline__ void expose_min(const Dev &d, Ctrl *ctrl, uint32_t m, uint32_t w, uint32_t s, uint32_t bus)
{
    const uint32_t cand = CW_MAKE(s + TE_BIAS, bus | (w & CW_KEEP));
    const uint32_t prev = atomicMin(&d.cit[m], cand);
    // exposed on a bus although the chunk's plan vaccinates it later: it leaves the eligible set with this exposure (simulator.rs:447-449),
    // so the plan of the steps from here on is off by this citizen -- noted for the repair (k_chunk_vax<true>)
    if (bus && cand < prev && CW_VAX_REL(w) != CW_VAX_NONE) {
        const uint32_t at = atomicAdd(&d.hot[HOT_LOST * HOT_STRIDE], 1u);
        if (at < LOST_CAP) d.lost_list[at] = m;
    }
    if (cand < prev && CW_TE(prev) == TE_SUSCEPTIBLE) {                       // first exposure in this chunk
        const uint32_t r = m & (SUBQ - 1u);
        d.newexp[(size_t)r * d.newexp_cap + atomicAdd(&d.hot[(HOT_NEWEXP + r) * HOT_STRIDE], 1u)] = m;
    }
}

struct ChunkShared {
    Decision dec[FREE_MAX];
    uint64_t thr[512];
};
struct RouteShared {
    uint32_t s_key[CHUNK_ROUTE_MAX];
    uint16_t s_bus[CHUNK_ROUTE_MAX];
    uint8_t s_inf[CHUNK_ROUTE_MAX];
    uint32_t s_cnt[CHUNK_ROUTE_MAX + 1];
};
// Per wavefront: the item's / the school's Infected per step; 64 staged members; the item's slots of four time steps
// (item_steps_regs).  A slot's descriptor is eight words: [0] first step of the slot + 3 (bits 0-6; a slot may begin up to three
// steps before the chunk) | its steps that are marked (8-11) | in which of them those with a work place are at work (12-15) | in
// which masks are worn everywhere (16-19); [1] the item's Infected & 255 in the four steps, a byte each (the threshold index,
// `as u8`); [2] the same for the school of a room; [3], [4] the item's Infected in steps 0-1 / 2-3, 16 bits each.
#define SLOT_STEPS 4u
struct WaveScratch { uint32_t rounds; uint32_t cnt[FREE_MAX]; uint32_t sch[FREE_MAX]; uint32_t mem_id[64]; uint32_t mem_w[64]; uint4 desc[2u * (FREE_MAX / SLOT_STEPS + 1u)]; };

// One member list of one item over the marked steps of the chunk.  The time steps 4k .. 4k+3 share one Philox block (RNG
// contract: step t takes word t & 3), so the unit of work is a (member, slot of four steps) pair: the pairs [p_lo, p_hi) are
// spread densely over the 64 lanes (the draws are Philox-bound -- 20 quarter-rate multiplies a block -- so idle lanes and
// blocks used for one draw only are what costs).  ws.desc: the item's slots with a marked step, in order, S of them.
// kind 0 residents, 1 workers, 2 room participants.
// pre_m / pre_w: members lo + pre_base + lane of the list and their words when the caller has already fetched them (have_pre).
__device__ __forceinline__ void member_pairs(const Dev &d, Ctrl *ctrl, const ChunkShared &sm, WaveScratch &ws, const uint32_t *idx,
                                             uint32_t lo, uint32_t p_lo, uint32_t p_hi, uint32_t lane, uint32_t kind, uint32_t S, uint32_t t0 WORK_ARG,
                                             bool have_pre = false, uint32_t pre_m = 0u, uint32_t pre_w = 0u, uint32_t pre_base = 0u)
{
    const uint64_t seed = ((uint64_t)d.seed_hi << 32) | d.seed_lo;
    // members touched by the slots [p_lo, p_hi): staged in LDS 64 at a time -- every member recurs once per slot
    const uint32_t m_first = p_lo / S, m_last = (p_hi - 1u) / S;
    for (uint32_t mb = m_first; mb <= m_last; mb += 64u) {
        __builtin_amdgcn_wave_barrier();
        if (mb + lane <= m_last) {
            WORK_ADD(WK_MEMBERS, 1); WORK_ADD(WK_MEMBERS_IDX, idx ? 1 : 0);
            if (have_pre && mb == pre_base) { ws.mem_id[lane] = pre_m; ws.mem_w[lane] = pre_w; }
            else {
                const uint32_t m = idx ? idx[lo + mb + lane] : lo + mb + lane;
                ws.mem_id[lane] = m;
                ws.mem_w[lane] = d.cit[m];
            }
        }
        __builtin_amdgcn_wave_barrier();
        const uint32_t q_lo = max(p_lo, mb * S), q_hi = min(p_hi, (mb + 64u) * S);
#ifdef ESIM_WAVE_PROFILE
        if (lane == 0) ws.rounds += (q_hi - q_lo + 63u) / 64u;
#endif
        for (uint32_t p = q_lo + lane; p < q_hi; p += 64u) {
            const uint32_t um = p / S, si = p - um * S;
            const uint32_t m = ws.mem_id[um - mb], w = ws.mem_w[um - mb];
            const uint32_t te = CW_TE(w);
            WORK_ADD(WK_PAIRS, 1);
            if (te >= TE_RECOVERED && te != TE_SUSCEPTIBLE) continue;
            const uint4 dsc = ws.desc[2u * si];
            const uint32_t cnt23 = ws.desc[2u * si + 1u].x;
            const int jb = (int)(dsc.x & 127u) - 3;                            // first step of the slot (may lie before the chunk)
            const uint32_t mk = (dsc.x >> 8) & 15u, atw = (dsc.x >> 12) & 15u, everywhere = (dsc.x >> 16) & 15u;
            // The steps of the slot in which this member takes a draw, all four at once: marked; Susceptible when this list is
            // walked in that step -- w > (step << 19 | the bits an exposure keeps), i.e. never exposed, or so far only by
            // something that comes later (a later step, or a bus of this step: that exposure may be undercut) --; not Vaccinated
            // by then (k_chunk_vax); standing in the building's area (simulator.rs:324)
            const uint32_t vrel = CW_VAX_REL(w);
            const int js = (int)te + (((w & ~CW_KEEP) & ((1u << CW_TE_SHIFT) - 1u)) ? 1 : 0) - (int)(t0 + TE_BIAS);   // Susceptible in steps j < js
            const int lim = min(js, vrel == CW_VAX_NONE ? (int)FREE_MAX : (int)vrel + 1) - jb;                  // ... of the slot: h < lim
            const uint32_t early = lim <= 0 ? 0u : lim >= (int)SLOT_STEPS ? 15u : (1u << lim) - 1u;
            const bool same = w & FL_SAME_AREA;
            const uint32_t here = kind == 0u ? (((w & FL_HAS_WORK) && !same) ? ~atw : 15u) : (same ? 15u : atw);
            const uint32_t act = mk & early & here;
            if (!act) continue;
            WORK_ADD(WK_PAIRS_ACTIVE, 1);
            const uint32_t nn = kind == 2u ? dsc.z : dsc.y;                    // exposure_count & 255 per step: infected in the building
            const uint32_t row = (w & FL_MASK_COMPLIANT) ? 0u : everywhere;   // steps in which this member's chance is the masked one
            uint64_t thr[SLOT_STEPS];
#pragma unroll
            for (uint32_t h = 0; h < SLOT_STEPS; ++h) thr[h] = sm.thr[(((row >> h) & 1u) << 8) + ((nn >> (8u * h)) & 255u)];
            const uint32_t gid = d.id_base + m;
            const uint32_t s_blk = (uint32_t)((int)t0 + jb) + (uint32_t)__builtin_ctz(act);   // a time step of the slot: names its block
            uint32_t hit = 0u;                                                // bit h: a draw of step h succeeded
            if (kind == 2u) {
                // School::find_exposures: one draw per Infected in the room (building.rs:494-522); the earliest step decides
                const uint32_t cnt[SLOT_STEPS] = { dsc.w & 0xFFFFu, dsc.w >> 16, cnt23 & 0xFFFFu, cnt23 >> 16 };
                uint32_t kmax = 0u;
#pragma unroll
                for (uint32_t h = 0; h < SLOT_STEPS; ++h) if ((act >> h) & 1u) kmax = max(kmax, cnt[h]);
                const uint32_t first_act = act & (0u - act);
#ifdef ESIM_COUNT_WORK
                for (uint32_t h = 0; h < SLOT_STEPS; ++h) if ((act >> h) & 1u) WORK_ADD(WK_DRAWS, cnt[h]);
#endif
                for (uint32_t k = 0; k < kmax && !(hit & first_act); ++k) {
                    WORK_ADD(WK_BLOCKS, 1);
                    const philox_out o = esim_draw_block(seed, gid, s_blk, ESIM_SLOT_ROOM0 + k);
                    const uint32_t wd[SLOT_STEPS] = { o.w0, o.w1, o.w2, o.w3 };
#pragma unroll
                    for (uint32_t h = 0; h < SLOT_STEPS; ++h) if (((act >> h) & 1u) && k < cnt[h] && (uint64_t)wd[h] < thr[h]) hit |= 1u << h;
                }
            } else {
                WORK_ADD(WK_BLOCKS, 1); WORK_ADD(WK_DRAWS, __popc(act));
                const philox_out o = esim_draw_block(seed, gid, s_blk, kind == 0u ? ESIM_SLOT_HOME : ESIM_SLOT_WORK);
                const uint32_t wd[SLOT_STEPS] = { o.w0, o.w1, o.w2, o.w3 };
#pragma unroll
                for (uint32_t h = 0; h < SLOT_STEPS; ++h) if ((uint64_t)wd[h] < thr[h]) hit |= 1u << h;
                hit &= act;
            }
            if (hit) { WORK_ADD(WK_HITS, 1); expose_min(d, ctrl, m, w, (uint32_t)((int)t0 + jb) + (uint32_t)__builtin_ctz(hit), 0u); }   // (the earliest wins anyway)
        }
    }
}

// The marked steps of item v as slots of four time steps (4k .. 4k+3), in order, with what member_pairs needs of each step,
// into this wavefront's scratch (ws.sch holds the school's counts when the item is a room).  Returns S, the number of slots
// with a marked step.
// (four bits of a set of steps from step p on; p may be up to three steps before the chunk)
__device__ __forceinline__ uint32_t m96_nibble(const M96 &m, int p)
{
    if (p < 0) return (uint32_t)(m.lo << (-p)) & 15u;
    if (p >= 64) return p >= 96 ? 0u : (m.hi >> (p - 64)) & 15u;
    return (uint32_t)((m.lo >> p) | (p > 60 ? (unsigned long long)m.hi << (64 - p) : 0ull)) & 15u;
}

// AW / EV: the steps of the chunk in which those with a work place are at work / masks are worn everywhere.
__device__ __forceinline__ uint32_t item_steps_regs(uint32_t c0, uint32_t c1, uint32_t lane, WaveScratch &ws, uint32_t t0, const M96 &AW, const M96 &EV)
{
    ws.cnt[lane] = c0;
    if (lane < FREE_MAX - 64u) ws.cnt[64u + lane] = c1;
    const M96 MK = { __ballot(c0 != 0u), (uint32_t)__ballot(lane < FREE_MAX - 64u && c1 != 0u) };
    __builtin_amdgcn_wave_barrier();
    // lane L looks at the slot whose first time step is step j0 = 4L - (t0 & 3) of the chunk (negative: before the chunk)
    const int j0 = (int)(SLOT_STEPS * lane) - (int)(t0 & (SLOT_STEPS - 1u));
    const uint32_t mk = lane <= FREE_MAX / SLOT_STEPS ? m96_nibble(MK, j0) : 0u;
    const unsigned long long present = __ballot(mk != 0u);
    if (mk) {
        // the counts of the slot's four steps (the item's, the school's), all eight reads in flight together
        uint32_t c[SLOT_STEPS], sc[SLOT_STEPS];
#pragma unroll
        for (uint32_t h = 0; h < SLOT_STEPS; ++h) {
            const int j = j0 + (int)h;
            const uint32_t jc = (uint32_t)(j < 0 ? 0 : j >= (int)FREE_MAX ? (int)FREE_MAX - 1 : j);
            c[h] = ws.cnt[jc]; sc[h] = ws.sch[jc];
        }
        uint32_t nn = 0u, ns = 0u;
#pragma unroll
        for (uint32_t h = 0; h < SLOT_STEPS; ++h) {
            if (!((mk >> h) & 1u)) c[h] = 0u;
            nn |= (c[h] & 255u) << (8u * h);
            ns |= (sc[h] & 255u) << (8u * h);
        }
        const uint32_t i = (uint32_t)__popcll(present & ((1ull << lane) - 1ull));
        ws.desc[2u * i] = make_uint4((uint32_t)(j0 + 3) | (mk << 8) | (m96_nibble(AW, j0) << 12) | (m96_nibble(EV, j0) << 16), nn, ns,
                                     min(c[0], 0xFFFFu) | (min(c[1], 0xFFFFu) << 16));
        ws.desc[2u * i + 1u] = make_uint4(min(c[2], 0xFFFFu) | (min(c[3], 0xFFFFu) << 16), 0u, 0u, 0u);
    }
    return (uint32_t)__popcll(present);
}

__device__ __forceinline__ uint32_t fetch_slot(const Dev &d, uint32_t slot, uint32_t lane);
__device__ __forceinline__ void item_counts(const Dev &d, uint32_t x, uint32_t slot, uint32_t lane, uint32_t n, const M96 &AW,
                                            const M96 &BUS, uint32_t &c0, uint32_t &c1);
__device__ __forceinline__ void pitem_counts(const Dev &d, uint32_t x, uint32_t slot, uint32_t lane, const ChunkT &ct, const M96 &AW,
                                             const M96 &BUS, bool vec_only, uint32_t &c0, uint32_t &c1);
__device__ __forceinline__ void school_counts(const Dev &d, uint32_t s_sch, uint32_t lane, uint32_t n, const Decision &q0, const Decision &q1, WaveScratch &ws)
{
    // s_sch: the hash slot of the room's school (k_chunk_marks left it in the room's record): infected in the whole school, per
    // step.  Everybody Infected in a school building counts itself in the slot's per-step counters (k_chunk_marks).
    uint32_t c0 = 0u, c1 = 0u;
    if (s_sch != 0xFFFFFFFFu) {
        if (lane < n) c0 = d.vec[(size_t)s_sch * FREE_MAX + lane];
        if (64u + lane < n) c1 = d.vec[(size_t)s_sch * FREE_MAX + 64u + lane];
    }
    ws.sch[lane] = c0;
    if (lane < FREE_MAX - 64u) ws.sch[64u + lane] = c1;
}

// Lists with more pairs than this are cut into units that any wavefront can take (k_chunk_units), so that one
// 200-member workplace does not keep a single wavefront busy while the chip idles.
// A deferred unit carries everything its consumer needs, so that it is three dependent loads away from drawing: the
// item's hash slot and its claimer's stretch (the Infected per step), the school's slot for a room, the member list, and
// where in it the unit's first pair falls.
struct UnitSrc { uint32_t slot, link, own; };
__device__ __forceinline__ void list_or_units(const Dev &d, Ctrl *ctrl, const ChunkShared &sm, WaveScratch &ws, const uint32_t *idx,
                                              uint32_t lo, uint32_t hi, const UnitSrc &src, uint32_t lane, uint32_t kind, uint32_t S, uint32_t t0 WORK_ARG,
                                              bool have_pre = false, uint32_t pre_m = 0u, uint32_t pre_w = 0u)
{
    const uint32_t pairs = (hi - lo) * S;
    if (pairs == 0) return;
    if (pairs <= UNIT_INLINE) { member_pairs(d, ctrl, sm, ws, idx, lo, 0u, pairs, lane, kind, S, t0 WORK_PASS, have_pre, pre_m, pre_w); return; }
    const uint32_t n_units = (pairs + UNIT_PAIRS - 1u) / UNIT_PAIRS;
    const uint32_t r = ((blockIdx.x * TPB + threadIdx.x) >> 6) & (SUBQ - 1u);  // this wavefront's queue
    uint32_t start = 0;
    if (lane == 0) start = atomicAdd(&d.hot[(HOT_UNITS + r) * HOT_STRIDE], n_units);
    start = __shfl(start, 0, 64);
    UnitRec *q = d.units + (size_t)r * d.unit_qcap;
    if (start + n_units > d.unit_qcap) {
        // queue full: what was reserved of it becomes no-ops and the list is drawn here
        for (uint32_t i = lane; i < n_units && start + i < d.unit_qcap; i += 64u) q[start + i].code = UNIT_NOOP;
        member_pairs(d, ctrl, sm, ws, idx, lo, 0u, pairs, lane, kind, S, t0 WORK_PASS, have_pre, pre_m, pre_w);
        return;
    }
    for (uint32_t i = lane; i < n_units; i += 64u) {
        const uint32_t p_lo = i * UNIT_PAIRS;
        q[start + i] = UnitRec{ src.slot, kind == 2u ? src.link : 0xFFFFFFFFu, lo, hi - lo, (kind << 30) | p_lo, src.own, p_lo / S, 0u };
    }
}

// What a wavefront needs of item v before it can start on it; depends on v alone, so the fetch of the next item is
// issued before the work on the current one (the pass is bound by chains of dependent loads, not by bandwidth).
// The fetch of an item is one register in two hops: lanes 0..7 its record and lane 17 its hash slot (ITEM_UNUSED: id not
// handed out) by item id; then lanes 8..14 the slot's interval records and lane 16 their number by slot.
struct ItemFetch { uint32_t slot, id, a_lo, a_hi, b_lo, b_hi, aux, link, c0, c1; };
__device__ __forceinline__ uint32_t fetch_item(const Dev &d, uint32_t v, uint32_t lane)
{
    uint32_t x = 0u;
    if (lane < 8u) x = reinterpret_cast<const uint32_t *>(d.item_rec)[(size_t)v * 8u + lane];
    else if (lane == LANE_HSLOT) x = d.hitems[v];
    return x;
}
__device__ __forceinline__ uint32_t fetch_slot(const Dev &d, uint32_t slot, uint32_t lane)
{
    uint32_t x = 0u;
    if (slot < d.hcap) {                                                       // (ITEM_UNUSED, or anything else that is no slot: nothing fetched)
        if (lane >= 8u && lane < 8u + ITEM_RECS) x = d.slot_iv[(size_t)slot * SLOT_IV_STRIDE + (lane - 8u)];
        else if (lane == LANE_STATE) x = d.slot_state[slot];
    }
    return x;
}
__device__ __forceinline__ uint32_t merge_fetch(uint32_t by_id, uint32_t by_slot, uint32_t lane)
{
    return (lane < 8u || lane == LANE_HSLOT) ? by_id : by_slot;
}

// Infected standing in the item in step `lane` (c0) and `64 + lane` (c1) of the chunk: the records of its slot, plus the
// per-step counters of those that found no record free.  (The claimer's own stretch is added by decode_item.)
__device__ __forceinline__ void item_counts(const Dev &d, uint32_t x, uint32_t slot, uint32_t lane, uint32_t n, const M96 &AW,
                                            const M96 &BUS, uint32_t &c0, uint32_t &c1)
{
    const uint32_t state = FX(x, LANE_STATE);
    c0 = 0u; c1 = 0u;
    if (state > ITEM_RECS) {
        // (summed up by k_chunk_fold from the records beyond ITEM_RECS)
        if (lane < n) c0 = d.vec[(size_t)slot * FREE_MAX + lane];
        if (64u + lane < n) c1 = d.vec[(size_t)slot * FREE_MAX + 64u + lane];
    }
    const uint32_t n_rec = state >= SLOT_COUNTERS_ONLY ? 0u : state < ITEM_RECS ? state : ITEM_RECS;
    for (uint32_t k = 0; k < n_rec; ++k) {
        const uint32_t iv = (uint32_t)__builtin_amdgcn_readlane((int)x, (int)(8u + k));
        iv_count(iv, lane, AW, BUS, c0, c1);
    }
}

// The same for a slot of the persistent map: its records are absolute (piv_count); beyond ITEM_RECS of them (or all of them:
// vec_only, a school building) k_map_fold has left the sums in `vec`.
__device__ __forceinline__ void pitem_counts(const Dev &d, uint32_t x, uint32_t slot, uint32_t lane, const ChunkT &ct, const M96 &AW,
                                             const M96 &BUS, bool vec_only, uint32_t &c0, uint32_t &c1)
{
    const uint32_t cnt = FX(x, LANE_STATE) & PSLOT_COUNT;
    c0 = 0u; c1 = 0u;
    if (vec_only || cnt > ITEM_RECS) {                                     // (a school: k_map_fold stores its counters in every chunk it is listed for)
        if (lane < ct.n) c0 = d.vec[(size_t)slot * FREE_MAX + lane];
        if (64u + lane < ct.n) c1 = d.vec[(size_t)slot * FREE_MAX + 64u + lane];
    }
    if (vec_only) return;
    const uint32_t n_rec = cnt < ITEM_RECS ? cnt : ITEM_RECS;
    for (uint32_t k = 0; k < n_rec; ++k) piv_count(FX(x, 8u + k), lane, ct, AW, BUS, c0, c1);
}

template <bool PM>
__device__ __forceinline__ ItemFetch decode_item(const Dev &d, uint32_t x, uint32_t lane, uint32_t n, const ChunkT &ct, const M96 &AW, const M96 &BUS)
{
    ItemFetch f;
    f.slot = FX(x, LANE_HSLOT); f.id = FX(x, 0); f.a_lo = FX(x, 1); f.a_hi = FX(x, 2); f.b_lo = FX(x, 3); f.b_hi = FX(x, 4); f.aux = FX(x, 5); f.link = FX(x, 6);
    f.c0 = 0u; f.c1 = 0u;
    if (f.slot < d.hcap) {
        if (PM) pitem_counts(d, x, f.slot, lane, ct, AW, BUS, false, f.c0, f.c1);
        else {
            item_counts(d, x, f.slot, lane, n, AW, BUS, f.c0, f.c1);
            iv_count(FX(x, 7), lane, AW, BUS, f.c0, f.c1);
        }
    }
    return f;
}

// What a chunk table says is checked against the capacities before it is used as an index: a table that does not hold what
// k_chunk_marks writes (a diagnostics build that leaves a write out, a defect) ends in ESIM_ERANGE, not in a memory fault.
__device__ __forceinline__ bool item_ok(const Dev &d, const ItemFetch &it)
{
    if (it.slot >= d.hcap) return false;
    if (it.id < d.n_bld) return it.a_lo <= it.a_hi && it.a_hi <= d.n && it.b_lo <= it.b_hi && it.b_hi <= d.n_wrk_idx;
    if (it.id < d.n_bld + d.n_room) return it.a_lo <= it.a_hi && it.a_hi <= d.n_room_idx && (it.link == 0xFFFFFFFFu || it.link < d.hcap);
    return true;
}
#define PAIR_SPREAD 1237u
// apply_exposures (simulator.rs:262-405) for every item and every step of the chunk.
// One (route of <= 64 riders, bus step j) pair with an Infected rider: rank the riders by (Philox key, id) with shuffles, buses
// are runs of bus_capacity ranks, every bus with an Infected rider draws for its Susceptible riders (simulator.rs:362-401).
__device__ __forceinline__ void route_pair_small(const Dev &d, Ctrl *ctrl, const ChunkShared &sm, uint32_t off, uint32_t sz, uint32_t j, uint32_t t0, uint32_t lane WORK_ARG)
{
    const uint64_t seed = ((uint64_t)d.seed_hi << 32) | d.seed_lo;
    const uint32_t s = t0 + j, mask = sm.dec[j].mask;
    uint32_t c = 0, w = 0, key = 0;
    bool inf = false;
    WORK_ADD(WK_ROUTE_PAIRS, lane == 0 ? 1 : 0); WORK_ADD(WK_RIDERS, lane < sz ? 1 : 0);
    bool can = false;                                                          // could take a draw on this bus step at all
    if (lane < sz) {
        c = d.route_riders[off + lane];
        w = d.cit[c];
        inf = status_in_chunk(d, w, t0, j) == ESIM_INFECTED;
        const uint32_t te = CW_TE(w);
        can = !(w <= CW_MAKE(s + TE_BIAS, CW_BUS_EXPOSED | (w & CW_KEEP)) || (te >= TE_RECOVERED && te != TE_SUSCEPTIBLE) || j > CW_VAX_REL(w));
    }
    // Nobody Infected aboard, or nobody who could still be exposed: no draw is made, whatever the buses (the order of the riders
    // is only needed to tell who shares a bus with whom).  A route that fills one bus at most needs no order either.
    if (!__any(inf) || !__any(can)) return;
    uint32_t rank = 0;
    if (sz > d.bus_capacity) {
        if (lane < sz) key = philox4x32_10(d.id_base + c, s, ESIM_SLOT_BUS_ORDER, 0u, d.seed_lo, d.seed_hi).w0;
        for (uint32_t i = 0; i < sz; ++i) {
            const uint32_t ki = (uint32_t)__builtin_amdgcn_readlane((int)key, (int)i);   // i is uniform: a scalar broadcast
            rank += ki < key || (ki == key && i < lane);                     // ids ascend with the lane
        }
    }
    const uint32_t bus = rank / d.bus_capacity;
    // Infected riders on my bus: one ballot per bus of the route
    const unsigned long long inf_m = __ballot(inf);
    uint32_t k = 0;
    const uint32_t n_bus = (sz + d.bus_capacity - 1u) / d.bus_capacity;
    for (uint32_t b = 0; b < n_bus; ++b) {
        const unsigned long long on_b = __ballot(lane < sz && bus == b);
        if (bus == b) k = (uint32_t)__popcll(on_b & inf_m);
    }
    if (lane < sz && k && can) {                                               // not exposed before this bus, not Vaccinated by then
        const uint32_t row = (!(w & FL_MASK_COMPLIANT) && mask == ESIM_MASK_EVERYWHERE) ? 1u : 0u;
        WORK_ADD(WK_BUS_DRAWS, 1);
        if (esim_u32(seed, d.id_base + c, s, ESIM_SLOT_BUS) < sm.thr[row * 256u + (k & 255u)]) { WORK_ADD(WK_HITS, 1); expose_min(d, ctrl, c, w, s, CW_BUS_EXPOSED); }
    }
}

// One (route of more than 64 riders, bus step) pair, by a whole workgroup of NT threads: ranks through LDS (simulator.rs:362-401).
template <uint32_t NT>
__device__ __forceinline__ void route_pair_big(const Dev &d, Ctrl *ctrl, const ChunkShared &sm, RouteShared &rs, uint32_t code, uint32_t t0, uint32_t n WORK_ARG)
{
    const uint64_t seed = ((uint64_t)d.seed_hi << 32) | d.seed_lo;
    const uint32_t r = code >> 7, j = code & 127u;
    if (r >= d.n_routes || j >= n) { if (threadIdx.x == 0) ctrl->error = (uint32_t)(-ESIM_ERANGE); return; }   // (block-uniform)
    const uint32_t off = d.route_off[r], sz = d.route_off[r + 1] - off;
    if (sz > CHUNK_ROUTE_MAX) { if (threadIdx.x == 0) ctrl->error = (uint32_t)(-ESIM_ERANGE); return; }
    const uint32_t s = t0 + j, mask = sm.dec[j].mask;
    WORK_ADD(WK_ROUTE_PAIRS, threadIdx.x == 0 ? 1 : 0);
    int loc_inf = 0, loc_can = 0;
    for (uint32_t i = threadIdx.x; i < sz; i += NT) {
        const uint32_t c = d.route_riders[off + i];
        WORK_ADD(WK_RIDERS, 1);
        const uint32_t w = d.cit[c], te = CW_TE(w);
        rs.s_inf[i] = status_in_chunk(d, w, t0, j) == ESIM_INFECTED ? 1 : 0;
        loc_inf |= rs.s_inf[i];
        loc_can |= !(w <= CW_MAKE(s + TE_BIAS, CW_BUS_EXPOSED | (w & CW_KEEP)) || (te >= TE_RECOVERED && te != TE_SUSCEPTIBLE) || j > CW_VAX_REL(w)) ? 1 : 0;
    }
    // (nobody Infected aboard, or nobody who could still be exposed: no draw is made, whatever the order of the riders)
    const int any_inf = __syncthreads_or(loc_inf), any_can = __syncthreads_or(loc_can);
    if (!any_inf || !any_can) return;
    for (uint32_t i = threadIdx.x; i < sz; i += NT)
        rs.s_key[i] = philox4x32_10(d.id_base + d.route_riders[off + i], s, ESIM_SLOT_BUS_ORDER, 0u, d.seed_lo, d.seed_hi).w0;
    for (uint32_t i = threadIdx.x; i < sz / d.bus_capacity + 1u; i += NT) rs.s_cnt[i] = 0u;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < sz; i += NT) {
        const uint32_t ki = rs.s_key[i];
        uint32_t rank = 0;
        for (uint32_t qq = 0; qq < sz; ++qq) { const uint32_t kq = rs.s_key[qq]; rank += kq < ki || (kq == ki && qq < i); }
        const uint32_t bus = rank / d.bus_capacity;
        rs.s_bus[i] = (uint16_t)bus;
        if (rs.s_inf[i]) atomicAdd(&rs.s_cnt[bus], 1u);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < sz; i += NT) {
        const uint32_t k = rs.s_cnt[rs.s_bus[i]];
        if (!k) continue;
        const uint32_t c = d.route_riders[off + i];
        const uint32_t w = d.cit[c], te = CW_TE(w);
        if (w <= CW_MAKE(s + TE_BIAS, CW_BUS_EXPOSED | (w & CW_KEEP)) || (te >= TE_RECOVERED && te != TE_SUSCEPTIBLE) || j > CW_VAX_REL(w)) continue;   // exposed before this bus, or Vaccinated by then
        const uint32_t row = (!(w & FL_MASK_COMPLIANT) && mask == ESIM_MASK_EVERYWHERE) ? 1u : 0u;
        WORK_ADD(WK_BUS_DRAWS, 1);
        if (esim_u32(seed, d.id_base + c, s, ESIM_SLOT_BUS) < sm.thr[row * 256u + (k & 255u)]) { WORK_ADD(WK_HITS, 1); expose_min(d, ctrl, c, w, s, CW_BUS_EXPOSED); }
    }
    __syncthreads();
}

// PM: the items are those of the persistent map -- ids handed out from SUBQ sub-lists (n_mw = SUBQ), counts from absolute records,
// routes are items among the others (their records are their Infected riders: the bus steps with any are ranked right here, longer
// routes go to k_chunk_units' list), and there is no list of (route, step) pairs.
template <bool PM>
__global__ __launch_bounds__(TPB) void k_chunk_draw(Dev d, uint32_t n_mw)
{
    __shared__ ChunkShared sm;
    __shared__ WaveScratch wsc[TPB / 64];
    Ctrl *ctrl = d.ctrl;
    const uint32_t t0 = ctrl->chunk_t0, n = ctrl->chunk_ok;
    if (!ctrl->chunk_parallel || n == 0u) return;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (blockIdx.x * TPB + threadIdx.x) >> 6, n_waves = (gridDim.x * TPB) >> 6;   // <= CHUNK_WAVES_MAX (esim_create)
    const uint32_t pt0 = PROF_NOW();
    WORK_TALLY;
    uint32_t p_items = 0u, p_item_max = 0u;
    // The wavefronts of k_chunk_marks (same grid) each handed out the first used_cnt[w] ids of [w * per_wave, ...): whoever
    // reaches a key first claims its item, so the early wavefronts hold far more items than the late ones.  The pass
    // therefore takes the items in id order as ONE dense sequence and every wavefront draws an equal stretch of it
    // (k_chunk_fold left the prefix sums of used_cnt in used_pref).
    // n_mw = wavefronts of k_chunk_marks (whose id ranges the items sit in); this kernel's own grid is a multiple of that:
    // equal stretches are equal in ITEMS, not in work -- 20 to 40 items per wavefront with member lists of 2 to 200, then the
    // wavefront's share of the routes: in round 2 the slowest of 4096 wavefronts ran 1.4x (items) and 2x (routes) the median,
    // and 2.8 of the 4 wavefronts a SIMD had been given were resident on average.  Taking blocks from shared counters does
    // not help at 20 items per wavefront (measured: +-0); what does is MORE, SHORTER wavefronts than the chip holds at once
    // (ESIM_DRAW_MULT x the marks grid): the dispatcher starts the next workgroup where one has finished.
    const uint32_t per_wave = ld(&ctrl->items_per_wave);
    if ((unsigned long long)per_wave * n_mw > d.items_cap || n_waves % n_mw != 0u) return;   // (k_chunk_marks raised ESIM_ERANGE and left no items)
    const uint32_t G = n_waves / n_mw;
    const uint32_t T = d.used_pref[n_mw];
    const uint32_t coarse = d.used_pref[min(64u * lane, n_mw)];
    const uint32_t Tq = T / n_waves, Tr = T % n_waves;                         // (T * wave / n_waves without 64-bit division)
    const uint32_t d_lo = Tq * wave + (uint32_t)(((unsigned long long)Tr * wave) / n_waves), d_hi = Tq * (wave + 1u) + (uint32_t)(((unsigned long long)Tr * (wave + 1u)) / n_waves);
    // the wavefront of k_chunk_marks that owns dense index i: the last one whose ids start at or before it -- first among
    // every 64th (`coarse`), then among the 64 from there on (`win`: lane l holds where the ids of owner ow_base + l start; a
    // window of 64 owners, moved on when used up)
    uint32_t ow_base = 0u, win = 0u, ow = 0u;
    auto seek = [&](uint32_t i) {
        ow_base = 64u * ((uint32_t)__popcll(__ballot(64u * lane < n_mw && coarse <= i)) - 1u);
        win = d.used_pref[min(ow_base + lane, n_mw)];
        ow = ow_base + (uint32_t)__popcll(__ballot(ow_base + lane < n_mw && win <= i)) - 1u;
    };
    seek(d_lo);
    // item id of dense index i; called with ascending i
    auto id_of = [&](uint32_t i) -> uint32_t {
        for (;;) {
            const uint32_t rel = ow - ow_base;
            if (rel == 63u) { ow_base = ow; win = d.used_pref[min(ow_base + lane, n_mw)]; continue; }
            if (ow + 1u < n_mw && (uint32_t)__builtin_amdgcn_readlane((int)win, (int)(rel + 1u)) <= i) { ++ow; continue; }
            return ow * per_wave + (i - (uint32_t)__builtin_amdgcn_readlane((int)win, (int)rel));
        }
    };
    // three items in flight: the record of the one after next (by id), the slot records of the next (by its slot), this one
    uint32_t id_cur = 0u, id_nxt = 0u, sl_cur = 0u;
    if (d_lo < d_hi) id_cur = fetch_item(d, id_of(d_lo), lane);
    if (d_lo + 1u < d_hi) id_nxt = fetch_item(d, id_of(d_lo + 1u), lane);
    // ... and the first look at the (route, bus step) pairs dealt to this wavefront (phase 2 below), so that they are here
    // when the items are done
    // Wavefront w of k_chunk_marks left pair_cnt[w] pairs in its own stretch of K places.  Its k-th pair goes to the wavefront
    // of this kernel with number ((w + k * PAIR_SPREAD) mod n_mw) + n_mw * (k mod G): lane l of wavefront (base, r) looks at
    // k = l * G + r of the stretch it may have been dealt from.
    const uint32_t K = PAIR_K(per_wave, ld(&ctrl->chunk_bus));                 // pairs a stretch of the list can hold
    const uint32_t w_base = wave % n_mw, w_rep = wave / n_mw;
    uint32_t code_l = 0u, off_l = 0u, sz_l = 0u;
    bool have = false;
    if (!PM) {
        const uint32_t k = lane * G + w_rep;
        if (k < K) {
            const uint32_t src = (w_base + n_mw - (uint32_t)(((unsigned long long)k * PAIR_SPREAD) % n_mw)) % n_mw;
            code_l = d.route_pairs[(size_t)src * K + k];                      // in bounds whether or not the pair exists
            have = k < d.pair_cnt[src] && (code_l >> 7) < d.n_routes && (code_l & 127u) < n;
        }
    }
    for (uint32_t i = threadIdx.x; i < n; i += TPB) sm.dec[i] = d.dec[i];
    for (uint32_t i = threadIdx.x; i < 512u; i += TPB) sm.thr[i] = d.thr[i];
    __syncthreads();
    const uint32_t route_base = d.n_bld + d.n_room;
    WaveScratch &ws = wsc[threadIdx.x >> 6];
    const Decision q0 = lane < n ? sm.dec[lane] : Decision{ 0u, 0u, 0u, 0u };
    const Decision q1 = 64u + lane < n ? sm.dec[64u + lane] : Decision{ 0u, 0u, 0u, 0u };
    M96 AW, BUS;
    schedule_masks(lane, n, q0, q1, AW, BUS);
    const M96 EV = { __ballot(lane < n && q0.mask == ESIM_MASK_EVERYWHERE), (uint32_t)__ballot(64u + lane < n && q1.mask == ESIM_MASK_EVERYWHERE) };
    const ChunkT ct = chunk_t(d, t0, n);
    if (d_lo < d_hi) sl_cur = fetch_slot(d, FX(id_cur, LANE_HSLOT), lane);
    if (have) { const uint32_t r = code_l >> 7; off_l = d.route_off[r]; sz_l = d.route_off[r + 1] - off_l; }
#ifdef ESIM_WAVE_PROFILE
    if (lane == 0) ws.rounds = 0u;
#endif
    uint32_t pst[5] = { 0u, 0u, 0u, 0u, 0u };
    const uint32_t pt1 = PROF_NOW();
    // (1) buildings and school rooms: one wavefront per item
    for (uint32_t v = d_lo; v < d_hi; ++v) {
        const uint32_t x = merge_fetch(id_cur, sl_cur, lane);
        id_cur = id_nxt;
        if (v + 2u < d_hi) id_nxt = fetch_item(d, id_of(v + 2u), lane);
        if (v + 1u < d_hi) sl_cur = fetch_slot(d, FX(id_cur, LANE_HSLOT), lane);
        const uint32_t pq0 = PROF_NOW();
        const ItemFetch it = decode_item<PM>(d, x, lane, n, ct, AW, BUS);
        if (it.slot == ITEM_UNUSED) continue;
        if (!item_ok(d, it)) { if (lane == 0) RAISE(ctrl, ESIM_ERANGE, ERR_AT_ITEM_CHECK); continue; }
        // (persistent map: an item whose records all lie outside the chunk -- recovered, or at home under a lockdown -- is left
        // here, before any member is fetched)
        if (PM && !__any((it.c0 | it.c1) != 0u)) continue;
        if (it.id >= route_base) {
            if (PM) {
                // the bus steps in which an Infected rider of this route is on the bus: c0 / c1 count them per step
                const uint32_t r = it.id - route_base, sz = it.a_hi - it.a_lo;
                if (r >= d.n_routes || it.a_hi < it.a_lo || it.a_hi > d.n_pt) { if (lane == 0) RAISE(ctrl, ESIM_ERANGE, ERR_AT_ROUTE_ITEM); continue; }
                const unsigned long long m_lo = __ballot(it.c0 != 0u) & BUS.lo;
                const uint32_t m_hi = (uint32_t)__ballot(lane < FREE_MAX - 64u && it.c1 != 0u) & BUS.hi;
                if (sz <= 64u) {
                    // (registered for k_chunk_units, which deals the pairs of all wavefronts out evenly -- ranked right here they
                    // made the slowest wavefront twice the median --; a stretch that is full: ranked here after all)
                    const uint32_t np = (uint32_t)__popcll(m_lo) + (uint32_t)__popc(m_hi), rp_cap = 2u * d.items_cap / SUBQ, qr = wave & (SUBQ - 1u);
                    uint32_t at = 0u;
                    if (lane == 0) at = atomicAdd(&d.hot[(HOT_RPAIRS + qr) * HOT_STRIDE], np);      // (one reservation per route item, 64 lists)
                    at = FX(at, 0);
                    uint32_t *list = d.route_pairs + (size_t)qr * rp_cap;
                    uint32_t i = 0u;
                    for (unsigned long long m = m_lo; m; m &= m - 1ull, ++i) {
                        const uint32_t j = (uint32_t)__builtin_ctzll(m);
                        if (at + i < rp_cap) { if (lane == 0) list[at + i] = (r << 7) | j; } else route_pair_small(d, ctrl, sm, it.a_lo, sz, j, t0, lane WORK_PASS);
                    }
                    for (uint32_t m = m_hi; m; m &= m - 1u, ++i) {
                        const uint32_t j = 64u + (uint32_t)__builtin_ctz(m);
                        if (at + i < rp_cap) { if (lane == 0) list[at + i] = (r << 7) | j; } else route_pair_small(d, ctrl, sm, it.a_lo, sz, j, t0, lane WORK_PASS);
                    }
                } else {
                    const uint32_t np = (uint32_t)__popcll(m_lo) + (uint32_t)__popc(m_hi);
                    uint32_t at = 0u;
                    if (lane == 0) at = atomicAdd(&d.hot[HOT_BIGPAIRS * HOT_STRIDE], np);
                    at = FX(at, 0);
                    if (at + np > 2u * d.items_cap) { if (lane == 0) RAISE(ctrl, ESIM_ERANGE, ERR_AT_BIGPAIRS); continue; }
                    uint32_t i = 0u;
                    for (unsigned long long m = m_lo; m; m &= m - 1ull, ++i) if (lane == 0) d.route_pairs_big[at + i] = (r << 7) | (uint32_t)__builtin_ctzll(m);
                    for (uint32_t m = m_hi; m; m &= m - 1u, ++i) if (lane == 0) d.route_pairs_big[at + i] = (r << 7) | (64u + (uint32_t)__builtin_ctz(m));
                }
            }
            continue;
        }
        const uint32_t pi0 = PROF_NOW();
        pst[0] += pi0 - pq0;
        (void)pi0; ++p_items; WORK_ADD(WK_ITEMS, lane == 0 ? 1 : 0);
        if (it.id < d.n_bld) {
            if (it.aux == ESIM_SCHOOL) continue;                              // School::find_exposures works per room
            // first 64 residents and workers and their words: both lists' loads are in flight together
            const uint32_t n_res = it.a_hi - it.a_lo, n_wrk = it.b_hi - it.b_lo;
            uint32_t rm = 0u, wm = 0u, rw = 0u, ww = 0u;
            if (lane < n_res) rm = d.res_idx ? d.res_idx[it.a_lo + lane] : it.a_lo + lane;
            if (lane < n_wrk) wm = d.wrk_idx[it.b_lo + lane];
            if (lane < n_res) rw = d.cit[rm];
            if (lane < n_wrk) ww = d.cit[wm];
            const uint32_t pq1 = PROF_NOW();
            const uint32_t S = item_steps_regs(it.c0, it.c1, lane, ws, t0, AW, EV);
            __builtin_amdgcn_wave_barrier();
            const uint32_t pq2 = PROF_NOW();
            // Household / Workplace::find_exposures: every registered occupant (building.rs:202-204,278-280)
            const UnitSrc src = { it.slot, it.link, FX(x, 7) };
            list_or_units(d, ctrl, sm, ws, d.res_idx, it.a_lo, it.a_hi, src, lane, 0u, S, t0 WORK_PASS, true, rm, rw);
            const uint32_t pq3 = PROF_NOW();
            list_or_units(d, ctrl, sm, ws, d.wrk_idx, it.b_lo, it.b_hi, src, lane, 1u, S, t0 WORK_PASS, true, wm, ww);
            const uint32_t pq4 = PROF_NOW();
            pst[1] += pq1 - pi0; pst[2] += pq2 - pq1; pst[3] += pq3 - pq2; pst[4] += pq4 - pq3;
        } else {
            const uint32_t n_mem = it.a_hi - it.a_lo;
            uint32_t mm = 0u, mw = 0u;
            if (lane < n_mem) mm = d.room_idx[it.a_lo + lane];
            school_counts(d, it.link, lane, n, q0, q1, ws);
            if (lane < n_mem) mw = d.cit[mm];
            const uint32_t S = item_steps_regs(it.c0, it.c1, lane, ws, t0, AW, EV);
            __builtin_amdgcn_wave_barrier();
            // School::find_exposures: the room once per infected in it (building.rs:494-522)
            const UnitSrc src = { it.slot, it.link, FX(x, 7) };
            list_or_units(d, ctrl, sm, ws, d.room_idx, it.a_lo, it.a_hi, src, lane, 2u, S, t0 WORK_PASS, true, mm, mw);
        }
        __builtin_amdgcn_wave_barrier();
        { const uint32_t dt = PROF_NOW() - pi0; p_item_max = dt > p_item_max ? dt : p_item_max; }
    }
    const uint32_t pt2 = PROF_NOW();
    // (2) routes of <= 64 riders: one wavefront per (route, bus step) with an Infected rider (the pairs dealt to this wavefront --
    // see the first look above --, taken one by one)
    for (uint32_t k0 = 0; !PM && k0 * G < K; k0 += 64u) {
        if (k0) {                                                             // (beyond the 64 looked at up front: many Infected)
            const uint32_t kk = (k0 + lane) * G + w_rep;
            have = false;
            if (kk < K) {
                const uint32_t src = (w_base + n_mw - (uint32_t)(((unsigned long long)kk * PAIR_SPREAD) % n_mw)) % n_mw;
                code_l = d.route_pairs[(size_t)src * K + kk];
                have = kk < d.pair_cnt[src] && (code_l >> 7) < d.n_routes && (code_l & 127u) < n;
            }
            if (have) { const uint32_t r = code_l >> 7; off_l = d.route_off[r]; sz_l = d.route_off[r + 1] - off_l; }
        }
        unsigned long long todo = __ballot(have);
        while (todo) {
            const int src_lane = __ffsll((long long)todo) - 1;
            todo &= todo - 1ull;
            const uint32_t code = __shfl(code_l, src_lane, 64), off = __shfl(off_l, src_lane, 64), sz = __shfl(sz_l, src_lane, 64);
            route_pair_small(d, ctrl, sm, off, sz, code & 127u, t0, lane WORK_PASS);
        }
    }
    WORK_FLUSH(d);
    const uint32_t pt3 = PROF_NOW();
#ifndef ESIM_PROFILE_UNITS
    PROF_PUT(d, 0, pt0); PROF_PUT(d, 1, pt1); PROF_PUT(d, 2, pt2); PROF_PUT(d, 3, pt3);   // start, after preamble, after items, end
    PROF_PUT(d, 4, p_items); PROF_PUT(d, 5, p_item_max); PROF_PUT(d, 6, wsc[threadIdx.x >> 6].rounds);
    PROF_PUT(d, 11, pst[0]); PROF_PUT(d, 12, pst[1]); PROF_PUT(d, 13, pst[2]); PROF_PUT(d, 14, pst[3]); PROF_PUT(d, 15, pst[4]);
#endif
    (void)pst;
    (void)pt0; (void)pt1; (void)pt2; (void)pt3; (void)p_items; (void)p_item_max;
}

// The deferred units of long member lists, dealt to the wavefronts round-robin.
// Then the routes of more than 64 riders: one workgroup per (route, bus step), ranks through LDS.
template <bool PM>
__global__ __launch_bounds__(TPB) void k_chunk_units(Dev d)
{
    __shared__ ChunkShared sm;
    __shared__ WaveScratch wsc[TPB / 64];
    __shared__ RouteShared rs;
    Ctrl *ctrl = d.ctrl;
    const uint32_t t0 = ctrl->chunk_t0, n = ctrl->chunk_ok;
    if (!ctrl->chunk_parallel || n == 0u) return;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (blockIdx.x * TPB + threadIdx.x) >> 6, n_waves = (gridDim.x * TPB) >> 6;
    const uint32_t pu0 = PROF_NOW();
    WORK_TALLY;
    uint32_t pu_n = 0u, pu_max = 0u, pu_it = 0u;
    // queue `wave & 63`, every (n_waves / 64)-th unit of it
    const uint32_t qr = wave & (SUBQ - 1u), first = wave / SUBQ, step = n_waves / SUBQ;
    const uint32_t n_units = step ? min(ld(&d.hot[(HOT_UNITS + qr) * HOT_STRIDE]), d.unit_qcap) : 0u;
    const uint32_t n_pairs = min(ld(&d.hot[HOT_BIGPAIRS * HOT_STRIDE]), 2u * d.items_cap);
    // (persistent map: the pairs of routes of <= 64 riders that k_chunk_draw registered are dealt out here, see below)
    const uint32_t rp_cap = 2u * d.items_cap / SUBQ;
    const uint32_t n_rp = (PM && step) ? min(ld(&d.hot[(HOT_RPAIRS + qr) * HOT_STRIDE]), rp_cap) : 0u;
    if (__syncthreads_or(first < n_units || first < n_rp) == 0 && n_pairs == 0u) return;
    for (uint32_t i = threadIdx.x; i < n; i += TPB) sm.dec[i] = d.dec[i];
    for (uint32_t i = threadIdx.x; i < 512u; i += TPB) sm.thr[i] = d.thr[i];
    __syncthreads();
    WaveScratch &ws = wsc[threadIdx.x >> 6];
    const uint32_t *q_words = reinterpret_cast<const uint32_t *>(d.units + (size_t)qr * d.unit_qcap);
    const Decision q0 = lane < n ? sm.dec[lane] : Decision{ 0u, 0u, 0u, 0u };
    const Decision q1 = 64u + lane < n ? sm.dec[64u + lane] : Decision{ 0u, 0u, 0u, 0u };
    M96 AW, BUS;
    schedule_masks(lane, n, q0, q1, AW, BUS);
    const M96 EV = { __ballot(lane < n && q0.mask == ESIM_MASK_EVERYWHERE), (uint32_t)__ballot(64u + lane < n && q1.mask == ESIM_MASK_EVERYWHERE) };
    const ChunkT ct = chunk_t(d, t0, n);
    // Per unit: its record (lanes 0..7 of one register); then, together, the slot's interval records, the school's, and the
    // ids of the first members its pairs touch; then those members' words.  The record of the unit after next and the
    // second stage of the next are in flight while this one draws.
    // (unit records are written by k_chunk_draw from items it has checked (item_ok), into queues that are initialised to
    // no-ops; what a record names as hash slots is checked by fetch_slot, which fetches nothing for a value that is no slot)
    auto unit_words = [&](uint32_t q) -> uint32_t { return lane < 8u ? q_words[(size_t)q * 8u + lane] : 0u; };
    auto member_id = [&](uint32_t u) -> uint32_t {
        const uint32_t code = FX(u, 4), kind = code >> 30, lo = FX(u, 2), n_mem = FX(u, 3), mf = FX(u, 6);
        if (code == UNIT_NOOP || mf + lane >= n_mem) return 0u;
        const uint32_t *idx = kind == 2u ? d.room_idx : kind == 1u ? d.wrk_idx : d.res_idx;
        return idx ? idx[lo + mf + lane] : lo + mf + lane;
    };
    const uint32_t pu1 = PROF_NOW();
    uint32_t u_0 = 0xFFFFFFFFu, u_1 = 0xFFFFFFFFu;                            // this unit, the next (code word UNIT_NOOP: none)
    uint32_t xs_0 = 0u, ys_0 = 0u, mid_0 = 0u;
    if (first < n_units) u_0 = unit_words(first);
    if (first + step < n_units) u_1 = unit_words(first + step);
    if (first < n_units && FX(u_0, 4) != UNIT_NOOP) { xs_0 = fetch_slot(d, FX(u_0, 0), lane); ys_0 = fetch_slot(d, FX(u_0, 1), lane); mid_0 = member_id(u_0); }
    for (uint32_t q = first; q < n_units; q += step) {
        const uint32_t u = u_0, xs = xs_0, ys = ys_0, mid = mid_0;
        u_0 = u_1;
        u_1 = 0xFFFFFFFFu;
        if (q + 2u * step < n_units) u_1 = unit_words(q + 2u * step);
        if (q + step < n_units && FX(u_0, 4) != UNIT_NOOP) { xs_0 = fetch_slot(d, FX(u_0, 0), lane); ys_0 = fetch_slot(d, FX(u_0, 1), lane); mid_0 = member_id(u_0); }
        const uint32_t code = FX(u, 4);
        if (code == UNIT_NOOP) continue;
        const uint32_t pui = PROF_NOW();
        ++pu_n;
        const uint32_t kind = code >> 30, p_lo = code & 0x3FFFFFFFu, slot = FX(u, 0), link = FX(u, 1), lo = FX(u, 2), n_mem = FX(u, 3), own = FX(u, 5), mf = FX(u, 6);
        const uint32_t mw = (mf + lane < n_mem) ? d.cit[mid] : 0u;
        uint32_t c0, c1;
        if (PM) pitem_counts(d, xs, slot, lane, ct, AW, BUS, false, c0, c1);
        else { item_counts(d, xs, slot, lane, n, AW, BUS, c0, c1); iv_count(own, lane, AW, BUS, c0, c1); }
        if (kind == 2u) {
            uint32_t s0 = 0u, s1 = 0u;
            if (link != 0xFFFFFFFFu) { if (PM) pitem_counts(d, ys, link, lane, ct, AW, BUS, true, s0, s1); else item_counts(d, ys, link, lane, n, AW, BUS, s0, s1); }
            ws.sch[lane] = s0;
            if (lane < FREE_MAX - 64u) ws.sch[64u + lane] = s1;
        }
        const uint32_t *idx = kind == 2u ? d.room_idx : kind == 1u ? d.wrk_idx : d.res_idx;
        const uint32_t S = item_steps_regs(c0, c1, lane, ws, t0, AW, EV);
        __builtin_amdgcn_wave_barrier();
        const uint32_t pairs = n_mem * S;
        WORK_ADD(WK_UNITS, lane == 0 ? 1 : 0);
        member_pairs(d, ctrl, sm, ws, idx, lo, p_lo, min(pairs, p_lo + UNIT_PAIRS), lane, kind, S, t0 WORK_PASS, true, mid, mw, mf);
        __builtin_amdgcn_wave_barrier();
        { const uint32_t dt = PROF_NOW() - pui; pu_max = dt > pu_max ? dt : pu_max; pu_it += (min(pairs, p_lo + UNIT_PAIRS) - p_lo + 63u) / 64u; }
    }
    const uint32_t pu2 = PROF_NOW();
    WORK_FLUSH(d);
#ifdef ESIM_PROFILE_UNITS
    PROF_PUT(d, 0, pu0); PROF_PUT(d, 1, pu1); PROF_PUT(d, 2, pu2); PROF_PUT(d, 4, pu_n); PROF_PUT(d, 5, pu_max); PROF_PUT(d, 7, pu_it);
#endif
    (void)pu0; (void)pu1; (void)pu2; (void)pu_n; (void)pu_max; (void)pu_it;
    if (PM) {
        // list `wave & 63` of the (route, bus step) pairs k_chunk_draw registered, every (n_waves / 64)-th entry of it; the next
        // pair's route is looked up while this one is ranked
        const uint32_t *list = d.route_pairs + (size_t)qr * rp_cap;
        uint32_t code = first < n_rp ? list[first] : 0u, off = 0u, sz = 0u;
        if (first < n_rp && (code >> 7) < d.n_routes) { off = d.route_off[code >> 7]; sz = d.route_off[(code >> 7) + 1u] - off; }
        for (uint32_t q = first; q < n_rp; q += step) {
            const uint32_t c_code = code, c_off = off, c_sz = sz;
            if (q + step < n_rp) { code = list[q + step]; if ((code >> 7) < d.n_routes) { off = d.route_off[code >> 7]; sz = d.route_off[(code >> 7) + 1u] - off; } else sz = 0u; }
            if ((c_code >> 7) < d.n_routes && (c_code & 127u) < n && c_sz <= 64u) route_pair_small(d, ctrl, sm, c_off, c_sz, c_code & 127u, t0, lane WORK_PASS);
        }
    }
    for (uint32_t q = blockIdx.x; q < n_pairs; q += gridDim.x) route_pair_big<TPB>(d, ctrl, sm, rs, d.route_pairs_big[q], t0, n WORK_PASS);
    WORK_FLUSH(d);
}

// Exposures per step (statistics.rs:181) from the final citizen words -- the many-workgroup form, for chunks with many new
// exposures (k_chunk_books does it itself otherwise).
__global__ __launch_bounds__(TPB) void k_chunk_count(Dev d)
{
    Ctrl *ctrl = d.ctrl;
    if (!ctrl->chunk_parallel || ctrl->chunk_ok == 0u) return;
    const uint32_t tid = blockIdx.x * TPB + threadIdx.x, r = tid & (SUBQ - 1u), step = (gridDim.x * TPB) / SUBQ;
    const uint32_t n_new = min(ld(&d.hot[(HOT_NEWEXP + r) * HOT_STRIDE]), d.newexp_cap);
    const uint32_t *list = d.newexp + (size_t)r * d.newexp_cap;
    const uint32_t t0 = ctrl->chunk_t0;
    // exposures per (step of the chunk, building | bus): counted in LDS, then added to one of EXP_ROWS rows of exp_part
    // (k_chunk_books adds the rows up and zeroes them) -- a hundred thousand atomics on the same dozen cache lines of one
    // global array are served one by one
    __shared__ uint32_t e_cnt[2u * FREE_MAX];
    __shared__ uint32_t s_cut;
    if (threadIdx.x < 2u * FREE_MAX) e_cnt[threadIdx.x] = 0u;
    if (threadIdx.x == 0) s_cut = 0xFFFFFFFFu;
    __syncthreads();
    for (uint32_t i = tid / SUBQ; i < n_new; i += step) {
        if (list[i] >= d.n) continue;
        const uint32_t w = d.cit[list[i]];
        const uint32_t j = CW_TE(w) - TE_BIAS - t0;
        if (j < FREE_MAX) atomicAdd(&e_cnt[2u * j + ((w & CW_BUS_EXPOSED) ? 1u : 0u)], 1u);
        // exposed on a bus although the plan vaccinates it later in the chunk: from this step on the plan is void (k_chunk_vax)
        if ((w & CW_BUS_EXPOSED) && CW_VAX_REL(w) != CW_VAX_NONE) { atomicMin(&s_cut, j); if (d.world > 1u) d.xc[j] = 1u; }
    }
    __syncthreads();
    if (threadIdx.x < 2u * FREE_MAX && e_cnt[threadIdx.x]) atomicAdd(&d.exp_part[(size_t)(blockIdx.x % EXP_ROWS) * 2u * FREE_MAX + threadIdx.x], e_cnt[threadIdx.x]);
    if (threadIdx.x == 0 && s_cut != 0xFFFFFFFFu) atomicMin(&ctrl->chunk_cut, s_cut);
    if (!ctrl->vax_chunk) return;
    // What the chunk's vaccinations do to the census of its later steps, from the words as the draws left them: one thread per
    // planned citizen, only the step that won counts.  A citizen vaccinated at the end of step j is Vaccinated from step j + 1
    // on instead of what its exposure step says (Susceptible; Exposed up to e_last; Infected up to i_last; Recovered after).
    // Difference arrays over the steps; events at or after a cut only touch steps that are not committed.
    __shared__ int dl[4][FREE_MAX + 2];
    for (uint32_t i = threadIdx.x; i < 4u * (FREE_MAX + 2u); i += TPB) (&dl[0][0])[i] = 0;
    __syncthreads();
    const uint32_t n = ctrl->chunk_ok;
    for (uint32_t j = blockIdx.x; j < n; j += gridDim.x) {
        const uint32_t cnt = d.vax_cnt[j];
        for (uint32_t i = threadIdx.x; i < cnt; i += TPB) {
            const uint32_t w = d.cit[d.vax_ev[(size_t)j * VACC_MAX_RATE + i]], te = CW_TE(w);
            if (CW_VAX_REL(w) != j) continue;                                  // (j = n - 1 only moves the totals after the chunk: index n)
            atomicAdd(&dl[3][j + 1u], 1);                                      // Vaccinated from j + 1 to the end
            // (exposed in a LATER step of this chunk: it was Susceptible when it was vaccinated -- only a citizen the repair of the plan
            // chose anew can look like this, and the chunk is then cut behind step j: that exposure never happened)
            if (te == TE_SUSCEPTIBLE || (te < TE_RECOVERED && te - TE_BIAS - t0 < n && te - TE_BIAS - t0 > j)) { atomicSub(&dl[0][j + 1u], 1); continue; }
            const int e_last = (int)te - (int)TE_BIAS + (int)d.exposed_time - (int)t0, i_last = e_last + 1 + (int)d.infected_time;
            const int lo = (int)j + 1;
            if (lo <= e_last) { atomicSub(&dl[1][lo], 1); atomicAdd(&dl[1][min(e_last, (int)n - 1) + 1], 1); }
            const int ilo = max(lo, e_last + 1);
            if (ilo <= i_last && ilo < (int)n) { atomicSub(&dl[2][ilo], 1); atomicAdd(&dl[2][min(i_last, (int)n - 1) + 1], 1); }
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < 4u * (FREE_MAX + 2u); i += TPB) { const int v = (&dl[0][0])[i]; if (v) atomicAdd(&d.vax_delta[i], (uint32_t)v); }
}

// The chunk's exposures enter the log grouped by step (after k_batch_finish wrote the offsets); the hash map and
// the count vectors are emptied for the next chunk.
__global__ __launch_bounds__(TPB) void k_chunk_scatter(Dev d)
{
    Ctrl *ctrl = d.ctrl;
    if (ctrl->chunk_done == 0u) return;
    // (the chunk as k_chunk_books noted it down: by now the control block may describe the next one)
    const uint32_t t0 = ctrl->prev_t0;
    const uint32_t n_items = min(ctrl->prev_n_items, d.items_cap);
    {
        const uint32_t tid = blockIdx.x * TPB + threadIdx.x, r = tid & (SUBQ - 1u), step = (gridDim.x * TPB) / SUBQ;
        const uint32_t n_new = min(d.hot[(HOT_PREV_NEWEXP + r) * HOT_STRIDE], d.newexp_cap);
        const uint32_t *list = d.newexp + (size_t)r * d.newexp_cap;
        const uint32_t n_eff = ctrl->prev_n_eff;
        for (uint32_t i = tid / SUBQ; i < n_new; i += step) {
            const uint32_t m = list[i];
            if (m >= d.n) continue;
            const uint32_t te = CW_TE(d.cit[m]);
            if (te - TE_BIAS - t0 < n_eff) d.log[d.log_off[te] + atomicAdd(&d.cursor[(blockIdx.x % EXP_ROWS) * FREE_MAX + te - TE_BIAS - t0], 1u)] = m;
            else {
                // exposed in a step that was not committed (a cut, or the disease was over before): Susceptible again; on a bus in
                // the very step of the cut: it will be again, and the next plan must know (CW_PLAN_SKIP)
                const bool again = ctrl->prev_cut && te - TE_BIAS - t0 == n_eff && (d.cit[m] & CW_BUS_EXPOSED);
                atomicOr(&d.cit[m], (TE_SUSCEPTIBLE << CW_TE_SHIFT) | (again ? CW_PLAN_SKIP : 0u));
                atomicAnd(&d.cit[m], ~CW_BUS_EXPOSED);
            }
        }
    }
    // the hash slots (and spilled count vectors) of the ids that were handed out: a thread per (wavefront of k_chunk_marks,
    // k-th id of its range), so that the whole clean-up is three dependent loads deep
    if (ctrl->prev_pmap) return;                                              // (the persistent map stays)
    const uint32_t per_wave = ctrl->prev_per_wave, n_mw = per_wave ? n_items / per_wave : 0u;
    const uint32_t tid = blockIdx.x * TPB + threadIdx.x, nth = gridDim.x * TPB;
    for (uint32_t i = tid; i < n_mw * per_wave; i += nth) {
        const uint32_t w = i / per_wave, k = i - w * per_wave;
        if (k >= d.used_cnt[w]) continue;
        const uint32_t h = d.hitems[i];
        if (h >= d.hcap) continue;                                             // ITEM_UNUSED (or no slot at all)
        const uint32_t state = d.slot_state[h];
        if (state > ITEM_RECS && d.item_rec[i].id < d.n_bld + d.n_room)       // somebody spilled into the per-step counters
            for (uint32_t j = 0; j < FREE_MAX; ++j) d.vec[(size_t)h * FREE_MAX + j] = 0u;
        d.hkey[h] = HKEY_EMPTY;
        if (state) d.slot_state[h] = 0u;
    }
}

// The planned vaccinations of the chunk k_chunk_scatter has just finished (a kernel of its own: the scatter reads the exposure
// steps this one overwrites): those of committed steps happen (simulator.rs:551: whatever the citizen was, it is Vaccinated;
// an exposure step leaves the histogram as vaccinate() does), the others are forgotten.
__global__ __launch_bounds__(TPB) void k_chunk_vax_final(Dev d)
{
    Ctrl *ctrl = d.ctrl;
    if (!ctrl->prev_vax) return;
    const uint32_t n_eff = ctrl->prev_n_eff;
    // persistent map: the cancellation records of steps that were not committed are taken back (zeroed where they stand); those of
    // committed steps stay for good.  (n_neg is reset by the plan kernel of the next chunk: other workgroups still read it here.)
    {
        const uint32_t n_neg = min(ctrl->n_neg, NEG_CAP);
        for (uint32_t i = blockIdx.x * TPB + threadIdx.x; i < n_neg; i += gridDim.x * TPB) {
            const uint32_t where = d.neg_list[2u * i], j = d.neg_list[2u * i + 1u];
            if (j < n_eff) continue;
            if (where & 0x80000000u) d.ovf[where & 0x7FFFFFFFu] = 0u; else d.slot_iv[where] = 0u;
        }
    }
    // (the plan may reach beyond the chunk: the decisions can end a chunk early, k_decide)
    const uint32_t n = ctrl->prev_planned;
    for (uint32_t j = blockIdx.x; j < n; j += gridDim.x) {
        const uint32_t cnt = d.vax_cnt[j];
        for (uint32_t i = threadIdx.x; i < cnt; i += TPB) {
            const uint32_t c = d.vax_ev[(size_t)j * VACC_MAX_RATE + i];
            const uint32_t w = d.cit[c];
            if (CW_VAX_REL(w) != j) continue;                         // another step of the chunk won, or Vaccinated before
            if (j < n_eff) {
                const uint32_t te = CW_TE(w);
                if (te < TE_RECOVERED) atomicSub(&d.hist[te], 1u);
                d.cit[c] = CW_MAKE(TE_VACCINATED, w & (CW_BUS_EXPOSED | CW_FLAGS));
            } else atomicAnd(&d.cit[c], ~CW_VAX_MASK);
        }
    }
}

// ----------------------------------------------------------------------------- k_batch_finish
// The books of a pipelined chunk [t0, t0+n): census (simulator.rs:178) by sliding the Exposed / Infected
// windows over the exposure histogram, the StatisticEntry of every step (statistics.rs:208-215, adjusted
// by citizen_exposed :275-287), hist / log offsets, and the control block as it stands after the chunk.
// (body shared by the two launch forms below)
// e: this chunk's exposure counts [2 * step of the chunk + (bus ? 1 : 0)] when the caller holds them (else d.exp_step has them);
// lo_out: receives the first log position of every step of the chunk.
// vax: the chunk ran under a vaccination programme with its vaccinations planned (k_chunk_vax): the census moves by the
// prefix sums of Dev::vax_delta, and only the steps before Ctrl::chunk_cut are committed.  Returns the steps committed.
__device__ __forceinline__ uint32_t batch_finish_body(const Dev &d, uint32_t t0, uint32_t n, const uint32_t *e = nullptr, uint32_t *lo_out = nullptr, bool vax = false,
                                                       bool marks_left = true)
{
    __shared__ uint32_t P[BF_WIN + 1];                 // P[i + 1] = sum of H[0..i], P[0] = 0
    __shared__ uint32_t wtmp[FIN_TPB / 64];
    __shared__ uint32_t n_eff_s;
    __shared__ uint32_t cum[5][FREE_MAX + 2];          // prefix sums of the vaccination deltas (S, E, I, V) and of the bus exposures
    Ctrl *ctrl = d.ctrl;
    const uint32_t tid = threadIdx.x;
    const int et = (int)d.exposed_time, it = (int)d.infected_time;
    const int base_idx = (int)(t0 + TE_BIAS) - et - 1 - it;              // lowest histogram entry any census of the chunk reads
    uint32_t n_cut = vax ? min(n, ld(&ctrl->chunk_cut)) : n;
    if (vax && d.world > 1u) {                                            // sharded: the earliest cut of any shard (buffer C, summed)
        n_cut = n;
        for (uint32_t j = 0; j < n; ++j) if (ld(&d.xc[j])) { n_cut = j; break; }
    }
    // H[i] = citizens exposed in "step" base_idx + i: the histogram before the chunk, this chunk's exposure counters inside it
    {
        const int k = base_idx + (int)tid;
        uint32_t h = 0;
        if (k >= (int)(t0 + TE_BIAS)) { const uint32_t j = (uint32_t)(k - (int)(t0 + TE_BIAS)); if (j < n) h = e ? e[2u * j] + e[2u * j + 1u] : d.exp_step[2u * (t0 + j)] + d.exp_step[2u * (t0 + j) + 1u]; }
        else if (k >= 0) h = d.hist[k];
        P[tid + 1] = h;
        if (tid == 0) { P[0] = 0u; n_eff_s = n_cut; }
        // cum[q][i] = sum of delta[q][0..i] (i.e. what applies to step i); cum[4][i] = bus exposures of steps < i.
        // (all loads at once, then one wavefront per row scans it: a thread walking a row load by load took 15-25 us)
        for (uint32_t i = tid; i < 5u * (FREE_MAX + 2u); i += FIN_TPB) {
            const uint32_t q = i / (FREE_MAX + 2u), k = i - q * (FREE_MAX + 2u);
            uint32_t v = 0u;
            if (q < 4u) v = vax ? ld(&d.vax_delta[i]) : 0u;
            else if (k < n) v = e ? e[2u * k + 1u] : d.exp_step[2u * (t0 + k) + 1u];
            cum[q][k] = v;
        }
    }
    __syncthreads();
    if (tid < 5u * 64u) {
        const uint32_t q = tid >> 6, l = tid & 63u;
        uint32_t v0 = cum[q][l], v1 = 64u + l < FREE_MAX + 2u ? cum[q][64u + l] : 0u;
        const uint32_t own0 = v0, own1 = v1;
        for (uint32_t o = 1; o < 64u; o <<= 1) { const uint32_t y0 = __shfl_up(v0, o, 64), y1 = __shfl_up(v1, o, 64); if (l >= o) { v0 += y0; v1 += y1; } }
        v1 += __shfl(v0, 63, 64);
        if (q == 4u) { v0 -= own0; v1 -= own1; }                              // (exclusive: bus exposures of the steps before)
        cum[q][l] = v0;
        if (64u + l < FREE_MAX + 2u) cum[q][64u + l] = v1;
    }
    __syncthreads();
    block_scan_1024(P + 1, wtmp);
    const uint32_t S0 = ctrl->n_susceptible, V0 = ctrl->n_vaccinated, run0 = d.log_off[t0 + TE_BIAS];
    const uint32_t elig0 = ctrl->elig_count;
    const int top0 = (int)(t0 + TE_BIAS) - base_idx;                      // index of hist[t0 + TE_BIAS] in H
    esim_step_result r;
    uint32_t exps = 0;
    if (tid < n) {
        const uint32_t s = t0 + tid;
        const int ts = top0 + (int)tid;                                   // index of this step's own entry
        exps = P[ts + 1] - P[ts];
        const uint32_t S = S0 - (P[ts] - P[top0]) + cum[0][tid];           // Susceptible before this step's exposures
        const uint32_t E = P[ts] - P[ts - et] + cum[1][tid];               // exposed in steps s - et .. s - 1 (census precedes exposures)
        const uint32_t I = P[ts - et] - P[ts - et - 1 - it] + cum[2][tid];
        const uint32_t V = V0 + cum[3][tid];
        r.time_step = s;
        if (exps > S && tid < n_cut) ctrl->error = (uint32_t)(-ESIM_ESIM); // citizen_exposed underflow, statistics.rs:275-287
        r.susceptible = S - exps; r.exposed = E + exps; r.infected = I;
        r.recovered = d.n - S - V - E - I; r.vaccinated = V;
        r.exposures_building = e ? e[2u * tid] : d.exp_step[2u * s]; r.exposures_bus = e ? e[2u * tid + 1u] : d.exp_step[2u * s + 1u];
        if (e || tid >= n_cut) {                                          // (steps that are not committed will be counted again)
            d.exp_step[2u * s] = tid < n_cut ? r.exposures_building : 0u; d.exp_step[2u * s + 1u] = tid < n_cut ? r.exposures_bus : 0u;
        }
        if (lo_out) lo_out[tid] = run0 + (P[ts] - P[top0]);
        r.lockdown = d.dec[tid + 1u].lockdown; r.vaccination_active = vax ? 1u : 0u; r.mask_status = d.dec[tid + 1u].mask;
        r.n_riders = d.dec[tid].bus_dir ? d.n_pt : 0u;
        r.vaccinated_now = vax ? d.vax_now[tid] : 0u;
        r.eligible_count = vax ? elig0 - cum[4][tid + 1u] : 0u;            // after this step's bus exposures (simulator.rs:447-449)
        r.disease_exists = (r.exposed != 0u || r.infected != 0u || r.susceptible != 0u) ? 1u : 0u;   // statistics.rs:289-291
        r.reserved = 0u;
        if (!r.disease_exists && ctrl->stop_when_done) atomicMin(&n_eff_s, tid + 1u);
    }
    __syncthreads();
    const uint32_t n_eff = n_eff_s;
    if (tid < n_eff) {
        const uint32_t s = t0 + tid;
        const int ts = top0 + (int)tid;
        d.hist[s + TE_BIAS] = exps;
        d.log_off[s + TE_BIAS + 1u] = run0 + (P[ts + 1] - P[top0]);
        if (s <= d.max_steps) d.records[s] = r;
        if (tid + 1u == n_eff) ctrl->quiet = (r.exposed == 0u && r.infected == 0u) ? 1u : 0u;
    }
    if (tid == 0) {
        ctrl->n_susceptible = S0 - (P[top0 + (int)n_eff] - P[top0]) + cum[0][n_eff];
        ctrl->n_vaccinated = V0 + cum[3][n_eff];
        if (vax) ctrl->elig_count = elig0 - cum[4][n_eff];
        ctrl->log_len = run0 + (P[top0 + (int)n_eff] - P[top0]);
        ctrl->t = t0 + n_eff; ctrl->steps_done = t0 + n_eff - 1u;
        if (n_eff < n_cut) ctrl->finished = 1u;
        else if (n_cut < n) ctrl->vax_cuts += 1u;
        // (a chunk whose plan was repaired and that is cut all the same is cut BEHIND a step whose newly chosen citizen mattered later:
        // what the attempt saw in the step of the cut is then not what will happen in it, so nobody is marked CW_PLAN_SKIP)
        ctrl->prev_cut = (n_eff == n_cut && n_cut < n && !(vax && ctrl->repair_ran)) ? 1u : 0u;
        if (n_eff) {
            ctrl->lockdown = d.dec[n_eff].lockdown; ctrl->mask = d.dec[n_eff].mask;
            ctrl->at_work = d.dec[n_eff - 1u].at_work; ctrl->bus_dir = d.dec[n_eff - 1u].bus_dir;
        }
        // ring slots: a chunk run step by step leaves the marks of its last step (the next exposure pass clears them); a chunk
        // drawn in one pass leaves none -- and the lists k_chunk_marks emptied for it must not keep their old lengths, or a
        // sequential step that comes back to that slot would walk stale entries (a route ranked twice at once)
        const uint32_t keep = marks_left ? ((t0 + n - 1u) & (MARK_SLOTS - 1u)) : MARK_SLOTS;
        for (uint32_t z = 0; z < MARK_SLOTS; ++z)
            if (z != keep) { ctrl->n_touched_bld[z] = 0u; ctrl->n_touched_room[z] = 0u; ctrl->n_touched_route[z] = 0u; ctrl->n_touched_route_big[z] = 0u; }
    }
    return n_eff;
}

__global__ __launch_bounds__(FIN_TPB) void k_batch_finish(Dev d, uint32_t t0, uint32_t n)
{
    batch_finish_body(d, t0, n);
}

// The books of a one-pass chunk, in ONE workgroup so that nothing but kernel boundaries of the wide kernels is left on
// the chunk's critical path (a kernel boundary costs ~4.5 us here, and these steps are small):
//   exposures per step (statistics.rs:181) from the final citizen words of the newly exposed
//   census, records, histogram, log offsets, control block (batch_finish_body)
//   [scatter] the new log entries in step order; hash slots of the chunk's items emptied
//   [next]    the census ahead and the decisions of the NEXT chunk (k_future + k_decide)
// It takes (t0, n) from the control block, so that the host can enqueue chunk after chunk without waiting; chunk_done tells
// k_chunk_scatter (the many-workgroup form of [scatter], used while many citizens are Infected) that the books were written.
struct BooksShared { uint32_t e_cnt[2 * FREE_MAX]; uint32_t lo_s[FREE_MAX], cur_s[FREE_MAX]; uint32_t win[BF_WIN]; uint32_t wtmp[FIN_TPB / 64]; };
__device__ __forceinline__ void books_body(const Dev &d, int fused, int do_next, uint32_t max_ahead, uint32_t limit_t, BooksShared &bs)
{
    uint32_t *e_cnt = bs.e_cnt, *lo_s = bs.lo_s, *cur_s = bs.cur_s, *win = bs.win, *wtmp = bs.wtmp;
    Ctrl *ctrl = d.ctrl;
    const uint32_t tid = threadIdx.x;
    const uint32_t pb0 = PROF_NOW();
    if (!ctrl->chunk_parallel || ctrl->chunk_ok == 0u) {
        if (tid == 0) {
            ctrl->chunk_done = 0u;
            // a plan was made but the chunk does not run: k_chunk_vax_final takes the plan's fields out of the words again
            ctrl->prev_vax = ld(&ctrl->vax_chunk); ctrl->prev_n_eff = 0u; ctrl->prev_planned = ld(&ctrl->vax_planned); ctrl->vax_chunk = 0u;
        }
        // a sharded burst all-reduces buffer F in place before every chunk: it must hold THIS shard's census again, whether
        // or not the chunk ran
        if (do_next == 2) future_body(d, max_ahead, limit_t, win, wtmp);
        return;
    }
    const uint32_t t0 = ctrl->chunk_t0, n = ctrl->chunk_ok;
    const uint32_t n_items = min(ld(&ctrl->n_items), d.items_cap);
    if (tid < 2u * FREE_MAX) e_cnt[tid] = 0u;
    if (tid < FREE_MAX) cur_s[tid] = 0u;
    __syncthreads();
    // sub-list `thread & 63` of the newly exposed, every 16th entry of it
    const uint32_t r = tid & (SUBQ - 1u);
    const uint32_t n_new = min(ld(&d.hot[(HOT_NEWEXP + r) * HOT_STRIDE]), d.newexp_cap);
    const uint32_t *list = d.newexp + (size_t)r * d.newexp_cap;
    if (fused) {
        for (uint32_t i = tid / SUBQ; i < n_new; i += FIN_TPB / SUBQ) {
            if (list[i] >= d.n) continue;
            const uint32_t w = d.cit[list[i]];
            const uint32_t j = CW_TE(w) - TE_BIAS - t0;
            if (j < FREE_MAX) atomicAdd(&e_cnt[2u * j + ((w & CW_BUS_EXPOSED) ? 1u : 0u)], 1u);
        }
    } else if (tid < 2u * n) {
        // k_chunk_count made them, in EXP_ROWS rows by workgroup.  k_chunk_scatter's workgroups visit the same citizens as their
        // namesakes there, so a row's counts are also what its workgroups will write into each step's stretch of the log: the
        // rows get their own write cursors (one shared cursor per step is a hundred thousand returning atomics on six lines)
        uint32_t a = 0u, run = 0u;
        uint32_t v[EXP_ROWS];
#pragma unroll
        for (uint32_t p = 0; p < EXP_ROWS; ++p) v[p] = d.exp_part[(size_t)p * 2u * FREE_MAX + tid];   // (all loads first: in flight together)
#pragma unroll
        for (uint32_t p = 0; p < EXP_ROWS; ++p) {
            d.exp_part[(size_t)p * 2u * FREE_MAX + tid] = 0u;
            a += v[p];
            if (!(tid & 1u)) d.cursor[p * FREE_MAX + (tid >> 1)] = run;      // (buildings + buses of the step, rows before this one)
            run += v[p] + __shfl_xor(v[p], 1, 64);
        }
        e_cnt[tid] = a;
    }
    __syncthreads();
    const uint32_t pb1 = PROF_NOW();
    const bool vax = ld(&ctrl->vax_chunk) != 0u;                              // (planned chunks always take the wide form: fused == 0)
    const uint32_t n_eff = batch_finish_body(d, t0, n, e_cnt, lo_s, vax, false);
    const uint32_t pb2 = PROF_NOW();
    if (!fused) {
        // k_chunk_scatter runs after this kernel, i.e. after the next chunk's decisions have reset what it reads: keep a copy
        if (tid < SUBQ) d.hot[(HOT_PREV_NEWEXP + tid) * HOT_STRIDE] = n_new;   // (thread r < 64 read sub-list r's length above)
        if (tid == 0) { ctrl->prev_t0 = t0; ctrl->prev_n_items = n_items; ctrl->prev_per_wave = ld(&ctrl->items_per_wave); ctrl->prev_pmap = ld(&ctrl->pmap_chunk);
                        ctrl->prev_n = n; ctrl->prev_n_eff = n_eff; ctrl->prev_vax = vax ? 1u : 0u; ctrl->prev_planned = ld(&ctrl->vax_planned); ctrl->vax_chunk = 0u; }
        // (its per-step write cursors were set above, a row per EXP_ROWS-th workgroup)
    }
    if (tid == 0) { ctrl->chunk_done = 1u; if (ld(&ctrl->pmap_chunk)) ctrl->map_t = t0 + n_eff; }   // the map now stands at the step the next chunk starts with
    if (tid < 64u) {
        // totals of the split lists, for esim_debug_counters
        uint32_t a = ld(&d.hot[(HOT_NEWEXP + tid) * HOT_STRIDE]), b = ld(&d.hot[(HOT_UNITS + tid) * HOT_STRIDE]);
        for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
        if (tid == 0) { ctrl->n_newexp = a; ctrl->n_units = b; ctrl->n_route_pairs_big = ld(&d.hot[HOT_BIGPAIRS * HOT_STRIDE]); }
    }
    __syncthreads();
    if (fused) {
        for (uint32_t i = tid / SUBQ; i < n_new; i += FIN_TPB / SUBQ) {
            const uint32_t m = list[i];
            if (m >= d.n) continue;
            const uint32_t j = CW_TE(d.cit[m]) - TE_BIAS - t0;
            if (j < FREE_MAX) d.log[lo_s[j] + atomicAdd(&cur_s[j], 1u)] = m;
        }
        // the ids each wavefront of k_chunk_marks handed out: thread t looks after the wavefronts t, t + 1024, ...; all loads
        // of a round are in flight together (this loop is nothing but memory latency)
        const uint32_t per_wave = ld(&ctrl->items_per_wave), n_mw = (per_wave && !ld(&ctrl->pmap_chunk)) ? n_items / per_wave : 0u;   // (the persistent map stays)
        for (uint32_t w0 = tid; w0 < n_mw; w0 += 4u * FIN_TPB) {
            uint32_t used[4], h[4][4];
#pragma unroll
            for (int a = 0; a < 4; ++a) { const uint32_t w = w0 + (uint32_t)a * FIN_TPB; used[a] = w < n_mw ? d.used_cnt[w] : 0u; }
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int k = 0; k < 4; ++k) h[a][k] = (uint32_t)k < used[a] ? d.hitems[(w0 + (uint32_t)a * FIN_TPB) * per_wave + (uint32_t)k] : ITEM_UNUSED;
            uint32_t st[4][4];
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int k = 0; k < 4; ++k) st[a][k] = h[a][k] < d.hcap ? d.slot_state[h[a][k]] : 0u;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const uint32_t base = (w0 + (uint32_t)a * FIN_TPB) * per_wave;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint32_t hs = h[a][k], state = st[a][k];
                    if (hs >= d.hcap) continue;
                    if (state > ITEM_RECS && d.item_rec[base + (uint32_t)k].id < d.n_bld + d.n_room)   // somebody spilled into the per-step counters
                        for (uint32_t j = 0; j < FREE_MAX; ++j) d.vec[(size_t)hs * FREE_MAX + j] = 0u;
                    d.hkey[hs] = HKEY_EMPTY;
                    if (state) d.slot_state[hs] = 0u;
                }
                for (uint32_t k = 4u; k < used[a]; ++k) {                     // (more than four ids per wavefront: many Infected)
                    const uint32_t hs = d.hitems[base + k];
                    if (hs >= d.hcap) continue;
                    const uint32_t state = d.slot_state[hs];
                    if (state > ITEM_RECS && d.item_rec[base + k].id < d.n_bld + d.n_room)
                        for (uint32_t j = 0; j < FREE_MAX; ++j) d.vec[(size_t)hs * FREE_MAX + j] = 0u;
                    d.hkey[hs] = HKEY_EMPTY;
                    if (state) d.slot_state[hs] = 0u;
                }
            }
        }
    }
    const uint32_t pb3 = PROF_NOW();
    if (do_next) {
        // 1: the next chunk's census ahead and decisions; 2: the census ahead only (sharded runs all-reduce it before deciding)
        __syncthreads();
        future_body(d, max_ahead, limit_t, win, wtmp);
        __syncthreads();
        const uint32_t pb4 = PROF_NOW();
        if (do_next == 1 && tid < 64u) decide_body(d, max_ahead, limit_t, 1);
        BOOKS_PROF(d, 4, pb4 - pb3);
    }
    BOOKS_PROF(d, 0, pb1 - pb0); BOOKS_PROF(d, 1, pb2 - pb1); BOOKS_PROF(d, 2, pb3 - pb2); BOOKS_PROF(d, 3, PROF_NOW() - pb3);
    (void)pb0; (void)pb1; (void)pb2; (void)pb3;
}

__global__ __launch_bounds__(FIN_TPB) void k_chunk_books(Dev d, int fused, int do_next, uint32_t max_ahead, uint32_t limit_t)
{
    __shared__ BooksShared bs;
    books_body(d, fused, do_next, max_ahead, limit_t, bs);
}
