// esim_kernels_tiny.h -- a whole time-parallel chunk in ONE launch of ONE workgroup, for chunks with few Infected.
//
// A chunk of the wide form is seven kernels (census ahead, decisions, marks, fold, draw, units, books); with a handful of Infected
// each of them is nothing but a chain of dependent memory round trips and a kernel boundary (round 3: 52 us of device time for a
// chunk with 7 Infected, 3 us of host time per launch).  Here the same chunk is: [census ahead + decisions, when the chunk before
// did not leave them], the Infected of the chunk and their keys into LDS, the distinct keys found by comparison (no hash map in
// memory, nothing to clean up), one wavefront per item for the draws (the same member_pairs / route_pair_small / route_pair_big as
// the wide form, so the draws are the same draws), then the books with the next chunk's decisions (books_body).
// It computes what k_chunk_marks -> k_chunk_fold -> k_chunk_draw -> k_chunk_units compute: an item's Infected per step are summed
// from the interval records of the entries that name its key (iv_count), a room's school from those that name its building.
// A chunk it cannot take (more than TINY_E Infected, a vaccination plan, shards) it turns into a no-op: the steps do not advance,
// the host sees that in its read-back and enqueues the wide form (esim_api.hip run_steps).
#pragma once

#define TINY_E 64u                 // Infected (log entries) of a chunk the one-workgroup form takes
#define TINY_WAVES (FIN_TPB / 64u)
#define TINY_INLINE 256u           // member lists of more (member, slot) pairs than this are cut into units of that many, which all wavefronts share
#define TINY_TASKS 512u
#define TINY_NONE 0xFFFFFFFFu
// What one wavefront of the workgroup wrote to memory, the others read after this: the writes have reached the L2 (the workgroup's
// waves share one CU and one L2: no write-back is needed), the readers drop what their L1 holds of it (a citizen word cached before
// an atomicMin on it).
#define TINY_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __syncthreads(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); } while (0)

struct TinyShared {
    uint32_t e_c[TINY_E], e_w[TINY_E], e_iv[TINY_E];      // citizen, word, its interval record as left in a home (iv | IV_AS_WORK: at work)
    __attribute__((aligned(16))) uint32_t e_key[TINY_E][4];   // home building, work building, n_bld + room, n_bld + n_room + route (TINY_NONE: none)
    uint32_t uniq[4u * TINY_E];                           // the distinct keys
    uint32_t n_uniq, n_tasks, n_big;
    uint32_t task[TINY_TASKS][3];                         // units of long member lists: key, kind, first pair
    uint32_t big[64];                             // (route of more than 64 riders, bus step) pairs: route << 7 | step
};

// Infected per step of the chunk in the item `key` (role 0: those who live there, 1: who work there, 2: the room; lanes = steps).
__device__ __forceinline__ void tiny_counts(const TinyShared &ts, uint32_t E, uint32_t key, uint32_t lane, const M96 &AW, const M96 &BUS,
                                            uint32_t &c0, uint32_t &c1)
{
    c0 = 0u; c1 = 0u;
    for (uint32_t e = 0; e < E; ++e) {
        const uint32_t iv = ts.e_iv[e];                                       // (uniform: LDS broadcasts)
        if (ts.e_key[e][0] == key) iv_count(iv, lane, AW, BUS, c0, c1);
        if (ts.e_key[e][1] == key || ts.e_key[e][2] == key) iv_count(iv | IV_AS_WORK, lane, AW, BUS, c0, c1);
    }
}

// One member list of an item, pairs [p_lo, p_hi) of it.
__device__ __forceinline__ void tiny_list(const Dev &d, Ctrl *ctrl, const ChunkShared &sm, WaveScratch &ws, const uint32_t *idx, uint32_t lo,
                                          uint32_t p_lo, uint32_t p_hi, uint32_t lane, uint32_t kind, uint32_t S, uint32_t t0 WORK_ARG)
{
    if (p_lo < p_hi) member_pairs(d, ctrl, sm, ws, idx, lo, p_lo, p_hi, lane, kind, S, t0 WORK_PASS);
}

__global__ __launch_bounds__(FIN_TPB) void k_chunk_tiny(Dev d, int do_first, int do_next, uint32_t max_ahead, uint32_t limit_t)
{
    __shared__ BooksShared bs;
    __shared__ ChunkShared sm;
    __shared__ TinyShared ts;
    __shared__ __align__(16) unsigned char pool[sizeof(WaveScratch) * TINY_WAVES > sizeof(RouteShared) ? sizeof(WaveScratch) * TINY_WAVES : sizeof(RouteShared)];
    WaveScratch *wsc = reinterpret_cast<WaveScratch *>(pool);
    RouteShared &rs = *reinterpret_cast<RouteShared *>(pool);
    Ctrl *ctrl = d.ctrl;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    TINY_PROF(d, 0);
    if (do_first) {
        future_body(d, max_ahead, limit_t, bs.win, bs.wtmp);
        TINY_SYNC();
        TINY_PROF(d, 1);
        if (tid < 64u) decide_body(d, max_ahead, limit_t, 1);
        TINY_SYNC();
    }
    TINY_PROF(d, 2);
    const uint32_t t0 = ld(&ctrl->chunk_t0), n = ld(&ctrl->chunk_ok);
    const uint32_t i0 = ld(&ctrl->chunk_i0), i1 = ld(&ctrl->chunk_i1);
    const uint32_t E = i1 - i0;
    bool run = ld(&ctrl->chunk_parallel) != 0u && n != 0u;
    const unsigned long long key_space = (unsigned long long)d.n_bld + d.n_room + d.n_routes;
    if (run && (i1 < i0 || E > TINY_E || ld(&ctrl->vax_chunk) != 0u || ld(&ctrl->have_elig) != 0u || d.world > 1u || ld(&ctrl->chunk_bus) > 8u || key_space >= 0xFFFFFFFFull || n > FREE_MAX)) {
        // not a chunk for this form: nobody advances (books_body: "the chunk does not run"), the host enqueues the wide form
        __syncthreads();
        if (tid == 0) ctrl->chunk_parallel = 0u;
        TINY_SYNC();
        run = false;
    }
    if (run) {
        WORK_TALLY;
        {
            // the marks of step t0 - 1 (made by a sequential or pipelined step) would have been cleared by the exposure pass of
            // step t0; this chunk has none, so clear them here (as k_chunk_marks does)
            const uint32_t q = (t0 + MARK_SLOTS - 1u) & (MARK_SLOTS - 1u);
            const uint32_t ob = ctrl->n_touched_bld[q], orr = ctrl->n_touched_room[q], ort = ctrl->n_touched_route[q], orb = ctrl->n_touched_route_big[q];
            for (uint32_t i = tid; i < ob; i += FIN_TPB) d.cnt_bld[q][d.touched_bld[q][i]] = 0u;
            for (uint32_t i = tid; i < orr; i += FIN_TPB) d.cnt_room[q][d.touched_room[q][i]] = 0u;
            for (uint32_t i = tid; i < ort; i += FIN_TPB) d.route_flag[q][d.touched_route[q][i]] = 0u;
            for (uint32_t i = tid; i < orb; i += FIN_TPB) d.route_flag[q][d.touched_route_big[q][i]] = 0u;
        }
        for (uint32_t i = tid; i < n; i += FIN_TPB) sm.dec[i] = d.dec[i];
        for (uint32_t i = tid; i < 512u; i += FIN_TPB) sm.thr[i] = d.thr[i];
        if (tid == 0) { ts.n_uniq = 0u; ts.n_tasks = 0u; ts.n_big = 0u; }
        __syncthreads();
        TINY_PROF(d, 3);
        const Decision q0 = lane < n ? sm.dec[lane] : Decision{ 0u, 0u, 0u, 0u };
        const Decision q1 = 64u + lane < n ? sm.dec[64u + lane] : Decision{ 0u, 0u, 0u, 0u };
        M96 AW, BUS;
        schedule_masks(lane, n, q0, q1, AW, BUS);
        const M96 EV = { __ballot(lane < n && q0.mask == ESIM_MASK_EVERYWHERE), (uint32_t)__ballot(64u + lane < n && q1.mask == ESIM_MASK_EVERYWHERE) };
        // (1) the chunk's Infected, one thread each: the stretch of the chunk in which each is Infected, where it stands in it, its keys
        if (tid < TINY_E) {
            uint32_t c = 0u, w = 0u, iv = 0u, key[4] = { TINY_NONE, TINY_NONE, TINY_NONE, TINY_NONE };
            if (tid < E) {
                c = d.log[i0 + tid];
                if (c < d.n) {
                    w = d.cit[c];
                    const int a_abs = (int)CW_TE(w) - (int)TE_BIAS + (int)d.exposed_time + 1;
                    const int b_rel = a_abs + (int)d.infected_time - (int)t0;
                    const uint32_t iv_a = a_abs > (int)t0 ? (uint32_t)(a_abs - (int)t0) : 0u;
                    const uint32_t iv_b = b_rel < 0 ? 0u : min(min((uint32_t)b_rel, n - 1u), CW_VAX_REL(w));
                    const bool act = !(CW_TE(w) >= TE_RECOVERED || b_rel < 0 || iv_a > iv_b);
                    if (act) {
                        const M96 I = m96_range(iv_a, iv_b);
                        const M96 onbus = (w & FL_USES_PT) ? m96_and(I, BUS) : M96{ 0ull, 0u };
                        const M96 rest = m96_andn(I, onbus);
                        const M96 atw = (w & FL_HAS_WORK) ? m96_and(rest, AW) : M96{ 0ull, 0u };
                        const M96 ath = m96_andn(rest, atw);
                        const bool any_home = m96_any(ath), any_work = m96_any(atw), any_bus = m96_any(onbus);
                        if (any_home || any_work || any_bus) {
                            WORK_ADD(WK_ENTRIES, 1);
                            const uint4 k4 = d.where4[c];
                            if (any_home) key[0] = k4.x;
                            if (any_work) key[1] = k4.y;
                            if (any_work && (w & FL_WORK_SCHOOL) && k4.z != 0xFFFFFFFFu) key[2] = d.n_bld + k4.z;
                            if (any_bus) key[3] = d.n_bld + d.n_room + k4.w;
                            iv = IV_VALID | iv_a | (iv_b << 7) | ((w & FL_USES_PT) ? IV_PT : 0u) | ((w & FL_HAS_WORK) ? IV_HW : 0u);
                        }
                    }
                }
            }
            ts.e_c[tid] = c; ts.e_w[tid] = w; ts.e_iv[tid] = iv;
            ts.e_key[tid][0] = key[0]; ts.e_key[tid][1] = key[1]; ts.e_key[tid][2] = key[2]; ts.e_key[tid][3] = key[3];
        }
        __syncthreads();
        TINY_PROF(d, 4);
        // (2) the distinct keys: a key is listed by its first occurrence
        if (tid < 4u * TINY_E) {
            const uint32_t key = (&ts.e_key[0][0])[tid];
            bool first = key != TINY_NONE;
            const uint32_t me = tid >> 2, mk = tid & 3u;
            for (uint32_t e = 0; e <= me && first; ++e) {                      // (an entry's four keys in one LDS read)
                const uint4 k4 = *reinterpret_cast<const uint4 *>(&ts.e_key[e][0]);
                const uint32_t upto = e < me ? 4u : mk;                         // keys of entry e that come before mine
                if ((upto > 0u && k4.x == key) || (upto > 1u && k4.y == key) || (upto > 2u && k4.z == key) || (upto > 3u && k4.w == key)) first = false;
            }
            if (first) { WORK_ADD(WK_KEYS, 1); ts.uniq[atomicAdd(&ts.n_uniq, 1u)] = key; }
        }
        __syncthreads();
        TINY_PROF(d, 5);
        // (3) one wavefront per item
        WaveScratch &ws = wsc[wv];
        const uint32_t n_uniq = ts.n_uniq, route_base = d.n_bld + d.n_room;
        for (uint32_t u = wv; u < n_uniq; u += TINY_WAVES) {
            const uint32_t key = ts.uniq[u];
            if (key >= route_base) {
                // a route: the bus steps of the chunk in which one of its riders is Infected and aboard
                const uint32_t r = key - route_base;
                if (r >= d.n_routes) { if (lane == 0) RAISE(ctrl, ESIM_ERANGE, ERR_AT_ROUTE_ITEM); continue; }
                M96 ride = { 0ull, 0u };
                for (uint32_t e = 0; e < E; ++e) if (ts.e_key[e][3] == key) { const uint32_t iv = ts.e_iv[e]; const M96 I = m96_range(iv & 127u, (iv >> 7) & 127u); ride.lo |= I.lo & BUS.lo; ride.hi |= I.hi & BUS.hi; }
                const uint32_t off = d.route_off[r], sz = d.route_off[r + 1u] - off;
                for (uint32_t half = 0; half < 2u; ++half)
                    for (unsigned long long m = half ? (unsigned long long)ride.hi : ride.lo; m; m &= m - 1ull) {
                        const uint32_t j = 64u * half + (uint32_t)__builtin_ctzll(m);
                        if (sz <= 64u) route_pair_small(d, ctrl, sm, off, sz, j, t0, lane WORK_PASS);
                        else if (lane == 0) { const uint32_t at = atomicAdd(&ts.n_big, 1u); if (at < 64u) ts.big[at] = (r << 7) | j; else RAISE(ctrl, ESIM_ERANGE, ERR_AT_BIGPAIRS); }
                    }
                continue;
            }
            uint32_t c0, c1;
            tiny_counts(ts, E, key, lane, AW, BUS, c0, c1);
            WORK_ADD(WK_ITEMS, lane == 0 ? 1 : 0);
            if (key < d.n_bld) {
                const BldRec b = d.bld8[key];
                if (b.type == ESIM_SCHOOL) continue;                            // School::find_exposures works per room
                if (b.res_lo > b.res_hi || b.res_hi > d.n || b.wrk_lo > b.wrk_hi || b.wrk_hi > d.n_wrk_idx) { if (lane == 0) RAISE(ctrl, ESIM_ERANGE, ERR_AT_ITEM_CHECK); continue; }
                // (the first 64 residents and workers and their words: both lists' loads in flight together, as in k_chunk_draw)
                const uint32_t n_res = b.res_hi - b.res_lo, n_wrk = b.wrk_hi - b.wrk_lo;
                uint32_t rm = 0u, wm = 0u, rw = 0u, ww = 0u;
                if (lane < n_res) rm = d.res_idx ? d.res_idx[b.res_lo + lane] : b.res_lo + lane;
                if (lane < n_wrk) wm = d.wrk_idx[b.wrk_lo + lane];
                if (lane < n_res) rw = d.cit[rm];
                if (lane < n_wrk) ww = d.cit[wm];
                const uint32_t S = item_steps_regs(c0, c1, lane, ws, t0, AW, EV);
                __builtin_amdgcn_wave_barrier();
                const uint32_t pr = n_res * S, pw = n_wrk * S;
                if (pr && pr <= TINY_INLINE) member_pairs(d, ctrl, sm, ws, d.res_idx, b.res_lo, 0u, pr, lane, 0u, S, t0 WORK_PASS, true, rm, rw);
                else if (!pr) { }
                else {
                    // units of TINY_INLINE pairs for everybody; what the queue cannot hold this wavefront draws itself
                    const uint32_t nu = (pr + TINY_INLINE - 1u) / TINY_INLINE; uint32_t at = 0u; if (lane == 0) at = atomicAdd(&ts.n_tasks, nu); at = FX(at, 0);
                    const uint32_t fit = at >= TINY_TASKS ? 0u : min(nu, TINY_TASKS - at);
                    for (uint32_t q = lane; q < fit; q += 64u) { ts.task[at + q][0] = key; ts.task[at + q][1] = 0u; ts.task[at + q][2] = q * TINY_INLINE; }
                    if (fit < nu) tiny_list(d, ctrl, sm, ws, d.res_idx, b.res_lo, fit * TINY_INLINE, pr, lane, 0u, S, t0 WORK_PASS);
                }
                if (pw && pw <= TINY_INLINE) member_pairs(d, ctrl, sm, ws, d.wrk_idx, b.wrk_lo, 0u, pw, lane, 1u, S, t0 WORK_PASS, true, wm, ww);
                else if (!pw) { }
                else {
                    // units of TINY_INLINE pairs for everybody; what the queue cannot hold this wavefront draws itself
                    const uint32_t nu = (pw + TINY_INLINE - 1u) / TINY_INLINE; uint32_t at = 0u; if (lane == 0) at = atomicAdd(&ts.n_tasks, nu); at = FX(at, 0);
                    const uint32_t fit = at >= TINY_TASKS ? 0u : min(nu, TINY_TASKS - at);
                    for (uint32_t q = lane; q < fit; q += 64u) { ts.task[at + q][0] = key; ts.task[at + q][1] = 1u; ts.task[at + q][2] = q * TINY_INLINE; }
                    if (fit < nu) tiny_list(d, ctrl, sm, ws, d.wrk_idx, b.wrk_lo, fit * TINY_INLINE, pw, lane, 1u, S, t0 WORK_PASS);
                }
            } else {
                const uint32_t r = key - d.n_bld;
                const uint32_t a_lo = d.room_off[r], a_hi = d.room_off[r + 1u], sch = d.room_bld[r];
                if (a_lo > a_hi || a_hi > d.n_room_idx) { if (lane == 0) RAISE(ctrl, ESIM_ERANGE, ERR_AT_ITEM_CHECK); continue; }
                uint32_t s0, s1;
                tiny_counts(ts, E, sch, lane, AW, BUS, s0, s1);                 // infected in the whole school, per step
                ws.sch[lane] = s0;
                if (lane < FREE_MAX - 64u) ws.sch[64u + lane] = s1;
                uint32_t mm = 0u, mw = 0u;
                if (lane < a_hi - a_lo) { mm = d.room_idx[a_lo + lane]; mw = d.cit[mm]; }
                const uint32_t S = item_steps_regs(c0, c1, lane, ws, t0, AW, EV);
                __builtin_amdgcn_wave_barrier();
                const uint32_t pm = (a_hi - a_lo) * S;
                if (pm && pm <= TINY_INLINE) member_pairs(d, ctrl, sm, ws, d.room_idx, a_lo, 0u, pm, lane, 2u, S, t0 WORK_PASS, true, mm, mw);
                else if (!pm) { }
                else {
                    // units of TINY_INLINE pairs for everybody; what the queue cannot hold this wavefront draws itself
                    const uint32_t nu = (pm + TINY_INLINE - 1u) / TINY_INLINE; uint32_t at = 0u; if (lane == 0) at = atomicAdd(&ts.n_tasks, nu); at = FX(at, 0);
                    const uint32_t fit = at >= TINY_TASKS ? 0u : min(nu, TINY_TASKS - at);
                    for (uint32_t q = lane; q < fit; q += 64u) { ts.task[at + q][0] = key; ts.task[at + q][1] = 2u; ts.task[at + q][2] = q * TINY_INLINE; }
                    if (fit < nu) tiny_list(d, ctrl, sm, ws, d.room_idx, a_lo, fit * TINY_INLINE, pm, lane, 2u, S, t0 WORK_PASS);
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        __syncthreads();
        TINY_PROF(d, 6);
        // (4) the units of the long member lists, dealt to the wavefronts round-robin
        const uint32_t n_tasks = min(ts.n_tasks, TINY_TASKS);
        for (uint32_t q = wv; q < n_tasks; q += TINY_WAVES) {
            const uint32_t key = ts.task[q][0], kind = ts.task[q][1], p_lo = ts.task[q][2];
            uint32_t c0, c1, lo, n_mem;
            const uint32_t *idx;
            tiny_counts(ts, E, key, lane, AW, BUS, c0, c1);
            if (kind == 2u) {
                const uint32_t r = key - d.n_bld;
                uint32_t s0, s1;
                tiny_counts(ts, E, d.room_bld[r], lane, AW, BUS, s0, s1);
                ws.sch[lane] = s0;
                if (lane < FREE_MAX - 64u) ws.sch[64u + lane] = s1;
                lo = d.room_off[r]; n_mem = d.room_off[r + 1u] - lo; idx = d.room_idx;
            } else {
                const BldRec b = d.bld8[key];
                lo = kind ? b.wrk_lo : b.res_lo; n_mem = (kind ? b.wrk_hi : b.res_hi) - lo; idx = kind ? d.wrk_idx : d.res_idx;
            }
            const uint32_t S = item_steps_regs(c0, c1, lane, ws, t0, AW, EV);
            __builtin_amdgcn_wave_barrier();
            const uint32_t P = n_mem * S;
            WORK_ADD(WK_UNITS, lane == 0 ? 1 : 0);
            tiny_list(d, ctrl, sm, ws, idx, lo, p_lo, min(P, p_lo + TINY_INLINE), lane, kind, S, t0 WORK_PASS);
            __builtin_amdgcn_wave_barrier();
        }
        __syncthreads();                                                      // (the wavefronts' scratch becomes the routes' LDS)
        TINY_PROF(d, 7);
        // (5) routes of more than 64 riders: the whole workgroup per (route, bus step)
        const uint32_t n_big = min(ts.n_big, 64u);
        for (uint32_t q = 0; q < n_big; ++q) route_pair_big<FIN_TPB>(d, ctrl, sm, rs, ts.big[q], t0, n WORK_PASS);
        WORK_FLUSH(d);
        TINY_SYNC();
    }
    TINY_PROF(d, 8);
    // (6) the books (exposures per step, census, records, log), and the census ahead and decisions of the next chunk
    books_body(d, 1, do_next, max_ahead, limit_t, bs);
    TINY_PROF(d, 9);
}
