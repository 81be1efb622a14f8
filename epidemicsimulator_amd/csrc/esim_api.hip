// esim_api.hip -- host side of libesim: the C ABI of include/esim.h over the kernels of
// esim_kernels.hip.  No CPU compute path exists here: without a HIP device every entry
// point that would compute fails with ESIM_ENODEVICE.
#include "esim_kernels.hip"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace {

thread_local std::string g_create_error;

struct esim_ctx_impl {
    esim_params P;
    Dev d;
    bool uploaded = false;
    hipStream_t stream = nullptr;
    bool own_stream = true;
    std::string err;
    // host copies needed for reset
    std::vector<uint32_t> init_state;
    std::vector<uint32_t> init_log;       // distinct seeds
    size_t cnt_bytes = 0;
    uint32_t *cnt_base = nullptr;
    uint32_t n_routes = 0;
    size_t xa_n = 0, xb_n = 0, xf_n = 0;
    uint32_t last_chunk_pairs = 0;               // Infected during the chunk last looked at (picks the form of the chunk's book-keeping)
    uint32_t free_limit = 0, free_first = 0;     // open burst of decoupled chunks: last step it may reach, first step
    uint32_t host_t = 1;          // next time step to enqueue
    uint64_t pop_hash = 0;        // of the uploaded population arrays: a checkpoint only goes back into the population it came from
    // device allocations
    std::vector<void *> allocs;
    // timing
    bool phase_timing = false, kernel_timing = false;
    uint32_t kernel_timing_stride = 16;
    bool timing_this_step = false;
    hipEvent_t ev[5] = { nullptr, nullptr, nullptr, nullptr, nullptr };
    double phase_s[3] = { 0, 0, 0 };
    std::vector<hipEvent_t> kev;       // two per timed step: before k_infected, after k_finish
    size_t kev_used = 0;
    uint32_t grid_citizens = 1, grid_infected = 1, grid_expose = 1;
    bool time_parallel = true;         // draw all steps of a chunk in one pass when its marks fit the hash map
    hipEvent_t cev[2] = { nullptr, nullptr }; double chunk_ms = 0; uint64_t chunk_steps = 0, chunk_count = 0;
    uint32_t grid_chunk = 1024;
    // persistent item map (unsharded contexts): the host's view of it -- valid as long as nothing but map-maintaining chunk passes
    // has been enqueued since it was (re)built; a rebuild every pmap_rebuild_every chunks sheds the items of the recovered
    bool pmap = false, map_valid = false, pmap_used = false;      // (off by default: measured slower than the per-chunk rebuild)
    uint32_t pmap_since_rebuild = 0, pmap_rebuild_every = 4;
    uint64_t vax_chunk_repairs = 0;
    bool quiet = false;                         // Ctrl::quiet at the last read-back of a burst of chunk passes
    bool repair_armed = false;                  // ... its two kernels are enqueued from the first cut of a run on (York never has one: 11 us a chunk saved)
    bool vax_repair_always = false;             // ESIM_VAX_REPAIR=2: from the start
    bool vax_repair = true;                     // planned chunks: repair the plan after bus exposures instead of cutting the chunk (ESIM_VAX_REPAIR=0: cut)
    uint32_t tiny_pairs = 2048;                 // chunks with at most this many (Infected, step) pairs at the last read-back run as ONE kernel (k_chunk_tiny; 0: off)
    uint32_t small_grid = 64, small_mult = 4;  // chunks with few Infected: workgroups of the marks / fold kernels, multiplier of the draw kernels (0: off)
    uint32_t draw_mult = 4, units_mult = 4;   // k_chunk_draw / k_chunk_units run this many times the marks grid: more, shorter wavefronts than the chip holds at once
    bool pipeline = true;              // run chunks of steps as one kernel per step while no vaccination programme runs
    bool vax_chunks = true;            // time-parallel chunks also under a vaccination programme (their vaccinations planned ahead, k_chunk_vax)
    uint64_t vax_chunk_steps = 0, vax_chunk_cuts = 0;
    bool elig_seen = false;            // the last control block read back had an eligible set (a vaccination programme runs)
    // the exchange between shards (esim_comm_*): RCCL owned by the library, or a caller's all-reduce
    int comm_rank = 0, comm_world = 1;
    ncclComm_t nccl = nullptr;
    esim_allreduce_fn comm_fn = nullptr; void *comm_user = nullptr;
    std::vector<uint32_t> comm_stage;
    uint32_t *xr = nullptr; size_t xr_n = 0;      // records exchange (sharded chunks)
    uint64_t shard_chunk_steps = 0, shard_step_steps = 0;
    uint64_t comm_calls = 0;
    bool xs_a2a = true;                 // the commuter exchange as an all-to-all of owner-addressed segments (ESIM_XS_MODE=gather: all-gather)
    uint32_t *xs_out = nullptr, *shared_mask = nullptr;
    // pinned host mirrors: the control block and the records of the call in flight come back with ONE stream wait (two blocking
    // copies into pageable memory cost more than a small chunk's kernels)
    Ctrl *pin_ctrl = nullptr;
    esim_step_result *pin_rec = nullptr; size_t pin_rec_n = 0;
    uint32_t pin_first = 0, pin_valid = 0;        // records [pin_first, pin_first + pin_valid) of the call in flight are in pin_rec
    bool pin_track = false;
    bool host_trace = false;                      // ESIM_TRACE_HOST: esim_run prints where its host time went (stderr)
    std::vector<std::pair<const char *, double>> ht;
    bool ctrl_fresh = false;                      // pin_ctrl holds the control block as it stands (nothing was enqueued since)
    uint32_t stop_flag_dev = 0;                   // what ctrl->stop_when_done holds (written only when it changes)
    // per-kernel device time of the chunk pass (esim_enable_chunk_kernel_timing): an event in front of every kernel of a chunk
    bool kdetail = false;
    std::vector<hipEvent_t> kdev; std::vector<int> kd_kind; size_t kd_used = 0;
    double kd_ms[ESIM_CK_N] = { 0 }; uint64_t kd_calls[ESIM_CK_N] = { 0 };
    double comm_timeout_s = 60.0;      // deadline of a host wait on a stream that holds collectives (esim_comm_set_timeout)
    std::vector<hipEvent_t> fev; size_t fev_used = 0;                               // chunks of an open decoupled burst
    std::vector<hipEvent_t> pkev; size_t pkev_used = 0; uint64_t pipe_steps = 0;   // sampled k_pipe launches
    uint32_t small_max = 128;          // infected-slice length up to which the persistent single-workgroup kernel runs a step
    hipEvent_t sev[2] = { nullptr, nullptr };   // k_small timing
    double small_ms = 0; uint64_t small_steps = 0;
};

#define CTX(c) (reinterpret_cast<esim_ctx_impl *>(c))

void comm_release(esim_ctx_impl *c);     // (defined with the exchange, below)

int fail(esim_ctx_impl *c, int code, const std::string &msg)
{
    if (c) c->err = msg; else g_create_error = msg;
    return code;
}

#define HIP_TRY(c, expr)                                                                         \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            return fail(c, ESIM_ENODEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));   \
    } while (0)

template <class T> int dev_alloc(esim_ctx_impl *c, T **p, size_t n)
{
    void *q = nullptr;
    hipError_t e = hipMalloc(&q, sizeof(T) * (n ? n : 1));
    if (e != hipSuccess) return fail(c, ESIM_ENOMEM, std::string("hipMalloc: ") + hipGetErrorString(e));
    c->allocs.push_back(q);
    *p = (T *)q;
    return ESIM_OK;
}

template <class T> int dev_upload(esim_ctx_impl *c, const T **p, const T *host, size_t n)
{
    T *q = nullptr;
    int rc = dev_alloc(c, &q, n);
    if (rc) return rc;
    if (n) HIP_TRY(c, hipMemcpy(q, host, sizeof(T) * n, hipMemcpyHostToDevice));
    *p = q;
    return ESIM_OK;
}

void free_device(esim_ctx_impl *c)
{
    for (void *p : c->allocs) (void)hipFree(p);
    c->allocs.clear();
    c->uploaded = false;
}

// one allocation back (buffers that are re-sized: the commuter segments, the records exchange)
void dev_free(esim_ctx_impl *c, void *p)
{
    if (!p) return;
    auto it = std::find(c->allocs.begin(), c->allocs.end(), p);
    if (it != c->allocs.end()) c->allocs.erase(it);
    (void)hipFree(p);
}

uint32_t grid_for(size_t items, uint32_t per_block, uint32_t cap)
{
    size_t g = (items + per_block - 1) / per_block;
    return (uint32_t)std::max<size_t>(1, std::min<size_t>(g, cap));
}

}  // namespace

extern "C" void esim_default_params(esim_params *p)
{
    if (!p) return;
    p->exposure_chance = 0.00055; p->mask_effectiveness = 0.70;            // disease.rs:120,127
    p->lockdown_threshold = 0.0034; p->vaccination_threshold = 0.005;      // interventions.rs:74-75
    p->mask_pt_threshold = 0.001; p->mask_everywhere_threshold = 0.0022;   // interventions.rs:55-56
    p->exposed_time = 4 * 24; p->infected_time = 14 * 24;                  // disease.rs:122-123
    p->vaccination_rate = 85 * 18;                                         // disease.rs:125
    p->bus_capacity = 20;                                                  // config.rs:37
    p->start_hour = 9; p->end_hour = 17;                                   // citizen.rs:154-155
    p->seed = 0x5EED2011ull;
    p->device = 0;
    p->max_steps = 5000;                                                   // disease.rs:124
}

// ceil(q * 2^32): `uniform < q` (citizen.rs:242) for uniform = w * 2^-32 (w a 32-bit word) is exactly
// `w < ceil(q * 2^32)`, because scaling a double by 2^32 is exact.
extern "C" int esim_threshold_lut(const esim_params *p, uint64_t out[512])
{
    if (!p || !out) return ESIM_EINVAL;
    for (int row = 0; row < 2; ++row) {
        // DiseaseModel::get_exposure_chance, disease.rs:131-154 (is_vaccinated = false: only
        // Susceptible citizens are ever tested, simulator.rs:337,436)
        double chance = p->exposure_chance - (row ? p->exposure_chance * p->mask_effectiveness : 0.0) - 0.0;
        if (std::signbit(chance)) chance = 0.0;
        for (int n = 0; n < 256; ++n) {
            const double q = 1.0 - std::pow(1.0 - chance, (double)n);      // binomial, citizen.rs:47-49
            const double scaled = std::ceil(std::ldexp(q, 32));
            out[row * 256 + n] = scaled <= 0.0 ? 0ull : (uint64_t)scaled;
        }
    }
    return ESIM_OK;
}

extern "C" int esim_create(const esim_params *p, esim_ctx **out)
{
    if (!p || !out) return fail(nullptr, ESIM_EINVAL, "esim_create: null argument");
    if (p->exposed_time + p->infected_time + 2u > TE_BIAS)
        return fail(nullptr, ESIM_ERANGE, "esim_create: exposed_time + infected_time + 2 exceeds the state encoding (512)");
    static_assert(SCH_RING >= TE_BIAS + 2u * FREE_MAX, "the school rings must hold an Infected window and two chunks");
    if (p->vaccination_rate > VACC_MAX_RATE)
        return fail(nullptr, ESIM_ERANGE, "esim_create: vaccination_rate above 8192 is not supported");
    if (p->bus_capacity == 0 || p->start_hour == 0 || p->end_hour == 0 || p->start_hour > 24 || p->end_hour > 24)
        return fail(nullptr, ESIM_EINVAL, "esim_create: bad bus_capacity / working hours");
    {
        // the schedule is evaluated once for everybody, which needs the four arms of citizen.rs:177-205 to
        // fall on four different hours
        const uint32_t h[4] = { (p->start_hour + 23u) % 24u, p->start_hour % 24u, (p->end_hour + 23u) % 24u, p->end_hour % 24u };
        for (int a = 0; a < 4; ++a) for (int b = a + 1; b < 4; ++b)
            if (h[a] == h[b]) return fail(nullptr, ESIM_EINVAL, "esim_create: start_hour-1, start_hour, end_hour-1, end_hour must be distinct");
        if (p->start_hour > 23 || p->end_hour > 23) return fail(nullptr, ESIM_EINVAL, "esim_create: working hours must be in 1..23");
    }
    if (p->max_steps == 0 || p->max_steps > ESIM_MAX_STEP)
        return fail(nullptr, ESIM_ERANGE, "esim_create: max_steps must be in 1..7600");
    int n_dev = 0;
    hipError_t e = hipGetDeviceCount(&n_dev);
    if (e != hipSuccess || n_dev <= 0)
        return fail(nullptr, ESIM_ENODEVICE, std::string("esim_create: no HIP device (") + hipGetErrorString(e) + ")");
    if (p->device < 0 || p->device >= n_dev) return fail(nullptr, ESIM_EINVAL, "esim_create: device ordinal out of range");
    e = hipSetDevice(p->device);
    if (e != hipSuccess) return fail(nullptr, ESIM_ENODEVICE, std::string("hipSetDevice: ") + hipGetErrorString(e));
    esim_ctx_impl *c = new esim_ctx_impl();
    c->P = *p;
    if (const char *e = std::getenv("ESIM_COMM_TIMEOUT_S")) { const double v = std::atof(e); if (v > 0.0) c->comm_timeout_s = v; }
    std::memset(&c->d, 0, sizeof c->d);
    e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete c; return fail(nullptr, ESIM_ENODEVICE, std::string("hipStreamCreate: ") + hipGetErrorString(e)); }
    for (auto &ev : c->ev) (void)hipEventCreate(&ev);
    *out = reinterpret_cast<esim_ctx *>(c);
    return ESIM_OK;
}

extern "C" void esim_destroy(esim_ctx *ctx)
{
    if (ctx) comm_release(CTX(ctx));
    if (!ctx) return;
    esim_ctx_impl *c = CTX(ctx);
    (void)hipSetDevice(c->P.device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    free_device(c);
    for (auto &ev : c->ev) if (ev) (void)hipEventDestroy(ev);
    for (auto &ev : c->kev) (void)hipEventDestroy(ev);
    for (auto &ev : c->sev) if (ev) (void)hipEventDestroy(ev);
    for (auto &ev : c->pkev) (void)hipEventDestroy(ev);
    for (auto &ev : c->fev) (void)hipEventDestroy(ev);
    for (auto &ev : c->cev) if (ev) (void)hipEventDestroy(ev);
    for (auto &ev : c->kdev) (void)hipEventDestroy(ev);
    if (c->pin_ctrl) (void)hipHostFree(c->pin_ctrl);
    if (c->pin_rec) (void)hipHostFree(c->pin_rec);
    if (c->stream && c->own_stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" const char *esim_last_error(const esim_ctx *ctx)
{
    if (!ctx) return g_create_error.c_str();
    return reinterpret_cast<const esim_ctx_impl *>(ctx)->err.c_str();
}

extern "C" int esim_upload_population(esim_ctx *ctx, const esim_population *pop)
{
    esim_ctx_impl *c = CTX(ctx);
    if (!c || !pop) return fail(c, ESIM_EINVAL, "esim_upload_population: null argument");
    HIP_TRY(c, hipSetDevice(c->P.device));
    const uint32_t N = pop->n_citizens, B = pop->n_buildings, R = pop->n_rooms;
    if (!pop->home_building || !pop->work_building || !pop->room || !pop->flags || !pop->building_area || !pop->building_type)
        return fail(c, ESIM_EINVAL, "esim_upload_population: a required array is NULL");
    if (R && !pop->room_building) return fail(c, ESIM_EINVAL, "esim_upload_population: room_building is NULL");
    if (pop->n_seeds && !pop->seeds) return fail(c, ESIM_EINVAL, "esim_upload_population: seeds is NULL");
    const uint32_t n_global = pop->n_citizens_global ? pop->n_citizens_global : N;
    if ((uint64_t)pop->citizen_id_base + N > n_global) return fail(c, ESIM_EINVAL, "esim_upload_population: shard range exceeds n_citizens_global");
    const bool sharded = n_global != N || pop->n_shared_buildings || pop->n_shared_rooms;
    // ---- validate the population contract and derive the static flags
    std::vector<uint8_t> fl(N);
    for (uint32_t b = 0; b < B; ++b) {
        if (pop->building_area[b] >= pop->n_areas) return fail(c, ESIM_EINVAL, "esim_upload_population: building_area out of range");
        if (pop->building_type[b] > ESIM_SCHOOL) return fail(c, ESIM_EINVAL, "esim_upload_population: unknown building_type");
    }
    for (uint32_t r = 0; r < R; ++r)
        if (pop->room_building[r] >= B || pop->building_type[pop->room_building[r]] != ESIM_SCHOOL)
            return fail(c, ESIM_EINVAL, "esim_upload_population: room_building must name a School");
    std::vector<uint32_t> room_fixed(N, 0);
    for (uint32_t i = 0; i < N; ++i) {
        const uint32_t hb = pop->home_building[i], wb = pop->work_building[i];
        if (hb >= B || wb >= B) return fail(c, ESIM_EINVAL, "esim_upload_population: building index out of range");
        if (pop->building_type[hb] == ESIM_SCHOOL) return fail(c, ESIM_EINVAL, "esim_upload_population: a School cannot be a home");
        uint8_t f = pop->flags[i] & (FL_USES_PT | FL_MASK_COMPLIANT);
        if (pop->building_area[hb] == pop->building_area[wb]) f |= FL_SAME_AREA;
        if (hb != wb) {
            f |= FL_HAS_WORK;
            if (pop->building_type[wb] == ESIM_SCHOOL) {
                f |= FL_WORK_SCHOOL;
                if (pop->room[i] >= R || pop->room_building[pop->room[i]] != wb)
                    return fail(c, ESIM_EINVAL, "esim_upload_population: school member without a room of that school");
                room_fixed[i] = pop->room[i];
            }
        }
        fl[i] = f;
    }
    for (uint32_t i = 0; i < pop->n_seeds; ++i)
        if (pop->seeds[i] >= N) return fail(c, ESIM_EINVAL, "esim_upload_population: seed index out of range");
    for (uint32_t i = 0; i < pop->n_shared_buildings; ++i)
        if (pop->shared_building_local[i] >= (int32_t)B) return fail(c, ESIM_EINVAL, "esim_upload_population: shared building out of range");
    for (uint32_t i = 0; i < pop->n_shared_rooms; ++i)
        if (pop->shared_room_local[i] >= (int32_t)R) return fail(c, ESIM_EINVAL, "esim_upload_population: shared room out of range");

    {
        // FNV-1a over what the path reads of the population (checkpoints carry it, esim_checkpoint_restore compares it)
        uint64_t h = 0xcbf29ce484222325ull;
        auto mix = [&](const void *p, size_t nbytes) { const uint8_t *q = (const uint8_t *)p; for (size_t i = 0; i < nbytes; ++i) { h ^= q[i]; h *= 0x100000001b3ull; } };
        mix(pop->home_building, sizeof(uint32_t) * (size_t)N); mix(pop->work_building, sizeof(uint32_t) * (size_t)N);
        mix(pop->room, sizeof(uint32_t) * (size_t)N); mix(pop->flags, (size_t)N);
        mix(pop->building_area, sizeof(uint32_t) * (size_t)B); mix(pop->building_type, (size_t)B);
        if (R) mix(pop->room_building, sizeof(uint32_t) * (size_t)R);
        if (pop->n_seeds) mix(pop->seeds, sizeof(uint32_t) * (size_t)pop->n_seeds);
        c->pop_hash = h;
    }
    // ---- public transport routes: riders sharing (home area, work area), simulator.rs:181-186.
    // Both travel directions group the same citizens, so one static list serves every bus step.
    std::vector<std::pair<uint64_t, uint32_t>> pairs;
    for (uint32_t i = 0; i < N; ++i)
        if (fl[i] & FL_USES_PT)
            pairs.emplace_back(((uint64_t)pop->building_area[pop->home_building[i]] << 32) | pop->building_area[pop->work_building[i]], i);
    std::sort(pairs.begin(), pairs.end());
    std::vector<uint32_t> route_off, riders(pairs.size()), route_of(N, NO_ROUTE);
    for (size_t i = 0; i < pairs.size(); ++i) {
        if (i == 0 || pairs[i].first != pairs[i - 1].first) route_off.push_back((uint32_t)i);
        riders[i] = pairs[i].second;
        route_of[pairs[i].second] = (uint32_t)route_off.size() - 1;
    }
    const uint32_t n_routes = (uint32_t)route_off.size();
    route_off.push_back((uint32_t)pairs.size());
    bool any_big = false;
    uint32_t max_route = 0;
    for (uint32_t r = 0; r < n_routes; ++r) {
        const uint32_t sz = route_off[r + 1] - route_off[r];
        max_route = std::max(max_route, sz);
        if (sz > 64) { any_big = true; for (uint32_t q = route_off[r]; q < route_off[r + 1]; ++q) fl[riders[q]] |= FL_BIG_ROUTE; }
    }
    { std::vector<std::pair<uint64_t, uint32_t>>().swap(pairs); }

    // ---- static member lists (the occupant lists the reference keeps per building:
    // output_area.rs:172-180, simulator_builder.rs:1076,1100, building.rs:404-431)
    auto csr = [](uint32_t n_keys, uint32_t n_items, auto key_of, auto use, std::vector<uint32_t> &off, std::vector<uint32_t> &idx) {
        off.assign((size_t)n_keys + 1, 0);
        for (uint32_t i = 0; i < n_items; ++i) if (use(i)) off[key_of(i) + 1]++;
        for (uint32_t k = 0; k < n_keys; ++k) off[k + 1] += off[k];
        idx.resize(off[n_keys]);
        std::vector<uint32_t> cur(off.begin(), off.end() - 1);
        for (uint32_t i = 0; i < n_items; ++i) if (use(i)) idx[cur[key_of(i)]++] = i;
    };
    std::vector<uint32_t> res_off, res_idx, wrk_off, wrk_idx, room_off, room_idx;
    csr(B, N, [&](uint32_t i) { return pop->home_building[i]; }, [&](uint32_t) { return true; }, res_off, res_idx);
    bool home_sorted = true;
    for (uint32_t i = 0; i < N && home_sorted; ++i) home_sorted = res_idx[i] == i;
    csr(B, N, [&](uint32_t i) { return pop->work_building[i]; },
        [&](uint32_t i) { return (fl[i] & FL_HAS_WORK) && !(fl[i] & FL_WORK_SCHOOL); }, wrk_off, wrk_idx);
    csr(R, N, [&](uint32_t i) { return room_fixed[i]; }, [&](uint32_t i) { return (fl[i] & FL_WORK_SCHOOL) != 0; }, room_off, room_idx);

    // ---- initial state: everyone Susceptible at home (citizen.rs:139-162), seeds Infected(0)
    c->init_state.resize(N);
    for (uint32_t i = 0; i < N; ++i) c->init_state[i] = CW_MAKE(TE_SUSCEPTIBLE, (uint32_t)fl[i]);
    c->init_log.clear();
    const uint32_t seed_te = TE_BIAS - (c->P.exposed_time + 1u);                        // Infected(0) before step 1
    for (uint32_t i = 0; i < pop->n_seeds; ++i) {
        const uint32_t sc = pop->seeds[i];
        if (CW_TE(c->init_state[sc]) != seed_te) { c->init_state[sc] = CW_MAKE(seed_te, (uint32_t)fl[sc]); c->init_log.push_back(sc); }
    }

    // a communicator belongs to the population it was set up for (its buffers are sized and its ranks checked against the
    // shard): a new upload invalidates it -- call esim_comm_init_* again afterwards
    comm_release(c);
    c->comm_fn = nullptr; c->comm_user = nullptr; c->comm_rank = 0; c->comm_world = 1;
    c->xr = nullptr; c->xr_n = 0;
    free_device(c);
    Dev &d = c->d;
    std::memset(&d, 0, sizeof d);
    d.n = N; d.n_global = n_global; d.id_base = pop->citizen_id_base; d.n_bld = B; d.n_room = R;
    d.n_pt = (uint32_t)riders.size(); d.n_routes = n_routes; c->n_routes = n_routes;
    d.max_route = max_route;
    int rc;
    if ((rc = dev_alloc(c, &d.cit, (size_t)N + 1))) return rc;
    if ((rc = dev_upload(c, &d.home, pop->home_building, N))) return rc;
    if ((rc = dev_upload(c, &d.work, pop->work_building, N))) return rc;
    if ((rc = dev_upload(c, &d.room, room_fixed.data(), N))) return rc;
    if ((rc = dev_upload(c, &d.res_off, res_off.data(), res_off.size()))) return rc;
    if (!home_sorted) { if ((rc = dev_upload(c, &d.res_idx, res_idx.data(), res_idx.size()))) return rc; }
    else d.res_idx = nullptr;
    if ((rc = dev_upload(c, &d.wrk_off, wrk_off.data(), wrk_off.size()))) return rc;
    if ((rc = dev_upload(c, &d.wrk_idx, wrk_idx.data(), wrk_idx.size()))) return rc;
    if ((rc = dev_upload(c, &d.room_off, room_off.data(), room_off.size()))) return rc;
    if ((rc = dev_upload(c, &d.room_idx, room_idx.data(), room_idx.size()))) return rc;
    if ((rc = dev_upload(c, &d.room_bld, pop->room_building, R))) return rc;
    if ((rc = dev_upload(c, &d.bld_type, pop->building_type, B))) return rc;
    uint32_t *cnt = nullptr;
    const size_t per_parity = (size_t)B + R + n_routes;
    if ((rc = dev_alloc(c, &cnt, MARK_SLOTS * per_parity))) return rc;
    c->cnt_base = cnt;
    c->cnt_bytes = sizeof(uint32_t) * MARK_SLOTS * per_parity;
    if ((rc = dev_alloc(c, &d.exp_step, 2 * ((size_t)c->P.max_steps + 2)))) return rc;
    if ((rc = dev_alloc(c, &d.exp_part, (size_t)EXP_ROWS * 2u * FREE_MAX))) return rc;
    HIP_TRY(c, hipMemset(d.exp_part, 0, sizeof(uint32_t) * EXP_ROWS * 2u * FREE_MAX));
    if ((rc = dev_alloc(c, &d.dec, FREE_MAX + 1))) return rc;
    // hash map of the time-parallel chunks: room for ~2 marks per (Infected, step) pair at < 1/2 load
    {
        // slots for half the citizens (a quarter of that many items: a chunk in which up to ~3 % of the citizens are Infected
        // still runs in the one-pass form), between 2^20 and 2^26; ~440 B per slot, most of it spill counters that stay cold
        uint32_t cap = 1u << 20;
        while (cap < (1u << 26) && cap < N / 2u) cap <<= 1;
        if (const char *e = std::getenv("ESIM_HASH_LOG2")) cap = 1u << std::min(28, std::max(4, std::atoi(e)));
        d.hcap = cap;
        d.items_cap = cap / 4u;                     // load factor <= 1/4; one count vector of FREE_MAX steps per item
        // (every chunk table is initialised at allocation: no kernel ever reads memory nobody wrote, whatever a diagnostics
        // build leaves out -- and the consumers check what they read from these tables against the capacities)
        if ((rc = dev_alloc(c, &d.hkey, cap))) return rc;
        if ((rc = dev_alloc(c, &d.hitems, d.items_cap))) return rc;
        if ((rc = dev_alloc(c, &d.item_rec, d.items_cap))) return rc;
        HIP_TRY(c, hipMemset(d.hitems, 0xFF, sizeof(uint32_t) * (size_t)d.items_cap));        // ITEM_UNUSED
        HIP_TRY(c, hipMemset(d.item_rec, 0, sizeof(ItemRec) * (size_t)d.items_cap));
        if ((rc = dev_alloc(c, &d.vec, (size_t)cap * FREE_MAX))) return rc;
        if ((rc = dev_alloc(c, &d.slot_state, cap))) return rc;
        if ((rc = dev_alloc(c, &d.slot_iv, (size_t)cap * SLOT_IV_STRIDE))) return rc;
        HIP_TRY(c, hipMemset(d.slot_state, 0, sizeof(uint32_t) * cap));
        HIP_TRY(c, hipMemset(d.slot_iv, 0, sizeof(uint32_t) * (size_t)cap * SLOT_IV_STRIDE));
        // deferred units: SUBQ queues; a queue that is full makes its producer draw the list itself, so the size is a
        // matter of speed only.  Room for the smaller of: every long member list marked in every step; a quarter of the
        // citizens -- twice over, since the queues fill unevenly.
        size_t units = 0;
        auto add_lists = [&](const std::vector<uint32_t> &off) {
            for (size_t i = 0; i + 1 < off.size(); ++i) {
                const size_t pairs = (size_t)(off[i + 1] - off[i]) * FREE_MAX;
                if (pairs > UNIT_INLINE) units += (pairs + UNIT_PAIRS - 1) / UNIT_PAIRS;
            }
        };
        add_lists(res_off); add_lists(wrk_off); add_lists(room_off);
        units = std::min<size_t>(units, (size_t)N / 8u + 65536u);
        d.unit_qcap = (uint32_t)std::max<size_t>(1024, units * 2u / SUBQ);
        if ((rc = dev_alloc(c, &d.units, (size_t)d.unit_qcap * SUBQ))) return rc;
        if ((rc = dev_alloc(c, &d.route_pairs, (size_t)d.items_cap * (CHUNK_BUS_STEPS / 4u)))) return rc;   // (PAIR_K: up to CHUNK_BUS_STEPS / 4 per item id)
        if ((rc = dev_alloc(c, &d.route_pairs_big, (size_t)d.items_cap * 2u))) return rc;
        HIP_TRY(c, hipMemset(d.units, 0xFF, sizeof(UnitRec) * (size_t)d.unit_qcap * SUBQ));     // code == UNIT_NOOP
        HIP_TRY(c, hipMemset(d.route_pairs, 0, sizeof(uint32_t) * (size_t)d.items_cap * (CHUNK_BUS_STEPS / 4u)));
        HIP_TRY(c, hipMemset(d.route_pairs_big, 0, sizeof(uint32_t) * (size_t)d.items_cap * 2u));
        {
            // (a school building's records are those of everybody who works or learns there: its members are in the room lists)
            std::vector<uint32_t> sch_members((size_t)B + 1, 0);
            for (uint32_t i = 0; i < N; ++i) if (fl[i] & FL_WORK_SCHOOL) sch_members[pop->work_building[i] + 1]++;
            for (uint32_t b = 0; b < B; ++b) sch_members[b + 1] += sch_members[b];
            std::vector<uint32_t> ovf_off(res_off.size());
            for (size_t i = 0; i < res_off.size(); ++i) ovf_off[i] = res_off[i] + wrk_off[i] + sch_members[i];
            if ((rc = dev_upload(c, &d.ovf_off, ovf_off.data(), ovf_off.size()))) return rc;
            {
                std::vector<int32_t> sch_of(B ? B : 1, -1);
                uint32_t n_sch = 0;
                for (uint32_t b = 0; b < B; ++b) if (pop->building_type[b] == ESIM_SCHOOL) sch_of[b] = (int32_t)n_sch++;
                d.n_sch = n_sch;
                if ((rc = dev_upload(c, &d.sch_of_bld, sch_of.data(), sch_of.size()))) return rc;
                if ((rc = dev_alloc(c, &d.sch_ring, (size_t)(n_sch ? n_sch : 1) * 2u * SCH_RING))) return rc;
                HIP_TRY(c, hipMemset(d.sch_ring, 0, sizeof(uint32_t) * (size_t)(n_sch ? n_sch : 1) * 2u * SCH_RING));
            }
            {
                // the records k_chunk_marks reads with one request each (Dev::where4, Dev::bld8), and the schools' difference arrays
                std::vector<uint4> w4(N ? N : 1);
                for (uint32_t i = 0; i < N; ++i) w4[i] = make_uint4(pop->home_building[i], pop->work_building[i], room_fixed[i], route_of[i]);
                if ((rc = dev_upload(c, &d.where4, w4.data(), w4.size()))) return rc;
                std::vector<BldRec> b8(B ? B : 1);
                for (uint32_t b = 0; b < B; ++b) b8[b] = BldRec{ res_off[b], res_off[b + 1], wrk_off[b], wrk_off[b + 1], (uint32_t)pop->building_type[b], ovf_off[b], ovf_off[b + 1], 0u };
                if ((rc = dev_upload(c, &d.bld8, b8.data(), b8.size()))) return rc;
                const size_t sd = (size_t)(d.n_sch ? d.n_sch : 1) * SD_REPL * 2u * FREE_MAX;
                if ((rc = dev_alloc(c, &d.sch_diff, sd))) return rc;
                HIP_TRY(c, hipMemset(d.sch_diff, 0, sizeof(uint32_t) * sd));
            }
            d.ovf_room_base = ovf_off.back();
            // (the persistent map keeps two places per member -- a record and a cancellation -- and the routes' riders behind the rooms)
            d.ovf_route_base = d.ovf_room_base + room_off.back();
            d.ovf_n = 2u * (d.ovf_route_base + (uint32_t)riders.size()) + 2u;
            if ((rc = dev_alloc(c, &d.ovf, (size_t)d.ovf_n))) return rc;
            HIP_TRY(c, hipMemset(d.ovf, 0, sizeof(uint32_t) * (size_t)d.ovf_n));
        }
        d.n_wrk_idx = (uint32_t)wrk_idx.size(); d.n_room_idx = (uint32_t)room_idx.size();
        d.big_qcap = d.items_cap / SUBQ;             // (a slot is listed at most once a chunk, and there are at most items_cap of them)
        if ((rc = dev_alloc(c, &d.big_list, (size_t)d.big_qcap * SUBQ * 3u))) return rc;
        if ((rc = dev_alloc(c, &d.used_pref, CHUNK_WAVES_MAX + 1u))) return rc;
        if ((rc = dev_alloc(c, &d.pbig_cnt, SUBQ))) return rc;
        if ((rc = dev_alloc(c, &d.neg_list, (size_t)NEG_CAP * 2u))) return rc;
        if ((rc = dev_alloc(c, &d.cancel_list, (size_t)NEG_CAP * 2u))) return rc;
        HIP_TRY(c, hipMemset(d.cancel_list, 0, sizeof(uint32_t) * (size_t)NEG_CAP * 2u));
        HIP_TRY(c, hipMemset(d.pbig_cnt, 0, sizeof(uint32_t) * SUBQ));
        HIP_TRY(c, hipMemset(d.neg_list, 0, sizeof(uint32_t) * (size_t)NEG_CAP * 2u));
        HIP_TRY(c, hipMemset(d.big_list, 0, sizeof(uint32_t) * (size_t)d.big_qcap * SUBQ * 3u));
        HIP_TRY(c, hipMemset(d.used_pref, 0, sizeof(uint32_t) * (CHUNK_WAVES_MAX + 1u)));
        if ((rc = dev_alloc(c, &d.pair_cnt, 16384u))) return rc;
        if ((rc = dev_alloc(c, &d.used_cnt, 16384u))) return rc;
        HIP_TRY(c, hipMemset(d.used_cnt, 0, sizeof(uint32_t) * 16384u));
        if ((rc = dev_alloc(c, &d.hot, (size_t)HOT_COUNT * HOT_STRIDE))) return rc;
        HIP_TRY(c, hipMemset(d.hot, 0, sizeof(uint32_t) * HOT_COUNT * HOT_STRIDE));
        HIP_TRY(c, hipMemset(d.pair_cnt, 0, sizeof(uint32_t) * 16384u));
        d.newexp_cap = N / SUBQ + 1u;                 // citizens with the same id & 63: nobody is listed twice in a chunk
        if ((rc = dev_alloc(c, &d.newexp, (size_t)d.newexp_cap * SUBQ))) return rc;
        HIP_TRY(c, hipMemset(d.newexp, 0, sizeof(uint32_t) * (size_t)d.newexp_cap * SUBQ));
        if ((rc = dev_alloc(c, &d.cursor, (size_t)EXP_ROWS * FREE_MAX))) return rc;
        HIP_TRY(c, hipMemset(d.hkey, 0xFF, sizeof(unsigned long long) * cap));
        HIP_TRY(c, hipMemset(d.vec, 0, sizeof(uint32_t) * (size_t)cap * FREE_MAX));
        HIP_TRY(c, hipMemset(d.cursor, 0, sizeof(uint32_t) * EXP_ROWS * FREE_MAX));
    }
    for (int p = 0; p < (int)MARK_SLOTS; ++p) {
        uint32_t *base = cnt + p * per_parity;
        d.cnt_bld[p] = base; d.cnt_room[p] = base + B; d.route_flag[p] = base + B + R;
        if ((rc = dev_alloc(c, &d.touched_bld[p], B))) return rc;
        if ((rc = dev_alloc(c, &d.touched_room[p], R))) return rc;
        if ((rc = dev_alloc(c, &d.touched_route[p], n_routes))) return rc;
        if ((rc = dev_alloc(c, &d.touched_route_big[p], n_routes))) return rc;
    }
    if ((rc = dev_alloc(c, &d.vax_ev, (size_t)FREE_MAX * VACC_MAX_RATE))) return rc;
    if ((rc = dev_alloc(c, &d.vax_cnt, FREE_MAX))) return rc;
    if ((rc = dev_alloc(c, &d.vax_now, FREE_MAX))) return rc;
    if ((rc = dev_alloc(c, &d.vax_delta, 4u * (FREE_MAX + 2u)))) return rc;
    if ((rc = dev_alloc(c, &d.lost_list, LOST_CAP))) return rc;
    HIP_TRY(c, hipMemset(d.lost_list, 0, sizeof(uint32_t) * LOST_CAP));
    if ((rc = dev_alloc(c, &d.xf_adj, FREE_MAX + 2u))) return rc;
    HIP_TRY(c, hipMemset(d.vax_cnt, 0, sizeof(uint32_t) * FREE_MAX));
    HIP_TRY(c, hipMemset(d.vax_now, 0, sizeof(uint32_t) * FREE_MAX));
    HIP_TRY(c, hipMemset(d.vax_delta, 0, sizeof(uint32_t) * 4u * (FREE_MAX + 2u)));
    HIP_TRY(c, hipMemset(d.xf_adj, 0, sizeof(uint32_t) * (FREE_MAX + 2u)));
    if ((rc = dev_alloc(c, &d.hist, TE_SLOTS))) return rc;
    if ((rc = dev_alloc(c, &d.log, (size_t)N + 1))) return rc;
    if ((rc = dev_alloc(c, &d.log_off, TE_SLOTS + 1))) return rc;
    uint64_t lut[512];
    esim_threshold_lut(&c->P, lut);
    if ((rc = dev_upload(c, &d.thr, lut, 512))) return rc;
    if ((rc = dev_alloc(c, &d.ctrl, 1))) return rc;
    if ((rc = dev_alloc(c, &d.records, (size_t)c->P.max_steps + 1))) return rc;
    if ((rc = dev_upload(c, &d.route_off, route_off.data(), route_off.size()))) return rc;
    if ((rc = dev_upload(c, &d.route_riders, riders.data(), riders.size()))) return rc;
    if ((rc = dev_upload(c, &d.route_of, route_of.data(), N))) return rc;
    const size_t big_scratch = any_big ? riders.size() : 0;
    if ((rc = dev_alloc(c, &d.bus_key, big_scratch))) return rc;
    if ((rc = dev_alloc(c, &d.bus_idx, big_scratch))) return rc;
    if ((rc = dev_alloc(c, &d.bus_cnt, big_scratch))) return rc;
    if ((rc = dev_alloc(c, &d.bus_flag, big_scratch))) return rc;
    d.exposed_time = c->P.exposed_time; d.infected_time = c->P.infected_time;
    d.vaccination_rate = c->P.vaccination_rate; d.bus_capacity = c->P.bus_capacity;
    d.start_hour = c->P.start_hour; d.end_hour = c->P.end_hour;
    d.seed_lo = (uint32_t)c->P.seed; d.seed_hi = (uint32_t)(c->P.seed >> 32);
    d.thr_lockdown = c->P.lockdown_threshold; d.thr_vacc = c->P.vaccination_threshold;
    d.thr_mask_pt = c->P.mask_pt_threshold; d.thr_mask_all = c->P.mask_everywhere_threshold;
    d.max_steps = c->P.max_steps;
    d.n_shards = sharded ? 2u : 1u;
    d.n_shared_bld = pop->n_shared_buildings; d.n_shared_room = pop->n_shared_rooms;
    if ((rc = dev_upload(c, &d.shared_bld, pop->shared_building_local, pop->n_shared_buildings))) return rc;
    if ((rc = dev_upload(c, &d.shared_room, pop->shared_room_local, pop->n_shared_rooms))) return rc;
    {
        // the inverse of the shared tables: which shared slot a local building / room is (sharded chunks, k_shared_pack)
        std::vector<int32_t> of_b(B ? B : 1, -1), of_r(R ? R : 1, -1);
        for (uint32_t i = 0; i < pop->n_shared_buildings; ++i) if (pop->shared_building_local[i] >= 0) of_b[pop->shared_building_local[i]] = (int32_t)i;
        for (uint32_t i = 0; i < pop->n_shared_rooms; ++i) if (pop->shared_room_local[i] >= 0) of_r[pop->shared_room_local[i]] = (int32_t)i;
        if ((rc = dev_upload(c, &d.shared_of_bld, of_b.data(), of_b.size()))) return rc;
        if ((rc = dev_upload(c, &d.shared_of_room, of_r.data(), of_r.size()))) return rc;
        if ((rc = dev_alloc(c, &d.xv, XV_HEADER + (size_t)FREE_MAX * (PLAN_W / 32u)))) return rc;
        if ((rc = dev_alloc(c, &d.xc, FREE_MAX + 2u))) return rc;
        if ((rc = dev_alloc(c, &d.xl, FREE_MAX + 2u))) return rc;
        HIP_TRY(c, hipMemset(d.xl, 0, sizeof(uint32_t) * (FREE_MAX + 2u)));
        if ((rc = dev_alloc(c, &d.xe, XE_WORDS))) return rc;
        HIP_TRY(c, hipMemset(d.xe, 0, sizeof(uint32_t) * XE_WORDS));
        HIP_TRY(c, hipMemset(d.xv, 0, sizeof(uint32_t) * (XV_HEADER + (size_t)FREE_MAX * (PLAN_W / 32u))));
        HIP_TRY(c, hipMemset(d.xc, 0, sizeof(uint32_t) * (FREE_MAX + 2u)));
        d.rank = 0; d.world = 1; d.xs = nullptr;
    }
#ifdef ESIM_COUNT_WORK
    if ((rc = dev_alloc(c, &d.work_cnt, (size_t)WK_N))) return rc;
    HIP_TRY(c, hipMemset(d.work_cnt, 0, sizeof(unsigned long long) * WK_N));
#endif
#ifdef ESIM_WAVE_PROFILE
    if ((rc = dev_alloc(c, &d.prof_buf, (size_t)16384 * 16))) return rc;
    HIP_TRY(c, hipMemset(d.prof_buf, 0, sizeof(uint32_t) * 16384 * 16));
#endif
    c->xa_n = XA_HEADER + (size_t)d.n_shared_bld + d.n_shared_room;
    c->xb_n = XB_HEADER + VACC_WINDOW / 32u;
    if ((rc = dev_alloc(c, &d.xa, c->xa_n))) return rc;
    if ((rc = dev_alloc(c, &d.xb, c->xb_n))) return rc;
    c->xf_n = std::min<uint32_t>(FREE_MAX, c->P.exposed_time + 1u);
    d.xf_n = (uint32_t)c->xf_n;
    if ((rc = dev_alloc(c, &d.xf, FREE_MAX + 1))) return rc;
    HIP_TRY(c, hipMemset(d.xf, 0, sizeof(uint32_t) * (FREE_MAX + 1)));
    HIP_TRY(c, hipMemset(d.xa, 0, sizeof(uint32_t) * c->xa_n));
    HIP_TRY(c, hipMemset(d.xb, 0, sizeof(uint32_t) * c->xb_n));

    if (!c->pin_ctrl) HIP_TRY(c, hipHostMalloc((void **)&c->pin_ctrl, sizeof(Ctrl), hipHostMallocDefault));
    if (c->pin_rec_n < (size_t)c->P.max_steps + 1) {
        if (c->pin_rec) (void)hipHostFree(c->pin_rec);
        c->pin_rec = nullptr; c->pin_rec_n = 0;
        HIP_TRY(c, hipHostMalloc((void **)&c->pin_rec, sizeof(esim_step_result) * ((size_t)c->P.max_steps + 1), hipHostMallocDefault));
        c->pin_rec_n = (size_t)c->P.max_steps + 1;
    }
    c->grid_citizens = grid_for(N, TPB, 2048);
    c->grid_infected = 1024;
    c->grid_expose = 1024;
    if (const char *e = std::getenv("ESIM_GRID_INFECTED")) c->grid_infected = (uint32_t)std::max(1, std::atoi(e));   // tuning knobs
    if (const char *e = std::getenv("ESIM_GRID_CHUNK")) c->grid_chunk = (uint32_t)std::min((int)(CHUNK_WAVES_MAX * 64u / TPB), std::max(16, std::atoi(e) / 16 * 16));   // whole groups of 64 wavefronts
    if (std::getenv("ESIM_TRACE_HOST")) c->host_trace = true;
    if (const char *e = std::getenv("ESIM_VAX_REPAIR")) { c->vax_repair = std::atoi(e) != 0; c->vax_repair_always = std::atoi(e) >= 2; }
    if (const char *e = std::getenv("ESIM_TINY_PAIRS")) c->tiny_pairs = (uint32_t)std::max(0, std::atoi(e));
    if (const char *e = std::getenv("ESIM_SMALL_GRID")) c->small_grid = (uint32_t)std::max(0, std::atoi(e) / 16 * 16);
    if (const char *e = std::getenv("ESIM_SMALL_MULT")) c->small_mult = (uint32_t)std::min(16, std::max(1, std::atoi(e)));
    if (const char *e = std::getenv("ESIM_PMAP")) c->pmap = std::atoi(e) != 0;
    if (const char *e = std::getenv("ESIM_PMAP_REBUILD")) c->pmap_rebuild_every = (uint32_t)std::max(1, std::atoi(e));
    if (const char *e = std::getenv("ESIM_DRAW_MULT")) c->draw_mult = (uint32_t)std::min(4, std::max(1, std::atoi(e)));      // (16 384 wavefronts at most: Dev::pair_cnt)
    if (const char *e = std::getenv("ESIM_UNITS_MULT")) c->units_mult = (uint32_t)std::min(16, std::max(1, std::atoi(e)));
    if (const char *e = std::getenv("ESIM_GRID_EXPOSE")) c->grid_expose = (uint32_t)std::max(1, std::atoi(e));
    c->uploaded = true;
    return esim_reset(ctx);
}

extern "C" int esim_reset(esim_ctx *ctx)
{
    esim_ctx_impl *c = CTX(ctx);
    if (!c || !c->uploaded) return fail(c, ESIM_ESTATE, "esim_reset: no population uploaded");
    HIP_TRY(c, hipSetDevice(c->P.device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    const Dev &d = c->d;
    const uint32_t n_seeds = (uint32_t)c->init_log.size();
    Ctrl h;
    std::memset(&h, 0, sizeof h);
    h.t = 1;
    h.mask = ESIM_MASK_NONE;
    h.n_susceptible = d.n - n_seeds;
    h.log_len = n_seeds;
    HIP_TRY(c, hipMemcpy(d.ctrl, &h, sizeof h, hipMemcpyHostToDevice));
    if (d.n) HIP_TRY(c, hipMemcpy(d.cit, c->init_state.data(), sizeof(uint32_t) * (size_t)d.n, hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemset(c->cnt_base, 0, c->cnt_bytes));
    HIP_TRY(c, hipMemset(d.exp_step, 0, sizeof(uint32_t) * 2 * ((size_t)c->P.max_steps + 2)));
    HIP_TRY(c, hipMemset(d.records, 0, sizeof(esim_step_result) * ((size_t)c->P.max_steps + 1)));
    // census histogram and exposure log: the seeds are Infected(0) before step 1, i.e. "exposed" at
    // step -(exposed_time + 1)
    const uint32_t seed_te = TE_BIAS - (c->P.exposed_time + 1u);
    std::vector<uint32_t> hist(TE_SLOTS, 0), off(TE_SLOTS + 1, 0);
    hist[seed_te] = n_seeds;
    for (uint32_t k = seed_te + 1; k <= TE_SLOTS; ++k) off[k] = n_seeds;
    HIP_TRY(c, hipMemcpy(d.hist, hist.data(), sizeof(uint32_t) * TE_SLOTS, hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(d.log_off, off.data(), sizeof(uint32_t) * (TE_SLOTS + 1), hipMemcpyHostToDevice));
    if (n_seeds) HIP_TRY(c, hipMemcpy(d.log, c->init_log.data(), sizeof(uint32_t) * n_seeds, hipMemcpyHostToDevice));
    c->host_t = 1;
    c->map_valid = false;                         // (the device arrays may still hold the last run's map: the first chunk clears them)
    c->stop_flag_dev = 0;
    c->pin_track = false;
    c->free_limit = 0;
    c->last_chunk_pairs = (uint32_t)c->init_log.size();
    c->phase_s[0] = c->phase_s[1] = c->phase_s[2] = 0;
    c->kev_used = 0;
    c->small_ms = 0; c->small_steps = 0;
    c->pkev_used = 0; c->pipe_steps = 0;
    c->chunk_ms = 0; c->chunk_steps = 0; c->chunk_count = 0;
    c->vax_chunk_steps = 0; c->vax_chunk_cuts = 0; c->elig_seen = false; c->repair_armed = false; c->quiet = false;
    return ESIM_OK;
}

namespace {

int enqueue_begin(esim_ctx_impl *c, bool time_kernel)
{
    Dev &d = c->d;
    if (c->phase_timing) HIP_TRY(c, hipEventRecord(c->ev[0], c->stream));
    if (time_kernel) HIP_TRY(c, hipEventRecord(c->kev[c->kev_used + 0], c->stream));
    hipLaunchKernelGGL(k_infected, dim3(c->grid_infected), dim3(TPB), 0, c->stream, d);
    if (d.n_shards > 1) {
        const uint32_t n = (uint32_t)std::max<size_t>(XA_HEADER, std::max(d.n_shared_bld, d.n_shared_room));
        hipLaunchKernelGGL(k_pack_a, dim3(grid_for(n, TPB, 1u << 20)), dim3(TPB), 0, c->stream, d);
    }
    if (c->phase_timing) HIP_TRY(c, hipEventRecord(c->ev[1], c->stream));
    return ESIM_OK;
}

int enqueue_exposures(esim_ctx_impl *c, bool time_kernel)
{
    Dev &d = c->d;
    if (d.n_shards > 1) {
        const uint32_t n = (uint32_t)std::max<size_t>(XA_HEADER, std::max(d.n_shared_bld, d.n_shared_room));
        hipLaunchKernelGGL(k_unpack_a, dim3(grid_for(n, TPB, 1u << 20)), dim3(TPB), 0, c->stream, d);
    }
    (void)time_kernel;
    hipLaunchKernelGGL(k_expose, dim3(c->grid_expose), dim3(TPB), 0, c->stream, d);
    if (d.n_shards > 1) hipLaunchKernelGGL(k_pack_b, dim3(VACC_WINDOW / TPB), dim3(TPB), 0, c->stream, d);
    if (c->phase_timing) HIP_TRY(c, hipEventRecord(c->ev[2], c->stream));
    return ESIM_OK;
}

int enqueue_finish(esim_ctx_impl *c, bool time_kernel, int mode = -1)
{
    Dev &d = c->d;
    if (mode < 0) mode = d.n_shards > 1 ? 1 : 0;
    hipLaunchKernelGGL(k_finish, dim3(1), dim3(FIN_TPB), 0, c->stream, d, mode);
    if (time_kernel) { HIP_TRY(c, hipEventRecord(c->kev[c->kev_used + 1], c->stream)); c->kev_used += 2; }
    if (c->phase_timing) {
        HIP_TRY(c, hipEventRecord(c->ev[3], c->stream));
        HIP_TRY(c, hipEventSynchronize(c->ev[3]));
        float ms;
        for (int i = 0; i < 3; ++i) { HIP_TRY(c, hipEventElapsedTime(&ms, c->ev[i], c->ev[i + 1])); c->phase_s[i] += ms * 1e-3; }
    }
    c->host_t++;
    HIP_TRY(c, hipGetLastError());
    return ESIM_OK;
}

int check_budget(esim_ctx_impl *c, uint32_t n_steps)
{
    if (!c || !c->uploaded) return fail(c, ESIM_ESTATE, "no population uploaded");
    if ((uint64_t)c->host_t + n_steps - 1 > c->P.max_steps)
        return fail(c, ESIM_ERANGE, "step budget exhausted: max_steps reached (DiseaseModel::max_time_step)");
    return ESIM_OK;
}

bool want_kernel_timing(esim_ctx_impl *c)
{
    if (!c->kernel_timing || (c->host_t % c->kernel_timing_stride) != 0) return false;
    if (c->kev_used + 2 > c->kev.size()) {
        for (int i = 0; i < 2; ++i) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return false; c->kev.push_back(e); }
    }
    return true;
}

int fail_dev(esim_ctx_impl *c, uint32_t err)
{
    return fail(c, -(int)err, "device-side error (S underflow / vaccination window exhausted / a chunk table overflowed)");
}

static inline void ht_mark(esim_ctx_impl *c, const char *what)
{
    if (!c->host_trace) return;
    timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
    c->ht.emplace_back(what, ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3);
}

// The control block through the pinned mirror: an asynchronous copy and one wait.
int read_ctrl(esim_ctx_impl *c, Ctrl *h)
{
    HIP_TRY(c, hipMemcpyAsync(c->pin_ctrl, c->d.ctrl, sizeof(Ctrl), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    *h = *c->pin_ctrl;
    return ESIM_OK;
}

void kd_resolve(esim_ctx_impl *c);

// After a burst of chunk passes that started at step `first` and can have advanced `span` steps at most: the control block and
// the records of those steps come back with one wait (esim_run hands the records on from the mirror).
int burst_readback(esim_ctx_impl *c, uint32_t first, uint32_t span, Ctrl *h)
{
    const Dev &d = c->d;
    ht_mark(c, "kernels enqueued");
    HIP_TRY(c, hipMemcpyAsync(c->pin_ctrl, d.ctrl, sizeof(Ctrl), hipMemcpyDeviceToHost, c->stream));
    const bool rec = c->pin_track && first == c->pin_first + c->pin_valid && (size_t)first + span <= c->pin_rec_n;
    if (rec) HIP_TRY(c, hipMemcpyAsync(c->pin_rec + first, d.records + first, sizeof(esim_step_result) * span, hipMemcpyDeviceToHost, c->stream));
    ht_mark(c, "copies enqueued");
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    ht_mark(c, "stream drained");
    *h = *c->pin_ctrl;
    if (rec && h->t >= first) c->pin_valid += std::min<uint32_t>(h->t - first, span);
    c->ctrl_fresh = true;
    if (c->kdetail) kd_resolve(c);
    return ESIM_OK;
}

int device_error(esim_ctx_impl *c)
{
    Ctrl h;
    HIP_TRY(c, hipMemcpy(&h, c->d.ctrl, sizeof h, hipMemcpyDeviceToHost));
    if (h.error) return fail(c, -(int)h.error, "device-side error (S underflow / vaccination window exhausted / a chunk table check: " + std::to_string(h.err_where) + ")");
    return ESIM_OK;
}

}  // namespace

extern "C" int esim_step_begin(esim_ctx *ctx)
{
    esim_ctx_impl *c = CTX(ctx);
    int rc = check_budget(c, 1);
    if (rc) return rc;
    HIP_TRY(c, hipSetDevice(c->P.device));
    c->timing_this_step = want_kernel_timing(c);
    return enqueue_begin(c, c->timing_this_step);
}

extern "C" int esim_step_exposures(esim_ctx *ctx)
{
    esim_ctx_impl *c = CTX(ctx);
    if (!c || !c->uploaded) return fail(c, ESIM_ESTATE, "no population uploaded");
    HIP_TRY(c, hipSetDevice(c->P.device));
    return enqueue_exposures(c, c->timing_this_step);
}

extern "C" int esim_step_finish(esim_ctx *ctx, esim_step_result *out)
{
    esim_ctx_impl *c = CTX(ctx);
    if (!c || !c->uploaded) return fail(c, ESIM_ESTATE, "no population uploaded");
    HIP_TRY(c, hipSetDevice(c->P.device));
    int rc = enqueue_finish(c, c->timing_this_step);
    c->timing_this_step = false;
    if (rc) return rc;
    if (out) {
        HIP_TRY(c, hipMemcpyAsync(out, &c->d.records[c->host_t - 1], sizeof *out, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        return device_error(c);
    }
    return ESIM_OK;
}

extern "C" int esim_step(esim_ctx *ctx, esim_step_result *out);

namespace {

// Sequential steps (three kernels per step, or the persistent single-workgroup kernel while few citizens
// are Infected): the only form that can vaccinate.
int run_sequential(esim_ctx_impl *c, uint32_t n_steps, bool allow_early_stop, uint32_t *executed)
{
    Dev &d = c->d;
    c->ctrl_fresh = false;                   // (whatever a burst read back is out of date once more steps are enqueued)
    c->quiet = false;
    c->map_valid = false;                    // (sequential steps do not maintain the persistent item map)
    uint32_t remaining = n_steps, total = 0;
    int rc;
    while (remaining > 0) {
        if (c->small_max > 0 && !c->phase_timing) {
            if (c->kernel_timing) { if (!c->sev[0]) { (void)hipEventCreate(&c->sev[0]); (void)hipEventCreate(&c->sev[1]); } HIP_TRY(c, hipEventRecord(c->sev[0], c->stream)); }
            hipLaunchKernelGGL(k_small, dim3(1), dim3(FIN_TPB), 0, c->stream, d, remaining, c->small_max, 0);
            if (c->kernel_timing) HIP_TRY(c, hipEventRecord(c->sev[1], c->stream));
            Ctrl h;
            HIP_TRY(c, hipMemcpyAsync(&h, d.ctrl, sizeof h, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            if (c->kernel_timing && h.small_done) { float ms; HIP_TRY(c, hipEventElapsedTime(&ms, c->sev[0], c->sev[1])); c->small_ms += ms; c->small_steps += h.small_done; }
            c->host_t += h.small_done; total += h.small_done; remaining -= h.small_done;
            if (h.error) return fail(c, -(int)h.error, "device-side error (S underflow / vaccination window exhausted / a chunk table check: " + std::to_string(h.err_where) + ")");
            if (h.finished && allow_early_stop) break;
        }
        if (remaining == 0) break;
        const uint32_t chunk = std::min<uint32_t>(remaining, (c->small_max > 0 && !c->phase_timing) ? 32u : remaining);
        for (uint32_t s = 0; s < chunk; ++s) {
            const bool tk = want_kernel_timing(c);
            if ((rc = enqueue_begin(c, tk))) return rc;
            if ((rc = enqueue_exposures(c, tk))) return rc;
            if ((rc = enqueue_finish(c, tk, 0))) return rc;
        }
        total += chunk; remaining -= chunk;
    }
    if (executed) *executed = total;
    return ESIM_OK;
}

// Per-kernel timing of the chunk pass: an event in front of every kernel (kind = which one), ESIM_CK_N closes a sequence.
// kd_resolve turns consecutive events into durations once the stream has drained.
void kd_mark(esim_ctx_impl *c, int kind)
{
    if (!c->kdetail) return;
    if (c->kd_used == c->kdev.size()) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return; c->kdev.push_back(e); c->kd_kind.push_back(0); }
    if (hipEventRecord(c->kdev[c->kd_used], c->stream) != hipSuccess) return;
    c->kd_kind[c->kd_used++] = kind;
}

void kd_resolve(esim_ctx_impl *c)
{
    for (size_t i = 0; i + 1 < c->kd_used; ++i) {
        if (c->kd_kind[i] >= ESIM_CK_N) continue;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, c->kdev[i], c->kdev[i + 1]) == hipSuccess) { c->kd_ms[c->kd_kind[i]] += ms; c->kd_calls[c->kd_kind[i]]++; }
    }
    c->kd_used = 0;
}

// The kernels of one time-parallel chunk; they take the chunk (first step, length, whether it may run this way)
// from the control block as k_decide left it, and do nothing when it may not.
// then_next: also prepare the chunk after it (census ahead + decisions: what k_future and k_decide do), for steps up to limit_t.
// While few citizens are Infected the books, the log scatter, the clean-up and that preparation are ONE single-workgroup
// kernel (a kernel boundary costs more than these steps); with many, the scatter and clean-up need the whole chip.
// then_next: 0 nothing, 1 census ahead + decisions of the next chunk, 2 census ahead only.
// marks -> fold -> draw -> units of one chunk: on the persistent item map (unsharded contexts; k_map_enter only enters who turns
// Infected in the chunk, after a k_map_clear everybody who is Infected in it) or with the map rebuilt and torn down per chunk
// (sharded contexts, ESIM_PMAP=0).
// A chunk with few Infected is nothing but the latency of its kernels: those run on 64 workgroups instead of 1024 then (measured
// on york, whose chunks are all of that kind: 3.56 instead of 4.0 ms for the 5000 steps).  The choice follows what the last
// read-back showed, so bursts are kept short while it is in force (the epidemic may double within a hundred steps).
bool tiny_chunk(const esim_ctx_impl *c) { return c->tiny_pairs && c->last_chunk_pairs <= c->tiny_pairs && c->d.world == 1u && c->d.n_shards == 1u && !c->pmap; }
bool small_chunk(const esim_ctx_impl *c) { return c->small_grid && c->last_chunk_pairs < 4096u && c->grid_chunk > c->small_grid && !std::getenv("ESIM_GRID_CHUNK"); }

void enqueue_chunk_front(esim_ctx_impl *c)
{
    Dev &d = c->d;
    const bool pm = c->pmap && d.world == 1u;
    const uint32_t g = small_chunk(c) ? c->small_grid : c->grid_chunk, g_draw = g * (small_chunk(c) ? c->small_mult : c->draw_mult), g_units = g * (small_chunk(c) ? c->small_mult : c->units_mult);
    if (pm) {
        if (!c->map_valid || c->pmap_since_rebuild >= c->pmap_rebuild_every) {
            kd_mark(c, ESIM_CK_MAP_CLEAR);
            hipLaunchKernelGGL(k_map_clear, dim3(COUNT_GRID), dim3(TPB), 0, c->stream, d);
            hipLaunchKernelGGL(k_map_reset, dim3(1), dim3(64), 0, c->stream, d);
            (void)hipMemsetAsync(d.sch_ring, 0, sizeof(uint32_t) * (size_t)(d.n_sch ? d.n_sch : 1) * 2u * SCH_RING, c->stream);
            c->pmap_since_rebuild = 0;
        }
        c->map_valid = true; c->pmap_used = true; c->pmap_since_rebuild++;
        kd_mark(c, ESIM_CK_MARKS);
        hipLaunchKernelGGL(k_map_enter, dim3(g), dim3(TPB), 0, c->stream, d);
        kd_mark(c, ESIM_CK_FOLD);
        hipLaunchKernelGGL(k_chunk_fold<true>, dim3(g), dim3(TPB), 0, c->stream, d);
        kd_mark(c, ESIM_CK_DRAW);
        hipLaunchKernelGGL(k_chunk_draw<true>, dim3(g_draw), dim3(TPB), 0, c->stream, d, SUBQ);
        kd_mark(c, ESIM_CK_UNITS);
        hipLaunchKernelGGL(k_chunk_units<true>, dim3(g_units), dim3(TPB), 0, c->stream, d);
        return;
    }
    c->map_valid = false;
    kd_mark(c, ESIM_CK_MARKS);
    hipLaunchKernelGGL(k_chunk_marks, dim3(g), dim3(TPB), 0, c->stream, d);
    kd_mark(c, ESIM_CK_FOLD);
    hipLaunchKernelGGL(k_chunk_fold<false>, dim3(g), dim3(TPB), 0, c->stream, d);
    kd_mark(c, ESIM_CK_DRAW);
    hipLaunchKernelGGL(k_chunk_draw<false>, dim3(g_draw), dim3(TPB), 0, c->stream, d, g * (TPB / 64u));
    kd_mark(c, ESIM_CK_UNITS);
    hipLaunchKernelGGL(k_chunk_units<false>, dim3(g_units), dim3(TPB), 0, c->stream, d);
}

void enqueue_parallel_chunk(esim_ctx_impl *c, int then_next, uint32_t limit_t)
{
    Dev &d = c->d;
    const bool small = c->last_chunk_pairs < 1024u;
    enqueue_chunk_front(c);
    if (!small) { kd_mark(c, ESIM_CK_COUNT); hipLaunchKernelGGL(k_chunk_count, dim3(COUNT_GRID), dim3(TPB), 0, c->stream, d); }
    kd_mark(c, ESIM_CK_BOOKS);
    hipLaunchKernelGGL(k_chunk_books, dim3(1), dim3(FIN_TPB), 0, c->stream, d, small ? 1 : 0, then_next, (uint32_t)c->xf_n, limit_t);
    if (!small) { kd_mark(c, ESIM_CK_SCATTER); hipLaunchKernelGGL(k_chunk_scatter, dim3(COUNT_GRID), dim3(TPB), 0, c->stream, d); }   // (same grid as k_chunk_count: their workgroups pair up)
    kd_mark(c, ESIM_CK_N);
}

// One time-parallel chunk under a vaccination programme: census ahead, the plan of the chunk's vaccinations and what it does
// to the Infected census, the decisions, then the pass itself in its wide form.  Every kernel takes the chunk from the
// control block; a chunk that cannot run (no plan possible, a step that must run sequentially first) is a no-op.
void enqueue_vax_chunk(esim_ctx_impl *c, uint32_t limit_t)
{
    Dev &d = c->d;
    if (c->quiet && !(c->pmap && d.world == 1u) && d.world == 1u) {
        // Nobody is Exposed or Infected any more (the last read-back said so, and nobody is infected from outside): what is left of the
        // run is the vaccination programme.  No marks, no draws, nothing to scatter: the plan, the decisions, the census the
        // vaccinations move, the books, the words (York: the last 3400 of its 5000 steps are of this kind).
        kd_mark(c, ESIM_CK_VAX);
        hipLaunchKernelGGL(k_chunk_vax<false>, dim3(FREE_MAX + 1u), dim3(FIN_TPB), 0, c->stream, d, (uint32_t)c->xf_n, limit_t, 0);
        kd_mark(c, ESIM_CK_DECIDE);
        hipLaunchKernelGGL(k_decide, dim3(1), dim3(64), 0, c->stream, d, (uint32_t)c->xf_n, limit_t, 1, 0);
        c->map_valid = false;
        kd_mark(c, ESIM_CK_COUNT);
        hipLaunchKernelGGL(k_chunk_count, dim3(COUNT_GRID), dim3(TPB), 0, c->stream, d);
        kd_mark(c, ESIM_CK_BOOKS);
        hipLaunchKernelGGL(k_chunk_books, dim3(1), dim3(FIN_TPB), 0, c->stream, d, 0, 0, (uint32_t)c->xf_n, limit_t);
        kd_mark(c, ESIM_CK_VAX_FINAL);
        hipLaunchKernelGGL(k_chunk_vax_final, dim3(FREE_MAX), dim3(TPB), 0, c->stream, d);
        kd_mark(c, ESIM_CK_N);
        return;
    }
    kd_mark(c, ESIM_CK_VAX);
    hipLaunchKernelGGL(k_chunk_vax<false>, dim3(FREE_MAX + 1u), dim3(FIN_TPB), 0, c->stream, d, (uint32_t)c->xf_n, limit_t, 0);   // (+ the census ahead)
    // (persistent map: a rebuild has to be decided BEFORE the plan's cancellation records go into the map)
    const bool pm = c->pmap && d.world == 1u;
    if (pm && (!c->map_valid || c->pmap_since_rebuild >= c->pmap_rebuild_every)) {
        kd_mark(c, ESIM_CK_MAP_CLEAR);
        hipLaunchKernelGGL(k_map_clear, dim3(COUNT_GRID), dim3(TPB), 0, c->stream, d);
        hipLaunchKernelGGL(k_map_reset, dim3(1), dim3(64), 0, c->stream, d);
        (void)hipMemsetAsync(d.sch_ring, 0, sizeof(uint32_t) * (size_t)(d.n_sch ? d.n_sch : 1) * 2u * SCH_RING, c->stream);
        c->pmap_since_rebuild = 0; c->map_valid = true;
    }
    if (pm) {                                     // (the plan's vaccinations of citizens the persistent map holds: noted for k_map_enter)
        kd_mark(c, ESIM_CK_VAX_ADJ);
        hipLaunchKernelGGL(k_chunk_vax_adj, dim3(FREE_MAX), dim3(TPB), 0, c->stream, d, (uint32_t)c->xf_n, limit_t, 1);
    }
    kd_mark(c, ESIM_CK_DECIDE);
    hipLaunchKernelGGL(k_decide, dim3(1), dim3(64), 0, c->stream, d, (uint32_t)c->xf_n, limit_t, 1, 0);
    enqueue_chunk_front(c);
    if (!pm && d.world == 1u && c->vax_repair && (c->repair_armed || c->vax_repair_always)) {
        // bus exposures of citizens the plan vaccinates later: the plan of the steps behind is repaired instead of the chunk being cut
        kd_mark(c, ESIM_CK_VAX_REPAIR);
        hipLaunchKernelGGL(k_chunk_lost, dim3(1), dim3(FIN_TPB), 0, c->stream, d);
        hipLaunchKernelGGL(k_chunk_vax<true>, dim3(FREE_MAX), dim3(FIN_TPB), 0, c->stream, d, (uint32_t)c->xf_n, limit_t, 0);
    }
    kd_mark(c, ESIM_CK_COUNT);
    hipLaunchKernelGGL(k_chunk_count, dim3(COUNT_GRID), dim3(TPB), 0, c->stream, d);
    kd_mark(c, ESIM_CK_BOOKS);
    hipLaunchKernelGGL(k_chunk_books, dim3(1), dim3(FIN_TPB), 0, c->stream, d, 0, 0, (uint32_t)c->xf_n, limit_t);
    kd_mark(c, ESIM_CK_SCATTER);
    hipLaunchKernelGGL(k_chunk_scatter, dim3(COUNT_GRID), dim3(TPB), 0, c->stream, d);   // (same grid as k_chunk_count: their workgroups pair up)
    kd_mark(c, ESIM_CK_VAX_FINAL);
    hipLaunchKernelGGL(k_chunk_vax_final, dim3(FREE_MAX), dim3(TPB), 0, c->stream, d);
    kd_mark(c, ESIM_CK_N);
}

// One pipelined chunk.  Precondition: k_future ran for the current step (and, when sharded, buffer F was
// all-reduced).  k_decide finds how many of the next n_ahead steps can run before a vaccination programme
// would start; those run as one k_pipe each and k_batch_finish writes their books.  *executed = steps run.
int run_chunk(esim_ctx_impl *c, uint32_t n_ahead, uint32_t *executed, Ctrl *state_before)
{
    Dev &d = c->d;
    c->ctrl_fresh = false;
    hipLaunchKernelGGL(k_decide, dim3(1), dim3(64), 0, c->stream, d, n_ahead, c->P.max_steps, c->time_parallel ? 1 : 0, 0);
    Ctrl h;
    HIP_TRY(c, hipMemcpyAsync(&h, d.ctrl, sizeof h, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (state_before) *state_before = h;
    if (h.error) return fail(c, -(int)h.error, "device-side error (raised at check " + std::to_string(h.err_where) + ", esim_device.h ERR_AT_*)");
    const uint32_t n = h.chunk_ok, t0 = h.t;
    c->last_chunk_pairs = h.chunk_pairs;
    *executed = 0;
    if (n == 0) return ESIM_OK;
    if (h.chunk_parallel) {
        // every step of the chunk in one pass: marks of all steps, draws of all (item, step) pairs, then the books
        const bool tk = c->kernel_timing;
        if (tk) { if (!c->cev[0]) { (void)hipEventCreate(&c->cev[0]); (void)hipEventCreate(&c->cev[1]); } HIP_TRY(c, hipEventRecord(c->cev[0], c->stream)); }
        enqueue_parallel_chunk(c, 0, 0u);
        if (tk) { HIP_TRY(c, hipEventRecord(c->cev[1], c->stream)); HIP_TRY(c, hipEventSynchronize(c->cev[1])); float ms; HIP_TRY(c, hipEventElapsedTime(&ms, c->cev[0], c->cev[1])); c->chunk_ms += ms; }
        HIP_TRY(c, hipGetLastError());
        // The chunk takes itself back when the persistent map cannot serve it (k_map_enter: a map built under a lockdown, and a
        // schedule with working hours): what was executed is read, not assumed; the map is rebuilt and the chunk enqueued again.
        Ctrl after;
        int rc2 = read_ctrl(c, &after);
        if (rc2) return rc2;
        if (after.error) return fail(c, -(int)after.error, "device-side error (raised at check " + std::to_string(after.err_where) + ")");
        if (after.t == t0 && c->map_valid) {
            c->map_valid = false;
            hipLaunchKernelGGL(k_decide, dim3(1), dim3(64), 0, c->stream, d, n_ahead, c->P.max_steps, 1, 0);
            enqueue_parallel_chunk(c, 0, 0u);
            if ((rc2 = read_ctrl(c, &after))) return rc2;
            if (after.error) return fail(c, -(int)after.error, "device-side error (raised at check " + std::to_string(after.err_where) + ")");
        }
        const uint32_t ran = after.t - t0;
        c->chunk_steps += ran; c->chunk_count += ran ? 1u : 0u;
        *executed = ran;
        return ESIM_OK;
    }
    hipLaunchKernelGGL(k_infected_dec, dim3(c->grid_infected), dim3(TPB), 0, c->stream, d, t0, 0u);
    for (uint32_t j = 0; j < n; ++j) {
        bool tk = c->kernel_timing && ((t0 + j) % c->kernel_timing_stride) == 0;
        if (tk && c->pkev_used + 2 > c->pkev.size())
            for (int i = 0; i < 2 && tk; ++i) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) tk = false; else c->pkev.push_back(e); }
        if (tk) HIP_TRY(c, hipEventRecord(c->pkev[c->pkev_used], c->stream));
        hipLaunchKernelGGL(k_pipe, dim3(c->grid_expose + c->grid_infected), dim3(TPB), 0, c->stream, d, t0 + j, j, c->grid_expose, j + 1 < n ? 1 : 0);
        if (tk) { HIP_TRY(c, hipEventRecord(c->pkev[c->pkev_used + 1], c->stream)); c->pkev_used += 2; }
    }
    hipLaunchKernelGGL(k_batch_finish, dim3(1), dim3(FIN_TPB), 0, c->stream, d, t0, n);
    HIP_TRY(c, hipGetLastError());
    c->pipe_steps += n;
    *executed = n;
    return ESIM_OK;
}

// Runs up to n_steps steps of an unsharded context: pipelined chunks while no vaccination programme runs,
// sequential steps from the step that starts it.
int run_steps(esim_ctx_impl *c, uint32_t n_steps, bool allow_early_stop, uint32_t *executed)
{
    Dev &d = c->d;
    uint32_t remaining = n_steps, total = 0;
    int rc;
    bool sequential_only = !c->pipeline || c->phase_timing;
    bool stalled = false, probing = false;
    uint32_t backoff = 0, sync_chunks_left = 0;
    // a vaccination programme runs: chunks with their vaccinations planned, or sequential steps (short runs: sequential)
    const bool vax_ok = c->time_parallel && c->vax_chunks && d.n_shards == 1u;
    bool vax_regime = vax_ok && c->elig_seen;
    if (c->elig_seen && (!vax_ok || n_steps < 8u)) sequential_only = true;
    uint32_t vax_fail = 0;
    while (remaining > 0) {
        c->ctrl_fresh = false;
        if (sequential_only) {
            uint32_t done = 0;
            if ((rc = run_sequential(c, remaining, allow_early_stop, &done))) return rc;
            total += done;
            break;
        }
        if (vax_regime) {
            // bursts of planned chunks; whatever stops one (a cut: the step at ctrl->t must run sequentially; no plan possible;
            // a chunk that does not fit the one-pass form) is answered with sequential steps, more of them when it keeps happening
            const uint32_t first = c->host_t, limit_t = first + remaining - 1u;
            const uint32_t bursts = vax_fail ? 1u : std::min<uint32_t>((remaining + (uint32_t)c->xf_n - 1u) / (uint32_t)c->xf_n + 1u, 8u);   // (one more than fit: cuts)
            const bool tk = c->kernel_timing;
            if (tk) { if (!c->cev[0]) { (void)hipEventCreate(&c->cev[0]); (void)hipEventCreate(&c->cev[1]); } HIP_TRY(c, hipEventRecord(c->cev[0], c->stream)); }
            for (uint32_t g = 0; g < bursts; ++g) enqueue_vax_chunk(c, limit_t);
            if (tk) HIP_TRY(c, hipEventRecord(c->cev[1], c->stream));
            Ctrl h;
            if ((rc = burst_readback(c, first, std::min<uint32_t>(remaining, bursts * (uint32_t)c->xf_n), &h))) return rc;
            HIP_TRY(c, hipGetLastError());
            if (h.error) return fail(c, -(int)h.error, "device-side error (raised at check " + std::to_string(h.err_where) + ", esim_device.h ERR_AT_*)");
            const uint32_t done = h.t - first;
            c->last_chunk_pairs = h.chunk_pairs;
            if (tk && done) { float ms; HIP_TRY(c, hipEventElapsedTime(&ms, c->cev[0], c->cev[1])); c->chunk_ms += ms; c->chunk_steps += done; c->chunk_count += (done + (uint32_t)c->xf_n - 1u) / (uint32_t)c->xf_n; }
            c->quiet = h.quiet != 0u;
            if (h.vax_cuts > c->vax_chunk_cuts) c->repair_armed = true;   // (a chunk was cut: from now on the plan is repaired instead)
            c->vax_chunk_steps += done; c->vax_chunk_cuts = h.vax_cuts; c->vax_chunk_repairs = h.vax_repairs;
            c->host_t = h.t; total += done; remaining -= done;
            if (h.finished && allow_early_stop) break;
            if (remaining == 0) break;
            if (done) { vax_fail = 0; continue; }                       // (cut chunks advance less; the next one starts at the cut)
            c->map_valid = false;                                        // (no progress: whatever the reason, the map is rebuilt next)
            if (std::getenv("ESIM_DEBUG"))
                std::fprintf(stderr, "[esim] vax burst without progress at t=%u: chunk_ok=%u parallel=%u vax_chunk=%u cut=%u pairs=%u fits_flag=%u elig=%u bursts=%u\n",
                             h.t, h.chunk_ok, h.chunk_parallel, h.vax_chunk, h.chunk_cut, h.chunk_pairs, 0u, h.elig_count, bursts);
            vax_fail = std::min<uint32_t>(vax_fail + 1u, 8u);
            uint32_t seq = 0;
            const uint32_t want = std::min<uint32_t>(remaining, vax_fail <= 1u ? 1u : (vax_fail <= 3u ? 8u : (uint32_t)c->xf_n));
            if ((rc = run_sequential(c, want, allow_early_stop, &seq))) return rc;
            total += seq; remaining -= seq;
            if (seq < want) break;                                       // the run ended
            continue;
        }
        if (c->time_parallel && !stalled && sync_chunks_left == 0) {
            // Chunks are enqueued back to back without waiting for their k_decide: every kernel takes the chunk from the
            // control block and is a no-op when the chunk cannot run time-parallel (then the steps simply do not advance,
            // which the read-back below sees, and the synchronous path further down takes over for one chunk).
            const uint32_t first = c->host_t, limit_t = first + remaining - 1u;
            const uint32_t bursts = std::min<uint32_t>((remaining + (uint32_t)c->xf_n - 1u) / (uint32_t)c->xf_n, probing ? 1u : (small_chunk(c) ? 4u : 16u));   // (the form of a chunk's book-keeping is chosen from what the last read-back showed)
            const bool tk = c->kernel_timing;
            if (tk) { if (!c->cev[0]) { (void)hipEventCreate(&c->cev[0]); (void)hipEventCreate(&c->cev[1]); } HIP_TRY(c, hipEventRecord(c->cev[0], c->stream)); }
            if (tiny_chunk(c)) {
                // few Infected: every chunk of the burst is ONE launch of one workgroup (esim_kernels_tiny.h); a chunk that has
                // outgrown that form does not advance, which the read-back below sees
                c->map_valid = false;
                for (uint32_t g = 0; g < bursts; ++g) {
                    kd_mark(c, ESIM_CK_TINY);
                    hipLaunchKernelGGL(k_chunk_tiny, dim3(1), dim3(FIN_TPB), 0, c->stream, d, g == 0u ? 1 : 0, g + 1u < bursts ? 1 : 0, (uint32_t)c->xf_n, limit_t);
                }
                kd_mark(c, ESIM_CK_N);
            } else {
            kd_mark(c, ESIM_CK_FUTURE);
            hipLaunchKernelGGL(k_future, dim3(1), dim3(FIN_TPB), 0, c->stream, d, (uint32_t)c->xf_n, limit_t);
            kd_mark(c, ESIM_CK_DECIDE);
            hipLaunchKernelGGL(k_decide, dim3(1), dim3(64), 0, c->stream, d, (uint32_t)c->xf_n, limit_t, 1, 0);
            for (uint32_t g = 0; g < bursts; ++g) enqueue_parallel_chunk(c, g + 1u < bursts ? 1 : 0, limit_t);
            }
            if (tk) HIP_TRY(c, hipEventRecord(c->cev[1], c->stream));
            Ctrl h;
            if ((rc = burst_readback(c, first, std::min<uint32_t>(remaining, bursts * (uint32_t)c->xf_n), &h))) return rc;
            HIP_TRY(c, hipGetLastError());
            if (h.error) return fail(c, -(int)h.error, "device-side error (raised at check " + std::to_string(h.err_where) + ", esim_device.h ERR_AT_*)");
            const uint32_t done = h.t - first;
            c->last_chunk_pairs = h.chunk_pairs;
            if (tk && done) { float ms; HIP_TRY(c, hipEventElapsedTime(&ms, c->cev[0], c->cev[1])); c->chunk_ms += ms; c->chunk_steps += done; c->chunk_count += (done + (uint32_t)c->xf_n - 1u) / (uint32_t)c->xf_n; }
            c->host_t = h.t; total += done; remaining -= done;
            if (h.finished) break;
            if (done == 0) { backoff = std::min<uint32_t>(64u, backoff ? backoff * 2u : 1u); sync_chunks_left = backoff; probing = true; c->map_valid = false; }   // (e.g. a map built under a lockdown: rebuilt next)
            else { backoff = 0; probing = done < std::min<uint32_t>(remaining + done, bursts * (uint32_t)c->xf_n); }
            if (done < std::min<uint32_t>(remaining + done, bursts * (uint32_t)c->xf_n)) stalled = true;   // something other than a full time-parallel chunk is next
            continue;
        }
        stalled = false;
        if (sync_chunks_left) --sync_chunks_left;
        const uint32_t n_ahead = std::min<uint32_t>(remaining, (uint32_t)c->xf_n);
        hipLaunchKernelGGL(k_future, dim3(1), dim3(FIN_TPB), 0, c->stream, d, n_ahead, c->P.max_steps);
        uint32_t done = 0;
        Ctrl before;
        if ((rc = run_chunk(c, n_ahead, &done, &before))) return rc;
        c->host_t = before.t + done; total += done; remaining -= done;
        if (before.finished) break;
        if (done < n_ahead && remaining > 0) {
            if (before.have_elig || before.vacc_active) {
                c->elig_seen = true;
                if (vax_ok && remaining >= 8u) vax_regime = true; else sequential_only = true;
                continue;
            }
            // the next step starts the vaccination programme (or a limit was hit): one sequential step, then look again
            uint32_t one = 0;
            if ((rc = run_sequential(c, 1, allow_early_stop, &one))) return rc;
            total += one; remaining -= one;
            if (one == 0) break;
            Ctrl h;
            HIP_TRY(c, hipMemcpyAsync(&h, d.ctrl, sizeof h, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            if (h.have_elig) {                                            // that step started the programme
                c->elig_seen = true;
                if (vax_ok && remaining >= 8u) vax_regime = true; else sequential_only = true;
            }
        }
        if (allow_early_stop && done > 0) {
            // a chunk may have ended the run (disease gone): k_batch_finish set `finished`
            Ctrl h;
            HIP_TRY(c, hipMemcpyAsync(&h, d.ctrl, sizeof h, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            if (h.finished) { c->host_t = h.t; break; }
        }
    }
    if (executed) *executed = total;
    return ESIM_OK;
}

}  // namespace

extern "C" int esim_step(esim_ctx *ctx, esim_step_result *out)
{
    esim_ctx_impl *c = CTX(ctx);
    int rc = check_budget(c, 1);
    if (rc) return rc;
    if (c->d.n_shards > 1) return fail(c, ESIM_ESTATE, "esim_step: a sharded population needs the split-phase calls and an all-reduce");
    HIP_TRY(c, hipSetDevice(c->P.device));
    if ((rc = run_steps(c, 1, false, nullptr))) return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (out) HIP_TRY(c, hipMemcpy(out, &c->d.records[c->host_t - 1], sizeof *out, hipMemcpyDeviceToHost));
    return device_error(c);
}

extern "C" int esim_run(esim_ctx *ctx, uint32_t n_steps, int stop_when_done, esim_step_result *out_array, uint32_t *n_done)
{
    esim_ctx_impl *c = CTX(ctx);
    if (c && c->host_trace) { c->ht.clear(); ht_mark(c, "enter"); }
    int rc = check_budget(c, n_steps);
    if (rc) return rc;
    if (c->d.n_shards > 1) return fail(c, ESIM_ESTATE, "esim_run: a sharded population needs the split-phase calls and an all-reduce");
    HIP_TRY(c, hipSetDevice(c->P.device));
    const uint32_t first = c->host_t;
    const uint32_t flag = stop_when_done ? 1u : 0u;
    if (flag != c->stop_flag_dev) {                      // (the flag lives in the control block; written only when it changes)
        HIP_TRY(c, hipMemcpyAsync(&c->d.ctrl->stop_when_done, &flag, sizeof flag, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        c->stop_flag_dev = flag;
    }
    // bursts of chunk passes bring their records back together with the control block (burst_readback); whatever other forms
    // ran is fetched below
    c->pin_track = out_array != nullptr; c->pin_first = first; c->pin_valid = 0; c->ctrl_fresh = false;
    rc = run_steps(c, n_steps, stop_when_done != 0, nullptr);
    c->pin_track = false;
    if (rc) return rc;
    Ctrl h;
    if (c->ctrl_fresh) h = *c->pin_ctrl;
    else if ((rc = read_ctrl(c, &h))) return rc;
    if (h.error) return fail(c, -(int)h.error, "device-side error (S underflow / vaccination window exhausted / a chunk table check: " + std::to_string(h.err_where) + ")");
    const uint32_t done = h.steps_done >= first ? h.steps_done - first + 1 : 0;
    if (std::getenv("ESIM_DEBUG"))
        std::fprintf(stderr, "[esim] esim_run(%u steps from %u): done %u, t=%u steps_done=%u finished=%u chunk_ok=%u parallel=%u, records mirrored %u, control block %s\n",
                     n_steps, first, done, h.t, h.steps_done, h.finished, h.chunk_ok, h.chunk_parallel, c->pin_valid, c->ctrl_fresh ? "from the burst" : "read now");
    c->host_t = first + done;
    if (out_array && done) {
        const uint32_t have = std::min(c->pin_valid, done);
        if (have < done) {
            HIP_TRY(c, hipMemcpyAsync(c->pin_rec + first + have, c->d.records + first + have, sizeof(esim_step_result) * (done - have), hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
        }
        std::memcpy(out_array, c->pin_rec + first, sizeof(esim_step_result) * done);
    }
    if (n_done) *n_done = done;
    if (c->host_trace) {
        ht_mark(c, "exit");
        std::fprintf(stderr, "[esim host trace] esim_run(%u):", n_steps);
        for (size_t i = 1; i < c->ht.size(); ++i) std::fprintf(stderr, " %s +%.1f us;", c->ht[i].first, c->ht[i].second - c->ht[i - 1].second);
        std::fprintf(stderr, " total %.1f us\n", c->ht.back().second - c->ht.front().second);
    }
    return ESIM_OK;
}

// ---- the exchange between shards -----------------------------------------------------------------------------------
// SUM all-reduces of small uint32 device buffers (the commuter exchange and the census).  Two transports:
// RCCL over xGMI, with the communicator owned by the library and the collective enqueued on the context's own stream between
// its kernels (no host synchronisation per step); or a caller-supplied function (tests on one GPU: gloo through the Python
// binding), which is called with the stream drained.  librccl is loaded on first use, so a build without it still runs.
namespace {

struct RcclApi {
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};

RcclApi &rccl()
{
    static RcclApi api;
    static bool tried = false;
    if (tried) return api;
    tried = true;
    void *h = nullptr;
    // a copy the process has loaded already (e.g. the one PyTorch ships) is reused: one RCCL runtime per process
    for (const char *name : { "librccl.so", "librccl.so.1" }) if ((h = dlopen(name, RTLD_NOW | RTLD_NOLOAD))) break;
    if (!h) for (const char *name : { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" }) if ((h = dlopen(name, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!h) return api;
    api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    api.CommInitRank = (decltype(api.CommInitRank))dlsym(h, "ncclCommInitRank");
    api.CommDestroy = (decltype(api.CommDestroy))dlsym(h, "ncclCommDestroy");
    api.CommAbort = (decltype(api.CommAbort))dlsym(h, "ncclCommAbort");
    api.Send = (decltype(api.Send))dlsym(h, "ncclSend");
    api.Recv = (decltype(api.Recv))dlsym(h, "ncclRecv");
    api.GroupStart = (decltype(api.GroupStart))dlsym(h, "ncclGroupStart");
    api.GroupEnd = (decltype(api.GroupEnd))dlsym(h, "ncclGroupEnd");
    api.AllReduce = (decltype(api.AllReduce))dlsym(h, "ncclAllReduce");
    api.AllGather = (decltype(api.AllGather))dlsym(h, "ncclAllGather");
    api.GetErrorString = (decltype(api.GetErrorString))dlsym(h, "ncclGetErrorString");
    api.ok = api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.CommAbort && api.AllReduce && api.AllGather && api.Send && api.Recv &&
             api.GroupStart && api.GroupEnd && api.GetErrorString;
    return api;
}

void comm_release(esim_ctx_impl *c)
{
    if (c->nccl && rccl().ok) rccl().CommDestroy(c->nccl);
    c->nccl = nullptr;
}

// SUM all-reduce of n uint32 at device pointer buf over the shards, in place, ordered after everything enqueued so far.
// `which` names the buffer for a caller's transport (0 A, 1 B, 2 F, 3 plan liveness, 4 commuter records, 5 cuts, 6 records).
int exchange_buf(esim_ctx_impl *c, int which, uint32_t *buf, size_t n)
{
    c->comm_calls++;
    if (c->nccl) {
        ncclResult_t r = rccl().AllReduce(buf, buf, n, ncclUint32, ncclSum, c->nccl, c->stream);
        if (r != ncclSuccess) return fail(c, ESIM_ENODEVICE, std::string("ncclAllReduce: ") + rccl().GetErrorString(r));
        return ESIM_OK;
    }
    if (c->comm_fn) {
        // the caller's transport works on host memory: stage through a host buffer with the stream drained
        c->comm_stage.resize(n);
        HIP_TRY(c, hipMemcpyAsync(c->comm_stage.data(), buf, sizeof(uint32_t) * n, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        if (c->comm_fn(c->comm_user, which, c->comm_stage.data(), n) != 0) return fail(c, ESIM_ENODEVICE, "the caller's all-reduce failed");
        HIP_TRY(c, hipMemcpyAsync(buf, c->comm_stage.data(), sizeof(uint32_t) * n, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        return ESIM_OK;
    }
    return fail(c, ESIM_ESTATE, "sharded run without a communicator (esim_comm_init_rccl / esim_comm_init_callback)");
}

// All-gather: every rank contributes the `per_rank` words at buf + rank * per_rank and receives everybody's.  Over a caller's
// transport: a SUM all-reduce of the whole buffer, the other ranks' segments zeroed first (k_zero_segments).
int exchange_gather(esim_ctx_impl *c, int which, uint32_t *buf, size_t per_rank)
{
    if (c->nccl) {
        c->comm_calls++;
        ncclResult_t r = rccl().AllGather(buf + (size_t)c->comm_rank * per_rank, buf, per_rank, ncclUint32, c->nccl, c->stream);
        if (r != ncclSuccess) return fail(c, ESIM_ENODEVICE, std::string("ncclAllGather: ") + rccl().GetErrorString(r));
        return ESIM_OK;
    }
    return exchange_buf(c, which, buf, per_rank * (size_t)c->comm_world);
}

int wait_stream(esim_ctx_impl *c);

// All-to-all: rank s sends the `seg` words at out + d * seg to rank d, and receives rank r's words for it at in + r * seg.  RCCL:
// one group of ncclSend / ncclRecv pairs on the context's stream (every pair of shards talks over its own xGMI link).  A caller's
// transport only has a SUM all-reduce: the ranks' rows of the [sender][receiver] matrix are summed and each picks its column.
int exchange_alltoall(esim_ctx_impl *c, int which, const uint32_t *out, uint32_t *in, size_t seg)
{
    const int W = c->comm_world, me = c->comm_rank;
    if (W <= 1) return ESIM_OK;
    c->comm_calls++;
    if (c->nccl) {
        ncclResult_t r = rccl().GroupStart();
        for (int p = 0; p < W && r == ncclSuccess; ++p) {
            if (p == me) continue;
            r = rccl().Send(out + (size_t)p * seg, seg, ncclUint32, p, c->nccl, c->stream);
            if (r == ncclSuccess) r = rccl().Recv(in + (size_t)p * seg, seg, ncclUint32, p, c->nccl, c->stream);
        }
        const ncclResult_t e = rccl().GroupEnd();
        if (r == ncclSuccess) r = e;
        if (r != ncclSuccess) return fail(c, ESIM_ENODEVICE, std::string("ncclSend/ncclRecv: ") + rccl().GetErrorString(r));
        return ESIM_OK;
    }
    if (c->comm_fn) {
        const size_t n = (size_t)W * W * seg;
        c->comm_stage.assign(n, 0u);
        HIP_TRY(c, hipMemcpyAsync(c->comm_stage.data() + (size_t)me * W * seg, out, sizeof(uint32_t) * (size_t)W * seg, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        for (size_t i = 0; i < seg; ++i) c->comm_stage[((size_t)me * W + me) * seg + i] = 0u;          // (nothing goes to oneself)
        if (c->comm_fn(c->comm_user, which, c->comm_stage.data(), n) != 0) return fail(c, ESIM_ENODEVICE, "the caller's all-reduce failed");
        for (int p = 0; p < W; ++p)
            if (p != me) HIP_TRY(c, hipMemcpyAsync(in + (size_t)p * seg, c->comm_stage.data() + ((size_t)p * W + me) * seg, sizeof(uint32_t) * seg, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        return ESIM_OK;
    }
    return fail(c, ESIM_ESTATE, "sharded run without a communicator (esim_comm_init_rccl / esim_comm_init_callback)");
}

// what the exchange of sharded chunks needs once the number of ranks is known; and the ranks' shards are checked against each
// other -- one world (n_citizens_global, shared tables of the same size), rank r holding the r-th stretch of the global
// citizen ids -- with one small all-reduce: a communicator over shards that do not belong together would run without an
// error and give wrong records.
int comm_buffers(esim_ctx_impl *c)
{
    if (!c->uploaded) return fail(c, ESIM_ESTATE, "esim_comm_init: upload the population first");
    HIP_TRY(c, hipSetDevice(c->P.device));
    Dev &d = c->d;
    int rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (d.xs) { dev_free(c, d.xs); d.xs = nullptr; }
    if (c->xs_out) { dev_free(c, c->xs_out); c->xs_out = nullptr; }
    if (c->shared_mask) { dev_free(c, c->shared_mask); c->shared_mask = nullptr; }
    d.xs_out = nullptr; d.shared_mask = nullptr;
    if (const char *e = std::getenv("ESIM_XS_MODE")) c->xs_a2a = std::string(e) != "gather";
    if (c->xr) { dev_free(c, c->xr); c->xr = nullptr; c->xr_n = 0; }
    d.rank = (uint32_t)c->comm_rank; d.world = (uint32_t)c->comm_world;
    const size_t n = (size_t)d.world * (1u + 3u * (size_t)XS_CAP_MAX);
    d.xs_cap = 4096u;
    if (const char *e = std::getenv("ESIM_XS_CAP")) d.xs_cap = (uint32_t)std::min<long>(XS_CAP_MAX, std::max<long>(1, std::atol(e)));   // (tests: a segment that has to grow)
    if ((rc = dev_alloc(c, &d.xs, n))) return rc;
    HIP_TRY(c, hipMemset(d.xs, 0, sizeof(uint32_t) * n));
    // the layout check
    const uint32_t W = d.world;
    std::vector<uint32_t> rows((size_t)W * 5u, 0u);
    uint32_t *mine = &rows[(size_t)d.rank * 5u];
    mine[0] = d.id_base; mine[1] = d.n; mine[2] = d.n_global; mine[3] = d.n_shared_bld; mine[4] = d.n_shared_room;
    uint32_t *dv = nullptr;
    if ((rc = dev_alloc(c, &dv, rows.size()))) return rc;
    HIP_TRY(c, hipMemcpy(dv, rows.data(), sizeof(uint32_t) * rows.size(), hipMemcpyHostToDevice));
    rc = exchange_buf(c, 8, dv, rows.size());
    if (!rc) rc = wait_stream(c);
    if (!rc && hipMemcpy(rows.data(), dv, sizeof(uint32_t) * rows.size(), hipMemcpyDeviceToHost) != hipSuccess) rc = fail(c, ESIM_ENODEVICE, "esim_comm_init: read-back of the layout check failed");
    dev_free(c, dv);
    if (rc) return rc;
    uint64_t next = 0;
    for (uint32_t r = 0; r < W; ++r) {
        const uint32_t *q = &rows[(size_t)r * 5u];
        if (q[2] != d.n_global || q[3] != d.n_shared_bld || q[4] != d.n_shared_room || q[0] != next) {
            char msg[256];
            std::snprintf(msg, sizeof msg, "esim_comm_init: rank %u holds citizens [%u, %u) of %u with %u / %u shared buildings / rooms -- not shard %u of the world this rank's shard belongs to "
                          "(expected ids from %llu, %u citizens in all, %u / %u shared)", r, q[0], q[0] + q[1], q[2], q[3], q[4], r, (unsigned long long)next, d.n_global, d.n_shared_bld, d.n_shared_room);
            return fail(c, ESIM_EINVAL, msg);
        }
        next += q[1];
    }
    if (next != d.n_global) return fail(c, ESIM_EINVAL, "esim_comm_init: the ranks' shards do not add up to n_citizens_global (world size differs from the number of shards)");
    if (c->xs_a2a && W > 1) {
        // which shards have members in each shared building: every shard sets its own bit where it has, the bits are summed
        if ((rc = dev_alloc(c, &c->shared_mask, (size_t)d.n_shared_bld + 1u))) return rc;
        std::vector<uint32_t> bits((size_t)d.n_shared_bld + 1u, 0u);
        std::vector<int32_t> local((size_t)d.n_shared_bld + 1u, -1);
        if (d.n_shared_bld) HIP_TRY(c, hipMemcpy(local.data(), d.shared_bld, sizeof(int32_t) * d.n_shared_bld, hipMemcpyDeviceToHost));
        for (uint32_t k = 0; k < d.n_shared_bld; ++k) bits[k] = local[k] >= 0 ? 1u << d.rank : 0u;
        HIP_TRY(c, hipMemcpy(c->shared_mask, bits.data(), sizeof(uint32_t) * bits.size(), hipMemcpyHostToDevice));
        if ((rc = exchange_buf(c, 9, c->shared_mask, bits.size()))) return rc;
        if ((rc = wait_stream(c))) return rc;
        if ((rc = dev_alloc(c, &c->xs_out, n))) return rc;
        HIP_TRY(c, hipMemset(c->xs_out, 0, sizeof(uint32_t) * n));
        d.xs_out = c->xs_out; d.shared_mask = c->shared_mask;
    }
    return ESIM_OK;
}

int exchange(esim_ctx_impl *c, int which)
{
    Dev &d = c->d;
    uint32_t *buf = which == 2 ? d.xf : which ? d.xb : d.xa;
    const size_t n = which == 2 ? c->xf_n + 1 : which ? c->xb_n : c->xa_n;
    return exchange_buf(c, which, buf, n);
}

// The host's wait for a stream that holds RCCL collectives has a deadline: a peer that died or left (a crash, an exchange that
// failed on its side) would otherwise leave this rank inside a collective for ever.  On expiry the communicator is aborted
// (ncclCommAbort ends the collective kernels), the context is left without one, and the call fails with ESIM_ETIMEDOUT -- the
// caller is expected to exit with an error, as the reference does when step() fails (run/src/main.rs:306-308).
int wait_stream(esim_ctx_impl *c)
{
    if (!c->nccl) { HIP_TRY(c, hipStreamSynchronize(c->stream)); return ESIM_OK; }
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t e = hipStreamQuery(c->stream);
        if (e == hipSuccess) return ESIM_OK;
        if (e != hipErrorNotReady) return fail(c, ESIM_ENODEVICE, std::string("hipStreamQuery: ") + hipGetErrorString(e));
        const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (waited > c->comm_timeout_s) {
            (void)rccl().CommAbort(c->nccl);
            c->nccl = nullptr;
            char msg[200];
            std::snprintf(msg, sizeof msg, "rank %d: no progress on the stream for %.1f s inside a sharded run (a peer left or died); the RCCL communicator was aborted",
                          c->comm_rank, waited);
            return fail(c, ESIM_ETIMEDOUT, msg);
        }
        if (waited > 2e-3) std::this_thread::sleep_for(std::chrono::microseconds(waited > 0.1 ? 1000 : 20));
    }
}

// Every read-back of the control block in a sharded run: the shards' error fields are summed first (k_status_pack ->
// all-reduce -> k_status_unpack), so every rank sees any rank's device-side error in the same collective and takes the same
// return decision from the same word.
int sync_status(esim_ctx_impl *c, bool ex, Ctrl *h)
{
    Dev &d = c->d;
    int rc;
    hipLaunchKernelGGL(k_status_pack, dim3(1), dim3(64), 0, c->stream, d);
    if (ex && (rc = exchange_buf(c, 7, d.xe, XE_WORDS))) return rc;
    hipLaunchKernelGGL(k_status_unpack, dim3(1), dim3(64), 0, c->stream, d);
    HIP_TRY(c, hipMemcpyAsync(h, d.ctrl, sizeof *h, hipMemcpyDeviceToHost, c->stream));
    if ((rc = wait_stream(c))) return rc;
    if (h->peer_error) {
        const uint32_t code = err_decode(h->peer_error);
        char msg[240];
        std::snprintf(msg, sizeof msg, "device-side error %d on at least one shard (S underflow / vaccination window exhausted / a chunk table overflowed); every rank returns it",
                      -(int)code);
        return fail(c, -(int)code, msg);
    }
    return ESIM_OK;
}

}  // namespace

extern "C" int esim_comm_unique_id(void *out, size_t cap)
{
    if (!out || cap < sizeof(ncclUniqueId)) return ESIM_EINVAL;
    if (!rccl().ok) return fail(nullptr, ESIM_ENODEVICE, "librccl could not be loaded");
    ncclUniqueId id;
    ncclResult_t r = rccl().GetUniqueId(&id);
    if (r != ncclSuccess) return fail(nullptr, ESIM_ENODEVICE, std::string("ncclGetUniqueId: ") + rccl().GetErrorString(r));
    std::memcpy(out, &id, sizeof id);
    return ESIM_OK;
}

extern "C" int esim_comm_init_rccl(esim_ctx *ctx, const void *unique_id, size_t id_bytes, int rank, int world)
{
    esim_ctx_impl *c = CTX(ctx);
    if (!c || !unique_id || id_bytes < sizeof(ncclUniqueId) || rank < 0 || rank >= world || world > (int)ERR_MAX_WORLD) return fail(c, ESIM_EINVAL, "esim_comm_init_rccl: bad argument (0 <= rank < world <= 31)");
    if (!c->uploaded) return fail(c, ESIM_ESTATE, "esim_comm_init_rccl: upload the population first");
    if (!rccl().ok) return fail(c, ESIM_ENODEVICE, "librccl could not be loaded");
    HIP_TRY(c, hipSetDevice(c->P.device));
    if (c->nccl) { rccl().CommDestroy(c->nccl); c->nccl = nullptr; }
    ncclUniqueId id;
    std::memcpy(&id, unique_id, sizeof id);
    ncclResult_t r = rccl().CommInitRank(&c->nccl, world, id, rank);
    if (r != ncclSuccess) { c->nccl = nullptr; return fail(c, ESIM_ENODEVICE, std::string("ncclCommInitRank: ") + rccl().GetErrorString(r)); }
    c->comm_rank = rank; c->comm_world = world; c->comm_fn = nullptr;
    const int rc = comm_buffers(c);
    if (rc) { comm_release(c); c->comm_rank = 0; c->comm_world = 1; c->d.rank = 0; c->d.world = 1; }
    return rc;
}

extern "C" int esim_comm_init_callback(esim_ctx *ctx, esim_allreduce_fn fn, void *user, int rank, int world)
{
    esim_ctx_impl *c = CTX(ctx);
    if (!c || !fn || rank < 0 || rank >= world || world > (int)ERR_MAX_WORLD) return fail(c, ESIM_EINVAL, "esim_comm_init_callback: bad argument (0 <= rank < world <= 31)");
    if (!c->uploaded) return fail(c, ESIM_ESTATE, "esim_comm_init_callback: upload the population first");
    if (c->nccl) { rccl().CommDestroy(c->nccl); c->nccl = nullptr; }
    c->comm_fn = fn; c->comm_user = user; c->comm_rank = rank; c->comm_world = world;
    const int rc = comm_buffers(c);
    if (rc) { c->comm_fn = nullptr; c->comm_rank = 0; c->comm_world = 1; c->d.rank = 0; c->d.world = 1; }
    return rc;
}

extern "C" int esim_comm_set_timeout(esim_ctx *ctx, double seconds)
{
    esim_ctx_impl *c = CTX(ctx);
    if (!c || !(seconds > 0.0)) return fail(c, ESIM_EINVAL, "esim_comm_set_timeout: seconds must be positive");
    c->comm_timeout_s = seconds;
    return ESIM_OK;
}

extern "C" int esim_debug_inject_error(esim_ctx *ctx, int code)
{
    esim_ctx_impl *c = CTX(ctx);
    if (!c || !c->uploaded || code > -1 || code < -6) return fail(c, ESIM_EINVAL, "esim_debug_inject_error: code must be one of the ESIM_E* values");
    HIP_TRY(c, hipSetDevice(c->P.device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    const uint32_t v = (uint32_t)(-code);
    HIP_TRY(c, hipMemcpy(&c->d.ctrl->error, &v, sizeof v, hipMemcpyHostToDevice));
    return ESIM_OK;
}

extern "C" int esim_comm_stats(esim_ctx *ctx, uint64_t *collectives)
{
    esim_ctx_impl *c = CTX(ctx);
    if (!c) return ESIM_EINVAL;
    if (collectives) *collectives = c->comm_calls;
    return ESIM_OK;
}

namespace {

// One time-parallel chunk of a sharded run: what the shards exchange once per chunk instead of once per step --
// the liveness of the plan's candidates (V), the Infected commuters to shared buildings (S, all-gathered), the Infected census
// ahead with the "cannot" word (F), the steps with a cut (C).  Kernels and collectives are enqueued on the context's stream.
int enqueue_sharded_chunk(esim_ctx_impl *c, uint32_t limit_t, bool vax)
{
    Dev &d = c->d;
    int rc;
    hipLaunchKernelGGL(k_future, dim3(1), dim3(FIN_TPB), 0, c->stream, d, (uint32_t)c->xf_n, limit_t);
    if (vax) {
        hipLaunchKernelGGL(k_vax_live<false>, dim3(PLAN_W / TPB, FREE_MAX), dim3(TPB), 0, c->stream, d, (uint32_t)c->xf_n, limit_t);
        if ((rc = exchange_buf(c, 3, d.xv, XV_HEADER + (size_t)FREE_MAX * (PLAN_W / 32u)))) return rc;
        hipLaunchKernelGGL(k_chunk_vax<false>, dim3(FREE_MAX), dim3(FIN_TPB), 0, c->stream, d, (uint32_t)c->xf_n, limit_t, 1);
    }
    const size_t seg = 1u + 3u * (size_t)d.xs_cap;
    if (d.xs_out) {
        // all-to-all: a record goes to the shards that have members in its building; segments of the same size
        // between every pair of shards (their need is exchanged with the status, so they grow alike everywhere)
        for (uint32_t r = 0; r < d.world; ++r) HIP_TRY(c, hipMemsetAsync(d.xs_out + (size_t)r * seg, 0, sizeof(uint32_t), c->stream));
        hipLaunchKernelGGL(k_shared_pack, dim3(256), dim3(TPB), 0, c->stream, d, (uint32_t)c->xf_n, limit_t);
        if ((rc = exchange_alltoall(c, 4, d.xs_out, d.xs, seg))) return rc;
    } else {
        HIP_TRY(c, hipMemsetAsync(d.xs + (size_t)d.rank * seg, 0, sizeof(uint32_t), c->stream));
        if (c->comm_fn)          // (a caller's transport sums the whole buffer: the other ranks' segments must be zero)
            for (uint32_t r = 0; r < d.world; ++r) if (r != d.rank) HIP_TRY(c, hipMemsetAsync(d.xs + (size_t)r * seg, 0, sizeof(uint32_t) * seg, c->stream));
        hipLaunchKernelGGL(k_shared_pack, dim3(256), dim3(TPB), 0, c->stream, d, (uint32_t)c->xf_n, limit_t);
        if ((rc = exchange_gather(c, 4, d.xs, seg))) return rc;
    }
    hipLaunchKernelGGL(k_shard_prep, dim3(1), dim3(128), 0, c->stream, d, (uint32_t)c->xf_n, limit_t);
    if ((rc = exchange(c, 2))) return rc;
    hipLaunchKernelGGL(k_decide, dim3(1), dim3(64), 0, c->stream, d, (uint32_t)c->xf_n, limit_t, 1, 1);
    hipLaunchKernelGGL(k_chunk_marks, dim3(c->grid_chunk), dim3(TPB), 0, c->stream, d);
    hipLaunchKernelGGL(k_chunk_fold<false>, dim3(c->grid_chunk), dim3(TPB), 0, c->stream, d);
    hipLaunchKernelGGL(k_chunk_draw<false>, dim3(c->grid_chunk * c->draw_mult), dim3(TPB), 0, c->stream, d, c->grid_chunk * (TPB / 64u));
    hipLaunchKernelGGL(k_chunk_units<false>, dim3(c->grid_chunk * c->units_mult), dim3(TPB), 0, c->stream, d);
    if (vax && c->vax_repair && (c->repair_armed || c->vax_repair_always)) {
        // the repair of the plan (DESIGN.md 3.13 v), sharded: the shards agree on the step to walk again from (buffer L), exchange
        // the liveness of the candidates as it truly stood (buffer V a second time) and walk the same steps again
        hipLaunchKernelGGL(k_chunk_lost, dim3(1), dim3(FIN_TPB), 0, c->stream, d);
        if ((rc = exchange_buf(c, 10, d.xl, FREE_MAX + 2u))) return rc;
        hipLaunchKernelGGL(k_lost_global, dim3(1), dim3(64), 0, c->stream, d);
        hipLaunchKernelGGL(k_vax_live<true>, dim3(PLAN_W / TPB, FREE_MAX), dim3(TPB), 0, c->stream, d, (uint32_t)c->xf_n, limit_t);
        if ((rc = exchange_buf(c, 3, d.xv, XV_HEADER + (size_t)FREE_MAX * (PLAN_W / 32u)))) return rc;
        hipLaunchKernelGGL(k_chunk_vax<true>, dim3(FREE_MAX), dim3(FIN_TPB), 0, c->stream, d, (uint32_t)c->xf_n, limit_t, 1);
    }
    hipLaunchKernelGGL(k_chunk_count, dim3(COUNT_GRID), dim3(TPB), 0, c->stream, d);
    if (vax && (rc = exchange_buf(c, 5, d.xc, FREE_MAX + 2u))) return rc;
    hipLaunchKernelGGL(k_chunk_books, dim3(1), dim3(FIN_TPB), 0, c->stream, d, 0, 0, (uint32_t)c->xf_n, limit_t);
    hipLaunchKernelGGL(k_chunk_scatter, dim3(COUNT_GRID), dim3(TPB), 0, c->stream, d);   // (same grid as k_chunk_count: their workgroups pair up)
    hipLaunchKernelGGL(k_chunk_vax_final, dim3(FREE_MAX), dim3(TPB), 0, c->stream, d);
    HIP_TRY(c, hipGetLastError());
    return ESIM_OK;
}

// n coupled steps: three device phases around the two per-step exchanges (the form every step can take)
int run_coupled_steps(esim_ctx_impl *c, uint32_t n, bool ex)
{
    int rc;
    for (uint32_t s = 0; s < n; ++s) {
        const bool tk = want_kernel_timing(c);
        if ((rc = enqueue_begin(c, tk))) return rc;
        if (ex && (rc = exchange(c, 0))) return rc;
        if ((rc = enqueue_exposures(c, tk))) return rc;
        if (ex && (rc = exchange(c, 1))) return rc;
        if ((rc = enqueue_finish(c, tk))) return rc;
        if ((s & 255u) == 255u && (rc = wait_stream(c))) return rc;
    }
    c->shard_step_steps += n;
    return ESIM_OK;
}

}  // namespace

// Simulator::simulate's loop for one shard of a sharded population.  Steps run as time-parallel chunks with one round of
// exchanges per chunk wherever a chunk can run on every shard, and as coupled steps (three device phases around two exchanges
// per step) otherwise: the step that starts the vaccination programme, chunks that do not fit the one-pass form somewhere, plans
// that need more candidates than the exchanged window.  Every rank takes the same decisions from the same reduced words.
// Over RCCL nothing waits for the host inside a burst of chunks.
extern "C" int esim_run_sharded(esim_ctx *ctx, uint32_t n_steps, uint32_t *n_done)
{
    esim_ctx_impl *c = CTX(ctx);
    int rc = check_budget(c, n_steps);
    if (rc) return rc;
    if (c->d.n_shards > 1 && !c->nccl && !c->comm_fn) return fail(c, ESIM_ESTATE, "esim_run_sharded: no communicator (esim_comm_init_rccl / esim_comm_init_callback)");
    HIP_TRY(c, hipSetDevice(c->P.device));
    Dev &d = c->d;
    const uint32_t first = c->host_t;
    // no early stop here (a shard's local census says nothing about the disease elsewhere, and shards that stopped at
    // different steps would issue different collectives): a flag an earlier esim_run left on the device is cleared
    static const uint32_t zero = 0u;
    HIP_TRY(c, hipMemcpyAsync(&d.ctrl->stop_when_done, &zero, sizeof zero, hipMemcpyHostToDevice, c->stream));
    if (c->pmap_used) {                              // (a persistent map left by esim_run: the chunk pass of sharded runs builds its own per chunk)
        hipLaunchKernelGGL(k_map_clear, dim3(COUNT_GRID), dim3(TPB), 0, c->stream, d);
        hipLaunchKernelGGL(k_map_reset, dim3(1), dim3(64), 0, c->stream, d);
        (void)hipMemsetAsync(d.sch_ring, 0, sizeof(uint32_t) * (size_t)(d.n_sch ? d.n_sch : 1) * 2u * SCH_RING, c->stream);
        c->pmap_used = false;
    }
    c->map_valid = false;
    // (a communicator on an unsharded context -- one rank -- still makes its collectives: the sums over one rank change nothing,
    // which is how the RCCL path is exercised on a one-GPU box)
    const bool ex = d.n_shards > 1 || c->nccl || c->comm_fn;
    const bool chunks = ex && d.xs && c->pipeline && c->time_parallel && d.items_cap > 0;
    std::vector<std::pair<uint32_t, uint32_t>> local_ranges;     // [first step, count) whose records hold this shard's census
    uint32_t remaining = n_steps, stall = 0;
    Ctrl h;
    while (remaining > 0) {
        if (chunks && (!c->elig_seen || c->vax_chunks)) {
            const uint32_t t_first = c->host_t, limit_t = t_first + remaining - 1u;
            const uint32_t bursts = stall ? 1u : std::min<uint32_t>((remaining + (uint32_t)c->xf_n - 1u) / (uint32_t)c->xf_n + (c->elig_seen ? 1u : 0u), 4u);
            for (uint32_t g = 0; g < bursts; ++g) if ((rc = enqueue_sharded_chunk(c, limit_t, c->elig_seen))) return rc;
            // every rank reads the same decision words: steps advanced (all shards run a chunk or none does), the summed error
            // fields, the gathered segment need
            if ((rc = sync_status(c, ex, &h))) return rc;
            const uint32_t done = h.t - t_first;
            c->host_t = h.t; remaining -= done;
            c->shard_chunk_steps += done;
            if (h.vax_cuts > c->vax_chunk_cuts) c->repair_armed = true;  // (cuts are decided from summed words: every rank arms in the same burst)
            c->vax_chunk_cuts = h.vax_cuts; c->vax_chunk_repairs = h.vax_repairs;
            // the commuter segment follows the need (the same on every rank: the counts were gathered)
            const uint32_t cap_before = d.xs_cap;
            while (d.xs_cap < XS_CAP_MAX && 2u * h.xs_need_all > d.xs_cap) d.xs_cap *= 2u;     // (xs_need_all: the maximum over the shards, from the status exchange)
            if (done) { local_ranges.emplace_back(t_first, done); stall = 0; continue; }
            if (std::getenv("ESIM_DEBUG"))
                std::fprintf(stderr, "[esim] rank %d: sharded chunk without progress at t=%u: chunk_ok=%u parallel=%u vax_chunk=%u vax_fail=%u cannot=%u xs_need=%u xs_cap=%u pairs=%u\n",
                             c->comm_rank, h.t, h.chunk_ok, h.chunk_parallel, h.vax_chunk, h.vax_fail, 0u, h.xs_need_all, cap_before, h.chunk_pairs);
            if (d.xs_cap != cap_before) continue;                        // the segment was too short: again with the longer one
            stall = std::min<uint32_t>(stall + 1u, 8u);
        }
        const uint32_t k = std::min<uint32_t>(remaining, (!chunks || (c->elig_seen && !c->vax_chunks)) ? remaining : (stall <= 1u ? 1u : (stall <= 3u ? 8u : (uint32_t)c->xf_n)));
        if ((rc = run_coupled_steps(c, k, ex))) return rc;
        remaining -= k;
        if ((rc = sync_status(c, ex, &h))) return rc;
        c->elig_seen = h.have_elig != 0u;
    }
    // the records of the steps drawn as chunks: this shard's census -> everybody's
    for (auto &rg : local_ranges) {
        const size_t n = (size_t)rg.second * XR_FIELDS;
        if (n > c->xr_n) {
            if ((rc = wait_stream(c))) return rc;                        // (the buffer being replaced may still be in use)
            if (c->xr) dev_free(c, c->xr);
            c->xr = nullptr; c->xr_n = 0;
            if ((rc = dev_alloc(c, &c->xr, n))) return rc;
            c->xr_n = n;
        }
        hipLaunchKernelGGL(k_records_pack, dim3(grid_for(rg.second, TPB, 64)), dim3(TPB), 0, c->stream, d, rg.first, rg.second, c->xr);
        if (ex && (rc = exchange_buf(c, 6, c->xr, n))) return rc;
        hipLaunchKernelGGL(k_records_unpack, dim3(grid_for(rg.second, TPB, 64)), dim3(TPB), 0, c->stream, d, rg.first, rg.second, c->xr);
    }
    if ((rc = wait_stream(c))) return rc;
    if (n_done) *n_done = c->host_t - first;
    return ESIM_OK;
}

extern "C" int esim_shard_stats(esim_ctx *ctx, uint64_t *chunk_steps, uint64_t *coupled_steps)
{
    esim_ctx_impl *c = CTX(ctx);
    if (!c) return ESIM_EINVAL;
    if (chunk_steps) *chunk_steps = c->shard_chunk_steps;
    if (coupled_steps) *coupled_steps = c->shard_step_steps;
    return ESIM_OK;
}

extern "C" int esim_future_infected(esim_ctx *ctx)
{
    esim_ctx_impl *c = CTX(ctx);
    if (!c || !c->uploaded) return fail(c, ESIM_ESTATE, "no population uploaded");
    HIP_TRY(c, hipSetDevice(c->P.device));
    hipLaunchKernelGGL(k_future, dim3(1), dim3(FIN_TPB), 0, c->stream, c->d, (uint32_t)c->xf_n, c->free_limit ? c->free_limit : c->P.max_steps);
    HIP_TRY(c, hipGetLastError());
    return ESIM_OK;
}

// A burst of decoupled chunks without a host round trip per chunk: the caller repeats
// { esim_future_infected; all-reduce F; esim_free_enqueue } and then collects once.
extern "C" int esim_free_begin(esim_ctx *ctx, uint32_t n_steps)
{
    esim_ctx_impl *c = CTX(ctx);
    int rc = check_budget(c, n_steps);
    if (rc) return rc;
    if (n_steps == 0) return fail(c, ESIM_EINVAL, "esim_free_begin: no steps");
    if (c->d.n_shared_bld || c->d.n_shared_room) return fail(c, ESIM_ESTATE, "esim_free_begin: shards that share buildings need the coupled steps");
    if (!c->pipeline || !c->time_parallel) return fail(c, ESIM_ESTATE, "esim_free_begin: needs pipeline level 2");
    c->free_limit = c->host_t + n_steps - 1u;
    c->free_first = c->host_t;
    return ESIM_OK;
}

extern "C" int esim_free_enqueue(esim_ctx *ctx)
{
    esim_ctx_impl *c = CTX(ctx);
    if (!c || !c->uploaded || !c->free_limit) return fail(c, ESIM_ESTATE, "esim_free_enqueue: no burst open (esim_free_begin)");
    HIP_TRY(c, hipSetDevice(c->P.device));
    bool tk = c->kernel_timing;
    if (tk && c->fev_used + 2 > c->fev.size())
        for (int i = 0; i < 2 && tk; ++i) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) tk = false; else c->fev.push_back(e); }
    if (tk) HIP_TRY(c, hipEventRecord(c->fev[c->fev_used], c->stream));
    hipLaunchKernelGGL(k_decide, dim3(1), dim3(64), 0, c->stream, c->d, (uint32_t)c->xf_n, c->free_limit, 1, 0);
    enqueue_parallel_chunk(c, 2, c->free_limit);                  // leaves the census ahead of the NEXT chunk in buffer F
    if (tk) { HIP_TRY(c, hipEventRecord(c->fev[c->fev_used + 1], c->stream)); c->fev_used += 2; }
    HIP_TRY(c, hipGetLastError());
    return ESIM_OK;
}

extern "C" int esim_free_collect(esim_ctx *ctx, uint32_t *n_done)
{
    esim_ctx_impl *c = CTX(ctx);
    if (!c || !c->uploaded || !c->free_limit) return fail(c, ESIM_ESTATE, "esim_free_collect: no burst open (esim_free_begin)");
    HIP_TRY(c, hipSetDevice(c->P.device));
    c->free_limit = 0;
    Ctrl h;
    HIP_TRY(c, hipMemcpyAsync(&h, c->d.ctrl, sizeof h, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (h.error) return fail(c, -(int)h.error, "device-side error (raised at check " + std::to_string(h.err_where) + ", esim_device.h ERR_AT_*)");
    const uint32_t done = h.t - c->free_first;
    c->last_chunk_pairs = h.chunk_pairs;
    // device time of the chunks of the burst (k_future and the collective in front of each are not inside the pairs)
    for (size_t i = 0; i + 1 < c->fev_used; i += 2) { float ms; HIP_TRY(c, hipEventElapsedTime(&ms, c->fev[i], c->fev[i + 1])); if (done) c->chunk_ms += ms; }
    c->fev_used = 0;
    c->chunk_steps += done;
    c->chunk_count += (done + (uint32_t)c->xf_n - 1u) / (uint32_t)c->xf_n;
    c->host_t = h.t;
    if (n_done) *n_done = done;
    return ESIM_OK;
}

extern "C" int esim_run_free(esim_ctx *ctx, uint32_t n_steps, uint32_t *n_done)
{
    esim_ctx_impl *c = CTX(ctx);
    int rc = check_budget(c, n_steps);
    if (rc) return rc;
    if (n_steps > c->xf_n) return fail(c, ESIM_EINVAL, "esim_run_free: more steps than the future vector covers");
    if (c->d.n_shared_bld || c->d.n_shared_room) return fail(c, ESIM_ESTATE, "esim_run_free: shards that share buildings need the coupled steps");
    HIP_TRY(c, hipSetDevice(c->P.device));
    uint32_t done = 0;
    if ((rc = run_chunk(c, n_steps, &done, nullptr))) return rc;
    c->host_t += done;
    if (n_done) *n_done = done;
    return ESIM_OK;
}

extern "C" int esim_set_pipeline(esim_ctx *ctx, int enable)
{
    esim_ctx_impl *c = CTX(ctx);
    if (!c) return ESIM_EINVAL;
    c->pipeline = enable != 0;            // 0: sequential steps only
    c->time_parallel = enable >= 2;       // 1: one kernel per step (k_pipe); 2: all steps of a chunk in one pass
    c->vax_chunks = enable >= 3;          // 3 (default): ... also while a vaccination programme runs, its vaccinations planned per chunk
    if (enable >= 4 && !c->pmap) { c->pmap = true; c->map_valid = false; }   // 4: ... on the persistent item map (unsharded contexts)
    if (enable < 4 && c->pmap && !std::getenv("ESIM_PMAP")) {                // back to the per-chunk map: whatever the persistent one holds is emptied first
        if (c->pmap_used && c->uploaded) {
            hipLaunchKernelGGL(k_map_clear, dim3(COUNT_GRID), dim3(TPB), 0, c->stream, c->d);
            hipLaunchKernelGGL(k_map_reset, dim3(1), dim3(64), 0, c->stream, c->d);
            (void)hipMemsetAsync(c->d.sch_ring, 0, sizeof(uint32_t) * (size_t)(c->d.n_sch ? c->d.n_sch : 1) * 2u * SCH_RING, c->stream);
            c->pmap_used = false;
        }
        c->pmap = false; c->map_valid = false;
    }
    return ESIM_OK;
}

extern "C" int esim_vax_chunk_stats(esim_ctx *ctx, uint64_t *steps, uint64_t *cuts)
{
    esim_ctx_impl *c = CTX(ctx);
    if (!c) return ESIM_EINVAL;
    if (steps) *steps = c->vax_chunk_steps;
    if (cuts) *cuts = c->vax_chunk_cuts;
    return ESIM_OK;
}

extern "C" int esim_vax_repair_stats(esim_ctx *ctx, uint64_t *repairs)
{
    esim_ctx_impl *c = CTX(ctx);
    if (!c) return ESIM_EINVAL;
    if (repairs) *repairs = c->vax_chunk_repairs;
    return ESIM_OK;
}

extern "C" int esim_chunk_timing(esim_ctx *ctx, double *total_ms, uint64_t *steps, uint64_t *chunks)
{
    esim_ctx_impl *c = CTX(ctx);
    if (!c) return ESIM_EINVAL;
    if (total_ms) *total_ms = c->chunk_ms;
    if (steps) *steps = c->chunk_steps;
    if (chunks) *chunks = c->chunk_count;
    c->chunk_ms = 0; c->chunk_steps = 0; c->chunk_count = 0;
    return ESIM_OK;
}

extern "C" int esim_enable_chunk_kernel_timing(esim_ctx *ctx, int enable)
{
    esim_ctx_impl *c = CTX(ctx);
    if (!c) return ESIM_EINVAL;
    HIP_TRY(c, hipSetDevice(c->P.device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->kdetail = enable != 0;
    c->kd_used = 0;
    for (int i = 0; i < ESIM_CK_N; ++i) { c->kd_ms[i] = 0; c->kd_calls[i] = 0; }
    return ESIM_OK;
}

extern "C" int esim_chunk_kernel_timings(esim_ctx *ctx, double ms[ESIM_CK_N], uint64_t calls[ESIM_CK_N])
{
    esim_ctx_impl *c = CTX(ctx);
    if (!c || !ms || !calls) return ESIM_EINVAL;
    HIP_TRY(c, hipSetDevice(c->P.device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    kd_resolve(c);
    for (int i = 0; i < ESIM_CK_N; ++i) { ms[i] = c->kd_ms[i]; calls[i] = c->kd_calls[i]; c->kd_ms[i] = 0; c->kd_calls[i] = 0; }
    return ESIM_OK;
}

extern "C" int esim_pipeline_timing(esim_ctx *ctx, double *mean_step_ms, uint64_t *steps_timed, uint64_t *steps_run)
{
    esim_ctx_impl *c = CTX(ctx);
    if (!c) return ESIM_EINVAL;
    HIP_TRY(c, hipSetDevice(c->P.device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    double acc = 0;
    const size_t n = c->pkev_used / 2;
    for (size_t i = 0; i < n; ++i) { float ms; HIP_TRY(c, hipEventElapsedTime(&ms, c->pkev[2 * i], c->pkev[2 * i + 1])); acc += ms; }
    if (mean_step_ms) *mean_step_ms = n ? acc / (double)n : 0.0;
    if (steps_timed) *steps_timed = n;
    if (steps_run) *steps_run = c->pipe_steps;
    c->pkev_used = 0; c->pipe_steps = 0;
    return ESIM_OK;
}

extern "C" int esim_debug_counters(esim_ctx *ctx, uint32_t out[16])
{
    esim_ctx_impl *c = CTX(ctx);
    if (!c || !c->uploaded || !out) return ESIM_EINVAL;
    HIP_TRY(c, hipSetDevice(c->P.device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    Ctrl h;
    HIP_TRY(c, hipMemcpy(&h, c->d.ctrl, sizeof h, hipMemcpyDeviceToHost));
    const uint32_t v[16] = { h.t, h.chunk_ok, h.chunk_parallel, h.chunk_pairs, h.n_items, h.items_per_wave, h.n_units, h.n_route_pairs,
                             h.n_route_pairs_big, h.n_newexp, h.log_len, h.n_susceptible, h.lockdown, h.mask, h.at_work, h.bus_dir };
    std::memcpy(out, v, sizeof v);
    return ESIM_OK;
}

#ifdef ESIM_COUNT_WORK
// counting build only (not in include/esim.h): what the chunk pass worked on since the last call (WK_* in esim_kernels_common.h)
extern "C" int esim_work_counters(esim_ctx *ctx, unsigned long long *out, uint32_t n)
{
    esim_ctx_impl *c = CTX(ctx);
    if (!c || !c->uploaded || !out) return ESIM_EINVAL;
    HIP_TRY(c, hipSetDevice(c->P.device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    unsigned long long h[WK_N];
    HIP_TRY(c, hipMemcpy(h, c->d.work_cnt, sizeof h, hipMemcpyDeviceToHost));
    HIP_TRY(c, hipMemset(c->d.work_cnt, 0, sizeof h));
    for (uint32_t i = 0; i < n; ++i) out[i] = i < WK_N ? h[i] : 0ull;
    return ESIM_OK;
}
#endif

#ifdef ESIM_WAVE_PROFILE
// diagnostics build only (not in include/esim.h): rows of per-wavefront timers, and the timer's rate in kHz
extern "C" int esim_prof_read(esim_ctx *ctx, uint32_t *out, uint32_t n_words, int *clock_khz)
{
    esim_ctx_impl *c = CTX(ctx);
    if (!c || !c->uploaded || !out) return ESIM_EINVAL;
    HIP_TRY(c, hipSetDevice(c->P.device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpy(out, c->d.prof_buf, sizeof(uint32_t) * std::min<uint32_t>(n_words, 16384u * 16u), hipMemcpyDeviceToHost));
    if (clock_khz) HIP_TRY(c, hipDeviceGetAttribute(clock_khz, hipDeviceAttributeWallClockRate, c->P.device));
    return ESIM_OK;
}
#endif

extern "C" int esim_set_small_step_limit(esim_ctx *ctx, uint32_t max_infected)
{
    esim_ctx_impl *c = CTX(ctx);
    if (!c) return ESIM_EINVAL;
    c->small_max = max_infected;
    return ESIM_OK;
}

extern "C" int esim_set_tiny_chunk_limit(esim_ctx *ctx, uint32_t max_pairs)
{
    esim_ctx_impl *c = CTX(ctx);
    if (!c) return ESIM_EINVAL;
    c->tiny_pairs = max_pairs;
    return ESIM_OK;
}

extern "C" int esim_exchange_buffer(esim_ctx *ctx, int which, void **device_ptr, size_t *n_u32)
{
    esim_ctx_impl *c = CTX(ctx);
    if (!c || !c->uploaded) return fail(c, ESIM_ESTATE, "no population uploaded");
    if (which < 0 || which > 2) return fail(c, ESIM_EINVAL, "esim_exchange_buffer: which must be 0, 1 or 2");
    if (device_ptr) *device_ptr = which == 2 ? (void *)c->d.xf : which ? (void *)c->d.xb : (void *)c->d.xa;
    if (n_u32) *n_u32 = which == 2 ? c->xf_n + 1 : which ? c->xb_n : c->xa_n;
    return ESIM_OK;
}

extern "C" int esim_read_records(esim_ctx *ctx, uint32_t first_step, uint32_t n, esim_step_result *out)
{
    esim_ctx_impl *c = CTX(ctx);
    if (!c || !c->uploaded) return fail(c, ESIM_ESTATE, "no population uploaded");
    if (!out || first_step == 0 || (uint64_t)first_step + n > (uint64_t)c->P.max_steps + 1) return fail(c, ESIM_EINVAL, "esim_read_records: bad range");
    HIP_TRY(c, hipSetDevice(c->P.device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpy(out, &c->d.records[first_step], sizeof(esim_step_result) * n, hipMemcpyDeviceToHost));
    return device_error(c);
}

extern "C" int esim_stream(esim_ctx *ctx, void **stream)
{
    esim_ctx_impl *c = CTX(ctx);
    if (!c || !stream) return fail(c, ESIM_EINVAL, "esim_stream: null argument");
    *stream = (void *)c->stream;
    return ESIM_OK;
}

extern "C" int esim_set_stream(esim_ctx *ctx, void *stream)
{
    esim_ctx_impl *c = CTX(ctx);
    if (!c) return ESIM_EINVAL;
    HIP_TRY(c, hipSetDevice(c->P.device));
    if (c->stream) HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->stream && c->own_stream) (void)hipStreamDestroy(c->stream);
    c->stream = (hipStream_t)stream;
    c->own_stream = false;
    return ESIM_OK;
}

extern "C" int esim_set_exchange_buffer(esim_ctx *ctx, int which, void *device_ptr)
{
    esim_ctx_impl *c = CTX(ctx);
    if (!c || !c->uploaded) return fail(c, ESIM_ESTATE, "no population uploaded");
    if (which < 0 || which > 2 || !device_ptr) return fail(c, ESIM_EINVAL, "esim_set_exchange_buffer: bad argument");
    HIP_TRY(c, hipSetDevice(c->P.device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (which == 2) c->d.xf = (uint32_t *)device_ptr; else if (which) c->d.xb = (uint32_t *)device_ptr; else c->d.xa = (uint32_t *)device_ptr;
    return ESIM_OK;
}

extern "C" int esim_synchronize(esim_ctx *ctx)
{
    esim_ctx_impl *c = CTX(ctx);
    if (!c) return ESIM_EINVAL;
    HIP_TRY(c, hipSetDevice(c->P.device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return ESIM_OK;
}

extern "C" int esim_download_state(esim_ctx *ctx, uint8_t *status, uint16_t *timer, uint32_t *current_building,
                                   uint8_t *on_bus, uint8_t *eligible)
{
    esim_ctx_impl *c = CTX(ctx);
    if (!c || !c->uploaded) return fail(c, ESIM_ESTATE, "no population uploaded");
    HIP_TRY(c, hipSetDevice(c->P.device));
    const uint32_t N = c->d.n;
    uint8_t *d_status = nullptr, *d_bus = nullptr, *d_elig = nullptr; uint16_t *d_timer = nullptr; uint32_t *d_cur = nullptr;
    auto cleanup = [&]() { (void)hipFree(d_status); (void)hipFree(d_bus); (void)hipFree(d_elig); (void)hipFree(d_timer); (void)hipFree(d_cur); };
    const size_t n1 = N ? N : 1;
    if ((status && hipMalloc(&d_status, n1) != hipSuccess) || (on_bus && hipMalloc(&d_bus, n1) != hipSuccess) ||
        (eligible && hipMalloc(&d_elig, n1) != hipSuccess) || (timer && hipMalloc(&d_timer, 2 * n1) != hipSuccess) ||
        (current_building && hipMalloc(&d_cur, 4 * n1) != hipSuccess)) { cleanup(); return fail(c, ESIM_ENOMEM, "esim_download_state: hipMalloc"); }
    hipLaunchKernelGGL(k_decode_state, dim3(c->grid_citizens), dim3(TPB), 0, c->stream, c->d, d_status, d_timer, d_cur, d_bus, d_elig);
    hipError_t e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess && status) e = hipMemcpy(status, d_status, N, hipMemcpyDeviceToHost);
    if (e == hipSuccess && on_bus) e = hipMemcpy(on_bus, d_bus, N, hipMemcpyDeviceToHost);
    if (e == hipSuccess && eligible) e = hipMemcpy(eligible, d_elig, N, hipMemcpyDeviceToHost);
    if (e == hipSuccess && timer) e = hipMemcpy(timer, d_timer, 2 * (size_t)N, hipMemcpyDeviceToHost);
    if (e == hipSuccess && current_building) e = hipMemcpy(current_building, d_cur, 4 * (size_t)N, hipMemcpyDeviceToHost);
    cleanup();
    if (e != hipSuccess) return fail(c, ESIM_ENODEVICE, std::string("esim_download_state: ") + hipGetErrorString(e));
    return ESIM_OK;
}

extern "C" int esim_download_exposure_log(esim_ctx *ctx, uint32_t *citizen, uint32_t *step, uint8_t *on_bus, uint32_t cap, uint32_t *n_out)
{
    esim_ctx_impl *c = CTX(ctx);
    if (!c || !c->uploaded || !n_out) return fail(c, ESIM_ESTATE, "no population uploaded");
    HIP_TRY(c, hipSetDevice(c->P.device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    Ctrl h;
    HIP_TRY(c, hipMemcpy(&h, c->d.ctrl, sizeof h, hipMemcpyDeviceToHost));
    const uint32_t t_done = c->host_t - 1u;                       // steps run so far
    // log_off[TE_BIAS + s] = first entry of step s; the entries before step 1 are the seeds (simulator_builder.rs:1268-1287)
    std::vector<uint32_t> off((size_t)t_done + 2u);
    HIP_TRY(c, hipMemcpy(off.data(), c->d.log_off + TE_BIAS + 1u, sizeof(uint32_t) * (t_done + 1u), hipMemcpyDeviceToHost));
    off[t_done + 1u] = h.log_len;
    const uint32_t first = t_done ? off[0] : h.log_len, n = h.log_len - first;
    *n_out = n;
    if (n > cap || (n && (!citizen || !step || !on_bus))) return fail(c, ESIM_ERANGE, "esim_download_exposure_log: buffers too small (n_out holds the size needed)");
    if (n == 0) return ESIM_OK;
    uint32_t *d_c = nullptr; uint8_t *d_b = nullptr;
    if (hipMalloc(&d_c, sizeof(uint32_t) * (size_t)n) != hipSuccess || hipMalloc(&d_b, n) != hipSuccess) { (void)hipFree(d_c); return fail(c, ESIM_ENOMEM, "esim_download_exposure_log: hipMalloc"); }
    hipLaunchKernelGGL(k_export_log, dim3(grid_for(n, TPB, 2048)), dim3(TPB), 0, c->stream, c->d, first, n, d_c, d_b);
    hipError_t e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipMemcpy(citizen, d_c, sizeof(uint32_t) * (size_t)n, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(on_bus, d_b, n, hipMemcpyDeviceToHost);
    (void)hipFree(d_c); (void)hipFree(d_b);
    if (e != hipSuccess) return fail(c, ESIM_ENODEVICE, std::string("esim_download_exposure_log: ") + hipGetErrorString(e));
    for (uint32_t s = 1; s <= t_done; ++s)
        for (uint32_t i = off[s - 1u]; i < off[s] && i - first < n; ++i) step[i - first] = s;
    return ESIM_OK;
}

// ---- checkpoint / restore: everything a step reads that is not part of the uploaded population ----------------------
namespace {
struct CkptHeader {
    uint32_t magic, version, n, n_global, id_base, max_steps, host_t, log_len;
    uint32_t exposed_time, infected_time, vaccination_rate, bus_capacity, start_hour, end_hour, ctrl_bytes, layout_id;
    uint64_t seed;
    uint64_t pop_hash;
    double thresholds[6];
};
const uint32_t CKPT_MAGIC = 0x4D495345u /* "ESIM" */, CKPT_VERSION = 3u;

// What a checkpoint's bytes mean depends on how this build lays the state out: the citizen word's fields, the exposure-step
// bias and sentinels, the control block's fields.  The header carries a hash of all of that; a checkpoint written by a build
// with another layout (an older library, a diagnostics build that moved a field) is refused instead of reinterpreted.
constexpr uint32_t layout_mix(uint32_t h, uint32_t v) { return (h ^ v) * 16777619u; }
constexpr uint32_t ckpt_layout_id()
{
    uint32_t h = 2166136261u;
    const uint32_t parts[] = {
        CKPT_VERSION, (uint32_t)sizeof(Ctrl), (uint32_t)sizeof(esim_step_result), (uint32_t)sizeof(Decision),
        (uint32_t)offsetof(Ctrl, t), (uint32_t)offsetof(Ctrl, lockdown), (uint32_t)offsetof(Ctrl, mask), (uint32_t)offsetof(Ctrl, vacc_active),
        (uint32_t)offsetof(Ctrl, have_elig), (uint32_t)offsetof(Ctrl, trigger_step), (uint32_t)offsetof(Ctrl, elig_count), (uint32_t)offsetof(Ctrl, at_work),
        (uint32_t)offsetof(Ctrl, bus_dir), (uint32_t)offsetof(Ctrl, steps_done), (uint32_t)offsetof(Ctrl, error), (uint32_t)offsetof(Ctrl, n_susceptible),
        (uint32_t)offsetof(Ctrl, n_vaccinated), (uint32_t)offsetof(Ctrl, log_len), (uint32_t)offsetof(Ctrl, chunk_pairs), (uint32_t)offsetof(Ctrl, peer_error),
        CW_TE_SHIFT, CW_BUS_EXPOSED, CW_FLAGS, CW_VAX_SHIFT, CW_VAX_MASK, CW_PLAN_SKIP, TE_SUSCEPTIBLE, TE_VACCINATED, TE_RECOVERED, TE_BIAS, TE_SLOTS,
        FL_USES_PT, FL_MASK_COMPLIANT, FL_SAME_AREA, FL_WORK_SCHOOL, FL_HAS_WORK, FL_BIG_ROUTE, MARK_SLOTS, FREE_MAX };
    for (uint32_t v : parts) h = layout_mix(h, v);
    return h;
}

void ckpt_header(const esim_ctx_impl *c, const Ctrl &h, CkptHeader *o)
{
    std::memset(o, 0, sizeof *o);
    o->magic = CKPT_MAGIC; o->version = CKPT_VERSION; o->n = c->d.n; o->n_global = c->d.n_global; o->id_base = c->d.id_base;
    o->max_steps = c->P.max_steps; o->host_t = c->host_t; o->log_len = h.log_len;
    o->exposed_time = c->P.exposed_time; o->infected_time = c->P.infected_time; o->vaccination_rate = c->P.vaccination_rate;
    o->bus_capacity = c->P.bus_capacity; o->start_hour = c->P.start_hour; o->end_hour = c->P.end_hour; o->ctrl_bytes = (uint32_t)sizeof(Ctrl); o->layout_id = ckpt_layout_id();
    o->seed = c->P.seed; o->pop_hash = c->pop_hash;
    const double th[6] = { c->P.exposure_chance, c->P.mask_effectiveness, c->P.lockdown_threshold, c->P.vaccination_threshold,
                           c->P.mask_pt_threshold, c->P.mask_everywhere_threshold };
    std::memcpy(o->thresholds, th, sizeof th);
}

size_t ckpt_bytes(const CkptHeader &k)
{
    return sizeof(CkptHeader) + k.ctrl_bytes + sizeof(uint32_t) * ((size_t)TE_SLOTS + TE_SLOTS + 1 + k.n + k.log_len + 2u * ((size_t)k.host_t + 1u)) +
           sizeof(esim_step_result) * (size_t)k.host_t;
}
}  // namespace

extern "C" int esim_checkpoint_size(esim_ctx *ctx, size_t *bytes)
{
    esim_ctx_impl *c = CTX(ctx);
    if (!c || !c->uploaded || !bytes) return fail(c, ESIM_ESTATE, "no population uploaded");
    HIP_TRY(c, hipSetDevice(c->P.device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    Ctrl h;
    HIP_TRY(c, hipMemcpy(&h, c->d.ctrl, sizeof h, hipMemcpyDeviceToHost));
    CkptHeader k;
    ckpt_header(c, h, &k);
    *bytes = ckpt_bytes(k);
    return ESIM_OK;
}

extern "C" int esim_checkpoint_save(esim_ctx *ctx, void *buf, size_t cap)
{
    esim_ctx_impl *c = CTX(ctx);
    if (!c || !c->uploaded || !buf) return fail(c, ESIM_ESTATE, "no population uploaded");
    if (c->free_limit) return fail(c, ESIM_ESTATE, "esim_checkpoint_save: a burst of decoupled chunks is open");
    HIP_TRY(c, hipSetDevice(c->P.device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    const Dev &d = c->d;
    Ctrl h;
    HIP_TRY(c, hipMemcpy(&h, d.ctrl, sizeof h, hipMemcpyDeviceToHost));
    if (h.error) return fail(c, -(int)h.error, "esim_checkpoint_save: the context is in a device-side error state");
    CkptHeader k;
    ckpt_header(c, h, &k);
    if (cap < ckpt_bytes(k)) return fail(c, ESIM_ERANGE, "esim_checkpoint_save: buffer smaller than esim_checkpoint_size");
    uint8_t *p = (uint8_t *)buf;
    std::memcpy(p, &k, sizeof k); p += sizeof k;
    std::memcpy(p, &h, sizeof h); p += sizeof h;
    auto pull = [&](const void *src, size_t bytes) -> int { if (bytes) HIP_TRY(c, hipMemcpy(p, src, bytes, hipMemcpyDeviceToHost)); p += bytes; return ESIM_OK; };
    int rc;
    if ((rc = pull(d.hist, sizeof(uint32_t) * TE_SLOTS))) return rc;
    if ((rc = pull(d.log_off, sizeof(uint32_t) * (TE_SLOTS + 1)))) return rc;
    if ((rc = pull(d.cit, sizeof(uint32_t) * (size_t)d.n))) return rc;
    if ((rc = pull(d.log, sizeof(uint32_t) * (size_t)h.log_len))) return rc;
    if ((rc = pull(d.exp_step, sizeof(uint32_t) * 2u * ((size_t)c->host_t + 1u)))) return rc;
    if ((rc = pull(d.records, sizeof(esim_step_result) * (size_t)c->host_t))) return rc;
    return ESIM_OK;
}

extern "C" int esim_checkpoint_restore(esim_ctx *ctx, const void *buf, size_t bytes)
{
    esim_ctx_impl *c = CTX(ctx);
    if (!c || !c->uploaded || !buf) return fail(c, ESIM_ESTATE, "no population uploaded");
    if (bytes < sizeof(CkptHeader)) return fail(c, ESIM_EINVAL, "esim_checkpoint_restore: not a checkpoint");
    CkptHeader k, mine;
    std::memcpy(&k, buf, sizeof k);
    Ctrl zero;
    std::memset(&zero, 0, sizeof zero);
    ckpt_header(c, zero, &mine);
    if (k.magic != CKPT_MAGIC || k.version != CKPT_VERSION || k.ctrl_bytes != sizeof(Ctrl)) return fail(c, ESIM_EINVAL, "esim_checkpoint_restore: not a checkpoint of this library version");
    if (k.layout_id != ckpt_layout_id()) return fail(c, ESIM_EINVAL, "esim_checkpoint_restore: the checkpoint was written by a build with another state layout (citizen word / control block); it is refused, not reinterpreted");
    if (k.n != mine.n || k.n_global != mine.n_global || k.id_base != mine.id_base || k.seed != mine.seed || k.exposed_time != mine.exposed_time ||
        k.infected_time != mine.infected_time || k.vaccination_rate != mine.vaccination_rate || k.bus_capacity != mine.bus_capacity ||
        k.start_hour != mine.start_hour || k.end_hour != mine.end_hour || k.pop_hash != mine.pop_hash || std::memcmp(k.thresholds, mine.thresholds, sizeof k.thresholds) != 0)
        return fail(c, ESIM_EINVAL, "esim_checkpoint_restore: the checkpoint was taken with another population, shard or parameter set");
    if (k.host_t == 0 || k.host_t - 1u > c->P.max_steps || k.log_len > k.n) return fail(c, ESIM_EINVAL, "esim_checkpoint_restore: steps beyond this context's max_steps");
    if (bytes < ckpt_bytes(k)) return fail(c, ESIM_EINVAL, "esim_checkpoint_restore: truncated checkpoint");
    int rc = esim_reset(ctx);                                     // clean marks, chunk tables are clean between calls anyway
    if (rc) return rc;
    const Dev &d = c->d;
    const uint8_t *p = (const uint8_t *)buf + sizeof k;
    Ctrl h;
    std::memcpy(&h, p, sizeof h); p += sizeof h;
    // the control block goes to the device as it is: it must be the one of a context at rest at that step
    if (h.t != k.host_t || h.log_len != k.log_len || h.error != 0u || h.steps_done + 1u != k.host_t || h.n_susceptible > k.n || h.n_vaccinated > k.n)
        return fail(c, ESIM_EINVAL, "esim_checkpoint_restore: the control block does not match the checkpoint's header (corrupt file)");
    h.chunk_ok = 0; h.chunk_parallel = 0; h.chunk_done = 0; h.n_items = 0; h.n_newexp = 0; h.n_units = 0; h.unit_next = 0;
    h.n_route_pairs = 0; h.n_route_pairs_big = 0; h.prev_n_items = 0; h.prev_per_wave = 0; h.items_per_wave = 0; h.small_done = 0;
    h.free_base = 0; h.n_riders = 0; h.peer_error = 0;
    h.map_t = 0; h.pmap_chunk = 0; h.prev_pmap = 0; h.n_neg = 0; h.n_cancel = 0; h.map_work = 0;        // (the item map is derived state: the next chunk rebuilds it)
    for (int z = 0; z < 5; ++z) h.counts[z] = 0;
    // marks of the last step are only ever cleared, never read, by the step after it: start without them
    for (uint32_t z = 0; z < MARK_SLOTS; ++z) { h.n_touched_bld[z] = 0; h.n_touched_room[z] = 0; h.n_touched_route[z] = 0; h.n_touched_route_big[z] = 0; }
    auto push = [&](void *dst, size_t nb) -> int { if (nb) HIP_TRY(c, hipMemcpy(dst, p, nb, hipMemcpyHostToDevice)); p += nb; return ESIM_OK; };
    if ((rc = push(d.hist, sizeof(uint32_t) * TE_SLOTS))) return rc;
    if ((rc = push(d.log_off, sizeof(uint32_t) * (TE_SLOTS + 1)))) return rc;
    if ((rc = push(d.cit, sizeof(uint32_t) * (size_t)d.n))) return rc;
    if ((rc = push(d.log, sizeof(uint32_t) * (size_t)k.log_len))) return rc;
    if ((rc = push(d.exp_step, sizeof(uint32_t) * 2u * ((size_t)k.host_t + 1u)))) return rc;
    if ((rc = push(d.records, sizeof(esim_step_result) * (size_t)k.host_t))) return rc;
    HIP_TRY(c, hipMemcpy(d.ctrl, &h, sizeof h, hipMemcpyHostToDevice));
    c->host_t = k.host_t;
    c->last_chunk_pairs = h.chunk_pairs;
    c->elig_seen = h.have_elig != 0u;
    return ESIM_OK;
}

extern "C" int esim_enable_phase_timing(esim_ctx *ctx, int enable)
{
    esim_ctx_impl *c = CTX(ctx);
    if (!c) return ESIM_EINVAL;
    c->phase_timing = enable != 0;
    return ESIM_OK;
}

extern "C" int esim_phase_timings(esim_ctx *ctx, double out[4])
{
    esim_ctx_impl *c = CTX(ctx);
    if (!c || !out) return ESIM_EINVAL;
    out[0] = c->phase_s[0]; out[1] = c->phase_s[1]; out[2] = c->phase_s[2];
    out[3] = out[0] + out[1] + out[2];
    return ESIM_OK;
}

extern "C" int esim_enable_kernel_timing(esim_ctx *ctx, int enable)
{
    esim_ctx_impl *c = CTX(ctx);
    if (!c) return ESIM_EINVAL;
    c->kernel_timing = enable > 0;
    if (enable > 0) c->kernel_timing_stride = (uint32_t)enable;   // time every `enable`-th step
    c->kev_used = 0;
    if (enable > 0) {                                     // (the events the timed runs record: made here, not inside a timed call)
        HIP_TRY(c, hipSetDevice(c->P.device));
        if (!c->cev[0]) { (void)hipEventCreate(&c->cev[0]); (void)hipEventCreate(&c->cev[1]); }
        if (!c->sev[0]) { (void)hipEventCreate(&c->sev[0]); (void)hipEventCreate(&c->sev[1]); }
    }
    return ESIM_OK;
}

extern "C" int esim_small_kernel_timing(esim_ctx *ctx, double *total_ms, uint64_t *steps)
{
    esim_ctx_impl *c = CTX(ctx);
    if (!c) return ESIM_EINVAL;
    if (total_ms) *total_ms = c->small_ms;
    if (steps) *steps = c->small_steps;
    c->small_ms = 0; c->small_steps = 0;
    return ESIM_OK;
}

extern "C" int esim_kernel_timings(esim_ctx *ctx, double *step_ms, uint32_t *out_n)
{
    esim_ctx_impl *c = CTX(ctx);
    if (!c || !step_ms) return ESIM_EINVAL;
    HIP_TRY(c, hipSetDevice(c->P.device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    double acc = 0;
    const size_t n = c->kev_used / 2;
    for (size_t i = 0; i < n; ++i) {
        float ms;
        HIP_TRY(c, hipEventElapsedTime(&ms, c->kev[2 * i], c->kev[2 * i + 1]));
        acc += ms;
    }
    *step_ms = n ? acc / n : 0.0;
    if (out_n) *out_n = (uint32_t)n;
    c->kev_used = 0;
    return ESIM_OK;
}
