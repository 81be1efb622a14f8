/*
 * esim.h -- C ABI of libesim: the MI355X (gfx950) implementation of the reference
 * `sim` crate's per-timestep Citizen update loop.
 *
 * The reference (NoSuchThingAsRandom/EpidemicSimulator) has no FFI or plugin
 * interface: its boundary for this path is the Rust API `Simulator::step` /
 * `Simulator::simulate` (sim/src/simulator.rs:108,131).  Each entry point below names
 * the reference item it replaces.  A Rust shim implementing `Simulator` over this ABI
 * is shown in INTEGRATION.md.
 *
 * Conventions: every function returns ESIM_OK (0) or a negative ESIM_E* code and
 * never unwinds across the boundary; esim_last_error() gives the text.  The caller
 * owns every buffer it passes (the library copies in/out and retains no host
 * pointer).  A context is used from one host thread at a time (the reference
 * `Simulator` is !Send: it owns a ThreadRng, simulator.rs:102).  All compute runs
 * on the GPU; there is no CPU fallback -- without a usable HIP device
 * esim_create() fails with ESIM_ENODEVICE.
 */
#ifndef ESIM_H
#define ESIM_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ESIM_OK          0
#define ESIM_EINVAL     -1   /* bad argument / population contract violated */
#define ESIM_ENODEVICE  -2   /* no HIP device / HIP runtime error */
#define ESIM_ENOMEM     -3
#define ESIM_ESTATE     -4   /* call out of order (e.g. step before upload) */
#define ESIM_ERANGE     -5   /* step budget / encoding range exceeded */
#define ESIM_ESIM       -6   /* the reference's own error path (S underflow, statistics.rs:275-287) */
#define ESIM_ETIMEDOUT  -7   /* sharded run: no progress within the deadline (a peer left or died); the RCCL communicator was aborted */

/* DiseaseStatus codes, sim/src/disease.rs:36-44 */
enum { ESIM_SUSCEPTIBLE = 0, ESIM_EXPOSED = 1, ESIM_INFECTED = 2, ESIM_RECOVERED = 3, ESIM_VACCINATED = 4 };
/* BuildingType, sim/src/models/building.rs:46-53 (only the three the builder creates) */
enum { ESIM_HOUSEHOLD = 0, ESIM_WORKPLACE = 1, ESIM_SCHOOL = 2 };
/* MaskStatus, sim/src/interventions.rs:26-30 */
enum { ESIM_MASK_NONE = 0, ESIM_MASK_PUBLIC_TRANSPORT = 1, ESIM_MASK_EVERYWHERE = 2 };

#define ESIM_NO_ROOM 0xFFFFFFFFu
#define ESIM_FLAG_USES_PUBLIC_TRANSPORT 1u  /* Citizen::uses_public_transport, citizen.rs:132 */
#define ESIM_FLAG_MASK_COMPLIANT        2u  /* Citizen::is_mask_compliant, citizen.rs:131 */

/* Compile-time constants of the reference gathered into one runtime struct:
 * DiseaseModel::covid() (sim/src/disease.rs:118-129), InterventionThresholds and
 * MaskStatus::get_threshold (sim/src/interventions.rs:50-57,71-78), BUS_CAPACITY
 * (sim/src/config.rs:37), start/end_working_hour (sim/src/models/citizen.rs:154-155). */
typedef struct esim_params {
    double   exposure_chance;            /* 0.00055 */
    double   mask_effectiveness;         /* 0.70 */
    double   lockdown_threshold;         /* 0.0034 */
    double   vaccination_threshold;      /* 0.005 */
    double   mask_pt_threshold;          /* 0.001 */
    double   mask_everywhere_threshold;  /* 0.0022 */
    uint32_t exposed_time;               /* 96 */
    uint32_t infected_time;              /* 336 */
    uint32_t vaccination_rate;           /* 1530 */
    uint32_t bus_capacity;               /* 20 */
    uint32_t start_hour;                 /* 9 */
    uint32_t end_hour;                   /* 17 */
    uint64_t seed;                       /* Philox4x32-10 key (replaces thread_rng, simulator.rs:630) */
    int32_t  device;                     /* HIP device ordinal */
    uint32_t max_steps;                  /* capacity of the device-side record log; DiseaseModel::max_time_step = 5000 */
} esim_params;

/* What SimulatorBuilder::build (sim/src/simulator_builder.rs:1162-1292) leaves behind,
 * flattened to borrowed structure-of-arrays.  Citizens are indexed by
 * CitizenID::global_index (citizen.rs:52-57), buildings by a dense index over all
 * Output Areas (BuildingID, building.rs:62-67), rooms by a dense index over all
 * School classes and offices (School::occupant_to_class, building.rs:341).
 * Sharding (multi-GPU): a shard holds the citizens of a contiguous Output Area range;
 * citizen_id_base / n_citizens_global keep the Philox counters global, and the
 * shared_* tables name the buildings / rooms whose members live on several shards. */
typedef struct esim_population {
    uint32_t n_citizens, n_buildings, n_areas, n_rooms, n_seeds;
    uint32_t citizen_id_base;        /* global index of local citizen 0 (0 when unsharded) */
    uint32_t n_citizens_global;      /* == n_citizens when unsharded */
    uint32_t n_shared_buildings;     /* 0 when unsharded */
    uint32_t n_shared_rooms;         /* 0 when unsharded */
    const uint32_t *home_building;   /* [n_citizens] Citizen::household_code, citizen.rs:116 */
    const uint32_t *work_building;   /* [n_citizens] Citizen::workplace_code (== home when none), citizen.rs:118 */
    const uint32_t *room;            /* [n_citizens] class/office of a school member, else ESIM_NO_ROOM */
    const uint8_t  *flags;           /* [n_citizens] ESIM_FLAG_* */
    const uint16_t *age;             /* [n_citizens] or NULL -- carried for API fidelity, never read per step */
    const uint8_t  *occupation;      /* [n_citizens] or NULL -- idem (simulator_builder.rs:296-300) */
    const uint32_t *building_area;   /* [n_buildings] OutputAreaID::index of the building, building.rs:63 */
    const uint8_t  *building_type;   /* [n_buildings] ESIM_HOUSEHOLD/WORKPLACE/SCHOOL */
    const uint32_t *room_building;   /* [n_rooms] the School each room belongs to */
    const uint32_t *seeds;           /* [n_seeds] local indices starting Infected(0), simulator_builder.rs:1111-1140 */
    const int32_t  *shared_building_local; /* [n_shared_buildings] local building index or -1 */
    const int32_t  *shared_room_local;     /* [n_shared_rooms] local room index or -1 */
} esim_population;

/* One StatisticEntry (sim/src/statistics.rs:208-215) plus what the step decided. */
typedef struct esim_step_result {
    uint32_t time_step, susceptible, exposed, infected, recovered, vaccinated;
    uint32_t exposures_building;     /* successful Citizen::expose calls via buildings, simulator.rs:337-345 */
    uint32_t exposures_bus;          /* ... via PublicTransport, simulator.rs:436-446 */
    uint32_t lockdown;               /* InterventionStatus::lockdown_enabled() after this step */
    uint32_t vaccination_active;     /* vaccination_program_started() after this step */
    uint32_t mask_status;            /* ESIM_MASK_* after this step */
    uint32_t n_riders;               /* citizens on public transport this step */
    uint32_t vaccinated_now;         /* citizens set Vaccinated at the end of this step, simulator.rs:524-553 */
    uint32_t eligible_count;         /* |citizens_eligible_for_vaccine| after this step */
    uint32_t disease_exists;         /* StatisticsRecorder::disease_exists(), statistics.rs:289-291 */
    uint32_t reserved;
} esim_step_result;

typedef struct esim_ctx esim_ctx;

/* DiseaseModel::covid() + Default for InterventionThresholds/InterventionStatus. */
void esim_default_params(esim_params *p);

/* Replaces Simulator::from(SimulatorBuilder) (simulator.rs:601-644): creates the device
 * context; esim_upload_population copies the population to HBM and builds the derived
 * tables (route lists, probability-threshold LUT). */
int  esim_create(const esim_params *p, esim_ctx **out);
int  esim_upload_population(esim_ctx *ctx, const esim_population *pop);
/* Back to time step 0 with the uploaded population (all Susceptible, seeds Infected(0)). */
int  esim_reset(esim_ctx *ctx);

/* Replaces Simulator::step (simulator.rs:131-152).  out->disease_exists == 0 is the
 * reference's Ok(false). */
int  esim_step(esim_ctx *ctx, esim_step_result *out);
/* Replaces the loop of Simulator::simulate (simulator.rs:114-123): runs up to n_steps on the
 * device without host round trips; stops early when the disease is gone iff stop_when_done.
 * out_array has room for n_steps entries; *n_done receives the number written. */
int  esim_run(esim_ctx *ctx, uint32_t n_steps, int stop_when_done,
              esim_step_result *out_array, uint32_t *n_done);

/* Split-phase form of one step for sharded (multi-GPU) runs.  Between the phases the
 * caller SUM-all-reduces the exchange buffer (device memory, uint32) across shards:
 *   esim_step_begin     -- Citizen::execute_time_step for every citizen (generate_exposures,
 *                          simulator.rs:155-260); packs counts + shared infected counts
 *   [all-reduce A]
 *   esim_step_exposures -- apply_exposures (simulator.rs:262-405); packs vaccination liveness
 *   [all-reduce B]
 *   esim_step_finish    -- apply_interventions (simulator.rs:455-556), writes the record
 * With one shard the buffers need no reduction and esim_step() is exactly this sequence. */
int  esim_step_begin(esim_ctx *ctx);
int  esim_step_exposures(esim_ctx *ctx);
int  esim_step_finish(esim_ctx *ctx, esim_step_result *out /* may be NULL */);
int  esim_exchange_buffer(esim_ctx *ctx, int which /* 0 = A, 1 = B, 2 = F */, void **device_ptr, size_t *n_u32);
/* Pipelined chunks.  A citizen exposed in step t is Infected no earlier than t + exposed_time + 1
 * (disease.rs:47-71).  Hence, while no vaccination programme runs, the Infected census -- and with it every
 * intervention decision (interventions.rs:110-184), the schedule (citizen.rs:176-206) and who marks which
 * building -- is known for the next n <= exposed_time + 1 steps (n = size of buffer F, at most 96).  esim_run
 * uses this by itself (one kernel per step, books written once per chunk).  Shards that share no building
 * (n_shared_* == 0) use it to run without per-step collectives:
 *   esim_future_infected -- writes this shard's Infected census of the next n steps into buffer F
 *   [SUM all-reduce of F over the shards]
 *   esim_run_free(k, &done) -- runs min(k, steps before the one that would start vaccinating) whole steps
 *                              with no exchange; records hold THIS shard's census; done < k means the next
 *                              step must be a coupled one (esim_step_begin / _exposures / _finish).
 * Buffer F holds n + 1 words: the census of the next n steps and one word counting the shards whose chunk does
 * not fit the one-pass form below (after the all-reduce every shard therefore takes the same form).
 * Bursts: esim_run_free waits for the device once per chunk.  To keep several chunks in flight the caller
 * opens a burst with esim_free_begin(k) (k = steps it may cover), calls esim_future_infected once and then repeats
 * { all-reduce F; esim_free_enqueue } as often as it likes -- each round enqueues one whole chunk, which is a no-op on
 * EVERY shard when it cannot run in the one-pass form or would reach the step that starts vaccinating, and which ends by
 * writing the census ahead of the chunk after it into F -- and calls esim_free_collect(&done) once: done = steps the
 * burst advanced (the same on all shards).
 * esim_set_pipeline(ctx, level): 0 = sequential steps only; 1 = chunks run as one kernel per step (k_pipe);
 * 3 (default) = as 2, and chunks keep running under a vaccination programme (esim_vax_chunk_stats); 4 = as 3 on the persistent
 * item map (a citizen is entered into the items it stands in once, when it turns Infected, instead of in every chunk; unsharded contexts);
 * 2 = additionally to 1, when the chunk's marks fit the hash map, ALL steps of a chunk are drawn in one
 * pass (a citizen's exposure step is the earliest step at which any of its draws succeeds -- one atomicMin on
 * the citizen word per successful draw).  esim_chunk_timing: device time (ms), steps and number of such chunks
 * since the last call (measured while kernel timing is enabled). */
/* ---- the exchange between shards, owned by the library ("library owns streams / RCCL communicators") ----
 * esim_comm_unique_id     -- ncclGetUniqueId on one rank (cap >= 128 bytes); the caller carries it to the other ranks
 *                            (the Rust caller over whatever it launches its processes with; bench.py over torch.distributed)
 * esim_comm_init_rccl     -- collective over all ranks: an RCCL communicator on the context's device; every exchange is then
 *                            an ncclAllReduce enqueued on the context's stream between its kernels
 * esim_comm_init_callback -- instead: the caller's own SUM all-reduce over the ranks, in place, of `n_u32` uint32 in HOST memory
 *                            at `host_ptr` (the library stages the device buffer through it with the stream drained; `which`
 *                            names the buffer: 0 A, 1 B, 2 F, 3 plan liveness, 4 commuter records, 5 cuts, 6 records, 7 status,
 *                            8 the set-up's layout check, 9 which shards have members in each shared building);
 *                            returns 0 on success.  For transports other than RCCL and for tests with several ranks on one GPU.
 * esim_run_sharded        -- replaces the loop of Simulator::simulate (simulator.rs:114-123) for this rank's shard: n_steps
 *                            time steps, every rank calling it with the same n_steps; records of these steps hold the census
 *                            of the WHOLE population on every rank. */
/* Set-up.  esim_comm_init_* come AFTER esim_upload_population (a new upload invalidates the communicator) and are collective:
 * the ranks' shards are checked against each other with one small all-reduce -- rank r must hold the r-th stretch of the global
 * citizen ids of ONE world (same n_citizens_global, shared tables of the same size), else ESIM_EINVAL on every rank that sees
 * the mismatch.  world <= 31.
 * Failure semantics (the reference bubbles a failed step() up to main, run/src/main.rs:306-308): the shards' device-side error
 * words are summed in the same collectives that carry the data, and before every read-back of the control block, so ALL ranks
 * return from esim_run_sharded together and with the same ESIM_E* code (records of that call are then undefined); the host's
 * waits on a stream that holds RCCL collectives have a deadline (esim_comm_set_timeout, default 60 s, or ESIM_COMM_TIMEOUT_S):
 * on expiry the communicator is aborted (ncclCommAbort) and the call returns ESIM_ETIMEDOUT -- exit with an error then.
 * esim_run_sharded has no early stop (a shard cannot know that the disease is gone elsewhere): it always runs n_steps.
 * esim_debug_inject_error: diagnostics -- raises a sticky device-side error on this context (tests of the above). */
typedef int (*esim_allreduce_fn)(void *user, int which, void *host_ptr, size_t n_u32);
int  esim_comm_unique_id(void *out, size_t cap);
int  esim_comm_init_rccl(esim_ctx *ctx, const void *unique_id, size_t id_bytes, int rank, int world);
int  esim_comm_init_callback(esim_ctx *ctx, esim_allreduce_fn fn, void *user, int rank, int world);
int  esim_comm_set_timeout(esim_ctx *ctx, double seconds);
int  esim_debug_inject_error(esim_ctx *ctx, int code);
int  esim_comm_stats(esim_ctx *ctx, uint64_t *collectives);
int  esim_run_sharded(esim_ctx *ctx, uint32_t n_steps, uint32_t *n_done);
/* How the steps of sharded runs were executed so far: as time-parallel chunks (one round of exchanges per chunk) / as coupled
 * steps (two exchanges per step). */
int  esim_shard_stats(esim_ctx *ctx, uint64_t *chunk_steps, uint64_t *coupled_steps);
int  esim_future_infected(esim_ctx *ctx);
int  esim_run_free(esim_ctx *ctx, uint32_t n_steps, uint32_t *n_done);
int  esim_free_begin(esim_ctx *ctx, uint32_t n_steps);
int  esim_free_enqueue(esim_ctx *ctx);
int  esim_free_collect(esim_ctx *ctx, uint32_t *n_done);
int  esim_set_pipeline(esim_ctx *ctx, int level);
int  esim_chunk_timing(esim_ctx *ctx, double *total_ms, uint64_t *steps, uint64_t *chunks);
/* Device time of the chunk pass per KERNEL (HIP events in front of every kernel of a chunk on the context's stream, resolved at
 * the read-back that ends a burst; a kernel's figure includes the boundary to the next one): accumulated ms and launches since
 * the last call, indexed by ESIM_CK_*.  The reference's three phase timers (simulator.rs:137-143) map onto them as
 * "Generate Exposures" = marks + fold, "Apply Exposures" = draw + units, "Apply Interventions" = everything else (plan,
 * decisions, counts, books, scatter) -- apportioned: a chunk pass works on up to 96 steps at once.  ESIM_CK_TINY: a whole chunk
 * with few Infected in one launch (census ahead, decisions, marks, draws, books): the Python binding shares it out over the
 * three labels by the kernel's own stage timers (entries and keys 10 %, draws 40 %, census, decisions and books 50 %). */
enum { ESIM_CK_MARKS = 0, ESIM_CK_FOLD, ESIM_CK_DRAW, ESIM_CK_UNITS, ESIM_CK_COUNT, ESIM_CK_BOOKS, ESIM_CK_SCATTER, ESIM_CK_VAX, ESIM_CK_VAX_ADJ,
       ESIM_CK_VAX_FINAL, ESIM_CK_DECIDE, ESIM_CK_FUTURE, ESIM_CK_MAP_CLEAR, ESIM_CK_TINY, ESIM_CK_VAX_REPAIR, ESIM_CK_N };
int  esim_enable_chunk_kernel_timing(esim_ctx *ctx, int enable);
int  esim_chunk_kernel_timings(esim_ctx *ctx, double ms[ESIM_CK_N], uint64_t calls[ESIM_CK_N]);
/* Steps run as time-parallel chunks under a vaccination programme (pipeline level 3: the chunk's vaccinations are planned
 * ahead, simulator.rs:524-553 being a pure function of the step and of citizens_eligible_for_vaccine), and how many of those
 * chunks were cut short because a citizen the plan had chosen left the eligible set on a bus first (simulator.rs:447-449). */
int  esim_vax_chunk_stats(esim_ctx *ctx, uint64_t *steps, uint64_t *cuts);
/* Planned chunks in which a citizen was exposed on a bus before the step the plan vaccinates it in, and whose plan was REPAIRED
 * for the steps behind that exposure (k_chunk_vax<true>: walked again with the eligible set as it truly stood) instead of the
 * chunk being cut there; a chunk is still cut -- behind the step concerned -- when a newly chosen citizen is Infected or
 * exposed later in the chunk.  Sharded runs do the same with two more exchanges per planned chunk (the steps in which a shard lost
 * a citizen, callback `which` 10; the candidates' liveness a second time, `which` 3).  ESIM_VAX_REPAIR=0 switches it off. */
int  esim_vax_repair_stats(esim_ctx *ctx, uint64_t *repairs);
/* Record log read-back for split-phase runs (records first..first+n-1, 1-based time steps). */
int  esim_read_records(esim_ctx *ctx, uint32_t first_step, uint32_t n, esim_step_result *out);
/* The HIP stream all work of this context is enqueued on (hipStream_t as void*).  esim_set_stream
 * makes the context use a caller-owned stream instead (e.g. the one a collective library orders
 * its all-reduce against); esim_set_exchange_buffer replaces exchange buffer `which` by caller-owned
 * device memory of at least the size esim_exchange_buffer reports (e.g. a tensor the collective
 * library can address). */
int  esim_stream(esim_ctx *ctx, void **stream);
int  esim_set_stream(esim_ctx *ctx, void *stream);
int  esim_set_exchange_buffer(esim_ctx *ctx, int which, void *device_ptr);
int  esim_synchronize(esim_ctx *ctx);

/* Per-citizen state in reference terms, for visualisation / lookup-table sync / checkpoints
 * (replaces reading Simulator.output_areas[..].citizens, run/src/main.rs:246-259,
 * visualisation/src/citizen_connections.rs:40-62).  Any pointer may be NULL.
 *   status  ESIM_* code           (Citizen::disease_status)
 *   timer   Exposed(t)/Infected(t) payload, 0 otherwise
 *   current_building               (Citizen::current_building_position)
 *   on_bus  0 None, 1 (home OA, work OA), 2 (work OA, home OA)   (Citizen::on_public_transport)
 *   eligible member of Simulator::citizens_eligible_for_vaccine  (simulator.rs:97) */
int  esim_download_state(esim_ctx *ctx, uint8_t *status, uint16_t *timer,
                         uint32_t *current_building, uint8_t *on_bus, uint8_t *eligible);
/* Every exposure so far, in time order: the citizen (local index), the time step and whether it happened on public
 * transport -- the calls of StatisticsRecorder::add_exposure (statistics.rs:181-195) that feed exposures.json's
 * per-Output-Area series (statistics.rs:119-136).  A building exposure is credited to the Output Area the citizen stands
 * in at that step (simulator.rs:324 only exposes members whose current area is the building's).  Order inside a time
 * step is unspecified.  *n_out = number of exposures; ESIM_ERANGE (with *n_out set) when cap is too small. */
int  esim_download_exposure_log(esim_ctx *ctx, uint32_t *citizen, uint32_t *step, uint8_t *on_bus, uint32_t cap, uint32_t *n_out);
/* Checkpoint / resume (the reference has none for the simulation state): everything a step reads that is
 * not part of the uploaded population -- the citizen words, the census histogram, the exposure log, the control block,
 * the records so far.  Restore goes into a context that holds the SAME population (or shard) and parameters; the run
 * continues bit for bit as if it had not been interrupted.  Between calls, i.e. never inside esim_step_begin..finish or
 * an open burst. */
int  esim_checkpoint_size(esim_ctx *ctx, size_t *bytes);
int  esim_checkpoint_save(esim_ctx *ctx, void *buf, size_t cap);
int  esim_checkpoint_restore(esim_ctx *ctx, const void *buf, size_t bytes);

/* GPU time per phase since the last reset, seconds, in the reference's timer labels
 * (simulator.rs:137,140,143; statistics.rs:138-140):
 *   out[0] "Generate Exposures", out[1] "Apply Exposures", out[2] "Apply Interventions", out[3] total.
 * Only measured while phase timing is enabled (it inserts events between phases). */
int  esim_enable_phase_timing(esim_ctx *ctx, int enable);
int  esim_phase_timings(esim_ctx *ctx, double out[4]);

/* Mean device time (ms) of a multi-workgroup time step -- HIP event before k_infected to HIP event after
 * k_finish on the context's stream -- over the steps timed since the last call.
 * esim_enable_kernel_timing(ctx, n) brackets every n-th such step (n = 0: off) and every launch of the
 * persistent kernel (esim_small_kernel_timing). */
int  esim_enable_kernel_timing(esim_ctx *ctx, int enable);
int  esim_kernel_timings(esim_ctx *ctx, double *step_ms, uint32_t *out_n);

/* While few citizens are Infected a time step is a handful of dependent memory round trips; a persistent
 * single-workgroup kernel then advances many steps per launch (workgroup barriers instead of kernel
 * boundaries).  It hands over to the multi-workgroup kernels whenever a step has more than `max_infected`
 * Infected citizens (default 128; 0 disables it).  esim_small_kernel_timing: accumulated duration (ms) and
 * steps of those launches while kernel timing is enabled, then resets the accumulators. */
int  esim_set_small_step_limit(esim_ctx *ctx, uint32_t max_infected);
int  esim_small_kernel_timing(esim_ctx *ctx, double *total_ms, uint64_t *steps);
/* The same idea for time-parallel chunks: while the chunk last read back had at most `max_pairs` (Infected citizen, step) pairs, a
 * chunk is ONE launch of one workgroup (k_chunk_tiny: census ahead, decisions, marks, draws and books of up to 64 Infected; a
 * chunk that has outgrown it does not advance and is run in the wide form next).  Default 2048; 0 disables it. */
int  esim_set_tiny_chunk_limit(esim_ctx *ctx, uint32_t max_pairs);
/* Pipelined steps: mean duration (ms) of the sampled k_pipe launches, how many were sampled, how many steps ran
 * pipelined since the last call. */
int  esim_pipeline_timing(esim_ctx *ctx, double *mean_step_ms, uint64_t *steps_timed, uint64_t *steps_run);

/* Diagnostics: the control block's view of the last chunk (t, chunk_ok, chunk_parallel, chunk_pairs, n_items,
 * items_per_wave, n_units, n_route_pairs, n_route_pairs_big, n_newexp, log_len, n_susceptible, lockdown, mask,
 * at_work, bus_dir). */
int  esim_debug_counters(esim_ctx *ctx, uint32_t out[16]);

const char *esim_last_error(const esim_ctx *ctx);   /* ctx may be NULL: last esim_create error */
void esim_destroy(esim_ctx *ctx);

/* The exposure-probability LUT the kernels use: thresholds[mask][n & 255] =
 * ceil(q * 2^32) with q = 1 - (1 - p_eff)^(n as u8) (citizen.rs:47-49,239;
 * disease.rs:131-154).  mask 0: p_eff = p, mask 1: p_eff = p - p*mask_effectiveness.
 * Pure host arithmetic; exported so parity tests can pin it. */
int  esim_threshold_lut(const esim_params *p, uint64_t out[512]);

/* ---- synthetic populations: stands in for load_census_data + osm_data +
 * SimulatorBuilder, whose inputs are not available.  Pure host code. ---- */
typedef struct esim_synth_spec {
    uint32_t n_citizens, n_areas, citizens_per_school, n_seeds;
    uint64_t seed;
    double   area_jitter;     /* per-area census population = mean * (1 +- jitter) */
    double   p_public_transport, p_mask_compliant;
    double   p_work_from_home; /* extra share of adults without a workplace (0: only what the build leaves at home) */
    double   p_teaching;       /* census group 9 "Elementary", mapped to Teaching */
    /* the OSM side of SimulatorBuilder's inputs, per Output Area: log-normal counts round(median * exp(sigma z)) */
    double   household_buildings_median, household_buildings_sigma;   /* dwellings; household size = pop / count + 1, output_area.rs:139 */
    double   p_area_without_households;                               /* such areas get no citizens, simulator_builder.rs:226-235 */
    double   workplace_buildings_median, workplace_buildings_sigma;   /* possible workplace buildings, simulator_builder.rs:717 */
    double   p_area_without_workplaces;                               /* "No Workplace buildings exist", simulator_builder.rs:827-835 */
    double   workplace_floor_median, workplace_floor_sigma;           /* floor area of one, m^2 (RawBuilding::size) */
    uint32_t teacher_candidate_schools;                               /* MAX_ITEMS_RETURNED = 200, osm_data/src/quadtree.rs:544 */
    uint32_t reserved;
} esim_synth_spec;
/* presets: "york", "yh_census", "syn3m5", "uk64m"  */
int  esim_synth_preset(const char *name, esim_synth_spec *out);
/* Follows SimulatorBuilder::build (simulator_builder.rs:1162-1292) on synthetic census / OSM inputs: Output Areas are the
 * cells of a near-square map in row-major order, households stand at points of their cell, students go to the closest
 * school, teachers to the closest one lacking class teachers (else they are its secondary staff), workers to a
 * workplace of their occupation inside their home area.  Allocates the arrays of *out (shared_* left empty); release
 * with esim_synth_free. */
int  esim_synth_create(const esim_synth_spec *spec, esim_population *out);
/* The shard `shard` of `n_shards` of the same world: the whole world is generated and cut (esim_shard_population) into
 * bands of the map with about the same expected work each (esim_shard_cuts, by_work = 1); citizen_id_base and n_citizens_global are set, Philox
 * counters stay global.  Commuters to a school across a cut make that school (and its rooms) shared. */
int  esim_synth_create_shard(const esim_synth_spec *spec, uint32_t shard, uint32_t n_shards, esim_population *out);
void esim_synth_free(esim_population *pop);
/* Cuts the shard of Output Areas [area_begin, area_end) out of a whole population:
 * citizens living there, every building/room they reference (remote ones become ghosts),
 * and shared tables laid out identically on every shard of the same `cuts`
 * (cuts[0..n_shards] are the area boundaries of all shards).  Release with esim_synth_free. */
int  esim_shard_population(const esim_population *whole, const uint32_t *cuts, uint32_t n_shards,
                           uint32_t shard, esim_population *out);
/* Area boundaries cuts_out[0..n_shards] of n_shards bands of the map: by_work == 0, about the same number of citizens each;
 * otherwise about the same expected work each -- a citizen is drawn for by the shard it lives on, in every list it is a member of
 * (household, work place, class room), so its weight is 1 + the sizes of those lists, a band's the sum over its residents.
 * Static weights: where the epidemic will sit is not known before the run. */
int  esim_shard_cuts(const esim_population *whole, uint32_t n_shards, int by_work, uint32_t *cuts_out);

#ifdef __cplusplus
}
#endif
#endif
