/*
 * esim_oracle.h -- CPU ORACLE for the per-timestep Citizen update loop.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, bench.py's
 * `cpu_baseline` leg and __graft_entry__.smoke() may load it, and only as the
 * checker.  Nothing under epidemicsimulator_amd/ links, imports or calls it.
 *
 * It is a literal restatement (single-threaded unless orc_set_threads asks otherwise) of the reference `sim` crate's
 * `Simulator::step` (sim/src/simulator.rs:131-556 and the files it calls), with
 * one deliberate substitution: every `thread_rng()` draw is replaced by a
 * counter-based Philox4x32-10 draw (contract below), because the reference RNG
 * is OS-seeded and cannot be reproduced.
 *
 * PARITY STATUS: "parity unpinned" by the reference's own tests -- the
 * reference has no test, fixture or golden vector for this path (SURVEY.md
 * section 4 / 8c) and cannot be built here (no cargo/rustc, no input data).
 * What pins this oracle instead (tests/test_oracle_*.py):
 *   - Philox4x32-10 known-answer vectors (Random123 kat_vectors);
 *   - the f64 exposure-probability bit patterns of sim/src/models/citizen.rs:47-49
 *     evaluated with host libm;
 *   - E/I/R timer windows (sim/src/disease.rs:47-71) against the step numbers in
 *     logs/pc_logs/v1.6/york.log:489-490;
 *   - schedule hours, intervention thresholds, conservation S+E+I+R+V == N;
 *   - a distributional envelope from statistics_results/v1.7.1 (not bit-exact).
 *
 * RNG CONTRACT (shared, by specification, with the HIP implementation):
 *   block(c0,c1,c2,c3) = Philox4x32-10(counter=(c0,c1,c2,c3), key=(seed_lo,seed_hi))
 *   uniform     = w * 2^-32  in [0,1) for a 32-bit word w of a block (the reference's Uniform<f64> has 52 random bits,
 *        citizen.rs:44; the probabilities differ by less than 2^-32)
 *   exposure draw for citizen g (GLOBAL index) in step t, slot s:
 *        w = word t & 3 of block(g, t >> 2, s, 0) -- four consecutive steps per block, all 128 bits used;
 *        success  <=>  uniform < q   (strict, citizen.rs:242)
 *        s = 0  draw from the home building's list          (building.rs:202)
 *        s = 1  draw from a non-school work building's list (building.rs:278)
 *        s = 2  draw on a bus                               (simulator.rs:436)
 *        s = 16+j  j-th draw from a school room             (building.rs:494-522)
 *   bus order key for rider g in step t: w0 of block(g, t, 3, 0); riders of one
 *        route are ordered by (key, g) ascending and cut into buses of
 *        BUS_CAPACITY (replaces shuffle + pop, simulator.rs:362-388).
 *   vaccination candidate i of step t: j = mulhi64(w0<<32|w1 of block(i,t,4,0), N)
 *        -- the chosen set is the first k distinct ELIGIBLE citizens in the
 *        sequence j_0, j_1, ... , k = min(vaccination_rate, |eligible|); when
 *        |eligible| <= vaccination_rate every eligible citizen is chosen
 *        (uniform k-subset, replaces choose_multiple, simulator.rs:525-527).
 */
#ifndef ESIM_ORACLE_H
#define ESIM_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* status codes (sim/src/disease.rs:36-44) */
enum { ORC_S = 0, ORC_E = 1, ORC_I = 2, ORC_R = 3, ORC_V = 4 };
/* building types (sim/src/models/building.rs:46-53) */
enum { ORC_HOUSEHOLD = 0, ORC_WORKPLACE = 1, ORC_SCHOOL = 2 };
/* mask states (sim/src/interventions.rs:26-30) */
enum { ORC_MASK_NONE = 0, ORC_MASK_PT = 1, ORC_MASK_EVERYWHERE = 2 };

#define ORC_NO_ROOM 0xFFFFFFFFu
#define ORC_FLAG_USES_PT 1u      /* citizen.rs:132 */
#define ORC_FLAG_MASK_COMPLIANT 2u /* citizen.rs:131 */

typedef struct {
    double exposure_chance;      /* disease.rs:120  0.00055 */
    double mask_effectiveness;   /* disease.rs:127  0.70 */
    double lockdown_threshold;   /* interventions.rs:74 0.0034 */
    double vaccination_threshold;/* interventions.rs:75 0.005 */
    double mask_pt_threshold;    /* interventions.rs:55 0.001 */
    double mask_everywhere_threshold; /* interventions.rs:56 0.0022 */
    uint32_t exposed_time;       /* disease.rs:122  96 */
    uint32_t infected_time;      /* disease.rs:123  336 */
    uint32_t vaccination_rate;   /* disease.rs:125  1530 */
    uint32_t bus_capacity;       /* config.rs:37    20 */
    uint32_t start_hour;         /* citizen.rs:154  9 */
    uint32_t end_hour;           /* citizen.rs:155  17 */
    uint64_t seed;               /* Philox key */
} orc_params;

typedef struct {
    uint32_t n_citizens, n_buildings, n_areas, n_rooms, n_seeds;
    const uint32_t *home;        /* [n_citizens] building index, citizen.rs:116 */
    const uint32_t *work;        /* [n_citizens] building index (== home if none), citizen.rs:118 */
    const uint32_t *room;        /* [n_citizens] room index or ORC_NO_ROOM, building.rs:341 */
    const uint8_t  *flags;       /* [n_citizens] ORC_FLAG_* */
    const uint32_t *bld_area;    /* [n_buildings] output-area index, building.rs:63 */
    const uint8_t  *bld_type;    /* [n_buildings] */
    const uint32_t *room_bld;    /* [n_rooms] school building of each room */
    const uint32_t *seeds;       /* [n_seeds] citizens starting Infected(0), simulator_builder.rs:1139 */
} orc_population;

/* one StatisticEntry (statistics.rs:208-215) + what the step decided */
typedef struct {
    uint32_t time_step, susceptible, exposed, infected, recovered, vaccinated;
    uint32_t exposures_building, exposures_bus;   /* add_exposure calls, statistics.rs:181 */
    uint32_t lockdown, vaccination_active, mask_status, n_riders;
    uint32_t vaccinated_now, eligible_count;
    uint32_t disease_exists;                      /* statistics.rs:289-291 */
    uint32_t pad;
} orc_record;

typedef struct orc_sim orc_sim;

void     orc_default_params(orc_params *p);
orc_sim *orc_create(const orc_params *p, const orc_population *pop);
void     orc_destroy(orc_sim *s);
/* One Simulator::step (simulator.rs:131-152).  Returns 0 ok, <0 on the
 * reference's error path (S underflow, statistics.rs:275-287). */
int      orc_step(orc_sim *s, orc_record *out);
/* Runs up to n steps; stops early when disease_exists == 0 iff stop_when_done
 * (simulator.rs:114-118).  Returns number of records written, <0 on error. */
int      orc_run(orc_sim *s, uint32_t n, orc_record *out, int stop_when_done);
/* Per-citizen state for full-state parity: status code, timer (E/I only),
 * at_work (current_building_position == workplace_code != household_code ... see .c),
 * bus (0 none, 1 home->work, 2 work->home), eligible-for-vaccine flag. */
void     orc_get_state(const orc_sim *s, uint8_t *status, uint16_t *timer,
                       uint8_t *at_work, uint8_t *bus, uint8_t *eligible);

/* Runs the per-citizen pass of every later step on `threads` host threads (OpenMP; the reference uses rayon there,
 * simulator.rs:167-260).  Results are identical to the single-threaded run.  Returns the thread count in force. */
int      orc_set_threads(orc_sim *s, int threads);

/* The add_exposure calls so far (statistics.rs:181-195), per citizen: time step (0 = none) and Output Area credited
 * (ORC_NO_ROOM: public transport). */
void     orc_get_exposures(const orc_sim *s, uint32_t *step, uint32_t *area);

/* primitives exposed so tests can pin them */
void     orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
double   orc_binomial(double probability, uint8_t n);                 /* citizen.rs:47-49 */
double   orc_exposure_chance(const orc_params *p, int is_vaccinated, int mask_status,
                             int on_pt_and_compliant);                /* disease.rs:131-154 */
double   orc_q(const orc_params *p, uint64_t n, int mask_compliant, int global_mask); /* citizen.rs:221-240 */
uint32_t orc_u32(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2);

#ifdef __cplusplus
}
#endif
#endif
