/*
 * esim_oracle.c -- CPU ORACLE (test infrastructure; see esim_oracle.h header).
 *
 * Literal restatement of the reference's per-timestep loop; single-threaded, except that orc_set_threads puts the
 * per-citizen pass on several host threads with identical results.
 * Every function cites the reference file:line (relative to /root/reference)
 * it follows.  Data structures deliberately mirror the reference's shape
 * (array-of-struct citizens, per-building member lists, per-step rider lists,
 * per-building infected lists) -- this file is written for checkability, not
 * speed.  PARITY STATUS: unpinned by reference tests (none exist); pinned by
 * the known answers listed in esim_oracle.h.
 */
#include "esim_oracle.h"
#include <math.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ Philox */
/* Philox4x32-10, Salmon et al. SC'11 (Random123).  The reference uses rand 0.8
 * thread_rng (sim/Cargo.toml:22; simulator.rs:342,630) which is OS-seeded; this
 * is the substitution documented in the header. */
#define PHILOX_M0 0xD2511F53u
#define PHILOX_M1 0xCD9E8D57u
#define PHILOX_W0 0x9E3779B9u
#define PHILOX_W1 0xBB67AE85u

void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)PHILOX_M0 * c0;
        uint64_t p1 = (uint64_t)PHILOX_M1 * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += PHILOX_W0; k1 += PHILOX_W1;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

static void philox_block(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t out[4])
{
    uint32_t ctr[4] = { c0, c1, c2, 0u };
    uint32_t key[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) };
    orc_philox4x32_10(ctr, key, out);
}

/* The 32-bit integer of the exposure draw of citizen c0 in step c1, slot c2: steps 4k .. 4k+3 share one block, step t takes
 * its word t & 3 -- all 128 bits of a block are used (RNG contract, esim_oracle.h). */
uint32_t orc_u32(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2)
{
    uint32_t w[4];
    philox_block(seed, c0, c1 >> 2, c2, w);
    return w[c1 & 3u];
}

/* Uniform in [0,1) replacing RANDOM_DISTRUBUTION.sample(rng) (citizen.rs:44,242) */
static double uniform01(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2)
{
    return (double)orc_u32(seed, c0, c1, c2) * 0x1.0p-32;
}

/* ------------------------------------------------------------- probability */
/* sim/src/models/citizen.rs:47-49 */
double orc_binomial(double probability, uint8_t n)
{
    return 1.0 - pow(1.0 - probability, (double)n);
}

/* sim/src/disease.rs:131-154 */
double orc_exposure_chance(const orc_params *p, int is_vaccinated, int mask_status,
                           int on_pt_and_compliant)
{
    double sub;
    switch (mask_status) {
    case ORC_MASK_NONE: sub = 0.0; break;
    case ORC_MASK_PT:
        sub = on_pt_and_compliant ? p->exposure_chance * p->mask_effectiveness : 0.0;
        break;
    default: sub = p->exposure_chance * p->mask_effectiveness; break;
    }
    double chance = p->exposure_chance - sub - (is_vaccinated ? 1.0 : 0.0);
    if (signbit(chance)) chance = 0.0;
    return chance;
}

/* sim/src/models/citizen.rs:228-240 -- note the inverted mask logic (Q7): a
 * compliant citizen is evaluated with MaskStatus::None. `on_bus` only matters
 * in the PublicTransport state, which a compliant citizen never reaches. */
static double q_literal(const orc_params *p, uint64_t exposure_total, int mask_compliant,
                        int on_bus, int is_vaccinated, int global_mask)
{
    int mask = mask_compliant ? ORC_MASK_NONE : global_mask;
    double chance = orc_exposure_chance(p, is_vaccinated, mask, mask_compliant && on_bus);
    return orc_binomial(chance, (uint8_t)exposure_total);   /* `as u8`, citizen.rs:239 */
}

double orc_q(const orc_params *p, uint64_t n, int mask_compliant, int global_mask)
{
    return q_literal(p, n, mask_compliant, 0, 0, global_mask);
}

/* ------------------------------------------------------------------- state */
typedef struct {
    uint8_t  status;       /* DiseaseStatus, disease.rs:36-44 */
    uint16_t timer;        /* Exposed(t)/Infected(t) */
    uint8_t  uses_pt;      /* citizen.rs:132 */
    uint8_t  mask_ok;      /* citizen.rs:131 */
    uint8_t  bus;          /* on_public_transport: 0 None, 1 (home OA, work OA), 2 (work OA, home OA) */
    uint8_t  at_work;      /* 1 iff current_building_position was last set by the "starts work" arm */
    uint8_t  eligible;     /* member of citizens_eligible_for_vaccine, simulator.rs:97 */
    uint32_t home, work;   /* household_code / workplace_code, citizen.rs:116-118 */
    uint32_t cur;          /* current_building_position, citizen.rs:127 */
    uint32_t room;         /* occupant_to_class, building.rs:341 */
    uint32_t school_draws; /* per-step count of room draws already taken (slot 16+j) */
    uint32_t exp_step;     /* time step of the add_exposure call for this citizen (0: none), statistics.rs:181-195 */
    uint32_t exp_area;     /* ... the Output Area credited (building exposures), or ORC_NO_ROOM for public transport */
    int32_t  inf_next;     /* link in the per-building infected list of this step */
} citizen_t;

typedef struct { uint32_t src, dst, key, id; uint8_t infected; } rider_t;

struct orc_sim {
    orc_params P;
    uint32_t n_cit, n_bld, n_area, n_room;
    citizen_t *cit;
    uint32_t *bld_area; uint8_t *bld_type; uint32_t *room_bld;
    /* static membership lists (CSR), output_area.rs:172-180, simulator_builder.rs:1076,1100,
     * building.rs:404-431 */
    uint32_t *res_off, *res_idx;     /* residents per building (home == b) */
    uint32_t *wrk_off, *wrk_idx;     /* workers per building (work == b, work != home) */
    uint32_t *room_off, *room_idx;   /* participants per room */
    /* per-step scratch */
    int32_t  *inf_head; uint32_t *inf_cnt; uint32_t *touched; uint32_t n_touched;
    rider_t  *riders; uint32_t n_riders;
    int threads;               /* > 1: the per-citizen pass of a step runs on that many host threads (orc_set_threads) */
    uint32_t *inf_list;        /* [n_cit] scratch of the threaded pass: Infected citizens not on a bus, ascending */
    uint64_t *thr_counts;      /* [threads][8] */
    uint32_t *chosen_stamp;
    /* global state */
    uint32_t time_step;              /* StatisticsRecorder.current_time_step, statistics.rs:104 */
    int lockdown;                    /* InterventionStatus.lockdown.is_some() */
    int vaccination;                 /* InterventionStatus.vaccination.is_some() */
    int mask;                        /* InterventionStatus.mask_status */
    int have_eligible;               /* citizens_eligible_for_vaccine.is_some() */
    uint32_t eligible_count;
};

void orc_default_params(orc_params *p)
{
    p->exposure_chance = 0.00055;            /* disease.rs:120 */
    p->mask_effectiveness = 0.70;            /* disease.rs:127 */
    p->lockdown_threshold = 0.0034;          /* interventions.rs:74 */
    p->vaccination_threshold = 0.005;        /* interventions.rs:75 */
    p->mask_pt_threshold = 0.001;            /* interventions.rs:55 */
    p->mask_everywhere_threshold = 0.0022;   /* interventions.rs:56 */
    p->exposed_time = 4 * 24;                /* disease.rs:122 */
    p->infected_time = 14 * 24;              /* disease.rs:123 */
    p->vaccination_rate = 85 * 18;           /* disease.rs:125 */
    p->bus_capacity = 20;                    /* config.rs:37 */
    p->start_hour = 9;                       /* citizen.rs:154 */
    p->end_hour = 17;                        /* citizen.rs:155 */
    p->seed = 0x5EED2011ull;
}

static void build_csr(uint32_t n_keys, uint32_t n_items, const uint32_t *key_of_item,
                      const uint8_t *use, uint32_t **off_out, uint32_t **idx_out)
{
    uint32_t *off = (uint32_t *)calloc((size_t)n_keys + 1, sizeof(uint32_t));
    for (uint32_t i = 0; i < n_items; ++i)
        if (!use || use[i]) off[key_of_item[i] + 1]++;
    for (uint32_t k = 0; k < n_keys; ++k) off[k + 1] += off[k];
    uint32_t *idx = (uint32_t *)malloc(sizeof(uint32_t) * (off[n_keys] ? off[n_keys] : 1));
    uint32_t *cur = (uint32_t *)malloc(sizeof(uint32_t) * ((size_t)n_keys + 1));
    memcpy(cur, off, sizeof(uint32_t) * ((size_t)n_keys + 1));
    for (uint32_t i = 0; i < n_items; ++i)
        if (!use || use[i]) idx[cur[key_of_item[i]]++] = i;
    free(cur);
    *off_out = off; *idx_out = idx;
}

orc_sim *orc_create(const orc_params *p, const orc_population *pop)
{
    /* validation of the population contract */
    for (uint32_t c = 0; c < pop->n_citizens; ++c) {
        if (pop->home[c] >= pop->n_buildings || pop->work[c] >= pop->n_buildings) return NULL;
        int school = pop->bld_type[pop->work[c]] == ORC_SCHOOL && pop->work[c] != pop->home[c];
        if (school) {
            if (pop->room[c] >= pop->n_rooms || pop->room_bld[pop->room[c]] != pop->work[c]) return NULL;
        }
    }
    for (uint32_t b = 0; b < pop->n_buildings; ++b)
        if (pop->bld_area[b] >= pop->n_areas) return NULL;
    for (uint32_t i = 0; i < pop->n_seeds; ++i)
        if (pop->seeds[i] >= pop->n_citizens) return NULL;

    orc_sim *s = (orc_sim *)calloc(1, sizeof(orc_sim));
    s->P = *p;
    s->n_cit = pop->n_citizens; s->n_bld = pop->n_buildings;
    s->n_area = pop->n_areas; s->n_room = pop->n_rooms;
    s->cit = (citizen_t *)calloc(s->n_cit ? s->n_cit : 1, sizeof(citizen_t));
    s->bld_area = (uint32_t *)malloc(sizeof(uint32_t) * (s->n_bld ? s->n_bld : 1));
    s->bld_type = (uint8_t *)malloc(s->n_bld ? s->n_bld : 1);
    s->room_bld = (uint32_t *)malloc(sizeof(uint32_t) * (s->n_room ? s->n_room : 1));
    memcpy(s->bld_area, pop->bld_area, sizeof(uint32_t) * s->n_bld);
    memcpy(s->bld_type, pop->bld_type, s->n_bld);
    memcpy(s->room_bld, pop->room_bld, sizeof(uint32_t) * s->n_room);

    uint8_t *has_work = (uint8_t *)malloc(s->n_cit ? s->n_cit : 1);
    uint8_t *has_room = (uint8_t *)malloc(s->n_cit ? s->n_cit : 1);
    uint32_t *room_key = (uint32_t *)malloc(sizeof(uint32_t) * (s->n_cit ? s->n_cit : 1));
    for (uint32_t c = 0; c < s->n_cit; ++c) {
        citizen_t *z = &s->cit[c];
        /* Citizen::new, citizen.rs:139-162: starts Susceptible, at home, off the bus */
        z->status = ORC_S; z->timer = 0;
        z->home = pop->home[c]; z->work = pop->work[c]; z->cur = z->home;
        z->uses_pt = (pop->flags[c] & ORC_FLAG_USES_PT) != 0;
        z->mask_ok = (pop->flags[c] & ORC_FLAG_MASK_COMPLIANT) != 0;
        z->bus = 0; z->at_work = 0; z->eligible = 0;
        has_work[c] = z->work != z->home;
        int school = has_work[c] && s->bld_type[z->work] == ORC_SCHOOL;
        z->room = school ? pop->room[c] : ORC_NO_ROOM;
        has_room[c] = (uint8_t)school;
        room_key[c] = school ? z->room : 0;
        /* school members are registered in rooms only (building.rs:404-431), other workers in
         * the Workplace occupant list (simulator_builder.rs:1076) */
        if (school) has_work[c] = 0;
    }
    /* apply_initial_infections, simulator_builder.rs:1139 */
    for (uint32_t i = 0; i < pop->n_seeds; ++i) {
        s->cit[pop->seeds[i]].status = ORC_I;
        s->cit[pop->seeds[i]].timer = 0;
    }
    build_csr(s->n_bld, s->n_cit, pop->home, NULL, &s->res_off, &s->res_idx);
    build_csr(s->n_bld, s->n_cit, pop->work, has_work, &s->wrk_off, &s->wrk_idx);
    build_csr(s->n_room, s->n_cit, room_key, has_room, &s->room_off, &s->room_idx);
    free(has_work); free(has_room); free(room_key);

    s->inf_head = (int32_t *)malloc(sizeof(int32_t) * (s->n_bld ? s->n_bld : 1));
    for (uint32_t b = 0; b < s->n_bld; ++b) s->inf_head[b] = -1;
    s->inf_cnt = (uint32_t *)calloc(s->n_bld ? s->n_bld : 1, sizeof(uint32_t));
    s->touched = (uint32_t *)malloc(sizeof(uint32_t) * (s->n_bld ? s->n_bld : 1));
    s->riders = (rider_t *)malloc(sizeof(rider_t) * (s->n_cit ? s->n_cit : 1));
    s->chosen_stamp = (uint32_t *)calloc(s->n_cit ? s->n_cit : 1, sizeof(uint32_t));
    s->time_step = 0; s->lockdown = 0; s->vaccination = 0; s->mask = ORC_MASK_NONE;
    s->have_eligible = 0; s->eligible_count = 0;
    return s;
}

void orc_destroy(orc_sim *s)
{
    if (!s) return;
    free(s->cit); free(s->bld_area); free(s->bld_type); free(s->room_bld);
    free(s->res_off); free(s->res_idx); free(s->wrk_off); free(s->wrk_idx);
    free(s->room_off); free(s->room_idx);
    free(s->inf_head); free(s->inf_cnt); free(s->touched); free(s->riders); free(s->inf_list); free(s->thr_counts);
    free(s->chosen_stamp);
    free(s);
}

/* sim/src/disease.rs:47-71 */
static void disease_tick(const orc_params *P, citizen_t *z)
{
    switch (z->status) {
    case ORC_E:
        if (P->exposed_time <= z->timer) { z->status = ORC_I; z->timer = 0; }
        else z->timer++;
        break;
    case ORC_I:
        if (P->infected_time <= z->timer) { z->status = ORC_R; z->timer = 0; }
        else z->timer++;
        break;
    default: break;
    }
}

/* sim/src/models/citizen.rs:168-216 */
static void citizen_execute_time_step(const orc_params *P, citizen_t *z, uint32_t hour, int lockdown)
{
    disease_tick(P, z);                                   /* citizen.rs:175 */
    if (!lockdown) {                                      /* citizen.rs:176 */
        uint32_t h = hour % 24;
        if (h == P->start_hour - 1 && z->uses_pt)         z->bus = 1;            /* :179-184 */
        else if (h == P->start_hour) { z->cur = z->work; z->at_work = 1; z->bus = 0; } /* :186-189 */
        else if (h == P->end_hour - 1 && z->uses_pt)      z->bus = 2;            /* :191-196 */
        else if (h == P->end_hour) { z->cur = z->home; z->at_work = 0; z->bus = 0; }   /* :198-201 */
        else                                              z->bus = 0;            /* :202-204 */
    }
}

/* sim/src/models/citizen.rs:221-248 (caller guarantees Susceptible, simulator.rs:337,436) */
static int citizen_expose(orc_sim *s, uint32_t c, uint64_t exposure_total, uint32_t slot)
{
    citizen_t *z = &s->cit[c];
    double q = q_literal(&s->P, exposure_total, z->mask_ok, z->bus != 0,
                         z->status == ORC_V, s->mask);
    if (z->status == ORC_S && uniform01(s->P.seed, c, s->time_step, slot) < q) {
        z->status = ORC_E; z->timer = 0;
        return 1;
    }
    return 0;
}

/* statistics.rs:275-287 */
static int citizen_exposed_stat(orc_record *r)
{
    if (r->susceptible == 0) return -1;
    r->susceptible--; r->exposed++;
    return 0;
}

static int rider_cmp(const void *a, const void *b)
{
    const rider_t *x = (const rider_t *)a, *y = (const rider_t *)b;
    if (x->src != y->src) return x->src < y->src ? -1 : 1;
    if (x->dst != y->dst) return x->dst < y->dst ? -1 : 1;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    if (x->id != y->id) return x->id < y->id ? -1 : 1;
    return 0;
}

/* Building exposure candidate (simulator.rs:308-350) */
static int try_building_exposure(orc_sim *s, orc_record *rec, uint32_t b, uint32_t m, uint64_t n)
{
    citizen_t *z = &s->cit[m];
    /* "If the Citizen is not currently in the Area, they haven't been exposed!" simulator.rs:324 */
    if (s->bld_area[z->cur] != s->bld_area[b]) return 0;
    if (z->status != ORC_S) return 0;                      /* is_susceptible(), :337 */
    uint32_t slot;
    if (b == z->home) slot = 0;
    else if (s->bld_type[b] == ORC_SCHOOL) slot = 16 + z->school_draws++;
    else slot = 1;
    if (citizen_expose(s, m, n, slot)) {
        rec->exposures_building++;
        z->exp_step = s->time_step; z->exp_area = s->bld_area[b];   /* ID::Building -> its Output Area too, statistics.rs:186-190 */
        if (citizen_exposed_stat(rec)) return -1;          /* add_exposure, :356-358 */
        /* area.citizens_eligible_for_vaccine is always None (output_area.rs:113) => no removal, Q10 */
    }
    return 0;
}

/* The per-citizen pass of generate_exposures on several host threads (the reference runs it under rayon,
 * simulator.rs:167-260).  Same result as the loop in orc_step: every citizen's tick is independent, the census is a sum,
 * and riders / Infected are collected in ascending citizen order exactly as the sequential loop meets them. */
static void generate_exposures_threaded(orc_sim *s, orc_record *rec, uint32_t hour, int lockdown)
{
    const orc_params *P = &s->P;
    const int T = s->threads;
    memset(s->thr_counts, 0, sizeof(uint64_t) * 8u * (size_t)T);
#ifdef _OPENMP
#pragma omp parallel num_threads(T)
#endif
    {
#ifdef _OPENMP
        const int t = omp_get_thread_num();
#else
        const int t = 0;
#endif
        const uint32_t lo = (uint32_t)((uint64_t)s->n_cit * (uint64_t)t / (uint64_t)T);
        const uint32_t hi = (uint32_t)((uint64_t)s->n_cit * (uint64_t)(t + 1) / (uint64_t)T);
        uint64_t k[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
        for (uint32_t c = lo; c < hi; ++c) {
            citizen_t *z = &s->cit[c];
            citizen_execute_time_step(P, z, hour, lockdown);
            z->school_draws = 0;
            switch (z->status) {
            case ORC_S: k[0]++; break;
            case ORC_E: k[1]++; break;
            case ORC_I: k[2]++; break;
            case ORC_R: k[3]++; break;
            default:    k[4]++; break;
            }
            if (z->bus) k[5]++;
            else if (z->status == ORC_I) k[6]++;
        }
        memcpy(s->thr_counts + 8u * (size_t)t, k, sizeof k);
#ifdef _OPENMP
#pragma omp barrier
#endif
        uint64_t r_off = 0, i_off = 0;
        for (int q = 0; q < t; ++q) { r_off += s->thr_counts[8u * (size_t)q + 5]; i_off += s->thr_counts[8u * (size_t)q + 6]; }
        for (uint32_t c = lo; c < hi; ++c) {
            const citizen_t *z = &s->cit[c];
            if (z->bus) {
                rider_t *r = &s->riders[r_off++];
                uint32_t ah = s->bld_area[z->home], aw = s->bld_area[z->work];
                r->src = z->bus == 1 ? ah : aw;
                r->dst = z->bus == 1 ? aw : ah;
                uint32_t w[4];
                philox_block(P->seed, c, s->time_step, 3, w);
                r->key = w[0]; r->id = c; r->infected = z->status == ORC_I;
            } else if (z->status == ORC_I) s->inf_list[i_off++] = c;
        }
    }
    uint64_t n_inf = 0;
    for (int q = 0; q < T; ++q) {
        const uint64_t *k = s->thr_counts + 8u * (size_t)q;
        rec->susceptible += (uint32_t)k[0]; rec->exposed += (uint32_t)k[1]; rec->infected += (uint32_t)k[2];
        rec->recovered += (uint32_t)k[3]; rec->vaccinated += (uint32_t)k[4];
        s->n_riders += (uint32_t)k[5]; n_inf += k[6];
    }
    for (uint64_t i = 0; i < n_inf; ++i) {                 /* :187-198, in the order the sequential loop meets them */
        const uint32_t c = s->inf_list[i];
        citizen_t *z = &s->cit[c];
        uint32_t b = z->cur;
        if (s->inf_cnt[b] == 0) s->touched[s->n_touched++] = b;
        s->inf_cnt[b]++;
        z->inf_next = s->inf_head[b]; s->inf_head[b] = (int32_t)c;
    }
}

int orc_set_threads(orc_sim *s, int threads)
{
#ifndef _OPENMP
    threads = 1;
#endif
    if (threads < 1) threads = 1;
    if (threads > 1 && !s->inf_list) {
        s->inf_list = (uint32_t *)malloc(sizeof(uint32_t) * (s->n_cit ? s->n_cit : 1));
        if (!s->inf_list) return -1;
    }
    free(s->thr_counts);
    s->thr_counts = (uint64_t *)calloc(8u * (size_t)threads, sizeof(uint64_t));
    if (!s->thr_counts) return -1;
    s->threads = threads;
    return threads;
}

int orc_step(orc_sim *s, orc_record *out)
{
    const orc_params *P = &s->P;
    orc_record rec;
    memset(&rec, 0, sizeof rec);
    /* statistics_recorder.next(), statistics.rs:156-171 */
    s->time_step += 1;
    rec.time_step = s->time_step;
    const uint32_t hour = s->time_step;                    /* simulator.rs:156 */
    const int lockdown = s->lockdown;                      /* simulator.rs:158 */

    /* ---- generate_exposures, simulator.rs:155-260 ---- */
    s->n_riders = 0; s->n_touched = 0;
    if (s->threads > 1) generate_exposures_threaded(s, &rec, hour, lockdown);
    else
    for (uint32_t c = 0; c < s->n_cit; ++c) {
        citizen_t *z = &s->cit[c];
        citizen_execute_time_step(P, z, hour, lockdown);   /* :175-177 */
        z->school_draws = 0;
        switch (z->status) {                               /* statistics.add_citizen, :178 */
        case ORC_S: rec.susceptible++; break;
        case ORC_E: rec.exposed++; break;
        case ORC_I: rec.infected++; break;
        case ORC_R: rec.recovered++; break;
        default:    rec.vaccinated++; break;
        }
        if (z->bus) {                                      /* :181-186 */
            rider_t *r = &s->riders[s->n_riders++];
            uint32_t ah = s->bld_area[z->home], aw = s->bld_area[z->work];
            r->src = z->bus == 1 ? ah : aw;
            r->dst = z->bus == 1 ? aw : ah;
            uint32_t w[4];
            philox_block(P->seed, c, s->time_step, 3, w);
            r->key = w[0]; r->id = c; r->infected = z->status == ORC_I;
        } else if (z->status == ORC_I) {                   /* :187-198 */
            uint32_t b = z->cur;
            if (s->inf_cnt[b] == 0) s->touched[s->n_touched++] = b;
            s->inf_cnt[b]++;
            z->inf_next = s->inf_head[b]; s->inf_head[b] = (int32_t)c;
        }
        /* moving between areas (:199-257) only re-buckets storage: the lookup entry always ends
         * up naming the area of current_building_position. */
    }
    rec.n_riders = s->n_riders;

    /* ---- apply_exposures: buildings, simulator.rs:268-358 ---- */
    int err = 0;
    for (uint32_t ti = 0; ti < s->n_touched && !err; ++ti) {
        uint32_t b = s->touched[ti];
        uint64_t n = s->inf_cnt[b];                        /* exposure_count, :307 */
        if (s->bld_type[b] == ORC_SCHOOL) {
            /* School::find_exposures, building.rs:494-522: one copy of the room per infected */
            for (int32_t i = s->inf_head[b]; i >= 0 && !err; i = s->cit[i].inf_next) {
                uint32_t r = s->cit[i].room;
                if (r == ORC_NO_ROOM || s->room_bld[r] != b) continue;   /* :499-506 */
                for (uint32_t k = s->room_off[r]; k < s->room_off[r + 1] && !err; ++k)
                    err = try_building_exposure(s, &rec, b, s->room_idx[k], n);
            }
        } else {
            /* Household / Workplace::find_exposures, building.rs:202-204,278-280: all occupants */
            for (uint32_t k = s->res_off[b]; k < s->res_off[b + 1] && !err; ++k)
                err = try_building_exposure(s, &rec, b, s->res_idx[k], n);
            for (uint32_t k = s->wrk_off[b]; k < s->wrk_off[b + 1] && !err; ++k)
                err = try_building_exposure(s, &rec, b, s->wrk_idx[k], n);
        }
    }
    for (uint32_t ti = 0; ti < s->n_touched; ++ti) {
        s->inf_cnt[s->touched[ti]] = 0; s->inf_head[s->touched[ti]] = -1;
    }
    if (err) return -1;

    /* ---- apply_exposures: public transport, simulator.rs:360-401 ---- */
    if (s->n_riders) {
        qsort(s->riders, s->n_riders, sizeof(rider_t), rider_cmp);
        uint32_t i = 0;
        while (i < s->n_riders && !err) {
            uint32_t j = i;
            while (j < s->n_riders && s->riders[j].src == s->riders[i].src &&
                   s->riders[j].dst == s->riders[i].dst) ++j;
            /* route [i,j): buses of bus_capacity, public_transport_route.rs:70-78 */
            for (uint32_t b0 = i; b0 < j && !err; b0 += P->bus_capacity) {
                uint32_t b1 = b0 + P->bus_capacity < j ? b0 + P->bus_capacity : j;
                uint64_t exposure_count = 0;
                for (uint32_t k = b0; k < b1; ++k) exposure_count += s->riders[k].infected;
                if (exposure_count == 0) continue;         /* :368,:389 */
                /* expose_citizens, simulator.rs:407-453 */
                for (uint32_t k = b0; k < b1 && !err; ++k) {
                    uint32_t m = s->riders[k].id;
                    if (s->cit[m].status == ORC_S && citizen_expose(s, m, exposure_count, 2)) {
                        rec.exposures_bus++;
                        s->cit[m].exp_step = s->time_step; s->cit[m].exp_area = ORC_NO_ROOM;   /* ID::PublicTransport */
                        if (citizen_exposed_stat(&rec)) err = -1;
                        if (s->have_eligible && s->cit[m].eligible) {   /* :447-449 */
                            s->cit[m].eligible = 0; s->eligible_count--;
                        }
                    }
                }
            }
            i = j;
        }
    }
    if (err) return -1;

    /* ---- apply_interventions, simulator.rs:455-556 ---- */
    uint32_t total = rec.susceptible + rec.exposed + rec.infected + rec.recovered + rec.vaccinated;
    double x = (double)rec.infected / (double)total;       /* statistics.rs:252-254 */
    int ev_vaccination = 0;
    /* InterventionStatus::update_status, interventions.rs:110-184 */
    if (P->lockdown_threshold < x) s->lockdown = 1;        /* :116-123 */
    else if (s->lockdown) s->lockdown = 0;                 /* :126-128 */
    if (P->vaccination_threshold < x) {                    /* :132-141 */
        if (!s->vaccination) { s->vaccination = 1; ev_vaccination = 1; }
    }
    switch (s->mask) {                                     /* :142-180 */
    case ORC_MASK_NONE:
        if (P->mask_pt_threshold < x) s->mask = ORC_MASK_PT;
        break;
    case ORC_MASK_PT:
        if (x < P->mask_pt_threshold) s->mask = ORC_MASK_NONE;
        else if (P->mask_everywhere_threshold < x) s->mask = ORC_MASK_EVERYWHERE;
        break;
    default:
        if (x < P->mask_everywhere_threshold) s->mask = ORC_MASK_PT;
        break;
    }
    /* Lockdown event body is a no-op (simulator.rs:462-480). */
    if (ev_vaccination) {                                  /* simulator.rs:481-513 */
        s->have_eligible = 1; s->eligible_count = 0;
        for (uint32_t c = 0; c < s->n_cit; ++c) {
            s->cit[c].eligible = s->cit[c].status == ORC_S;
            s->eligible_count += s->cit[c].eligible;
        }
    }
    uint32_t vaccinated_now = 0;
    if (s->have_eligible) {                                /* simulator.rs:524-553 */
        if (s->eligible_count <= P->vaccination_rate) {
            /* choose_multiple returns the whole set when it is not larger than `amount` */
            for (uint32_t c = 0; c < s->n_cit; ++c)
                if (s->cit[c].eligible) { s->cit[c].status = ORC_V; s->cit[c].timer = 0; vaccinated_now++; }
        } else {
            /* uniform k-subset by rejection over the candidate sequence (header contract) */
            uint32_t i = 0;
            while (vaccinated_now < P->vaccination_rate) {
                uint32_t w[4];
                philox_block(P->seed, i++, s->time_step, 4, w);
                uint64_t x64 = ((uint64_t)w[0] << 32) | w[1];
                uint32_t j = (uint32_t)(((unsigned __int128)x64 * s->n_cit) >> 64);
                if (!s->cit[j].eligible || s->chosen_stamp[j] == s->time_step) continue;
                s->chosen_stamp[j] = s->time_step;
                s->cit[j].status = ORC_V; s->cit[j].timer = 0;   /* unconditional, :551 (Q10) */
                vaccinated_now++;
            }
        }
    }
    rec.lockdown = (uint32_t)s->lockdown;
    rec.vaccination_active = (uint32_t)s->vaccination;
    rec.mask_status = (uint32_t)s->mask;
    rec.vaccinated_now = vaccinated_now;
    rec.eligible_count = s->eligible_count;
    /* disease_exists, statistics.rs:289-291 */
    rec.disease_exists = rec.exposed != 0 || rec.infected != 0 || rec.susceptible != 0;
    if (out) *out = rec;
    return 0;
}

int orc_run(orc_sim *s, uint32_t n, orc_record *out, int stop_when_done)
{
    uint32_t k = 0;
    for (; k < n; ++k) {                                   /* simulator.rs:114-123 */
        if (orc_step(s, &out[k])) return -1;
        if (stop_when_done && !out[k].disease_exists) { ++k; break; }
    }
    return (int)k;
}

/* The add_exposure calls so far (statistics.rs:181-195): per citizen the time step (0: never exposed by the simulation)
 * and the Output Area credited, ORC_NO_ROOM for an exposure on public transport. */
void orc_get_exposures(const orc_sim *s, uint32_t *step, uint32_t *area)
{
    for (uint32_t c = 0; c < s->n_cit; ++c) { step[c] = s->cit[c].exp_step; area[c] = s->cit[c].exp_area; }
}

void orc_get_state(const orc_sim *s, uint8_t *status, uint16_t *timer,
                   uint8_t *at_work, uint8_t *bus, uint8_t *eligible)
{
    for (uint32_t c = 0; c < s->n_cit; ++c) {
        const citizen_t *z = &s->cit[c];
        if (status) status[c] = z->status;
        if (timer) timer[c] = (z->status == ORC_E || z->status == ORC_I) ? z->timer : 0;
        if (at_work) at_work[c] = z->at_work;
        if (bus) bus[c] = z->bus;
        if (eligible) eligible[c] = z->eligible;
    }
}
