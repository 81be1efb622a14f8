// esim_device.h -- device-resident layout shared by the kernels and the host side of libesim.
#pragma once
#include <stdint.h>

// ---- per-citizen dynamic state word (uint16, one per citizen, HBM) -------------------------
// bits 0..12  te : BIAS + (time step at which the citizen became Exposed(0)), or a sentinel.
//                  DiseaseStatus (disease.rs:36-44) is a pure function of (current step - te):
//                  the E/I timers of disease.rs:47-71 never have to be written back.
// bit  13     at_work : current_building_position == workplace_code (citizen.rs:127,187,199)
// bit  14     on_bus  : on_public_transport.is_some() (citizen.rs:134)
// bit  15     eligible: member of citizens_eligible_for_vaccine (simulator.rs:97)
#define ST_TE_MASK   0x1FFFu
#define ST_AT_WORK   0x2000u
#define ST_ON_BUS    0x4000u
#define ST_ELIGIBLE  0x8000u
#define TE_SUSCEPTIBLE 0x1FFFu
#define TE_VACCINATED  0x1FFEu
#define TE_RECOVERED   0x1FFDu
#define TE_BIAS        512u          // >= exposed_time + infected_time + 2
#define ESIM_MAX_STEP  7600u         // TE_BIAS + step must stay below TE_RECOVERED

// ---- per-citizen static flags (uint8) ------------------------------------------------------
#define FL_USES_PT        0x01u      // ESIM_FLAG_USES_PUBLIC_TRANSPORT
#define FL_MASK_COMPLIANT 0x02u      // ESIM_FLAG_MASK_COMPLIANT
#define FL_SAME_AREA      0x04u      // area(work building) == area(home building)  (Q4, simulator.rs:324)
#define FL_WORK_SCHOOL    0x08u      // work building is a School (room draws, building.rs:494-522)
#define FL_HAS_WORK       0x10u      // workplace_code != household_code

#define VACC_BATCH 4096u             // vaccination candidates examined per batch
#define VACC_TABLE 16384u            // LDS hash-set slots (>= max rate + VACC_BATCH, power of two)
#define VACC_MAX_RATE 8192u

struct Ctrl {
    uint32_t t;                 // time step being processed (1-based; statistics.rs:167)
    uint32_t lockdown;          // InterventionStatus.lockdown.is_some() as decided at the end of step t-1
    uint32_t mask;              // mask_status in force during this step's exposures
    uint32_t vacc_active;       // InterventionStatus.vaccination.is_some()
    uint32_t have_elig;         // citizens_eligible_for_vaccine.is_some()
    uint32_t elig_count;        // |citizens_eligible_for_vaccine| (this shard until exchanged)
    uint32_t finished;          // disease_exists() was false and the run asked to stop
    uint32_t stop_when_done;
    uint32_t bus_dir;           // direction of everyone currently on a bus: 1 home->work, 2 work->home
    uint32_t steps_done;
    uint32_t error;             // sticky ESIM_E* (negated) raised on the device
    // accumulators of the step in flight (zeroed by k_finish)
    uint32_t counts[5];         // S,E,I,R,V census after Citizen::execute_time_step (simulator.rs:178)
    uint32_t n_riders;
    uint32_t exp_bld, exp_bus;
    uint32_t pad[12];
};

struct Dev {
    uint32_t n;                 // citizens on this shard
    uint32_t n_global;          // citizens over all shards
    uint32_t id_base;           // global index of local citizen 0
    uint32_t n_bld, n_room;
    uint16_t *state;
    const uint8_t  *flags;
    const uint32_t *home, *work, *room;
    uint32_t *cnt_bld;          // [n_bld] infected citizens standing in each building this step
    uint32_t *cnt_room;         // [n_room] ... in each school room
    const uint64_t *thr;        // [2][256] ceil(q * 2^53)
    Ctrl *ctrl;
    struct esim_step_result *records;   // [max_steps + 1]
    // public transport: static route lists (riders of a route share (home area, work area))
    uint32_t n_routes_small, n_routes_big;
    const uint32_t *route_small;        // route ids with <= 64 riders
    const uint32_t *route_big;
    const uint32_t *route_off;          // [n_routes + 1]
    const uint32_t *route_riders;       // local citizen ids, ascending inside a route
    uint32_t *bus_key; uint32_t *bus_idx; uint32_t *bus_cnt; uint8_t *bus_flag;   // scratch for big routes
    // parameters
    uint32_t exposed_time, infected_time, vaccination_rate, bus_capacity, start_hour, end_hour;
    uint32_t seed_lo, seed_hi;
    double thr_lockdown, thr_vacc, thr_mask_pt, thr_mask_all;
    uint32_t max_steps;
    // sharding
    uint32_t n_shards;
    uint32_t n_shared_bld, n_shared_room;
    const int32_t *shared_bld, *shared_room;
    uint32_t *xa, *xb;                  // exchange buffers A and B
};

// exchange buffer A: [0..4] census, [5] riders, then shared building counts, then shared room counts
#define XA_HEADER 8u
// exchange buffer B: [0] building exposures, [1] bus exposures, [2] eligible count, [3] error, then
// VACC_BATCH/32 words of candidate liveness bits
#define XB_HEADER 8u
