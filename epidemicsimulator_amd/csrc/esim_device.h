// esim_device.h -- device-resident layout shared by the kernels and the host side of libesim.
#pragma once
#include <stdint.h>

// ---- per-citizen word (uint32, one per citizen, HBM) ---------------------------------------
// bits 31..19  te : TE_BIAS + (time step at which the citizen became Exposed(0)), or a sentinel.
//                   DiseaseStatus (disease.rs:36-44) is a pure function of (current step - te):
//                   the E/I timers of disease.rs:47-71 never have to be written back.
// bit  18      bus_exposed : the exposure happened on public transport (the citizen leaves
//                   citizens_eligible_for_vaccine, simulator.rs:447-449)
// bits 17..11  vax : only inside a time-parallel chunk that runs under a vaccination programme: 127 - j when the citizen
//                   is set Vaccinated at the END of step j of the chunk (simulator.rs:524-553), 0 = not in this chunk.  The
//                   earliest such step wins by atomicMax; k_chunk_scatter turns it into te = TE_VACCINATED and clears it.
// bit  9       in_map : the persistent item map holds this citizen's records (set when it is entered, k_map_enter; meaningless while
//                   Ctrl::map_t does not equal the chunk's first step: a rebuild clears the bits of everybody it may concern)
// bit  10      plan_skip : the citizen is Susceptible and WILL be exposed on public transport in the step at ctrl->t (a chunk was
//                   cut there, k_chunk_vax): it leaves citizens_eligible_for_vaccine in that step, so the next plan must not
//                   choose it.  Any exposure drops the bit.
// bits 7..0    static flags (below)
// te sits in the most significant bits and Susceptible is the largest te, so "exposed at the earliest step
// at which any draw succeeds; a building beats a bus within a step" is one atomicMin on this word.
// Where a citizen stands (home / work / on a bus) is NOT per-citizen state: every citizen has
// the same working hours (citizen.rs:154-155), so current_building_position and
// on_public_transport are global functions of the clock and the lockdown history (Ctrl).
#define CW_TE_SHIFT    19u
#define CW_BUS_EXPOSED (1u << 18)
#define CW_FLAGS       0xFFu
#define CW_VAX_SHIFT   11u
#define CW_VAX_MASK    (0x7Fu << CW_VAX_SHIFT)
#define CW_KEEP        (CW_FLAGS | CW_VAX_MASK)            // what an exposure inside a chunk leaves as it is
#define CW_VAX_NONE    0xFFFFFFFFu
#define CW_PLAN_SKIP   (1u << 10)
#define CW_IN_MAP      (1u << 9)    // persistent item map (k_map_enter): the citizen's records have been entered into the items it stands in
// step of the chunk at whose end the citizen becomes Vaccinated (CW_VAX_NONE: not in this chunk)
#define CW_VAX_REL(w)  ((((w) >> CW_VAX_SHIFT) & 0x7Fu) ? 127u - (((w) >> CW_VAX_SHIFT) & 0x7Fu) : CW_VAX_NONE)
#define CW_VAX_FIELD(j) ((127u - (j)) << CW_VAX_SHIFT)
#define CW_TE(w)       ((w) >> CW_TE_SHIFT)
#define CW_MAKE(te, rest) (((te) << CW_TE_SHIFT) | (rest))
#define TE_SUSCEPTIBLE 0x1FFFu
#define TE_VACCINATED  0x1FFEu
#define TE_RECOVERED   0x1FFDu
#define TE_BIAS        512u          // >= exposed_time + infected_time + 2
#define TE_SLOTS       8192u
#define ESIM_MAX_STEP  7600u         // TE_BIAS + step must stay below TE_RECOVERED

// ---- static flags (low byte of the citizen word) -------------------------------------------
#define FL_USES_PT        0x01u      // ESIM_FLAG_USES_PUBLIC_TRANSPORT
#define FL_MASK_COMPLIANT 0x02u      // ESIM_FLAG_MASK_COMPLIANT
#define FL_SAME_AREA      0x04u      // area(work building) == area(home building)  (Q4, simulator.rs:324)
#define FL_WORK_SCHOOL    0x08u      // work building is a School (room draws, building.rs:494-522)
#define FL_HAS_WORK       0x10u      // workplace_code != household_code
#define FL_BIG_ROUTE      0x20u      // rides a route of more than 64 riders (ranked by a workgroup, not by a wavefront)

#define VACC_BATCH 4096u             // vaccination candidates examined per batch
#define VACC_WINDOW 32768u           // sharded runs: candidates per step whose liveness the shards exchange (eligible fraction >= rate / window)
#define VACC_TABLE 16384u            // LDS hash-set slots (>= max rate + VACC_BATCH, power of two)
#define VACC_MAX_RATE 8192u
#define NO_ROUTE 0xFFFFFFFFu

#define MARK_SLOTS 4

struct Ctrl {
    uint32_t t;                 // time step being processed (1-based; statistics.rs:167)
    uint32_t lockdown;          // InterventionStatus.lockdown.is_some() as decided at the end of step t-1
    uint32_t mask;              // mask_status in force during this step's exposures
    uint32_t vacc_active;       // InterventionStatus.vaccination.is_some()
    uint32_t have_elig;         // citizens_eligible_for_vaccine.is_some()
    uint32_t trigger_step;      // step whose end created the eligible set (simulator.rs:481-513)
    uint32_t elig_count;        // |citizens_eligible_for_vaccine| (this shard)
    uint32_t finished;          // disease_exists() was false and the run asked to stop
    uint32_t stop_when_done;
    uint32_t at_work;           // global position: 1 after the "starts work" arm (citizen.rs:186-189)
    uint32_t bus_dir;           // everyone using public transport: 0 off, 1 home->work, 2 work->home
    uint32_t steps_done;
    uint32_t error;             // sticky ESIM_E* (negated) raised on the device
    // census bookkeeping (this shard): Susceptible and Vaccinated citizens; the rest sits in hist[]
    uint32_t n_susceptible, n_vaccinated, n_recovered_sentinel;
    uint32_t log_len;           // entries in the exposure log
    // per-step work lists (zeroed by k_finish)
    // (ring of MARK_SLOTS by step: the marks of step t are cleared by the exposure pass of step t+1)
    uint32_t n_touched_bld[MARK_SLOTS], n_touched_room[MARK_SLOTS], n_touched_route[MARK_SLOTS], n_touched_route_big[MARK_SLOTS];
    uint32_t counts[5];         // census of the step in flight (global when sharded, after unpack)
    uint32_t n_riders;
    uint32_t free_base;         // first step of the current free-running batch (decoupled sharded mode)
    uint32_t small_done;        // steps executed by the last k_small launch
    uint32_t chunk_ok;          // steps of the current chunk that may run pipelined (k_decide)
    uint32_t chunk_t0;          // first step of the current chunk
    uint32_t chunk_i0, chunk_i1; // log entries [i0, i1): exposure steps that are Infected in some step of the chunk
    uint32_t items_per_wave;    // item ids every wavefront of k_chunk_marks owns
    uint32_t chunk_parallel;    // 1: the chunk's marks fit the hash map, all its steps can be drawn in one pass
    uint32_t chunk_pairs;       // log entries that are Infected in some step of the chunk, this shard (k_future)
    uint32_t n_items;           // marked (building | room | route, step) entries of the chunk
    uint32_t n_newexp;          // citizens exposed in the chunk
    uint32_t n_units, unit_next; // deferred units of the last chunk (diagnostics; the queues' own counters live in Dev::hot)
    uint32_t n_route_pairs;     // (unused: those pairs are counted per wavefront, Dev::pair_cnt)
    uint32_t n_route_pairs_big; // ... routes of more riders
    uint32_t chunk_done;        // the books of the last time-parallel chunk were written (k_chunk_books)
    uint32_t prev_t0, prev_n_items, prev_per_wave; // that chunk, for k_chunk_scatter
    // time-parallel chunks under a vaccination programme (k_chunk_vax)
    uint32_t vax_chunk;         // 1: the chunk in preparation has its vaccinations planned (events in Dev::vax_ev, fields in the words)
    uint32_t chunk_cut;         // first step of the chunk (relative) that must NOT be committed: a citizen exposed on a bus there had been
                                // planned for vaccination at or after it (it left the eligible set, simulator.rs:447-449) -- FREE_MAX + 1: none
    uint32_t need_seq;          // (unused: the step of a cut needs no special form, see CW_PLAN_SKIP)
    uint32_t prev_cut;          // the chunk k_chunk_scatter is finishing was cut at prev_n_eff
    uint32_t prev_n, prev_n_eff, prev_vax; // the chunk k_chunk_scatter is finishing: its length, the steps committed, whether it was planned
    uint32_t vax_cuts;          // diagnostics: chunks that were cut short
    uint32_t vax_planned, prev_planned; // steps the plan of the chunk in preparation / being finished covers (>= the chunk's length)
    uint32_t xs_need;           // sharded chunks: the most commuter records THIS shard saw in one segment of the chunk last prepared
    uint32_t xs_need_all;       // ... any shard did (summed by slot in the status exchange: the segments grow by the same factor everywhere)
    uint32_t vax_fail;          // sharded plans: steps whose candidates beyond the exchanged window would have been needed (plan void)
    // persistent item map: valid for a chunk that starts at step map_t -- every form that advances the clock
    // without maintaining the map (sequential steps, k_pipe chunks, a restore) leaves map_t behind and so invalidates it
    uint32_t map_t;
    uint32_t chunk_e0;          // log entries [chunk_e0, chunk_i1): exposure steps that BECOME Infected in some step of the chunk (k_decide)
    uint32_t n_neg;             // cancellation records appended for this chunk's plan (Dev::neg_list)
    uint32_t n_cancel;          // planned vaccinations of citizens the map holds (Dev::cancel_list), noted by k_chunk_vax_adj
    uint32_t pmap_chunk;        // the chunk in flight runs on the persistent map (k_map_enter decided)
    uint32_t prev_pmap;         // ... and the chunk k_chunk_scatter is finishing did
    uint32_t map_work;          // the map holds work buildings, rooms and routes too (it was built for a schedule with working hours)
    uint32_t err_where;         // diagnostics: which check raised `error` (ERR_AT_*), reported in esim_last_error
    uint32_t peer_error;        // sharded runs: the error fields of ALL shards, summed (ERR_FIELD): every rank takes its return code
                                // from this word, so that all leave esim_run_sharded together (k_status_unpack)
    uint32_t quiet;             // nobody was Exposed or Infected in the last step the chunk pass committed: the epidemic is over for good (nobody
                                // is infected from outside), whatever is left to simulate is the vaccination programme (k_chunk_books)
    uint32_t chunk_bus;         // steps of the chunk in preparation with riders on a bus (k_decide)
    uint32_t replan_from;       // first step of the chunk whose plan is walked again (k_chunk_lost; FREE_MAX + 1: none)
    uint32_t repair_ran;        // the plan of the chunk in flight was repaired: a cut it ends in is not one of round 2's kind (no CW_PLAN_SKIP marks)
    uint32_t vax_repairs;       // diagnostics: planned chunks whose plan was repaired after bus exposures instead of being cut (k_chunk_vax<true>)
};

// A rank's sticky error (1..6 = -ESIM_E*) as a 5-bit field of a word the shards SUM: field c-1 counts the shards that raised
// code c (at most 31 shards).  Every shard decodes the lowest code present, i.e. the same one.
#define ERR_FIELD(code) ((code) ? 1u << (5u * (((code) < 1u || (code) > 6u ? 6u : (code)) - 1u)) : 0u)
#define ERR_MAX_WORLD 31u
#define XE_WORDS (2u + ERR_MAX_WORLD)
static inline __host__ __device__ uint32_t err_decode(uint32_t word)
{
    for (uint32_t c = 1u; c <= 6u; ++c) if ((word >> (5u * (c - 1u))) & 31u) return c;
    return word ? 6u : 0u;
}

// A deferred unit of a long member list: UNIT_PAIRS (member, marked step) pairs from pair p_lo on.  code = kind << 30 | p_lo
// (UNIT_NOOP: skip); m_first = the member its first pair belongs to.
struct UnitRec { uint32_t slot, link, lo, n_mem, code, own, m_first, pad; };

// A building as k_chunk_marks needs it when it claims its item or spills a record: residents, workers, type, overflow stretch.
struct BldRec { uint32_t res_lo, res_hi, wrk_lo, wrk_hi, type, ovf_lo, ovf_hi, pad; };
#define SD_REPL 8u

// An item of a time-parallel chunk: a building (a = residents, b = workers, aux = type), a school room (a =
// participants, aux = its school building) or a route.
struct ItemRec { uint32_t id, a_lo, a_hi, b_lo, b_hi, aux, link, own; };   // link: a room's school item; own: the claimer's interval record
#define CHUNK_WAVES_MAX 4096u      // wavefronts of a chunk-pass kernel (1024 workgroups of 256)
#define SLOT_COUNTERS_ONLY 0x40000000u // slot_state of a school building: no interval records, everybody counted in `vec`
#ifndef ITEM_RECS
#define ITEM_RECS 7u               // interval records a slot holds (besides the claimer's, which travels in the item); 15 / 16 measured
#define SLOT_IV_STRIDE 8u          // 2 % slower on uk64m: the readers' loop over the records costs more than k_chunk_fold saves
#endif
#define LANE_STATE 24u             // lanes of a fetched item (fetch_item / fetch_slot): 0..7 its record, 8..8+ITEM_RECS-1 the slot's
#define LANE_HSLOT 25u             // interval records, then their number and the item's hash slot

// What is in force during one step of a pipelined chunk (k_decide fills dec[0..chunk_ok]).
struct Decision {
    uint32_t lockdown;          // InterventionStatus.lockdown.is_some() while the step runs
    uint32_t mask;              // mask_status while the step's exposures are drawn
    uint32_t at_work, bus_dir;  // global position / bus direction after the step's schedule arm
};

struct Dev {
    uint32_t n;                 // citizens on this shard
    uint32_t n_global;          // citizens over all shards
    uint32_t id_base;           // global index of local citizen 0
    uint32_t n_bld, n_room, n_pt;
    uint32_t *cit;              // per-citizen word (te | bus_exposed | flags)
    const uint32_t *home, *work, *room;
    // static membership lists (the reference's occupant lists: output_area.rs:172-180,
    // simulator_builder.rs:1076,1100, building.rs:404-431)
    const uint32_t *res_off, *res_idx;      // residents per building; res_idx == nullptr: citizens are home-sorted
    const uint32_t *wrk_off, *wrk_idx;      // workers per non-school building
    const uint32_t *room_off, *room_idx;    // participants per school room
    const uint32_t *room_bld;               // [n_room] school of each room
    const uint8_t  *bld_type;
    // marks of a step, one set per ring slot t % MARK_SLOTS
    uint32_t *cnt_bld[MARK_SLOTS];       // [n_bld] infected citizens standing in each building this step
    uint32_t *cnt_room[MARK_SLOTS];      // [n_room] ... in each school room
    uint32_t *touched_bld[MARK_SLOTS], *touched_room[MARK_SLOTS], *touched_route[MARK_SLOTS], *touched_route_big[MARK_SLOTS];
    uint32_t *route_flag[MARK_SLOTS];    // [n_routes]
    uint32_t *exp_step;         // [2 * (max_steps + 2)] successful exposures per step: [2t] buildings, [2t+1] buses
    uint32_t *exp_part;         // [EXP_ROWS][2 * FREE_MAX] the same for the steps of a chunk, by workgroup of k_chunk_count % EXP_ROWS
    struct Decision *dec;       // [FREE_MAX + 1]
    // time-parallel chunks: infected per (building | room | route, step of the chunk) in an open-addressing hash map
    unsigned long long *hkey;   // [hcap] slot id (building | n_bld + room | n_bld + n_room + route), HKEY_EMPTY when free
    uint32_t hcap;              // power of two
    uint32_t *hitems;           // [items_cap] hash slot of each item of the chunk (ITEM_UNUSED: id not handed out)
    struct ItemRec *item_rec;   // [items_cap] what the draw pass needs of an item, written at claim time
    uint32_t *slot_state;       // [hcap] buildings, rooms: interval records asked for (beyond ITEM_RECS they went into `vec`);
                                // routes: bit i = the i-th bus step of the chunk has been registered
    uint32_t *slot_iv;          // [hcap][SLOT_IV_STRIDE] interval records (k_chunk_marks: IV_*)
    uint32_t *vec;              // [hcap][FREE_MAX] per-step counts of the Infected that found no record free
    uint32_t items_cap;
    struct UnitRec *units;      // [SUBQ][unit_qcap] deferred units of long member lists
    uint32_t *route_pairs;      // [2 * items_cap] route << 7 | step of the chunk, routes of <= 64 riders: wavefront w of
                                // k_chunk_marks owns entries [w * 2 * items_per_wave, ...), pair_cnt[w] of them are filled
    uint32_t *pair_cnt;         // [wavefronts of k_chunk_marks]
    uint32_t *used_cnt;         // [wavefronts of k_chunk_marks] item ids the wavefront handed out
    uint32_t *route_pairs_big;  // [2 * items_cap] the same for longer routes, one shared list
    // interval records beyond a slot's ITEM_RECS: a building's go to ovf[ovf_off[b] ...) (room for one per resident and worker), a
    // room's to ovf[ovf_room_base + room_off[r] ...); k_chunk_fold sums them into `vec` before the draw pass
    const uint32_t *ovf_off;    // [n_bld + 1] res_off + wrk_off
    uint32_t ovf_room_base;     // ovf_off[n_bld]
    uint32_t *ovf;              // [ovf_n = ovf_room_base + room_off[n_room] + 1]
    uint32_t ovf_n;
    uint32_t n_wrk_idx, n_room_idx;   // lengths of wrk_idx / room_idx (what a member range read from a chunk table is checked against)
    // persistent item map: per-sub-list cursors of the slots listed for k_map_fold (those with records in `ovf`), the addresses of
    // this chunk's cancellation records (so that those of uncommitted steps can be taken back)
    uint32_t *pbig_cnt;         // [SUBQ]
    uint32_t *neg_list;         // [NEG_CAP][2] word index (bit 31: in `ovf`, else in `slot_iv`), step of the chunk
    uint32_t *cancel_list;      // [NEG_CAP][2] citizen, step of the chunk
    // school buildings of the persistent map: instead of one record per Infected member (hundreds in one list when the epidemic
    // sits in a few catchments), two histograms of exposure steps per school -- everybody, and those who ride public transport --
    // in rings of SCH_RING steps: how many members are Infected in a step is a window sum over them, like the census
    const int32_t *sch_of_bld;  // [n_bld] dense school index, -1: not a school
    uint32_t *sch_ring;         // [n_sch][2][SCH_RING]
    uint32_t n_sch;
    // school buildings of the per-chunk map: everybody Infected in one adds its stretch of the chunk to a DIFFERENCE array of the
    // school (+1 at its first step, -1 behind its last; a second array for those who ride public transport) instead of one
    // counter atomic per step; SD_REPL copies by adding wavefront, so that the hundreds of Infected of one school do not queue
    // on one address; k_chunk_fold sums them up into the slot's `vec` and zeroes them
    uint32_t *sch_diff;         // [n_sch][SD_REPL][2][FREE_MAX]
    // what k_chunk_marks needs of a citizen / of a building in one 16- / 32-byte record (one memory request instead of four / three)
    const uint4 *where4;        // [n] home building, work building, room, route (as home / work / room / route_of)
    const struct BldRec *bld8;  // [n_bld]
    uint32_t ovf_route_base;    // records of a route's Infected riders: ovf[2 * (ovf_route_base + route_off[r]) ...)
    uint32_t *big_list;         // [SUBQ][big_qcap][3] slots with records in `ovf`, where those start and how many fit, listed by the first
                                // to put one there; 64 lists by listing wavefront & 63, lengths in hot[HOT_BIG ...]
    uint32_t big_qcap;
    uint32_t *used_pref;        // [CHUNK_WAVES_MAX + 1] prefix sums of used_cnt (k_chunk_fold)
    uint32_t unit_qcap;
    uint32_t *newexp;           // [SUBQ][newexp_cap] citizens exposed in the chunk, by id & 63
    uint32_t newexp_cap;
    uint32_t *hot;              // [HOT_COUNT * HOT_STRIDE] the counters of the lists above
    uint32_t *cursor;           // [EXP_ROWS][FREE_MAX] write cursors into each step's stretch of the log, a row per EXP_ROWS-th workgroup
    uint32_t max_route;         // riders of the largest route
    // vaccination inside time-parallel chunks
    uint32_t *vax_ev;           // [FREE_MAX][VACC_MAX_RATE] citizens chosen in each step of the chunk, in candidate order
    uint32_t *vax_cnt;          // [FREE_MAX] how many (this shard's)
    uint32_t *vax_now;          // [FREE_MAX] vaccinated_now of the step's record (all shards)
    uint32_t *lost_list;        // [LOST_CAP] citizens whose bus exposure in this chunk found a planned vaccination in their word (expose_min)
    uint32_t *vax_delta;        // [4][FREE_MAX + 2] difference arrays over the chunk's steps: Susceptible, Exposed, Infected, Vaccinated
                                // census changes caused by the chunk's vaccinations (two's complement)
    uint32_t *xf_adj;           // [FREE_MAX + 2] difference array: Infected census ahead minus those vaccinated before
    uint32_t *hist;             // [TE_SLOTS] citizens per exposure time (census without a pass over citizens)
    uint32_t *log;              // exposure log: citizen ids in order of exposure step
    uint32_t *log_off;          // [TE_SLOTS + 1] first log entry whose te >= k
    const uint64_t *thr;        // [2][256] ceil(q * 2^32)
    Ctrl *ctrl;
    struct esim_step_result *records;   // [max_steps + 1]
    // public transport: static route lists (riders of a route share (home area, work area))
    uint32_t n_routes;
    const uint32_t *route_off;          // [n_routes + 1]
    const uint32_t *route_riders;       // local citizen ids, ascending inside a route
    const uint32_t *route_of;           // [n] route of a citizen using public transport, else NO_ROUTE
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
    uint32_t *xa, *xb, *xf;             // exchange buffers A, B and the future-infected vector
    // sharded time-parallel chunks: what the shards exchange once per chunk
    uint32_t rank, world;
    uint32_t *xv;                       // [XV_HEADER + FREE_MAX * PLAN_W / 32] liveness of every step's first PLAN_W vaccination candidates
    uint32_t *xs;                       // [world][1 + 3 * xs_cap] Infected commuters to shared buildings: (citizen word, shared building, shared room | -1)
    uint32_t xs_cap;                    // records per shard in this chunk's exchange
    // the commuter exchange as an all-to-all: a record goes to the shards that have members in its building, and to no other
    uint32_t *xs_out;                   // [world][1 + 3 * xs_cap] what this shard sends to each of the others (nullptr: the all-gather form)
    const uint32_t *shared_mask;        // [n_shared_bld] bit r: shard r has members in the shared building (summed at set-up)
    uint32_t *xc;                       // [FREE_MAX + 2] steps of the chunk with a cut (k_chunk_count)
    uint32_t *xl;                       // [FREE_MAX + 2] steps of the chunk in which a citizen of THIS shard left the eligible set on a bus with a planned
                                        // vaccination in its word (k_chunk_lost; summed: the shards walk the plan again from the same step)
    uint32_t *xe;                       // [XE_WORDS] status exchange: [0] error fields (ERR_FIELD), [1] shards that are finished, [2 + r] the most
                                        // commuter records shard r wanted to send to one destination in the chunk last prepared
    const int32_t *shared_of_bld, *shared_of_room;   // [n_bld], [n_room]: index into the shared tables, -1 when not shared
#ifdef ESIM_COUNT_WORK
    unsigned long long *work_cnt;       // counting build only: [WK_N] what the chunk pass worked on (tools/work_counts.py)
#endif
#ifdef ESIM_WAVE_PROFILE
    uint32_t *prof_buf;                 // diagnostics build only: [wavefronts][16] timers of the last chunk (tools/wave_profile.py)
#endif
    uint32_t xf_n;                      // steps buffer F covers (min(FREE_MAX, exposed_time + 1)); word xf[xf_n]: shards whose
                                        // chunk does not fit the one-pass form
};

// exchange buffer A: [0..4] census, [5] riders, then shared building counts, then shared room counts
#define XA_HEADER 8u
// exchange buffer B: [0] building exposures, [1] bus exposures, [2] eligible count, [3] error, then
// VACC_WINDOW/32 words of candidate liveness bits
#define XB_HEADER 8u
// exchange buffer F: Infected census of the next xf_n <= FREE_MAX steps, then one word counting the shards that cannot
// draw the chunk in one pass
#define FREE_MAX 96u
#define HKEY_EMPTY 0xFFFFFFFFFFFFFFFFull
#define PLAN_W 4096u               // sharded plans: candidates per step whose liveness is exchanged (one batch)
#define XV_HEADER 8u               // [0] eligible count, [1] riders, [2] shards that cannot plan -- summed over the shards
#define XS_CAP_MAX (1u << 20)      // commuter records a shard can send per chunk at most (the segment in use, Dev::xs_cap, grows with the need)
// Shared lists of the chunk pass are split so that no single address takes more than a few atomics per pass (atomics on
// one address are served one at a time, ~10 ns each): every counter has a 128-byte line of its own in Dev::hot.
#define HOT_STRIDE 32u             // uint32 per counter
#define SUBQ 64u                   // sub-lists per shared list
#define HOT_NEWEXP 0u              // [SUBQ] citizens exposed in the chunk, by citizen id & 63
#define HOT_UNITS 64u              // [SUBQ] deferred units, by producing wavefront & 63
#define HOT_BIGPAIRS 128u          // (route, bus step) pairs of routes with more than 64 riders
#define HOT_BIG 129u               // [SUBQ] slots listed for k_chunk_fold, by listing wavefront & 63
#define HOT_RPAIRS 193u            // [SUBQ] persistent map: (route of <= 64 riders, bus step) pairs k_chunk_draw registered for k_chunk_units, by wavefront & 63
#define HOT_PREV_NEWEXP 257u       // [SUBQ] copy of HOT_NEWEXP of the chunk whose log entries k_chunk_scatter is writing
#define HOT_RESET 257u             // counters k_decide zeroes for a new chunk
#define HOT_LOST 321u               // citizens exposed on a bus while the chunk's plan had a vaccination for them (Dev::lost_list; zeroed by k_chunk_vax)
#define HOT_COUNT 322u
// where a device-side error was raised (Ctrl::err_where)
#define RAISE(ctrl, code, where) do { (ctrl)->error = (uint32_t)(-(code)); (ctrl)->err_where = (where); } while (0)
enum { ERR_AT_OVF_FULL = 1, ERR_AT_BIG_LIST, ERR_AT_NEG_LIST, ERR_AT_ITEM_IDS, ERR_AT_HASH_FULL, ERR_AT_ITEM_CHECK, ERR_AT_ROUTE_ITEM, ERR_AT_MAP_STATE,
       ERR_AT_CANCEL_SLOT, ERR_AT_BIGPAIRS, ERR_AT_SCHOOL };
#define SCH_RING 1024u             // >= exposed_time + infected_time + 2 + 2 * FREE_MAX
#define PBIG_STRIDE 4u             // words per entry of the persistent map's fold list: slot, overflow base, capacity | school flag, school
#define NEG_CAP (1u << 18)          // cancellation records per chunk (a chunk plans at most 96 x 8192 vaccinations, few of them of Infected citizens)
#define UNIT_NOOP 0xFFFFFFFFu
#ifndef LOST_CAP
#define LOST_CAP 8192u              // entries of Dev::lost_list (more: k_chunk_lost looks at everybody exposed in the chunk instead; -DLOST_CAP=2 tested)
#endif
#define CHUNK_BUS_STEPS 32u        // a one-pass chunk has at most this many steps with riders on a bus (a route item keeps a bit per such step;
                                   // round 3: 8 -- a lockdown that froze the riders on a bus made every chunk 8 steps long)
// entries a wavefront of k_chunk_marks has for its (route, bus step) pairs: two per item id it owns, more when the chunk has more than 8 bus steps
#define PAIR_K(per_wave, nbus) ((per_wave) * ((nbus) > 8u ? ((nbus) + 3u) / 4u : 2u))
#define COUNT_GRID 256u            // workgroups of k_chunk_count
#define EXP_ROWS 32u
#ifndef UNIT_INLINE
#define UNIT_INLINE 1024u          // the longest list (in pairs) a wavefront of k_chunk_draw draws itself
#endif
#ifndef UNIT_PAIRS
#define UNIT_PAIRS 1024u           // (member, marked step) pairs per deferred unit of a long member list, and the longest list a wavefront of
                                   // k_chunk_draw draws itself (128 ... 16384 measured: DESIGN.md 3.9, 3.13 viii)
#endif
#define CHUNK_ROUTE_MAX 2048u      // routes up to this many riders are ranked in LDS by the time-parallel pass
