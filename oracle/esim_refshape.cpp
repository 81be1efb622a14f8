/*
 * esim_refshape.cpp -- the REFERENCE-SHAPED CPU path (test infrastructure / bench.py's cpu_baseline; see esim_oracle.h).
 *
 * Same semantics and the same Philox contract as esim_oracle.c (so its records must equal the oracle's, which
 * tests/test_refshape.py checks), but with the data structures and the parallel shape of the reference's own
 * implementation, so that it can stand beside the GPU path as "what the reference does on this host":
 *
 *   reference (sim/src/simulator.rs)                              here
 *   output_areas: RwLock<Vec<Mutex<OutputArea>>>          :94     std::vector<Area>, a std::mutex per area
 *   OutputArea.citizens: Vec<Citizen>, array of structs            Area::citizens (Citizen carries its three BuildingIDs,
 *     of ~280 B with three BuildingIDs     citizen.rs:110-135       uuids included, like the reference's)
 *   OutputArea.buildings: Vec<Box<dyn Building>>                   Area::buildings with occupant lists / School rooms and
 *     Household / Workplace / School  building.rs:162-522           the occupant -> room hash map
 *   citizen_output_area_lookup: RwLock<Vec<Mutex<(area, idx)>>> :96  std::vector<LookupEntry>, a mutex per citizen
 *   generate_exposures: par_iter_mut over areas, drain and       threads take areas from a shared counter (rayon's work
 *     re-push every Citizen, a Mutex lock per citizen,            stealing); every citizen is moved into a fresh vector and
 *     per-area HashMaps of infected per building and riders        its lookup entry rewritten under its mutex; per-thread
 *     per route, reduced pairwise                   :167-229       hash maps merged afterwards
 *   serial cross-area move + lookup rewrite         :231-257       serial
 *   apply_exposures: par_iter over areas, lookup lock per          the same
 *     candidate, area filter, Citizen::expose       :268-355
 *   serial add_exposure, serial bus loop            :356-401       serial
 *   apply_interventions: serial, three locks per vaccinee :455-556 serial
 *
 * One thing is NOT reproduced: the reference allocates vec![Vec::new(); n_areas] per area per step (:172) and reduces
 * vectors of that length pairwise (:218-229), which is O(areas^2) per step and would not finish a single step at 290 000
 * areas; here the citizens that change area go into per-thread lists.  The baseline is therefore kinder to the CPU than
 * the reference is to itself.
 */
#include "esim_oracle.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <unordered_set>
#include <vector>

namespace {

struct BuildingID {             // building.rs:62-67
    uint32_t area, index;       // output_area_id.index, building_index
    uint8_t type;
    uint8_t uuid[16];
};

struct Citizen {                // citizen.rs:110-135
    uint8_t uuid[16];
    uint32_t global_index;
    uint16_t age;
    BuildingID household_code, workplace_code, current_building_position;
    uint8_t occupation, start_working_hour, end_working_hour;
    uint8_t status; uint16_t timer;                  // DiseaseStatus
    bool is_mask_compliant, uses_public_transport;
    bool on_bus; uint32_t bus_src, bus_dst;          // on_public_transport: Option<(OutputAreaID, OutputAreaID)>
    uint32_t room;                                   // index into School::rooms (occupant_to_class), or ORC_NO_ROOM
    uint32_t school_draws;
    bool eligible;
};

struct Building {               // Household / Workplace / School, building.rs:162-522
    uint8_t type;
    std::vector<uint32_t> occupants;                     // Household, Workplace
    std::vector<std::vector<uint32_t>> rooms;            // School: classes then offices
    std::unordered_map<uint32_t, uint32_t> occupant_to_room;
};

struct Area {
    std::mutex m;
    std::vector<Citizen> citizens;
    std::vector<Building> buildings;
    std::unordered_map<uint32_t, std::vector<uint32_t>> exposure_list;   // building index -> infected ids, building_exposure_list[area]
};

struct LookupEntry { std::mutex m; uint32_t area, index; };

struct Rider { uint32_t key, id; bool infected; };

void philox(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t out[4])
{
    const uint32_t ctr[4] = { c0, c1, c2, 0u }, key[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) };
    orc_philox4x32_10(ctr, key, out);
}

}  // namespace

struct rsh_sim {
    orc_params P;
    uint32_t n_cit = 0, n_area = 0;
    std::vector<Area> areas;
    std::vector<LookupEntry> lookup;
    std::vector<uint32_t> bld_area_of, bld_local;        // global building -> (area, index)
    int threads = 1;
    uint32_t time_step = 0;
    bool lockdown = false, vaccination = false, have_eligible = false;
    int mask = ORC_MASK_NONE;
    std::unordered_set<uint32_t> eligible;               // citizens_eligible_for_vaccine: Option<HashSet<CitizenID>>
    std::vector<uint32_t> chosen_stamp;
    std::vector<uint32_t> touched_areas;
    rsh_sim() {}
};

static bool expose(rsh_sim *s, Citizen &z, uint64_t exposure_total, uint32_t slot)
{
    // Citizen::expose, citizen.rs:221-248 (Q7: a compliant citizen is evaluated with MaskStatus::None)
    const int mask = z.is_mask_compliant ? ORC_MASK_NONE : s->mask;
    const double chance = orc_exposure_chance(&s->P, z.status == ORC_V, mask, z.is_mask_compliant && z.on_bus);
    const double q = orc_binomial(chance, (uint8_t)exposure_total);
    if (z.status == ORC_S && (double)orc_u32(s->P.seed, z.global_index, s->time_step, slot) * 0x1.0p-32 < q) {
        z.status = ORC_E; z.timer = 0;
        return true;
    }
    return false;
}

extern "C" rsh_sim *rsh_create(const orc_params *p, const orc_population *pop, int threads)
{
    rsh_sim *s = new rsh_sim();
    s->P = *p;
    s->n_cit = pop->n_citizens; s->n_area = pop->n_areas;
    s->threads = threads < 1 ? 1 : threads;
    s->areas = std::vector<Area>(pop->n_areas);
    s->lookup = std::vector<LookupEntry>(pop->n_citizens);
    s->chosen_stamp.assign(pop->n_citizens, 0);
    s->bld_area_of.assign(pop->bld_area, pop->bld_area + pop->n_buildings);
    s->bld_local.resize(pop->n_buildings);
    for (uint32_t b = 0; b < pop->n_buildings; ++b) {
        Area &a = s->areas[pop->bld_area[b]];
        s->bld_local[b] = (uint32_t)a.buildings.size();
        a.buildings.emplace_back();
        a.buildings.back().type = pop->bld_type[b];
    }
    // rooms of each school, in global room order
    std::vector<uint32_t> room_local(pop->n_rooms);
    for (uint32_t r = 0; r < pop->n_rooms; ++r) {
        Building &sch = s->areas[pop->bld_area[pop->room_bld[r]]].buildings[s->bld_local[pop->room_bld[r]]];
        room_local[r] = (uint32_t)sch.rooms.size();
        sch.rooms.emplace_back();
    }
    auto make_id = [&](uint32_t b) {
        BuildingID id; std::memset(&id, 0, sizeof id);
        id.area = pop->bld_area[b]; id.index = s->bld_local[b]; id.type = pop->bld_type[b];
        std::memcpy(id.uuid, &b, sizeof b);
        return id;
    };
    for (uint32_t c = 0; c < pop->n_citizens; ++c) {
        Citizen z; std::memset(&z, 0, sizeof z);
        std::memcpy(z.uuid, &c, sizeof c);
        z.global_index = c;
        z.household_code = make_id(pop->home[c]); z.workplace_code = make_id(pop->work[c]);
        z.current_building_position = z.household_code;                       // Citizen::new, citizen.rs:139-162
        z.start_working_hour = (uint8_t)p->start_hour; z.end_working_hour = (uint8_t)p->end_hour;
        z.status = ORC_S;
        z.uses_public_transport = (pop->flags[c] & ORC_FLAG_USES_PT) != 0;
        z.is_mask_compliant = (pop->flags[c] & ORC_FLAG_MASK_COMPLIANT) != 0;
        z.room = ORC_NO_ROOM;
        const bool has_work = pop->work[c] != pop->home[c];
        Area &ah = s->areas[pop->bld_area[pop->home[c]]];
        ah.buildings[s->bld_local[pop->home[c]]].occupants.push_back(c);      // Household::add_citizen, output_area.rs:172-180
        if (has_work) {
            Building &wb = s->areas[pop->bld_area[pop->work[c]]].buildings[s->bld_local[pop->work[c]]];
            if (wb.type == ORC_SCHOOL) {                                      // School::with_students_and_teachers, building.rs:404-431
                const uint32_t r = room_local[pop->room[c]];
                wb.rooms[r].push_back(c); wb.occupant_to_room[c] = r; z.room = r;
            } else wb.occupants.push_back(c);                                 // Workplace::add_citizen, simulator_builder.rs:1076
        }
        s->lookup[c].area = pop->bld_area[pop->home[c]];
        s->lookup[c].index = (uint32_t)ah.citizens.size();
        ah.citizens.push_back(z);
    }
    for (uint32_t i = 0; i < pop->n_seeds; ++i) {                             // apply_initial_infections, simulator_builder.rs:1139
        const LookupEntry &e = s->lookup[pop->seeds[i]];
        s->areas[e.area].citizens[e.index].status = ORC_I;
        s->areas[e.area].citizens[e.index].timer = 0;
    }
    return s;
}

extern "C" void rsh_destroy(rsh_sim *s) { delete s; }

namespace {

template <class F> void parallel_over(int threads, uint32_t n, F fn)
{
    std::atomic<uint32_t> next{ 0 };
    auto body = [&](int t) {
        for (;;) {
            const uint32_t i0 = next.fetch_add(16u);                      // work stealing in grains of 16 areas
            if (i0 >= n) break;
            for (uint32_t i = i0; i < std::min(n, i0 + 16u); ++i) fn(t, i);
        }
    };
    if (threads <= 1) { body(0); return; }
    std::vector<std::thread> th;
    for (int t = 0; t < threads; ++t) th.emplace_back(body, t);
    for (auto &x : th) x.join();
}

struct ThreadOut {
    uint32_t counts[5] = { 0, 0, 0, 0, 0 };
    std::unordered_map<uint64_t, std::vector<Rider>> pt;                  // (src area, dst area) -> riders
    std::unordered_map<uint64_t, std::vector<uint32_t>> bexp;             // (area, building index) -> infected ids
    std::vector<std::pair<uint32_t, Citizen>> moving;
    std::vector<std::pair<uint32_t, uint8_t>> exposures;                  // apply_exposures: (area credited, 0) per success
};

}  // namespace

extern "C" int rsh_step(rsh_sim *s, orc_record *out)
{
    const orc_params *P = &s->P;
    orc_record rec; std::memset(&rec, 0, sizeof rec);
    s->time_step += 1; rec.time_step = s->time_step;                      // statistics_recorder.next()
    const uint32_t hour = s->time_step;
    const bool lockdown = s->lockdown;
    std::vector<ThreadOut> tout((size_t)s->threads);

    // ---- generate_exposures (simulator.rs:155-260)
    parallel_over(s->threads, s->n_area, [&](int t, uint32_t ai) {
        Area &area = s->areas[ai];
        std::lock_guard<std::mutex> g(area.m);
        ThreadOut &o = tout[(size_t)t];
        std::vector<Citizen> kept;
        kept.reserve(area.citizens.size());
        for (Citizen &z : area.citizens) {                                // drain(0..)
            // Citizen::execute_time_step, citizen.rs:168-216
            uint32_t old_area = z.current_building_position.area;
            if (z.status == ORC_E) { if (P->exposed_time <= z.timer) { z.status = ORC_I; z.timer = 0; } else z.timer++; }
            else if (z.status == ORC_I) { if (P->infected_time <= z.timer) { z.status = ORC_R; z.timer = 0; } else z.timer++; }
            if (!lockdown) {
                const uint32_t h = hour % 24;
                if (h == (uint32_t)z.start_working_hour - 1 && z.uses_public_transport) { z.on_bus = true; z.bus_src = z.household_code.area; z.bus_dst = z.workplace_code.area; }
                else if (h == z.start_working_hour) { z.current_building_position = z.workplace_code; z.on_bus = false; }
                else if (h == (uint32_t)z.end_working_hour - 1 && z.uses_public_transport) { z.on_bus = true; z.bus_src = z.workplace_code.area; z.bus_dst = z.household_code.area; }
                else if (h == z.end_working_hour) { z.current_building_position = z.household_code; z.on_bus = false; }
                else z.on_bus = false;
            }
            z.school_draws = 0;
            const bool need_to_move = z.current_building_position.area != old_area;
            o.counts[z.status]++;                                         // statistics.add_citizen
            if (z.on_bus) {
                uint32_t w[4]; philox(P->seed, z.global_index, s->time_step, 3, w);
                o.pt[((uint64_t)z.bus_src << 32) | z.bus_dst].push_back(Rider{ w[0], z.global_index, z.status == ORC_I });
            } else if (z.status == ORC_I) {
                o.bexp[((uint64_t)z.current_building_position.area << 32) | z.current_building_position.index].push_back(z.global_index);
            }
            if (need_to_move) o.moving.emplace_back(z.current_building_position.area, z);
            else {
                LookupEntry &e = s->lookup[z.global_index];
                std::lock_guard<std::mutex> lg(e.m);
                e.area = ai; e.index = (uint32_t)kept.size();
                kept.push_back(z);
            }
        }
        area.citizens.swap(kept);
    });
    // reduce (:218-229) and the serial cross-area move (:231-257)
    std::unordered_map<uint64_t, std::vector<Rider>> pt_all;
    s->touched_areas.clear();
    for (ThreadOut &o : tout) {
        rec.susceptible += o.counts[ORC_S]; rec.exposed += o.counts[ORC_E]; rec.infected += o.counts[ORC_I];
        rec.recovered += o.counts[ORC_R]; rec.vaccinated += o.counts[ORC_V];
        for (auto &kv : o.pt) { auto &dst = pt_all[kv.first]; dst.insert(dst.end(), kv.second.begin(), kv.second.end()); }
        for (auto &kv : o.bexp) {
            Area &a = s->areas[(uint32_t)(kv.first >> 32)];
            if (a.exposure_list.empty()) s->touched_areas.push_back((uint32_t)(kv.first >> 32));
            auto &dst = a.exposure_list[(uint32_t)kv.first];
            dst.insert(dst.end(), kv.second.begin(), kv.second.end());
        }
        for (auto &mv : o.moving) {
            Area &a = s->areas[mv.first];
            std::lock_guard<std::mutex> g(a.m);
            LookupEntry &e = s->lookup[mv.second.global_index];
            e.area = mv.first; e.index = (uint32_t)a.citizens.size();
            a.citizens.push_back(mv.second);
        }
        o.pt.clear(); o.bexp.clear(); o.moving.clear();
    }
    for (auto &kv : pt_all) rec.n_riders += (uint32_t)kv.second.size();

    // ---- apply_exposures: buildings (simulator.rs:268-358)
    parallel_over(s->threads, (uint32_t)s->touched_areas.size(), [&](int t, uint32_t k) {
        const uint32_t ai = s->touched_areas[k];
        Area &area = s->areas[ai];
        std::lock_guard<std::mutex> g(area.m);
        ThreadOut &o = tout[(size_t)t];
        for (auto &kv : area.exposure_list) {
            const Building &b = area.buildings[kv.first];
            const uint64_t exposure_count = kv.second.size();
            std::vector<uint32_t> cand;                                   // Building::find_exposures
            if (b.type == ORC_SCHOOL) {
                for (uint32_t inf : kv.second) {                          // building.rs:494-522
                    auto it = b.occupant_to_room.find(inf);
                    if (it == b.occupant_to_room.end()) continue;
                    cand.insert(cand.end(), b.rooms[it->second].begin(), b.rooms[it->second].end());
                }
            } else cand = b.occupants;                                    // building.rs:202-204,278-280: a clone of the list
            for (uint32_t cid : cand) {
                uint32_t la, li;
                { LookupEntry &e = s->lookup[cid]; std::lock_guard<std::mutex> lg(e.m); la = e.area; li = e.index; }
                if (la != ai) continue;                                   // "not currently in the Area", :324
                Citizen &z = area.citizens[li];
                if (z.status != ORC_S) continue;
                uint32_t slot;
                if (z.household_code.area == ai && z.household_code.index == kv.first) slot = 0;
                else if (b.type == ORC_SCHOOL) slot = 16 + z.school_draws++;
                else slot = 1;
                if (expose(s, z, exposure_count, slot)) o.exposures.emplace_back(ai, (uint8_t)0);
            }
        }
        area.exposure_list.clear();
    });
    int err = 0;
    for (ThreadOut &o : tout)
        for (size_t i = 0; i < o.exposures.size(); ++i) {                 // add_exposure, :356-358; statistics.rs:275-287
            rec.exposures_building++;
            if (rec.susceptible == 0) err = -1; else { rec.susceptible--; rec.exposed++; }
        }
    if (err) return -1;

    // ---- apply_exposures: public transport, serial (simulator.rs:360-453)
    for (auto &kv : pt_all) {
        std::vector<Rider> &r = kv.second;
        std::sort(r.begin(), r.end(), [](const Rider &a, const Rider &b) { return a.key != b.key ? a.key < b.key : a.id < b.id; });
        for (size_t b0 = 0; b0 < r.size(); b0 += P->bus_capacity) {
            const size_t b1 = std::min(r.size(), b0 + P->bus_capacity);
            uint64_t exposure_count = 0;
            for (size_t k = b0; k < b1; ++k) exposure_count += r[k].infected;
            if (!exposure_count) continue;
            for (size_t k = b0; k < b1; ++k) {                            // expose_citizens, :407-453
                uint32_t la, li;
                { LookupEntry &e = s->lookup[r[k].id]; std::lock_guard<std::mutex> lg(e.m); la = e.area; li = e.index; }
                Area &a = s->areas[la];
                std::lock_guard<std::mutex> g(a.m);
                Citizen &z = a.citizens[li];
                if (z.status == ORC_S && expose(s, z, exposure_count, 2)) {
                    rec.exposures_bus++;
                    if (rec.susceptible == 0) return -1;
                    rec.susceptible--; rec.exposed++;
                    if (s->have_eligible) s->eligible.erase(z.global_index);      // :447-449
                }
            }
        }
    }

    // ---- apply_interventions, serial (simulator.rs:455-556; interventions.rs:110-184)
    const uint32_t total = rec.susceptible + rec.exposed + rec.infected + rec.recovered + rec.vaccinated;
    const double x = (double)rec.infected / (double)total;
    bool ev_vaccination = false;
    if (P->lockdown_threshold < x) s->lockdown = true; else if (s->lockdown) s->lockdown = false;
    if (P->vaccination_threshold < x && !s->vaccination) { s->vaccination = true; ev_vaccination = true; }
    switch (s->mask) {
    case ORC_MASK_NONE: if (P->mask_pt_threshold < x) s->mask = ORC_MASK_PT; break;
    case ORC_MASK_PT:
        if (x < P->mask_pt_threshold) s->mask = ORC_MASK_NONE;
        else if (P->mask_everywhere_threshold < x) s->mask = ORC_MASK_EVERYWHERE;
        break;
    default: if (x < P->mask_everywhere_threshold) s->mask = ORC_MASK_PT; break;
    }
    if (ev_vaccination) {                                                 // :481-513
        s->have_eligible = true;
        for (Area &a : s->areas) {
            std::lock_guard<std::mutex> g(a.m);
            for (Citizen &z : a.citizens) if (z.status == ORC_S) s->eligible.insert(z.global_index);
        }
    }
    uint32_t vaccinated_now = 0;
    if (s->have_eligible) {                                               // :524-553
        auto vaccinate = [&](uint32_t cid) {
            uint32_t la, li;
            { LookupEntry &e = s->lookup[cid]; std::lock_guard<std::mutex> lg(e.m); la = e.area; li = e.index; }
            Area &a = s->areas[la];
            std::lock_guard<std::mutex> g(a.m);
            a.citizens[li].status = ORC_V; a.citizens[li].timer = 0;      // unconditional, :551 (Q10)
            vaccinated_now++;
        };
        if (s->eligible.size() <= P->vaccination_rate) { for (uint32_t cid : s->eligible) vaccinate(cid); }
        else {
            uint32_t i = 0;
            while (vaccinated_now < P->vaccination_rate) {
                uint32_t w[4]; philox(P->seed, i++, s->time_step, 4, w);
                const uint64_t x64 = ((uint64_t)w[0] << 32) | w[1];
                const uint32_t j = (uint32_t)(((unsigned __int128)x64 * s->n_cit) >> 64);
                if (s->chosen_stamp[j] == s->time_step || !s->eligible.count(j)) continue;
                s->chosen_stamp[j] = s->time_step;
                vaccinate(j);
            }
        }
    }
    rec.lockdown = s->lockdown; rec.vaccination_active = s->vaccination; rec.mask_status = (uint32_t)s->mask;
    rec.vaccinated_now = vaccinated_now; rec.eligible_count = (uint32_t)s->eligible.size();
    rec.disease_exists = rec.exposed != 0 || rec.infected != 0 || rec.susceptible != 0;
    if (out) *out = rec;
    return 0;
}

extern "C" int rsh_run(rsh_sim *s, uint32_t n, orc_record *out)
{
    for (uint32_t k = 0; k < n; ++k) if (rsh_step(s, &out[k])) return -1;
    return (int)n;
}
