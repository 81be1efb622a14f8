// popgen.cpp -- seeded synthetic populations and Output-Area sharding (host only).
//
// Stands in for the reference's load_census_data + osm_data + SimulatorBuilder::build
// (sim/src/simulator_builder.rs:1162-1292), whose census tables and OSM extract are not in
// the repository.  Shapes follow SURVEY.md 8(d) / Appendix B: per-area population
// mean*(1 +- jitter); household size in 2..5 (output_area.rs:139); age uniform 0..90 with
// < MAX_STUDENT_AGE = 18 -> Student (config.rs:38, output_area.rs:155); adults get one of 9
// occupations, "Teaching" with weight p_teaching (Q12); workplaces inside the home area with
// capacity max(max(floor,2000)/density, 20) filled first-fit per occupation
// (building.rs:40,239-250, simulator_builder.rs:1042-1108, Q11); one School per
// citizens_per_school with a contiguous catchment, classes of <= ceil(n/26.6) per age group
// and offices of 12 (building.rs:307-308,346-443); STARTING_INFECTED_COUNT seeds drawn as
// uniform area then uniform citizen (simulator_builder.rs:1111-1140).
#include "../../include/esim.h"
#include "philox.h"
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <unordered_map>
#include <vector>

namespace {

struct Rng {
    uint64_t seed;
    philox_out block(uint32_t a, uint32_t b, uint32_t domain) const {
        return philox4x32_10(a, b, domain, 0x504F5047u /* "POPG" */, (uint32_t)seed, (uint32_t)(seed >> 32));
    }
    static double u(uint32_t w) { return (double)w * (1.0 / 4294967296.0); }
};

enum { DOM_AREA = 1, DOM_CITIZEN = 2, DOM_WORKPLACE = 3, DOM_SEED = 4 };

const uint32_t kDensity[8] = { 10, 12, 12, 12, 36, 19, 19, 47 };  // m^2 per worker, employment_densities.rs:31-44

template <class T> T *dup(const std::vector<T> &v)
{
    T *p = (T *)std::malloc(sizeof(T) * (v.size() ? v.size() : 1));
    if (p && !v.empty()) std::memcpy(p, v.data(), sizeof(T) * v.size());
    return p;
}

}  // namespace

extern "C" int esim_synth_preset(const char *name, esim_synth_spec *out)
{
    if (!name || !out) return ESIM_EINVAL;
    esim_synth_spec s;
    std::memset(&s, 0, sizeof s);
    s.n_seeds = 10;                     // STARTING_INFECTED_COUNT, config.rs:27
    s.seed = 0x5EED2011ull;
    s.area_jitter = 0.3;
    s.p_public_transport = 0.2;         // PUBLIC_TRANSPORT_PERCENTAGE, config.rs:36
    s.p_mask_compliant = 0.8;           // mask_percentage, disease.rs:126
    s.p_work_from_home = 0.14;
    s.p_teaching = 0.12;
    if (!std::strcmp(name, "york"))            { s.n_citizens = 197603;   s.n_areas = 637;    s.citizens_per_school = 7900; }
    else if (!std::strcmp(name, "yh_census"))  { s.n_citizens = 5249772;  s.n_areas = 17246;  s.citizens_per_school = 20600; }
    else if (!std::strcmp(name, "syn3m5"))     { s.n_citizens = 3457142;  s.n_areas = 15669;  s.citizens_per_school = 20600; }
    else if (!std::strcmp(name, "uk64m"))      { s.n_citizens = 64000000; s.n_areas = 290000; s.citizens_per_school = 20600; }
    else return ESIM_EINVAL;
    *out = s;
    return ESIM_OK;
}

// Generates the citizens of the school catchments [S*shard/n_shards, S*(shard+1)/n_shards) of the world
// described by `spec` (all of it for n_shards == 1).  Every random draw is keyed by GLOBAL area / citizen
// indices, so the shards of a world are exactly the pieces of the whole.  Catchments are closed under
// home, work and school membership, hence such a shard shares no building with any other shard.
static int synth_generate(const esim_synth_spec *spec, uint32_t shard, uint32_t n_shards, esim_population *out)
{
    if (!spec || !out || spec->n_areas == 0 || spec->n_citizens == 0 || n_shards == 0 || shard >= n_shards) return ESIM_EINVAL;
    const uint32_t N = spec->n_citizens, A = spec->n_areas;
    Rng rng{ spec->seed };

    // ---- per-area population, exactly N in total
    std::vector<double> w(A);
    double wsum = 0;
    for (uint32_t a = 0; a < A; ++a) {
        w[a] = 1.0 + spec->area_jitter * (2.0 * Rng::u(rng.block(a, 0, DOM_AREA).w0) - 1.0);
        wsum += w[a];
    }
    std::vector<uint32_t> area_off(A + 1, 0);
    {
        uint64_t given = 0;
        std::vector<uint32_t> pop(A);
        for (uint32_t a = 0; a < A; ++a) { pop[a] = (uint32_t)std::floor((double)N * w[a] / wsum); given += pop[a]; }
        for (uint32_t a = 0; given < N; a = (a + 1) % A) { pop[a]++; given++; }
        for (uint32_t a = 0; a < A; ++a) area_off[a + 1] = area_off[a] + pop[a];
    }

    // ---- schools: contiguous catchments of areas; the shard is a run of whole catchments
    uint32_t n_schools = (uint32_t)std::max<int64_t>(1, std::llround((double)N / std::max(1u, spec->citizens_per_school)));
    n_schools = std::min(n_schools, A);
    if (n_shards > n_schools) return ESIM_EINVAL;
    auto school_area_begin = [&](uint32_t s) { return (uint32_t)(((uint64_t)A * s) / n_schools); };
    const uint32_t s_lo = (uint32_t)(((uint64_t)n_schools * shard) / n_shards);
    const uint32_t s_hi = (uint32_t)(((uint64_t)n_schools * (shard + 1)) / n_shards);
    const uint32_t a_lo = school_area_begin(s_lo), a_hi = school_area_begin(s_hi);
    const uint32_t c_lo = area_off[a_lo], c_hi = area_off[a_hi], n = c_hi - c_lo;

    // ---- citizen attributes (local index = global index - c_lo)
    std::vector<uint32_t> home(n), work(n), room(n, ESIM_NO_ROOM);
    std::vector<uint8_t> flags(n), occ(n);
    std::vector<uint16_t> age(n);
    enum { OCC_STUDENT = 9, OCC_TEACHING = 8 };   // OccupationType::get_index, citizen.rs:312-324
    std::vector<uint8_t> wfh(n);
    for (uint32_t c = 0; c < n; ++c) {
        philox_out o = rng.block(c_lo + c, 0, DOM_CITIZEN);
        age[c] = (uint16_t)(((uint64_t)o.w0 * 91) >> 32);
        uint8_t f = 0;
        if (Rng::u(o.w1) < spec->p_public_transport) f |= ESIM_FLAG_USES_PUBLIC_TRANSPORT;
        philox_out o2 = rng.block(c_lo + c, 1, DOM_CITIZEN);
        if (Rng::u(o2.w0) < spec->p_mask_compliant) f |= ESIM_FLAG_MASK_COMPLIANT;
        flags[c] = f;
        if (age[c] < 18) { occ[c] = OCC_STUDENT; wfh[c] = 0; }
        else {
            double uo = Rng::u(o.w2);
            if (uo < spec->p_teaching) occ[c] = OCC_TEACHING;
            else occ[c] = (uint8_t)std::min(7.0, std::floor((uo - spec->p_teaching) / (1.0 - spec->p_teaching) * 8.0));
            wfh[c] = Rng::u(o.w3) < spec->p_work_from_home;
        }
    }

    std::vector<uint32_t> school_host_area(n_schools), school_building(n_schools);
    std::vector<int32_t> area_hosts_school(A, -1);
    for (uint32_t s = s_lo; s < s_hi; ++s) {
        uint32_t b = school_area_begin(s), e = school_area_begin(s + 1);
        school_host_area[s] = b + (e - b) / 2;
        area_hosts_school[school_host_area[s]] = (int32_t)s;
    }

    // ---- buildings area by area: households, workplaces, (school)
    std::vector<uint32_t> bld_area; std::vector<uint8_t> bld_type;
    bld_area.reserve(n / 2); bld_type.reserve(n / 2);
    for (uint32_t a = a_lo; a < a_hi; ++a) {
        const uint32_t c0 = area_off[a] - c_lo, c1 = area_off[a + 1] - c_lo;
        philox_out oa = rng.block(a, 1, DOM_AREA);
        const uint32_t hh = 2 + (uint32_t)(((uint64_t)oa.w0 * 4) >> 32);     // household size 2..5
        for (uint32_t c = c0; c < c1; ++c) {
            if ((c - c0) % hh == 0) { bld_area.push_back(a); bld_type.push_back(ESIM_HOUSEHOLD); }
            home[c] = (uint32_t)bld_area.size() - 1;
            work[c] = home[c];
        }
        // first-fit workplaces per occupation (simulator_builder.rs:1042-1108)
        for (uint32_t o = 0; o < 8; ++o) {
            uint32_t cap = 0, used = 0, wp = 0, nth = 0;
            for (uint32_t c = c0; c < c1; ++c) {
                if (occ[c] != o || wfh[c]) continue;
                if (used >= cap) {
                    philox_out ow = rng.block(a, o * 4096u + nth++, DOM_WORKPLACE);
                    uint32_t floor_space = 500 + (uint32_t)(((uint64_t)ow.w0 * 5500) >> 32);
                    cap = std::max(std::max(floor_space, 2000u) / kDensity[o], 20u);
                    used = 0;
                    bld_area.push_back(a); bld_type.push_back(ESIM_WORKPLACE);
                    wp = (uint32_t)bld_area.size() - 1;
                }
                work[c] = wp; used++;
            }
        }
        if (area_hosts_school[a] >= 0) {
            bld_area.push_back(a); bld_type.push_back(ESIM_SCHOOL);
            school_building[area_hosts_school[a]] = (uint32_t)bld_area.size() - 1;
        }
    }

    // ---- school rooms (School::with_students_and_teachers, building.rs:346-443)
    std::vector<uint32_t> room_bld;
    for (uint32_t s = s_lo; s < s_hi; ++s) {
        const uint32_t c0 = area_off[school_area_begin(s)] - c_lo, c1 = area_off[school_area_begin(s + 1)] - c_lo;
        std::vector<uint32_t> by_age[18], teachers;
        for (uint32_t c = c0; c < c1; ++c) {
            if (occ[c] == OCC_STUDENT) by_age[age[c]].push_back(c);
            else if (occ[c] == OCC_TEACHING && !wfh[c]) teachers.push_back(c);
        }
        size_t next_teacher = 0;
        for (int g = 0; g < 18; ++g) {
            if (by_age[g].empty()) continue;
            size_t classes = std::max<size_t>(1, (size_t)std::ceil((double)by_age[g].size() / 26.6));
            size_t class_size = (size_t)std::ceil((double)by_age[g].size() / (double)classes);
            for (size_t k = 0; k < by_age[g].size(); k += class_size) {
                uint32_t r = (uint32_t)room_bld.size();
                room_bld.push_back(school_building[s]);
                for (size_t i = k; i < std::min(by_age[g].size(), k + class_size); ++i) {
                    work[by_age[g][i]] = school_building[s]; room[by_age[g][i]] = r;
                }
                if (next_teacher < teachers.size()) {       // a class without a teacher is legal here
                    work[teachers[next_teacher]] = school_building[s]; room[teachers[next_teacher]] = r;
                    next_teacher++;
                }
            }
        }
        for (size_t k = next_teacher; k < teachers.size(); k += 12) {       // AVERAGE_OFFICE_SIZE
            uint32_t r = (uint32_t)room_bld.size();
            room_bld.push_back(school_building[s]);
            for (size_t i = k; i < std::min(teachers.size(), k + 12); ++i) {
                work[teachers[i]] = school_building[s]; room[teachers[i]] = r;
            }
        }
    }

    // ---- seeds
    std::vector<uint32_t> seeds;
    for (uint32_t i = 0; i < spec->n_seeds; ++i) {
        for (uint32_t attempt = 0; attempt < 64; ++attempt) {
            philox_out o = rng.block(i, attempt, DOM_SEED);
            uint32_t a = (uint32_t)(((uint64_t)o.w0 * A) >> 32);
            uint32_t na = area_off[a + 1] - area_off[a];
            if (!na) continue;                                  // empty area: the reference logs and skips
            const uint32_t g = area_off[a] + (uint32_t)(((uint64_t)o.w1 * na) >> 32);
            if (g >= c_lo && g < c_hi) seeds.push_back(g - c_lo);
            break;
        }
    }

    std::memset(out, 0, sizeof *out);
    out->n_citizens = n; out->n_buildings = (uint32_t)bld_area.size(); out->n_areas = A;
    out->n_rooms = (uint32_t)room_bld.size(); out->n_seeds = (uint32_t)seeds.size();
    out->citizen_id_base = c_lo; out->n_citizens_global = N;
    out->home_building = dup(home); out->work_building = dup(work); out->room = dup(room);
    out->flags = dup(flags); out->age = dup(age); out->occupation = dup(occ);
    out->building_area = dup(bld_area); out->building_type = dup(bld_type);
    out->room_building = dup(room_bld); out->seeds = dup(seeds);
    out->shared_building_local = nullptr; out->shared_room_local = nullptr;
    if (!out->home_building || !out->work_building || !out->room || !out->flags || !out->age ||
        !out->occupation || !out->building_area || !out->building_type || !out->room_building || !out->seeds) {
        esim_synth_free(out);
        return ESIM_ENOMEM;
    }
    return ESIM_OK;
}

extern "C" int esim_synth_create(const esim_synth_spec *spec, esim_population *out)
{
    return synth_generate(spec, 0, 1, out);
}

extern "C" int esim_synth_create_shard(const esim_synth_spec *spec, uint32_t shard, uint32_t n_shards, esim_population *out)
{
    return synth_generate(spec, shard, n_shards, out);
}

extern "C" void esim_synth_free(esim_population *p)
{
    if (!p) return;
    std::free((void *)p->home_building); std::free((void *)p->work_building); std::free((void *)p->room);
    std::free((void *)p->flags); std::free((void *)p->age); std::free((void *)p->occupation);
    std::free((void *)p->building_area); std::free((void *)p->building_type);
    std::free((void *)p->room_building); std::free((void *)p->seeds);
    std::free((void *)p->shared_building_local); std::free((void *)p->shared_room_local);
    std::memset(p, 0, sizeof *p);
}

// Output-Area sharding.  Citizens are owned by the shard of their HOME area.
// A building / room is "shared" when its members (residents, workers, room participants) live
// on more than one shard; every shard lists the shared ones in the same global order so that a
// plain SUM all-reduce over the exchange buffer combines the per-shard infected counts.
extern "C" int esim_shard_population(const esim_population *whole, const uint32_t *cuts, uint32_t n_shards,
                                     uint32_t shard, esim_population *out)
{
    if (!whole || !cuts || !out || shard >= n_shards || whole->citizen_id_base != 0) return ESIM_EINVAL;
    const uint32_t N = whole->n_citizens, B = whole->n_buildings, R = whole->n_rooms;
    if (cuts[0] != 0 || cuts[n_shards] != whole->n_areas) return ESIM_EINVAL;
    auto shard_of_area = [&](uint32_t a) {
        return (uint32_t)(std::upper_bound(cuts, cuts + n_shards + 1, a) - cuts) - 1;
    };
    // citizens must be ordered by home area for a shard to be a contiguous id range
    uint32_t c_begin = N, c_end = 0;
    uint32_t prev_area = 0;
    for (uint32_t c = 0; c < N; ++c) {
        uint32_t a = whole->building_area[whole->home_building[c]];
        if (a < prev_area) return ESIM_EINVAL;
        prev_area = a;
        if (a >= cuts[shard] && a < cuts[shard + 1]) { c_begin = std::min(c_begin, c); c_end = c + 1; }
    }
    if (c_begin > c_end) { c_begin = c_end = 0; }

    // which shards touch each building / room (bitmask for <= 64 shards)
    if (n_shards > 64) return ESIM_EINVAL;
    std::vector<uint64_t> bmask(B, 0), rmask(R, 0);
    for (uint32_t c = 0; c < N; ++c) {
        uint64_t bit = 1ull << shard_of_area(whole->building_area[whole->home_building[c]]);
        bmask[whole->home_building[c]] |= bit;
        bmask[whole->work_building[c]] |= bit;
        if (whole->room[c] != ESIM_NO_ROOM && whole->room[c] < R) rmask[whole->room[c]] |= bit;
    }
    auto multi = [](uint64_t m) { return (m & (m - 1)) != 0; };

    std::vector<int32_t> bmap(B, -1), rmap(R, -1);
    std::vector<uint32_t> bld_area; std::vector<uint8_t> bld_type; std::vector<uint32_t> room_bld_old;
    auto local_building = [&](uint32_t b) {
        if (bmap[b] < 0) { bmap[b] = (int32_t)bld_area.size(); bld_area.push_back(whole->building_area[b]); bld_type.push_back(whole->building_type[b]); }
        return (uint32_t)bmap[b];
    };
    const uint32_t n = c_end - c_begin;
    std::vector<uint32_t> home(n), work(n), room(n, ESIM_NO_ROOM);
    std::vector<uint8_t> flags(n), occ(n); std::vector<uint16_t> age(n);
    for (uint32_t i = 0; i < n; ++i) {
        uint32_t c = c_begin + i;
        home[i] = local_building(whole->home_building[c]);
        work[i] = local_building(whole->work_building[c]);
        if (whole->room[c] != ESIM_NO_ROOM) {
            uint32_t r = whole->room[c];
            if (rmap[r] < 0) { rmap[r] = (int32_t)room_bld_old.size(); room_bld_old.push_back(whole->room_building[r]); }
            room[i] = (uint32_t)rmap[r];
        }
        flags[i] = whole->flags[c];
        age[i] = whole->age ? whole->age[c] : 0;
        occ[i] = whole->occupation ? whole->occupation[c] : 0;
    }
    std::vector<uint32_t> room_bld(room_bld_old.size());
    for (size_t r = 0; r < room_bld_old.size(); ++r) room_bld[r] = local_building(room_bld_old[r]);

    std::vector<int32_t> shared_b, shared_r;
    for (uint32_t b = 0; b < B; ++b) if (multi(bmask[b])) shared_b.push_back(bmap[b]);
    for (uint32_t r = 0; r < R; ++r) if (multi(rmask[r])) shared_r.push_back(rmap[r]);

    std::vector<uint32_t> seeds;
    for (uint32_t i = 0; i < whole->n_seeds; ++i)
        if (whole->seeds[i] >= c_begin && whole->seeds[i] < c_end) seeds.push_back(whole->seeds[i] - c_begin);

    std::memset(out, 0, sizeof *out);
    out->n_citizens = n; out->n_buildings = (uint32_t)bld_area.size(); out->n_areas = whole->n_areas;
    out->n_rooms = (uint32_t)room_bld.size(); out->n_seeds = (uint32_t)seeds.size();
    out->citizen_id_base = c_begin; out->n_citizens_global = N;
    out->n_shared_buildings = (uint32_t)shared_b.size(); out->n_shared_rooms = (uint32_t)shared_r.size();
    out->home_building = dup(home); out->work_building = dup(work); out->room = dup(room);
    out->flags = dup(flags); out->age = dup(age); out->occupation = dup(occ);
    out->building_area = dup(bld_area); out->building_type = dup(bld_type);
    out->room_building = dup(room_bld); out->seeds = dup(seeds);
    out->shared_building_local = dup(shared_b); out->shared_room_local = dup(shared_r);
    return ESIM_OK;
}
