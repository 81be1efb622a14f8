// popgen.cpp -- seeded synthetic populations and Output-Area sharding (host only).
//
// Stands in for the reference's load_census_data + osm_data inputs, which are not in the repository, and then
// follows SimulatorBuilder::build (sim/src/simulator_builder.rs:1162-1292) on those synthetic inputs, step by step:
//
//   inputs (synthetic)                                   what the reference reads there
//   per-area census population, mean * (1 +- jitter)     PopulationRecord, output_area.rs:137
//   per-area count of OSM buildings tagged as dwellings  possible_buildings (TagClassifiedBuilding::Household), :139
//   per-area OSM "workplace" buildings with floor areas  possible_buildings_per_area, simulator_builder.rs:717
//   raw school buildings at points of the map            osm_data.voronoi() / building_locations, :351-365
//   age uniform 0..90, occupation by census major group  get_random_age / get_random_occupation, output_area.rs:152-153
//
//   build step                                           reference
//   households of pop / n_dwellings + 1 residents         output_area.rs:139-187
//   < MAX_STUDENT_AGE = 18 -> Student                     output_area.rs:155, config.rs:38
//   students -> the closest raw school                    simulator_builder.rs:432-486
//   "Teaching" (Q12) -> the closest school among the      simulator_builder.rs:489-545
//     nearest MAX_ITEMS_RETURNED = 200 that still lacks
//     class teachers, else secondary staff at the closest
//   classes of <= ceil(n / 26.6) per age group, offices   building.rs:346-443
//     of 12 from the teachers left over
//   per-area workplaces: floor-space bins per occupation, simulator_builder.rs:865-1108, building.rs:236-250,
//     first fit, capacity max(max(floor,2000)/density,20)   models/mod.rs:63-75; inside the HOME area (Q11)
//   STARTING_INFECTED_COUNT seeds: uniform area, then     simulator_builder.rs:1111-1140
//     uniform citizen of it (lost when the area is empty)
//
// Output Areas are the cells of a near-square grid in row-major order (so a contiguous range of areas is a band of the
// map); a household stands at a jittered point of its cell.  What the reference's logs say about the shape of these
// inputs is in SURVEY.md Appendix B and DESIGN.md 2 (the one input no log pins -- how many dwellings OSM tags per
// Output Area, which sets the household size -- is calibrated against the reference's recorded York trajectory).
// Only IEEE add/mul/div/sqrt-free arithmetic is used (no libm), so the same spec gives the same population everywhere.
#include "../../include/esim.h"
#include "philox.h"
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <unordered_map>
#include <vector>

namespace {

struct Rng {
    uint64_t seed;
    philox_out block(uint32_t a, uint32_t b, uint32_t domain) const {
        return philox4x32_10(a, b, domain, 0x504F5047u /* "POPG" */, (uint32_t)seed, (uint32_t)(seed >> 32));
    }
    static double u(uint32_t w) { return (double)w * (1.0 / 4294967296.0); }
    // ~N(0,1): sum of 12 uniforms - 6 (exact in double), from three blocks
    double normal(uint32_t a, uint32_t b, uint32_t domain) const {
        double s = 0;
        for (uint32_t k = 0; k < 3; ++k) {
            philox_out o = block(a, b * 4u + k, domain);
            s += u(o.w0) + u(o.w1) + u(o.w2) + u(o.w3);
        }
        return s - 6.0;
    }
};

enum { DOM_AREA = 1, DOM_CITIZEN = 2, DOM_WORKPLACE = 3, DOM_SEED = 4, DOM_SCHOOL = 5, DOM_HOUSEHOLD = 6,
       DOM_AREA_DWELLINGS = 7, DOM_AREA_WORKPLACES = 8 };

// m^2 per worker by OccupationType::get_index (citizen.rs:312-324): get_density_for_occupation, models/mod.rs:63-75,
// with the EmploymentDensities of load_census_data/src/tables/employment_densities.rs:31-44
const uint32_t kDensity[9] = { 12, 12, 10, 12, 36, 47, 19, 36, 19 };
// census major occupation groups 1-8 (share of the non-"Elementary" workforce, 2011 census England); group 9 is p_teaching (Q12)
const double kOccupationWeight[8] = { 10.9, 17.5, 12.8, 11.4, 11.4, 9.3, 8.4, 7.2 };
enum { OCC_TEACHING = 8, OCC_STUDENT = 9 };
const double kAverageClassSize = 26.6;      // AVERAGE_CLASS_SIZE, building.rs:307
const uint32_t kOfficeSize = 12;            // AVERAGE_OFFICE_SIZE, building.rs:308
const uint32_t kMinFloorSpace = 2000;       // MINIMUM_FLOOR_SPACE_SIZE, building.rs:40
const uint32_t kMinOccupants = 20;          // MIN_WORKPLACE_OCCUPANT_COUNT, config.rs:31

// 2^x by range reduction and a degree-7 polynomial in plain IEEE arithmetic (no libm: see the header comment)
double det_exp2(double x)
{
    if (x > 60) x = 60;
    if (x < -60) x = -60;
    const double fl = (double)(long long)(x < 0 ? x - 1.0 : x);     // floor for non-integers; integers below are fixed up
    double ip = fl, f = x - fl;
    if (f >= 1.0) { ip += 1.0; f -= 1.0; }
    const double t = f * 0.6931471805599453;                          // e^t, t in [0, ln 2)
    double p = 1.0 + t * (1.0 + t * (0.5 + t * (1.0 / 6 + t * (1.0 / 24 + t * (1.0 / 120 + t * (1.0 / 720 + t * (1.0 / 5040)))))));
    long long e = (long long)ip;
    while (e > 0) { p *= 2.0; --e; }
    while (e < 0) { p *= 0.5; ++e; }
    return p;
}

// count ~ round(median * exp(sigma * z)), z ~ N(0,1)
uint32_t lognormal_count(double median, double sigma, double z, uint32_t cap)
{
    const double v = median * det_exp2(sigma * z * 1.4426950408889634);
    if (!(v < (double)cap)) return cap;
    return (uint32_t)(v + 0.5);
}

template <class T> T *dup(const std::vector<T> &v)
{
    T *p = (T *)std::malloc(sizeof(T) * (v.size() ? v.size() : 1));
    if (p && !v.empty()) std::memcpy(p, v.data(), sizeof(T) * v.size());
    return p;
}

// The raw school buildings on the map and "the closest school" queries (osm_data.voronoi().find_seeds_for_point,
// simulator_builder.rs:407-408; MAX_ITEMS_RETURNED candidates, osm_data/src/quadtree.rs:544).
struct SchoolMap {
    std::vector<double> x, y;
    int G = 1; double cell = 1, width = 1, height = 1;
    std::vector<uint32_t> cell_off, cell_items;

    void build(double w, double h)
    {
        width = w; height = h;
        const size_t S = x.size();
        G = std::max(1, (int)std::ceil(std::sqrt((double)S / 1.5)));
        cell = std::max(w, h) / G;
        cell_off.assign((size_t)G * G + 1, 0);
        auto cell_of = [&](size_t s) { return (size_t)std::min(G - 1, (int)(y[s] / cell)) * G + std::min(G - 1, (int)(x[s] / cell)); };
        for (size_t s = 0; s < S; ++s) cell_off[cell_of(s) + 1]++;
        for (size_t c = 0; c < (size_t)G * G; ++c) cell_off[c + 1] += cell_off[c];
        cell_items.resize(S);
        std::vector<uint32_t> cur(cell_off.begin(), cell_off.end() - 1);
        for (size_t s = 0; s < S; ++s) cell_items[cur[cell_of(s)]++] = (uint32_t)s;
    }
    // the k closest schools with use[s] != 0, closest first (ties by index)
    void nearest(double px, double py, size_t k, const uint8_t *use, std::vector<std::pair<double, uint32_t>> &out) const
    {
        out.clear();
        const int cx = std::min(G - 1, std::max(0, (int)(px / cell))), cy = std::min(G - 1, std::max(0, (int)(py / cell)));
        for (int r = 0; r <= G; ++r) {
            for (int yy = std::max(0, cy - r); yy <= std::min(G - 1, cy + r); ++yy)
                for (int xx = std::max(0, cx - r); xx <= std::min(G - 1, cx + r); ++xx) {
                    if (std::max(std::abs(xx - cx), std::abs(yy - cy)) != r) continue;     // the ring only
                    const size_t c = (size_t)yy * G + xx;
                    for (uint32_t q = cell_off[c]; q < cell_off[c + 1]; ++q) {
                        const uint32_t s = cell_items[q];
                        if (use && !use[s]) continue;
                        const double dx = x[s] - px, dy = y[s] - py;
                        out.emplace_back(dx * dx + dy * dy, s);
                    }
                }
            if (out.size() >= k) {
                // everything outside the searched block is at least r cells away
                std::nth_element(out.begin(), out.begin() + (k - 1), out.end());
                const double bound = (double)r * cell;
                if (out[k - 1].first <= bound * bound) break;
            }
        }
        std::sort(out.begin(), out.end());
        if (out.size() > k) out.resize(k);
    }
};

struct AreaWork {           // result of the workplace assignment of one area
    std::vector<int32_t> work_local;     // per citizen of the area: index into the area's workplaces, or -1
    uint32_t n_workplaces = 0;
};

// SimulatorBuilder::assign_buildings_per_output_area + assign_workplaces_to_citizens_per_occupation
// (simulator_builder.rs:865-1108) for one area.  occ: per citizen of the area.
void assign_area_workplaces(const Rng &rng, const esim_synth_spec &sp, uint32_t a, uint32_t g0, const uint8_t *occ, const uint8_t *no_work,
                            uint32_t n, uint32_t n_buildings, AreaWork &out)
{
    out.work_local.assign(n, -1);
    out.n_workplaces = 0;
    if (n_buildings == 0 || n == 0) return;                            // "No Workplace buildings exist", :827-835
    std::vector<int64_t> size(n_buildings);
    int64_t available = 0;
    for (uint32_t j = 0; j < n_buildings; ++j) {                       // floor area of a raw building, osm_data/src/lib.rs:260
        const double z = rng.normal(a, j, DOM_WORKPLACE);
        size[j] = lognormal_count(sp.workplace_floor_median, sp.workplace_floor_sigma, z, 1u << 20);
        available += size[j];
    }
    // citizen_ids.shuffle(), :898; grouped by detailed occupation (students have none), :901-908
    std::vector<std::pair<uint32_t, uint32_t>> order;
    order.reserve(n);
    for (uint32_t i = 0; i < n; ++i)
        if (occ[i] <= OCC_TEACHING && !no_work[i]) order.emplace_back(rng.block(g0 + i, 2, DOM_CITIZEN).w0, i);
    if (order.empty()) return;                                          // "no workers exist", :872-878
    std::sort(order.begin(), order.end());
    std::vector<uint32_t> by_occ[9];
    for (auto &kv : order) by_occ[occ[kv.second]].push_back(kv.second);
    int64_t required[9], required_total = 0;
    for (int o = 0; o < 9; ++o) { required[o] = (int64_t)kDensity[o] * (int64_t)by_occ[o].size(); required_total += required[o]; }
    if (available == 0) return;
    // scale = ceil(required / available * BUILDING_PER_OCCUPATION_OVERCAPACITY), :939-941
    const int64_t scale = (int64_t)std::ceil(((double)required_total / (double)available) * 1.1);
    int64_t current[9] = { 0 }, diff[9];
    std::vector<uint32_t> bins[9];
    for (int o = 0; o < 9; ++o) diff[o] = required[o];
    for (uint32_t j = 0; j < n_buildings; ++j) {                        // :958-997 (buildings arrive shuffled: they are random already)
        const int64_t bs = size[j] * scale;
        if (bs == 0) continue;
        bool added = false;
        for (int o = 0; o < 9; ++o)                                     // no `break` in the reference: every bin it fits, :966-974
            if (current[o] + bs < required[o]) { current[o] += bs; bins[o].push_back(j); diff[o] -= bs; added = true; }
        if (!added) {
            int best = -1; int64_t best_diff = INT64_MAX;
            for (int o = 0; o < 9; ++o) if (0 < diff[o] && diff[o] < best_diff) { best_diff = diff[o]; best = o; }
            if (best >= 0) { current[best] += bs; bins[best].push_back(j); diff[best] -= bs; }
        }
    }
    for (int o = 0; o < OCC_TEACHING; ++o) {                            // Teaching is handled in build_schools, :1014-1017
        if (bins[o].empty()) continue;                                  // Err -> `continue`, :1027-1030: these workers stay at home
        size_t next = 0;
        auto capacity = [&](uint32_t j) {                               // Workplace::max_occupant_count, building.rs:236-250
            const uint32_t floor_space = (uint32_t)std::max<int64_t>(size[j], kMinFloorSpace);
            return std::max(floor_space / kDensity[o], kMinOccupants);
        };
        uint32_t cap = capacity(bins[o][next++]), used = 0;
        uint32_t wp = out.n_workplaces++;
        for (uint32_t i : by_occ[o]) {
            if (used >= cap) {
                if (next >= bins[o].size()) break;                      // "Ran out of Workplaces", :1083-1086
                cap = capacity(bins[o][next++]); used = 0; wp = out.n_workplaces++;
            }
            out.work_local[i] = (int32_t)wp; used++;
        }
    }
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
    s.p_work_from_home = 0.0;           // extra, not in the reference: there "working from home" is what is left without a workplace
    s.p_teaching = 0.123;               // 19 948 of 161 852 adults, logs/pc_logs/v1.6/york.log:439-440
    // OSM "workplace" buildings per Output Area and their floor areas: debug_dumps/pre_duplicate_removal/log.txt
    // (214 York areas: 10 % / median / 90 % = 2 / 12 / 64 buildings; floor median 98 m^2, 90 % 600 m^2);
    // 11 of 637 areas without any, logs/pc_logs/v1.6/york.log:453-470
    s.workplace_buildings_median = 12.0; s.workplace_buildings_sigma = 1.3; s.p_area_without_workplaces = 0.017;
    s.workplace_floor_median = 98.0; s.workplace_floor_sigma = 1.4;
    s.teacher_candidate_schools = 200;  // MAX_ITEMS_RETURNED, osm_data/src/quadtree.rs:544
    // dwellings OSM tags per Output Area (sets the household size, output_area.rs:139): 5 of 637 York areas have none
    // (york.log:436); the rest is calibrated, DESIGN.md 2
    s.household_buildings_median = 50.0; s.household_buildings_sigma = 1.0; s.p_area_without_households = 5.0 / 637.0;
    if (!std::strcmp(name, "york"))            { s.n_citizens = 197603;   s.n_areas = 637;    s.citizens_per_school = 7900; }
    else if (!std::strcmp(name, "yh_census"))  { s.n_citizens = 5249772;  s.n_areas = 17246;  s.citizens_per_school = 20600; }
    // the reference's Yorkshire-and-Humber run: 4397 of its 15 669 areas have no dwelling tagged and stay empty
    // (epidemic_sim_v1.6_17739074.log:2536-2537); the 64 M world keeps that share
    else if (!std::strcmp(name, "syn3m5"))     { s.n_citizens = 3457142;  s.n_areas = 15669;  s.citizens_per_school = 20600; s.p_area_without_households = 4397.0 / 15669.0; }
    else if (!std::strcmp(name, "uk64m"))      { s.n_citizens = 64000000; s.n_areas = 290000; s.citizens_per_school = 20600; s.p_area_without_households = 4397.0 / 15669.0; }
    else return ESIM_EINVAL;
    *out = s;
    return ESIM_OK;
}

static int synth_generate(const esim_synth_spec *spec, esim_population *out)
{
    if (!spec || !out || spec->n_areas == 0 || spec->n_citizens == 0) return ESIM_EINVAL;
    const esim_synth_spec &sp = *spec;
    const uint32_t N = sp.n_citizens, A = sp.n_areas;
    Rng rng{ sp.seed };
    const uint32_t W = (uint32_t)std::ceil(std::sqrt((double)A));            // the map: W columns, row-major areas
    const uint32_t H = (A + W - 1) / W;

    // ---- census: per-area population (exactly N in total) and OSM dwellings per area
    std::vector<double> w(A, 0.0);
    std::vector<uint32_t> dwellings(A, 0);
    double wsum = 0;
    uint32_t n_populated = 0;
    for (uint32_t a = 0; a < A; ++a) {
        philox_out o = rng.block(a, 0, DOM_AREA);
        const bool empty = Rng::u(o.w1) < sp.p_area_without_households;       // "no households exist", simulator_builder.rs:226-235
        if (empty) continue;
        dwellings[a] = std::max(1u, lognormal_count(sp.household_buildings_median, sp.household_buildings_sigma,
                                                     rng.normal(a, 0, DOM_AREA_DWELLINGS), 1u << 20));
        w[a] = 1.0 + sp.area_jitter * (2.0 * Rng::u(o.w0) - 1.0);
        wsum += w[a];
        n_populated++;
    }
    if (!n_populated) { dwellings[0] = std::max(1u, (uint32_t)(sp.household_buildings_median + 0.5)); w[0] = 1.0; wsum = 1.0; n_populated = 1; }
    std::vector<uint32_t> area_off(A + 1, 0);
    {
        uint64_t given = 0;
        std::vector<uint32_t> pop(A, 0);
        for (uint32_t a = 0; a < A; ++a) if (dwellings[a]) { pop[a] = (uint32_t)std::floor((double)N * w[a] / wsum); given += pop[a]; }
        for (uint32_t a = 0; given < N; a = (a + 1) % A) if (dwellings[a]) { pop[a]++; given++; }
        for (uint32_t a = 0; a < A; ++a) area_off[a + 1] = area_off[a] + pop[a];
    }

    // ---- households: pop / dwellings + 1 residents each, output_area.rs:139 (the last one of an area takes what is left)
    std::vector<uint32_t> hh_size(A, 1), hh_off(A + 1, 0);
    for (uint32_t a = 0; a < A; ++a) {
        const uint32_t pop = area_off[a + 1] - area_off[a];
        hh_size[a] = dwellings[a] ? pop / dwellings[a] + 1 : 1;
        hh_off[a + 1] = hh_off[a] + (pop + hh_size[a] - 1) / hh_size[a];
    }

    // ---- citizens: age, occupation, flags (Citizen::new, citizen.rs:139-162; output_area.rs:152-169)
    std::vector<uint8_t> flags(N), occ(N), no_work(N, 0);
    std::vector<uint16_t> age(N);
    double occ_cum[8]; { double s = 0; for (int o = 0; o < 8; ++o) { s += kOccupationWeight[o]; occ_cum[o] = s; } for (int o = 0; o < 8; ++o) occ_cum[o] /= s; }
    const unsigned n_threads = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    auto parallel_for = [&](uint32_t n_items, auto fn) {
        if (n_items < 4096 || n_threads == 1) { fn(0u, n_items); return; }
        std::vector<std::thread> th;
        for (unsigned t = 0; t < n_threads; ++t)
            th.emplace_back(fn, (uint32_t)((uint64_t)n_items * t / n_threads), (uint32_t)((uint64_t)n_items * (t + 1) / n_threads));
        for (auto &x : th) x.join();
    };
    parallel_for(N, [&](uint32_t lo, uint32_t hi) {
        for (uint32_t c = lo; c < hi; ++c) {
            philox_out o = rng.block(c, 0, DOM_CITIZEN);
            age[c] = (uint16_t)(((uint64_t)o.w0 * 91) >> 32);
            uint8_t f = 0;
            if (Rng::u(o.w1) < sp.p_public_transport) f |= ESIM_FLAG_USES_PUBLIC_TRANSPORT;
            philox_out o2 = rng.block(c, 1, DOM_CITIZEN);
            if (Rng::u(o2.w0) < sp.p_mask_compliant) f |= ESIM_FLAG_MASK_COMPLIANT;
            flags[c] = f;
            if (age[c] < 18) occ[c] = OCC_STUDENT;
            else {
                const double uo = Rng::u(o.w2);
                if (uo < sp.p_teaching) occ[c] = OCC_TEACHING;
                else {
                    const double v = (uo - sp.p_teaching) / (1.0 - sp.p_teaching);
                    uint8_t k = 0;
                    while (k < 7 && v >= occ_cum[k]) ++k;
                    occ[c] = k;
                }
                no_work[c] = Rng::u(o.w3) < sp.p_work_from_home;
            }
        }
    });

    // ---- raw school buildings on the map
    uint32_t S = (uint32_t)std::max<int64_t>(1, std::llround((double)N / std::max(1u, sp.citizens_per_school)));
    SchoolMap map;
    std::vector<uint32_t> school_area(S);
    map.x.resize(S); map.y.resize(S);
    for (uint32_t s = 0; s < S; ++s) {
        philox_out o = rng.block(s, 0, DOM_SCHOOL);
        const uint32_t a = (uint32_t)(((uint64_t)o.w0 * A) >> 32);
        school_area[s] = a;
        map.x[s] = (double)(a % W) + Rng::u(o.w1);
        map.y[s] = (double)(a / W) + Rng::u(o.w2);
    }
    map.build((double)W, (double)H);

    // ---- students: the closest school to the household (simulator_builder.rs:432-486)
    std::vector<uint32_t> school_of(N, UINT32_MAX);
    parallel_for(A, [&](uint32_t lo, uint32_t hi) {
        std::vector<std::pair<double, uint32_t>> near;
        for (uint32_t a = lo; a < hi; ++a) {
            const uint32_t c0 = area_off[a], c1 = area_off[a + 1], hs = hh_size[a];
            for (uint32_t h0 = c0, k = 0; h0 < c1; h0 += hs, ++k) {
                bool any = false;
                for (uint32_t c = h0; c < std::min(c1, h0 + hs); ++c) any |= occ[c] == OCC_STUDENT;
                if (!any) continue;
                philox_out o = rng.block(a, k, DOM_HOUSEHOLD);
                map.nearest((double)(a % W) + Rng::u(o.w0), (double)(a / W) + Rng::u(o.w1), 1, nullptr, near);
                for (uint32_t c = h0; c < std::min(c1, h0 + hs); ++c) if (occ[c] == OCC_STUDENT) school_of[c] = near[0].second;
            }
        }
    });
    // students per school and age group; class teachers each school needs (the reference counts an empty age group
    // below the oldest one present as one class, simulator_builder.rs:507-513)
    std::vector<uint32_t> per_age((size_t)S * 18, 0);
    for (uint32_t c = 0; c < N; ++c) if (school_of[c] != UINT32_MAX) per_age[(size_t)school_of[c] * 18 + age[c]]++;
    std::vector<uint32_t> deficit(S, 0);
    std::vector<uint8_t> has_students(S, 0);
    for (uint32_t s = 0; s < S; ++s) {
        int top = -1;
        for (int g = 0; g < 18; ++g) if (per_age[(size_t)s * 18 + g]) top = g;
        has_students[s] = top >= 0;
        for (int g = 0; g <= top; ++g)
            deficit[s] += (uint32_t)std::max(1.0, std::ceil((double)per_age[(size_t)s * 18 + g] / kAverageClassSize));
    }
    // ---- teachers, in citizen order (simulator_builder.rs:489-545)
    std::vector<std::vector<uint32_t>> teachers(S);
    {
        std::vector<std::pair<double, uint32_t>> near;
        const size_t K = std::max<size_t>(1, sp.teacher_candidate_schools);
        for (uint32_t a = 0; a < A; ++a) {
            const uint32_t c0 = area_off[a], c1 = area_off[a + 1];
            bool listed = false; size_t p = 0;
            for (uint32_t c = c0; c < c1; ++c) {
                if (occ[c] != OCC_TEACHING || no_work[c]) continue;
                if (!listed) { map.nearest((double)(a % W) + 0.5, (double)(a / W) + 0.5, K, has_students.data(), near); listed = true; }
                if (near.empty()) break;                                 // failed_teacher_count, :541-543
                while (p < near.size() && deficit[near[p].second] == 0) ++p;
                const uint32_t s = p < near.size() ? near[p].second : near[0].second;   // else secondary staff at the closest, :529-540
                if (p < near.size()) deficit[s]--;
                teachers[s].push_back(c);
                school_of[c] = s;
            }
        }
    }

    // ---- workplaces, area by area (build_workplaces, simulator_builder.rs:717-863; Q11: inside the home area)
    std::vector<AreaWork> area_work(A);
    std::vector<uint32_t> wp_buildings(A, 0);
    for (uint32_t a = 0; a < A; ++a) {
        philox_out o = rng.block(a, 1, DOM_AREA);
        if (Rng::u(o.w0) < sp.p_area_without_workplaces) continue;
        wp_buildings[a] = lognormal_count(sp.workplace_buildings_median, sp.workplace_buildings_sigma, rng.normal(a, 0, DOM_AREA_WORKPLACES), 4096);
    }
    parallel_for(A, [&](uint32_t lo, uint32_t hi) {
        for (uint32_t a = lo; a < hi; ++a)
            assign_area_workplaces(rng, sp, a, area_off[a], occ.data() + area_off[a], no_work.data() + area_off[a],
                                   area_off[a + 1] - area_off[a], wp_buildings[a], area_work[a]);
    });

    // ---- buildings: per area its households, then its workplaces, then the schools standing in it
    std::unordered_map<uint32_t, std::vector<uint32_t>> schools_in_area;
    for (uint32_t s = 0; s < S; ++s)
        if (has_students[s] && !teachers[s].empty()) schools_in_area[school_area[s]].push_back(s);   // no teachers: not built, :611-616
    std::vector<uint32_t> bld_base(A + 1, 0);
    for (uint32_t a = 0; a < A; ++a) {
        uint32_t n_sch = 0;
        auto it = schools_in_area.find(a);
        if (it != schools_in_area.end()) n_sch = (uint32_t)it->second.size();
        bld_base[a + 1] = bld_base[a] + (hh_off[a + 1] - hh_off[a]) + area_work[a].n_workplaces + n_sch;
    }
    const uint32_t B = bld_base[A];
    std::vector<uint32_t> bld_area(B); std::vector<uint8_t> bld_type(B);
    std::vector<uint32_t> school_building(S, UINT32_MAX);
    std::vector<uint32_t> home(N), work(N), room(N, ESIM_NO_ROOM);
    parallel_for(A, [&](uint32_t lo, uint32_t hi) {
        for (uint32_t a = lo; a < hi; ++a) {
            const uint32_t nh = hh_off[a + 1] - hh_off[a], nw = area_work[a].n_workplaces;
            uint32_t b = bld_base[a];
            for (uint32_t k = 0; k < nh; ++k, ++b) { bld_area[b] = a; bld_type[b] = ESIM_HOUSEHOLD; }
            for (uint32_t k = 0; k < nw; ++k, ++b) { bld_area[b] = a; bld_type[b] = ESIM_WORKPLACE; }
            for (; b < bld_base[a + 1]; ++b) { bld_area[b] = a; bld_type[b] = ESIM_SCHOOL; }
            const uint32_t c0 = area_off[a], c1 = area_off[a + 1], hs = hh_size[a];
            for (uint32_t c = c0; c < c1; ++c) {
                home[c] = bld_base[a] + (c - c0) / hs;
                const int32_t wl = area_work[a].work_local[c - c0];
                work[c] = wl >= 0 ? bld_base[a] + nh + (uint32_t)wl : home[c];
            }
        }
    });
    for (auto &kv : schools_in_area) {
        const uint32_t a = kv.first;
        uint32_t b = bld_base[a] + (hh_off[a + 1] - hh_off[a]) + area_work[a].n_workplaces;
        for (uint32_t s : kv.second) school_building[s] = b++;
    }
    { std::vector<AreaWork>().swap(area_work); }

    // ---- school rooms (School::with_students_and_teachers, building.rs:346-443)
    std::vector<uint32_t> room_bld;
    {
        // students of each school by age, in citizen order
        std::vector<uint64_t> start((size_t)S * 18 + 1, 0);
        for (size_t i = 0; i < (size_t)S * 18; ++i) start[i + 1] = start[i] + per_age[i];
        std::vector<uint32_t> sorted(start.back());
        std::vector<uint64_t> cur(start.begin(), start.end() - 1);
        for (uint32_t c = 0; c < N; ++c)
            if (occ[c] == OCC_STUDENT && school_of[c] != UINT32_MAX) sorted[cur[(size_t)school_of[c] * 18 + age[c]]++] = c;
        for (uint32_t s = 0; s < S; ++s) {
            const uint32_t sb = school_building[s];
            if (sb == UINT32_MAX) continue;                              // its students and teachers stay at home
            size_t next_teacher = 0;
            for (int g = 0; g < 18; ++g) {
                const uint64_t b0 = start[(size_t)s * 18 + g], n = per_age[(size_t)s * 18 + g];
                if (!n) continue;                                        // "Remove any empty age groups", :359-364
                const uint64_t classes = std::max<uint64_t>(1, (uint64_t)std::ceil((double)n / kAverageClassSize));
                const uint64_t class_size = (uint64_t)std::ceil((double)n / (double)classes);
                for (uint64_t k = 0; k < n; k += class_size) {
                    const uint32_t r = (uint32_t)room_bld.size();
                    room_bld.push_back(sb);
                    for (uint64_t i = k; i < std::min(n, k + class_size); ++i) { work[sorted[b0 + i]] = sb; room[sorted[b0 + i]] = r; }
                    if (next_teacher < teachers[s].size()) {             // the reference panics without one (:377-383); tiny test worlds may lack them
                        const uint32_t t = teachers[s][next_teacher++];
                        work[t] = sb; room[t] = r;
                    }
                }
            }
            for (size_t k = next_teacher; k < teachers[s].size(); k += kOfficeSize) {    // offices, :420-431
                const uint32_t r = (uint32_t)room_bld.size();
                room_bld.push_back(sb);
                for (size_t i = k; i < std::min(teachers[s].size(), k + kOfficeSize); ++i) { work[teachers[s][i]] = sb; room[teachers[s][i]] = r; }
            }
        }
    }

    // ---- seeds (apply_initial_infections, simulator_builder.rs:1111-1140): an empty area loses its seed
    std::vector<uint32_t> seeds;
    for (uint32_t i = 0; i < sp.n_seeds; ++i) {
        philox_out o = rng.block(i, 0, DOM_SEED);
        const uint32_t a = (uint32_t)(((uint64_t)o.w0 * A) >> 32);
        const uint32_t na = area_off[a + 1] - area_off[a];
        if (!na) continue;
        seeds.push_back(area_off[a] + (uint32_t)(((uint64_t)o.w1 * na) >> 32));
    }

    std::memset(out, 0, sizeof *out);
    out->n_citizens = N; out->n_buildings = B; out->n_areas = A;
    out->n_rooms = (uint32_t)room_bld.size(); out->n_seeds = (uint32_t)seeds.size();
    out->citizen_id_base = 0; out->n_citizens_global = N;
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
    return synth_generate(spec, out);
}

// Area boundaries of n_shards bands of the map.  by_work == 0: about the same number of citizens each.  by_work != 0: about the
// same EXPECTED WORK each -- a citizen is drawn for by the shard it lives on, once per Infected occupant and step in every list it
// is a member of (its household, its work place, its class room: building.rs:202-204,278-280,494-522), so at a uniform prevalence
// a citizen's share of the draws is the sum of the sizes of those lists, plus one for its own marks; a band's weight is the sum
// over the citizens living in it.  (Where the epidemic actually sits is not known before the run: these are static weights.)
extern "C" int esim_shard_cuts(const esim_population *whole, uint32_t n_shards, int by_work, uint32_t *cuts_out)
{
    if (!whole || !cuts_out || n_shards == 0) return ESIM_EINVAL;
    const uint32_t N = whole->n_citizens, B = whole->n_buildings, R = whole->n_rooms, A = whole->n_areas;
    std::vector<uint64_t> cum((size_t)A + 1, 0);
    if (!by_work) {
        for (uint32_t c = 0; c < N; ++c) cum[whole->building_area[whole->home_building[c]] + 1]++;
    } else {
        std::vector<uint32_t> res(B, 0), wrk(B, 0), part(R, 0);
        for (uint32_t c = 0; c < N; ++c) {
            res[whole->home_building[c]]++;
            if (whole->room[c] != ESIM_NO_ROOM && whole->room[c] < R) part[whole->room[c]]++;
            else if (whole->work_building[c] != whole->home_building[c]) wrk[whole->work_building[c]]++;
        }
        for (uint32_t c = 0; c < N; ++c) {
            uint64_t w = 1u + res[whole->home_building[c]];
            if (whole->room[c] != ESIM_NO_ROOM && whole->room[c] < R) w += part[whole->room[c]];
            else if (whole->work_building[c] != whole->home_building[c]) w += wrk[whole->work_building[c]];
            cum[whole->building_area[whole->home_building[c]] + 1] += w;
        }
    }
    for (uint32_t a = 0; a < A; ++a) cum[a + 1] += cum[a];
    cuts_out[0] = 0;
    for (uint32_t k = 1; k < n_shards; ++k) {
        const uint64_t want = (uint64_t)((unsigned __int128)cum[A] * k / n_shards);
        uint32_t cut = (uint32_t)(std::lower_bound(cum.begin(), cum.end(), want) - cum.begin());
        cut = std::max(cut, cuts_out[k - 1]);
        cuts_out[k] = std::min(cut, A);
    }
    cuts_out[n_shards] = A;
    return ESIM_OK;
}

extern "C" int esim_synth_create_shard(const esim_synth_spec *spec, uint32_t shard, uint32_t n_shards, esim_population *out)
{
    if (n_shards == 0 || shard >= n_shards) return ESIM_EINVAL;
    if (n_shards == 1) return synth_generate(spec, out);
    esim_population whole;
    int rc = synth_generate(spec, &whole);
    if (rc) return rc;
    std::vector<uint32_t> cuts(n_shards + 1, 0);
    rc = esim_shard_cuts(&whole, n_shards, 1, cuts.data());            // bands of about the same expected work
    if (!rc) rc = esim_shard_population(&whole, cuts.data(), n_shards, shard, out);
    esim_synth_free(&whole);
    return rc;
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
