"""Hand-over from the reference's population to the arrays `esim_upload_population` takes.

The reference keeps its population as `Vec<OutputArea>`, each with `citizens: Vec<Citizen>` and
`buildings: Vec<Box<dyn Building>>` (sim/src/models/output_area.rs:85-100).  `Citizen`, `CitizenID`,
`BuildingID`, `OutputAreaID`, `Household`, `Workplace`, `School` and `Class` all derive or implement
`serde::Serialize` (citizen.rs:51,109; building.rs:61,142-159,161,219,310,331; output_area.rs:41), so a
maintainer can dump what `SimulatorBuilder::build` (simulator_builder.rs:1162-1292) produced with one
`serde_json::to_writer` per area.  This module reads that JSON shape -- field names exactly as serde writes
them -- and produces a `Population`; `population_to_reference_json` writes the same shape from a
`Population` (used by the tests, and as documentation of what is expected).

One area (list element) looks like:
  {"output_area_id": {"code": "E00067299", "index": 0},
   "citizens":  [{"id": {"global_index": 0, "uuid_id": "..."}, "age": 34,
                  "household_code": {"output_area_id": {"code": .., "index": 0}, "building_index": 3,
                                     "building_unique_id": "...", "building_type": "Household"},
                  "workplace_code": {... "building_type": "Workplace" | "School" | "Household"},
                  "occupation": "Student" | "Unemployed" | {"Normal": {"occupation": "Teaching"}} | {"Essential": {..}},
                  "start_working_hour": 9, "end_working_hour": 17,
                  "disease_status": "Susceptible" | {"Infected": 0} | ...,
                  "is_mask_compliant": true, "uses_public_transport": false, ...}, ...],
   "buildings": [{"building_code": {...}, "occupants": [{"global_index": ..}, ..], ...},            # Household / Workplace
                 {"building_code": {...}, "classes": [{"students": [ids], "teacher": id}, ..],
                  "offices": [[ids], ..]}, ...]}                                                     # School
Only the fields named above are read; `uuid`s, polygons and locations are ignored.
"""
import numpy as np

from . import _lib
from .population import Population

_BUILDING_TYPES = {"Household": _lib.HOUSEHOLD, "Workplace": _lib.WORKPLACE, "School": _lib.SCHOOL}
# occupation byte: 0 unemployed, 1 student, 2 + OccupationType index (citizen.rs:299-309), +16 when Essential
_OCCUPATION_TYPES = ("Manager", "Professional", "Technical", "Administrative", "SkilledTrades", "Caring", "Sales",
                     "MachineOperatives", "Teaching")


class ReferenceFormatError(ValueError):
    pass


def _occupation_byte(o):
    if o == "Unemployed":
        return 0
    if o == "Student":
        return 1
    if isinstance(o, dict) and len(o) == 1:
        kind, inner = next(iter(o.items()))
        if kind in ("Normal", "Essential") and inner.get("occupation") in _OCCUPATION_TYPES:
            return 2 + _OCCUPATION_TYPES.index(inner["occupation"]) + (16 if kind == "Essential" else 0)
    raise ReferenceFormatError("unknown occupation %r" % (o,))


def _occupation_json(b):
    if b == 0:
        return "Unemployed"
    if b == 1:
        return "Student"
    kind = "Essential" if b >= 18 else "Normal"
    return {kind: {"occupation": _OCCUPATION_TYPES[(b - 2) % 16]}}


def population_from_reference_json(areas, expect_hours=(9, 17)):
    """`areas`: the list described in the module docstring (already parsed JSON).  Returns
    (Population, area_codes).  Citizens are indexed by `id.global_index` (must be 0..N-1), buildings by area order
    and `building_index`, school rooms by (school, classes then offices) order.  Citizens whose `disease_status`
    is `{"Infected": _}` become the seeds (simulator_builder.rs:1268-1287)."""
    areas = sorted(areas, key=lambda a: a["output_area_id"]["index"])
    area_index = {}
    for pos, a in enumerate(areas):
        if a["output_area_id"]["index"] != pos:
            raise ReferenceFormatError("output area indexes must be 0..n-1 without gaps")
        area_index[a["output_area_id"]["code"]] = pos
    bld_base = np.zeros(len(areas) + 1, np.int64)
    for pos, a in enumerate(areas):
        for k, b in enumerate(a["buildings"]):
            code = b["building_code"]
            if code["building_index"] != k or code["output_area_id"]["index"] != pos:
                raise ReferenceFormatError("building %d of area %s carries code %r" % (k, a["output_area_id"]["code"], code))
        bld_base[pos + 1] = bld_base[pos] + len(a["buildings"])
    n_bld = int(bld_base[-1])
    n = sum(len(a["citizens"]) for a in areas)
    building_area = np.zeros(n_bld, np.uint32)
    building_type = np.zeros(n_bld, np.uint8)
    home = np.full(n, 0xFFFFFFFF, np.uint32)
    work = np.zeros(n, np.uint32)
    room = np.full(n, _lib.NO_ROOM, np.uint32)
    flags = np.zeros(n, np.uint8)
    age = np.zeros(n, np.uint16)
    occupation = np.zeros(n, np.uint8)
    seeds = []
    room_building = []

    def gid(code):
        a = code["output_area_id"]["index"]
        if not (0 <= a < len(areas)) or not (0 <= code["building_index"] < bld_base[a + 1] - bld_base[a]):
            raise ReferenceFormatError("building code out of range: %r" % (code,))
        return int(bld_base[a]) + code["building_index"]

    for pos, a in enumerate(areas):
        for k, b in enumerate(a["buildings"]):
            g = int(bld_base[pos]) + k
            building_area[g] = pos
            building_type[g] = _BUILDING_TYPES.get(b["building_code"].get("building_type"), _lib.WORKPLACE)
            if "classes" in b:                                     # School (building.rs:331-342): classes, then offices
                building_type[g] = _lib.SCHOOL
                rooms = [[s["global_index"] for s in c["students"]] + [c["teacher"]["global_index"]] for c in b["classes"]]
                rooms += [[s["global_index"] for s in office] for office in b.get("offices", [])]
                for members in rooms:
                    r = len(room_building)
                    room_building.append(g)
                    for c in members:
                        if not (0 <= c < n):
                            raise ReferenceFormatError("school %d names citizen %d" % (g, c))
                        room[c] = r
        for c in a["citizens"]:
            i = c["id"]["global_index"]
            if not (0 <= i < n) or home[i] != 0xFFFFFFFF:
                raise ReferenceFormatError("citizen global_index %r is out of range or repeated" % (i,))
            if (c.get("start_working_hour", expect_hours[0]), c.get("end_working_hour", expect_hours[1])) != tuple(expect_hours):
                raise ReferenceFormatError("citizen %d works %r-%r: the device schedule is global (DESIGN.md 3.2)"
                                           % (i, c.get("start_working_hour"), c.get("end_working_hour")))
            home[i] = gid(c["household_code"])
            work[i] = gid(c["workplace_code"])
            flags[i] = (_lib.FLAG_USES_PUBLIC_TRANSPORT if c.get("uses_public_transport") else 0) | \
                       (_lib.FLAG_MASK_COMPLIANT if c.get("is_mask_compliant") else 0)
            age[i] = c.get("age", 0)
            occupation[i] = _occupation_byte(c.get("occupation", "Unemployed"))
            st = c.get("disease_status", "Susceptible")
            if isinstance(st, dict) and "Infected" in st:
                seeds.append(i)
            elif st != "Susceptible":
                raise ReferenceFormatError("citizen %d starts as %r: only Susceptible / Infected are built by the reference" % (i, st))
    school_member = building_type[work] == _lib.SCHOOL
    if (school_member & (home != work) & (room == _lib.NO_ROOM)).any():
        raise ReferenceFormatError("a school member is in no class or office of its school")
    room[~school_member] = _lib.NO_ROOM
    pop = Population(home_building=home, work_building=work, room=room, flags=flags, age=age, occupation=occupation,
                     building_area=building_area, building_type=building_type,
                     room_building=np.asarray(room_building, np.uint32), seeds=np.asarray(sorted(seeds), np.uint32),
                     n_areas=len(areas))
    return pop, [a["output_area_id"]["code"] for a in areas]


def population_to_reference_json(pop, area_codes=None):
    """The inverse: the serde shape of the reference for a `Population` (whole population, not a shard)."""
    if pop.citizen_id_base != 0 or pop.n_citizens_global != pop.n_citizens:
        raise ValueError("population_to_reference_json takes a whole population")
    codes = area_codes or ["OA%07d" % a for a in range(pop.n_areas)]
    first = np.zeros(pop.n_areas + 1, np.int64)
    np.add.at(first, pop.building_area.astype(np.int64) + 1, 1)
    first = np.cumsum(first)
    order = np.argsort(pop.building_area, kind="stable")           # buildings grouped by area, original order inside
    local = np.zeros(pop.n_buildings, np.int64)
    local[order] = np.arange(pop.n_buildings) - first[pop.building_area[order]]
    names = {v: k for k, v in _BUILDING_TYPES.items()}

    def code(g):
        a = int(pop.building_area[g])
        return {"output_area_id": {"code": codes[a], "index": a}, "building_index": int(local[g]),
                "building_unique_id": "00000000-0000-0000-0000-%012x" % g, "building_type": names[int(pop.building_type[g])]}

    def cid(i):
        return {"global_index": int(i), "uuid_id": "00000000-0000-0000-0000-%012x" % int(i)}

    seeds = set(int(s) for s in pop.seeds)
    areas = [{"output_area_id": {"code": codes[a], "index": a}, "citizens": [], "buildings": []} for a in range(pop.n_areas)]
    occupants = [[] for _ in range(pop.n_buildings)]
    for i in range(pop.n_citizens):
        occupants[int(pop.home_building[i])].append(i)
        if pop.work_building[i] != pop.home_building[i] and pop.building_type[pop.work_building[i]] != _lib.SCHOOL:
            occupants[int(pop.work_building[i])].append(i)
    rooms_of = {}
    for r, g in enumerate(pop.room_building):
        rooms_of.setdefault(int(g), []).append(r)
    members = [[] for _ in range(pop.n_rooms)]
    for i in np.nonzero(pop.room != _lib.NO_ROOM)[0]:
        members[int(pop.room[i])].append(int(i))
    for g in order:
        g = int(g)
        b = {"building_code": code(g), "location": {"x": 0, "y": 0}}
        if pop.building_type[g] == _lib.SCHOOL:
            # every room is written as an office (a list of ids): the device treats classes and offices alike
            b["classes"] = []
            b["offices"] = [[cid(i) for i in members[r]] for r in rooms_of.get(g, [])]
        else:
            b["occupants"] = [cid(i) for i in occupants[g]]
        areas[int(pop.building_area[g])]["buildings"].append(b)
    for i in range(pop.n_citizens):
        h = int(pop.home_building[i])
        areas[int(pop.building_area[h])]["citizens"].append({
            "id": cid(i), "age": int(pop.age[i]), "household_code": code(h), "workplace_code": code(int(pop.work_building[i])),
            "occupation": _occupation_json(int(pop.occupation[i])), "start_working_hour": 9, "end_working_hour": 17,
            "current_building_position": code(h), "disease_status": {"Infected": 0} if i in seeds else "Susceptible",
            "is_mask_compliant": bool(pop.flags[i] & _lib.FLAG_MASK_COMPLIANT),
            "uses_public_transport": bool(pop.flags[i] & _lib.FLAG_USES_PUBLIC_TRANSPORT), "on_public_transport": None})
    return areas
