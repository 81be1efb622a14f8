"""The hand-over from the reference's serde JSON shape of a built population -- CPU only."""
import json

import numpy as np
import pytest

import _oracle
from epidemicsimulator_amd import Population, _lib
from epidemicsimulator_amd.reference_io import (ReferenceFormatError, population_from_reference_json,
                                                population_to_reference_json)


def small_pop():
    return Population.synthetic("york", n_citizens=1500, n_areas=6, citizens_per_school=700, n_seeds=5)


def test_round_trip_keeps_every_array_the_path_reads():
    pop = small_pop()
    areas = json.loads(json.dumps(population_to_reference_json(pop)))       # through real JSON text
    back, codes = population_from_reference_json(areas)
    assert codes == ["OA%07d" % a for a in range(pop.n_areas)]
    assert back.n_citizens == pop.n_citizens and back.n_areas == pop.n_areas and back.n_buildings == pop.n_buildings
    # buildings are renumbered area by area; compare through the area / type of every citizen's buildings and the grouping
    for arr in ("flags", "age", "occupation"):
        assert (getattr(back, arr) == getattr(pop, arr)).all(), arr
    assert (back.building_area[back.home_building] == pop.building_area[pop.home_building]).all()
    assert (back.building_area[back.work_building] == pop.building_area[pop.work_building]).all()
    assert (back.building_type[back.work_building] == pop.building_type[pop.work_building]).all()
    same_home = lambda p: p.home_building[:, None] == p.home_building[None, :]
    idx = np.arange(0, pop.n_citizens, 7)
    assert (back.home_building[idx][:, None] == back.home_building[None, idx]).tolist() == \
           (pop.home_building[idx][:, None] == pop.home_building[None, idx]).tolist()
    assert sorted(back.seeds.tolist()) == sorted(set(pop.seeds.tolist()))
    in_room = pop.room != _lib.NO_ROOM
    assert ((back.room != _lib.NO_ROOM) == in_room).all()
    assert (back.room_building[back.room[in_room]] == back.work_building[in_room]).all()


def test_converted_population_runs_like_the_original_in_the_oracle():
    # same citizens, same buildings up to renumbering, same global citizen ids => identical Philox draws and records
    pop = small_pop()
    back, _ = population_from_reference_json(population_to_reference_json(pop))
    ep = _lib.default_params(exposure_chance=0.01, vaccination_threshold=0.05, seed=17)
    a = _oracle.Oracle(pop, _oracle.params_from_esim(ep)).run(300)
    b = _oracle.Oracle(back, _oracle.params_from_esim(ep)).run(300)
    for f in a.dtype.names:
        assert (a[f] == b[f]).all(), f


def test_serde_shape_of_the_reference_types_is_accepted():
    # hand-written in the layout serde derives for Citizen / BuildingID / Household / School (citizen.rs:109-135,
    # building.rs:61-67,161-168,310-342): a household of two, one of them a pupil of the one school, one worker
    def area(i):
        return {"code": "E0000000%d" % i, "index": i}

    def bcode(a, k, t):
        return {"output_area_id": area(a), "building_index": k, "building_unique_id": "x", "building_type": t}

    def cit(i, home, work, occ, status="Susceptible", pt=False):
        return {"id": {"global_index": i, "uuid_id": "u"}, "age": 30, "household_code": home, "workplace_code": work,
                "occupation": occ, "start_working_hour": 9, "end_working_hour": 17, "current_building_position": home,
                "disease_status": status, "is_mask_compliant": True, "uses_public_transport": pt, "on_public_transport": None}

    h0, w1, s1 = bcode(0, 0, "Household"), bcode(1, 0, "Workplace"), bcode(1, 1, "School")
    areas = [
        {"output_area_id": area(1), "citizens": [cit(2, bcode(1, 2, "Household"), s1, {"Normal": {"occupation": "Teaching"}})],
         "buildings": [{"building_code": w1, "occupants": [{"global_index": 1}], "floor_space": 100},
                       {"building_code": s1, "classes": [{"students": [{"global_index": 0}], "teacher": {"global_index": 2}}], "offices": []},
                       {"building_code": bcode(1, 2, "Household"), "occupants": [{"global_index": 2}]}]},
        {"output_area_id": area(0), "citizens": [cit(0, h0, s1, "Student"), cit(1, h0, w1, {"Essential": {"occupation": "Caring"}}, {"Infected": 0}, True)],
         "buildings": [{"building_code": h0, "occupants": [{"global_index": 0}, {"global_index": 1}]}]},
    ]
    pop, codes = population_from_reference_json(areas)
    assert codes == ["E00000000", "E00000001"]
    assert pop.home_building.tolist() == [0, 0, 3] and pop.work_building.tolist() == [2, 1, 2]
    assert pop.building_type.tolist() == [_lib.HOUSEHOLD, _lib.WORKPLACE, _lib.SCHOOL, _lib.HOUSEHOLD]
    assert pop.building_area.tolist() == [0, 1, 1, 1]
    assert pop.room.tolist() == [0, _lib.NO_ROOM, 0] and pop.room_building.tolist() == [2]
    assert pop.seeds.tolist() == [1] and pop.flags.tolist() == [2, 3, 2]
    assert pop.occupation.tolist() == [1, 2 + 5 + 16, 2 + 8]
    _oracle.Oracle(pop, _oracle.params_from_esim(_lib.default_params())).run(48)      # accepted by the path's checks


def test_what_the_device_cannot_represent_is_refused():
    areas = population_to_reference_json(small_pop())
    areas = [a for a in areas if len(a["citizens"]) > 1] + [a for a in areas if len(a["citizens"]) <= 1]
    areas[0]["citizens"][0]["start_working_hour"] = 7
    with pytest.raises(ReferenceFormatError, match="schedule is global"):
        population_from_reference_json(areas)
    areas = population_to_reference_json(small_pop())
    areas = [a for a in areas if len(a["citizens"]) > 1] + [a for a in areas if len(a["citizens"]) <= 1]
    areas[1]["citizens"][0]["id"]["global_index"] = areas[1]["citizens"][1]["id"]["global_index"]
    with pytest.raises(ReferenceFormatError, match="repeated"):
        population_from_reference_json(areas)


@pytest.mark.gpu
def test_converted_population_runs_on_the_gpu_like_the_oracle():
    # a population that arrives in the reference's own serde JSON shape (SimulatorBuilder's output, dumped with
    # serde_json) is converted and fed to the HIP path; records and per-citizen state equal the oracle's on the same arrays
    from epidemicsimulator_amd import Simulator
    src = Population.synthetic("york", n_citizens=9000, n_areas=30, citizens_per_school=3000, n_seeds=15, p_public_transport=0.4)
    pop, codes = population_from_reference_json(json.loads(json.dumps(population_to_reference_json(src))))
    assert len(codes) == src.n_areas and pop.n_citizens == src.n_citizens
    ep = _lib.default_params(exposure_chance=0.004, vaccination_rate=40, vaccination_threshold=0.02, lockdown_threshold=0.03,
                             mask_pt_threshold=0.005, mask_everywhere_threshold=0.01, seed=5, max_steps=600)
    sim = Simulator(pop, ep, area_codes=codes)
    got = sim.run(600)
    orc = _oracle.Oracle(pop, _oracle.params_from_esim(ep))
    want = orc.run(600)
    for f in ("susceptible", "exposed", "infected", "recovered", "vaccinated", "exposures_building", "exposures_bus", "vaccinated_now", "eligible_count", "lockdown", "mask_status"):
        assert (got[f] == want[f]).all(), f
    assert want["vaccinated"][-1] > 0 and want["exposures_bus"].sum() > 0
    g, o = sim.download_state(), orc.state()
    for k in ("status", "timer", "current_building", "on_bus", "eligible"):
        assert (g[k] == o[k]).all(), k
    # the per-Output-Area exposure series are keyed by the reference's area codes
    series = sim.exposures_per_output_area(codes)
    assert set(series) <= set(codes) and sum(sum(v) for v in series.values()) == int(want["exposures_building"].sum())
    sim.close()
