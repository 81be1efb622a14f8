"""Parity of the HIP path (through the C ABI) with the CPU oracle: every field of every per-step
record, and the full per-citizen state, bit-exact under the same Philox seed."""
import numpy as np
import pytest

import _oracle
from epidemicsimulator_amd import Population, Simulator, _lib

pytestmark = pytest.mark.gpu

FIELDS = [f for f in _lib.RECORD_FIELDS if f != "reserved"]


def assert_same_records(gpu, orc):
    assert len(gpu) == len(orc)
    for f in FIELDS:
        a, b = gpu[f], orc[f]
        if not (a == b).all():
            i = int(np.argmax(a != b))
            raise AssertionError("field %s differs first at record %d (step %d): gpu %d oracle %d"
                                 % (f, i, int(orc["time_step"][i]), int(a[i]), int(b[i])))


def assert_same_state(sim, orc):
    g, o = sim.download_state(), orc.state()
    for k in ("status", "timer", "current_building", "on_bus", "eligible"):
        assert (g[k] == o[k]).all(), k


# every scenario runs six ways: time-parallel chunks all the way (default: all steps of a chunk drawn in one pass, under a
# vaccination programme with the chunk's vaccinations planned ahead), time-parallel chunks until a programme starts and
# sequential steps from there, chunks as one kernel per step, and sequential steps only with the default hand-over between the
# persistent single-workgroup kernel and the multi-workgroup kernels, multi-workgroup kernels only, persistent kernel only
SMALL_LIMITS = ("pmap", "vax", "wide", "tinymax", "tp", "pipe", None, 0, 1 << 30)      # the nine execution forms


def run_both(pop, steps, check_state_every=None, small_limits=SMALL_LIMITS, **params):
    for lim in small_limits:
        try:
            _run_both(pop, steps, check_state_every, lim, **params)
        except AssertionError as e:
            raise AssertionError("execution form %r: %s" % (lim, e)) from e


def _run_both(pop, steps, check_state_every, small_limit, **params):
    ep = _lib.default_params(**params)
    sim = Simulator(pop, ep)
    if small_limit == "pmap":
        sim.set_pipeline(4)                       # ... on the persistent item map (entered once per infection, not per chunk)
    elif small_limit == "vax":
        sim.set_pipeline(3)                       # time-parallel chunks, also under a vaccination programme (default)
    elif small_limit == "wide":
        sim.set_pipeline(3); sim.set_tiny_chunk_limit(0)         # ... every chunk in the seven-kernel form, however few the Infected
    elif small_limit == "tinymax":
        sim.set_pipeline(3); sim.set_tiny_chunk_limit(1 << 30)   # ... every chunk tried as one launch first (it declines beyond 64 Infected)
    elif small_limit == "tp":
        sim.set_pipeline(2)                       # time-parallel chunks until a vaccination programme starts
    elif small_limit == "pipe":
        sim.set_pipeline(1)                       # one kernel per step
    else:
        sim.set_pipeline(0)                       # sequential steps
        if small_limit is not None:
            sim.set_small_step_limit(small_limit)
    orc = _oracle.Oracle(pop, _oracle.params_from_esim(ep))
    if check_state_every:
        done = 0
        while done < steps:
            n = min(check_state_every, steps - done)
            assert_same_records(sim.run(n), orc.run(n))
            assert_same_state(sim, orc)
            done += n
    else:
        assert_same_records(sim.run(steps), orc.run(steps))
        assert_same_state(sim, orc)
    sim.close()
    return orc


AGGRESSIVE = dict(exposure_chance=0.004, vaccination_rate=40, vaccination_threshold=0.02,
                  lockdown_threshold=0.03, mask_pt_threshold=0.005, mask_everywhere_threshold=0.01, seed=77)


def test_small_world_all_paths_match_golden_and_oracle():
    import json, os
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "oracle_small_world.json")))
    pop = Population.synthetic("york", **g["spec"])
    sim = Simulator(pop, _lib.default_params(**g["params"]))
    rec = sim.run(g["steps"])
    for k, want in g["records"].items():
        assert rec[k].tolist() == want, k
    # the fixture exercises every branch
    assert rec["vaccinated"][-1] > 0 and rec["exposures_bus"].sum() > 0 and rec["lockdown"].sum() > 0
    assert set(rec["mask_status"].tolist()) == {0, 1, 2}
    sim.close()


def test_small_world_state_every_step_window():
    pop = Population.synthetic("york", n_citizens=6000, n_areas=24, citizens_per_school=1500, n_seeds=12)
    run_both(pop, 300, check_state_every=7, **AGGRESSIVE)


def test_step_by_step_equals_run():
    pop = Population.synthetic("york", n_citizens=3000, n_areas=10, citizens_per_school=1500, n_seeds=8)
    ep = _lib.default_params(**AGGRESSIVE)
    a, b = Simulator(pop, ep), Simulator(pop, ep)
    ra = a.run(120)
    for i in range(120):
        alive = b.step()
        assert alive == bool(ra["disease_exists"][i])
        for f in FIELDS:
            assert b.last[f] == int(ra[f][i]), (f, i)
    a.close(); b.close()


def test_york_default_params_1000_steps():
    run_both(Population.synthetic("york"), 1000, small_limits=("tp", "pipe", 0))


def test_york_full_5000_steps_vaccination_85():
    # BASELINE.json configs[1]: York, 5000 steps, fixed Philox seed vs CPU counts; v1.7.1's rate 85/step
    # (forms: planned chunks under the programme -- the default, level 3 --, chunks until it starts, one kernel per step, sequential)
    run_both(Population.synthetic("york"), 5000, small_limits=("pmap", "vax", "tp", "pipe", None), vaccination_rate=85, vaccination_threshold=0.003)


def test_yh_census_config_1500_steps():
    # BASELINE.json configs[2]: Yorkshire & Humber (5 249 772 citizens, 17 246 Output Areas) on one GPU;
    # 1500 of the 5000 steps keeps the oracle within ~20 s
    run_both(Population.synthetic("yh_census"), 1500, small_limits=("tp", "pipe"))


def test_big_routes_and_u8_truncation():
    # one very large Output Area: routes of > 64 riders (workgroup path), a school of > 256 members so the
    # infected count passes 255 (`as u8`, Q6), household and workplace draws in the same area
    pop = Population.synthetic("york", n_citizens=9000, n_areas=2, citizens_per_school=9000, n_seeds=40,
                               p_public_transport=0.5)
    orc = run_both(pop, 700, check_state_every=100, exposure_chance=0.02, vaccination_threshold=0.9,
                   lockdown_threshold=0.95, mask_pt_threshold=0.2, mask_everywhere_threshold=0.4, seed=5)


def test_route_longer_than_the_one_pass_limit():
    # a single Output Area: one route of several thousand riders (> CHUNK_ROUTE_MAX = 2048), so no chunk may take the
    # one-pass form and level 2 has to fall back to one kernel per step by itself
    pop = Population.synthetic("york", n_citizens=12000, n_areas=1, citizens_per_school=6000, n_seeds=30,
                               p_public_transport=0.6)
    run_both(pop, 400, check_state_every=200, small_limits=("tp", None), exposure_chance=0.01, seed=23)


def test_lockdown_freeze_on_a_bus_hour():
    # lockdown decided at the end of an hour-8 step keeps riders on the bus every step (Q8)
    pop = Population.synthetic("york", n_citizens=4000, n_areas=8, citizens_per_school=2000, n_seeds=30)
    ep = dict(exposure_chance=0.01, lockdown_threshold=0.0074, vaccination_threshold=0.5, seed=11)
    orc = run_both(pop, 200, check_state_every=50, **ep)


def test_tiny_and_degenerate_populations():
    # fewer eligible citizens than the vaccination rate: everybody eligible is vaccinated (choose_multiple)
    pop = Population.synthetic("york", n_citizens=500, n_areas=3, citizens_per_school=500, n_seeds=5)
    run_both(pop, 400, check_state_every=50, exposure_chance=0.01, vaccination_threshold=0.011, seed=3)
    # a single citizen, no seeds: disease_exists stays true while S != 0 (statistics.rs:289-291)
    one = Population(home_building=np.zeros(1, np.uint32), work_building=np.zeros(1, np.uint32),
                     flags=np.zeros(1, np.uint8), building_area=np.zeros(1, np.uint32),
                     building_type=np.zeros(1, np.uint8), seeds=np.zeros(0, np.uint32), n_areas=1)
    run_both(one, 30)


def test_stop_when_done_matches_reference_loop():
    # everyone infected at the start and no susceptibles => the loop ends when the last one recovers
    n = 64
    home = np.arange(n, dtype=np.uint32) // 4
    pop = Population(home_building=home, work_building=home.copy(), flags=np.zeros(n, np.uint8),
                     building_area=np.zeros(n // 4, np.uint32), building_type=np.zeros(n // 4, np.uint8),
                     seeds=np.arange(n, dtype=np.uint32), n_areas=1)
    ep = _lib.default_params(vaccination_threshold=2.0)
    sim = Simulator(pop, ep)
    orc = _oracle.Oracle(pop, _oracle.params_from_esim(ep))
    g, o = sim.run(600, stop_when_done=True), orc.run(600, stop_when_done=True)
    assert len(g) == len(o) == 337 and g["disease_exists"][-1] == 0
    assert_same_records(g, o)
    sim.close()


def test_read_access_for_visualisation():
    # citizen_output_area_lookup / per-area infected counts rebuilt from the device state
    pop = Population.synthetic("york", n_citizens=5000, n_areas=16, citizens_per_school=2500)
    ep = _lib.default_params(**AGGRESSIVE)
    sim = Simulator(pop, ep)
    orc = _oracle.Oracle(pop, _oracle.params_from_esim(ep))
    sim.run(130); orc.run(130)                                   # hour 10: commuters are at work
    ost = orc.state()
    area, local = sim.citizen_output_area_lookup()
    want_area = pop.building_area[ost["current_building"]]
    assert (area == want_area).all() and (area != pop.building_area[pop.home_building]).any()
    for a in (0, 7, 15):
        members = np.nonzero(area == a)[0]
        assert local[members].tolist() == list(range(len(members)))
    assert sim.infected_per_area().tolist() == np.bincount(want_area[ost["status"] == _lib.INFECTED], minlength=16).tolist()
    sim.close()


def test_exposure_log_and_per_output_area_series_match_the_oracle():
    # every add_exposure call of the run (statistics.rs:181-195): who, when, on a bus or in a building of which Output Area --
    # through chunks, the vaccination programme (sequential steps; vaccinated citizens keep their entry) and a lockdown
    pop = Population.synthetic("york", n_citizens=6000, n_areas=20, citizens_per_school=3000, n_seeds=12, p_public_transport=0.4)
    ep = _lib.default_params(**AGGRESSIVE)
    for level in (2, 0):
        sim = Simulator(pop, ep)
        sim.set_pipeline(level)
        orc = _oracle.Oracle(pop, _oracle.params_from_esim(ep))
        sim.run(420); orc.run(420)
        o_step, o_area = orc.exposures()
        cit, step, bus = sim.exposure_events()
        want = np.nonzero(o_step)[0]
        assert len(cit) == len(want) > 500
        order = np.lexsort((want, o_step[want]))
        assert (cit == want[order]).all() and (step == o_step[want][order]).all()
        assert (bus == (o_area[want][order] == 0xFFFFFFFF)).all() and bus.any() and not bus.all()
        per_area = sim.exposures_per_output_area()
        ref = {}
        b = o_area != 0xFFFFFFFF
        for a in np.unique(o_area[b & (o_step > 0)]):
            steps = o_step[b & (o_step > 0) & (o_area == a)]
            ref["OA%07d" % a] = np.unique(steps, return_counts=True)[1].tolist()
        assert per_area == ref
        assert sum(sum(v) for v in per_area.values()) == int((bus == 0).sum())
        sim.close()


def test_checkpoint_resume_continues_bit_for_bit(tmp_path):
    # save in the middle of the vaccination programme, continue in a NEW context: records, state and exposure log as if
    # the run had never stopped; a checkpoint of another population or seed is refused
    pop = Population.synthetic("york", n_citizens=6000, n_areas=20, citizens_per_school=3000, n_seeds=12)
    ep = _lib.default_params(**AGGRESSIVE)
    whole = Simulator(pop, ep)
    ref = whole.run(500)
    for at in (130, 330):                                        # inside a chunked stretch / inside the vaccination programme
        a = Simulator(pop, ep)
        first = a.run(at)
        path = str(tmp_path / ("ckpt%d.bin" % at))
        a.save_checkpoint(path)
        a.close()
        b = Simulator(pop, ep)
        b.load_checkpoint(path)
        assert b._steps == at and len(b.statistics_recorder.global_stats) == at
        rest = b.run(500 - at)
        assert_same_records(np.concatenate([first, rest]), ref)
        sa, sb = whole.download_state(), b.download_state()
        assert all((sa[k] == sb[k]).all() for k in sa)
        ea, eb = whole.exposure_events(), b.exposure_events()
        assert all((x == y).all() for x, y in zip(ea, eb))
        b.close()
    other = Simulator(pop, _lib.default_params(**dict(AGGRESSIVE, seed=78)))
    with pytest.raises(_lib.EsimError, match="another population"):
        other.load_checkpoint(path)
    other.close(); whole.close()
    # a checkpoint written by a build that lays the state out differently (its header carries a hash of the citizen word's
    # fields and of the control block's layout: CkptHeader.layout_id, the 16th u32 of the file) is refused, not reinterpreted
    raw = np.fromfile(path, np.uint8)
    raw[15 * 4] ^= 0x5A
    bad = str(tmp_path / "other_layout.bin")
    raw.tofile(bad)
    c = Simulator(pop, ep)
    with pytest.raises(_lib.EsimError, match="another state layout"):
        c.load_checkpoint(bad)
    c.load_checkpoint(path)                                   # the context is still usable, and the genuine file still goes in
    assert_same_records(np.concatenate([first, c.run(500 - at)]), ref)
    c.close()


def test_simulate_after_a_resume_finishes_the_same_run(tmp_path, capsys):
    # simulate() on a context that has already run (here: resumed from a checkpoint at step 330 of 500) continues from the
    # context's clock: it must stay inside the step budget and write the files of the uninterrupted run
    import json
    pop = Population.synthetic("york", n_citizens=6000, n_areas=20, citizens_per_school=3000, n_seeds=12)
    ep = _lib.default_params(max_steps=500, **AGGRESSIVE)
    whole = Simulator(pop, ep)
    whole.simulate(str(tmp_path) + "/whole/")
    lines_whole = capsys.readouterr().out.count("Completed  50 time steps")
    a = Simulator(pop, ep)
    a.run(330)
    a.save_checkpoint(str(tmp_path / "c.bin"))
    a.close()
    b = Simulator(pop, ep)
    b.load_checkpoint(str(tmp_path / "c.bin"))
    b.simulate(str(tmp_path) + "/resumed/")
    lines_resumed = capsys.readouterr().out.count("Completed  50 time steps")
    for name in ("global_stats.json", "exposures.json"):
        assert json.load(open(str(tmp_path) + "/whole/" + name)) == json.load(open(str(tmp_path) + "/resumed/" + name)), name
    assert lines_whole == 10 and lines_resumed == 3               # after the steps with index 0, 50, ..., 450 / 350, 400, 450
    whole.close(); b.close()


def test_reset_reproduces_the_run():
    pop = Population.synthetic("york", n_citizens=5000, n_areas=16, citizens_per_school=2500)
    sim = Simulator(pop, _lib.default_params(**AGGRESSIVE))
    a = sim.run(200)
    sim.reset()
    b = sim.run(200)
    assert_same_records(a, b)
    sim.close()


def test_size_independent_properties_large():
    # at a size the oracle would take minutes for: conservation, monotone S, determinism across contexts
    pop = Population.synthetic("syn3m5")
    ep = _lib.default_params(exposure_chance=0.002)
    s1, s2 = Simulator(pop, ep), Simulator(pop, ep)
    a, b = s1.run(400), s2.run(400)
    assert_same_records(a, b)
    tot = a["susceptible"] + a["exposed"] + a["infected"] + a["recovered"] + a["vaccinated"]
    assert (tot == pop.n_citizens).all()
    assert (np.diff(a["susceptible"].astype(np.int64)) <= 0).all()
    st = s1.download_state()
    assert np.bincount(st["status"], minlength=5).tolist() != []      # decodes
    s1.close(); s2.close()


def test_simulate_writes_the_reference_output_files(tmp_path):
    # Simulator::simulate (simulator.rs:108-127) + StatisticsRecorder::dump_to_file (statistics.rs:113-150)
    import json
    pop = Population.synthetic("york", n_citizens=3000, n_areas=10, citizens_per_school=1500, n_seeds=8)
    ep = _lib.default_params(max_steps=60, **{k: v for k, v in AGGRESSIVE.items()})
    sim = Simulator(pop, ep, record_timings=True)
    out = str(tmp_path) + "/run/"
    sim.simulate(out)
    stats = json.load(open(out + "global_stats.json"))
    assert len(stats) == 61                                            # 60 steps + the all-zero entry of Q14
    assert stats[-1] == {"time_step": 61, "susceptible": 0, "exposed": 0, "infected": 0, "recovered": 0, "vaccinated": 0}
    assert set(stats[0]) == {"time_step", "susceptible", "exposed", "infected", "recovered", "vaccinated"}
    orc = _oracle.Oracle(pop, _oracle.params_from_esim(ep)).run(60)
    for i in (0, 17, 59):
        assert stats[i]["infected"] == int(orc["infected"][i]) and stats[i]["susceptible"] == int(orc["susceptible"][i])
    timings = json.load(open(out + "timings.json"))
    assert len(timings) == 60
    assert set(timings[0]) == {"Generate Exposures", "Apply Exposures", "Apply Interventions", "total"}
    assert all(t["total"] > 0 for t in timings)
    exposures = json.load(open(out + "exposures.json"))
    assert exposures["All"]["All"] == [int(a) + int(b) for a, b in zip(orc["exposures_building"], orc["exposures_bus"])]
    assert sum(sum(v) for v in exposures["OutputArea"].values()) == int(orc["exposures_building"].sum())
    assert len(json.load(open(out + "memory.json"))) == 60
    sim.close()


def test_simulate_device_resident_blocks(tmp_path, capsys):
    # without per-step timers simulate() runs the steps between two progress lines on the device in one go; same
    # records, same files, same number of progress lines (after the steps with index 0, 50, 100)
    import json
    pop = Population.synthetic("york", n_citizens=3000, n_areas=10, citizens_per_school=1500, n_seeds=8)
    ep = _lib.default_params(max_steps=130, **{k: v for k, v in AGGRESSIVE.items()})
    sim = Simulator(pop, ep)
    out = str(tmp_path) + "/run/"
    sim.simulate(out)
    assert capsys.readouterr().out.count("Completed  50 time steps") == 3
    stats = json.load(open(out + "global_stats.json"))
    orc = _oracle.Oracle(pop, _oracle.params_from_esim(ep)).run(130)
    assert len(stats) == 131
    for f in ("susceptible", "exposed", "infected", "recovered", "vaccinated"):
        assert [e[f] for e in stats[:-1]] == [int(x) for x in orc[f]]
    assert [e["time_step"] for e in stats] == list(range(1, 132))
    timings = json.load(open(out + "timings.json"))
    assert len(timings) == 130 and all(t["total"] > 0 for t in timings)
    # the reference's three phase keys (simulator.rs:137-143 -> timings.json) in a device-resident run: apportioned from the
    # chunk kernels' HIP events (marks + fold / draw + units / the rest); every block that ran as chunk passes carries them
    phased = [t for t in timings if len(t) > 1]
    assert len(phased) >= 100
    assert all(set(t) == {"Generate Exposures", "Apply Exposures", "Apply Interventions", "total"} for t in phased)
    assert all(t["Generate Exposures"] > 0 and t["Apply Exposures"] > 0 and t["Apply Interventions"] > 0 for t in phased)
    assert len(json.load(open(out + "memory.json"))) == 130
    sim.close()
    # an epidemic that dies out stops the loop at the same step as the reference's `if !self.step() { break }`
    n = 64
    home = np.arange(n, dtype=np.uint32) // 4
    pop1 = Population(home_building=home, work_building=home.copy(), flags=np.zeros(n, np.uint8),
                      building_area=np.zeros(n // 4, np.uint32), building_type=np.zeros(n // 4, np.uint8),
                      seeds=np.arange(n, dtype=np.uint32), n_areas=1)
    sim1 = Simulator(pop1, _lib.default_params(vaccination_threshold=2.0, max_steps=2000))
    sim1.simulate(str(tmp_path) + "/gone/")
    gone = json.load(open(str(tmp_path) + "/gone/global_stats.json"))
    assert len(gone) == 337 + 1 and gone[336]["recovered"] == n
    sim1.close()


def random_population(seed, n=700, n_areas=5, n_buildings=90, n_schools=3, rooms_per_school=4):
    """Anything the ABI allows, not just what the reference's builder produces: workplaces in other areas,
    people working in somebody's household, tiny and empty buildings, rooms with one member, citizens that
    are not sorted by home (exercises res_idx), duplicate seeds."""
    rng = np.random.default_rng(seed)
    b_area = rng.integers(0, n_areas, n_buildings).astype(np.uint32)
    b_type = rng.choice([_lib.HOUSEHOLD, _lib.WORKPLACE], n_buildings, p=[0.7, 0.3]).astype(np.uint8)
    schools = rng.choice(n_buildings, n_schools, replace=False)
    b_type[schools] = _lib.SCHOOL
    room_bld = np.repeat(schools, rooms_per_school).astype(np.uint32)
    not_school = np.nonzero(b_type != _lib.SCHOOL)[0]
    home = rng.choice(not_school, n).astype(np.uint32)                 # unsorted on purpose
    work = home.copy()
    room = np.full(n, _lib.NO_ROOM, np.uint32)
    kind = rng.random(n)
    w = kind < 0.45                                                    # works in any non-school building, any area
    work[w] = rng.choice(not_school, int(w.sum()))
    sc = (kind >= 0.45) & (kind < 0.8)                                 # school member in a random room
    r = rng.integers(0, len(room_bld), int(sc.sum()))
    room[sc] = r
    work[sc] = room_bld[r]
    flags = (rng.random(n) < 0.5).astype(np.uint8) | ((rng.random(n) < 0.6).astype(np.uint8) << 1)
    seeds = rng.integers(0, n, 9).astype(np.uint32)
    seeds[-1] = seeds[0]
    return Population(home_building=home, work_building=work, room=room, flags=flags, building_area=b_area,
                      building_type=b_type, room_building=room_bld, seeds=seeds, n_areas=n_areas)


def test_one_huge_workplace_overflows_a_unit_queue():
    # 50 000 citizens who all work in ONE building: its worker list is 50 000 members x up to 25 slots of four steps = 1 250 000
    # (member, slot) pairs per chunk = 1221 units of 1024, more than the queue it goes to holds (1024): the producing wavefront
    # has to draw the list itself; households of four, a second area so that half of the workers fail the same-area filter
    n = 50000
    home = (np.arange(n, dtype=np.uint32) // 4)
    n_home = int(home.max()) + 1
    work = np.full(n, n_home, np.uint32)
    area = np.concatenate([(np.arange(n_home) % 2).astype(np.uint32), [0]]).astype(np.uint32)
    btype = np.concatenate([np.zeros(n_home, np.uint8), [_lib.WORKPLACE]]).astype(np.uint8)
    flags = (np.arange(n) % 3 == 0).astype(np.uint8) * 2
    pop = Population(home_building=home, work_building=work, flags=flags, building_area=area, building_type=btype,
                     seeds=np.arange(0, 40, 2, dtype=np.uint32), n_areas=2)
    run_both(pop, 250, check_state_every=125, small_limits=("tp", None), exposure_chance=0.00002, seed=99,
             vaccination_threshold=2.0, lockdown_threshold=2.0)


def test_many_infected_several_chunks_per_call():
    # found by tools/fuzz_parity.py (seed 1124): with more than 1024 Infected the chunk's log scatter is a kernel of its
    # own that runs AFTER the next chunk's decisions; two such chunks in one call must not share write cursors.  Big
    # buildings (more Infected per building than an item has records => the spilled per-step counters), a single school
    # room of ~900, short chunks (exposed_time 30), a lockdown threshold that toggles.
    pop = random_population(1124, n=2500, n_areas=12, n_buildings=40, n_schools=1, rooms_per_school=1)
    params = dict(exposure_chance=0.0005, seed=587668745022, vaccination_threshold=2.0, lockdown_threshold=0.01,
                  mask_pt_threshold=0.005, mask_everywhere_threshold=0.2, bus_capacity=2, exposed_time=30, infected_time=336)
    for block in (50, 160, 500):
        run_both(pop, 500, check_state_every=block, small_limits=("tp",), **params)


@pytest.mark.parametrize("grid", (16, 48))
def test_chunk_pass_with_few_wavefronts(grid, monkeypatch):
    # 64 / 192 wavefronts instead of 4096 (ESIM_GRID_CHUNK, read at upload): the lane-per-citizen marks pass then needs several
    # rounds of 64 citizens per wavefront, the draw pass's equal shares of the items span many owners (192 is no power of two:
    # the share arithmetic and the two-level owner search), k_chunk_fold's lists are long, an item's records overflow.
    monkeypatch.setenv("ESIM_GRID_CHUNK", str(grid))
    pop = random_population(4242 + grid, n=9000, n_areas=9, n_buildings=500, n_schools=4, rooms_per_school=6)
    params = dict(exposure_chance=0.01, seed=99173 + grid, vaccination_rate=400, vaccination_threshold=0.3, lockdown_threshold=0.15,
                  mask_pt_threshold=0.02, mask_everywhere_threshold=0.2, bus_capacity=20, exposed_time=30, infected_time=100)
    run_both(pop, 600, check_state_every=97, small_limits=("pmap", "vax", "tp"), **params)


@pytest.mark.parametrize("repair", (2, 1, 0))
def test_bus_exposures_of_planned_citizens_repair_the_plan(repair, monkeypatch):
    # tools/fuzz_parity.py seed 2026: 700 citizens, 3 vaccinations a step, many bus exposures under the programme.  A citizen the
    # plan vaccinates later is exposed on a bus first: the plan of the steps behind is walked again (ESIM_VAX_REPAIR=2: from the
    # start; 1, the default: from the run's first cut on; 0: every such exposure cuts the chunk, as in round 2) -- same records
    # and states as the oracle in all three, and the repairs and cuts are where they should be.  (The first version of the repair
    # counted a newly chosen citizen that was exposed LATER in the chunk as Exposed when it was vaccinated: one Susceptible too many.)
    monkeypatch.setenv("ESIM_VAX_REPAIR", str(repair))
    rng = np.random.default_rng(7000 + 2026)
    pop = random_population(2026, n=int(rng.choice([300, 700, 2500])), n_areas=int(rng.choice([1, 5, 12])),
                            n_buildings=int(rng.choice([40, 90, 400])), n_schools=int(rng.choice([1, 3])), rooms_per_school=int(rng.choice([1, 4, 9])))
    params = dict(exposure_chance=0.01, seed=207753198333, vaccination_rate=3, vaccination_threshold=0.02, lockdown_threshold=0.9,
                  mask_pt_threshold=0.02, mask_everywhere_threshold=0.2, bus_capacity=64, exposed_time=96, infected_time=336, start_hour=6, end_hour=17)
    ep = _lib.default_params(**params)
    sim = Simulator(pop, ep)
    orc = _oracle.Oracle(pop, _oracle.params_from_esim(ep))
    for _ in range(4):
        assert_same_records(sim.run(125), orc.run(125))
        assert_same_state(sim, orc)
    st = sim.vax_chunk_stats()
    assert st["steps"] > 100
    if repair == 2:
        assert st["repairs"] >= 2
    elif repair == 0:
        assert st["repairs"] == 0 and st["cuts"] >= 2
    else:
        assert st["cuts"] >= 1
    sim.close()


@pytest.mark.parametrize("seed", range(6))
def test_random_populations_all_paths(seed):
    pop = random_population(seed)
    rng = np.random.default_rng(1000 + seed)
    params = dict(exposure_chance=float(rng.choice([0.002, 0.01, 0.05])), seed=int(rng.integers(1, 1 << 40)),
                  vaccination_rate=int(rng.choice([3, 25, 400])), vaccination_threshold=float(rng.choice([0.02, 0.08, 0.3])),
                  lockdown_threshold=float(rng.choice([0.05, 0.15, 0.9])), mask_pt_threshold=0.02,
                  mask_everywhere_threshold=float(rng.choice([0.04, 0.2])), bus_capacity=int(rng.choice([3, 20])),
                  exposed_time=int(rng.choice([5, 96])), infected_time=int(rng.choice([17, 336])),
                  start_hour=int(rng.choice([9, 6])), end_hour=int(rng.choice([17, 20])))
    run_both(pop, 500, check_state_every=125, **params)
