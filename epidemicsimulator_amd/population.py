"""Populations in the layout `esim_upload_population` takes (include/esim.h: esim_population).

This is the hand-over point from the reference's `SimulatorBuilder::build`
(sim/src/simulator_builder.rs:1162-1292): whatever builds the population (the reference's
census/OSM pipeline, or the seeded synthetic generator here) ends up as these arrays.
"""
import ctypes as C

import numpy as np

from . import _lib

PRESETS = ("york", "yh_census", "syn3m5", "uk64m")

_ARRAYS = (
    ("home_building", np.uint32, "n_citizens"), ("work_building", np.uint32, "n_citizens"),
    ("room", np.uint32, "n_citizens"), ("flags", np.uint8, "n_citizens"),
    ("age", np.uint16, "n_citizens"), ("occupation", np.uint8, "n_citizens"),
    ("building_area", np.uint32, "n_buildings"), ("building_type", np.uint8, "n_buildings"),
    ("room_building", np.uint32, "n_rooms"), ("seeds", np.uint32, "n_seeds"),
    ("shared_building_local", np.int32, "n_shared_buildings"),
    ("shared_room_local", np.int32, "n_shared_rooms"),
)
_SCALARS = ("n_citizens", "n_buildings", "n_areas", "n_rooms", "n_seeds", "citizen_id_base",
            "n_citizens_global", "n_shared_buildings", "n_shared_rooms")


class Population:
    """Structure-of-arrays population; numpy arrays own the memory."""

    def __init__(self, **kw):
        for name, dtype, _ in _ARRAYS:
            a = kw.get(name)
            a = np.zeros(0, dtype) if a is None else np.ascontiguousarray(a, dtype=dtype)
            setattr(self, name, a)
        self.n_citizens = int(self.home_building.size)
        self.n_buildings = int(self.building_area.size)
        self.n_rooms = int(self.room_building.size)
        self.n_seeds = int(self.seeds.size)
        self.n_areas = int(kw.get("n_areas", (int(self.building_area.max()) + 1) if self.n_buildings else 0))
        self.citizen_id_base = int(kw.get("citizen_id_base", 0))
        self.n_citizens_global = int(kw.get("n_citizens_global", self.n_citizens))
        self.n_shared_buildings = int(self.shared_building_local.size)
        self.n_shared_rooms = int(self.shared_room_local.size)
        if self.room.size == 0 and self.n_citizens:
            self.room = np.full(self.n_citizens, _lib.NO_ROOM, np.uint32)
        if self.age.size == 0:
            self.age = np.zeros(self.n_citizens, np.uint16)
        if self.occupation.size == 0:
            self.occupation = np.zeros(self.n_citizens, np.uint8)

    # -- C view ------------------------------------------------------------------------
    def as_struct(self):
        s = _lib.PopulationStruct()
        for name in _SCALARS:
            setattr(s, name, getattr(self, name))
        for name, dtype, _ in _ARRAYS:
            a = getattr(self, name)
            ptr_t = dict(s._fields_)[name]
            setattr(s, name, a.ctypes.data_as(ptr_t) if a.size else ptr_t())
        return s

    @classmethod
    def _from_struct(cls, s):
        kw = {}
        for name, dtype, count in _ARRAYS:
            n = int(getattr(s, count))
            p = getattr(s, name)
            kw[name] = np.ctypeslib.as_array(p, shape=(n,)).astype(dtype, copy=True) if n and p else None
        pop = cls(n_areas=int(s.n_areas), citizen_id_base=int(s.citizen_id_base),
                  n_citizens_global=int(s.n_citizens_global), **kw)
        return pop

    # -- builders ----------------------------------------------------------------------
    @classmethod
    def synthetic(cls, preset=None, **spec_overrides):
        """Seeded synthetic population. `preset` in PRESETS, or give
        n_citizens/n_areas/citizens_per_school explicitly."""
        lib = _lib.load()
        spec = _lib.SynthSpec()
        _lib.check(lib.esim_synth_preset((preset or "york").encode(), C.byref(spec)))
        for k, v in spec_overrides.items():
            if not hasattr(spec, k):
                raise AttributeError("esim_synth_spec has no field %r" % k)
            setattr(spec, k, v)
        s = _lib.PopulationStruct()
        _lib.check(lib.esim_synth_create(C.byref(spec), C.byref(s)))
        try:
            return cls._from_struct(s)
        finally:
            lib.esim_synth_free(C.byref(s))

    @classmethod
    def synthetic_shard(cls, shard, n_shards, preset=None, **spec_overrides):
        """Shard `shard` of `n_shards` of the synthetic world, generated directly (esim_synth_create_shard):
        a run of whole school catchments, so it shares no building with the other shards."""
        lib = _lib.load()
        spec = _lib.SynthSpec()
        _lib.check(lib.esim_synth_preset((preset or "york").encode(), C.byref(spec)))
        for k, v in spec_overrides.items():
            if not hasattr(spec, k):
                raise AttributeError("esim_synth_spec has no field %r" % k)
            setattr(spec, k, v)
        s = _lib.PopulationStruct()
        _lib.check(lib.esim_synth_create_shard(C.byref(spec), shard, n_shards, C.byref(s)))
        try:
            return cls._from_struct(s)
        finally:
            lib.esim_synth_free(C.byref(s))

    def shard(self, cuts, index):
        """The shard of Output Areas [cuts[index], cuts[index+1]) (esim_shard_population)."""
        lib = _lib.load()
        cuts = np.ascontiguousarray(cuts, np.uint32)
        whole = self.as_struct()
        out = _lib.PopulationStruct()
        _lib.check(lib.esim_shard_population(C.byref(whole), cuts.ctypes.data_as(C.POINTER(C.c_uint32)),
                                             len(cuts) - 1, index, C.byref(out)))
        try:
            return Population._from_struct(out)
        finally:
            lib.esim_synth_free(C.byref(out))

    def work_cuts(self, n_shards):
        """Area boundaries giving each shard about the same expected work (esim_shard_cuts: a citizen weighs 1 + the sizes of
        the lists it is a member of -- household, work place, class room; what esim_synth_create_shard cuts by)."""
        lib = _lib.load()
        whole = self.as_struct()
        cuts = np.zeros(n_shards + 1, np.uint32)
        _lib.check(lib.esim_shard_cuts(C.byref(whole), n_shards, 1, cuts.ctypes.data_as(C.POINTER(C.c_uint32))))
        return cuts

    def even_cuts(self, n_shards):
        """Area boundaries giving each shard about the same number of citizens."""
        area_of_citizen = self.building_area[self.home_building]
        per_area = np.bincount(area_of_citizen, minlength=self.n_areas)
        cum = np.concatenate([[0], np.cumsum(per_area)])
        cuts = [0]
        for k in range(1, n_shards):
            cuts.append(int(np.searchsorted(cum, self.n_citizens * k / n_shards)))
        cuts.append(self.n_areas)
        return np.maximum.accumulate(np.asarray(cuts, np.uint32))
