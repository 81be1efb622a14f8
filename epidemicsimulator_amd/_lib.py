"""ctypes binding of libesim.so (C ABI: include/esim.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C epidemicsimulator_amd/csrc``.
There is no Python or CPU fallback: if the shared object is missing, or no HIP device is
usable, the calls fail loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ESIM_LIB") or os.path.join(_HERE, "libesim.so")   # ESIM_LIB: a diagnostics build of the same library

OK = 0
ERRORS = {-1: "EINVAL", -2: "ENODEVICE", -3: "ENOMEM", -4: "ESTATE", -5: "ERANGE", -6: "ESIM", -7: "ETIMEDOUT"}

SUSCEPTIBLE, EXPOSED, INFECTED, RECOVERED, VACCINATED = range(5)
HOUSEHOLD, WORKPLACE, SCHOOL = range(3)
MASK_NONE, MASK_PUBLIC_TRANSPORT, MASK_EVERYWHERE = range(3)
NO_ROOM = 0xFFFFFFFF
FLAG_USES_PUBLIC_TRANSPORT = 1
FLAG_MASK_COMPLIANT = 2


class EsimError(RuntimeError):
    def __init__(self, code, text):
        super().__init__("libesim %s (%d): %s" % (ERRORS.get(code, "?"), code, text))
        self.code = code


class Params(C.Structure):
    _fields_ = [
        ("exposure_chance", C.c_double), ("mask_effectiveness", C.c_double),
        ("lockdown_threshold", C.c_double), ("vaccination_threshold", C.c_double),
        ("mask_pt_threshold", C.c_double), ("mask_everywhere_threshold", C.c_double),
        ("exposed_time", C.c_uint32), ("infected_time", C.c_uint32),
        ("vaccination_rate", C.c_uint32), ("bus_capacity", C.c_uint32),
        ("start_hour", C.c_uint32), ("end_hour", C.c_uint32),
        ("seed", C.c_uint64), ("device", C.c_int32), ("max_steps", C.c_uint32),
    ]


_u32p, _u16p, _u8p, _i32p = (C.POINTER(C.c_uint32), C.POINTER(C.c_uint16),
                             C.POINTER(C.c_uint8), C.POINTER(C.c_int32))


class PopulationStruct(C.Structure):
    _fields_ = [
        ("n_citizens", C.c_uint32), ("n_buildings", C.c_uint32), ("n_areas", C.c_uint32),
        ("n_rooms", C.c_uint32), ("n_seeds", C.c_uint32),
        ("citizen_id_base", C.c_uint32), ("n_citizens_global", C.c_uint32),
        ("n_shared_buildings", C.c_uint32), ("n_shared_rooms", C.c_uint32),
        ("home_building", _u32p), ("work_building", _u32p), ("room", _u32p),
        ("flags", _u8p), ("age", _u16p), ("occupation", _u8p),
        ("building_area", _u32p), ("building_type", _u8p), ("room_building", _u32p),
        ("seeds", _u32p), ("shared_building_local", _i32p), ("shared_room_local", _i32p),
    ]


class StepResult(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in (
        "time_step", "susceptible", "exposed", "infected", "recovered", "vaccinated",
        "exposures_building", "exposures_bus", "lockdown", "vaccination_active", "mask_status",
        "n_riders", "vaccinated_now", "eligible_count", "disease_exists", "reserved")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_ if n != "reserved"}


RECORD_FIELDS = [n for n, _ in StepResult._fields_]


class SynthSpec(C.Structure):
    _fields_ = [
        ("n_citizens", C.c_uint32), ("n_areas", C.c_uint32), ("citizens_per_school", C.c_uint32),
        ("n_seeds", C.c_uint32), ("seed", C.c_uint64), ("area_jitter", C.c_double),
        ("p_public_transport", C.c_double), ("p_mask_compliant", C.c_double),
        ("p_work_from_home", C.c_double), ("p_teaching", C.c_double),
        ("household_buildings_median", C.c_double), ("household_buildings_sigma", C.c_double),
        ("p_area_without_households", C.c_double),
        ("workplace_buildings_median", C.c_double), ("workplace_buildings_sigma", C.c_double),
        ("p_area_without_workplaces", C.c_double),
        ("workplace_floor_median", C.c_double), ("workplace_floor_sigma", C.c_double),
        ("teacher_candidate_schools", C.c_uint32), ("reserved", C.c_uint32),
    ]


# every symbol include/esim.h declares (tests/test_abi.py checks the library exports them all)
SYMBOLS = [
    "esim_default_params", "esim_create", "esim_upload_population", "esim_reset", "esim_step",
    "esim_run", "esim_step_begin", "esim_step_exposures", "esim_step_finish",
    "esim_exchange_buffer", "esim_future_infected", "esim_run_free", "esim_free_begin", "esim_free_enqueue", "esim_free_collect", "esim_set_pipeline", "esim_chunk_timing", "esim_enable_chunk_kernel_timing", "esim_chunk_kernel_timings", "esim_vax_chunk_stats", "esim_vax_repair_stats", "esim_pipeline_timing",
    "esim_comm_unique_id", "esim_comm_init_rccl", "esim_comm_init_callback", "esim_comm_set_timeout", "esim_debug_inject_error", "esim_comm_stats", "esim_run_sharded", "esim_shard_stats",
    "esim_read_records", "esim_stream", "esim_set_stream",
    "esim_set_exchange_buffer", "esim_synchronize",
    "esim_download_state", "esim_download_exposure_log", "esim_checkpoint_size", "esim_checkpoint_save", "esim_checkpoint_restore", "esim_enable_phase_timing", "esim_phase_timings",
    "esim_enable_kernel_timing", "esim_kernel_timings", "esim_set_small_step_limit", "esim_set_tiny_chunk_limit", "esim_small_kernel_timing", "esim_debug_counters",
    "esim_last_error", "esim_destroy",
    "esim_threshold_lut", "esim_synth_preset", "esim_synth_create", "esim_synth_create_shard", "esim_synth_free",
    "esim_shard_population", "esim_shard_cuts",
]

# esim_allreduce_fn: int (*)(void *user, int which, void *device_ptr, size_t n_u32)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t)

_lib = None


def load():
    """Loads libesim.so (once).  Raises if it has not been built -- no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libesim.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C epidemicsimulator_amd/csrc`. There is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    vp, pvp = C.c_void_p, C.POINTER(C.c_void_p)
    sig = {
        "esim_default_params": (None, [C.POINTER(Params)]),
        "esim_create": (C.c_int, [C.POINTER(Params), pvp]),
        "esim_upload_population": (C.c_int, [vp, C.POINTER(PopulationStruct)]),
        "esim_reset": (C.c_int, [vp]),
        "esim_step": (C.c_int, [vp, C.POINTER(StepResult)]),
        "esim_run": (C.c_int, [vp, C.c_uint32, C.c_int, C.POINTER(StepResult), C.POINTER(C.c_uint32)]),
        "esim_step_begin": (C.c_int, [vp]),
        "esim_step_exposures": (C.c_int, [vp]),
        "esim_step_finish": (C.c_int, [vp, C.POINTER(StepResult)]),
        "esim_exchange_buffer": (C.c_int, [vp, C.c_int, pvp, C.POINTER(C.c_size_t)]),
        "esim_comm_unique_id": (C.c_int, [vp, C.c_size_t]),
        "esim_comm_init_rccl": (C.c_int, [vp, vp, C.c_size_t, C.c_int, C.c_int]),
        "esim_comm_init_callback": (C.c_int, [vp, ALLREDUCE_FN, vp, C.c_int, C.c_int]),
        "esim_comm_set_timeout": (C.c_int, [vp, C.c_double]),
        "esim_debug_inject_error": (C.c_int, [vp, C.c_int]),
        "esim_comm_stats": (C.c_int, [vp, C.POINTER(C.c_uint64)]),
        "esim_run_sharded": (C.c_int, [vp, C.c_uint32, C.POINTER(C.c_uint32)]),
        "esim_shard_stats": (C.c_int, [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
        "esim_future_infected": (C.c_int, [vp]),
        "esim_run_free": (C.c_int, [vp, C.c_uint32, C.POINTER(C.c_uint32)]),
        "esim_free_begin": (C.c_int, [vp, C.c_uint32]),
        "esim_free_enqueue": (C.c_int, [vp]),
        "esim_free_collect": (C.c_int, [vp, C.POINTER(C.c_uint32)]),
        "esim_set_pipeline": (C.c_int, [vp, C.c_int]),
        "esim_chunk_timing": (C.c_int, [vp, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
        "esim_enable_chunk_kernel_timing": (C.c_int, [vp, C.c_int]),
        "esim_chunk_kernel_timings": (C.c_int, [vp, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]),
        "esim_vax_chunk_stats": (C.c_int, [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
        "esim_vax_repair_stats": (C.c_int, [vp, C.POINTER(C.c_uint64)]),
        "esim_pipeline_timing": (C.c_int, [vp, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
        "esim_read_records": (C.c_int, [vp, C.c_uint32, C.c_uint32, C.POINTER(StepResult)]),
        "esim_stream": (C.c_int, [vp, pvp]),
        "esim_set_stream": (C.c_int, [vp, vp]),
        "esim_set_exchange_buffer": (C.c_int, [vp, C.c_int, vp]),
        "esim_synchronize": (C.c_int, [vp]),
        "esim_download_state": (C.c_int, [vp, _u8p, _u16p, _u32p, _u8p, _u8p]),
        "esim_download_exposure_log": (C.c_int, [vp, _u32p, _u32p, _u8p, C.c_uint32, _u32p]),
        "esim_checkpoint_size": (C.c_int, [vp, C.POINTER(C.c_size_t)]),
        "esim_checkpoint_save": (C.c_int, [vp, vp, C.c_size_t]),
        "esim_checkpoint_restore": (C.c_int, [vp, vp, C.c_size_t]),
        "esim_enable_phase_timing": (C.c_int, [vp, C.c_int]),
        "esim_phase_timings": (C.c_int, [vp, C.POINTER(C.c_double)]),
        "esim_enable_kernel_timing": (C.c_int, [vp, C.c_int]),
        "esim_kernel_timings": (C.c_int, [vp, C.POINTER(C.c_double), C.POINTER(C.c_uint32)]),
        "esim_debug_counters": (C.c_int, [vp, C.POINTER(C.c_uint32)]),
        "esim_set_small_step_limit": (C.c_int, [vp, C.c_uint32]),
        "esim_set_tiny_chunk_limit": (C.c_int, [vp, C.c_uint32]),
        "esim_small_kernel_timing": (C.c_int, [vp, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]),
        "esim_last_error": (C.c_char_p, [vp]),
        "esim_destroy": (None, [vp]),
        "esim_threshold_lut": (C.c_int, [C.POINTER(Params), C.POINTER(C.c_uint64)]),
        "esim_synth_preset": (C.c_int, [C.c_char_p, C.POINTER(SynthSpec)]),
        "esim_synth_create": (C.c_int, [C.POINTER(SynthSpec), C.POINTER(PopulationStruct)]),
        "esim_synth_create_shard": (C.c_int, [C.POINTER(SynthSpec), C.c_uint32, C.c_uint32, C.POINTER(PopulationStruct)]),
        "esim_shard_cuts": (C.c_int, [C.POINTER(PopulationStruct), C.c_uint32, C.c_int, C.POINTER(C.c_uint32)]),
        "esim_synth_free": (None, [C.POINTER(PopulationStruct)]),
        "esim_shard_population": (C.c_int, [C.POINTER(PopulationStruct), _u32p, C.c_uint32, C.c_uint32,
                                            C.POINTER(PopulationStruct)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


ESIM_OK, ESIM_ERANGE = 0, -5          # include/esim.h
CHUNK_KERNELS = ("marks", "fold", "draw", "units", "count", "books", "scatter", "vax", "vax_adj", "vax_final", "decide", "future", "map_clear", "tiny", "vax_repair")   # ESIM_CK_*
PHASE_OF_KERNEL = {"marks": "Generate Exposures", "fold": "Generate Exposures", "draw": "Apply Exposures", "units": "Apply Exposures"}   # the rest: "Apply Interventions"
TINY_PHASE_SHARES = {"Generate Exposures": 0.10, "Apply Exposures": 0.40, "Apply Interventions": 0.50}   # k_chunk_tiny: all three in one launch


def check(code, ctx=None):
    if code != OK:
        text = load().esim_last_error(ctx)
        raise EsimError(code, text.decode() if text else "")


def default_params(**overrides):
    p = Params()
    load().esim_default_params(C.byref(p))
    for k, v in overrides.items():
        if not hasattr(p, k):
            raise AttributeError("esim_params has no field %r" % k)
        setattr(p, k, v)
    return p
