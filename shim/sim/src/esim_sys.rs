//! `extern "C"` view of include/esim.h (libesim.so): the MI355X implementation of the per-timestep Citizen update loop.
//! Layouts are checked against the C header by tests/test_abi.py (sizes) on the Python side; keep the three in step.
#![allow(non_camel_case_types, dead_code)]
use std::os::raw::{c_char, c_int, c_void};

pub const ESIM_OK: c_int = 0;
pub const ESIM_NO_ROOM: u32 = 0xFFFF_FFFF;
pub const ESIM_FLAG_USES_PUBLIC_TRANSPORT: u8 = 1;
pub const ESIM_FLAG_MASK_COMPLIANT: u8 = 2;
pub const ESIM_HOUSEHOLD: u8 = 0;
pub const ESIM_WORKPLACE: u8 = 1;
pub const ESIM_SCHOOL: u8 = 2;
/// DiseaseStatus codes of esim_download_state
pub const ESIM_SUSCEPTIBLE: u8 = 0;
pub const ESIM_EXPOSED: u8 = 1;
pub const ESIM_INFECTED: u8 = 2;
pub const ESIM_RECOVERED: u8 = 3;
pub const ESIM_VACCINATED: u8 = 4;

#[repr(C)]
#[derive(Clone, Copy)]
pub struct EsimParams {
    pub exposure_chance: f64,
    pub mask_effectiveness: f64,
    pub lockdown_threshold: f64,
    pub vaccination_threshold: f64,
    pub mask_pt_threshold: f64,
    pub mask_everywhere_threshold: f64,
    pub exposed_time: u32,
    pub infected_time: u32,
    pub vaccination_rate: u32,
    pub bus_capacity: u32,
    pub start_hour: u32,
    pub end_hour: u32,
    pub seed: u64,
    pub device: i32,
    pub max_steps: u32,
}

#[repr(C)]
pub struct EsimPopulation {
    pub n_citizens: u32,
    pub n_buildings: u32,
    pub n_areas: u32,
    pub n_rooms: u32,
    pub n_seeds: u32,
    pub citizen_id_base: u32,
    pub n_citizens_global: u32,
    pub n_shared_buildings: u32,
    pub n_shared_rooms: u32,
    pub home_building: *const u32,
    pub work_building: *const u32,
    pub room: *const u32,
    pub flags: *const u8,
    pub age: *const u16,
    pub occupation: *const u8,
    pub building_area: *const u32,
    pub building_type: *const u8,
    pub room_building: *const u32,
    pub seeds: *const u32,
    pub shared_building_local: *const i32,
    pub shared_room_local: *const i32,
}

#[repr(C)]
#[derive(Default, Clone, Copy, Debug)]
pub struct EsimStepResult {
    pub time_step: u32,
    pub susceptible: u32,
    pub exposed: u32,
    pub infected: u32,
    pub recovered: u32,
    pub vaccinated: u32,
    pub exposures_building: u32,
    pub exposures_bus: u32,
    pub lockdown: u32,
    pub vaccination_active: u32,
    pub mask_status: u32,
    pub n_riders: u32,
    pub vaccinated_now: u32,
    pub eligible_count: u32,
    pub disease_exists: u32,
    pub reserved: u32,
}

/// esim_allreduce_fn: a caller-supplied SUM all-reduce for transports other than RCCL
pub type EsimAllreduceFn = extern "C" fn(user: *mut c_void, which: c_int, host_ptr: *mut c_void, n_u32: usize) -> c_int;

#[link(name = "esim")]
extern "C" {
    pub fn esim_default_params(p: *mut EsimParams);
    pub fn esim_create(p: *const EsimParams, out: *mut *mut c_void) -> c_int;
    pub fn esim_upload_population(ctx: *mut c_void, pop: *const EsimPopulation) -> c_int;
    pub fn esim_reset(ctx: *mut c_void) -> c_int;
    pub fn esim_step(ctx: *mut c_void, out: *mut EsimStepResult) -> c_int;
    pub fn esim_run(ctx: *mut c_void, n_steps: u32, stop_when_done: c_int, out: *mut EsimStepResult, n_done: *mut u32) -> c_int;
    pub fn esim_read_records(ctx: *mut c_void, first_step: u32, n: u32, out: *mut EsimStepResult) -> c_int;
    pub fn esim_download_state(ctx: *mut c_void, status: *mut u8, timer: *mut u16, current_building: *mut u32,
                               on_bus: *mut u8, eligible: *mut u8) -> c_int;
    pub fn esim_download_exposure_log(ctx: *mut c_void, citizen: *mut u32, step: *mut u32, on_bus: *mut u8, cap: u32, n_out: *mut u32) -> c_int;
    pub fn esim_checkpoint_size(ctx: *mut c_void, bytes: *mut usize) -> c_int;
    pub fn esim_checkpoint_save(ctx: *mut c_void, buf: *mut c_void, cap: usize) -> c_int;
    pub fn esim_checkpoint_restore(ctx: *mut c_void, buf: *const c_void, bytes: usize) -> c_int;
    pub fn esim_enable_phase_timing(ctx: *mut c_void, enable: c_int) -> c_int;
    pub fn esim_phase_timings(ctx: *mut c_void, out: *mut f64) -> c_int;
    // multi-GPU: one context per GPU and process; the exchange between the shards is the library's
    pub fn esim_shard_population(whole: *const EsimPopulation, cuts: *const u32, n_shards: u32, shard: u32, out: *mut EsimPopulation) -> c_int;
    pub fn esim_shard_cuts(whole: *const EsimPopulation, n_shards: u32, by_work: c_int, cuts_out: *mut u32) -> c_int;
    pub fn esim_synth_free(pop: *mut EsimPopulation);
    pub fn esim_comm_unique_id(out: *mut c_void, cap: usize) -> c_int;
    pub fn esim_comm_init_rccl(ctx: *mut c_void, unique_id: *const c_void, id_bytes: usize, rank: c_int, world: c_int) -> c_int;
    pub fn esim_comm_init_callback(ctx: *mut c_void, f: EsimAllreduceFn, user: *mut c_void, rank: c_int, world: c_int) -> c_int;
    // (set-up comes after esim_upload_population and checks the ranks' shards against each other; every rank returns from
    // esim_run_sharded together and with the same code; ESIM_ETIMEDOUT (-7) when a peer left -- exit with an error then)
    pub fn esim_comm_set_timeout(ctx: *mut c_void, seconds: f64) -> c_int;
    pub fn esim_run_sharded(ctx: *mut c_void, n_steps: u32, n_done: *mut u32) -> c_int;
    pub fn esim_last_error(ctx: *const c_void) -> *const c_char;
    pub fn esim_destroy(ctx: *mut c_void);
}

/// The library's error text for `ctx` (or for the last failed esim_create when null).
pub fn last_error(ctx: *const c_void) -> String {
    unsafe {
        let p = esim_last_error(ctx);
        if p.is_null() { String::new() } else { std::ffi::CStr::from_ptr(p).to_string_lossy().into_owned() }
    }
}
