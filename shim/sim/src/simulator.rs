//! `Simulator` over libesim (replaces sim/src/simulator.rs:87-152,601-644).  Same pub fields, `From<SimulatorBuilder>`,
//! `simulate`, `step`; the per-timestep work runs on the GPU, the host keeps the `OutputArea` mirror that `run` and
//! `visualisation` read (refreshed from the device on demand).
use std::collections::{HashMap, HashSet};
use std::os::raw::c_void;
use std::sync::{Mutex, RwLock};
use std::time::Instant;

use anyhow::anyhow;
use log::{debug, info};

use crate::config::{get_memory_usage, DEBUG_ITERATION_PRINT};
use crate::disease::{DiseaseModel, DiseaseStatus};
use crate::esim_sys::*;
use crate::flatten::{flatten, FlatPopulation};
use crate::interventions::InterventionThresholds;
use crate::models::citizen::CitizenID;
use crate::models::output_area::{OutputArea, OutputAreaID};
use crate::models::public_transport_route::{PublicTransport, PublicTransportID};
use crate::simulator_builder::SimulatorBuilder;
use crate::statistics::{StatisticEntry, StatisticsRecorder};

pub struct Simulator {
    pub area_code: String,
    pub output_area_lookup: HashMap<String, u32>,
    current_population: u32,
    pub output_areas: RwLock<Vec<Mutex<OutputArea>>>,
    pub citizen_output_area_lookup: RwLock<Vec<Mutex<(OutputAreaID, u32)>>>,
    pub citizens_eligible_for_vaccine: Option<HashSet<CitizenID>>,
    statistics_recorder: StatisticsRecorder,
    disease_model: DiseaseModel,
    pub public_transport: HashMap<PublicTransportID, PublicTransport>,
    /// the device context (one host thread at a time: `Simulator` was !Send before, too -- it owned a ThreadRng)
    ctx: *mut c_void,
    flat: FlatPopulation,
    mirror_step: u32,
}

fn check(rc: i32, ctx: *const c_void) -> anyhow::Result<()> {
    if rc == ESIM_OK { Ok(()) } else { Err(anyhow!("libesim error {}: {}", rc, last_error(ctx))) }
}

impl From<SimulatorBuilder> for Simulator {
    fn from(builder: SimulatorBuilder) -> Self {
        let flat = flatten(&builder.output_areas).expect("population does not fit the device layout");
        let mut p: EsimParams = unsafe { std::mem::zeroed() };
        unsafe { esim_default_params(&mut p) };
        // DiseaseModel::covid() (disease.rs:118-129) and the thresholds of interventions.rs:50-57,71-78
        let m = &builder.disease_model;
        p.exposure_chance = m.exposure_chance;
        p.mask_effectiveness = m.mask_effectiveness;
        p.exposed_time = m.exposed_time as u32;
        p.infected_time = m.infected_time as u32;
        p.vaccination_rate = m.vaccination_rate as u32;
        p.max_steps = m.max_time_step as u32;
        let th = InterventionThresholds::default();
        p.lockdown_threshold = th.lockdown();
        p.vaccination_threshold = th.vaccination_threshold();
        p.seed = rand::random::<u64>(); // the reference seeds thread_rng from the OS (simulator.rs:630); fix it for a reproducible run
        let mut ctx: *mut c_void = std::ptr::null_mut();
        check(unsafe { esim_create(&p, &mut ctx) }, std::ptr::null()).expect("esim_create");
        let pop = flat.as_struct();
        check(unsafe { esim_upload_population(ctx, &pop) }, ctx).expect("esim_upload_population"); // the library copies

        let current_population = builder.citizen_output_area_lookup.len() as u32;
        Simulator {
            area_code: builder.area_code,
            output_area_lookup: builder.output_area_lookup,
            current_population,
            output_areas: RwLock::new(builder.output_areas.into_iter().map(Mutex::new).collect()),
            citizen_output_area_lookup: RwLock::new(builder.citizen_output_area_lookup.into_iter().map(Mutex::new).collect()),
            citizens_eligible_for_vaccine: None,
            statistics_recorder: StatisticsRecorder::default(),
            disease_model: builder.disease_model,
            public_transport: Default::default(),
            ctx,
            flat,
            mirror_step: 0,
        }
    }
}

impl Simulator {
    /// simulator.rs:108-127: until the disease is gone or max_time_step; progress line every DEBUG_ITERATION_PRINT steps;
    /// statistics dump.  The steps between two progress lines are one device-resident run (esim_run with stop_when_done).
    pub fn simulate(&mut self, output_name: String) -> anyhow::Result<()> {
        let mut start_time = Instant::now();
        info!("Starting simulation with {} areas", self.output_areas.read().unwrap().len());
        let max = self.disease_model.max_time_step as u32;
        let mut done: u32 = 0;
        let mut buf = vec![EsimStepResult::default(); DEBUG_ITERATION_PRINT];
        while done < max {
            // the next progress line follows the step with index 0 (mod DEBUG_ITERATION_PRINT), simulator.rs:119
            let n = if done == 0 { 1 } else { (DEBUG_ITERATION_PRINT as u32).min(max - done) };
            let mut got: u32 = 0;
            check(unsafe { esim_run(self.ctx, n, 1, buf.as_mut_ptr(), &mut got) }, self.ctx)?;
            for r in &buf[..got as usize] {
                self.record(r)?;
            }
            done += got;
            let last = buf[(got.max(1) - 1) as usize];
            if got < n || last.disease_exists == 0 {
                debug!("{:?}", self.statistics_recorder.global_stats.last());
                break;
            }
            if (done - 1) % DEBUG_ITERATION_PRINT as u32 == 0 {
                println!("Completed {: >3} time steps, in: {: >6} seconds  Statistics: {:?},   Memory usage: {}", DEBUG_ITERATION_PRINT,
                         format!("{:.2}", start_time.elapsed().as_secs_f64()), self.statistics_recorder.global_stats.last().expect("No data recorded!"),
                         get_memory_usage()?);
                start_time = Instant::now();
            }
        }
        self.flush_exposures()?;
        self.statistics_recorder.dump_to_file(output_name);
        Ok(())
    }

    /// simulator.rs:131-152.  Returns false when the disease has finished.
    pub fn step(&mut self) -> anyhow::Result<bool> {
        let mut r = EsimStepResult::default();
        check(unsafe { esim_step(self.ctx, &mut r) }, self.ctx)?; // never unwinds across the boundary
        self.record(&r)?;
        Ok(r.disease_exists != 0)
    }

    /// One StatisticEntry (statistics.rs:208-215) from a step result: next() + the census as update_global_stats_entry leaves it
    /// after this step's citizen_exposed calls (statistics.rs:156-171,275-287).
    fn record(&mut self, r: &EsimStepResult) -> anyhow::Result<()> {
        self.statistics_recorder.next()?;
        self.statistics_recorder.update_global_stats_entry(StatisticEntry::from_counts(
            r.time_step, r.susceptible, r.exposed, r.infected, r.recovered, r.vaccinated)); // needs: a 6-argument constructor next to with_time_step (statistics.rs:218)
        Ok(())
    }

    /// exposures.json's per-Output-Area series (statistics.rs:119-136) from the device's exposure log: a building exposure
    /// is credited to the Output Area the citizen stands in at that step -- the building's (simulator.rs:324).
    fn flush_exposures(&mut self) -> anyhow::Result<()> {
        let mut n: u32 = 0;
        unsafe { esim_download_exposure_log(self.ctx, std::ptr::null_mut(), std::ptr::null_mut(), std::ptr::null_mut(), 0, &mut n) };
        let (mut cit, mut step, mut bus) = (vec![0u32; n as usize], vec![0u32; n as usize], vec![0u8; n as usize]);
        check(unsafe { esim_download_exposure_log(self.ctx, cit.as_mut_ptr(), step.as_mut_ptr(), bus.as_mut_ptr(), n, &mut n) }, self.ctx)?;
        let areas = self.output_areas.read().unwrap();
        for i in 0..n as usize {
            if bus[i] != 0 { continue; }
            let c = cit[i] as usize;
            // where the citizen stood in that step: at work between start and end hour unless a lockdown froze it -- the
            // library's records carry `lockdown`; the simple case (no lockdown in force) is hour-of-day only
            let building = if self.was_at_work(step[i]) && self.flat.work_building[c] != self.flat.home_building[c] { self.flat.work_building[c] } else { self.flat.home_building[c] };
            let area = areas[self.flat.building_area[building as usize] as usize].lock().unwrap().id();
            self.statistics_recorder.add_area_exposure(area, step[i]); // needs: the per-time-step push of statistics.rs:160-164 by (area, step)
        }
        Ok(())
    }

    fn was_at_work(&self, step: u32) -> bool {
        // citizen.rs:176-206 with start/end_working_hour 9/17 for everybody (citizen.rs:154-155); steps under lockdown keep the
        // position of the step before (Q8) -- global_stats' records say which steps those were
        let mut at_work = false;
        for t in 1..=step {
            if self.statistics_recorder.lockdown_during(t) { continue; } // needs: keep EsimStepResult::lockdown of step t-1 per step
            match t % 24 { 9 => at_work = true, 17 => at_work = false, _ => {} }
        }
        at_work
    }

    /// Brings the host mirror (`output_areas[..].citizens[..].disease_status / current_building_position`) up to the device's
    /// state; `run --visualise*` and `visualisation::citizen_connections` read it (run/src/main.rs:246-259).
    pub fn refresh_mirror(&mut self) -> anyhow::Result<()> {
        let n = self.flat.home_building.len();
        let (mut status, mut timer, mut cur) = (vec![0u8; n], vec![0u16; n], vec![0u32; n]);
        check(unsafe { esim_download_state(self.ctx, status.as_mut_ptr(), timer.as_mut_ptr(), cur.as_mut_ptr(), std::ptr::null_mut(), std::ptr::null_mut()) }, self.ctx)?;
        let mut areas = self.output_areas.write().unwrap();
        for area in areas.iter_mut() {
            let mut area = area.lock().unwrap();
            for citizen in area.citizens.iter_mut() {
                let c = citizen.id().global_index();
                citizen.disease_status = match status[c] {
                    ESIM_SUSCEPTIBLE => DiseaseStatus::Susceptible,
                    ESIM_EXPOSED => DiseaseStatus::Exposed(timer[c]),
                    ESIM_INFECTED => DiseaseStatus::Infected(timer[c]),
                    ESIM_RECOVERED => DiseaseStatus::Recovered,
                    _ => DiseaseStatus::Vaccinated,
                };
                citizen.current_building_position = if cur[c] == self.flat.work_building[c] { citizen.workplace_code.clone() } else { citizen.household_code.clone() };
            }
        }
        Ok(())
    }
}

impl Drop for Simulator {
    fn drop(&mut self) {
        unsafe { esim_destroy(self.ctx) }
    }
}
