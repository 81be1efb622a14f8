//! What `SimulatorBuilder::build` (simulator_builder.rs:1162-1292) leaves behind, flattened into the structure-of-arrays that
//! `esim_upload_population` takes (include/esim.h: esim_population).  The library copies the arrays; this struct only has to
//! live across the call.
use std::collections::HashMap;

use crate::disease::DiseaseStatus;
use crate::esim_sys::*;
use crate::models::building::{Building, BuildingID, BuildingType, School};
use crate::models::citizen::{Citizen, Occupation};
use crate::models::output_area::OutputArea;

pub struct FlatPopulation {
    pub home_building: Vec<u32>,
    pub work_building: Vec<u32>,
    pub room: Vec<u32>,
    pub flags: Vec<u8>,
    pub age: Vec<u16>,
    pub occupation: Vec<u8>,
    pub building_area: Vec<u32>,
    pub building_type: Vec<u8>,
    pub room_building: Vec<u32>,
    pub seeds: Vec<u32>,
    /// first dense building index of every Output Area: dense = area_base[area] + BuildingID::building_index()
    pub area_base: Vec<u32>,
    pub n_areas: u32,
}

impl FlatPopulation {
    /// Dense index of a building: Output Areas in `output_areas` order, buildings in their `Vec` order (BuildingID, building.rs:62-67).
    pub fn dense(&self, id: &BuildingID) -> u32 {
        self.area_base[id.output_area_code().index()] + id.building_index() as u32
    }

    pub fn as_struct(&self) -> EsimPopulation {
        EsimPopulation {
            n_citizens: self.home_building.len() as u32,
            n_buildings: self.building_area.len() as u32,
            n_areas: self.n_areas,
            n_rooms: self.room_building.len() as u32,
            n_seeds: self.seeds.len() as u32,
            citizen_id_base: 0,
            n_citizens_global: self.home_building.len() as u32,
            n_shared_buildings: 0,
            n_shared_rooms: 0,
            home_building: self.home_building.as_ptr(),
            work_building: self.work_building.as_ptr(),
            room: self.room.as_ptr(),
            flags: self.flags.as_ptr(),
            age: self.age.as_ptr(),
            occupation: self.occupation.as_ptr(),
            building_area: self.building_area.as_ptr(),
            building_type: self.building_type.as_ptr(),
            room_building: self.room_building.as_ptr(),
            seeds: self.seeds.as_ptr(),
            shared_building_local: std::ptr::null(),
            shared_room_local: std::ptr::null(),
        }
    }
}

fn type_code(t: &BuildingType) -> u8 {
    match t {
        BuildingType::Household => ESIM_HOUSEHOLD,
        BuildingType::School => ESIM_SCHOOL,
        _ => ESIM_WORKPLACE, // Workplace; Restaurant / SuperMarket / Shop are never built (building.rs:46-53)
    }
}

/// Citizens are indexed by `CitizenID::global_index` (citizen.rs:52-57); `generate_citizens` hands the indexes out area by area
/// in `output_areas` order (simulator_builder.rs:177-262), so walking the areas in order visits them in ascending index.
pub fn flatten(output_areas: &[OutputArea]) -> anyhow::Result<FlatPopulation> {
    let n_citizens: usize = output_areas.iter().map(|a| a.citizens.len()).sum();
    let mut f = FlatPopulation {
        home_building: vec![0; n_citizens],
        work_building: vec![0; n_citizens],
        room: vec![ESIM_NO_ROOM; n_citizens],
        flags: vec![0; n_citizens],
        age: vec![0; n_citizens],
        occupation: vec![0; n_citizens],
        building_area: Vec::new(),
        building_type: Vec::new(),
        room_building: Vec::new(),
        seeds: Vec::new(),
        area_base: Vec::with_capacity(output_areas.len()),
        n_areas: output_areas.len() as u32,
    };
    // buildings: dense numbering, and the rooms of every School (classes first, then offices: building.rs:346-443)
    let mut room_of: HashMap<crate::models::citizen::CitizenID, u32> = HashMap::new();
    for (area_index, area) in output_areas.iter().enumerate() {
        f.area_base.push(f.building_area.len() as u32);
        for building in &area.buildings {
            let dense = f.building_area.len() as u32;
            f.building_area.push(area_index as u32);
            // needs: `fn building_type(&self) -> &BuildingType` on BuildingID (the field exists, building.rs:66)
            f.building_type.push(type_code(building.id().building_type()));
            if let Some(school) = building.as_any().downcast_ref::<School>() {
                for class in school.classes() {
                    let r = f.room_building.len() as u32;
                    f.room_building.push(dense);
                    for id in class.get_participants() {
                        room_of.insert(id, r);
                    }
                }
                for office in school.offices() {
                    let r = f.room_building.len() as u32;
                    f.room_building.push(dense);
                    for id in office {
                        room_of.insert(*id, r);
                    }
                }
            }
        }
    }
    for area in output_areas {
        for citizen in &area.citizens {
            let c = citizen.id().global_index();
            anyhow::ensure!(c < n_citizens, "citizen index {} out of range", c);
            f.home_building[c] = f.dense(&citizen.household_code);
            f.work_building[c] = f.dense(&citizen.workplace_code);
            f.room[c] = room_of.get(&citizen.id()).copied().unwrap_or(ESIM_NO_ROOM);
            f.flags[c] = (if citizen.uses_public_transport { ESIM_FLAG_USES_PUBLIC_TRANSPORT } else { 0 })
                | (if citizen.is_mask_compliant { ESIM_FLAG_MASK_COMPLIANT } else { 0 });
            f.age[c] = citizen.age;
            f.occupation[c] = occupation_code(citizen);
            // apply_initial_infections leaves the seeds Infected(0) (simulator_builder.rs:1139)
            if let DiseaseStatus::Infected(_) = citizen.disease_status {
                f.seeds.push(c as u32);
            }
        }
    }
    Ok(f)
}

/// OccupationType::get_index (citizen.rs:312-324), 9 for students, 10 for unemployed -- carried for API fidelity only.
fn occupation_code(citizen: &Citizen) -> u8 {
    match citizen.occupation() {
        Occupation::Normal { occupation } | Occupation::Essential { occupation } => occupation.get_index() as u8,
        Occupation::Student => 9,
        Occupation::Unemployed => 10,
    }
}
