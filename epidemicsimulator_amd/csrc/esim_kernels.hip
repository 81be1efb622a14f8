// esim_kernels.hip -- gfx950 kernels of the per-timestep Citizen update loop.
//
// One time step (reference: Simulator::step, sim/src/simulator.rs:131-152) is, on the device:
//   k_infected  generate_exposures (simulator.rs:155-260): every currently Infected citizen that is
//               not on a bus marks the building (and school room) it stands in; riders mark their route
//   k_expose    apply_exposures (simulator.rs:262-405): for every marked building / room / route,
//               walk its registered members exactly as Building::find_exposures does and draw
//   k_finish    apply_interventions (simulator.rs:455-556), the census and the StatisticEntry
//
// While no vaccination programme runs, everything a step needs from the past except who is still Susceptible is known
// up to exposed_time steps ahead (who is Infected, where everybody stands, the intervention decisions).  Then a chunk
// of <= 96 steps is drawn in ONE pass (esim_kernels_chunk.h: a citizen's exposure step is the earliest step at which
// any of its draws succeeds -- one atomicMin on its word), or, when it does not fit that form, as one k_pipe launch
// per step with the books written once per chunk.
//
// Work-efficient by construction:
//  * DiseaseStatus is a function of (step - exposure step), so the census of simulator.rs:178 is a
//    sliding-window sum over a histogram of exposure steps, not a pass over citizens;
//  * citizens are appended to an exposure log when they become Exposed(0), so "everyone Infected in
//    step t" is one contiguous slice of that log;
//  * where citizens stand is global (same working hours for everybody, citizen.rs:154-155).
// All draws are Philox4x32-10 keyed (global citizen, step, slot); probabilities are integer
// thresholds ceil(q*2^32) from a host-built LUT, so every comparison is exact integer work.
#include "esim_kernels_common.h"
#include "esim_kernels_step.h"
#include "esim_kernels_chunk.h"
#include "esim_kernels_tiny.h"
#include "esim_kernels_state.h"
